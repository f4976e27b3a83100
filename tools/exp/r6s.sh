# round 6, GPU call s: long soak on the final tree + the concurrency and multi-rank tests five times over (one process each time).
mkdir -p gpurun_out/r6s && cd /root/repo
timeout -k 10 900 python tools/soak.py 6000 1500 > gpurun_out/r6s/soak_final.log 2>&1; echo "soak rc=$?"; tail -1 gpurun_out/r6s/soak_final.log
python -m pytest tests/test_gpu_round6.py tests/test_gpu_multirank_stand_in.py tests/test_gpu_robustness.py --count 1 -x -q -p no:cacheprovider > gpurun_out/r6s/repeat_0.log 2>&1 || true
for i in 1 2 3 4 5; do python -m pytest tests/test_gpu_round6.py tests/test_gpu_multirank_stand_in.py tests/test_gpu_robustness.py -x -q > gpurun_out/r6s/repeat_$i.log 2>&1; echo "repeat $i rc=$? $(tail -1 gpurun_out/r6s/repeat_$i.log)"; done
