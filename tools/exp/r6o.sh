mkdir -p gpurun_out/r6o && cd /root/repo
python -m pytest tests/test_gpu_round6.py tests/test_gpu_fuzz.py tests/test_gpu_robustness.py tests/test_gpu_facade.py -x -q > gpurun_out/r6o/tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r6o/tests.log
for nt in 40 120; do timeout -k 10 500 python tools/host_pipeline_bench_levels.py $nt > gpurun_out/r6o/levels_$nt.json 2> gpurun_out/r6o/levels_$nt.err; echo "levels $nt rc=$?"; cat gpurun_out/r6o/levels_$nt.json; done
