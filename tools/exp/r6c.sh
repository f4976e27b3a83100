mkdir -p gpurun_out/r6c && cd /root/repo
python -m pytest tests/test_gpu_round6.py tests/test_gpu_long_rows.py tests/test_gpu_direct_blocks.py tests/test_gpu_fuzz.py tests/test_gpu_batch_fastest.py tests/test_gpu_user_path.py tests/test_gpu_robustness.py -x -q > gpurun_out/r6c/tests.log 2>&1
echo "tests rc=$?" ; tail -5 gpurun_out/r6c/tests.log
for knobs in "" "host_pack_stores=1" ""; do
  tag=$(echo "$knobs" | tr ' =' '__'); [ -z "$tag" ] && tag=default
  n=$(ls gpurun_out/r6c/h2h_${tag}_*.json 2>/dev/null | wc -l)
  timeout -k 10 300 python tools/user_path_bench.py 512 --only-h2h $knobs > gpurun_out/r6c/h2h_${tag}_$n.json 2> gpurun_out/r6c/h2h_${tag}_$n.err || echo "h2h $tag failed"
done
bash tools/exp/cfg4_placement.sh gpurun_out/r6c/placement > gpurun_out/r6c/placement.txt 2>&1
tail -70 gpurun_out/r6c/placement.txt
