mkdir -p gpurun_out/r6b && cd /root/repo
python -m pytest tests/test_gpu_round6.py tests/test_gpu_long_rows.py tests/test_gpu_direct_blocks.py tests/test_gpu_fuzz.py tests/test_gpu_batch_fastest.py tests/test_gpu_user_path.py tests/test_gpu_robustness.py -x -q > gpurun_out/r6b/tests.log 2>&1
echo "tests rc=$?" ; tail -15 gpurun_out/r6b/tests.log
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; numactl -H 2>/dev/null | head -20 > gpurun_out/r6b/numa.txt; lscpu | grep -i "numa\|model name\|socket" >> gpurun_out/r6b/numa.txt
for knobs in "" "host_pack_stores=1" "host_staging_numa=1" "host_pack_stores=1 host_staging_numa=1"; do
  tag=$(echo "$knobs" | tr ' =' '__'); [ -z "$tag" ] && tag=default
  timeout -k 10 300 python tools/user_path_bench.py 512 --only-h2h $knobs > gpurun_out/r6b/h2h_$tag.json 2> gpurun_out/r6b/h2h_$tag.err || echo "h2h $tag failed"
done
ls -la gpurun_out/r6b
