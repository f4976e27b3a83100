# round 6, GPU call w (final tree): the whole GPU suite, then the driver's command plain and under the kernel trace.
mkdir -p gpurun_out/r6w && cd /root/repo
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/r6w/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r6w/tests.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6w/r06_driver_cmd_bench.json 2> gpurun_out/r6w/bench.err; echo "bench rc=$?"
rm -rf /tmp/prof_driver
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_driver/kt -- python3 /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 > /root/repo/gpurun_out/r6w/r06_driver_cmd_bench_under_rocprof.json 2> /root/repo/gpurun_out/r6w/rocprof.err); echo "rocprof rc=$?"
cp $(ls /tmp/prof_driver/kt/*/*_kernel_stats.csv | head -1) gpurun_out/r6w/r06_driver_cmd_kernel_stats.csv
tail -n 1 gpurun_out/r6w/r06_driver_cmd_bench.json | wc -c
