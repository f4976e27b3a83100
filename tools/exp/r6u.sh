# round 6, GPU call u (final tree): the evidence under profiles/ (bench line, kernel trace, PMC passes per
# workload), the driver's command plain and under the kernel trace with the fresh traffic.json.
mkdir -p gpurun_out/r6u && cd /root/repo
export TMPDIR=/tmp
SMM_GIT_HEAD=$(cat .git_head 2>/dev/null || echo unknown) bash tools/collect_profiles.sh r06 "cfg2 cfg2sb cfg2sbk cfg3 cfg3c cfg3sb cfg4s cfg5tile" pmc > gpurun_out/r6u/collect.log 2>&1; echo "collect rc=$?"; tail -2 gpurun_out/r6u/collect.log
cp gpurun_out/profiles_new/traffic.json profiles/traffic.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/profiles_new/r06_driver_cmd_bench.json 2> gpurun_out/r6u/bench.err; echo "bench rc=$?"
rm -rf /tmp/prof_driver
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_driver/kt -- python3 /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 > /root/repo/gpurun_out/profiles_new/r06_driver_cmd_bench_under_rocprof.json 2> /root/repo/gpurun_out/r6u/rocprof.err); echo "rocprof rc=$?"
cp $(ls /tmp/prof_driver/kt/*/*_kernel_stats.csv | head -1) gpurun_out/profiles_new/r06_driver_cmd_kernel_stats.csv
tail -n 1 gpurun_out/profiles_new/r06_driver_cmd_bench.json | wc -c
