# round 6, GPU call f: new GPU tests, then the evidence kept under profiles/ (bench line, kernel trace, PMC passes per workload,
# the driver's command under the kernel trace), then bench.py --gpus 2 through the stand-in collective library.
mkdir -p gpurun_out/r6f && cd /root/repo
python -m pytest tests/test_gpu_round6.py tests/test_gpu_multirank_stand_in.py tests/test_gpu_batch_fastest.py -x -q > gpurun_out/r6f/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r6f/tests.log
SMM_GIT_HEAD=$(cat gpurun_out/../.git_head 2>/dev/null || echo unknown) SMM_DRIVER_CMD=1 bash tools/collect_profiles.sh r06 "cfg2 cfg2sb cfg2sbk cfg3 cfg3c cfg3sb cfg4s cfg5tile" pmc > gpurun_out/r6f/collect.log 2>&1; echo "collect rc=$?"; tail -5 gpurun_out/r6f/collect.log
g++ -O1 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/cpp/fake_rccl.cpp -o /tmp/libfake_rccl.so -L/opt/rocm/lib -lamdhip64 -lrt -Wl,-rpath,/opt/rocm/lib
SMM_RCCL_LIB=/tmp/libfake_rccl.so SMM_BENCH_SHARE_GPUS=1 timeout -k 10 600 python bench.py --gpus 2 --gather root --steps 5 --warmup 2 --configs none --no-cpu-baseline > gpurun_out/r6f/bench_2ranks_stand_in.json 2> gpurun_out/r6f/bench_2ranks_stand_in.err; echo "2-rank rc=$?"; tail -c 1500 gpurun_out/r6f/bench_2ranks_stand_in.json
