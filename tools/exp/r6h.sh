# round 6, GPU call h: soak (seeded random operators through every kernel form and the host pipelines), the whole GPU suite,
# the default bench line (traffic replayed from the refreshed profiles/traffic.json), bench.py --gpus 2 through the stand-in.
mkdir -p gpurun_out/r6h && cd /root/repo
timeout -k 10 600 python tools/soak.py 2000 400 > gpurun_out/r6h/soak.log 2>&1; echo "soak rc=$?"; tail -2 gpurun_out/r6h/soak.log
python -m pytest tests -m gpu -x -q > gpurun_out/r6h/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/r6h/gpu_suite.log
python bench.py > gpurun_out/r6h/bench_default.json 2> gpurun_out/r6h/bench_default.err; echo "bench rc=$?"; tail -n 1 gpurun_out/r6h/bench_default.json | wc -c
g++ -O1 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/cpp/fake_rccl.cpp -o /tmp/libfake_rccl.so -L/opt/rocm/lib -lamdhip64 -lrt -Wl,-rpath,/opt/rocm/lib
SMM_RCCL_LIB=/tmp/libfake_rccl.so SMM_BENCH_SHARE_GPUS=1 timeout -k 10 600 python bench.py --gpus 2 --gather root --steps 5 --warmup 2 --configs none --no-cpu-baseline > gpurun_out/r6h/bench_2ranks_stand_in.json 2> gpurun_out/r6h/bench_2ranks_stand_in.err; echo "2-rank rc=$?"
tail -n 1 gpurun_out/r6h/bench_default.json | cut -c1-1800
