mkdir -p gpurun_out/r6v && cd /root/repo
python -m pytest tests/test_gpu_round6.py tests/test_gpu_facade.py tests/test_gpu_user_path.py tests/test_gpu_resources.py -x -q > gpurun_out/r6v/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r6v/tests.log
python tools/exp/facade_big.py 512; python tools/exp/facade_big.py 64; python tools/exp/facade_big.py 2048
SMM_RESULT_CACHE=0 python tools/exp/facade_big.py 512
python - <<'PY'
import time, numpy as np, sys
sys.path.insert(0, "/root/repo")
from smmregrid_amd import pinned_empty
for mb in (64, 265, 1024):
    t0 = time.perf_counter(); a = pinned_empty((mb << 20,), np.uint8); t1 = time.perf_counter(); del a
    print("hipHostMalloc %d MiB: %.1f ms" % (mb, (t1 - t0) * 1e3))
PY
python tools/host_pipeline_bench_levels.py 40 | python -c "import json,sys; d=json.load(sys.stdin); print({k:(round(d[k]['seconds'],3)) for k in ('packed','whole_rows')}, d['bit_identical'])"
