mkdir -p gpurun_out/r6t && cd /root/repo
python - <<'PY' > gpurun_out/r6t/reference_sized.json 2> gpurun_out/r6t/reference_sized.err
import json, sys
sys.path.insert(0, "tools")
import user_path_bench as u
print(json.dumps(u.reference_sized(device=0, reps=25, budget_s=60.0)))
PY
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6t/reference_sized.json"))
for k,v in d.items(): print(k, v.get("regrid_ms"), v.get("regrid_ms_min"), v.get("bit_equal"), v.get("init_ms"))
PY
timeout -k 10 400 python tools/soak.py 1500 300 > gpurun_out/r6t/soak.log 2>&1; echo "soak rc=$?"; tail -1 gpurun_out/r6t/soak.log
python -m pytest tests -m gpu -x -q > gpurun_out/r6t/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r6t/tests.log
