mkdir -p gpurun_out/r6p && cd /root/repo
for nt in 8 12 16 24 32; do timeout -k 10 300 python tools/host_pipeline_bench_levels.py $nt > gpurun_out/r6p/levels_$nt.json 2> gpurun_out/r6p/levels_$nt.err; echo "levels $nt rc=$?"; python - <<PY
import json
d=json.load(open("gpurun_out/r6p/levels_$nt.json"))
print($nt, "packed %.3f s (%d chunks, kernel %.1f ms) whole rows %.3f s  same bits %s" % (d["packed"]["seconds"], d["packed"]["stages"]["chunks"], d["packed"]["stages"]["kernel_ms"], d["whole_rows"]["seconds"], d["bit_identical"]))
PY
done
python -m pytest tests/test_gpu_round6.py tests/test_gpu_fuzz.py tests/test_gpu_facade.py -x -q > gpurun_out/r6p/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r6p/tests.log
