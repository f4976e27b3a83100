mkdir -p gpurun_out/r6z && cd /root/repo
for rep in 1 2 3; do python tools/user_path_bench.py 512 --only-h2h > gpurun_out/r6z/h2h_$rep.json 2>/dev/null; python - <<PY
import json
d=json.load(open("gpurun_out/r6z/h2h_$rep.json"))["host_to_host"]
print({k: (round(d[k]["cells_per_s"]/1e6), d[k]["stage_ms"]["stage_in"]) for k in ("pageable_packed","pinned_packed","pageable_whole_rows","pinned_whole_rows")}, d["ceilings"]["host_copy_GBs"])
PY
done
python tools/host_pipeline_bench_levels.py 40 | python -c "import json,sys; d=json.load(sys.stdin); print({k:(round(d[k]['seconds'],3)) for k in ('packed','whole_rows')}, d['bit_identical'])"
python -m pytest tests/test_gpu_round6.py tests/test_gpu_user_path.py tests/test_gpu_robustness.py -x -q 2>&1 | tail -2
