mkdir -p gpurun_out/r6y && cd /root/repo
cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head -3; rocm-smi --showtoponuma 2>/dev/null | grep -i "numa" | head -5
for cpus in "0-63,128-191" "64-127,192-255" "all"; do
  for rep in 1 2; do
    if [ "$cpus" = "all" ]; then python tools/user_path_bench.py 512 --only-h2h > gpurun_out/r6y/h2h_all_$rep.json 2>/dev/null; f=gpurun_out/r6y/h2h_all_$rep.json
    else taskset -c $cpus python tools/user_path_bench.py 512 --only-h2h > gpurun_out/r6y/h2h_${cpus%%-*}_$rep.json 2>/dev/null; f=gpurun_out/r6y/h2h_${cpus%%-*}_$rep.json; fi
    python - <<PY
import json
d=json.load(open("$f"))["host_to_host"]
print("$cpus", {k: (round(d[k]["cells_per_s"]/1e6), d[k]["stage_ms"]["stage_in"]) for k in ("pageable_packed","pinned_packed","pageable_whole_rows")}, d["ceilings"])
PY
  done
done
