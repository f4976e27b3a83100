mkdir -p gpurun_out/r6q && cd /root/repo
for rows in 8 12 16 24 48; do timeout -k 10 300 python tools/user_path_bench.py $rows --only-h2h > gpurun_out/r6q/h2h_$rows.json 2> gpurun_out/r6q/h2h_$rows.err; python - <<PY
import json
d=json.load(open("gpurun_out/r6q/h2h_$rows.json"))["host_to_host"]
print($rows, {k: round(d[k]["ms"],3) for k in ("pageable_packed","pageable_whole_rows","pinned_packed","pinned_whole_rows")}, d["spot_check"])
PY
done
