#!/bin/bash
# Same-box A/B of bench argument sets for one workload; each set is one quoted string:
#   bash tools/exp/ab_args.sh cfg3 "--tune tile_walk=64" "--tune tile_walk=120"        (the environment is inherited)
wl=$1; shift
out=gpurun_out/ab_args_$wl
rm -rf $out; mkdir -p $out
for rep in 1 2 3; do
  i=0
  for set in "$@"; do
    i=$((i+1))
    python bench.py --workload $wl --steps 8 --warmup 2 --no-cpu-baseline --others none --configs none $set > $out/s${i}_$rep.json 2> $out/s${i}_$rep.err
  done
done
i=0
for set in "$@"; do
  i=$((i+1))
  python - "$out" "$i" "$set" <<'PY'
import json, glob, sys
out, i, name = sys.argv[1:4]
v = []
for f in sorted(glob.glob(f"{out}/s{i}_*.json")):
    try:
        v.append(json.loads(open(f).read().strip().splitlines()[-1])["roofline"]["kernel_ms"])
    except Exception as e:
        print(f, "ERR", e)
if v:
    print("%-28s min %.3f med %.3f  %s" % (name, min(v), sorted(v)[len(v) // 2], ["%.3f" % x for x in v]))
PY
done
