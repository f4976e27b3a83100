#!/bin/bash
# Same-box A/B of tuning-knob sets of one workload (0 = library default):
#   bash tools/exp/ab_variants.sh cfg4s "0 tile_staging=1 tile_staging=1,tile_rows_per_step=1" [extra bench args]
wl=$1; vars=$2; shift 2
out=gpurun_out/ab_var_$wl
rm -rf $out; mkdir -p $out
for rep in 1 2 3; do
  for v in $vars; do
    t=""; [ "$v" != 0 ] && t="--tune $v"
    python bench.py --workload $wl $t --steps 20 --warmup 5 --no-cpu-baseline --others none --configs none --user-path none "$@" > $out/v${v}_$rep.json 2> $out/v${v}_$rep.err
  done
done
python - <<PY
import json, glob, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("$out/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        res[f.split("/")[-1].rsplit("_", 1)[0]].append(d["roofline"]["kernel_ms"])
    except Exception as e:
        print(f, "ERR", e)
for k in sorted(res):
    v = res[k]
    print("$wl %-40s min %.3f med %.3f  %s" % (k, min(v), sorted(v)[len(v) // 2], ["%.3f" % x for x in v]))
PY
