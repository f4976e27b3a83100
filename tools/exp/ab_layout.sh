#!/bin/bash
# Same-box A/B of the two destination layouts: bash tools/exp/ab_layout.sh "cfg3 conmid cfg5tile"
wls=${1:-cfg3}
out=gpurun_out/ab_layout
rm -rf $out; mkdir -p $out
for rep in 1 2 3; do
  for wl in $wls; do
    for lay in rows patches; do
      SMM_LAYOUT=$lay python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline \
          > $out/${wl}_${lay}_$rep.json 2> $out/${wl}_${lay}_$rep.err
    done
  done
done
python - <<PY
import json, glob, collections
res = collections.defaultdict(list); extra = {}
for f in sorted(glob.glob("$out/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        wl, lay, _ = f.split("/")[-1][:-5].rsplit("_", 2)
        res[(wl, lay)].append(d["roofline"]["kernel_ms"])
        extra[(wl, lay)] = (d["config"]["plan"].get("dst_patches"), d["config"]["plan"]["rows_per_block"], d["roofline"]["line_granular_bytes"])
    except Exception as e:
        print(f, "ERR", e)
for k in sorted(res):
    v = res[k]
    print("%-10s %-8s min %.3f med %.3f  %s  plan %s" % (k[0], k[1], min(v), sorted(v)[len(v) // 2], ["%.3f" % x for x in v], extra[k]))
PY
