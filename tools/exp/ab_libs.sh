#!/bin/bash
# Same-box A/B of two builds of the library (box-to-box HBM rates differ by 10-20 %):
#   bash tools/exp/ab_libs.sh "cfg2 cfg3" tools/exp/libsmm_r01.so smmregrid_amd/libsmmregrid_hip.so
# alternates the libraries three times per workload and prints the kernel times.
wls=${1:-cfg2}
shift
libs=("$@")
out=gpurun_out/ab
mkdir -p $out
for rep in 1 2 3; do
  for wl in $wls; do
    for lib in "${libs[@]}"; do
      tag=$(basename $lib .so)
      SMM_LIB_ALLOW_MISSING=1 SMM_LIB_PATH=$PWD/$lib python bench.py --workload $wl --steps 10 --warmup 3 --others none --configs none \
          --no-cpu-baseline > $out/${wl}_${tag}_$rep.json 2> $out/${wl}_${tag}_$rep.err
    done
  done
done
python - <<PY
import json, glob, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("$out/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        wl, tag = f.split("/")[-1].rsplit("_", 1)[0].split("_", 1)
        res[(wl, tag)].append(d["roofline"]["kernel_ms"])
    except Exception as e:
        print(f, "ERR", e)
for k in sorted(res):
    v = res[k]
    print("%-10s %-22s min %.3f  med %.3f  all %s" % (k[0], k[1], min(v), sorted(v)[len(v) // 2], ["%.3f" % x for x in v]))
PY
