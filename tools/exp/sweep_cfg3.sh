#!/bin/bash
# cfg3 knob sweep in one box: ablations + walk length + XCD run length + cache policy
out=gpurun_out/sweep3
rm -rf $out; mkdir -p $out
run() { tag=$1; shift; python bench.py --workload cfg3 --steps 8 --warmup 2 --no-cpu-baseline --others none --configs none "$@" > $out/$tag.json 2> $out/$tag.err; }
run base
SMM_LIB_PATH=$PWD/tools/exp/libsmm_skipcompute.so run skipcompute
SMM_LIB_PATH=$PWD/tools/exp/libsmm_skipstage.so run skipstage
for j in 16 32 120; do run jpb$j --jpb $j; done
for v in 3 4 6 7 13; do run var$v --variant $v; done
run base2
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
        print("%-14s ms=%.3f frac=%.3f" % (f.split("/")[-1][:-5], r["kernel_ms"], r["frac"]))
    except Exception as e:
        print(f, "ERR", e)
PY
