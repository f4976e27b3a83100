#!/bin/bash
# How much of the config-4 shape's rate depends on the bytes a wave keeps in flight and on a workgroup barrier per unit?
# (tools/exp/ceiling.hip with exact_rounds = 1: a unit of n KiB issues n 1-KiB loads, waits, stores, goes on.)
out=${1:-gpurun_out/cfg4_inflight.jsonl}
bin=$(dirname "$0")/ceiling
[ -x "$bin" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "$(dirname "$0")/ceiling.hip" -o "$bin" || exit 1
: > $out
run() { local label=$1; shift; line=$("$bin" "$@") || { echo "FAILED $label" >&2; return 1; }; echo "{\"shape\": \"$label\", ${line#\{}" >> $out; }
for wg in 8; do
for bar in 0 1; do
#   label                                                   run rd/unit seg  wr/unit stride    units   wg reps pitch rows slab     loads/round barrier
run "unit 2 KiB = 1 batch row (tile 8 x 256 B), 2 loads in flight"  256 2048  2048 2048  100663296 3200000 $wg 5 20480 8  52428800 2 $bar
run "unit 4 KiB = 2 batch rows, 4 loads in flight"                  256 4096  2048 4096  100663296 1600000 $wg 5 20480 8  52428800 4 $bar
run "unit 8 KiB = 4 batch rows, 8 loads in flight"                  256 8192  2048 8192  100663296 800000  $wg 5 20480 8  52428800 8 $bar
run "unit 16 KiB = 8 batch rows, 8 loads in flight"                 256 16384 2048 16384 100663296 400000  $wg 5 20480 8  52428800 8 $bar
run "unit 16 KiB = 8 batch rows, 2 loads in flight"                 256 16384 2048 16384 100663296 400000  $wg 5 20480 8  52428800 2 $bar
done
done
python3 - $out <<'PY'
import sys, json
for l in open(sys.argv[1]):
    d = json.loads(l)
    print("%-62s barrier %d  total %.0f GB/s" % (d["shape"], d["barrier"], d["total_GBs"]))
PY
