#!/bin/bash
# Round-6 bounded pass on BASELINE config 4: three of 26 launches of the config-4 access shape ran at
# 5.9 TB/s against 4.8 - 5.4 for the rest -- "the rate follows where hipMalloc puts the buffers".  Which relation of the
# two base addresses do the fast placements share, and can the library produce it for buffers it owns?
#   (1) the shape as >= 24 separate processes, a dummy allocation of varying size made first so that hipMalloc places
#       the two buffers differently; every line carries both base pointers, their difference mod 2 MiB / 64 MiB / 1 GiB
#       and the allocation ranges;
#   (2) the write buffer's base moved inside one over-sized allocation: 0 ... 64 MiB in 8 steps, 3 launches each.
# bash tools/exp/cfg4_placement.sh [out_dir]
out=${1:-gpurun_out/cfg4_placement}
mkdir -p $out
bin=$(dirname "$0")/ceiling
[ -x "$bin" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "$(dirname "$0")/ceiling.hip" -o "$bin" || exit 1
#      run rd/unit seg  wr/unit stride    units  wg reps pitch rows slab     exact barrier sweep
shape="256 16384  2048 16384  100663296 400000 8  10   20480 8    52428800 0     0       0"
: > $out/placement.jsonl
i=0
for pre in 0 0 0 0 1048576 3145728 17825792 68157440 538968064 1073741824 1075838976 2147483648 \
           0 0 0 0 1048576 3145728 17825792 68157440 538968064 1073741824 1075838976 2147483648; do
  i=$((i + 1))
  line=$($bin $shape 0 $pre | grep '^placement' | sed 's/^placement //') || { echo "launch $i failed" >&2; break; }
  echo "{\"pass\": \"processes\", \"launch\": $i, \"pre_alloc\": $pre, ${line#\{}" >> $out/placement.jsonl
done
for off in 0 8388608 16777216 25165824 33554432 41943040 50331648 58720256 67108864 4096 65536 2097152; do
  for rep in 1 2 3; do
    line=$($bin $shape $off 0 | grep '^placement' | sed 's/^placement //') || { echo "offset $off failed" >&2; break 2; }
    echo "{\"pass\": \"y_offset\", \"dst_off\": $off, \"rep\": $rep, ${line#\{}" >> $out/placement.jsonl
  done
done
python3 - <<PY
import json
rows = [json.loads(l) for l in open("$out/placement.jsonl")]
print("%-10s %-12s %8s  %-14s %-14s %10s %10s %11s %9s %9s" % ("pass", "knob", "GB/s", "src", "dst", "d mod 2M", "d mod 64M", "d mod 1G", "src%1G/2M", "dst%1G/2M"))
for r in rows:
    knob = r.get("pre_alloc", r.get("dst_off"))
    print("%-10s %-12d %8.0f  %-14s %-14s %10d %10d %11d %9d %9d" % (r["pass"], knob, r["total_GBs"], r["src"], r["dst"], r["diff_mod_2M"],
          r["diff_mod_64M"], r["diff_mod_1G"], r["src_mod_1G"] >> 21, r["dst_mod_1G"] >> 21))
PY
