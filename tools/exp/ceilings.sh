#!/bin/bash
# Practical HBM ceilings for the access shapes of the regrid kernels (tools/exp/ceiling.hip), one JSON
# line per shape -> $1 (default gpurun_out/ceilings.jsonl).  bash tools/exp/ceilings.sh [out] [wg_per_cu]
out=${1:-gpurun_out/ceilings.jsonl}
wg=${2:-8}
bin=$(dirname "$0")/ceiling
[ -x "$bin" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "$(dirname "$0")/ceiling.hip" -o "$bin" || exit 1
: > "$out"
run() { # label, args...
  local label=$1; shift
  line=$("$bin" "$@") || { echo "FAILED $label" >&2; return 1; }
  echo "{\"shape\": \"$label\", ${line#\{}" >> "$out"
}
#            label                                   run    rd/unit seg   wr/unit stride     units  wg
run "read only, 64-KiB runs"                          65536  65536   1024  0       1024       400000 $wg
run "write only, 64-KiB segments"                     1024   0       65536 65536   65536      400000 $wg
run "copy 1:1, 64-KiB runs / segments"                65536  65536   65536 65536   65536      200000 $wg
run "cfg2 native mix 8:1, 8-KiB runs, 2-KiB segs"     8192   65536   2048  8192    524288     400000 $wg
run "cfg2sb 4:1, 1-KiB runs, 128-B segs on 512 KiB"   1024   65536   128   16384   524288     400000 $wg
run "cfg2sb 4:1, 1-KiB runs, 256-B segs"              1024   65536   256   16384   524288     400000 $wg
run "cfg2sb 4:1, 1-KiB runs, 512-B segs"              1024   65536   512   16384   524288     400000 $wg
run "cfg2sb 4:1, 1-KiB runs, 1-KiB segs"              1024   65536   1024  16384   524288     400000 $wg
run "cfg2sb reads only, 1-KiB runs"                   1024   65536   128   0       524288     400000 $wg
run "cfg3 30:1, 2-KiB runs, 512-B segs"               2048   30720   512   1024    524288     800000 $wg
run "cfg3 30:1, 1-KiB runs"                           1024   30720   512   1024    524288     800000 $wg
run "cfg3 30:1, 512-B runs"                           512    30720   512   1024    524288     800000 $wg
run "cfg3 30:1, 128-B runs"                           128    30720   512   1024    524288     800000 $wg
run "cfg4s 1:1 of fetched lines, 128-B runs, 2-KiB segs on 96 MiB" 128 2048 2048 2048 100663296 2000000 $wg
run "cfg4s 2:1 (f32 out), 128-B runs, 1-KiB segs"     128    2048    1024  1024    50331648   2000000 $wg
run "cfg5 4:1, 8-KiB runs, 2-KiB segs"                8192   32768   2048  8192    2097152    400000 $wg
cat "$out"
