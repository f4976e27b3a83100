#!/bin/bash
# config-2 native mix (8 : 1) against the write-segment length and the read-run length: does a longer Y segment or a
# longer X run lift the ceiling of the mix?  bash tools/exp/ceil_mix.sh [out]
out=${1:-gpurun_out/ceil_mix.jsonl}
bin=$(dirname "$0")/ceiling
[ -x "$bin" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "$(dirname "$0")/ceiling.hip" -o "$bin" || exit 1
: > "$out"
for seg in 512 2048 8192 65536; do
  for run in 2048 8192 65536; do
    line=$("$bin" $run 65536 $seg 8192 524288 400000 8) || exit 1
    echo "{\"shape\": \"8:1 run $run seg $seg\", ${line#\{}" >> "$out"
  done
done
for wg in 4 16; do
  line=$("$bin" 8192 65536 2048 8192 524288 400000 $wg) || exit 1
  echo "{\"shape\": \"8:1 run 8192 seg 2048 wg_per_cu $wg\", ${line#\{}" >> "$out"
done
python3 - "$out" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    r = json.loads(l); print(f"{r['shape']:40s} {r['total_GBs']:.0f} GB/s")
PY
