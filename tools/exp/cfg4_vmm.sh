#!/bin/bash
# Round 6, config-4 placement, third question: the rate of the config-4 access shape follows the WRITE buffer's allocation
# (tools/exp/cfg4_placement.sh + the in-process probe: profiles/r06_cfg4_placement.txt).  Can an allocation made through the
# virtual-memory API with a large virtual alignment (so that the driver may map large fragments) make the fast class the rule?
#   bash tools/exp/cfg4_vmm.sh [out_dir]
out=${1:-gpurun_out/cfg4_vmm}
mkdir -p $out
bin=$(dirname "$0")/ceiling
[ -x "$bin" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "$(dirname "$0")/ceiling.hip" -o "$bin" || exit 1
shape="256 16384 2048 16384 100663296 400000 8 10 20480 8 52428800 0 0 0"
: > $out/vmm.jsonl
for rep in 1 2 3 4 5 6; do
  for v in "0 0" "2097152 0" "67108864 0" "1073741824 0" "1073741824 1"; do
    set -- $v
    line=$($bin $shape 0 0 0 $1 $2 2>> $out/vmm.err | grep '^placement' | sed 's/^placement //') || { echo "variant $v failed" >&2; continue; }
    [ -z "$line" ] && { echo "variant $v: no output" >&2; continue; }
    echo "{\"vmm_align\": $1, \"vmm_src\": $2, \"rep\": $rep, ${line#\{}" >> $out/vmm.jsonl
  done
done
python3 - <<PY
import json, collections
rows = [json.loads(l) for l in open("$out/vmm.jsonl")]
by = collections.defaultdict(list)
for r in rows: by[(r["vmm_align"], r["vmm_src"])].append(r["total_GBs"])
for k in sorted(by): print("dst via %-28s %s" % (("hipMalloc" if not k[0] else "VMM, alignment %d MiB%s" % (k[0] >> 20, ", src too" if k[1] else "")), " ".join("%.0f" % v for v in by[k])))
PY
tail -3 $out/vmm.err
