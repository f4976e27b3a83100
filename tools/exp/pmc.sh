# a few derived counters for one workload: bash tools/exp/pmc.sh cfg3 "LDSBankConflict MemUnitStalled" "VALUBusy SALUBusy" ...
# extra bench arguments (e.g. tuning knobs) through SMM_BENCH_ARGS="--tune tile_staging=2"
set -e
export TMPDIR=/tmp
root=$PWD
wl=$1; shift
i=0
for set in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pm_$i
  (cd /tmp && rocprofv3 --pmc $set --output-format csv -d /tmp/pm_$i -- python3 $root/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --others none --configs none --user-path none $SMM_BENCH_ARGS > /dev/null 2>&1) || { echo "set '$set' failed"; continue; }
  python3 - $i <<'PY'
import csv,glob,sys,collections
i=sys.argv[1]
fs=glob.glob(f"/tmp/pm_{i}/*/*_counter_collection.csv")
if not fs: print("no output"); sys.exit(0)
agg=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if "_apply_tile" in r["Kernel_Name"] or "apply_sb_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items(): print(f"{k}: mean {sum(v)/len(v):.6g} (n={len(v)})")
PY
done
