# FETCH_SIZE / WRITE_SIZE of one workload for several tuning-knob sets (0 = default): bash tools/exp/fetch.sh cfg4s "0 xcd_run=-1"
set -e
export TMPDIR=/tmp
root=$PWD
for v in $2; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pf_$v_$c
    (cd /tmp && rocprofv3 --pmc $c --output-format csv -d /tmp/pf_${v}_$c -- python3 $root/bench.py --workload $1 --steps 3 --warmup 1 --no-cpu-baseline --others none --configs none --user-path none $( [ "$v" != 0 ] && echo --tune $v ) $3 > /dev/null 2>&1)
    python3 - $v $c <<'PY'
import csv,glob,sys,collections
v,c=sys.argv[1:3]
f=glob.glob(f"/tmp/pf_{v}_{c}/*/*_counter_collection.csv")[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "smm_apply_" in r["Kernel_Name"] and r["Counter_Name"]==c: agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k,vals in agg.items(): print("variant",v,c,"KiB mean %.4g"%(sum(vals)/len(vals)),"x2 GB %.2f"%(2*1024*sum(vals)/len(vals)/1e9),"n",len(vals))
PY
  done
done
