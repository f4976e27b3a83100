# FETCH_SIZE of the batch-fastest kernels of one workload per tuning variant (bytes per bench step):
#   bash tools/exp/fetch_sb.sh cfg3sb "0 8" 75      (launches per step: 75 for the level groups, 1 otherwise)
set -e
export TMPDIR=/tmp
root=$PWD
for v in $2; do
  rm -rf /tmp/pfs_$v
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pfs_$v -- python3 $root/bench.py --workload $1 --steps 3 --warmup 1 --no-cpu-baseline --others none --configs none --variant $v > /dev/null 2>&1)
  python3 - $v ${3:-1} <<'PY'
import csv,glob,sys,collections
v,lps=sys.argv[1],int(sys.argv[2])
f=glob.glob(f"/tmp/pfs_{v}/*/*_counter_collection.csv")[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "smm_apply_sb" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE": agg[r["Kernel_Name"][28:70]].append(float(r["Counter_Value"]))
for k,vals in agg.items(): print("variant",v,k,"fetch x2 GB per step %.2f"%(lps*2*1024*sum(vals)/len(vals)/1e9),"launches",len(vals))
PY
done
