#!/bin/bash
# Collects the evidence kept under profiles/ on the GPU box (results land in gpurun_out/profiles_new/,
# copy what should be judged into profiles/):
#   bash tools/collect_profiles.sh r01 "cfg2 cfg3 cfg5tile cfg4s" [pmc]
# Per workload: the bench line, the rocprofv3 kernel-trace stats of the same command and (with "pmc")
# the FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ passes, each in its own run as the pool requires.
set -e
tag=${1:-r02}
wls=${2:-cfg2}
pmc=${3:-}
root=$PWD
out=$root/gpurun_out/profiles_new
mkdir -p $out
export TMPDIR=/tmp
for wl in $wls; do
  python3 bench.py --workload $wl --steps 20 --warmup 3 --others none --configs none --user-path none > $out/${tag}_bench_$wl.json
  d=/tmp/prof_$wl
  rm -rf $d
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d/kt -- python3 $root/bench.py --workload $wl \
      --steps 20 --warmup 3 --no-cpu-baseline --others none --configs none --user-path none > $out/${tag}_${wl}_bench_under_rocprof.json)
  cp $(ls $d/kt/*/*_kernel_stats.csv | head -1) $out/${tag}_${wl}_kernel_stats.csv
  if [ -n "$pmc" ]; then
    (cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d/fetch -- python3 $root/bench.py --workload $wl \
        --steps 5 --warmup 1 --no-cpu-baseline --others none --configs none --user-path none > /dev/null)
    (cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $d/write -- python3 $root/bench.py --workload $wl \
        --steps 5 --warmup 1 --no-cpu-baseline --others none --configs none --user-path none > /dev/null)
    (cd /tmp && rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $d/rdreq -- python3 \
        $root/bench.py --workload $wl --steps 5 --warmup 1 --no-cpu-baseline --others none --configs none --user-path none > /dev/null)
    lps=1   # launches per bench step (the batch-fastest level groups are one grouped launch since round 5)
    python3 tools/summarize_pmc.py --launches-per-step $lps --fetch $d/fetch --write $d/write --rdreq $d/rdreq --key $wl/default/auto/0 \
        --out $out/${tag}_${wl}_pmc_summary.json --traffic $out/traffic.json --git-head "${SMM_GIT_HEAD:-unknown}" > /dev/null
  fi
  echo "done $wl"
done
if [ -n "$SMM_DRIVER_CMD" ]; then
  # the driver's own command (headline + the secondary workloads of the same line) under the kernel trace
  rm -rf /tmp/prof_driver
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_driver/kt -- python3 $root/bench.py \
      --gpus 1 --steps 20 --warmup 5 > $out/${tag}_driver_cmd_bench_under_rocprof.json)
  cp $(ls /tmp/prof_driver/kt/*/*_kernel_stats.csv | head -1) $out/${tag}_driver_cmd_kernel_stats.csv
fi
ls $out
