# A/B of prebuilt library variants under tools/exp/ (timing only)
# usage: bash tools/exp/ab.sh "cfg2 cfg4s" "old new"   ("new" = the in-tree library)
set -e
mkdir -p gpurun_out
: > gpurun_out/ab.log
for rep in 1 2 3; do
  for wl in ${1:-cfg2}; do
    for lib in ${2:-old new}; do
      if [ $lib = new ]; then unset SMM_LIB_PATH; else export SMM_LIB_PATH=$PWD/tools/exp/libsmm_$lib.so; fi
      python bench.py --workload $wl --steps 30 --warmup 3 --no-cpu-baseline --others none --configs none 2>/dev/null | python tools/short.py $wl $lib >> gpurun_out/ab.log
    done
  done
done
cut -c1-75 gpurun_out/ab.log
