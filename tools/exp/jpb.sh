# walk-length sweep on one workload for several library builds: bash tools/exp/jpb.sh cfg5 "old new" "8 16 32 64"
set -e
mkdir -p gpurun_out
: > gpurun_out/jpb.log
for rep in 1 2; do
for lib in ${2:-new}; do
  if [ $lib = new ]; then unset SMM_LIB_PATH; else export SMM_LIB_PATH=$PWD/tools/exp/libsmm_$lib.so; fi
  for j in ${3:-0}; do
    python bench.py --workload $1 --steps 10 --warmup 2 --no-cpu-baseline --others none --configs none --jpb $j $4 2>/dev/null | python tools/short.py $1 $lib jpb$j >> gpurun_out/jpb.log
  done
done
done
cut -c1-80 gpurun_out/jpb.log
