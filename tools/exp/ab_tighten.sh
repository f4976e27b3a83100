for rep in 1 2 3; do
 for lib in smmregrid_amd/libsmmregrid_hip.so tools/exp/libsmm_tighten3.so tools/exp/libsmm_tighten6.so; do
  for v in 0 9; do
   SMM_LIB_ALLOW_MISSING=1 SMM_LIB_PATH=$PWD/$lib python bench.py --workload cfg4s --variant $v --steps 15 --warmup 4 --no-cpu-baseline --others none --configs none 2>/dev/null | python tools/short.py $(basename $lib .so) v$v
  done
 done
done
