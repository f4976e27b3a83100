#!/bin/bash
# Diagnostic builds of the HIP library (timing experiments only, outputs are wrong by design):
#   libsmm_skipcompute.so : tile kernel stages and stores but skips the link loop
#   libsmm_skipstage.so   : tile kernel skips the HBM->LDS staging loads
# Extra variants: bash tools/exp/build_exp.sh NAME=-DFLAG ...   (-> tools/exp/libsmm_NAME.so)
# Use with SMM_LIB_PATH=tools/exp/libsmm_<name>.so python bench.py ...
set -e
cd "$(dirname "$0")/../../smmregrid_amd/csrc"
specs=("$@")
[ ${#specs[@]} -eq 0 ] && specs=(skipcompute=-DSMM_EXP_SKIP_COMPUTE skipstage=-DSMM_EXP_SKIP_STAGE)
for spec in "${specs[@]}"; do
  name=${spec%%=*}
  flags=${spec#*=}
  make -j8 BUILD=build_$name OUT=../../tools/exp/libsmm_$name.so EXTRA="$flags" > /dev/null
done
ls -la ../../tools/exp/*.so
