#!/bin/bash
# Diagnostic builds of the HIP library (timing experiments only, outputs are wrong by design):
#   libsmm_skipcompute.so : tile kernel stages and stores but skips the link loop
#   libsmm_skipstage.so   : tile kernel skips the HBM->LDS staging loads
set -e
cd "$(dirname "$0")/../../smmregrid_amd/csrc"
for v in SKIP_COMPUTE SKIP_STAGE; do
  out=../../tools/exp/libsmm_$(echo $v | tr 'A-Z_' 'a-z ' | tr -d ' ').so
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -shared -x hip \
      -DSMM_EXP_$v smm_device.hip smm_build.cpp -o $out &
done
wait
ls -la ../../tools/exp/
