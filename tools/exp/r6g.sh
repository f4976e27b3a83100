mkdir -p gpurun_out/r6g && cd /root/repo
python tools/exp/cfg3_knobs.py 3 > gpurun_out/r6g/cfg3_knobs.txt 2> gpurun_out/r6g/cfg3_knobs.err; echo "rc=$?"; cat gpurun_out/r6g/cfg3_knobs.txt; tail -3 gpurun_out/r6g/cfg3_knobs.err
for t in "" "--tune tile_x_loads=1"; do
  echo "== counters cfg3 '$t'" >> gpurun_out/r6g/cfg3_counters.txt
  SMM_BENCH_ARGS="$t" bash tools/exp/pmc.sh cfg3 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" >> gpurun_out/r6g/cfg3_counters.txt 2>&1
done
cat gpurun_out/r6g/cfg3_counters.txt
