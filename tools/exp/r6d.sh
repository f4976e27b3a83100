# round 6, GPU call d: (1) is the config-4 shape's rate a property of the process or of the allocation?  (2) cfg3sb with the
# latitude-pair tile order, A/B on one box, with the L2 counters;  (3) the whole GPU suite;  (4) the default bench line.
mkdir -p gpurun_out/r6d && cd /root/repo
bin=tools/exp/ceiling
shape="256 16384 2048 16384 100663296 400000 8 10 20480 8 52428800 0 0 0"
for p in 1 2 3; do timeout -k 10 200 $bin $shape 0 0 8 > gpurun_out/r6d/realloc_$p.txt 2>&1 || echo "realloc $p failed"; done
cat gpurun_out/r6d/realloc_*.txt
for rep in 1 2 3; do
  for t in "" "--tune sb_pair_tiles=45" "--tune sb_pair_tiles=90"; do
    tag=$(echo "$t" | tr -c 'a-z0-9' '_'); [ -z "$t" ] && tag=default
    python bench.py --workload cfg3sb $t --steps 20 --warmup 5 --no-cpu-baseline --others none --configs none --user-path none > gpurun_out/r6d/cfg3sb_${tag}_$rep.json 2> gpurun_out/r6d/cfg3sb_${tag}_$rep.err || echo "cfg3sb $tag failed"
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6d/cfg3sb_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(d["roofline"]["kernel_ms"],3), d["spot_check"]["bit_equal_to_oracle"])
    except Exception as e: print(f, "ERR", e)
PY
for t in "" "--tune sb_pair_tiles=45"; do
  echo "== counters cfg3sb '$t'" >> gpurun_out/r6d/cfg3sb_counters.txt
  SMM_BENCH_ARGS="$t" bash tools/exp/pmc.sh cfg3sb "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" >> gpurun_out/r6d/cfg3sb_counters.txt 2>&1
done
cat gpurun_out/r6d/cfg3sb_counters.txt
python -m pytest tests -m gpu -x -q > gpurun_out/r6d/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -5 gpurun_out/r6d/gpu_suite.log
python bench.py > gpurun_out/r6d/bench_default.json 2> gpurun_out/r6d/bench_default.err; echo "bench rc=$?"; tail -c 3600 gpurun_out/r6d/bench_default.json
