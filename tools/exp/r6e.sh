# round 6, GPU call e: (1) config-4 shape with the write buffer from the virtual-memory API at large alignments;
# (2) cfg3sb latitude-pair order once more with XCD runs that hold whole bands; (3) the default bench line (length check).
mkdir -p gpurun_out/r6e && cd /root/repo
bash tools/exp/cfg4_vmm.sh gpurun_out/r6e/vmm > gpurun_out/r6e/vmm.txt 2>&1; cat gpurun_out/r6e/vmm.txt
for rep in 1 2 3; do
  for t in "" "--tune sb_pair_tiles=45" "--tune sb_pair_tiles=45,xcd_run=45" "--tune sb_pair_tiles=45,xcd_run=90" "--tune xcd_run=45"; do
    tag=$(echo "$t" | tr -c 'a-z0-9' '_'); [ -z "$t" ] && tag=default
    python bench.py --workload cfg3sb $t --steps 20 --warmup 5 --no-cpu-baseline --others none --configs none --user-path none > gpurun_out/r6e/cfg3sb_${tag}_$rep.json 2> gpurun_out/r6e/cfg3sb_${tag}_$rep.err || echo "cfg3sb $tag failed"
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6e/cfg3sb_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(d["roofline"]["kernel_ms"],3), d["spot_check"]["bit_equal_to_oracle"])
    except Exception as e: print(f, "ERR", e)
PY
python bench.py > gpurun_out/r6e/bench_default.json 2> gpurun_out/r6e/bench_default.err; echo "bench rc=$?"; tail -n 1 gpurun_out/r6e/bench_default.json | wc -c
