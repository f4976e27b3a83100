#!/bin/bash
# Round-5 bounded pass on BASELINE config 4: what holds the config-4 kernel at ~5 TB/s?
#   (1) ceiling lines whose shape is the kernel's: TILES of neighbouring 128-B lines (8 source rows x 256 B, row
#       pitch 20 480 B = 5120 f32) + 2-KiB f64 write segments on the 96-MiB row pitch of Y, mix 1 : 1;
#   (2) SQ wait / busy split and TCC write-side counters of cfg4s against cfg5tile (the control, 5.96 TB/s).
# bash tools/exp/cfg4_pass.sh [out_dir]
out=${1:-gpurun_out/cfg4_pass}
mkdir -p $out
bin=$(dirname "$0")/ceiling
[ -x "$bin" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "$(dirname "$0")/ceiling.hip" -o "$bin" || exit 1
: > $out/ceilings.jsonl
run() { local label=$1; shift; line=$("$bin" "$@") || { echo "FAILED $label" >&2; return 1; }; echo "{\"shape\": \"$label\", ${line#\{}" >> $out/ceilings.jsonl; }
for wg in 8 16; do
# a unit = a workgroup walking 8 batch rows: the same tile (8 source rows x 256 B = 2 x 8 neighbouring 128-B lines, row
# pitch 20 480 B) of 8 consecutive X rows 52 428 800 B apart, and a 2-KiB f64 segment into 8 Y rows 100 663 296 B apart
#    label                                                         run  rd/unit seg  wr/unit stride     units   wg  reps pitch rows slab
run "copy 1:1, 64-KiB runs / segments"                             65536 65536 65536 65536  65536      200000  $wg 5
run "write only, 2-KiB segs of 8 rows on 96 MiB"                   1024  0     2048  16384  100663296  400000  $wg 5
run "write only, 64-KiB segments"                                  1024  0     65536 65536  65536      200000  $wg 5
run "read only, walk of 8 slabs, tile 8 rows x 256 B"              256   16384 2048  0      100663296  400000  $wg 5  20480 8 52428800
run "read only, walk of 8 slabs, tile 16 rows x 128 B"             128   16384 2048  0      100663296  400000  $wg 5  20480 16 52428800
run "read only, scattered 128-B lines, 16-KiB units"               128   16384 2048  0      100663296  400000  $wg 5
run "read only, 64-KiB runs"                                       65536 65536 1024  0      1024       200000  $wg 5
run "cfg4 1:1, walk of 8 slabs, tile 8 x 256 B, 2-KiB segs/96 MiB" 256   16384 2048  16384  100663296  400000  $wg 5  20480 8 52428800
run "cfg4 1:1, walk of 8 slabs, tile 16 x 128 B, 2-KiB segs"       128   16384 2048  16384  100663296  400000  $wg 5  20480 16 52428800
run "cfg4 1:1, walk of 8 slabs, tile 4 x 512 B, 2-KiB segs"        512   16384 2048  16384  100663296  400000  $wg 5  20480 4 52428800
run "cfg4 1:1, 8 tiles of one slab, tile 8 x 256 B, 2-KiB segs"    256   16384 2048  16384  100663296  400000  $wg 5  20480 8
run "cfg4 1:1, scattered 128-B lines, 16-KiB units, 2-KiB segs"    128   16384 2048  16384  100663296  400000  $wg 5
run "cfg4 1:1, walk of 8 slabs, tile 8 x 256 B, 8-KiB segs"        256   16384 8192  16384  100663296  400000  $wg 5  20480 8 52428800
done
cat $out/ceilings.jsonl
export TMPDIR=/tmp
(cd /tmp && rocprofv3 -L > $OLDPWD/$out/counters_avail.txt 2>&1) || true
grep -o "TCC_EA0_W[A-Z0-9_]*\|TCC_[A-Z_]*STALL[A-Z_]*\|TCC_EA0_RD[A-Z0-9_]*" $out/counters_avail.txt | sort -u > $out/tcc_names.txt
for wl in cfg4s cfg5tile; do
  echo "== $wl" >> $out/counters.txt
  bash tools/exp/pmc.sh $wl \
    "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
    "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
    "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
    "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" \
    "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" \
    "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
    "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_RDREQ_IO_CREDIT_STALL_sum" \
    "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_BUSY_sum TCC_REQ_sum" \
    "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum" >> $out/counters.txt 2>&1
done
cat $out/counters.txt
