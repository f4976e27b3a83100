#!/bin/bash
# Does the config-4 shape's rate depend on the row pitches?  Y rows of HEALPix 1024 in f64 are exactly 96 MiB apart
# (12 * 4^10 * 8 B = 3 * 2^25), X rows of n1280 in f32 exactly 50 MiB (25 * 2^21): batch rows j and j + 1 of one block
# fall on the same channel / bank bits.  Same shape with skewed pitches, three process launches each (placement noise).
bin=$(dirname "$0")/ceiling
[ -x "$bin" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "$(dirname "$0")/ceiling.hip" -o "$bin" || exit 1
for rep in 1 2 3; do
for spec in "100663296 52428800 exact" "100665344 52428800 Y+2KiB" "100669440 52428800 Y+6KiB" "100663296 52429056 X+256B" "100669440 52432896 Y+6KiB,X+4KiB" "100794368 52494336 Y+128KiB,X+64KiB" "101187584 52691968 Y+512KiB+2KiB,X+257KiB"; do
  set -- $spec
  "$bin" 256 16384 2048 16384 $1 400000 8 10 20480 8 $2 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s Y pitch %d X pitch %d  %.0f GB/s' % ('$3', d['write_stride'], d['slab_pitch'], d['total_GBs']))"
done
done
