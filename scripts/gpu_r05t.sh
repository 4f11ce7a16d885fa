#!/bin/bash
# round 5: seg_upsample_softmax timing ablations (scripts/micro/r05_variants/seg_abl.sh)
OUT=gpurun_out/${1:-r05t}
mkdir -p $OUT
for V in "" noatomic noexp nodiv gx16 gx256 nostore; do
  L=hand3d_amd/libhp3d.so; [ -n "$V" ] && L=hand3d_amd/libhp3d_seg_$V.so
  HP3D_LIB=$(pwd)/$L python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 10 --warmup 3 --layers > $OUT/b_$V.json 2> $OUT/b_$V.txt
  echo "== ${V:-shipped}: $(grep -E '^seg_upsample' $OUT/b_$V.txt)"
done
