#!/bin/bash
# round 6: timing ablations / variants of conv_wino4s.hip (libraries built by scripts/build_variant.sh <name> conv_wino4s.hip ... -DHP3D_W4S_ABL=n):
# the B = 32 bench with option wino4_split=auto and the per-layer table, two layers quoted per variant
OUT=gpurun_out/${1:-r06b}; mkdir -p $OUT; shift
for V in base "$@"; do
  if [ "$V" == base ]; then LIB=hand3d_amd/libhp3d.so; else LIB=hand3d_amd/libhp3d_$V.so; fi
  HP3D_LIB=$LIB timeout 300 python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 8 --warmup 2 --layers --option wino4_split=auto > $OUT/bench_$V.json 2> $OUT/layers_$V.txt
  echo "$V: $(python -c "
import json; d=json.loads(open('$OUT/bench_$V.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>/dev/null) | $(grep -E "HandSegNet/conv4_2|PoseNet2D/conv3_2|HandSegNet/conv2_2|PoseNet2D/conv4_4 " $OUT/layers_$V.txt | awk '{printf "%s %s | ", $1, $3}')"
done
