#!/bin/bash
OUT=gpurun_out/${1:-lf4}; mkdir -p $OUT
for abl in 10 11 12 13 14 15 16 17 18; do
  HP3D_LIFT_ABL=$abl HP3D_LIFT_WGS=128 timeout 300 python bench.py --gpus 1 --cpu-seconds 0 --no-host-path --batch 1 --height 240 --width 320 --steps 50 --warmup 10 --layers --option lift_fused=1 > $OUT/b1.json 2> $OUT/b1_layers.txt
  echo "phases 0..$((abl-10)) $(grep lift_fused $OUT/b1_layers.txt | awk '{print $3}') ms"
done
