#!/bin/bash
OUT=gpurun_out/${1:-lf2}; mkdir -p $OUT
for wgs in 16 32 64 128 256; do
  HP3D_LIFT_WGS=$wgs timeout 300 python bench.py --gpus 1 --cpu-seconds 0 --no-host-path --batch 1 --height 240 --width 320 --steps 50 --warmup 10 --layers --option lift_fused=1 > $OUT/b1_$wgs.json 2> $OUT/b1_layers_$wgs.txt
  echo "wgs=$wgs $(grep lift_fused $OUT/b1_layers_$wgs.txt | awk '{print $3}') ms; total $(python -c "import json;print(json.load(open('$OUT/b1_$wgs.json'))['ms_per_step'])")"
done
timeout 300 python bench.py --gpus 1 --cpu-seconds 0 --no-host-path --batch 1 --height 240 --width 320 --steps 50 --warmup 10 --option lift_fused=0 > $OUT/b1_off.json 2>/dev/null; python -c "import json;print('off',json.load(open('$OUT/b1_off.json'))['ms_per_step'])"
