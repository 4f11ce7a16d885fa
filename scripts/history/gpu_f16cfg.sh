#!/bin/bash
OUT=gpurun_out/${1:-f16cfg}; mkdir -p $OUT
for C in "" 6; do
  for S in "480 640" "320 320"; do
    set -- $S
    HP3D_CONV_CFG=$C timeout 300 python bench.py --gpus 1 --steps 4 --warmup 1 --layers --cpu-seconds 0 --no-host-path --option streams=1 --dtype f16 --batch 32 --height $1 --width $2 > $OUT/b_$C_$1.json 2> $OUT/b_$C_$1.txt
    python - <<PY
import json
r=json.load(open("$OUT/b_$C_$1.json")); print("cfg '$C' $1x$2:", r["value"], "img/s", r["ms_per_step"], "ms; conv_mfma TF", r["roofline"]["achieved_algorithmic"])
PY
    grep -E "HandSegNet/conv(1_1|1_2|2_1|2_2|3_2|4_2|5_2) |PoseNet2D/conv(4_2|6_2) " $OUT/b_$C_$1.txt
  done
done
HP3D_CONV_CFG=6 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "f16" -p no:cacheprovider 2>&1 | tail -3
