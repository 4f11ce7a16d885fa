#!/bin/bash
# timing ablations of conv_wino (wrong results on purpose): which stall is worth how much?
OUT=gpurun_out/${1:-abl}; mkdir -p $OUT
for V in "" _abl1 _abl2 _abl3; do
  L=hand3d_amd/libhp3d$V.so
  HP3D_LIB=$PWD/$L timeout 300 python bench.py --gpus 1 --steps 6 --warmup 2 --layers --cpu-seconds 0 --no-host-path --option streams=1 > $OUT/b$V.json 2> $OUT/b$V.txt
  python - <<PY
import json
r=json.load(open("$OUT/b$V.json")); print("lib '$V':", r["value"], "img/s", r["ms_per_step"], "ms; wino alg TF", r["roofline"]["achieved_algorithmic"])
PY
  grep -E "HandSegNet/conv(1_2|2_1|3_2|4_2) |PoseNet2D/conv6_2" $OUT/b$V.txt
done
