#!/bin/bash
# round-3 timing ablations of conv_wino (wrong results on purpose): 4 = no transform arithmetic, 8 = no A-fragment reads, 16 = no epilogue, 28 = all three
OUT=gpurun_out/${1:-abl3}; mkdir -p $OUT
for V in "" _abl4 _abl8 _abl16 _abl28; do
  L=hand3d_amd/libhp3d$V.so
  HP3D_LIB=$PWD/$L timeout 300 python bench.py --gpus 1 --steps 6 --warmup 2 --layers --cpu-seconds 0 --no-host-path --option streams=1 --option wino2=0 > $OUT/b$V.json 2> $OUT/b$V.txt
  python - <<PY
import json
r=json.load(open("$OUT/b$V.json")); print("lib '$V':", r["value"], "img/s", r["ms_per_step"], "ms; wino alg TF", r["roofline"]["achieved_algorithmic"])
PY
  grep -E "HandSegNet/conv(1_2|2_1|3_2|4_2) |PoseNet2D/conv6_2" $OUT/b$V.txt | awk '{printf "%s %s  ", $1, $3} END {print ""}'
done
