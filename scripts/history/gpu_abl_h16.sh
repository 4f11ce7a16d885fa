#!/bin/bash
OUT=gpurun_out/${1:-ablh16}; mkdir -p $OUT
for V in "" _nt1 _nt2 _nt3; do
  HP3D_LIB=$PWD/hand3d_amd/libhp3d$V.so timeout 300 python bench.py --gpus 1 --steps 4 --warmup 1 --layers --cpu-seconds 0 --no-host-path --option streams=1 --dtype f16 --batch 32 --height 480 --width 640 > $OUT/b$V.json 2> $OUT/b$V.txt
  python - <<PY
import json
r=json.load(open("$OUT/b$V.json")); print("lib '$V':", r["value"], "img/s", r["ms_per_step"], "ms;", r["roofline"]["kernel"], r["roofline"]["achieved_algorithmic"])
PY
  grep -E "HandSegNet/conv(1_2|2_1|2_2|3_1|3_2|4_2) " $OUT/b$V.txt
done
