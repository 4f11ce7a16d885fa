#!/bin/bash
# timing ablations of the f16 direct kernel (wrong results on purpose) + the conv_first f16 variant
OUT=gpurun_out/${1:-ablf16}; mkdir -p $OUT
for V in "" _m1 _m2 _m4 _m8 _m15; do
  L=hand3d_amd/libhp3d$V.so
  HP3D_LIB=$PWD/$L timeout 300 python bench.py --gpus 1 --steps 4 --warmup 1 --layers --cpu-seconds 0 --no-host-path --option streams=1 --dtype f16 --batch 32 --height 480 --width 640 > $OUT/b$V.json 2> $OUT/b$V.txt
  python - <<PY
import json
r=json.load(open("$OUT/b$V.json")); print("lib '$V':", r["value"], "img/s", r["ms_per_step"], "ms; conv_mfma TF", r["roofline"]["achieved_algorithmic"], r["roofline"]["kernel"])
PY
  grep -E "HandSegNet/conv(1_1|1_2|2_1|3_2|4_2) |PoseNet2D/conv6_2" $OUT/b$V.txt
done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "f16" -p no:cacheprovider 2>&1 | tail -3
