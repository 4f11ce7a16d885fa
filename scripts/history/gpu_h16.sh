#!/bin/bash
# conv_h16 (half-precision 3x3 trunk kernel) vs the general kernel in f16 mode
OUT=gpurun_out/${1:-h16}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "f16" -p no:cacheprovider --tb=short 2>&1 | tail -5
for M in h16 mfma; do
  for S in "32 480 640" "32 320 320" "128 480 640"; do
    set -- $S
    timeout 300 python bench.py --gpus 1 --steps 4 --warmup 1 --layers --cpu-seconds 0 --no-host-path --option streams=1 --option f16_impl=$M --dtype f16 --batch $1 --height $2 --width $3 > $OUT/b_${M}_$1_$2.json 2> $OUT/b_${M}_$1_$2.txt
    python - <<PY
import json
r=json.load(open("$OUT/b_${M}_$1_$2.json")); print("$M B=$1 $2x$3:", r["value"], "img/s", r["ms_per_step"], "ms;", r["roofline"]["kernel"], r["roofline"]["achieved_algorithmic"], r["roofline"]["frac"])
PY
    grep -E "HandSegNet/conv(1_2|2_1|2_2|3_1|3_2|4_1|4_2|5_2) |PoseNet2D/conv(4_2|4_4) " $OUT/b_${M}_$1_$2.txt
  done
done
