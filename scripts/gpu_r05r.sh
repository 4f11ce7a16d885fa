#!/bin/bash
# round 5: conv1_1's read pass beside the convolution (child stream) against in front of it; per-layer rows + bench line, alternating
OUT=gpurun_out/${1:-r05r}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "cold_input or first_layer or batch" -p no:cacheprovider 2>&1 | tail -3
for R in 1 2; do
for TB in 1 0; do
  python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 20 --warmup 5 --layers --option first_touch_beside=$TB > $OUT/b32_$TB.json 2> $OUT/b32_$TB.txt
  echo "== B=32 320x320 first_touch_beside=$TB: $(python -c "import json; d=json.load(open('$OUT/b32_$TB.json')); print(d['ms_per_step'], d['value'], [ (f['achieved'], f['frac']) for f in d['roofline_other_conv'] if f['kernel'].startswith('conv_first')])")"; grep -E "conv1_1" $OUT/b32_$TB.txt
done
done
