#!/bin/bash
# two streams vs one with conv_wino4 as the default (B = 32 320x320 and 240x320), one box
OUT=gpurun_out/${1:-w4streams}; mkdir -p $OUT
for hw in "320 320" "240 320"; do set -- $hw
  for st in auto 1; do
    timeout 300 python bench.py --gpus 1 --cpu-seconds 0 --no-host-path --batch 32 --height $1 --width $2 --steps 10 --warmup 3 --option streams=$st > $OUT/s_${1}_$st.json 2>/dev/null
    python - <<PY
import json
d=json.load(open('$OUT/s_${1}_$st.json')); print('${1}x${2} streams=$st', d['value'], d['ms_per_step'])
PY
  done
done
