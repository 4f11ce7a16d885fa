#!/bin/bash
# per-layer times of both Winograd kernels at small batches (one stream): which layer shapes should conv_wino2 take?
OUT=gpurun_out/${1:-w2sweep}; mkdir -p $OUT
for B in 2 4 8 16; do
 for w in 0 1; do
  timeout 300 python bench.py --gpus 1 --cpu-seconds 0 --no-host-path --batch $B --height 240 --width 320 --steps 20 --warmup 5 --layers --option wino2=$w --option streams=1 > $OUT/b${B}_$w.json 2> $OUT/b${B}_layers_$w.txt
  python - <<PY
import json
d=json.load(open('$OUT/b${B}_$w.json')); print('B=$B wino2=$w', d['value'], d['ms_per_step'])
PY
 done
done
