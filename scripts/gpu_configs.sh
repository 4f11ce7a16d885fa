#!/bin/bash
# BASELINE.json configs on one GPU (bench lines only; parity for these shapes is in tests/test_gpu_parity.py)
OUT=gpurun_out/${1:-cfg}; mkdir -p $OUT
run() { tag=$1; shift; timeout 400 python bench.py --gpus 1 --cpu-seconds 0 --no-host-path "$@" 2>$OUT/$tag.err | tail -1 > $OUT/$tag.json; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$tag.json').read())
    print('$tag', d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['config'].get('workload'), d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['frac'])
except Exception as e:
    print('$tag FAILED', e)
PY
}
run c2_posenet_b1 --workload posenet --batch 1 --steps 50 --warmup 10
run c3_b32_320 --batch 32 --steps 10 --warmup 3
run c1_b1_240x320 --batch 1 --height 240 --width 320 --steps 50 --warmup 10
run c5_f16_b128_480x640 --dtype f16 --batch 128 --height 480 --width 640 --steps 3 --warmup 1
run f32_b32_480x640 --batch 32 --height 480 --width 640 --steps 3 --warmup 1
run f16_b32_320 --dtype f16 --batch 32 --steps 10 --warmup 3
run c4_b32_240x320 --batch 32 --height 240 --width 320 --steps 10 --warmup 3
run f32_b8_240x320 --batch 8 --height 240 --width 320 --steps 20 --warmup 5
