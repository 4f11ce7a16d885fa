#!/bin/bash
# Round-2 GPU visit C: dual-stream execution inside the engine
TAG=${1:-r02c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "two_streams or batch32 or batch_and_2d or determinism or c5_shape or uint8 or hipgraph" > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed|FAILED|Error" $OUT/pytest_gpu.log | tail -10
b() { tag=$1; shift; timeout 600 python bench.py --gpus 1 --cpu-seconds 0 --no-host-path "$@" > $OUT/$tag.json 2> $OUT/$tag.txt; echo "$tag exit $?"; python - <<PY
import json
try:
    r = json.load(open("$OUT/$tag.json")); print("$tag", r["value"], r["unit"], r["ms_per_step"], "ms/step", r["roofline"]["kernel"], r["roofline"]["frac"])
except Exception as e: print("$tag: no result", e)
PY
}
b s_auto_320 --steps 10 --warmup 3
b s1_320 --steps 10 --warmup 3 --option streams=1
b s_auto_240 --steps 10 --warmup 3 --height 240 --width 320
b s1_240 --steps 10 --warmup 3 --height 240 --width 320 --option streams=1
b s2_480 --steps 4 --warmup 1 --height 480 --width 640 --option streams=2
b s1_480 --steps 4 --warmup 1 --height 480 --width 640 --option streams=1
b s2_f16 --steps 8 --warmup 2 --dtype f16 --option streams=2
b s1_f16 --steps 8 --warmup 2 --dtype f16 --option streams=1
b s2_b16 --steps 10 --warmup 3 --batch 16 --option streams=2
b s1_b16 --steps 10 --warmup 3 --batch 16 --option streams=1
b s2_b64 --steps 6 --warmup 2 --batch 64 --option streams=2
b s1_b64 --steps 6 --warmup 2 --batch 64 --option streams=1
