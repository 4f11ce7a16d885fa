#!/bin/bash
# quick GPU iteration: conv parity + bench line + per-layer table.  Usage: gpu_quick.sh tag [pytest -k expr]
TAG=${1:-q}; K=${2:-conv}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "$K" > $OUT/pytest.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed|FAILED" $OUT/pytest.log | tail -8
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --layers --cpu-seconds 0 --no-host-path > $OUT/bench.json 2> $OUT/bench_layers.txt
echo "bench exit $?"; python - <<PY
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['roofline'])
PY
