#!/bin/bash
# quick visit: selected parity tests + the default bench line with the per-layer table.  Usage: gpu_quick3.sh <tag> [pytest -k expr]
OUT=gpurun_out/${1:-q3}; mkdir -p $OUT
K=${2:-"winograd or 7x7 or batch32 or posenet_parity"}
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "$K" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --layers --cpu-seconds 0 --no-host-path > $OUT/bench.json 2> $OUT/bench_layers.txt
echo "bench exit $?"; python - <<PY
import json
d=json.load(open('$OUT/bench.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['frac'], r['achieved_algorithmic'])
PY
grep -E "conv6_|conv7_1|conv1_2|conv2_1" $OUT/bench_layers.txt
