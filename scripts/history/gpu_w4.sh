#!/bin/bash
# conv_wino4 (Winograd F(4x4,3x3)): parity tests, then the same box runs the B = 32 bench with wino4 = 0 / pose / 1 and per-layer tables
OUT=gpurun_out/${1:-w4}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -s -k "f4x4" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; grep -E "conv_wino4|passed|failed|Error" $OUT/pytest.log | tail -40
for w in 0 pose 1; do
  timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --layers --cpu-seconds ${2:-8} --no-host-path --option wino4=$w > $OUT/bench_$w.json 2> $OUT/layers_$w.txt
  python - <<PY
import json
d=json.load(open('$OUT/bench_$w.json')); print('B32 wino4=$w', d['value'], d['ms_per_step'], d.get('epe_vs_oracle'))
PY
done
paste <(awk '{print $1, $2, $3}' $OUT/layers_0.txt) <(awk '{print $2, $3}' $OUT/layers_pose.txt) <(awk '{print $2, $3}' $OUT/layers_1.txt) | grep -E "conv" | head -60
