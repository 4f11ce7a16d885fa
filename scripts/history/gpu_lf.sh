#!/bin/bash
# fused lifting stage: parity test + B = 1 / 2 / 4 full-path timings with lift_fused = 0 / auto
OUT=gpurun_out/${1:-lf}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -s -k "lift_fused or pose3d or full_pipeline_parity_config_c1" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; grep -E "lift_fused B|passed|failed|Error|assert" $OUT/pytest.log | tail -10
for m in 0 auto; do
 for B in 1 2 4; do
  timeout 300 python bench.py --gpus 1 --cpu-seconds 0 --no-host-path --batch $B --height 240 --width 320 --steps 50 --warmup 10 --layers --option lift_fused=$m > $OUT/b${B}_$m.json 2> $OUT/b${B}_layers_$m.txt
  python - <<PY
import json
d=json.load(open('$OUT/b${B}_$m.json')); print('B=$B lift_fused=$m', d['value'], d['ms_per_step'])
PY
 done
done
grep -E "lift_fused|PosePrior|Viewpoint" $OUT/b1_layers_auto.txt | head
