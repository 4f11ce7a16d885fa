#!/bin/bash
# conv_wino2 round 2: parity (incl. in-launch split-K reduction), full-pipeline tests, B = 1 configs with wino2 = 0 / auto, B = 32 check
OUT=gpurun_out/${1:-w2b}; mkdir -p $OUT

timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -s -k "two_workgroups or full_pipeline or posenet_parity or handsegnet_parity or two_streams or batch32" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; grep -E "passed|failed|Error|assert" $OUT/pytest.log | tail -8
for w in 0 auto; do
  timeout 300 python bench.py --gpus 1 --cpu-seconds 0 --no-host-path --workload posenet --batch 1 --steps 50 --warmup 10 --layers --option wino2=$w > $OUT/c2_$w.json 2> $OUT/c2_layers_$w.txt
  timeout 300 python bench.py --gpus 1 --cpu-seconds 0 --no-host-path --batch 1 --height 240 --width 320 --steps 50 --warmup 10 --layers --option wino2=$w > $OUT/c1_$w.json 2> $OUT/c1_layers_$w.txt
  timeout 300 python bench.py --gpus 1 --cpu-seconds 0 --no-host-path --batch 8 --height 240 --width 320 --steps 20 --warmup 5 --option wino2=$w > $OUT/b8_$w.json 2> /dev/null
  timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --cpu-seconds 0 --no-host-path --option wino2=$w > $OUT/b32_$w.json 2> /dev/null
  python - <<PY
import json
for t in ('c2','c1','b8','b32'):
    d=json.load(open('$OUT/%s_$w.json'%t)); print(t,'wino2=$w', d['value'], d['ms_per_step'])
PY
done
