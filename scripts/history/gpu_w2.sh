#!/bin/bash
# conv_wino2 (two workgroups per CU): parity tests, then the same box runs the bench with wino2 = 0 / auto / 1 (B = 32) and the B = 1 configs
OUT=gpurun_out/${1:-w2}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -s -k "two_workgroups" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; grep -E "conv_wino2|passed|failed|Error" $OUT/pytest.log | tail -40
for w in 0 auto 1; do
  timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --layers --cpu-seconds 0 --no-host-path --option wino2=$w > $OUT/bench_$w.json 2> $OUT/layers_$w.txt
  python - <<PY
import json
d=json.load(open('$OUT/bench_$w.json')); print('B32 wino2=$w', d['value'], d['ms_per_step'])
PY
done
for w in 0 auto 1; do
  timeout 300 python bench.py --gpus 1 --cpu-seconds 0 --no-host-path --workload posenet --batch 1 --steps 50 --warmup 10 --layers --option wino2=$w > $OUT/c2_$w.json 2> $OUT/c2_layers_$w.txt
  timeout 300 python bench.py --gpus 1 --cpu-seconds 0 --no-host-path --batch 1 --height 240 --width 320 --steps 50 --warmup 10 --layers --option wino2=$w > $OUT/c1_$w.json 2> $OUT/c1_layers_$w.txt
  python - <<PY
import json
for t in ('c2','c1'):
    d=json.load(open('$OUT/%s_$w.json'%t)); print(t,'wino2=$w', d['value'], d['ms_per_step'])
PY
done
paste <(awk '{print $1, $2, $3}' $OUT/layers_0.txt) <(awk '{print $2, $3}' $OUT/layers_1.txt) | grep -E "conv" | head -60
