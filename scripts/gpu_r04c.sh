#!/bin/bash
# round 4, call C: tail pieces (option wino4_tail) -- parity tests, then single- and two-stream bench with the option on / off,
# and the lifting stage as one launch at B = 32 (lift_fused = 1) against the per-layer kernels
OUT=gpurun_out/${1:-r04c}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tail_pieces or f4x4 or batch32 or lift_fused" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/pytest.log
timeout 600 python -m pytest tests/test_gpu_reference_fixtures.py -m gpu -q -x -s -p no:cacheprovider > $OUT/pytest_ref.log 2>&1; echo "pytest ref exit $?"; grep -E "passed|failed|batch of 8|conv_wino4 launches" $OUT/pytest_ref.log | tail -5
for cfg in "tail1_s1:--option wino4_tail=1 --option streams=1" "tail0_s1:--option wino4_tail=0 --option streams=1" "tail1_s2:--option wino4_tail=1" "tail0_s2:--option wino4_tail=0" "lf1_s1:--option lift_fused=1 --option streams=1" "lf1_s2:--option lift_fused=1"; do
  n=${cfg%%:*}; o=${cfg#*:}
  timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --layers --cpu-seconds 0 --no-host-path $o > $OUT/bench_$n.json 2> $OUT/layers_$n.txt
  python - <<PY
import json
try:
    d=json.load(open('$OUT/bench_$n.json')); print('$n', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('stage_ms'))
except Exception as e: print('$n FAILED', e)
PY
done
grep -E "conv4_2|conv3_2|conv5_2|conv2_2 " $OUT/layers_tail1_s1.txt $OUT/layers_tail0_s1.txt | awk '{print $1, $2, $3, $4}'
grep -E "lift_fused|fc_rel0|conv_pose_0_1" $OUT/layers_lf1_s1.txt | head -5
