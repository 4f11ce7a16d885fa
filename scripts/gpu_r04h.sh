#!/bin/bash
# round 4, call H: forked last stage (option fork_tail) -- bit-identity test, then the default bench with the fork on / off
OUT=gpurun_out/${1:-r04h}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_c5_fixture.py -m gpu -q -x -k "fork_tail or two_streams or c5 or batch32" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
for cfg in "default:" "fork0:--option fork_tail=0" "default_again:" "B16:--batch 16" "B16_fork0:--batch 16 --option fork_tail=0" "b32_240:--height 240" "b32_240_fork0:--height 240 --option fork_tail=0"; do
  n=${cfg%%:*}; o=${cfg#*:}
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 3 --cpu-seconds 0 --no-host-path $o > $OUT/bench_$n.json 2> $OUT/bench_$n.err
  python - <<PY
import json
try:
    d=json.load(open('$OUT/bench_$n.json')); print('%-16s %8.1f img/s %8.3f ms/step  frac %.4f'%('$n', d['value'], d['ms_per_step'], d['roofline']['frac']))
except Exception as e: print('$n FAILED', e)
PY
done
