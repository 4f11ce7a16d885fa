#!/bin/bash
# round 4, call G: (1) one vs two streams now that tail pieces remove the last-round quantisation, over batch sizes and resolutions;
# (2) the lifting stage as one launch at B = 32; (3) hipGraph replay at B = 32; (4) the config-5 fixture test with its printed errors
OUT=gpurun_out/${1:-r04g}; mkdir -p $OUT
run() { # name, args...
  n=$1; shift
  timeout 300 python bench.py --gpus 1 --cpu-seconds 0 --no-host-path --steps 10 --warmup 3 "$@" > $OUT/$n.json 2> $OUT/$n.err
  python - <<PY
import json
try:
    d=json.load(open('$OUT/$n.json')); print('%-28s %8.1f img/s %8.3f ms/step'%('$n', d['value'], d['ms_per_step']))
except Exception as e: print('$n FAILED', e)
PY
}
for cfg in "32 320 320" "16 320 320" "64 320 320" "32 240 320" "32 480 640" "8 320 320" "24 320 320"; do set -- $cfg
  for st in 1 2; do run s_B$1_$2x$3_st$st --batch $1 --height $2 --width $3 --option streams=$st; done
done
run lf1_st1 --option streams=1 --option lift_fused=1
run lf0_st1 --option streams=1 --option lift_fused=0
run graph_st1 --option streams=1 --graph
run f16_B128_st1 --dtype f16 --batch 128 --height 480 --width 640 --option streams=1
run f16_B128_st2 --dtype f16 --batch 128 --height 480 --width 640 --option streams=2
timeout 600 python -m pytest tests/test_gpu_c5_fixture.py -m gpu -q -x -s -p no:cacheprovider > $OUT/pytest_c5.log 2>&1; echo "pytest c5 exit $?"; grep -E "C5 480|passed|failed" $OUT/pytest_c5.log
