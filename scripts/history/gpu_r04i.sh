#!/bin/bash
# round 4, call I: small batches -- hipGraph replay against plain launches (config 1 / config 2 shapes, B = 1, 2, 4)
OUT=gpurun_out/${1:-r04i}; mkdir -p $OUT
run() { n=$1; shift
  timeout 300 python bench.py --gpus 1 --cpu-seconds 0 --no-host-path --steps 200 --warmup 20 "$@" > $OUT/$n.json 2> $OUT/$n.err
  python - <<PY
import json
try:
    d=json.load(open('$OUT/$n.json')); print('%-24s %8.1f img/s %8.3f ms/step replays %s'%('$n', d['value'], d['ms_per_step'], d['config'].get('hipgraph_replays')))
except Exception as e: print('$n FAILED', e)
PY
}
for g in "" "--graph"; do
  t=plain; [ -n "$g" ] && t=graph
  run c2_b1_$t --workload posenet --batch 1 $g
  run c1_b1_$t --batch 1 --height 240 --width 320 $g
  run full_b2_$t --batch 2 --height 240 --width 320 $g
  run full_b4_$t --batch 4 --height 240 --width 320 $g
  run c2_b4_$t --workload posenet --batch 4 $g
done
