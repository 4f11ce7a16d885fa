#!/bin/bash
# conv_wino4 policy check: full-path latency / throughput per batch size with wino4 = 0 / auto (PoseNet2D) / all, one box
OUT=gpurun_out/${1:-w4sweep}; mkdir -p $OUT
for B in 1 2 4 8 16 32; do
  for w in 0 auto all; do
    st=30; [ $B -ge 8 ] && st=10
    timeout 300 python bench.py --gpus 1 --cpu-seconds 0 --no-host-path --batch $B --height 240 --width 320 --steps $st --warmup 5 --option wino4=$w > $OUT/b${B}_$w.json 2> $OUT/b${B}_$w.err
  done
  python - <<PY
import json
r=[]
for w in ('0','auto','all'):
    try: d=json.load(open('$OUT/b${B}_%s.json'%w)); r.append('%s %.3f ms (%.0f img/s)'%(w,d['ms_per_step'],d['value']))
    except Exception as e: r.append('%s FAILED'%w)
print('B=$B 240x320:', ' | '.join(r))
PY
done
for w in 0 auto all; do
  timeout 300 python bench.py --gpus 1 --cpu-seconds 0 --no-host-path --workload posenet --batch 1 --steps 50 --warmup 10 --option wino4=$w > $OUT/c2_$w.json 2> $OUT/c2_$w.err
  python - <<PY
import json
d=json.load(open('$OUT/c2_$w.json')); print('PoseNet2D B=1 wino4=$w', d['ms_per_step'], 'ms')
PY
done
