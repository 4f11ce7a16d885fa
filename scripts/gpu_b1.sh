#!/bin/bash
OUT=gpurun_out/b1; mkdir -p $OUT
for B in 1 4 32; do
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --no-host-path --layers --workload posenet --batch $B > $OUT/pose_b$B.json 2> $OUT/pose_b$B.txt
python -c "
import json;d=json.loads(open('$OUT/pose_b$B.json').read().strip().splitlines()[-1]);print('posenet B=$B', d['value'],d['ms_per_step'],d['roofline']['achieved'])"
done
for B in 1 8; do
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --no-host-path --layers --batch $B --height 240 --width 320 > $OUT/full_b$B.json 2> $OUT/full_b$B.txt
python -c "
import json;d=json.loads(open('$OUT/full_b$B.json').read().strip().splitlines()[-1]);print('full 240x320 B=$B', d['value'],d['ms_per_step'],d['roofline']['achieved'])"
done
