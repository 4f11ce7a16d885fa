#!/bin/bash
# round 5: kp_detect with a padded LDS row pitch (8-way bank conflicts before): tests, per-layer row, bench line
OUT=gpurun_out/${1:-r05v}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_fixtures.py -q -m gpu -k "keypoint or detect or kp or full or batch or fixture" -p no:cacheprovider 2>&1 | tail -3
python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 20 --warmup 5 --layers > $OUT/b.json 2> $OUT/b.txt
echo "== $(python -c "import json; d=json.load(open('$OUT/b.json')); print(d['ms_per_step'], d['value'])")"; grep -E "^kp_|^seg_up|^mask_grow" $OUT/b.txt
for R in 1 2; do python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 30 --warmup 5 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['value'])"; done
