#!/bin/bash
# round 5: mask_grow visiting only the rows the object can have reached; kp_detect with a thread per output column.  Tests, per-layer rows, bench.
OUT=gpurun_out/${1:-r05w}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_fixtures.py tests/test_gpu_c5_fixture.py -q -m gpu -k "mask or keypoint or detect or kp or full or batch or fixture or u8 or handseg or c5" -p no:cacheprovider 2>&1 | tail -3
python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 20 --warmup 5 --layers > $OUT/b.json 2> $OUT/b.txt
echo "== $(python -c "import json; d=json.load(open('$OUT/b.json')); print(d['ms_per_step'], d['value'])")"; grep -E "^kp_|^seg_up|^mask_grow|^crop" $OUT/b.txt
for R in 1 2; do python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 30 --warmup 5 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['value'])"; done
python bench.py --dtype f16 --batch 128 --height 480 --width 640 --steps 3 --warmup 2 --cpu-seconds 0 --no-host-path --no-other-configs --layers 2>&1 >/dev/null | grep -E "^seg_upsample|^mask_grow|^kp_"
