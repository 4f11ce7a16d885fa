#!/bin/bash
# round 5: mask_grow with guard-word rows (no divisions in the pass loop) and two barriers per pass: mask tests, per-layer rows at the bench shape and at C5's
OUT=gpurun_out/${1:-r05s}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_fixtures.py tests/test_gpu_c5_fixture.py -q -m gpu -k "mask or full or batch or fixture or c5 or u8 or handseg" -p no:cacheprovider 2>&1 | tail -4
python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 20 --warmup 5 --layers > $OUT/b32.json 2> $OUT/b32.txt
echo "== B=32 320x320: $(python -c "import json; d=json.load(open('$OUT/b32.json')); print(d['ms_per_step'], d['value'])")"; grep -E "mask_grow|seg_upsample|crop_and" $OUT/b32.txt
python bench.py --dtype f16 --batch 128 --height 480 --width 640 --steps 3 --warmup 2 --cpu-seconds 0 --no-host-path --no-other-configs --layers > $OUT/c5.json 2> $OUT/c5.txt
echo "== C5 shard: $(python -c "import json; d=json.load(open('$OUT/c5.json')); print(d['ms_per_step'], d['value'])")"; grep -E "mask_grow|seg_upsample|crop_and" $OUT/c5.txt
python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 100 --warmup 10 --layers --batch 1 --height 240 --width 320 > $OUT/c1.json 2> $OUT/c1.txt
echo "== C1: $(python -c "import json; d=json.load(open('$OUT/c1.json')); print(d['ms_per_step'], d['value'])")"; grep -E "mask_grow|seg_upsample|crop_and" $OUT/c1.txt
