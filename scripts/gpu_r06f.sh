#!/bin/bash
# round 6 item 3: the config-5 line (half-precision trunks, B = 128 at 480x640) with per-layer table + its parity test
OUT=gpurun_out/${1:-r06f}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_c5_fixture.py tests/test_gpu_parity.py -m gpu -q -x -k "c5 or f16" -p no:cacheprovider > $OUT/pytest_c5.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_c5.log
timeout 600 python bench.py --dtype f16 --batch 128 --height 480 --width 640 --steps 10 --warmup 2 --cpu-seconds 0 --no-host-path --no-other-configs --layers > $OUT/bench_c5_f16.json 2> $OUT/bench_layers_c5_f16.txt; echo "c5 exit $?"
python -c "
import json; c=json.loads(open('$OUT/bench_c5_f16.json').read().strip().splitlines()[-1]); print('c5', c['value'], c['value_min'], c['value_max'], c['ms_per_step'], c['roofline']['frac'])"
head -16 $OUT/bench_layers_c5_f16.txt
