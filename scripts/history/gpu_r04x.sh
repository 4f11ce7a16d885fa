#!/bin/bash
# round 4: conv_wino4w.hip (wide items, option wino4_wide; default off): its GPU tests, then the default bench line and the one-stream per-layer
# table with the option off / on, same box
OUT=gpurun_out/${1:-r04x}; mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wide_items" -p no:cacheprovider -s > $OUT/pytest.log 2>&1; echo "pytest exit $?"; grep -E "wide items|passed|failed" $OUT/pytest.log | tail -12
for v in 0 1; do
  timeout 200 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-host-path --option wino4_wide=$v > $OUT/bench_w$v.json 2> $OUT/bench_w$v.err; python -c "
import json,sys; d=json.loads(open('$OUT/bench_w$v.json').read().strip().splitlines()[-1]); print('wide=$v', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])"
  timeout 200 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-host-path --layers --option streams=1 --option wino4_wide=$v > $OUT/layers_w$v.json 2> $OUT/layers_w$v.txt
done
paste <(grep -E "conv_wino4" $OUT/layers_w0.txt | awk '{print $1, $2, $3}') <(grep -E "conv_wino4" $OUT/layers_w1.txt | awk '{print $2, $3}') | column -t | head -40
