#!/bin/bash
# round 6, first visit of conv_wino4s.hip (split-operand F(4x4,3x3)): per-shape error table against float64 for both kernels, then the B = 32 bench
# with per-layer tables, float32-operand kernel vs option wino4_split=auto
OUT=gpurun_out/${1:-r06a}; mkdir -p $OUT
timeout 900 python tests/helpers/split_numerics.py $OUT/split_numerics.md > $OUT/split_numerics.log 2>&1; echo "numerics exit $?"; tail -22 $OUT/split_numerics.log
for OPT in "" "--option wino4_split=auto"; do
  TAG=$( [ -z "$OPT" ] && echo base || echo split )
  timeout 600 python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 10 --warmup 3 --layers $OPT > $OUT/bench_$TAG.json 2> $OUT/layers_$TAG.txt; echo "bench $TAG exit $?"
  python -c "
import json; d=json.loads(open('$OUT/bench_$TAG.json').read().strip().splitlines()[-1]); print('$TAG', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d.get('epe_vs_oracle'))"
done
grep -E "conv(2_2|3_|4_|5_)" $OUT/layers_base.txt | head -40
echo ---- split
grep -E "conv(2_2|3_|4_|5_)" $OUT/layers_split.txt | head -40
