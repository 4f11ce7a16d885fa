#!/bin/bash
# round 5, second visit: conv_wino7.hip (the 7x7 layers as Winograd F(4x4,4x4)) -- its GPU tests, the driver's command with per-layer table,
# the nine-block form against it at the bench shape and at B = 16 / 24 (where the item count no longer fills the chip)
OUT=gpurun_out/${1:-r05b}; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s -k "7x7 or batch32_winograd_active or fc_vs or pose3d or posenet_parity or config_c1 or splitk or two_streams" > $OUT/pytest_quick.log 2>&1; echo "quick pytest exit $?"; grep -E "passed|failed|7x7 as four|B=32 320x320" $OUT/pytest_quick.log | tail -12
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --layers > $OUT/bench.json 2> $OUT/bench_layers.txt ) 2> $OUT/bench_time.txt; tail -3 $OUT/bench_time.txt
python - <<PY
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], 'epe', d['epe_vs_oracle'], 'other_configs wall', d.get('other_configs_wall_s'))
for c in d.get('other_configs', []):
    print('  ', {k: c.get(k) for k in ('config', 'images_per_s', 'ms_per_step', 'dominant_family', 'executed_frac_of_dense_peak', 'parity_spot', 'error')})
PY
grep -E "conv6_2|conv7_1|conv6_1 |fc_rel0|fc_vp0|conv4_7" $OUT/bench_layers.txt
run() { timeout 200 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-host-path --no-other-configs "$@" 2> $OUT/err.txt | tail -1 > $OUT/line.json; python -c "import json; d=json.loads(open('$OUT/line.json').read()); print('$*', d['value'], d['ms_per_step'], d['roofline']['frac'])"; }
run --option wino7=0
run --option wino7=auto
run --batch 16 --option wino7=0
run --batch 16 --option wino7=1
run --batch 24 --option wino7=0
run --batch 24 --option wino7=1
run --batch 12 --option wino7=0
run --batch 12 --option wino7=1
