#!/bin/bash
# round 5, first visit: (1) the tests around this round's host / FC changes, (2) the driver's command with the new other_configs leg,
# (3) wino4_wide off / on at the bench shape, (4) the WHOLE GPU suite with wino4_wide=1
OUT=gpurun_out/${1:-r05a}; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "fc or pose3d or poseprior or lift or config_c1 or stage_timing or weight_files" > $OUT/pytest_quick.log 2>&1; echo "quick pytest exit $?"; tail -3 $OUT/pytest_quick.log
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --layers > $OUT/bench.json 2> $OUT/bench_layers.txt ) 2> $OUT/bench_time.txt; tail -3 $OUT/bench_time.txt
python - <<PY
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], 'other_configs wall', d.get('other_configs_wall_s'))
for c in d.get('other_configs', []):
    print('  ', {k: c.get(k) for k in ('config', 'images_per_s', 'ms_per_step', 'dominant_family', 'executed_frac_of_dense_peak', 'parity_spot', 'error')})
PY
for v in 0 1; do
  timeout 200 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-host-path --no-other-configs --option wino4_wide=$v > $OUT/bench_w$v.json 2> $OUT/bench_w$v.err
  python -c "import json; d=json.loads(open('$OUT/bench_w$v.json').read().strip().splitlines()[-1]); print('wide=$v', d['value'], d['ms_per_step'])"
done
HP3D_TEST_OPTIONS=wino4_wide=1 timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -rx > $OUT/pytest_gpu_wide.log 2>&1; echo "wide pytest exit $?"; tail -3 $OUT/pytest_gpu_wide.log
