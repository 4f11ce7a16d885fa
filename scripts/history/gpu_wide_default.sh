#!/bin/bash
# NEXT round, first call: the whole GPU suite with conv_wino4w.hip switched on for the layers its mode 1 selects (the suite's engine takes the option from
# HP3D_TEST_OPTIONS, tests/conftest.py), then the default bench line with the option off / on.  Green + faster -> make "wino4_wide" default 1 in engine.hip.
OUT=gpurun_out/${1:-wide_default}; mkdir -p $OUT
HP3D_TEST_OPTIONS=wino4_wide=1 timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -rx > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_gpu.log
for v in 0 1; do
  timeout 200 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-host-path --option wino4_wide=$v > $OUT/bench_w$v.json 2> $OUT/bench_w$v.err
  python -c "import json; d=json.loads(open('$OUT/bench_w$v.json').read().strip().splitlines()[-1]); print('wide=$v', d['value'], d['ms_per_step'])"
done
