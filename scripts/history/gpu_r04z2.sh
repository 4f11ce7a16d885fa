#!/bin/bash
# round 4, last visit: the committed tree -- conv_wino4w's GPU tests under the final policy, smoke(), the default bench line
OUT=gpurun_out/${1:-r04z2}; mkdir -p $OUT
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "wide_items" -p no:cacheprovider -s > $OUT/pytest.log 2>&1; echo "pytest exit $?"; grep -E "whole path|passed|failed" $OUT/pytest.log | tail -4
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke exit $?"
timeout 200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])"
