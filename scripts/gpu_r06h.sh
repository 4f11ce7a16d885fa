#!/bin/bash
# round 6: whole GPU suite + the bench line (three times, for the spread)
OUT=gpurun_out/${1:-r06h}; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest_gpu.log
for r in 1 2 3; do
  timeout 300 python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['value_min'], d['value_max'], d['ms_per_step'], d['roofline']['frac'])"
done
