#!/bin/bash
# round 5: batch sweep of the full path on the final tree (float32, inputs in HBM) -> profiles/r05_batch_sweep.md
OUT=gpurun_out/${1:-r05x}
mkdir -p $OUT
for HW in "320 320" "240 320"; do
  set -- $HW
  for N in 1 2 4 8 12 16 24 32 48 64; do
    S=30; [ $N -le 4 ] && S=100
    python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps $S --warmup 5 --batch $N --height $1 --width $2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1x$2 B=$N', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'])"
  done
done | tee $OUT/sweep.txt
