#!/bin/bash
# round 6 item 2: the batch sweep of the final policy (both trunks on the tail-aware plan, two streams from B = 40 / 2.4 M pixels per half)
OUT=gpurun_out/${1:-r06e}; mkdir -p $OUT
for HW in "320 320" "240 320"; do
  set -- $HW
  for N in 8 12 16 24 32 40 48 64; do
    python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 20 --warmup 5 --batch $N --height $1 --width $2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1x$2 B=$N', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'])"
  done
done | tee $OUT/sweep.txt
