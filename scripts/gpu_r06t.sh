#!/bin/bash
# round 6: 32 < B < 64 at 320x320 as ONE chunk (option micro_batch = B, one stream) against the default (two halves on two streams)
for N in 40 48 56; do
  for O in "" "--option micro_batch=$N --option streams=1"; do
    python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 20 --warmup 5 --batch $N $O 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('B=$N [$O]', d['ms_per_step'], d['value'], d['roofline']['frac'])"
  done
done
python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 20 --warmup 5 --batch 48 --height 240 --width 320 --option micro_batch=48 --option streams=1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('240x320 B=48 one chunk', d['ms_per_step'], d['value'])"
