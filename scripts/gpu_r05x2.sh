#!/bin/bash
# round 5: the batch sweep again after the kernel-choice model learnt about tail pieces and the chunk policy became 32 + remainder;
# GPU tests that exercise several batch sizes first.
OUT=gpurun_out/${1:-r05x2}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_fixtures.py -q -m gpu -k "batch or full or fixture or micro or chunk or stream" -p no:cacheprovider 2>&1 | tail -3
for HW in "320 320" "240 320"; do
  set -- $HW
  for N in 8 12 16 20 24 28 32 40 48 64; do
    python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 30 --warmup 5 --batch $N --height $1 --width $2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1x$2 B=$N', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'])"
  done
done | tee $OUT/sweep.txt
