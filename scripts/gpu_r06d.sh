#!/bin/bash
# round 6 item 2: the reference-fixture tests with the margin-aware mask gate, then the batch sweep (both trunks on the tail-aware plan)
OUT=gpurun_out/${1:-r06d}; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_reference_fixtures.py tests/test_gpu_parity.py -m gpu -q -x -k "reference_fixtures or batch32 or f4x4_policy or micro_batch or batch_and_2d" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/pytest.log
for HW in "320 320" "240 320"; do
  set -- $HW
  for N in 8 12 16 20 24 32 40 48 64; do
    python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 20 --warmup 5 --batch $N --height $1 --width $2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1x$2 B=$N', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'])"
  done
done | tee $OUT/sweep.txt
python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 20 --warmup 5 --batch 40 --option micro_batch=32 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('320x320 B=40 as 32+8', d['ms_per_step'], d['value'])" | tee -a $OUT/sweep.txt
