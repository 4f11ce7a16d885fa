#!/bin/bash
# round 6, config 5's fused first block: the filter-resident form (conv_h16_first_kernel) against the ring form, same box, alternating; parity first
OUT=gpurun_out/r06l; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_c5_fixture.py -x -q -m gpu -k "f16 or c5" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
for r in 1 2; do
  for F in ring resident; do
    timeout 300 python bench.py --dtype f16 --batch 128 --height 480 --width 640 --steps 10 --warmup 2 --cpu-seconds 0 --no-host-path --no-other-configs --option f16_fuse12=$F 2>/dev/null | python -c "
import sys,json; c=json.loads(sys.stdin.read()); print('$F', c['value'], c['value_min'], c['value_max'], c['ms_per_step'], c['roofline']['frac'])"
  done
done
timeout 300 python bench.py --dtype f16 --batch 128 --height 480 --width 640 --steps 5 --warmup 2 --cpu-seconds 0 --no-host-path --no-other-configs --layers > $OUT/bench_c5.json 2> $OUT/layers_c5.txt
grep "conv1_2\|conv2_1 " $OUT/layers_c5.txt
# the f16 trunks at the primary shape (B = 32, 320 x 320: 12800 items, 50 per CU)
for F in ring resident; do
  timeout 300 python bench.py --dtype f16 --steps 20 --warmup 5 --cpu-seconds 0 --no-host-path --no-other-configs --option f16_fuse12=$F 2>/dev/null | python -c "
import sys,json; c=json.loads(sys.stdin.read()); print('B32 320 $F', c['value'], c['ms_per_step'])"
done
