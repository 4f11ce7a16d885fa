#!/bin/bash
# round 5: heat-map up-sampling + keypoint detection beside ViewpointNet (the whole path's lifting stage on two streams), on / off alternating
OUT=gpurun_out/${1:-r05q}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_fixtures.py -q -m gpu -k "lifting or full or batch or fixture or keypoint or kp" -p no:cacheprovider 2>&1 | tail -4
B="python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 30 --warmup 5"
for R in 1 2 3; do
for N in 32 16; do
  for LO in 1 0; do
    echo "== full B=$N 320x320 lift_overlap=$LO: $($B --batch $N --option lift_overlap=$LO 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")"
  done
done
done
