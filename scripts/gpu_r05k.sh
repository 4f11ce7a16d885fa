#!/bin/bash
# round 5: the unfused lifting stage's two towers on two streams (option lift_overlap).  New GPU test + the lifting tests, then the bench
# line at B = 32 / 16 / 8 with the option on and off (alternating, same box), and the lifting stage alone (hp3d_pose3d from host buffers).
OUT=gpurun_out/${1:-r05k}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "lifting or pose3d or lift or batch or full" -p no:cacheprovider 2>&1 | tail -5
B="python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 30 --warmup 5"
for R in 1 2; do
for N in 32 16 8; do
  for LO in 1 0; do
    echo "== full B=$N 320x320 lift_overlap=$LO"; $B --batch $N --option lift_overlap=$LO 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  done
done
done
python - <<'PY'
import time, numpy as np, sys
sys.path.insert(0, '.')
from hand3d_amd import synth
from hand3d_amd._lib import Engine
e = Engine(0); e.load_weight_dict(synth.make_weights()); e.finalize_weights()
rng = np.random.default_rng(0)
for B in (32, 16, 8):
    sm = np.maximum(rng.standard_normal((B, 32, 32, 21)).astype(np.float32), 0) * 0.3
    hs = synth.hand_sides(B)
    for lo in ('1', '0', '1', '0'):
        e.set_option('lift_overlap', lo)
        for _ in range(20): e.pose3d(sm, hs)
        t = time.perf_counter()
        for _ in range(200): e.pose3d(sm, hs)
        print('pose3d from host buffers B=%d lift_overlap=%s: %.3f ms per call' % (B, lo, (time.perf_counter() - t) / 200 * 1e3))
PY
