#!/bin/bash
# round 6 item 6: the lifting stage -- fc_tail (the tail of a tower as one launch) and conv_s2_gemm (the towers' last stride-2 layer as split-K GEMM):
# parity tests, hp3d_pose3d wall time per option, event-timed rows, bench line A/B
OUT=gpurun_out/${1:-r06g}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "pose3d or fc_vs or lift or poseprior" -p no:cacheprovider > $OUT/pytest_lift.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_lift.log
python - <<'PY' | tee $OUT/pose3d_wall.txt
import time, numpy as np
from hand3d_amd import Engine, synth
e = Engine(0); e.load_weight_dict(synth.make_weights()); e.finalize_weights()
rng = np.random.default_rng(5)
for B in (8, 16, 32):
    sm = (rng.standard_normal((B, 32, 32, 21)) * 0.3).astype(np.float32); hs = synth.hand_sides(B)
    for opts in ({'fc_tail': '0', 'tiny_gemm': '0'}, {'fc_tail': '1', 'tiny_gemm': '0'}, {'fc_tail': '0', 'tiny_gemm': '1'}, {'fc_tail': '1', 'tiny_gemm': '1'}):
        for k, v in opts.items(): e.set_option(k, v)
        for _ in range(5): o = e.pose3d(sm, hs)
        t = time.perf_counter()
        for _ in range(50): o = e.pose3d(sm, hs)
        print('B=%d %s hp3d_pose3d %.3f ms' % (B, opts, (time.perf_counter() - t) / 50 * 1e3))
PY
for OPT in "--option fc_tail=0 --option tiny_gemm=0" ""; do
  TAG=$( [ -z "$OPT" ] && echo new || echo old )
  for r in 1 2; do
  timeout 300 python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 20 --warmup 5 --layers $OPT > $OUT/bench_$TAG.json 2> $OUT/layers_$TAG.txt
  python -c "
import json; d=json.loads(open('$OUT/bench_$TAG.json').read().strip().splitlines()[-1]); print('$TAG', d['value'], d['value_min'], d['value_max'], d['ms_per_step'])"
  done
done
grep -E "PosePrior|ViewpointNet" $OUT/layers_new.txt
