#!/bin/bash
# round 6: conv_wino4s.hip as an option -- its GPU tests, the per-shape error table, the mask-flip count on 256 images (oracle logits cached in
# tests/helpers/_cache by `mask_flip_vs_oracle.py --make-oracle` on the build box), and the bench line with other_configs (C3-split)
OUT=gpurun_out/${1:-r06c}; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "split_operands" -p no:cacheprovider > $OUT/pytest_split.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_split.log
timeout 900 python tests/helpers/split_numerics.py $OUT/split_numerics.md > $OUT/split_numerics.log 2>&1; echo "numerics exit $?"; tail -2 $OUT/split_numerics.log
timeout 1500 python tests/helpers/mask_flip_vs_oracle.py 256 $OUT/mask_flip.md > $OUT/mask_flip.log 2>&1; echo "flip exit $?"; tail -7 $OUT/mask_flip.log
timeout 900 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-host-path > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('primary', d['value'], d['ms_per_step'], d['roofline']['frac'])
for c in d.get('other_configs', []): print('  ', c.get('config'), c.get('images_per_s'), c.get('ms_per_step'), c.get('dominant_family'), c.get('executed_frac_of_dense_peak'), c.get('error'))"
