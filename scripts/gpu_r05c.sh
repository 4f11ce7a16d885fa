#!/bin/bash
# round 5, third visit: where conv_wino7.hip's time goes -- timing ablations (scripts/micro/r05_variants/w7_ablations.sh, wrong results on purpose),
# one stream, per-layer times of PoseNet2D conv6_1 (160 channels) / conv6_2 (128) and the step
OUT=gpurun_out/${1:-r05c}; mkdir -p $OUT
bash scripts/micro/r05_variants/w7_ablations.sh > $OUT/build.log 2>&1; tail -2 $OUT/build.log
for v in ${VARIANTS:-base hotw l2hot ntw rowmaj rowmajntw nowin input inputhotw}; do
  HP3D_LIB=hand3d_amd/libhp3d_w7_$v.so timeout 120 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-host-path --no-other-configs --layers > $OUT/$v.json 2> $OUT/$v.txt
  python - <<PY
import json,re
try:
    d=json.loads(open('$OUT/$v.json').read().strip().splitlines()[-1])
    rows={l.split()[0]: float(l.split()[2]) for l in open('$OUT/$v.txt') if l.startswith('PoseNet2D/conv')}
    print('%-6s step %.3f ms  %7.1f img/s | conv6_1 %.4f conv6_2 %.4f conv7_5 %.4f' % ('$v', d['ms_per_step'], d['value'], rows['PoseNet2D/conv6_1'], rows['PoseNet2D/conv6_2'], rows['PoseNet2D/conv7_5']))
except Exception as e:
    print('$v FAILED', e)
PY
done
