#!/bin/bash
# round 5, fourth visit: conv_wino7.hip's weight stream (L1-hot / L2-hot / nt hint) and window issue order; conv_first.hip with nt stores
OUT=gpurun_out/${1:-r05d}; mkdir -p $OUT
bash scripts/micro/r05_variants/cf_nt.sh > $OUT/build_cf.log 2>&1 &
VARIANTS="base hotw l2hot ntw rowmaj rowmajntw nowin input inputhotw" bash scripts/gpu_r05c.sh ${1:-r05d}
wait
for lib in libhp3d.so libhp3d_cfnt.so; do
  HP3D_LIB=hand3d_amd/$lib timeout 120 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-host-path --no-other-configs --layers > $OUT/cf_$lib.json 2> $OUT/cf_$lib.txt
  python -c "
import json
d=json.loads(open('$OUT/cf_$lib.json').read().strip().splitlines()[-1])
rows=[l.split() for l in open('$OUT/cf_$lib.txt') if 'conv1_1' in l]
print('$lib', d['value'], d['ms_per_step'], [(r[0], r[2], r[4]) for r in rows])"
done
