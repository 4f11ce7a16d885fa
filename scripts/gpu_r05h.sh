#!/bin/bash
# round 5, eighth visit: conv_first strip length, conv_wino4's pooled epilogue with the whole-tile fast path
OUT=gpurun_out/${1:-r05h}; mkdir -p $OUT
bash scripts/micro/r05_variants/cf_strips.sh > $OUT/build1.log 2>&1 &
bash scripts/micro/r05_variants/w4_poolfull.sh > $OUT/build2.log 2>&1 &
wait
run() { tag=$1; shift; timeout 200 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-host-path --no-other-configs --layers "$@" > $OUT/$tag.json 2> $OUT/$tag.txt; python -c "
import json
d=json.loads(open('$OUT/$tag.json').read().strip().splitlines()[-1])
rows={l.split()[0]: l.split()[2] for l in open('$OUT/$tag.txt') if l.startswith(('HandSegNet/conv1','PoseNet2D/conv1','HandSegNet/conv2_2','HandSegNet/conv3_4'))}
print('$tag', d['value'], d['ms_per_step'], rows)"; }
run base
HP3D_LIB=hand3d_amd/libhp3d_cfper10.so run cfper10
HP3D_LIB=hand3d_amd/libhp3d_cfper5.so run cfper5
HP3D_LIB=hand3d_amd/libhp3d_w4pf.so run w4pf
run base2
HP3D_LIB=hand3d_amd/libhp3d_w4pf.so run w4pf2
