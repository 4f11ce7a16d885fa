#!/bin/bash
# round 6 item 3: config 5 (f16, B = 128 at 480x640), XCD-affine item order of conv_h16 (tree) against the old cout-block-innermost order (variant h16noaff), alternating
for r in 1 2; do
  for LIB in hand3d_amd/libhp3d_h16noaff.so hand3d_amd/libhp3d.so; do
    HP3D_LIB=$LIB timeout 300 python bench.py --dtype f16 --batch 128 --height 480 --width 640 --steps 10 --warmup 2 --cpu-seconds 0 --no-host-path --no-other-configs 2>/dev/null | python -c "
import sys,json; c=json.loads(sys.stdin.read()); print('$LIB', c['value'], c['value_min'], c['value_max'], c['ms_per_step'], c['roofline']['frac'])"
  done
done
