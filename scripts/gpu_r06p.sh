#!/bin/bash
# round 6: conv_h16_first_kernel variants on one box: per-layer time of the fused first block (HandSegNet / PoseNet2D) + the whole C5 step
for L in "$@"; do
  HP3D_LIB=hand3d_amd/$L.so timeout 300 python bench.py --dtype f16 --batch 128 --height 480 --width 640 --steps 5 --warmup 2 --cpu-seconds 0 --no-host-path --no-other-configs --layers > /tmp/b_$L.json 2> /tmp/l_$L.txt
  echo $L $(python -c "import json; c=json.loads(open('/tmp/b_$L.json').read()); print(c['value'], c['ms_per_step'])") $(grep "HandSegNet/conv1_2\|PoseNet2D/conv1_2" /tmp/l_$L.txt | awk '{print $3}')
done
