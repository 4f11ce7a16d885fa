#!/bin/bash
# round 6: timing ablations of conv_h16_first_kernel (variants compute wrong results on purpose): 1 = no epilogue, 2 = no patch build, 3 = neither, 4 = no main MFMAs, 8 = no window fetch
OUT=gpurun_out/r06m; mkdir -p $OUT
for L in libhp3d libhp3d_abl1 libhp3d_abl2 libhp3d_abl3 libhp3d_abl4 libhp3d_abl8; do
  HP3D_LIB=hand3d_amd/$L.so timeout 300 python bench.py --dtype f16 --batch 128 --height 480 --width 640 --steps 3 --warmup 1 --cpu-seconds 0 --no-host-path --no-other-configs --layers > /dev/null 2> $OUT/layers_$L.txt
  echo $L $(grep "HandSegNet/conv1_2\|PoseNet2D/conv1_2" $OUT/layers_$L.txt | awk '{print $3, $4}')
done
