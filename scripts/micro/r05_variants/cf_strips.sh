#!/bin/bash
# conv_first.hip: how long a strip of tiles should a workgroup walk?  (shipped: the whole tile row unless that leaves fewer than 4 workgroups per CU)
#   -> hand3d_amd/libhp3d_cfper<N>.so for N in 10 5
C=hand3d_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-result -Wno-unused-value -I$C"
T=/tmp/cfvar; mkdir -p $T
for N in 10 5; do
  sed -E "s/int per = p.tiles_x;/int per = std::min(p.tiles_x, $N);/" $C/conv_first.hip > $T/conv_first_per$N.hip
  ( /opt/rocm/bin/hipcc $F -c $T/conv_first_per$N.hip -o $T/conv_first_per$N.o &&
    OBJS=""; for f in conv_mfma conv_wino conv_wino2 conv_wino4 conv_wino7 conv_pw2 conv_h16 glue lift_fused engine; do OBJS="$OBJS $C/$f.o"; done;
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o hand3d_amd/libhp3d_cfper$N.so $OBJS $T/conv_first_per$N.o && echo built per$N ) &
done
wait
