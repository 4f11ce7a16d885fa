#!/bin/bash
# conv_first.hip with non-temporal float32 stores (the 64-channel activation is written once and read back one layer later, after 0.8 GB
# have passed through the 256 MB memory-side cache): does the hint lift the store rate?  -> hand3d_amd/libhp3d_cfnt.so
C=hand3d_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-result -Wno-unused-value -I$C"
T=/tmp/cfvar; mkdir -p $T
sed -E 's/HP3D_BUFFER_STORE4\(orsrc, (v[01]), base, ([0-9]+)\);/HP3D_BUFFER_STORE4_NT(orsrc, \1, base, \2);/' $C/conv_first.hip > $T/conv_first_nt.hip
diff $C/conv_first.hip $T/conv_first_nt.hip | grep -c '^>'
/opt/rocm/bin/hipcc $F -c $T/conv_first_nt.hip -o $T/conv_first_nt.o || exit 1
OBJS=""; for f in conv_mfma conv_wino conv_wino2 conv_wino4 conv_wino7 conv_pw2 conv_h16 glue lift_fused engine; do OBJS="$OBJS $C/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o hand3d_amd/libhp3d_cfnt.so $OBJS $T/conv_first_nt.o && echo built cfnt
