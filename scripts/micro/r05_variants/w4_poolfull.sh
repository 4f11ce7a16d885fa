#!/bin/bash
# conv_wino4.hip: the pooled epilogue WITH the edge-select-free path when every tile is whole (round 4 measured it 1-3 % slower; the epilogue
# has since lost a quarter of its VALU)  -> hand3d_amd/libhp3d_w4pf.so
C=hand3d_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-result -Wno-unused-value -ffp-contract=fast -mllvm -pragma-unroll-threshold=100000 -I$C"
T=/tmp/w4var; mkdir -p $T
sed -E 's/if \(!POOL \&\& full\) store_tile\(std::true_type\{\}\);/if (full) store_tile(std::true_type{});/' $C/conv_wino4.hip > $T/conv_wino4_pf.hip
diff $C/conv_wino4.hip $T/conv_wino4_pf.hip | grep -c '^>'
/opt/rocm/bin/hipcc $F -c $T/conv_wino4_pf.hip -o $T/conv_wino4_pf.o || exit 1
OBJS=""; for f in conv_mfma conv_wino conv_wino2 conv_wino7 conv_pw2 conv_first conv_h16 glue lift_fused engine; do OBJS="$OBJS $C/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o hand3d_amd/libhp3d_w4pf.so $OBJS $T/conv_wino4_pf.o && echo built w4pf
