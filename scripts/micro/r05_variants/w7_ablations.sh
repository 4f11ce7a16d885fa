#!/bin/bash
# conv_wino7.hip timing ablations (WRONG RESULTS ON PURPOSE): sed-patched copies of the shipped source, one library each, built in parallel.
# The shipped file carries no ablation switches; this script is the record of what was measured (profiles/r05_tuning_notes.md).
#   usage: bash scripts/micro/r05_variants/w7_ablations.sh            -> hand3d_amd/libhp3d_w7_<name>.so
C=hand3d_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-result -Wno-unused-value -Wno-unused-variable -Wno-unused-but-set-variable -ffp-contract=fast -mllvm -pragma-unroll-threshold=100000 -I$C"
T=/tmp/w7var; mkdir -p $T
variant() {   # name, sed program
  sed -E "$2" $C/conv_wino7.hip > $T/conv_wino7_$1.hip
  ( /opt/rocm/bin/hipcc $F -c $T/conv_wino7_$1.hip -o $T/conv_wino7_$1.o &&
    OBJS=""; for f in conv_mfma conv_wino conv_wino2 conv_wino4 conv_pw2 conv_first conv_h16 glue lift_fused engine; do OBJS="$OBJS $C/$f.o"; done;
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o hand3d_amd/libhp3d_w7_$1.so $OBJS $T/conv_wino7_$1.o && echo built $1 ) &
}
variant base   's/^XXXX//'
variant hotw   's/HP3D_SADD\(wsb, entry_stride_b\);/(void)0;/'
variant l2hot  's/            a_bases\(cur\);/            a_bases(cur); wsb = cyoff + W7_RING * entry_stride_b; asm volatile("" : "+s"(wsb));/'
variant ntw    's/HP3D_BUFFER_LOAD16\(wrsrc, wv_lane, wsb\)/HP3D_BUFFER_LOAD16_NT(wrsrc, wv_lane, wsb)/'
variant rowmaj 's/window_load\(W7_ISSUE.e\[g\], wsoff\)/window_load(g, wsoff)/'
variant rowmajntw 's/window_load\(W7_ISSUE.e\[g\], wsoff\)/window_load(g, wsoff)/; s/HP3D_BUFFER_LOAD16\(wrsrc, wv_lane, wsb\)/HP3D_BUFFER_LOAD16_NT(wrsrc, wv_lane, wsb)/'
variant nowin  's/if \(g < W7_NP\) window_load\(W7_ISSUE.e\[g\], wsoff\);/(void)0;/'
variant input  's/if \(g < W7_NP\) window_load\(W7_ISSUE.e\[g\], wsoff\);/(void)0;/; s/if \(!lastc\) transform_arith\(\);/(void)0;/; s/if \(pl < W7_NP\) v_write\(cur \^ 1, pl\);/(void)0;/'
variant inputhotw 's/if \(g < W7_NP\) window_load\(W7_ISSUE.e\[g\], wsoff\);/(void)0;/; s/if \(!lastc\) transform_arith\(\);/(void)0;/; s/if \(pl < W7_NP\) v_write\(cur \^ 1, pl\);/(void)0;/; s/HP3D_SADD\(wsb, entry_stride_b\);/(void)0;/'
wait
