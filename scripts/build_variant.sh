#!/bin/bash
# build_variant.sh <name> <extra hipcc flags...>: hand3d_amd/libhp3d_<name>.so with conv_wino.hip recompiled under the flags
# (timing ablations / tuning variants; run with HP3D_LIB=hand3d_amd/libhp3d_<name>.so)
N=$1; shift
C=hand3d_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-result -Wno-unused-value"
/opt/rocm/bin/hipcc $F "$@" -c $C/conv_wino.hip -o /tmp/conv_wino_$N.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o hand3d_amd/libhp3d_$N.so $C/conv_mfma.o /tmp/conv_wino_$N.o $C/conv_first.o $C/glue.o $C/engine.o && echo built hand3d_amd/libhp3d_$N.so
