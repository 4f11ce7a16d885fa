#!/bin/bash
# build_variant.sh <name> <file.hip> <extra hipcc flags...>: hand3d_amd/libhp3d_<name>.so with ONE kernel source recompiled
# under the flags (timing ablations / tuning variants; run with HP3D_LIB=hand3d_amd/libhp3d_<name>.so)
N=$1; SRC=$2; shift; shift
C=hand3d_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-result -Wno-unused-value"
OBJS=""
for f in conv_mfma conv_wino conv_wino2 conv_wino4 conv_wino4s conv_wino7 conv_pw2 conv_first conv_h16 glue lift_fused engine; do
  if [ "$f.hip" == "$SRC" ]; then /opt/rocm/bin/hipcc $F "$@" -c $C/$f.hip -o /tmp/${f}_$N.o || exit 1; OBJS="$OBJS /tmp/${f}_$N.o"; else OBJS="$OBJS $C/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o hand3d_amd/libhp3d_$N.so $OBJS && echo built hand3d_amd/libhp3d_$N.so
