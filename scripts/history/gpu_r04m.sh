#!/bin/bash
# round 4, call M: where a conv_wino4 step waits -- shader-clock intervals summed per wave (diagnostic build -DHP3D_W4_TIMING=1), real and hot windows
OUT=gpurun_out/${1:-r04m}; mkdir -p $OUT
for lib in timing timinghot; do
  for cfg in "32 64 64 256 256 0 3" "32 256 256 64 64 1 3" "32 32 32 512 512 0 3" "32 80 80 256 256 0 3" "32 40 40 512 512 0 3" "32 32 32 128 128 0 7"; do
    HP3D_LIB=hand3d_amd/libhp3d_$lib.so timeout 120 python scripts/conv_probe.py $cfg wino4 2>&1 | grep w4_timing | tail -1 | sed -e "s/^/$lib /"
  done
done | tee $OUT/w4_timing.txt
