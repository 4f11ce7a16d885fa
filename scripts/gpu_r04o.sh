#!/bin/bash
# round 4, call O: weight ring 9 (base) / 12 / 18 on the pair-interleaved plane loop; in-kernel timing at ring 18
OUT=gpurun_out/${1:-r04o}; mkdir -p $OUT
for cfg in "32 64 64 256 256 0 3" "32 256 256 64 64 1 3" "32 80 80 256 256 0 3" "32 32 32 128 128 0 7"; do
  HP3D_LIB=hand3d_amd/libhp3d_timing18.so timeout 120 python scripts/conv_probe.py $cfg wino4 2>&1 | grep w4_timing | tail -1
done | tee $OUT/w4_timing18.txt
bash scripts/gpu_r04a.sh ${1:-r04o}/ab ring12 ring18
