#!/bin/bash
# round 4, call N: the plane as four MFMA pairs with one load behind each (base) against the plane as one block (nopairs): parity subset,
# per-layer tables, in-kernel timing of the new build
OUT=gpurun_out/${1:-r04n}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "f4x4 or wino4 or tail_pieces or batch32 or 7x7" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
for cfg in "32 64 64 256 256 0 3" "32 256 256 64 64 1 3" "32 80 80 256 256 0 3" "32 32 32 128 128 0 7"; do
  HP3D_LIB=hand3d_amd/libhp3d_timing.so timeout 120 python scripts/conv_probe.py $cfg wino4 2>&1 | grep w4_timing | tail -1
done | tee $OUT/w4_timing.txt
bash scripts/gpu_r04a.sh ${1:-r04n}/ab nopairs
