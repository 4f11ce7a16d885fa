#!/bin/bash
# round 4, call E: (1) LDS-DMA micro-benchmark with the lean piece form; (2) round-3 kernel with each plane's eight MFMAs split into two
# statements and the plane's loads between them (libhp3d_split.so) against the base: parity subset, then per-layer single-stream tables
OUT=gpurun_out/${1:-r04e}; mkdir -p $OUT
timeout 120 scripts/micro/dma_rows > $OUT/dma_rows.txt 2>&1; echo "dma_rows exit $?"; cat $OUT/dma_rows.txt
HP3D_LIB=hand3d_amd/libhp3d_split.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "f4x4 or wino4 or tail_pieces or batch32" -p no:cacheprovider > $OUT/pytest_split.log 2>&1; echo "pytest split exit $?"; tail -3 $OUT/pytest_split.log
bash scripts/gpu_r04a.sh ${1:-r04e}/ab split
