#!/bin/bash
# round 4, call F: L2 touch prefetch of the window lines (base = touch on) against touch off, a deep ring with early windows, both
OUT=gpurun_out/${1:-r04f}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "f4x4 or wino4 or tail_pieces or batch32" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
bash scripts/gpu_r04a.sh ${1:-r04f}/ab t0 r18w4 t1r18
