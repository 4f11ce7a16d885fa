#!/bin/bash
# round 4, call Q: window loads issued class by class (base) against row-major (issue0): parity subset + per-layer tables
OUT=gpurun_out/${1:-r04q}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "f4x4 or wino4 or tail_pieces or batch32 or 7x7" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
bash scripts/gpu_r04a.sh ${1:-r04q}/ab issue0
