#!/bin/bash
# round 4, call V: D[cout][tile] epilogue on whole accumulator tuples (base) against the previous build (prev)
OUT=gpurun_out/${1:-r04v}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "f4x4 or tail_pieces or batch32" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -2 $OUT/pytest.log
bash scripts/gpu_r04a.sh ${1:-r04v}/ab prev
