#!/bin/bash
# round 4, call U: MFMA operands (weights, V) -> D[cout][tile], epilogue with 16-byte stores (base) against the previous build (prev)
OUT=gpurun_out/${1:-r04u}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_fixtures.py -m gpu -q -x -k "f4x4 or wino4 or tail_pieces or batch32 or 7x7 or reference or config_c1 or arbitrary or sizes" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
bash scripts/gpu_r04a.sh ${1:-r04u}/ab prev
