#!/bin/bash
# round 4, call K: epilogue fast path (base) against the previous epilogue (nofp), the 12-operation input transform (bt12), scalar transforms (scalar)
OUT=gpurun_out/${1:-r04k}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "f4x4 or wino4 or tail_pieces or batch32" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
HP3D_LIB=hand3d_amd/libhp3d_bt12.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "f4x4 or wino4 or tail_pieces or batch32" -p no:cacheprovider > $OUT/pytest_bt12.log 2>&1; echo "pytest bt12 exit $?"; tail -3 $OUT/pytest_bt12.log
bash scripts/gpu_r04a.sh ${1:-r04k}/ab nofp bt12 scalar
