#!/bin/bash
# A/B of kernel builds on ONE box: parity subset on the tree's library, then the default bench and a one-stream per-layer table for the tree's
# library ("base") and for every variant library hand3d_amd/libhp3d_<name>.so (scripts/build_variant.sh <name> <file.hip> <flags>).
# Usage: gpu_ab.sh <tag> [--no-tests] <variant>...      (round 4's calls and their variants: scripts/README.md)
OUT=gpurun_out/${1:-ab}; mkdir -p $OUT; shift
if [ "$1" == "--no-tests" ]; then shift; else
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "f4x4 or wino4 or tail_pieces or batch32 or 7x7" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
fi
bash scripts/gpu_r04a.sh $(basename $OUT)/ab "$@"
