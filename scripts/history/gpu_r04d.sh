#!/bin/bash
# round 4, call D: LDS-DMA row-piece micro-benchmark (semantics + issue cost beside f32 MFMAs), the "before" L2 / L1 counters of
# conv_wino4 on PoseNet conv3_2 (B = 32), then the full round visit (tests, rocprof stats, PMC traffic, bench line)
OUT=gpurun_out/${1:-r04d}; mkdir -p $OUT
timeout 120 scripts/micro/dma_rows > $OUT/dma_rows.txt 2>&1; echo "dma_rows exit $?"; cat $OUT/dma_rows.txt
bash scripts/gpu_w4tcc.sh ${1:-r04d}/tcc 32 64 64 256 256 0 3 wino4 > $OUT/tcc.txt 2>&1; tail -8 $OUT/tcc.txt
bash scripts/gpu_round.sh ${1:-r04d}/round pmc
