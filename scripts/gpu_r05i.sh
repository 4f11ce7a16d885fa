#!/bin/bash
# round 5: SQ and L2 counters of one 7x7 layer (32 x 32x32, 128 -> 128: PoseNet2D conv6_2 at the bench shape) under the nine-block form on
# conv_wino4.hip (split in two + reduce) and under conv_wino7.hip  -> profiles/r05_sq_counters.md
OUT=${1:-r05i}
bash scripts/gpu_w2pmc.sh $OUT/sq 32 32 32 128 128 0 7 wino4,wino7 > gpurun_out/$OUT.sq.txt 2>&1; tail -12 gpurun_out/$OUT.sq.txt
bash scripts/gpu_w4tcc.sh $OUT/tcc 32 32 32 128 128 0 7 wino4,wino7 > gpurun_out/$OUT.tcc.txt 2>&1; tail -12 gpurun_out/$OUT.tcc.txt
