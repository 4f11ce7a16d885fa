#!/bin/bash
# round 4, call P: timing ablations on the pair-interleaved kernel (wrong results on purpose): window addresses of a channel-blocked tensor
# [B][H][C/16][W][16] (every fetched line fully used), windows from one hot 18 KB region
OUT=gpurun_out/${1:-r04p}; mkdir -p $OUT
bash scripts/gpu_r04a.sh ${1:-r04p}/ab cblk hot
