#!/bin/bash
# round 4, call R: cache-policy bits on the weight-fragment loads with the class-ordered window stream (nt = 2, sc0 = 1, sc1 = 16)
OUT=gpurun_out/${1:-r04r}; mkdir -p $OUT
bash scripts/gpu_r04a.sh ${1:-r04r}/ab baux2 baux1 baux16
