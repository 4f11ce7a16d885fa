#!/bin/bash
# round 4, call S: window loads four per plane over nine planes (a class inside one plane) against two per plane over eighteen (base)
OUT=gpurun_out/${1:-r04s}; mkdir -p $OUT
bash scripts/gpu_r04a.sh ${1:-r04s}/ab wpp4
