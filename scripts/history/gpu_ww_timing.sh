#!/bin/bash
# NEXT round's first question for conv_wino4w.hip: where does a wave's time go at Cin = 64 (+11..17 % against conv_wino4.hip) and at Cin = 512 (-5 %)?
# Build the two diagnostic libraries HERE first:
#   bash scripts/build_variant.sh wwt conv_wino4w.hip -ffp-contract=fast -mllvm -pragma-unroll-threshold=100000 -DHP3D_WW_TIMING=1
#   bash scripts/build_variant.sh w4t conv_wino4.hip  -ffp-contract=fast -mllvm -pragma-unroll-threshold=100000 -DHP3D_W4_TIMING=1
# then: gpurun -- 'bash scripts/gpu_ww_timing.sh <tag>'   (per-launch cycle sums per wave go to stderr: ww_timing / w4_timing lines)
OUT=gpurun_out/${1:-wwt}; mkdir -p $OUT
HP3D_LIB=hand3d_amd/libhp3d_wwt.so timeout 200 python bench.py --steps 1 --warmup 1 --cpu-seconds 0 --no-host-path --option streams=1 --option wino4_wide=force > $OUT/wide.json 2> $OUT/wide.txt
HP3D_LIB=hand3d_amd/libhp3d_w4t.so timeout 200 python bench.py --steps 1 --warmup 1 --cpu-seconds 0 --no-host-path --option streams=1 > $OUT/base.json 2> $OUT/base.txt
grep -E "w._timing" $OUT/wide.txt | sort | uniq -c | sort -rn | head -30
grep -E "w._timing" $OUT/base.txt | sort | uniq -c | sort -rn | head -30
