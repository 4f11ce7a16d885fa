#!/bin/bash
# round 4: conv_wino4w.hip on EVERY eligible layer (wino4_wide=force) with a weight ring of 9 half planes (libhp3d_ring9.so) against the default, per layer, one stream
OUT=gpurun_out/${1:-r04y2}; mkdir -p $OUT
timeout 100 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-host-path --layers --option streams=1 > $OUT/layers_base.json 2> $OUT/layers_base.txt
HP3D_LIB=hand3d_amd/libhp3d_ring9.so timeout 100 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-host-path --layers --option streams=1 --option wino4_wide=force > $OUT/layers_ring9_force.json 2> $OUT/layers_ring9_force.txt
paste <(grep -E "conv_wino4" $OUT/layers_base.txt | awk '{print $1, $3}') <(grep -E "conv_wino4" $OUT/layers_ring9_force.txt | awk '{print $2, $3}') | head -30
