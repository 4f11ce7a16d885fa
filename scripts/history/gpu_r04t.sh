#!/bin/bash
# round 4, final visit: L2 / L1 and SQ counters of conv_wino4 on PoseNet conv3_2 (B = 32) for the "after" column, the configurations table,
# then tests + rocprof stats + PMC traffic + bench line (gpu_round.sh)
OUT=gpurun_out/${1:-r04t}; mkdir -p $OUT
bash scripts/gpu_w4tcc.sh ${1:-r04t}/tcc 32 64 64 256 256 0 3 wino4 > $OUT/tcc.txt 2>&1; tail -6 $OUT/tcc.txt
bash scripts/gpu_w2pmc.sh ${1:-r04t}/sq 32 64 64 256 256 0 3 wino4 > $OUT/sq.txt 2>&1; tail -12 $OUT/sq.txt
bash scripts/gpu_configs.sh ${1:-r04t}/cfg > $OUT/cfg.txt 2>&1; cat $OUT/cfg.txt
bash scripts/gpu_round.sh ${1:-r04t}/round pmc
