#!/bin/bash
# L2 / L1 counters of the half-precision trunk kernel (one --pmc pass per counter set), B=32 480x640 f16, streams=1.
OUT=gpurun_out/${1:-pmch16}; R=$(pwd); mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for C in "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  N=$(echo $C | tr ' ' '_')
  timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/$N -o hp3d -- python $R/bench.py --gpus 1 --steps 1 --warmup 1 --cpu-seconds 0 --no-host-path --option streams=1 --dtype f16 --batch 32 --height 480 --width 640 > /dev/null 2> $R/$OUT/${N}_stderr.txt
  echo "pmc $C exit $?"
done
# HBM traffic per launch at the config-5 per-GPU shape (B=128): separate FETCH_SIZE / WRITE_SIZE passes -> profiles/conv_h16_traffic.json
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/c5_$C -o hp3d -- python $R/bench.py --gpus 1 --steps 1 --warmup 1 --cpu-seconds 0 --no-host-path --option streams=1 --dtype f16 --batch 128 --height 480 --width 640 > /dev/null 2> $R/$OUT/c5_${C}_stderr.txt
  echo "pmc $C (B=128) exit $?"
done
cd $R
python scripts/h16_counters.py $OUT ${2:-r02}
