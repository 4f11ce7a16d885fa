#!/bin/bash
# SQ / LDS counters of one layer under conv_wino (wino2=0) and conv_wino2 (wino2=1).  Usage: gpu_w2pmc.sh <tag> B H W Cin Cout pool
R=$(pwd); OUT=gpurun_out/${1:-w2pmc}; shift; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_VMEM"; do
  N=$(echo $SET | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $R/$OUT/$N -o p -- python $R/scripts/conv_probe.py "$@" > /dev/null 2> $R/$OUT/$N.err
  echo "$N exit $?"
done
cd $R
python scripts/w2pmc_summary.py $OUT
