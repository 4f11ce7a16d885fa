# SQ / GRBM / TCC counters of one PoseNet pass (separate --pmc passes, kernel-trace only).  Usage: pmc_sq.sh [tag] [nsets]
R=$(pwd); OUT=gpurun_out/${1:-pmc1}; NSETS=${2:-5}; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
I=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_WAVE32_LDS SQ_IFETCH SQ_INST_LEVEL_VMEM"; do
  I=$((I+1)); [ $I -gt $NSETS ] && break
  N=$(echo $SET | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $R/$OUT/$N -o p -- python $R/bench.py --gpus 1 --steps 2 --warmup 1 --cpu-seconds 0 --no-host-path --workload posenet > /dev/null 2> $R/$OUT/$N.err
  echo "$N exit $?"
done
