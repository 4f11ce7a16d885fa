R=$(pwd); OUT=gpurun_out/pmc1; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|GRBM_[A-Z_0-9]+|TCC_[A-Z_0-9]+|TCP_[A-Z_0-9]+)\b" | sort -u > $R/$OUT/counters.txt
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $SET | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $R/$OUT/$N -o p -- python $R/bench.py --gpus 1 --steps 1 --warmup 1 --cpu-images 0 --workload posenet > /dev/null 2> $R/$OUT/$N.err
  echo "$N exit $?"
done
