R=$(pwd); OUT=gpurun_out/pmc4; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for SET in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"; do
  N=$(echo $SET | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $R/$OUT/$N -o p -- python $R/bench.py --gpus 1 --steps 2 --warmup 1 --cpu-seconds 0 --no-host-path --workload posenet > /dev/null 2> $R/$OUT/$N.err
  echo "$N exit $?"
done
