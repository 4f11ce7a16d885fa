#!/bin/bash
# gpu_round.sh without the test suite: bench line, rocprofv3 kernel stats, PMC traffic passes.  Usage: tag
TAG=${1:-r01}; OUT=gpurun_out/$TAG; R=$(pwd); mkdir -p $OUT; export TMPDIR=/tmp
rocminfo | grep -m2 -E "Marketing" > $OUT/device.txt 2>&1; rocminfo | grep -m1 -E "gfx9" >> $OUT/device.txt 2>&1; nproc >> $OUT/device.txt
echo "no pytest in this pass (scripts/gpu_profile_only.sh)" > $OUT/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --layers > $OUT/bench.json 2> $OUT/bench_layers.txt
echo "bench exit $?"; cut -c1-200 $OUT/bench.json
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o hp3d -- python $R/bench.py --gpus 1 --steps 10 --warmup 3 --cpu-seconds 0 --no-host-path > $R/$OUT/prof_bench.json 2> $R/$OUT/prof_stderr.txt
echo "rocprof exit $?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/pmc_$C -o hp3d -- python $R/bench.py --gpus 1 --steps 2 --warmup 1 --cpu-seconds 0 --no-host-path > /dev/null 2> $R/$OUT/pmc_${C}_stderr.txt
  echo "pmc $C exit $?"
done
