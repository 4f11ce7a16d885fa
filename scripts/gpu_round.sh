#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats (+ optional PMC passes).
# Usage: gpu_round.sh [tag] [pmc]      (SKIP_TESTS=1: profiles and bench line only)
# (the profiled commands carry --no-other-configs: the other configurations' launches would enter the per-family averages and the PMC bytes per launch)
TAG=${1:-r01}
PMC=${2:-}
OUT=gpurun_out/$TAG
R=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo | grep -m2 -E "Marketing" > $OUT/device.txt 2>&1
rocminfo | grep -m1 -E "gfx9" >> $OUT/device.txt 2>&1
nproc >> $OUT/device.txt
if [ -z "$SKIP_TESTS" ]; then
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -rx --durations=10 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|FAILED|Error" $OUT/pytest_gpu.log | tail -15
fi
echo "== rocprofv3 kernel-trace --stats"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o hp3d -- python $R/bench.py --gpus 1 --steps 10 --warmup 3 --cpu-seconds 0 --no-host-path --no-other-configs --option streams=1 > $R/$OUT/prof_bench.json 2> $R/$OUT/prof_stderr.txt
echo "rocprof exit $?"
if [ -n "$PMC" ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/pmc_$C -o hp3d -- python $R/bench.py --gpus 1 --steps 2 --warmup 1 --cpu-seconds 0 --no-host-path --no-other-configs --option streams=1 > /dev/null 2> $R/$OUT/pmc_${C}_stderr.txt
    echo "pmc $C exit $?"
  done
fi
cd $R
# the PMC passes feed bench.py's roofline.traffic through profiles/<family>_traffic.json: summarise them here (box-local copy of
# profiles/; the committed files are regenerated from the merged gpurun_out/ by the same script), then take the bench line
python scripts/summarize_prof.py $OUT $TAG > /dev/null 2>&1
echo "== bench"
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 --layers > $OUT/bench.json 2> $OUT/bench_layers.txt
echo "bench exit $?"; cat $OUT/bench.json
find $OUT -name "*.csv" | head -20
