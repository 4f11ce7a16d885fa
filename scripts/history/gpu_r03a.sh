#!/bin/bash
# Round-3 visit A: GPU tests (incl. the config-5 fixture test) + the default bench line + the B=1 configs (baseline for the small-batch work)
OUT=gpurun_out/${1:-r03a}; mkdir -p $OUT
export TMPDIR=/tmp
rocminfo | grep -m2 -E "Marketing" > $OUT/device.txt 2>&1; nproc >> $OUT/device.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -rx -s --durations=8 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|FAILED|Error|C5 480x640" $OUT/pytest_gpu.log | tail -15
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 --layers > $OUT/bench.json 2> $OUT/bench_layers.txt
echo "bench exit $?"; cut -c1-1500 $OUT/bench.json
timeout 300 python bench.py --gpus 1 --cpu-seconds 0 --no-host-path --workload posenet --batch 1 --steps 50 --warmup 10 --layers > $OUT/c2.json 2> $OUT/c2_layers.txt; cut -c1-300 $OUT/c2.json
timeout 300 python bench.py --gpus 1 --cpu-seconds 0 --no-host-path --batch 1 --height 240 --width 320 --steps 50 --warmup 10 --layers > $OUT/c1.json 2> $OUT/c1_layers.txt; cut -c1-300 $OUT/c1.json
