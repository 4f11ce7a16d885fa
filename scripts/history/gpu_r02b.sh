#!/bin/bash
# Round-2 GPU visit B: re-check of the Winograd kernels after the uniformity fix + headline bench
TAG=${1:-r02b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_fixtures.py -m gpu -q --tb=short -p no:cacheprovider -k "wino or conv7x7 or batch32 or fixtures or conv_ or odd_tile or c1 or keypoints" > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed|FAILED|Error" $OUT/pytest_gpu.log | tail -10
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --layers --cpu-seconds 25 > $OUT/bench.json 2> $OUT/bench_layers.txt
echo "bench exit $?"; HP3D_FIRST_STORE=1 timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --layers --cpu-seconds 0 --no-host-path > $OUT/bench_first1.json 2> $OUT/bench_first1_layers.txt; grep conv1_1 $OUT/bench_layers.txt $OUT/bench_first1_layers.txt; scripts/micro/write_bw > $OUT/write_bw.txt 2>&1; cat $OUT/write_bw.txt; python - <<PY
import json
r = json.load(open("$OUT/bench.json"))
print(r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["achieved_algorithmic"])
print("cpu", r["cpu_baseline"]); print("epe", r["epe_vs_oracle"]); print("host", r["host_path"])
for o in r["roofline_other_conv"]: print(o["kernel"], o["frac"], o["achieved"], o["unit"], o["avg_launch_ms"])
PY
