#!/bin/bash
# Round-2 GPU visit A: full parity suite, the bench record (f32 headline), the other BASELINE shapes, the small-batch
# Winograd split-K experiment, rocprofv3 kernel stats.   Usage: bash scripts/gpu_r02a.sh [tag]
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
R=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo | grep -m2 -E "Marketing" > $OUT/device.txt 2>&1; nproc >> $OUT/device.txt
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s --durations=15 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|FAILED|Error|xfail" $OUT/pytest_gpu.log | tail -20
echo "== bench (headline)"
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 --layers > $OUT/bench.json 2> $OUT/bench_layers.txt
echo "bench exit $?"; cat $OUT/bench.json
b() { tag=$1; shift; timeout 600 python bench.py --gpus 1 --cpu-seconds 0 --no-host-path --layers "$@" > $OUT/$tag.json 2> $OUT/$tag.txt; echo "$tag exit $?"; python - <<PY
import json
try:
    r = json.load(open("$OUT/$tag.json")); print("$tag", r["value"], r["unit"], r["ms_per_step"], "ms/step", r["roofline"]["kernel"], r["roofline"]["frac"], r["roofline"].get("achieved_algorithmic"))
except Exception as e: print("$tag: no result", e)
PY
}
b f32_480x640 --steps 4 --warmup 1 --height 480 --width 640
b f32_240x320 --steps 8 --warmup 2 --height 240 --width 320
b f16_320 --steps 8 --warmup 2 --dtype f16
b f16_480x640_b128 --steps 3 --warmup 1 --dtype f16 --height 480 --width 640 --batch 128
b pose_b1_split --steps 50 --warmup 10 --workload posenet --batch 1
b pose_b1_nosplit --steps 50 --warmup 10 --workload posenet --batch 1 --option wino_splitk=0
b full_b1_split --steps 30 --warmup 5 --batch 1 --height 240 --width 320
b full_b1_nosplit --steps 30 --warmup 5 --batch 1 --height 240 --width 320 --option wino_splitk=0
b full_b8 --steps 10 --warmup 3 --batch 8 --height 240 --width 320
echo "== torchrun world 1 (native RCCL path through the launcher)"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --steps 3 --warmup 1 --cpu-seconds 0 --no-host-path > $OUT/torchrun1.json 2> $OUT/torchrun1.txt
echo "torchrun exit $?"; cat $OUT/torchrun1.json | head -c 400; echo
echo "== rocprofv3 kernel-trace --stats"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o hp3d -- python $R/bench.py --gpus 1 --steps 10 --warmup 3 --cpu-seconds 0 --no-host-path > $R/$OUT/prof_bench.json 2> $R/$OUT/prof_stderr.txt
echo "rocprof exit $?"
cd $R; find $OUT -name "*stats*.csv" | head
