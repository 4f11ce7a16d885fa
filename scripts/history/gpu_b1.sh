#!/bin/bash
# small-batch latency: PoseNet2D B=1 (config 2) and the full path B=1 / B=4, per-layer tables (and hipGraph replay with "graph")
OUT=gpurun_out/${1:-b1g}; mkdir -p $OUT
for G in "" ${2:+--graph}; do
  for WL in "posenet 1 256 256" "full 1 240 320" "full 4 240 320"; do
    set -- $WL
    timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --layers --cpu-seconds 0 --no-host-path --workload $1 --batch $2 --height $3 --width $4 $G > $OUT/b_$1_$2$G.json 2> $OUT/b_$1_$2$G.txt
    python - <<PY
import json
r=json.load(open("$OUT/b_$1_$2$G.json")); print("$1 B=$2 graph='$G':", r["value"], "img/s", r["ms_per_step"], "ms", r["config"].get("hipgraph_replays"))
PY
  done
done
