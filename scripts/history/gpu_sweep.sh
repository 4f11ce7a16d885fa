#!/bin/bash
OUT=gpurun_out/sweep; mkdir -p $OUT
run() { echo "== $1"; env $1 timeout 300 python bench.py --gpus 1 --steps 5 --warmup 2 --cpu-seconds 0 --no-host-path --layers --workload posenet > $OUT/b.json 2> $OUT/"$(echo $1 | tr ' =' '__')".txt; python -c "
import json;d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['achieved'])"; }










run "HP3D_CONV_DBG=0"
