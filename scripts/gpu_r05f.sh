#!/bin/bash
# round 5, sixth visit: conv_pw2.hip with its weight rings (per-layer times), and where the B = 1 configurations spend their time now
OUT=gpurun_out/${1:-r05f}; mkdir -p $OUT
run() { tag=$1; shift; timeout 200 python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --layers "$@" > $OUT/$tag.json 2> $OUT/$tag.txt; python -c "import json; d=json.loads(open('$OUT/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])"; }
run b32 --steps 10 --warmup 3
grep -E "conv_pw2" $OUT/b32.txt
run c2 --workload posenet --batch 1 --steps 50 --warmup 10
run c2_pw2force --workload posenet --batch 1 --steps 50 --warmup 10 --option pw2=force
run c2_wino7 --workload posenet --batch 1 --steps 50 --warmup 10 --option wino7=1
run c1 --batch 1 --height 240 --width 320 --steps 50 --warmup 10
cat $OUT/c2.txt
