#!/bin/bash
# round 5: per-layer tables at B = 12 / 16 / 24 / 32 (320x320): which layers make B = 12 and B = 24 cost more per image than their neighbours?
OUT=gpurun_out/${1:-r05y}
mkdir -p $OUT
for N in 8 12 16 24 32; do
  python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 20 --warmup 5 --batch $N --layers > $OUT/b$N.json 2> $OUT/b$N.txt
  echo "B=$N $(python -c "import json; d=json.load(open('$OUT/b$N.json')); print(d['ms_per_step'], d['value'])")"
done
