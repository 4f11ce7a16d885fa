#!/bin/bash
# round 5: is HandSegNet's conv1_1 (3.5 TB/s in the pipeline, 4.8 standalone) waiting for its COLD input image?  Option first_touch streams the
# image through the caches right before the launch.  Per-layer rows and the bench line, on / off alternating.
OUT=gpurun_out/${1:-r05o}
mkdir -p $OUT
for R in 1 2; do
for FT in 1 0; do
  python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 20 --warmup 5 --layers --option first_touch=$FT > $OUT/b32_$FT.json 2> $OUT/b32_$FT.txt
  echo "== B=32 320x320 first_touch=$FT: $(python -c "import json; d=json.load(open('$OUT/b32_$FT.json')); print(d['ms_per_step'], d['value'])")"; grep -E "conv1_1" $OUT/b32_$FT.txt
done
done
