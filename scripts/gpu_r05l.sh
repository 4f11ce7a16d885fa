#!/bin/bash
# round 5: conv_first.hip's balanced tile runs against the row walk (option first_walk), same box: tests, then per-layer tables.
OUT=gpurun_out/${1:-r05l}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "first_layer or conv_first or handsegnet or batch" -p no:cacheprovider 2>&1 | tail -4
for R in 1 2; do
for FW in balanced rows; do
  python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 20 --warmup 5 --layers --option first_walk=$FW > $OUT/b32_$FW.json 2> $OUT/b32_$FW.txt
  echo "== B=32 320x320 first_walk=$FW: $(python -c "import json; d=json.load(open('$OUT/b32_$FW.json')); print(d['ms_per_step'], d['value'])")"; grep -E "conv1_1|conv1_2" $OUT/b32_$FW.txt
done
done
for FW in balanced rows; do
  python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 20 --warmup 5 --layers --height 240 --width 320 --option first_walk=$FW > $OUT/c4_$FW.json 2> $OUT/c4_$FW.txt
  echo "== B=32 240x320 first_walk=$FW: $(python -c "import json; d=json.load(open('$OUT/c4_$FW.json')); print(d['ms_per_step'], d['value'])")"; grep -E "conv1_1" $OUT/c4_$FW.txt
  python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 100 --warmup 10 --layers --batch 1 --height 240 --width 320 --option first_walk=$FW > $OUT/c1_$FW.json 2> $OUT/c1_$FW.txt
  echo "== B=1 240x320 first_walk=$FW: $(python -c "import json; d=json.load(open('$OUT/c1_$FW.json')); print(d['ms_per_step'], d['value'])")"; grep -E "conv1_1" $OUT/c1_$FW.txt
  python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 5 --warmup 2 --layers --batch 128 --height 480 --width 640 --dtype f16 --option f16_fuse12=0 --option first_walk=$FW > $OUT/c5_$FW.json 2> $OUT/c5_$FW.txt
  echo "== C5 shard f16 (unfused first block) first_walk=$FW: $(python -c "import json; d=json.load(open('$OUT/c5_$FW.json')); print(d['ms_per_step'], d['value'])")"; grep -E "conv1_1" $OUT/c5_$FW.txt
done
