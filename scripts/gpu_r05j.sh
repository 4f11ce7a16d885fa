#!/bin/bash
# round 5: conv_wino7.hip's channel split for under-filled launches (small batches).  New GPU test, then C2 / C1 and small batches under
# wino7 = auto (split) vs wino7 = 0 (the nine-block forms on conv_wino2 / conv_wino4), forced split counts at B = 1, the B = 32 line.
OUT=gpurun_out/${1:-r05j}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "channel_split or four_4x4 or batch" -p no:cacheprovider -s 2>&1 | grep -E "channel split|passed|failed|Error" | tail -20
B="python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 200 --warmup 20"
for W7 in auto 0; do
  echo "== posenet B=1 wino7=$W7"; $B --workload posenet --batch 1 --height 256 --width 256 --option wino7=$W7 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  echo "== full B=1 240x320 wino7=$W7"; $B --batch 1 --height 240 --width 320 --option wino7=$W7 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
for KS in 2 4 6 8 10; do
  echo "== posenet B=1 wino7_ksplit=$KS"; $B --workload posenet --batch 1 --height 256 --width 256 --option wino7_ksplit=$KS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
B="python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 50 --warmup 10"
for N in 2 4 8 12 16; do
  for W7 in auto 0; do
    echo "== full B=$N 320x320 wino7=$W7"; $B --batch $N --option wino7=$W7 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  done
done
echo "== posenet B=1 layers"; python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 100 --warmup 10 --workload posenet --batch 1 --height 256 --width 256 --layers 2> $OUT/c2_layers.txt > $OUT/c2.json; grep -E "conv[4-7]_[1-7]|total" $OUT/c2_layers.txt | head -40
echo "== B=32"; python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 20 --warmup 5 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'])"
