#!/bin/bash
# round 5, fifth visit: conv_pw2.hip (1x1 head pairs as one launch) -- tests, per-layer table, off / on; the epilogue's select-free leaky-ReLU in
# conv_wino4 / conv_wino7 against the previous build (hand3d_amd/libhp3d_noslope.so, built before the call)
OUT=gpurun_out/${1:-r05e}; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s -k "head_pairs or batch32_winograd_active or 7x7_as_four or tail_pieces or winograd_f4x4_vs or handsegnet_parity or posenet_parity or conv7x7_on_winograd" > $OUT/pytest_quick.log 2>&1; echo "quick pytest exit $?"; grep -E "passed|failed|head pairs|B=32 320x320" $OUT/pytest_quick.log | tail -6
run() { tag=$1; shift; timeout 200 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-host-path --no-other-configs --layers "$@" > $OUT/$tag.json 2> $OUT/$tag.txt; python -c "import json; d=json.loads(open('$OUT/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'], d['roofline']['frac'])"; }
run pw2_on
run pw2_off --option pw2=0
HP3D_LIB=hand3d_amd/libhp3d_noslope.so run noslope
run slope_again
for b in 8 16; do run b${b}_pw2_on --batch $b --option pw2=force; run b${b}_pw2_off --batch $b --option pw2=0; done
grep -E "conv6_1|conv5_1|conv6_6|conv7_6|conv6_2 |conv5_2|conv6_7|conv7_7" $OUT/pw2_on.txt $OUT/pw2_off.txt | cut -c1-150
for t in noslope slope_again; do echo $t; grep -E "HandSegNet/conv1_2|HandSegNet/conv2_1|HandSegNet/conv3_2|PoseNet2D/conv3_2|PoseNet2D/conv6_2 " $OUT/$t.txt; done
