#!/bin/bash
# round 5, seventh visit: packed output transforms in the conv_wino4 / conv_wino7 epilogues against the previous build (hand3d_amd/libhp3d_prev.so)
OUT=gpurun_out/${1:-r05g}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_fixtures.py -m gpu -q --tb=short -p no:cacheprovider -k "winograd or f4x4 or 7x7 or tail or fixtures or batch32 or head_pairs or config_c1 or arbitrary or handsegnet or posenet" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
run() { tag=$1; shift; timeout 200 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-host-path --no-other-configs --layers "$@" > $OUT/$tag.json 2> $OUT/$tag.txt; python -c "import json; d=json.loads(open('$OUT/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'], d['roofline']['frac'])"; }
HP3D_LIB=hand3d_amd/libhp3d_prev.so run prev
run packed
HP3D_LIB=hand3d_amd/libhp3d_prev.so run prev2
run packed2
run streams2 --option streams=2
for t in prev packed; do echo $t; grep -E "HandSegNet/conv1_2|HandSegNet/conv2_1|HandSegNet/conv2_2|HandSegNet/conv3_2|HandSegNet/conv4_2|PoseNet2D/conv1_2|PoseNet2D/conv3_2|PoseNet2D/conv6_2 " $OUT/$t.txt; done
