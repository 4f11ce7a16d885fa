#!/bin/bash
# round 5: seg_upsample_softmax with one atomicMax per workgroup; 64 (shipped) / 32 / 16 workgroups per image; mask tests
OUT=gpurun_out/${1:-r05u}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_fixtures.py -q -m gpu -k "mask or full or batch or fixture or u8 or handseg or argmax" -p no:cacheprovider 2>&1 | tail -3
for V in "" gx32 gx16; do
  L=hand3d_amd/libhp3d.so; [ -n "$V" ] && L=hand3d_amd/libhp3d_seg_$V.so
  HP3D_LIB=$(pwd)/$L python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 20 --warmup 5 --layers > $OUT/b_$V.json 2> $OUT/b_$V.txt
  echo "== ${V:-shipped}: $(python -c "import json; d=json.load(open('$OUT/b_$V.json')); print(d['ms_per_step'], d['value'])") $(grep -E '^seg_upsample' $OUT/b_$V.txt)"
done
HP3D_LIB=$(pwd)/hand3d_amd/libhp3d.so python bench.py --dtype f16 --batch 128 --height 480 --width 640 --steps 3 --warmup 2 --cpu-seconds 0 --no-host-path --no-other-configs --layers 2>&1 >/dev/null | grep -E "^seg_upsample|^mask_grow"
