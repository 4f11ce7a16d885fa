#!/bin/bash
# mask_grow timing (and its ablation builds given as arguments), f32 B=32 480x640, one stream
for V in "" "$@"; do
HP3D_LIB=$PWD/hand3d_amd/libhp3d$V.so python bench.py --gpus 1 --steps 3 --warmup 1 --layers --cpu-seconds 0 --no-host-path --option streams=1 --batch 32 --height 480 --width 640 > gpurun_out/mg$V.json 2> gpurun_out/mg$V.txt
echo "lib '$V': $(grep mask_grow gpurun_out/mg$V.txt)"
done
