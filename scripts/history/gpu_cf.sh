#!/bin/bash
# first-block experiments of the half-precision trunks: libraries given as arguments ("" = the product build)
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "f16 or conv_first" -p no:cacheprovider --tb=short 2>&1 | tail -3
for V in "" "$@"; do
for S in "32 480 640" "128 480 640"; do set -- $S
HP3D_LIB=$PWD/hand3d_amd/libhp3d$V.so timeout 300 python bench.py --gpus 1 --steps 4 --warmup 1 --layers --cpu-seconds 0 --no-host-path --option streams=1 --dtype f16 --batch $1 --height $2 --width $3 > gpurun_out/cf_$1$V.json 2> gpurun_out/cf_$1$V.txt
python -c "
import json; r=json.load(open('gpurun_out/cf_$1$V.json')); print('lib \'$V\' B=$1', r['value'], 'img/s', r['ms_per_step'])"
grep -E "conv1_1|conv1_2 |conv2_1 " gpurun_out/cf_$1$V.txt
done
done
