#!/bin/bash
# bench the library variants hand3d_amd/libhp3d_<TAG>.so (experiment builds) next to the product library
OUT=gpurun_out/${1:-var}; mkdir -p $OUT
for L in hand3d_amd/libhp3d.so hand3d_amd/libhp3d_*.so; do
  T=$(basename $L .so)
  HP3D_LIB=$PWD/$L timeout 300 python bench.py --gpus 1 --steps 5 --warmup 2 --layers --cpu-seconds 0 --no-host-path > $OUT/$T.json 2> $OUT/$T.layers.txt
  python - <<PY
import json
d=json.loads(open('$OUT/$T.json').read().strip().splitlines()[-1])
print('$T', d['value'], d['ms_per_step'])
PY
done
