#!/bin/bash
# conv_wino4 variants (weight-ring depth): per-layer tables of the B = 32 bench with every eligible layer on the kernel
OUT=gpurun_out/${1:-w4b}; mkdir -p $OUT
shift
for v in base "$@"; do
  lib=hand3d_amd/libhp3d_$v.so; [ $v == base ] && lib=hand3d_amd/libhp3d.so
  HP3D_LIB=$lib timeout 300 python bench.py --gpus 1 --steps 6 --warmup 2 --layers --cpu-seconds 0 --no-host-path --option wino4=1 > $OUT/bench_$v.json 2> $OUT/layers_$v.txt
  python - <<PY
import json
d=json.load(open('$OUT/bench_$v.json')); print('$v', d['value'], d['ms_per_step'])
PY
done
python - "$OUT" base "$@" <<'PY'
import sys
out=sys.argv[1]; vs=sys.argv[2:]
tabs=[]
for v in vs:
    t={}
    for l in open('%s/layers_%s.txt'%(out,v)):
        f=l.split()
        if len(f)>=3 and f[1].startswith('conv_wino4'): t[f[0]+' '+f[1][11:]]=f[2]
    tabs.append(t)
for k in tabs[0]:
    print('%-44s'%k, ' '.join('%7s'%t.get(k,'-') for t in tabs))
PY
