#!/bin/bash
# round 4, call A: where does conv_wino4's time go on the FULL B = 32 path (both trunks)?  One column per build variant
# (timing ablations compute wrong results on purpose), single stream, per-layer tables.
OUT=gpurun_out/${1:-r04a}; mkdir -p $OUT
shift
rocminfo | grep -m2 -E "Marketing" > $OUT/device.txt 2>&1; nproc >> $OUT/device.txt
timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --cpu-seconds 0 --no-host-path > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<PY
import json
d=json.load(open('$OUT/bench_default.json')); print('default two-stream', d['value'], d['ms_per_step'], d['roofline']['frac'])
PY
for v in base "$@"; do
  lib=hand3d_amd/libhp3d_$v.so; [ $v == base ] && lib=hand3d_amd/libhp3d.so
  HP3D_LIB=$lib timeout 300 python bench.py --gpus 1 --steps 6 --warmup 2 --layers --cpu-seconds 0 --no-host-path --option streams=1 > $OUT/bench_$v.json 2> $OUT/layers_$v.txt
  python - <<PY
import json
try:
    d=json.load(open('$OUT/bench_$v.json')); print('$v', d['value'], d['ms_per_step'])
except Exception as e: print('$v FAILED', e)
PY
done
python - "$OUT" base "$@" <<'PY'
import sys
out=sys.argv[1]; vs=sys.argv[2:]
tabs=[]
for v in vs:
    t={}
    try:
        for l in open('%s/layers_%s.txt'%(out,v)):
            f=l.split()
            if len(f)>=3 and (f[1].startswith('conv_') ): t[f[0]+' '+f[1][5:28]]=f[2]
    except Exception: pass
    tabs.append(t)
print('%-52s'%'layer', ' '.join('%8s'%v[:8] for v in vs))
for k in tabs[0]:
    print('%-52s'%k, ' '.join('%8s'%t.get(k,'-') for t in tabs))
PY
