#!/bin/bash
# round 5: first_touch as a policy (auto: cold images of 8 ... 128 MB).  Test, the driver's command (with other_configs), per-layer rows.
OUT=gpurun_out/${1:-r05p2}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "cold_input or first_layer or batch or full" -p no:cacheprovider 2>&1 | tail -4
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver.json 2> $OUT/driver.err; python - <<PY
import json
d = json.load(open('$OUT/driver.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for f in d.get('roofline_by_family', d.get('families', [])) if isinstance(d.get('roofline_by_family', d.get('families', [])), list) else []:
    print(f)
for c in d['other_configs']:
    print(c['config'], c.get('ms_per_step'), c.get('value'))
PY
python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 20 --warmup 5 --layers > $OUT/b32.json 2> $OUT/b32.txt; grep -E "conv1_1" $OUT/b32.txt
python -c "
import json; d=json.load(open('$OUT/b32.json')); print(d['value']); print([ (k, v) for k, v in d.items() if 'first' in str(v)[:400] and k != 'config'][:3])"
for FT in auto 0; do
python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 20 --warmup 5 --height 240 --width 320 --option first_touch=$FT | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('240x320 first_touch=$FT', d['ms_per_step'], d['value'])"
python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 20 --warmup 5 --option first_touch=$FT | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('320x320 first_touch=$FT', d['ms_per_step'], d['value'])"
python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 30 --warmup 5 --batch 8 --option first_touch=$FT | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=8 320x320 first_touch=$FT', d['ms_per_step'], d['value'])"
done
