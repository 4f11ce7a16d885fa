#!/bin/bash
# round 6 item 2: where B = 24 / 12 lose per image against B = 32 (per-layer event timings)
OUT=gpurun_out/${1:-r06k}; mkdir -p $OUT
for N in 12 24 32; do
  timeout 300 python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 10 --warmup 3 --batch $N --layers > $OUT/bench_b$N.json 2> $OUT/layers_b$N.txt
done
python - <<'PY'
import re
def load(n):
    d = {}
    for l in open('gpurun_out/r06k/layers_b%d.txt' % n):
        p = l.split()
        if len(p) >= 5 and p[0] != 'layer':
            try: d[(p[0], p[1])] = float(p[2])
            except ValueError: pass
    return d
a, b, c = load(12), load(24), load(32)
names = []
for l in open('gpurun_out/r06k/layers_b32.txt'):
    p = l.split()
    if len(p) >= 5 and p[0] != 'layer': names.append(p[0])
def by_layer(d):
    o = {}
    for (n, k), v in d.items(): o.setdefault(n, [0.0, []]); o[n][0] += v; o[n][1].append(k)
    return o
A, B_, C = by_layer(a), by_layer(b), by_layer(c)
print('%-28s %8s %8s %8s   per-image time relative to B=32 (12, 24)   kernels at 24' % ('layer', 'B=12', 'B=24', 'B=32'))
seen = set()
for n in names:
    if n in seen: continue
    seen.add(n)
    x, y, z = A.get(n, [0, []])[0], B_.get(n, [0, []])[0], C[n][0]
    print('%-28s %8.3f %8.3f %8.3f   %5.2f %5.2f   %s' % (n, x, y, z, (x / 12) / (z / 32) if z else 0, (y / 24) / (z / 32) if z else 0, ','.join(B_.get(n, [0, ['-']])[1])))
print('total', sum(v[0] for v in A.values()), sum(v[0] for v in B_.values()), sum(v[0] for v in C.values()))
PY
