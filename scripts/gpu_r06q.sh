#!/bin/bash
# round 6: same-box A/B of library variants on the primary line (alternating, three rounds): bash scripts/gpu_r06q.sh libhp3d libhp3d_<variant> ...
for r in 1 2 3; do
  for L in "$@"; do
    HP3D_LIB=hand3d_amd/$L.so timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-host-path --no-other-configs 2>/dev/null | python -c "
import sys,json; c=json.loads(sys.stdin.read()); print('$L', c['value'], c['value_min'], c['value_max'], c['ms_per_step'], c['roofline']['frac'])"
  done
done
