#!/bin/bash
# round 6: the split-operand option (wino4_split=auto) with conv_wino4s's window addresses formed in one block (variant) against the tree, same box, alternating
for r in 1 2; do
  for L in "$@"; do
    HP3D_LIB=hand3d_amd/$L.so timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-host-path --no-other-configs --option wino4_split=auto 2>/dev/null | python -c "
import sys,json; c=json.loads(sys.stdin.read()); print('$L split', c['value'], c['ms_per_step'])"
  done
done
