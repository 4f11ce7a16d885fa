#!/bin/bash
# round 6 item 6: same-box A/B of the lifting-stage changes on the bench line (alternating, 3 rounds)
for r in 1 2 3; do
  for OPT in "--option fc_tail=0 --option tiny_gemm=0 --option kp_up_side=0" "--option kp_up_side=0" ""; do
    timeout 300 python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 20 --warmup 5 $OPT 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('[$OPT]', d['value'], d['value_min'], d['value_max'], d['ms_per_step'])"
  done
done
