#!/bin/bash
# round 5, last visit: scripts/gpu_r05z.sh (whole GPU suite, rocprofv3 stats, PMC passes, bench, smoke, C5 line, the driver's command) + a short batch sweep
bash scripts/gpu_r05z.sh ${1:-r05}
OUT=gpurun_out/${1:-r05}
for HW in "320 320" "240 320"; do
  set -- $HW
  for N in 8 12 16 24 32 48; do
    python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 20 --warmup 5 --batch $N --height $1 --width $2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1x$2 B=$N', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'])"
  done
done | tee $OUT/sweep.txt
