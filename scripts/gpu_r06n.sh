#!/bin/bash
# round 6: timing ablation -- conv_wino4's window loads without the per-load v_add_u32 (row offset VGPR + column offset through the scalar offset: WRONG
# results at tile columns that differ between lanes; measures what the 36 dependent VALU -> VMEM pairs between the MFMA pairs of a step cost)
for r in 1 2; do
  for L in libhp3d libhp3d_w4noadd; do
    HP3D_LIB=hand3d_amd/$L.so timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-host-path --no-other-configs 2>/dev/null | python -c "
import sys,json; c=json.loads(sys.stdin.read()); print('$L', c['value'], c['ms_per_step'], c['roofline']['frac'], c['roofline']['avg_launch_ms'])"
  done
done
