#!/bin/bash
# round 5: conv_first.hip kernel time by launch size and image shape (rocprofv3 kernel trace of scripts/first_probe.py)
OUT=$(pwd)/gpurun_out/${1:-r05n}
R=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for S in "32 320 320" "20 320 320" "32 256 256" "50 256 256" "32 256 320" "32 320 256" "40 256 256" "25 320 320" "32 304 304" "32 336 336"; do
  T=$(echo $S | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_$T -o p -- python $R/scripts/first_probe.py $S > $OUT/$T.txt 2>&1
  F=$(find $OUT/p_$T -name "*kernel_stats.csv" | head -1)
  python - "$F" "$S" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
B, H, W = [int(v) for v in sys.argv[2].split()]
for r in rows:
    if 'conv_first' in r['Name']:
        us = float(r['AverageNs']) / 1e3
        mb = B * H * W * (64 * 4 + 12) / 1e6
        print('%3d x %3d x %3d: %7.1f us  min %7.1f  %6.0f MB  %5.2f TB/s (min: %5.2f)' % (B, H, W, us, float(r['MinNs']) / 1e3, mb, mb / us / 1e6 * 1e6 / 1e6 * 1e0, mb / (float(r['MinNs']) / 1e3)))
PY
done
