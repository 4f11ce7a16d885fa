#!/bin/bash
# L2 / L1 counters of the half-precision trunk kernel (one --pmc pass per counter set), B=32 480x640 f16, streams=1.
OUT=gpurun_out/${1:-pmch16}; R=$(pwd); mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for C in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  N=$(echo $C | tr ' ' '_')
  timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/$N -o hp3d -- python $R/bench.py --gpus 1 --steps 1 --warmup 1 --cpu-seconds 0 --no-host-path --option streams=1 --dtype f16 --batch 32 --height 480 --width 640 > /dev/null 2> $R/$OUT/${N}_stderr.txt
  echo "pmc $C exit $?"
done
cd $R
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'conv_h16' not in k: continue
        key = (k[k.find('conv_h16'):][:40], r['Counter_Name'])
        agg[key][0] += float(r['Counter_Value']); agg[key][1] += 1
    for (k, c), (v, n) in sorted(agg.items()):
        print("%-42s %-32s per-launch %.4g (n=%d)" % (k, c, v / n, n))
PY
