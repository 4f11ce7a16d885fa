#!/bin/bash
# gpu_quick.sh + one FETCH_SIZE / WRITE_SIZE counter pass (HBM-side traffic of the conv families).  Usage: tag [pytest -k]
TAG=${1:-qp}; K=${2:-wino}
bash scripts/gpu_quick.sh $TAG "$K"
R=$(pwd); OUT=gpurun_out/$TAG; export TMPDIR=/tmp; cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/pmc_$C -o hp3d -- python $R/bench.py --gpus 1 --steps 2 --warmup 1 --cpu-seconds 0 --no-host-path > /dev/null 2> $R/$OUT/pmc_${C}_stderr.txt
  echo "pmc $C exit $?"
done
cd $R
python - <<PY
import csv, collections
for C, mul in (('FETCH_SIZE', 2.0), ('WRITE_SIZE', 1.0)):     # FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md); units: KB? -> see summarize_prof.py
    rows = list(csv.DictReader(open('$OUT/pmc_%s/hp3d_counter_collection.csv' % C)))
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in rows:
        if r['Counter_Name'] != C: continue
        k = 'conv_wino' if 'conv_wino' in r['Kernel_Name'] else 'conv_mfma' if 'conv_mfma' in r['Kernel_Name'] else None
        if k: agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
    for k, (v, n) in agg.items():
        print(C, k, 'raw per launch %.1f' % (v / n), 'launches', n)
PY
