#!/bin/bash
# L2 (TCC) counters of one layer under conv_wino / conv_wino4.  Usage: gpu_w4tcc.sh <tag> B H W Cin Cout pool k modes
R=$(pwd); OUT=gpurun_out/${1:-w4tcc}; shift; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for SET in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_WRITE_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_NC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  N=$(echo $SET | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $R/$OUT/$N -o p -- python $R/scripts/conv_probe.py "$@" > /dev/null 2> $R/$OUT/$N.err
  echo "$N exit $?"
done
cd $R
python - <<PY
import csv,glob,collections,re
agg=collections.OrderedDict()
for f in sorted(glob.glob('$OUT/*/p_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        m=re.search(r'(conv_wino[247]?_kernel(?:<[^>]*>)?)', r['Kernel_Name'])
        if not m: continue
        d=agg.setdefault(m.group(1),collections.defaultdict(list))
        d[r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    m={c:sum(v)/len(v) for c,v in d.items()}
    print(k); print('  ', {c:round(v) for c,v in m.items()})
    if m.get('TCC_REQ_sum'): print('   L2 hit rate %.3f'%(m.get('TCC_HIT_sum',0)/max(m.get('TCC_HIT_sum',0)+m.get('TCC_MISS_sum',0),1)))
PY
