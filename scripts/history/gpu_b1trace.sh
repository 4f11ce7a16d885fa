#!/bin/bash
# kernel timeline of the B=1 configurations (rocprofv3 --kernel-trace: start / end of every dispatch) -> gaps vs kernel time
OUT=gpurun_out/${1:-b1trace}; mkdir -p $OUT; R=$(pwd)
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/c1 -o t -- python $R/bench.py --gpus 1 --cpu-seconds 0 --no-host-path --batch 1 --height 240 --width 320 --steps 20 --warmup 5 > $R/$OUT/c1.json 2> $R/$OUT/c1.err
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/c2 -o t -- python $R/bench.py --gpus 1 --cpu-seconds 0 --no-host-path --workload posenet --batch 1 --steps 20 --warmup 5 > $R/$OUT/c2.json 2> $R/$OUT/c2.err
cd $R
python - <<PY
import csv,glob,collections
for tag in ('c1','c2'):
    f=glob.glob('$OUT/%s/**/*kernel_trace.csv'%tag, recursive=True)
    if not f: print(tag,'no trace'); continue
    rows=list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    # last timed step: take the final N dispatches spanning one step: find step boundaries by the first kernel name
    names=[r['Kernel_Name'] for r in rows]
    first=names.index(next(n for n in names if 'conv_first' in n))
    idx=[i for i,n in enumerate(names) if 'conv_first' in n]
    # steps start at every 2nd conv_first for the full path (HandSegNet, PoseNet) and at every one for posenet
    per = 2 if tag=='c1' else 1
    starts=idx[::per]
    a,b=starts[-12],starts[-11]      # one timed (unprofiled) step in the middle of the run
    seg=rows[a:b]
    t0=int(seg[0]['Start_Timestamp']); t1=int(seg[-1]['End_Timestamp'])
    busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seg)
    gaps=[int(seg[i+1]['Start_Timestamp'])-int(seg[i]['End_Timestamp']) for i in range(len(seg)-1)]
    print(tag,'kernels',len(seg),'span us',(t1-t0)/1e3,'sum kernel us',busy/1e3,'sum gaps us',sum(gaps)/1e3,'median gap us',sorted(gaps)[len(gaps)//2]/1e3)
    agg=collections.OrderedDict()
    for r in seg:
        n=r['Kernel_Name'].split('(')[0][-60:]
        d=agg.setdefault(n,[0,0]); d[0]+=1; d[1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    for n,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:14]:
        print('   %-62s x%-3d %8.1f us  avg %6.1f'%(n,c,t/1e3,t/1e3/c))
    with open('$OUT/%s_timeline.txt'%tag,'w') as fo:
        for i,r in enumerate(seg):
            fo.write('%8.1f %8.1f %s\n'%((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,r['Kernel_Name'][:90]))
PY
