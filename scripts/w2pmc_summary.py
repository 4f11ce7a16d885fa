"""Summarise scripts/gpu_w2pmc.sh output: python scripts/w2pmc_summary.py gpurun_out/<tag>"""
import collections, csv, glob, re, sys
agg = collections.OrderedDict()
for f in sorted(glob.glob(sys.argv[1] + '/*/p_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(conv_wino[247]?_kernel(?:<[^>]*>)?)', r['Kernel_Name'])
        if not m:
            continue
        d = agg.setdefault(m.group(1), collections.defaultdict(list))
        d[r['Counter_Name']].append(float(r['Counter_Value']))
        d['dur'].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    dur = m['dur']; clock = m.get('SQ_BUSY_CYCLES', 0) / 32 / dur
    wc = max(m.get('SQ_WAVE_CYCLES', 1), 1)
    print(k)
    print('  dur us %.1f clock %.2f GHz | MFMA busy %.3f | wave cycles: parked %.3f issue-wait %.3f active %.3f (VALU %.3f VMEM %.3f LDS %.3f)' % (
        dur / 1e3, clock, m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / (dur * clock), m.get('SQ_WAIT_ANY', 0) / wc, m.get('SQ_WAIT_INST_ANY', 0) / wc,
        m.get('SQ_ACTIVE_INST_ANY', 0) / wc, m.get('SQ_ACTIVE_INST_VALU', 0) / wc, m.get('SQ_ACTIVE_INST_VMEM', 0) / wc, m.get('SQ_ACTIVE_INST_LDS', 0) / wc))
    print('  insts: MFMA %d VALU %d SALU %d LDS %d VMEM_RD %d | LDS conflict cycles %d / active %d, LDS issue stall %d, VMEM issue stall %d' % (
        m.get('SQ_INSTS_MFMA', 0), m.get('SQ_INSTS_VALU', 0), m.get('SQ_INSTS_SALU', 0), m.get('SQ_INSTS_LDS', 0), m.get('SQ_INSTS_VMEM_RD', 0),
        m.get('SQ_LDS_BANK_CONFLICT', 0), m.get('SQ_LDS_IDX_ACTIVE', 0), m.get('SQ_WAIT_INST_LDS', 0), m.get('SQ_WAIT_INST_VMEM', 0)))
