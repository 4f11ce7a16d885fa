"""profiles/<tag>_h16_counters.md from a scripts/gpu_pmc_h16.sh output directory (rocprofv3 --pmc passes of the half-precision
trunk kernel): per instantiation, duration-weighted over launches longer than 100 us,
  clock     = SQ_BUSY_CYCLES / 32 shader engines / duration
  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x clock)
  L2 hit    = TCC_HIT / (TCC_HIT + TCC_MISS)
The passes run the same command, so dispatch ids line up across them.

  python scripts/h16_counters.py gpurun_out/pmch16 r02
"""
import collections
import csv
import os
import sys


def load(path):
    rows = collections.defaultdict(dict)
    if not os.path.exists(path):
        return rows
    for r in csv.DictReader(open(path)):
        if 'conv_h16' not in r['Kernel_Name']:
            continue
        k = r['Kernel_Name']
        if 'conv_h16_first_kernel' in k:
            k = 'conv_h16_first_kernel: fused first block, filters resident in registers, 1 workgroup per CU'      # (round 6; no template arguments)
        else:
            k = k[k.find('conv_h16_kernel'):]
            k = k[:k.find('(')] if '(' in k else k
        d = rows[int(r['Dispatch_Id'])]
        d['k'] = k
        d[r['Counter_Name']] = float(r['Counter_Value'])
        d['dur_ns'] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    return rows


def main():
    src, tag = sys.argv[1], sys.argv[2]
    a = load(os.path.join(src, 'SQ_INSTS_VALU_MFMA_MOPS_F16_SQ_BUSY_CYCLES', 'hp3d_counter_collection.csv'))
    b = load(os.path.join(src, 'SQ_VALU_MFMA_BUSY_CYCLES_SQ_WAVE_CYCLES', 'hp3d_counter_collection.csv'))
    c = load(os.path.join(src, 'TCC_HIT_sum_TCC_MISS_sum', 'hp3d_counter_collection.csv'))
    agg = collections.defaultdict(list)
    for did, r in a.items():
        if did not in b or b[did]['k'] != r['k'] or r['dur_ns'] < 100e3:
            continue
        clock = r['SQ_BUSY_CYCLES'] / 32 / r['dur_ns']
        busy = b[did]['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (b[did]['dur_ns'] * clock)
        hit = None
        if did in c and c[did].get('TCC_HIT_sum') is not None:
            hit = c[did]['TCC_HIT_sum'] / max(1.0, c[did]['TCC_HIT_sum'] + c[did].get('TCC_MISS_sum', 0.0))
        agg[r['k']].append((r['dur_ns'] / 1e3, clock, busy, hit))
    lines = ['# conv_h16 (half-precision 3x3 trunk kernel) SQ / L2 counters, %s' % tag, '',
             '`bash scripts/gpu_pmc_h16.sh` (rocprofv3 --kernel-trace --pmc, one pass per counter pair; bench.py --dtype f16 --batch 32',
             '--height 480 --width 640 --option streams=1), summarised by `scripts/h16_counters.py`; launches longer than 100 us,',
             'duration-weighted.  clock = SQ_BUSY_CYCLES / 32 SEs / duration; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles).', '',
             '| instantiation <cout blocks per wave, pool, workgroups per CU, fused first block, filter size> | launches | launch us | clock GHz | MFMA busy | busy x clock / 2.4 GHz | L2 hit rate |',
             '|---|---|---|---|---|---|---|']
    for k, v in sorted(agg.items()):
        w = sum(x[0] for x in v)
        clock = sum(x[0] * x[1] for x in v) / w
        busy = sum(x[0] * x[2] for x in v) / w
        hv = [x for x in v if x[3] is not None]
        hit = '%.2f' % (sum(x[0] * x[3] for x in hv) / sum(x[0] for x in hv)) if hv else '-'
        lines.append('| %s | %d | %.0f..%.0f | %.2f | %.3f | %.3f | %s |' % (k.replace('conv_h16_kernel', ''), len(v), min(x[0] for x in v), max(x[0] for x in v),
                                                                      clock, busy, busy * clock / 2.4, hit))
    lines += ['', 'Reading: the matrix pipe is busy 0.5-0.73 of the time, and the chip runs these kernels at 1.5-1.7 GHz (it clocks to its',
              'power budget; profiled passes run a little lower than unprofiled ones), so busy x clock / 2.4 GHz -- the fraction of the',
              '2.5 PF dense f16 peak the MFMAs executed -- comes to 0.36-0.48, which is what bench.py reports from its own event timing.',
              'Raising `MFMA busy` returns only partly as throughput (denser bodies clock lower: MI355X_MICROARCH.md, DVFS give-back).']
    # kernel families of the same run (kernel trace of the first SQ pass: one warm-up + one timed step, B=32 480x640, f16 trunks)
    kt = os.path.join(src, 'SQ_INSTS_VALU_MFMA_MOPS_F16_SQ_BUSY_CYCLES', 'hp3d_kernel_trace.csv')
    if os.path.exists(kt):
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from summarize_prof import family
        fam = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(kt)):
            f = fam[family(r['Kernel_Name'])]
            f[0] += 1
            f[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
        tot = sum(v[1] for v in fam.values())
        lines += ['', '## Kernel families of the same command (rocprofv3 --kernel-trace of the counter pass; half-precision trunks, B=32, 480x640)', '',
                  '| kernel family | launches | total ms | avg launch us | % GPU time |', '|---|---|---|---|---|']
        for k, (n, ms) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
            if ms / tot > 0.002:
                lines.append('| %s | %d | %.2f | %.1f | %.1f |' % (k, n, ms, ms / n * 1e3, 100 * ms / tot))
    # HBM traffic per conv_h16 launch at the config-5 per-GPU shape (FETCH_SIZE under-reports wide reads by 2x on gfx950; KiB units)
    import json
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from summarize_prof import pmc_sum, _tree_id
    fe = pmc_sum(os.path.join(src, 'c5_FETCH_SIZE', 'hp3d_counter_collection.csv'), 'FETCH_SIZE').get('conv_h16')
    wr = pmc_sum(os.path.join(src, 'c5_WRITE_SIZE', 'hp3d_counter_collection.csv'), 'WRITE_SIZE').get('conv_h16')
    if fe and wr and fe[1] and wr[1]:
        rdb, wtb = 2.0 * fe[0] * 1024 / fe[1], wr[0] * 1024 / wr[1]
        lines += ['', 'HBM traffic per conv_h16 launch at the config-5 per-GPU shape (B=128, 480x640; separate --pmc FETCH_SIZE / WRITE_SIZE passes, '
                  'FETCH_SIZE x 2): %.1f MB read + %.1f MB write = %.1f MB (bench.py quotes it as roofline.traffic for that workload).' % (rdb / 1e6, wtb / 1e6, (rdb + wtb) / 1e6)]
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        json.dump({"kernel": "conv_h16", "hbm_bytes_per_launch": rdb + wtb, "read_bytes": rdb, "write_bytes": wtb,
                   "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 per MI355X_MICROARCH.md, scripts/gpu_pmc_h16.sh",
                   "stamp": "profile tag %s, summarised %s, tree %s" % (tag, __import__('datetime').date.today().isoformat(), _tree_id()),
                   "workload": "ColorHandPose3DNetwork.inference, 480x640x3 f32 in HBM, 128 images/GPU/step", "dtype": "f16"},
                  open(os.path.join(root, 'profiles', 'conv_h16_traffic.json'), 'w'), indent=1)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', tag + '_h16_counters.md')
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
