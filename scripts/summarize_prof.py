"""Condenses a gpu_round.sh output directory (rocprofv3 CSVs) into profiles/<tag>_*.{csv,md}.

  python scripts/summarize_prof.py gpurun_out/r01b r01

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are
collected in separate --pmc passes; units are KiB; on gfx950 FETCH_SIZE under-reports wide coalesced
reads by exactly 2x, so read bytes = 2 * FETCH_SIZE * 1024 (WRITE_SIZE taken as reported).
"""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict


def _tree_id():
    if os.environ.get('HP3D_TREE_ID'):
        return os.environ['HP3D_TREE_ID']
    root0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if os.path.exists(os.path.join(root0, '.tree_id')):          # written next to the snapshot before a gpurun visit (the GPU box has no .git)
        return open(os.path.join(root0, '.tree_id')).read().strip() or 'unknown'
    try:
        import subprocess
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        return subprocess.check_output(['git', 'rev-parse', '--short', 'HEAD'], cwd=root, text=True, stderr=subprocess.DEVNULL).strip()
    except Exception:
        return 'unknown'


def family(name):
    if 'conv_wino_kernel' in name:
        return 'conv_wino'
    if 'conv_wino2_kernel' in name:
        return 'conv_wino2'
    if 'conv_wino4s_kernel' in name:      # (round 6: F(4x4,3x3) with split bf16x3 operands, option wino4_split; off by default)
        return 'conv_wino4s'
    if 'conv_wino4_kernel' in name or 'conv_wino7_kernel' in name:      # (bench.py's family: Winograd with 4x4 output tiles, F(4x4,3x3) and the 7x7 layers' F(4x4,4x4))
        return 'conv_wino4'
    if 'conv_pw2_kernel' in name:
        return 'conv_pw2'
    if 'lift_fused_kernel' in name:
        return 'lift_fused'
    if 'conv_mfma_kernel' in name:
        return 'conv_mfma'
    if 'conv_first_kernel' in name:
        return 'conv_first'
    if 'conv_h16_kernel' in name:
        return 'conv_h16'
    if 'touch_kernel' in name:          # (the read pass in front of HandSegNet's conv1_1: a cold input image through the memory-side cache)
        return 'conv_first_touch'
    if 'wino4_tail_r' in name:
        return 'wino4_tail_reduce'
    for k in ('conv_splitk_reduce', 'preprocess_u8', 'bone_rel_inv', 'fc_partial', 'fc_reduce', 'fc_tail', 'fc_kernel', 'im2col3x3', 'mask_grow', 'resize_bilinear', 'seg_upsample_softmax', 'crop_and_resize',
              'kp_detect', 'copy_channels', 'concat_handside', 'lift_epilogue', 'avgpool8', 'pad_channels'):
        if k in name:
            return k
    return 'other:' + name[:40]


def pmc_sum(path, counter):
    agg = defaultdict(lambda: [0.0, 0])
    if not os.path.exists(path):
        return agg
    with open(path) as f:
        for r in csv.DictReader(f):
            if r['Counter_Name'] != counter:
                continue
            a = agg[family(r['Kernel_Name'])]
            a[0] += float(r['Counter_Value'])
            a[1] += 1
    return agg


def main():
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dst = os.path.join(root, 'profiles')
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, 'prof', 'hp3d_kernel_stats.csv'), os.path.join(dst, tag + '_kernel_stats.csv'))
    for f in ('bench.json', 'bench_layers.txt', 'device.txt'):
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), os.path.join(dst, tag + '_' + f))
    fam = defaultdict(lambda: [0, 0.0])
    with open(os.path.join(src, 'prof', 'hp3d_kernel_stats.csv')) as f:
        for r in csv.DictReader(f):
            a = fam[family(r['Name'])]
            a[0] += int(r['Calls'])
            a[1] += float(r['TotalDurationNs'])
    total = sum(v[1] for v in fam.values())
    fetch = pmc_sum(os.path.join(src, 'pmc_FETCH_SIZE', 'hp3d_counter_collection.csv'), 'FETCH_SIZE')
    write = pmc_sum(os.path.join(src, 'pmc_WRITE_SIZE', 'hp3d_counter_collection.csv'), 'WRITE_SIZE')
    bench = {}
    if os.path.exists(os.path.join(src, 'bench.json')):
        try:
            bench = json.loads(open(os.path.join(src, 'bench.json')).read().strip().splitlines()[-1])
        except Exception:
            pass
    wl_src = bench
    if not wl_src and os.path.exists(os.path.join(src, 'prof_bench.json')):      # (run before the bench line exists: the profiled command's own line)
        try:
            wl_src = json.loads(open(os.path.join(src, 'prof_bench.json')).read().strip().splitlines()[-1])
        except Exception:
            wl_src = {}
    lines = ['# rocprofv3 summary %s' % tag, '',
             'command: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --gpus 1 --steps 10 --warmup 3 --cpu-seconds 0 --no-host-path --no-other-configs --option streams=1`',
             '(one HIP stream, so that kernel durations are not inflated by the overlap the default two-stream mode is there to create;',
             ' the bench line below is the DEFAULT command, whose roofline block comes from its own serial, event-timed pass)',
             '(PMC: separate `--kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, `--steps 2 --warmup 1`)', '',
             '| kernel family | launches | total ms | avg launch us | % GPU time | HBM read MB/launch (2x FETCH_SIZE) | HBM write MB/launch |',
             '|---|---|---|---|---|---|---|']
    for k, (calls, ns) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        fr = fetch.get(k)
        wr = write.get(k)
        rd = '%.2f' % (2.0 * fr[0] * 1024 / fr[1] / 1e6) if fr and fr[1] else '-'
        wt = '%.2f' % (wr[0] * 1024 / wr[1] / 1e6) if wr and wr[1] else '-'
        lines.append('| %s | %d | %.2f | %.1f | %.2f | %s | %s |' % (k, calls, ns / 1e6, ns / calls / 1e3, 100 * ns / total, rd, wt))
    if bench:
        r = bench.get('roofline', {})
        lines += ['', 'bench line of the same box: value %.1f %s, ms_per_step %.2f; roofline %s achieved %.1f / peak %.1f %s = %.3f '
                  '(avg launch %.4f ms from HIP events; rocprof avg for that family above must agree)'
                  % (bench['value'], bench['unit'], bench['ms_per_step'], r.get('kernel'), r.get('achieved', 0), r.get('peak', 0),
                     r.get('unit'), r.get('frac', 0), r.get('avg_launch_ms', 0))]
    for kf in ('conv_wino4', 'conv_wino', 'conv_mfma'):
        if wl_src and kf in fetch and fetch[kf][1]:
            rdb = 2.0 * fetch[kf][0] * 1024 / fetch[kf][1]
            wtb = write[kf][0] * 1024 / write[kf][1] if write[kf][1] else 0
            lines.append('%s HBM traffic per launch (PMC): %.1f MB read + %.1f MB write = %.1f MB' % (kf, rdb / 1e6, wtb / 1e6, (rdb + wtb) / 1e6))
            json.dump({"kernel": kf, "hbm_bytes_per_launch": rdb + wtb, "read_bytes": rdb, "write_bytes": wtb,
                       "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 per MI355X_MICROARCH.md, profile tag " + tag,
                       "stamp": "profile tag %s, summarised %s, tree %s" % (tag, __import__('datetime').date.today().isoformat(), _tree_id()),
                       "workload": wl_src.get('config', {}).get('workload'), "dtype": wl_src.get('dtype', 'f32')},
                      open(os.path.join(dst, kf + '_traffic.json'), 'w'), indent=1)
    open(os.path.join(dst, tag + '_summary.md'), 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
