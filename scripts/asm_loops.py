"""Per-kernel summary of a hipcc -S listing: MFMA runs, scratch / waterfall instructions inside them.
usage: python scripts/asm_loops.py file.s [kernel-substring]"""
import re
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ''
starts = [(m.start(), m.group(1)) for m in re.finditer(r'^(_Z\S+):\s*;\s*@', txt, re.M)]
for n, (pos, name) in enumerate(starts):
    if want not in name:
        continue
    body = txt[pos:starts[n + 1][0] if n + 1 < len(starts) else len(txt)]
    lines = [l.strip() for l in body.splitlines()]
    ops = [l.split()[0] if l and not l.startswith((';', '.')) and not l.endswith(':') else '' for l in lines]
    mf = [i for i, o in enumerate(ops) if o.startswith('v_mfma')]
    sc = [i for i, o in enumerate(ops) if o.startswith('scratch_')]
    wf = [i for i, o in enumerate(ops) if o == 's_cbranch_execnz']
    print(name[:90], '| instr', sum(1 for o in ops if o), 'mfma', len(mf), 'scratch', len(sc), 'execnz', len(wf))
    runs = []
    for i in mf:
        if runs and i - runs[-1][1] < 150:
            runs[-1][1] = i
        else:
            runs.append([i, i])
    for r in runs:
        seg = ops[r[0]:r[1] + 1]
        import collections
        c = collections.Counter('mfma' if o.startswith('v_mfma') else 'valu' if o.startswith('v_') else 'wait' if o == 's_waitcnt' else
                                'nop' if o == 's_nop' else 'salu' if o.startswith('s_') else o.split('_')[0] + '_' + (o.split('_') + [''])[1] for o in seg if o)
        print('   run %d..%d: %s | scratch inside: %d' % (r[0], r[1], dict(c), sum(1 for i in sc if r[0] <= i <= r[1])))
