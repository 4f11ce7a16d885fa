"""VALU -> MFMA operand hazard check over a hipcc -S listing (or llvm-objdump -d output): an MFMA that reads, as SrcA / SrcB, a VGPR
written by a VALU instruction fewer than `need` issue slots earlier sees the old value on gfx950 (measured: scripts/micro/split_unit.hip;
hipcc covers it for its own MFMAs, but not for MFMAs inside inline-asm statements).  Prints every violation; exit code 1 if any.
usage: python scripts/mfma_hazard_check.py file.s [kernel-substring] [need=2]"""
import re
import sys


def regs(tok):
    """'v[12:15]' / 'v7' -> set of VGPR numbers; anything else -> empty"""
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'v(\d+)', tok)
    return {int(m.group(1))} if m else set()


def check(lines, need=2):
    """lines: instruction strings of ONE kernel in program order (labels / directives removed). Returns the violations."""
    bad = []
    for i, l in enumerate(lines):
        if not l.startswith('v_mfma'):
            continue
        ops = [t.strip() for t in l.split(None, 1)[1].split(',')]
        src = regs(ops[1]) | regs(ops[2])
        for back in range(1, need + 1):
            if i - back < 0:
                break
            p = lines[i - back]
            if not p.startswith('v_') or p.startswith('v_mfma') or p.startswith('v_cmp') or p.startswith('v_readfirstlane') or p.startswith('v_readlane'):
                continue
            dst = regs(p.split(None, 1)[1].split(',')[0].strip())
            if dst & src:
                bad.append((i, back, p, l))
    return bad


def kernels_of(txt):
    out = {}
    cur = None
    for line in txt.splitlines():
        m = re.match(r'^(?:[0-9a-f]+ <)?(_Z\w+)>?:', line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        s = line.strip()
        if cur is None or not s or s.startswith((';', '.', '//')) or s.endswith(':'):
            continue
        s = re.sub(r'\s*//.*$', '', s)          # objdump's trailing address / encoding comment
        s = re.sub(r'\s*;.*$', '', s)
        if s:
            cur.append(s)
    return out


if __name__ == '__main__':
    txt = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ''
    need = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    total = 0
    for name, lines in kernels_of(txt).items():
        if want not in name:
            continue
        bad = check(lines, need)
        n_mfma = sum(1 for l in lines if l.startswith('v_mfma'))
        print('%s: %d MFMAs, %d violations' % (name[:100], n_mfma, len(bad)))
        for i, back, p, l in bad[:10]:
            print('   #%d: "%s"  %d slot(s) before  "%s"' % (i, p, back, l))
        total += len(bad)
    sys.exit(1 if total else 0)
