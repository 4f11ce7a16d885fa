"""VALU -> MFMA operand hazard check over a hipcc -S listing (or llvm-objdump -d output): an MFMA that reads, as SrcA / SrcB, a VGPR
written by a VALU instruction fewer than `need` = 2 wait states earlier sees the OLD value on gfx950 (measured:
scripts/micro/mfma_operand_hazard.hip; hipcc pads its own MFMAs with s_nop 1, but cannot see MFMAs inside inline-asm statements).
Prints every violation; exit code 1 if any.
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
    """lines: instruction strings of ONE kernel in program order (labels / directives removed).  A VALU write of an MFMA's SrcA / SrcB register
    needs `need` wait states in front of the MFMA (measured on MI355X, scripts/micro/mfma_operand_hazard.hip: with 0 or 1 the MFMA reads the
    OLD value, with 2 the new one; an instruction is one wait state, `s_nop N` is N + 1 -- hipcc pads its own MFMAs with `s_nop 1`).
    Returns the violations as (index of the MFMA, wait states found, the VALU instruction, the MFMA)."""
    bad = []
    for i, l in enumerate(lines):
        if not l.startswith('v_mfma'):
            continue
        ops = [t.strip() for t in l.split(None, 1)[1].split(',')]
        src = regs(ops[1]) | regs(ops[2])
        states = 0
        k = i - 1
        while k >= 0 and states < need:
            p = lines[k]
            if p.startswith('v_') and not p.startswith(('v_mfma', 'v_cmp', 'v_readfirstlane', 'v_readlane')):
                dst = regs(p.split(None, 1)[1].split(',')[0].strip())
                if dst & src:
                    bad.append((i, states, p, l))
            m = re.match(r's_nop\s+(\d+)', p)
            states += int(m.group(1)) + 1 if m else 1
            k -= 1
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
            print('   #%d: "%s"  with %d wait state(s) in front of  "%s"' % (i, p, back, l))
        total += len(bad)
    sys.exit(1 if total else 0)
