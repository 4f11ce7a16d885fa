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


MFMA_PASSES = (('32x32x16', 8), ('16x16x32', 4), ('32x32x2_f32', 16), ('32x32x2f32', 16), ('16x16x4_f32', 8), ('16x16x4f32', 8), ('32x32x8', 16), ('16x16x16', 8))


def check_result(lines, extra=4):
    """The other direction (round 6, conv_h16_first_kernel's asm MFMAs with VGPR accumulators): a VALU / LDS / VMEM instruction that READS a VGPR an
    MFMA wrote needs passes + `extra` wait states behind that MFMA (hipcc counts them for its own MFMAs -- its `s_nop` in front of the first
    v_accvgpr_read / VALU read of a result -- and cannot for MFMAs inside asm statements; another MFMA reading the result as SrcC is interlocked
    by hardware).  Returns (index of the MFMA, wait states found, needed, the reader, the MFMA)."""
    bad = []
    for i, l in enumerate(lines):
        if not l.startswith('v_mfma'):
            continue
        ops = [t.strip() for t in l.split(None, 1)[1].split(',')]
        dst = regs(ops[0])
        if not dst:
            continue                     # accumulator in AGPRs: read back by v_accvgpr_read, which the compiler schedules itself
        passes = next((n for key, n in MFMA_PASSES if key in l.split()[0]), 16)
        need = passes + extra
        states = 0
        for k in range(i + 1, len(lines)):
            if states >= need:
                break
            p = lines[k]
            m = re.match(r's_nop\s+(\d+)', p)
            if m:
                states += int(m.group(1)) + 1
                continue
            if p.startswith(('s_barrier', 's_cbranch', 's_branch', 's_endpgm')):
                break                    # (control flow: not followed; a barrier is not counted on either)
            toks = p.split(None, 1)
            if len(toks) == 2 and not p.startswith('v_mfma') and p[0] in 'vdbg':     # VALU, ds_*, buffer_*, global_*
                srcs = [t.strip() for t in toks[1].split(',')]
                if p.startswith('v_') or '_read' in toks[0] or '_load' in toks[0]:
                    srcs = srcs[1:]                                                 # first operand of a VALU instruction / a load is its destination
                read = set()
                for t in srcs:
                    read |= regs(t.split()[0]) if t else set()
                if read & dst:
                    bad.append((i, states, need, p, l))
                    break
            if p.startswith('v_mfma') and regs(p.split(None, 1)[1].split(',')[0].strip()) & dst:
                break                    # overwritten by a later MFMA: that one's own check takes over
            states += 1
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
