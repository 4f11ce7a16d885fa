"""Builds libhp3d.so (hand-written HIP for gfx950) in-tree with hipcc.

No torch, no cmake: one translation unit per kernel family + the executor -> one shared library with a C ABI (include/hp3d.h).
`-ffp-contract=off`: the glue kernels restate float32 op-by-op arithmetic of the reference
(box / interpolation coordinates feed discontinuous decisions); the MFMA conv is unaffected.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libhp3d.so')
SOURCES = ['conv_mfma.hip', 'conv_wino.hip', 'conv_wino2.hip', 'conv_wino4.hip', 'conv_wino4s.hip', 'conv_wino7.hip', 'conv_pw2.hip', 'conv_first.hip', 'conv_h16.hip', 'glue.hip', 'lift_fused.hip', 'engine.hip']
HEADERS = ['hp3d_common.h', 'lift_fused.h', 'wino4_shared.h', 'wino4_diag.h', os.path.join('..', '..', 'include', 'hp3d.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wall',
         '-Wno-unused-function', '-Wno-unused-result', '-Wno-unused-value']
# per-file additions.  conv_wino4.hip: its F(4x4,3x3) transforms are multiply-adds by 2, 4, 5, 8 -- contracted to (packed) FMAs they are
# a third fewer VALU instructions per step, and nothing in that file feeds a discontinuous decision (the later -ffp-contract wins)
# (-pragma-unroll-threshold: its 36-plane step body is one straight-line block by design; past the default limit hipcc silently
# stops unrolling and the 288 accumulators land in scratch)
EXTRA_FLAGS = {'conv_wino4.hip': ['-ffp-contract=fast', '-mllvm', '-pragma-unroll-threshold=100000'],
               'conv_wino4s.hip': ['-ffp-contract=fast', '-mllvm', '-pragma-unroll-threshold=100000'],     # (the split-operand form of the same kernel)
               'conv_wino7.hip': ['-ffp-contract=fast', '-mllvm', '-pragma-unroll-threshold=100000']}      # (F(4x4,4x4) for the 7x7 layers: the same reasons)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on %s' % src)
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
