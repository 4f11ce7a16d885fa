"""Builds tests/emu/libhp3d_emu.so: the product kernel sources interpreted on the CPU (tests only)."""
import os
import subprocess

CXX = os.environ.get('HP3D_EMU_CXX', '/opt/rocm/lib/llvm/bin/clang++' if os.path.exists('/opt/rocm/lib/llvm/bin/clang++') else 'g++')
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'hand3d_amd', 'csrc')
LIB = os.path.join(HERE, 'libhp3d_emu.so')
SRCS = [os.path.join(CSRC, f) for f in ('conv_mfma.hip', 'conv_wino.hip', 'conv_wino2.hip', 'conv_wino4.hip', 'conv_first.hip', 'conv_h16.hip', 'glue.hip', 'lift_fused.hip', 'engine.hip')] + [os.path.join(HERE, 'hp3d_emu.cpp')]
DEPS = SRCS + [os.path.join(CSRC, 'hp3d_common.h'), os.path.join(CSRC, 'lift_fused.h'), os.path.join(HERE, 'hp3d_emu.h'), os.path.join(ROOT, 'include', 'hp3d.h')]


def build(force=False):
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in DEPS):
        return LIB
    objs, procs = [], []
    for s in SRCS:
        o = os.path.join(HERE, os.path.basename(s).rsplit('.', 1)[0] + '.emu.o')
        objs.append(o)
        procs.append(subprocess.Popen([CXX, '-x', 'c++', '-std=c++17', '-O2', '-mavx2', '-mfma', '-mf16c', '-fPIC', '-DHP3D_EMU', '-ffp-contract=off',
                                       '-fno-strict-aliasing', '-w', '-Wno-psabi', '-I', HERE, '-I', CSRC, '-c', s, '-o', o]))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError('emu build failed')
    subprocess.check_call([CXX, '-shared', '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    print(build(force=True))
