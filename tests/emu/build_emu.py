"""Builds tests/emu/libhp3d_emu.so: the product kernel sources interpreted on the CPU (tests only)."""
import os
import subprocess

CXX = os.environ.get('HP3D_EMU_CXX', '/opt/rocm/lib/llvm/bin/clang++' if os.path.exists('/opt/rocm/lib/llvm/bin/clang++') else 'g++')
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'hand3d_amd', 'csrc')
LIB = os.path.join(HERE, 'libhp3d_emu.so')
SRCS = [os.path.join(CSRC, f) for f in ('conv_mfma.hip', 'conv_wino.hip', 'conv_wino2.hip', 'conv_wino4.hip', 'conv_wino4s.hip', 'conv_wino7.hip', 'conv_pw2.hip', 'conv_first.hip', 'conv_h16.hip', 'glue.hip', 'lift_fused.hip', 'engine.hip')] + [os.path.join(HERE, 'hp3d_emu.cpp')]
DEPS = SRCS + [os.path.join(CSRC, 'hp3d_common.h'), os.path.join(CSRC, 'lift_fused.h'), os.path.join(CSRC, 'wino4_shared.h'), os.path.join(CSRC, 'wino4_diag.h'), os.path.join(HERE, 'hp3d_emu.h'), os.path.join(ROOT, 'include', 'hp3d.h')]


def isa_flags():
    """The portable x86-64 baseline the interpreter's f16 conversions like (-mavx2 -mfma -mf16c), or nothing where the compiler does
    not take them (non-x86 hosts; the interpreter's plain-C++ paths are then used)."""
    flags = ['-mavx2', '-mfma', '-mf16c']
    try:
        r = subprocess.run([CXX, '-x', 'c++', '-std=c++17'] + flags + ['-fsyntax-only', '-'], input=b'int main() { return 0; }\n',
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=60)
        return flags if r.returncode == 0 else []
    except Exception:
        return []


def build(force=False):
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in DEPS):
        return LIB
    objs, procs = [], []
    isa = isa_flags()
    for s in SRCS:
        o = os.path.join(HERE, os.path.basename(s).rsplit('.', 1)[0] + '.emu.o')
        objs.append(o)
        procs.append(subprocess.Popen([CXX, '-x', 'c++', '-std=c++17', '-O2'] + isa + ['-fPIC', '-DHP3D_EMU', '-ffp-contract=off',
                                       '-fno-strict-aliasing', '-w', '-Wno-psabi', '-I', HERE, '-I', CSRC, '-c', s, '-o', o]))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError('emu build failed')
    subprocess.check_call([CXX, '-shared', '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    print(build(force=True))
