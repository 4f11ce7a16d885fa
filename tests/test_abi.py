"""The C-ABI library loads and exports every symbol include/hp3d.h declares; the ctypes binding
covers all of them; without a GPU the product path fails loudly (no CPU fallback); the product
package never touches the oracle."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'hp3d.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(hp3d_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from hand3d_amd import _lib, build
    lib = build.build(verbose=False)        # hipcc cross-compiles gfx950 without a GPU
    dl = ctypes.CDLL(lib)
    syms = _declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(dl, s), "libhp3d.so does not export %s" % s
    assert sorted(_lib.EXPORTS) == syms, "hand3d_amd/_lib.py and include/hp3d.h disagree"
    assert dl.hp3d_abi_version() == 1


def test_library_contains_gfx950_mfma_code(tmp_path):
    import shutil
    from hand3d_amd import _lib
    lib = str(tmp_path / 'libhp3d.so')          # llvm-objdump --offloading drops the extracted bundles next to its input
    shutil.copy(_lib.DEFAULT_LIB, lib)
    out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump', '--offloading', lib],
                         capture_output=True, text=True).stdout
    assert 'gfx950' in out


@pytest.mark.skipif(os.path.exists('/dev/kfd'), reason="GPU present")
def test_no_cpu_fallback_without_gpu():
    from hand3d_amd import _lib
    with pytest.raises(_lib.Hp3dError):
        _lib.Engine(0, path=_lib.DEFAULT_LIB)
    with pytest.raises(_lib.Hp3dError):
        _lib.load('/nonexistent/libhp3d.so')


def test_product_never_imports_oracle_or_frameworks():
    bad = re.compile(r'^\s*(from|import)\s+(oracle|torch|tensorflow|triton)\b', re.M)
    pkg = os.path.join(ROOT, 'hand3d_amd')
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if not f.endswith(('.py', '.hip', '.h', '.cpp')):
                continue
            src = open(os.path.join(dp, f)).read()
            m = bad.search(src)
            if m and not (f == 'dist.py' and 'torch' in m.group(0)):    # dist.py: torch.distributed plumbing only
                raise AssertionError("%s: %s" % (os.path.join(dp, f), m.group(0)))
    code = "import sys; sys.path.insert(0, %r); import hand3d_amd, hand3d_amd.nets, hand3d_amd.utils.general; " \
           "assert 'oracle' not in sys.modules and 'torch' not in sys.modules and 'tensorflow' not in sys.modules" % ROOT
    subprocess.check_call([sys.executable, '-c', code])
