import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
EMU_DIR = os.path.join(ROOT, 'tests', 'emu')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU-interpreter run, skipped unless HP3D_SLOW=1")


def _has_gpu():
    return os.path.exists('/dev/kfd')


@pytest.fixture(scope='session')
def emu_engine():
    """Engine on the CPU interpreter of the kernel sources (tests/emu) -- CPU suite only."""
    sys.path.insert(0, EMU_DIR)
    try:
        import build_emu
        lib = build_emu.build()
    finally:
        sys.path.pop(0)
    from hand3d_amd._lib import Engine
    e = Engine(0, path=lib)
    yield e
    e.close()


@pytest.fixture(scope='session')
def gpu_engine():
    """Engine on the real libhp3d.so; fails loudly (never skips to a fallback) when marked gpu."""
    from hand3d_amd import _lib
    assert os.path.exists(_lib.DEFAULT_LIB), "libhp3d.so not built (python -m hand3d_amd.build)"
    e = _lib.Engine(0, path=_lib.DEFAULT_LIB)
    # test infrastructure only: HP3D_TEST_OPTIONS="key=value,key=value" applies hp3d_set_option to the suite's engine, so that a non-default
    # kernel policy (round 5: wino4_wide=1, the since-removed wide-item kernel) can be taken through the WHOLE GPU suite before it becomes the default
    for kv in filter(None, os.environ.get('HP3D_TEST_OPTIONS', '').split(',')):
        e.set_option(*kv.split('=', 1))
    yield e
    e.close()


@pytest.fixture(scope='session')
def synth_weights():
    from hand3d_amd import synth
    return synth.make_weights()
