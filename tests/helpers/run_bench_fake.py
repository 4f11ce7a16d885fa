"""Runs bench.py's main() with a stand-in Engine (no GPU, no kernels): exercises the launcher protocol of the benchmark --
environment parsing, TCP rendezvous, weight-sync call order, barrier / max-over-ranks, per-step gather on every rank, the
rank-0-only JSON line -- under two real processes (tests/test_bench_protocol.py)."""
import os
import runpy
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hand3d_amd  # noqa: E402
from hand3d_amd import _lib  # noqa: E402


class FakeBuf(object):
    def __init__(self, n):
        self.a = np.zeros(max(int(n), 1), np.uint8)
        self.ptr = self.a.ctypes.data

    def __int__(self):
        return self.ptr

    __index__ = __int__

    def free(self):
        pass


class FakeLib(object):
    calls = 0

    def hp3d_posenet2d_dev(self, *a):
        time.sleep(0.001)
        return 0

    def hp3d_infer_full_kp_u8(self, h, B, Hin, Win, img, H, W, hs, sm, crop, scale, center, kpmap, c3, mask, kpc, khw):
        import ctypes
        FakeLib.calls += 1
        ctypes.memset(c3, FakeLib.calls & 1, 4)          # (two alternating batches give two different results)
        time.sleep(0.002)
        return 0


class FakeEngine(object):
    log = []

    def __init__(self, device=0, path=None):
        if os.environ.get('HP3D_FAKE_DIE_RANK') == os.environ.get('RANK', '-'):
            raise RuntimeError('no HIP device visible (fake): rank %s dies before the rendezvous' % os.environ.get('RANK'))
        self.device, self.h, self.lib = device, 1, FakeLib()
        self.rank = self.world = None
        self.prof = 0

    def _chk(self, rc):
        assert rc == 0

    def _rec(self, what):
        FakeEngine.log.append(what)

    def load_weight_dict(self, w):
        self._rec('load %d' % len(w))

    def finalize_weights(self, dtype=0):
        self._rec('finalize')

    def comm_unique_id(self):
        return bytes(range(128))

    def comm_init(self, rank, world, uid):
        if os.environ.get('HP3D_FAKE_RCCL_FAIL') == 'all' or os.environ.get('HP3D_FAKE_RCCL_FAIL') == str(rank):
            raise RuntimeError('ncclCommInitRank: unhandled system error (fake)')
        if os.environ.get('HP3D_FAKE_RCCL_HANG') in ('all', str(rank)):
            time.sleep(1e6)               # ncclCommInitRank that neither returns nor fails (seen once on a real box, round 5)
        assert uid == bytes(range(128)), "the rendezvous must deliver rank 0's id unchanged"
        self.rank, self.world = rank, world
        self._rec('comm_init %d/%d' % (rank, world))

    def bcast_weights(self, root=0):
        if os.environ.get('HP3D_FAKE_RCCL_HANG') == 'bcast':
            time.sleep(1e6)               # a communicator that came up and whose first collective never completes
        assert self.world is not None
        self._rec('bcast')

    def comm_destroy(self):
        self._rec('comm_destroy')

    def set_option(self, k, v):
        pass

    def close(self):
        pass

    def handsegnet(self, image, want_small=False):
        B, H, W, _ = image.shape
        return np.zeros((B, H, W, 2), np.float32), np.zeros((B, H // 8, W // 8, 2), np.float32)

    def to_device(self, a):
        return FakeBuf(np.asarray(a).nbytes)

    def dev_alloc(self, n):
        return FakeBuf(n)

    def to_host(self, buf, shape, dtype=np.float32, offset_bytes=0):
        return np.zeros(shape, dtype)

    def pinned_empty(self, shape, dtype=np.float32):
        return np.zeros(shape, dtype)

    def upload_async(self, *a):
        pass

    def wait_upload(self):
        pass

    def infer_full_dev(self, *a, **k):
        time.sleep(0.002)

    def sync(self):
        pass

    def set_profiling(self, on):
        self.prof = on

    def profile(self):
        return [('HandSegNet/conv3_2', 'conv_wino_f2x2_3x3', 1.0, 1.0e11, 1.0e8), ('HandSegNet/conv1_1', 'conv_first_3x3_c3', 0.2, 1.0e9, 7.0e8)]

    def allgather_dev(self, buf, count, world):
        assert self.world == world
        return np.zeros(world * count, np.float32)

    def counter(self, name):
        return (self.world or 0) if name == 'comm_ranks' else 0


os.environ['HP3D_BENCH_ENTRY'] = os.path.abspath(__file__)      # bench.py's self-launch starts THIS script per rank
hand3d_amd.Engine = FakeEngine
_lib.Engine = FakeEngine
_lib.device_count = lambda path=None: int(os.environ.get('HP3D_FAKE_DEVICES', '64'))     # (the stand-in box has as many devices as a test says)
sys.argv = ['bench.py'] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, 'bench.py'), run_name='__main__')
sys.stderr.write('FAKELOG rank %s: %s\n' % (os.environ.get('RANK', '-'), ' | '.join(FakeEngine.log)))
