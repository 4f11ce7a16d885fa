"""The torch-free N>1 plumbing (hand3d_amd/dist.py): TCP rendezvous that carries the RCCL id and the benchmark's host
scalars, ragged shard bookkeeping, and -- on the GPU box -- the native RCCL exchange itself."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rdzv_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from hand3d_amd.dist import Rendezvous
    try:
        r = Rendezvous(rank, world, '127.0.0.1', port, timeout=120)
        got = r.allgather({'rank': rank, 'x': np.arange(3) + rank})
        uid = r.broadcast(bytes(range(128)) if rank == 0 else None, 0)
        r.barrier()
        mx = r.max(0.5 + rank)
        late = r.broadcast('from-2' if rank == 2 else None, src=2) if world > 2 else 'from-2'
        r.close()
        q.put((rank, [g['rank'] for g in got], [int(g['x'][2]) for g in got], uid == bytes(range(128)), mx, late))
    except Exception as e:      # pragma: no cover
        q.put((rank, 'ERR', repr(e)))


@pytest.mark.parametrize('block_first_port', [False, True])
def test_tcp_rendezvous_three_ranks(block_first_port):
    from hand3d_amd.dist import rendezvous_ports
    port = _free_port()
    blocker = None
    if block_first_port:        # a foreign listener on the first candidate port: rank 0 moves on, the others notice
        blocker = socket.socket()
        blocker.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        try:
            blocker.bind(('127.0.0.1', rendezvous_ports(port)[0]))
            blocker.listen(4)
        except OSError:
            pytest.skip("candidate port already taken on this box")
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    world = 3
    procs = [ctx.Process(target=_rdzv_worker, args=(r, world, port, q)) for r in range(world)]
    for p in reversed(procs):       # rank 0 last: the spokes have to retry
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(30)
    if blocker:
        blocker.close()
    assert all(len(r) == 6 for r in res), res
    for rank, ranks, xs, uid_ok, mx, late in res:
        assert ranks == [0, 1, 2] and xs == [2, 3, 4] and uid_ok and mx == 2.5 and late == 'from-2', res


def test_world_one_needs_no_socket_and_ragged_gather(emu_engine):
    from hand3d_amd.dist import Rendezvous, ShardedPipeline
    r = Rendezvous(0, 1)
    assert r.allgather(7) == [7] and r.max(1.5) == 1.5 and r.broadcast('a') == 'a'
    sp = ShardedPipeline(emu_engine, 0, 1, r)
    kp = np.arange(5 * 63, dtype=np.float32).reshape(5, 21, 3)
    assert np.array_equal(sp.gather_ragged(kp, 5), kp)
    buf = emu_engine.to_device(kp)
    assert np.array_equal(sp.gather_keypoints(buf, 5), kp)
    buf.free()
    with pytest.raises(ValueError):
        ShardedPipeline(emu_engine, 0, 2)


def _rccl_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    try:
        from hand3d_amd import Engine, synth
        from hand3d_amd.dist import Rendezvous, ShardedPipeline
        eng = Engine(0)                     # both ranks on the one GPU the box has
        sp = ShardedPipeline(eng, rank, world, Rendezvous(rank, world, '127.0.0.1', port, timeout=120))
        w = synth.make_weights() if rank == 0 else None
        os.environ.setdefault('HP3D_RCCL_TIMEOUT', '90')     # the product's own deadline around ncclCommInitRank / the first broadcast
        sp.sync_weights(w)
        q.put((rank, 'STAGE', 'synced', None))              # the communicator is up and has carried the weight blob
        img = synth.make_batch(40 + rank, 1, 240, 320)
        hs = synth.hand_sides(1)
        d_img, d_hs, d_c = eng.to_device(img), eng.to_device(hs), eng.dev_alloc(63 * 4)
        eng.infer_full_dev(1, 240, 320, int(d_img), int(d_hs), coord3d=int(d_c))
        eng.sync()
        mine = eng.to_host(d_c, (1, 21, 3))
        allk = sp.gather_keypoints(d_c, 1)
        sp.close()
        q.put((rank, 'OK', mine, allk))
    except Exception as e:
        q.put((rank, 'ERR', repr(e), None))


@pytest.mark.gpu
def test_native_rccl_two_ranks_on_one_gpu(gpu_engine, synth_weights):
    """Two processes, two contexts on the SAME device, the engine's own RCCL communicator of size 2: weight-blob
    broadcast from rank 0, per-rank inference, keypoint all-gather.  RCCL builds that refuse two ranks on one device
    make this an expected failure (reported, not hidden): the 8-GPU run is then the first multi-rank exercise."""
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    import queue as _queue
    import time as _time
    res, synced, hung = [], set(), False
    deadline = _time.monotonic() + 240          # both ranks together (a fresh box pages the image in for up to two minutes)
    try:
        while len(res) < 2:
            try:
                m = q.get(timeout=max(1.0, deadline - _time.monotonic()))
            except _queue.Empty:
                hung = True
                break
            if m[1] == 'STAGE':
                synced.add(m[0])
            else:
                res.append(m)
    finally:
        for p in procs:
            p.join(5 if hung else 30)
            if p.is_alive():
                p.kill()
    # Expected failure ONLY while no rank got a working communicator: RCCL refusing two ranks on one device, or its set-up running into
    # the product's deadline (hand3d_amd/dist.py, HP3D_RCCL_TIMEOUT).  A rank that hangs or fails AFTER the communicator carried the
    # weights is a defect of the exchange and fails the test.
    assert not (hung and synced), "ranks %s had a working communicator and the run still hung (%d of 2 answered)" % (sorted(synced), len(res))
    if hung:
        pytest.xfail("this RCCL did not complete a 2-rank communicator on one device within the deadline (%d of 2 ranks answered)" % len(res))
    res.sort(key=lambda t: t[0])
    if any(r[1] != 'OK' for r in res):
        msg = '; '.join(str(r[2]) for r in res if r[1] != 'OK')
        setup = 'uplicate' in msg or 'invalid usage' in msg.lower() or 'ncclCommInitRank' in msg or 'rccl init timeout' in msg
        if setup and not synced:
            pytest.xfail("no 2-rank communicator on one device with this RCCL: %s" % msg[:400])
        raise AssertionError(msg)
    # rank 1 never saw the weight dictionary: its keypoints must equal what the session engine computes for its image
    from hand3d_amd import synth
    for r in range(2):
        exp = gpu_engine.infer_full(synth.make_batch(40 + r, 1, 240, 320), synth.hand_sides(1), outputs=('coord3d',))['coord3d']
        assert np.abs(res[r][2] - exp).max() < 1e-6
        assert np.array_equal(res[0][3][r], res[r][2][0]) and np.array_equal(res[1][3][r], res[r][2][0])


def test_wire_codec_round_trip_and_depth_cap():
    """The rendezvous codec carries None / bool / int / float / str / bytes / ndarray / list / map and nothing else, and refuses
    messages nested deeper than its cap (a peer must not be able to run the receiver out of stack)."""
    import struct
    from hand3d_amd import dist
    msg = {'a': [1, 2.5, None, True, 'x', b'\x00\x01', np.arange(6, dtype=np.float32).reshape(2, 3)], 'b': {'c': [[], {}]}}
    parts = []
    dist._enc(msg, parts)
    buf = b''.join(parts)
    back, pos = dist._dec(buf, 0)
    assert pos == len(buf) and back['a'][:6] == msg['a'][:6] and np.array_equal(back['a'][6], msg['a'][6]) and back['b'] == msg['b']
    deep = b''.join(b'L' + struct.pack('<Q', 1) for _ in range(dist._MAX_DEPTH + 2)) + b'N'
    with pytest.raises(ValueError, match='nested deeper'):
        dist._dec(deep, 0)
    ok = b''.join(b'L' + struct.pack('<Q', 1) for _ in range(dist._MAX_DEPTH)) + b'N'
    assert dist._dec(ok, 0)[1] == len(ok)
    with pytest.raises(TypeError):
        dist._enc(object(), [])
