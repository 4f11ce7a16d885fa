"""N>1 path on CPU: two gloo ranks shard a batch, broadcast the weight blob, gather keypoints."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges():
    from hand3d_amd.dist import shard_range, shard_sizes
    assert [shard_range(256, r, 8) for r in range(8)] == [(32 * r, 32 * r + 32) for r in range(8)]
    assert shard_sizes(10, 4) == [3, 3, 2, 2]
    for n, w in [(1, 4), (7, 3), (1024, 8)]:
        rs = [shard_range(n, r, w) for r in range(w)]
        assert rs[0][0] == 0 and rs[-1][1] == n and all(a[1] == b[0] for a, b in zip(rs, rs[1:]))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from examples.torch_dist_helpers import broadcast_blob, gather_keypoints
    from hand3d_amd.dist import shard_range
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        # weights: rank 0 owns the packed blob, everyone ends up with identical bytes
        blob = torch.arange(1000, dtype=torch.float32) if rank == 0 else torch.zeros(1000)
        broadcast_blob(blob, 0)
        ok_blob = bool(torch.equal(blob, torch.arange(1000, dtype=torch.float32)))
        # a ragged global batch of 5 images: each rank "infers" keypoints for its shard only
        n_total = 5
        lo, hi = shard_range(n_total, rank, world)
        full = torch.arange(n_total * 63, dtype=torch.float32).reshape(n_total, 21, 3)
        gathered = gather_keypoints(full[lo:hi].clone(), n_total=n_total)
        gathered2 = gather_keypoints(full[lo:hi].clone())          # sizes exchanged instead of derived
        q.put((rank, ok_blob, bool(torch.equal(gathered, full)), bool(torch.equal(gathered2, full))))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_broadcast_and_gather():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] and r[2] and r[3] for r in res), res


def _product_worker(rank, world, gloo_port, rdzv_port, q):
    """One rank of the PRODUCT's sharded path (hand3d_amd.dist.ShardedPipeline) on the CPU interpreter engine, inside a gloo group."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'emu'))
    import torch
    import torch.distributed as dist
    import build_emu
    from hand3d_amd import synth
    from hand3d_amd._lib import Engine, Hp3dError
    from hand3d_amd.dist import Rendezvous, ShardedPipeline, shard_range
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(gloo_port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    eng = Engine(0, path=build_emu.build())
    sp = ShardedPipeline(eng, rank, world, Rendezvous(rank, world, '127.0.0.1', rdzv_port, timeout=120.0))
    try:
        # the interpreter build has no RCCL: the native set-up fails loudly on every rank, all ranks agree and switch (bench.py's protocol)
        err = None
        try:
            sp.comm_init()
        except (Hp3dError, NotImplementedError, RuntimeError) as e:
            err = '%s: %s' % (type(e).__name__, e)
        assert all(sp.rdzv.allgather(err)), "the interpreter build cannot have built a communicator"
        sp.use_tcp_only()
        # weights exist on rank 0 ONLY (the lifting nets: PosePrior + ViewpointNet)
        w = {k: v for k, v in synth.make_weights().items() if k.startswith(('PosePrior/', 'ViewpointNet/'))} if rank == 0 else None
        sp.sync_weights(w)
        n_total = 3                                     # ragged: shards of 2 and 1
        rng = np.random.default_rng(5)
        sm32 = (rng.standard_normal((n_total, 32, 32, 21)) * 0.3).astype(np.float32)
        hs = synth.hand_sides(n_total)
        lo, hi = shard_range(n_total, rank, world)
        rel, _, _ = eng.pose3d(sm32[lo:hi], hs[lo:hi])
        got = sp.gather_ragged(rel, n_total)
        # independent transport for the comparison: the same shards all-gathered by gloo
        pad = torch.zeros(2, 21, 3)
        pad[:hi - lo] = torch.from_numpy(rel)
        outs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(outs, pad)
        via_gloo = np.concatenate([outs[0].numpy()[:2], outs[1].numpy()[:1]], 0)
        full = eng.pose3d(sm32, hs)[0] if rank == 0 else None          # the single-process result (rank 0 computes it once)
        q.put((rank, got, via_gloo, full))
    finally:
        sp.close()
        eng.close()
        dist.destroy_process_group()


def test_two_rank_product_pipeline_on_interpreter_engine():
    """World 2 through hand3d_amd.dist.ShardedPipeline itself: TCP rendezvous, the RCCL set-up failing loudly on
    the interpreter build and all ranks switching together, weights from rank 0 only, ragged shards (2 + 1), the gathered keypoints equal
    to the single-process result on every rank and to the same shards gathered by gloo."""
    import torch.multiprocessing as mp

    def free_port():
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        p = s.getsockname()[1]
        s.close()
        return p
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    gp, rp = free_port(), free_port()
    procs = [ctx.Process(target=_product_worker, args=(r, 2, gp, rp, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    full = res[0][3]
    assert full is not None and full.shape == (3, 21, 3)
    for rank, got, via_gloo, _ in res:
        assert got.shape == (3, 21, 3) and np.array_equal(got, via_gloo), rank
        assert np.array_equal(got, full), "rank %d: gathered shards != the single-process result" % rank


def test_single_process_degrades_to_identity():
    import torch
    sys.path.insert(0, ROOT)
    from examples.torch_dist_helpers import gather_keypoints
    x = torch.randn(3, 21, 3)
    assert gather_keypoints(x) is x
