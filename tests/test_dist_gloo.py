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


def test_single_process_degrades_to_identity():
    import torch
    sys.path.insert(0, ROOT)
    from examples.torch_dist_helpers import gather_keypoints
    x = torch.randn(3, 21, 3)
    assert gather_keypoints(x) is x
