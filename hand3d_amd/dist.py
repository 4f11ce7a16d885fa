"""Batch sharding across the GPUs of one node: every image is independent, so
the path shards with NO data-path collective.  torch.distributed (backend "nccl" == RCCL over
xGMI; "gloo" in the CPU tests) is used as plumbing for two tiny exchanges only:
  * once: broadcast of the packed weight blob (468 MB: direct + Winograd-domain filters + f16 section) from rank 0;
  * per batch: gather of the [B/n,21,3] keypoints (252 B/image).
The reference has no counterpart (single tf.Session everywhere, run.py:50).
"""
import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of rank `rank`; sizes differ by at most one."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(n_items, world):
    return [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]


def broadcast_blob(blob_tensor, src=0, group=None):
    """In-place broadcast of the packed weight blob (a flat torch tensor on the rank's device)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.broadcast(blob_tensor, src=src, group=group)
    return blob_tensor


def gather_keypoints(local_kp, n_total=None, group=None):
    """all_gather of per-rank keypoints [b_r,21,3] -> [sum b_r,21,3] in rank order (every rank gets
    the result; rank 0 is the consumer).  Ragged shards are padded to the largest shard."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local_kp
    world = dist.get_world_size(group)
    if n_total is None:
        sizes_t = [torch.zeros(1, dtype=torch.int64, device=local_kp.device) for _ in range(world)]
        dist.all_gather(sizes_t, torch.tensor([local_kp.shape[0]], dtype=torch.int64, device=local_kp.device), group=group)
        sizes = [int(s.item()) for s in sizes_t]
    else:
        sizes = shard_sizes(n_total, world)
    mx = max(sizes)
    pad = local_kp
    if local_kp.shape[0] < mx:
        pad = torch.cat([local_kp, local_kp.new_zeros((mx - local_kp.shape[0],) + tuple(local_kp.shape[1:]))], 0)
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad.contiguous(), group=group)
    return torch.cat([o[:s] for o, s in zip(outs, sizes)], 0)


def native_comm_init(engine, rank, world, group=None):
    """Set up the engine's own RCCL communicator (include/hp3d.h hp3d_comm_*): rank 0 draws the 128-byte id, the
    launcher's process group (any backend) only carries those bytes to the other ranks."""
    import torch.distributed as dist
    ids = [engine.comm_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(ids, src=0, group=group)
    engine.comm_init(rank, world, ids[0])


class ShardedPipeline(object):
    """One rank's share of a sharded batch: engine + device-resident I/O (torch tensors are used
    only as device memory).  `weights` is needed on rank 0 only."""

    def __init__(self, engine, rank=0, world=1, group=None):
        self.engine, self.rank, self.world, self.group = engine, rank, world, group

    def sync_weights_native(self, weights=None, dtype=0):
        """Same as sync_weights through the C ABI only: hp3d_comm_init + hp3d_bcast_weights (RCCL on the engine's
        stream, no torch tensors)."""
        if self.rank == 0:
            self.engine.load_weight_dict(weights)
            self.engine.finalize_weights(dtype)
        native_comm_init(self.engine, self.rank, self.world, self.group)
        self.engine.bcast_weights(0)

    def sync_weights(self, weights=None, device=None, dtype=0):
        import torch
        from . import _lib
        full = _lib.NET_SEG | _lib.NET_POSE | _lib.NET_PRIOR | _lib.NET_VP | (32 if dtype in (1, 'f16') else 0)
        if self.rank == 0:
            self.engine.load_weight_dict(weights)
            self.engine.finalize_weights(dtype)
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return
        n = (self.engine.blob_bytes() + 3) // 4
        blob = torch.empty(n, dtype=torch.float32, device=device)
        if self.rank == 0:
            self.engine.blob_export(blob.data_ptr())
        broadcast_blob(blob, 0, self.group)
        torch.cuda.synchronize(device)
        if self.rank != 0 or self.world == 1:     # world 1 (torchrun with one rank) re-imports its own blob: exercises the path
            self.engine.blob_import(blob.data_ptr(), full)
        del blob
