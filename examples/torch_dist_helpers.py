"""torch.distributed variants of the two exchanges of the sharded pipeline -- for callers that ALREADY live in a torch process group.

Not part of the product package (BASELINE.json's north star: no PyTorch in the engine; `hand3d_amd/` does not import torch): the
native path is `hand3d_amd.dist.ShardedPipeline` (TCP rendezvous + RCCL behind the C ABI).  These helpers only move the packed weight
blob and the keypoints through `torch.distributed` (backend "nccl" = RCCL on ROCm, "gloo" in the CPU test) and hand plain pointers to
the engine.  Covered by tests/test_dist_gloo.py (two gloo ranks)."""
from hand3d_amd.dist import shard_sizes


def broadcast_blob(blob_tensor, src=0, group=None):
    """In-place broadcast of the packed weight blob (a flat torch tensor on the rank's device) inside an existing
    torch process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.broadcast(blob_tensor, src=src, group=group)
    return blob_tensor


def gather_keypoints(local_kp, n_total=None, group=None):
    """all_gather of per-rank keypoints [b_r,21,3] (torch tensors) -> [sum b_r,21,3] in rank order.  Ragged shards are
    padded to the largest shard."""
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


def sync_weights_torch(engine, rank, world, weights=None, device=None, dtype=0, group=None):
    """The weight exchange through torch.distributed (backend nccl = RCCL) for callers inside a torch process group:
    hp3d_weights_blob_export -> dist.broadcast -> hp3d_weights_blob_import, with rank 0's real nets mask."""
    import torch
    import torch.distributed as dist
    if rank == 0:
        engine.load_weight_dict(weights)
        engine.finalize_weights(dtype)
    if not (dist.is_available() and dist.is_initialized()):
        return
    mask = torch.tensor([engine.nets_mask() if rank == 0 else 0], dtype=torch.int64, device=device)
    dist.broadcast(mask, src=0, group=group)
    n = (engine.blob_bytes() + 3) // 4
    blob = torch.empty(n, dtype=torch.float32, device=device)
    if rank == 0:
        engine.blob_export(blob.data_ptr())
    broadcast_blob(blob, 0, group)
    if device is not None and getattr(device, 'type', 'cpu') == 'cuda':
        torch.cuda.synchronize(device)
    if rank != 0 or world == 1:
        engine.blob_import(blob.data_ptr(), int(mask.item()))
    del blob
