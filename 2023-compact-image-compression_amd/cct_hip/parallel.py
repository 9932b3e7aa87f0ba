"""Multi-GPU sharding of a slice corpus: one process per GPU, contiguous slice ranges, and the
path's only exchange step -- an all-gather of per-slice compressed sizes (RCCL over xGMI when
the process group is "nccl"; gloo on CPU for tests).  No pixel or payload byte crosses GPUs:
slices are independent units (the reference treats them so too, scripts/evaluate.py:107-119).
"""
import numpy as np


def shard_range(n_total, rank, world):
    """Contiguous slice range [lo, hi) owned by `rank`: balanced, the first n % G ranks own one slice more
    (3954 slices over 8 GPUs = shards of 495/494)."""
    if world <= 0:
        return 0, n_total
    per, extra = divmod(n_total, world)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


def gather_sizes(local_sizes, dist=None, local_rank=0, counts=None):
    """All ranks' per-slice compressed sizes, in global slice order.

    local_sizes: uint32 array of this rank's sizes.  `dist` is torch.distributed (initialised) or
    None for single-process runs.  Ranks may own different counts (last shard shorter): arrays are
    padded to the longest shard for the collective and trimmed afterwards.
    """
    local_sizes = np.ascontiguousarray(local_sizes, dtype=np.uint32)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_sizes.copy()
    import torch
    world = dist.get_world_size()
    on_gpu = dist.get_backend() == "nccl"
    dev = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")
    if counts is None:
        cnt = torch.tensor([local_sizes.size], dtype=torch.int64, device=dev)
        all_cnt = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(all_cnt, cnt)
        counts = [int(c.item()) for c in all_cnt]
    width = max(counts) if counts else 0
    buf = torch.zeros(max(width, 1), dtype=torch.int64, device=dev)
    if local_sizes.size:
        buf[: local_sizes.size] = torch.from_numpy(local_sizes.astype(np.int64)).to(dev)
    out = torch.empty(world * buf.numel(), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(out, buf) if hasattr(dist, "all_gather_into_tensor") and on_gpu else \
        _all_gather_list(dist, out, buf, world)
    out = out.cpu().numpy().reshape(world, -1)
    return np.concatenate([out[r, : counts[r]] for r in range(world)]).astype(np.uint32)


def _all_gather_list(dist, out, buf, world):
    import torch
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    out.copy_(torch.cat(parts))


def file_offsets(all_sizes):
    """Exclusive scan of the gathered sizes: byte offset of every slice's file in the archive."""
    offs = np.zeros(len(all_sizes) + 1, dtype=np.uint64)
    np.cumsum(np.asarray(all_sizes, dtype=np.uint64), out=offs[1:])
    return offs
