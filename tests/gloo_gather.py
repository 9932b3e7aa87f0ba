"""torch.distributed twin of cct_hip.parallel.gather_sizes_rccl, for the world-size-2 gloo test on CPU only (the product
gathers through the library's C ABI and has no PyTorch in it)."""
import numpy as np


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


