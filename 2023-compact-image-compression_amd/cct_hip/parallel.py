"""Multi-GPU sharding of a slice corpus: one process per GPU, contiguous slice ranges, and the
path's only exchange step -- an all-gather of per-slice compressed sizes (RCCL over xGMI when
the process group is "nccl"; gloo on CPU for tests).  No pixel or payload byte crosses GPUs:
slices are independent units (the reference treats them so too, scripts/evaluate.py:107-119).
"""
import ctypes as C
import os
import struct
import time

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


# ---- RCCL through the C ABI (no PyTorch): cct_comm_* / cct_allgather_u32 ---------------------------------------

def exchange_unique_id(rank, world, make_id, directory=None, key=None, timeout_s=120.0):
    """Rank 0 creates the 128-byte communicator id (make_id()) and leaves it in a file the other ranks wait for.  The file
    name is derived from what the launcher gives every rank alike (MASTER_PORT, the launcher's pid, TORCHELASTIC_RUN_ID),
    so that two launches on one host do not see each other's id."""
    directory = directory or os.environ.get("CCT_RENDEZVOUS_DIR", "/tmp")
    if key is None:
        key = "_".join(str(x) for x in (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                                        os.getppid()))
    path = os.path.join(directory, f"cct_rccl_id_{key}")
    if rank == 0:
        blob = bytes(make_id())
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            f.write(blob)
        os.replace(tmp, path)  # atomic: readers see nothing or all of it
        return blob, path
    t0 = time.time()
    while True:
        try:
            with open(path, "rb") as f:
                blob = f.read()
            if len(blob) == 128:
                return blob, path
        except OSError:
            pass
        if time.time() - t0 > timeout_s:
            raise TimeoutError(f"rank {rank}: no communicator id at {path} after {timeout_s:.0f} s")
        time.sleep(0.02)


def comm_init(rank, world):
    """One RCCL communicator over the library's device; the id travels through exchange_unique_id."""
    from . import _ffi
    L = _ffi.lib()

    def make_id():
        buf = (C.c_char * 128)()
        _ffi.check(L.cct_comm_unique_id(buf))
        return bytes(buf)

    blob, path = exchange_unique_id(rank, world, make_id)
    _ffi.check(L.cct_comm_init(blob, rank, world))
    allgather_u32(np.zeros(1, np.uint32), 1)  # everyone has read the id once this returns
    if rank == 0:
        try:
            os.remove(path)
        except OSError:
            pass


def allgather_u32(values, max_local=None):
    """(world, max_local) uint32 array: row r holds rank r's values, zero padded."""
    from . import _ffi
    L = _ffi.lib()
    rk, wd = C.c_int(0), C.c_int(0)
    _ffi.check(L.cct_comm_info(C.byref(rk), C.byref(wd)))
    world = max(1, wd.value)
    values = np.ascontiguousarray(values, dtype=np.uint32)
    max_local = int(max_local if max_local is not None else values.size)
    out = np.zeros((world, max(max_local, 1)), dtype=np.uint32)
    _ffi.check(L.cct_allgather_u32(values.ctypes.data, int(values.size), max(max_local, 1), out.ctypes.data))
    return out[:, :max_local] if max_local else out[:, :0]


def gather_sizes_rccl(local_sizes):
    """All ranks' per-slice compressed sizes in global slice order through RCCL (counts first, then the padded sizes)."""
    local_sizes = np.ascontiguousarray(local_sizes, dtype=np.uint32)
    counts = allgather_u32(np.array([local_sizes.size], np.uint32), 1)[:, 0]
    rows = allgather_u32(local_sizes, int(counts.max()) if counts.size else 0)
    return np.concatenate([rows[r, : counts[r]] for r in range(len(counts))]).astype(np.uint32)


def allreduce_max_float(x):
    """max over ranks of one float (bench.py: the slowest rank's time) by gathering the bit patterns."""
    bits = struct.unpack("<I", struct.pack("<f", float(x)))[0]
    rows = allgather_u32(np.array([bits], np.uint32), 1)[:, 0]
    return max(struct.unpack("<f", struct.pack("<I", int(b)))[0] for b in rows)


def barrier():
    allgather_u32(np.zeros(1, np.uint32), 1)
