"""Multi-GPU sharding of a slice corpus: one process per GPU, contiguous slice ranges, and the
path's only exchange step -- an all-gather of per-slice compressed sizes over RCCL / xGMI, reached through the
library's C ABI (cct_comm_*; no PyTorch here -- the torch.distributed twin used by the gloo test lives in
tests/gloo_gather.py).  No pixel or payload byte crosses GPUs: slices are independent units (the reference treats
them so too, scripts/evaluate.py:107-119).
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


def file_offsets(all_sizes):
    """Exclusive scan of the gathered sizes: byte offset of every slice's file in the archive."""
    offs = np.zeros(len(all_sizes) + 1, dtype=np.uint64)
    np.cumsum(np.asarray(all_sizes, dtype=np.uint64), out=offs[1:])
    return offs


# ---- RCCL through the C ABI (no PyTorch): cct_comm_* / cct_allgather_u32 ---------------------------------------

def rendezvous_key():
    """What every rank of one launch knows alike and another launch on the host does not share: the rendezvous address
    and port the launcher exports, its run id and restart count.  Only a launch that sets none of them falls back to the
    parent's pid (ranks started by hand from one shell)."""
    env = os.environ
    if env.get("MASTER_PORT") or env.get("TORCHELASTIC_RUN_ID"):
        parts = (env.get("MASTER_ADDR", "localhost"), env.get("MASTER_PORT", "0"), env.get("TORCHELASTIC_RUN_ID", "none"),
                 env.get("TORCHELASTIC_RESTART_COUNT", "0"))
    else:
        parts = ("ppid", os.getppid())
    return "_".join(str(x).replace("/", "-").replace(":", "-") for x in parts)


def _process_start_time():
    try:
        import psutil
        return psutil.Process().create_time()
    except Exception:  # noqa: BLE001 -- no psutil: the age check is skipped
        return None


def exchange_unique_id(rank, world, make_id, directory=None, key=None, timeout_s=120.0, max_age_s=5.0):
    """Rank 0 creates the 128-byte communicator id (make_id()) and leaves it in a file the other ranks wait for
    (rendezvous_key() names it).  A file left behind by a launch that died is not taken for this launch's id: rank 0 removes
    whatever carries the name before it does anything slow, and the other ranks ignore a file older than their own process
    (minus max_age_s: the ranks of a launch start together, rank 0 writes seconds later)."""
    directory = directory or os.environ.get("CCT_RENDEZVOUS_DIR", "/tmp")
    path = os.path.join(directory, f"cct_rccl_id_{key if key is not None else rendezvous_key()}")
    if rank == 0:
        try:
            os.remove(path)
        except OSError:
            pass
        blob = bytes(make_id())
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            f.write(blob)
        os.replace(tmp, path)  # atomic: readers see nothing or all of it
        return blob, path
    born = _process_start_time()
    t0 = time.time()
    while True:
        try:
            st = os.stat(path)
            if born is None or st.st_mtime >= born - max_age_s:
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

    def forget():
        try:
            os.remove(path)
        except OSError:
            pass

    if rank == 0:
        import atexit
        atexit.register(forget)  # also when this rank dies before the first gather
    _ffi.check(L.cct_comm_init(blob, rank, world))
    allgather_u32(np.zeros(1, np.uint32), 1)  # everyone has read the id once this returns
    if rank == 0:
        forget()


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
