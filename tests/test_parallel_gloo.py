"""N>1 path on CPU: world_size-2 gloo process group exercising shard_range / gather_sizes."""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "2023-compact-image-compression_amd")

WORKER = r"""
import os, sys, numpy as np
sys.path.insert(0, {pkg!r}); sys.path.insert(0, {here!r})
import torch.distributed as dist
from cct_hip.parallel import shard_range, file_offsets
from gloo_gather import gather_sizes
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n_total = 3954 if len(sys.argv) < 2 else int(sys.argv[1])
lo, hi = shard_range(n_total, rank, world)
sizes = (np.arange(lo, hi, dtype=np.uint32) * 7 + 150000) % 244138
allsz = gather_sizes(sizes, dist, 0)
expect = (np.arange(n_total, dtype=np.uint32) * 7 + 150000) % 244138
assert allsz.dtype == np.uint32 and np.array_equal(allsz, expect), (rank, allsz[:4], expect[:4])
offs = file_offsets(allsz)
assert offs[-1] == expect.astype(np.uint64).sum() and offs[lo] == expect[:lo].astype(np.uint64).sum()
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok", lo, hi)
"""


def _run(world, n_total):
    code = WORKER.format(pkg=PKG, here=HERE)
    procs = []
    port = 29500 + (os.getpid() % 2000)
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", code, str(n_total)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    return outs


def test_gather_sizes_world2_corpus_shards():
    outs = _run(2, 3954)  # BASELINE config 3: 3954 slices, 1977 per rank
    assert "ok 0 1977" in outs[0] and "ok 1977 3954" in outs[1]


def test_gather_sizes_world2_ragged():
    _run(2, 7)  # 4 + 3 slices: padded collective, trimmed result


def test_shard_range_covers_everything():
    from cct_hip.parallel import shard_range
    for n in (0, 1, 7, 256, 3954):
        for world in (1, 2, 4, 8):
            got = []
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                got.extend(range(lo, hi))
            assert got == list(range(n))
    assert shard_range(3954, 0, 8) == (0, 495) and shard_range(3954, 2, 8) == (990, 1484) and shard_range(3954, 7, 8) == (3460, 3954)
    assert sorted({hi - lo for lo, hi in (shard_range(3954, r, 8) for r in range(8))}) == [494, 495]


def _id_worker(rank, d, q):
    import os
    os.environ.pop("MASTER_PORT", None)
    from cct_hip.parallel import exchange_unique_id
    blob, _ = exchange_unique_id(rank, 2, lambda: bytes(range(128)), directory=d, key="unit", timeout_s=20)
    q.put((rank, blob))


def test_unique_id_file_rendezvous(tmp_path):
    """bench.py's N > 1 bootstrap without PyTorch: rank 0 leaves the 128-byte communicator id in a file, the others wait."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_id_worker, args=(r, str(tmp_path), q)) for r in (1, 0)]  # the reader starts first
    for p in ps:
        p.start()
    got = dict(q.get(timeout=60) for _ in ps)
    for p in ps:
        p.join(30)
    assert got[0] == got[1] == bytes(range(128))


def test_rendezvous_ignores_a_stale_id_and_keys_on_the_launch(tmp_path, monkeypatch):
    """A 128-byte file left by a launch that died must not be read as this launch's id: ranks other than 0 skip files older
    than their own process; rank 0 replaces the file.  The file name comes from what the launcher exports to every rank
    (address, port, run id), the parent pid only when none of that exists."""
    import time
    from cct_hip import parallel
    stale = tmp_path / "cct_rccl_id_unit2"
    stale.write_bytes(b"\xee" * 128)
    old = time.time() - 3600
    os.utime(stale, (old, old))
    with __import__("pytest").raises(TimeoutError):
        parallel.exchange_unique_id(1, 2, None, directory=str(tmp_path), key="unit2", timeout_s=0.3)
    blob, path = parallel.exchange_unique_id(0, 2, lambda: bytes(range(128)), directory=str(tmp_path), key="unit2")
    assert blob == bytes(range(128)) and open(path, "rb").read() == blob
    assert parallel.exchange_unique_id(1, 2, None, directory=str(tmp_path), key="unit2", timeout_s=5)[0] == blob
    for k in ("MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT"):
        monkeypatch.delenv(k, raising=False)
    assert parallel.rendezvous_key() == f"ppid_{os.getppid()}"
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1"); monkeypatch.setenv("MASTER_PORT", "29511")
    k1 = parallel.rendezvous_key()
    assert "29511" in k1 and str(os.getppid()) not in k1.split("_")
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job/7")
    assert parallel.rendezvous_key() != k1 and "/" not in parallel.rendezvous_key()


def test_allgather_without_communicator_is_a_copy():
    """cct_allgather_u32 / gather_sizes_rccl in a single process (no communicator): identity, no device needed."""
    import numpy as np
    from cct_hip import parallel
    v = np.arange(7, dtype=np.uint32) * 1000
    assert np.array_equal(parallel.allgather_u32(v, 9)[0], np.concatenate([v, [0, 0]]))
    assert np.array_equal(parallel.gather_sizes_rccl(v), v)
    assert abs(parallel.allreduce_max_float(1.25) - 1.25) < 1e-6
