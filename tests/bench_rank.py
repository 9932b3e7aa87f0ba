"""One rank of bench.py's N > 1 control flow on a box with ONE GPU (helper of test_gpu_parity.py, not a test itself).

RCCL refuses two ranks on one device, so the collective -- and only the collective -- is replaced here by an exchange through
files in CCT_TEST_EXCHANGE_DIR; rank-disjoint batches, the gather after every encode, the barriers around the timed region,
the max over ranks of the elapsed time and rank 0's single JSON line are bench.py's own code, unchanged.  The real RCCL
entry points are exercised with a world of one in test_rccl_communicator_of_one.

    RANK=r WORLD_SIZE=n LOCAL_RANK=0 LOCAL_WORLD_SIZE=n CCT_TEST_EXCHANGE_DIR=... python tests/bench_rank.py --gpus n ...
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402  (puts the package directory on sys.path)
from cct_hip import parallel  # noqa: E402

DIR = os.environ["CCT_TEST_EXCHANGE_DIR"]
RANK, WORLD = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
_calls = [0]


def _allgather_u32(values, max_local=None):
    values = np.ascontiguousarray(values, dtype=np.uint32)
    max_local = int(max_local if max_local is not None else values.size)
    width = max(max_local, 1)
    k = _calls[0]
    _calls[0] += 1
    row = np.zeros(width, dtype=np.uint32)
    row[: values.size] = values
    mine = os.path.join(DIR, f"c{k}_r{RANK}")
    row.tofile(mine + ".tmp")
    os.replace(mine + ".tmp", mine)
    out = np.zeros((WORLD, width), dtype=np.uint32)
    t0 = time.time()
    for r in range(WORLD):
        path = os.path.join(DIR, f"c{k}_r{r}")
        while not os.path.exists(path):
            if time.time() - t0 > 240:
                raise TimeoutError(f"rank {RANK}: collective {k} never heard from rank {r}")
            time.sleep(0.001)
        got = np.fromfile(path, dtype=np.uint32)
        assert got.size == width, (k, r, got.size, width)  # every rank must make the same call in the same order
        out[r] = got
    return out[:, :max_local]


parallel.allgather_u32 = _allgather_u32
parallel.comm_init = lambda rank, world: None

if __name__ == "__main__":
    bench.main()
