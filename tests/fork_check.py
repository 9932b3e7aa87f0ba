"""Helper of test_gpu_parity.test_fork_before_first_use_and_fork_after_init (run as a script in a fresh process).

scripts/evaluate.py:107 of the reference fans slices out over a fork()ed multiprocessing pool after importing codec.core.
The library binds the GPU on the first device call, so workers forked BEFORE that call each get their own context;
a child forked AFTER the parent has initialised the GPU must be refused with CCT_E_DEVICE (ROCm cannot share a runtime
across fork), not crash and not re-create the context."""
import hashlib
import json
import multiprocessing as mp
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "2023-compact-image-compression_amd"))

import golden_inputs as gi  # noqa: E402
from codec.core import Encoder, Decoder  # noqa: E402  (imports the package: loads the library, touches no device)
import cct_hip  # noqa: E402


def work(seed):
    cfg = cct_hip.default_config()
    cfg["verbose"] = False
    img = gi.ct_phantom(seed, 128)
    out = Encoder(cfg, img).encode()
    back = Decoder(cfg, out).decode()
    assert bytes(back) == img.tobytes()
    return os.getpid(), hashlib.sha1(out).hexdigest()


def child_after_init(q):
    from cct_hip import _ffi
    try:
        work(1)
        q.put(("encoded", 0))
    except Exception as e:  # noqa: BLE001
        q.put((type(e).__name__, getattr(e, "code", None), str(e)))


def main():
    ctx = mp.get_context("fork")
    with ctx.Pool(2) as pool:  # the evaluate.py pattern: fork first, the children make the first device calls
        res = pool.map(work, [1, 2, 3, 4])
    pids = {p for p, _ in res}
    mine = [work(s)[1] for s in (1, 2, 3, 4)]  # now the parent initialises its own context
    q = ctx.Queue()
    p = ctx.Process(target=child_after_init, args=(q,))
    p.start()
    late = q.get(timeout=60)
    p.join(60)
    print(json.dumps({"pool_pids": len(pids), "parent_pid_in_pool": os.getpid() in pids,
                      "pool_hashes": [h for _, h in res], "parent_hashes": mine, "late_child": list(late),
                      "late_exit": p.exitcode}))


if __name__ == "__main__":
    main()
