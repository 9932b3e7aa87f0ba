#!/usr/bin/env python3
"""Soak run (not part of the product): the bench pipeline (two encode threads, one decode thread, three buffer
sets) for many steps, checking EVERY step: the archive must hash to what the same input produced the first time
(the encoder is deterministic) and the decoded rasters must equal the input.  Usage: python tools/soak.py [steps]"""
import ctypes as C
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "2023-compact-image-compression_amd")]
import cct_hip  # noqa: E402
from cct_hip import _ffi  # noqa: E402
from bench import make_batches  # noqa: E402
W = H = 512
import xxhash  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    n = 256
    batches = make_batches(0, n)  # forks a process pool: before the first GPU call
    L = _ffi.lib()
    _ffi.check(L.cct_init(0))
    cfg = cct_hip.default_config()
    flags, bs, eof, magic, ch, bpc = cct_hip.codec_params(cfg, np.uint16)
    d_imgs = [cct_hip.DeviceBuffer.from_numpy(b) for b in batches]
    NSET = 3
    cap = n * L.cct_file_bound(W, H, bs)
    pins = [cct_hip.PinnedArray(cap) for _ in range(NSET)]
    d_back = [cct_hip.DeviceBuffer(batches[0].nbytes) for _ in range(NSET)]
    offs = [np.zeros(n + 1, dtype=np.uint64) for _ in range(NSET)]
    sizes = [np.zeros(n, dtype=np.uint32) for _ in range(NSET)]
    status = [np.zeros(n, dtype=np.uint32) for _ in range(NSET)]
    want_hash = {}
    bad = []
    pool_enc, pool_dec = ThreadPoolExecutor(2), ThreadPoolExecutor(1)

    def enc(i, k):
        _ffi.check(L.cct_encode_batch_packed(d_imgs[i % 3].ptr, 1, n, W, H, bs, flags, eof, magic, ch, bpc, pins[k].array.ctypes.data,
                                             cap, offs[k].ctypes.data, sizes[k].ctypes.data, status[k].ctypes.data, None, None))

    def dec(i, k, e):
        e.result()
        hsh = xxhash.xxh64(pins[k].array[: int(offs[k][n])]).hexdigest()
        if want_hash.setdefault(i % 3, hsh) != hsh:
            bad.append((i, "archive differs from the first encode of this batch"))
        st = np.zeros(n, dtype=np.uint32)
        _ffi.check(L.cct_decode_batch(pins[k].array.ctypes.data, offs[k].ctypes.data, n, bs, magic, d_back[k].ptr, 1, n * W * H, st.ctypes.data))
        back = d_back[k].download(np.uint16, n * W * H).reshape(batches[0].shape)
        if not np.array_equal(back, batches[i % 3]):
            bad.append((i, "decoded rasters differ from the input"))

    t0 = time.time()
    in_flight, prev = [], None
    for i in range(steps):
        k = i % NSET
        while len(in_flight) >= NSET:
            in_flight.pop(0).result()
        e = pool_enc.submit(enc, i, k)
        in_flight.append(pool_dec.submit(dec, i, k, e))
        if prev is not None:
            prev.result()
        prev = e
        if i % 20 == 19:
            print(f"step {i + 1}/{steps}  {time.time() - t0:.1f} s  failures: {len(bad)}", flush=True)
    for f in in_flight:
        f.result()
    print("FAILURES:" if bad else "all steps verified", bad[:5])
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
