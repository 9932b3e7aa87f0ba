#!/usr/bin/env python3
"""Tuning harness (not part of the product): whole encode_batch / decode_batch calls on the bench workload, serial
(no overlap), with the library's own stage timings.  CCT_INF_PROF=1 prints the INFLATE kernel's per-phase cycle counts.

    python tools/prof_codec.py [--reps 10] [--slices 256] [--real] [--what enc,dec]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "2023-compact-image-compression_amd"), os.path.join(ROOT, "tests")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--slices", type=int, default=256)
    ap.add_argument("--real", action="store_true")
    ap.add_argument("--what", default="enc,dec")
    args = ap.parse_args()
    n = args.slices
    if args.real:
        import golden_inputs as gi
        a, b = gi.load_slice("slice0671"), gi.load_slice("slice3706")
        batch = np.stack([(a if i % 2 == 0 else b) for i in range(n)])
    else:
        from bench import make_batches
        batch = make_batches(0, n)[0]
    import cct_hip
    from cct_hip import _ffi
    L = _ffi.lib()
    cfg = cct_hip.default_config()
    files = cct_hip.encode_batch(batch, cfg)
    print(f"n={n} payload->file bytes {sum(len(f) for f in files)}", file=sys.stderr)
    what = args.what.split(",")
    import ctypes as C

    def tms():
        tm = (C.c_float * 6)()
        L.cct_last_timings(tm)
        return "kernel %.3f d2h %.3f deflate %.3f inflate %.3f dec_kernel %.3f" % tuple(tm[:5])
    for r in range(args.reps):
        if "enc" in what:
            t0 = time.perf_counter()
            f2 = cct_hip.encode_batch(batch, cfg)
            t1 = time.perf_counter()
            print(f"enc {1e3 * (t1 - t0):7.2f} ms {tms()}", file=sys.stderr)
            assert f2 == files
        if "dec" in what:
            t0 = time.perf_counter()
            back = cct_hip.decode_batch(files, cfg)
            t1 = time.perf_counter()
            print(f"dec {1e3 * (t1 - t0):7.2f} ms {tms()}", file=sys.stderr)
            assert np.array_equal(back, batch)


if __name__ == "__main__":
    main()
