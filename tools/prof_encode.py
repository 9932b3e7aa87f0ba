#!/usr/bin/env python3
"""Tuning harness (not part of the product): stage (i) alone -- cct_encode_payload_dev on the bench workload (256
slices of 512x512, three rotating batches so that HBM and not the Infinity Cache is measured), timed with HIP events
on the library's stream.  Run it under `rocprofv3 --kernel-trace --stats` for per-kernel times.

    python tools/prof_encode.py [--paths 1,2] [--reps 30] [--size 512] [--slices 256] [--real]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "2023-compact-image-compression_amd"), os.path.join(ROOT, "tests")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--paths", default="1,2")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--slices", type=int, default=256)
    ap.add_argument("--tpg", default="", help="tiles per workgroup of the streaming kernel for every path-4 run (e.g. 1,2,4)")
    ap.add_argument("--real", action="store_true", help="use the two real CT slices (tests/golden) instead of phantoms")
    args = ap.parse_args()
    n, w = args.slices, args.size
    # inputs first: nothing below may fork once the device is initialised
    if args.real:
        import golden_inputs as gi
        a, b = gi.load_slice("slice0671"), gi.load_slice("slice3706")
        b0 = np.stack([(a if i % 2 == 0 else b) for i in range(n)])
        batches = [b0, np.ascontiguousarray(b0[:, :, ::-1]), np.ascontiguousarray(b0[:, ::-1, :])]
    elif w == 512:
        from bench import make_batches
        batches = make_batches(0, n)
    else:
        from cct_hip.synth import ct_phantom
        base = [ct_phantom(s, w) for s in range(8)]
        b0 = np.stack([base[i % 8] for i in range(n)])
        batches = [b0, np.ascontiguousarray(b0[:, :, ::-1]), np.ascontiguousarray(b0[:, ::-1, :])]
    import cct_hip
    from cct_hip import _ffi, DeviceBuffer, Event, codec_params, encode_payload_dev
    from cct_hip.batch import payload_stride
    L = _ffi.lib()
    d_imgs = [DeviceBuffer.from_numpy(b) for b in batches]
    stride = payload_stride(w, w, 16)
    d_pay, d_sz, d_st = DeviceBuffer(n * stride), DeviceBuffer(4 * n), DeviceBuffer(4 * n)
    params = codec_params(cct_hip.default_config(), np.uint16)
    e0, e1 = Event(), Event()
    ref = None
    runs = []
    for path in [int(x) for x in args.paths.split(",")]:
        if path == 4 and args.tpg:
            runs += [(4, int(t)) for t in args.tpg.split(",")]
        else:
            runs.append((path, 0))
    for path, tpg in runs:
        L.cct_set_option(b"tile_path", path)
        if tpg:
            L.cct_set_option(b"stream_tpg", tpg)
        ts = []
        for it in range(args.reps + 3):
            e0.record()
            encode_payload_dev(d_imgs[it % 3], n, w, w, params, d_pay, d_sz, d_st)
            e1.record()
            ts.append(e1.elapsed_ms_since(e0) * 1e3)
        ts = sorted(ts[3:])
        sizes = d_sz.download(np.uint32, n)
        tot = int(sizes.sum())
        if ref is None:
            ref = tot
        px = n * w * w
        med = ts[len(ts) // 2]
        print(f"path {path}{'/' + str(tpg) if tpg else ''}: us min {ts[0]:8.1f} med {med:8.1f} max {ts[-1]:8.1f}   read {2 * px / med / 1e3:7.1f} GB/s"
              f" = {2 * px / med / 1e3 / 8000:.3f} of 8 TB/s   payload {tot} B {'(same)' if tot == ref else '(DIFFERS)'}", flush=True)
    L.cct_set_option(b"tile_path", 1)


if __name__ == "__main__":
    main()
