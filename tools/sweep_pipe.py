#!/usr/bin/env python3
"""Tuning harness: per-kernel event times of the staged pipeline (option pipe_timing) for several settings of pipe_tpw;
with CCT_PIPE_STAMPS=1 in the environment the library also prints cycles per phase of the analyse and pack kernels."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "2023-compact-image-compression_amd"), os.path.join(ROOT, "tests")]


def main():
    tpws = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
    real = "--real" in sys.argv
    n, w = 256, 512
    if real:
        import golden_inputs as gi
        a, b = gi.load_slice("slice0671"), gi.load_slice("slice3706")
        b0 = np.stack([(a if i % 2 == 0 else b) for i in range(n)])
        batches = [b0, np.ascontiguousarray(b0[:, :, ::-1]), np.ascontiguousarray(b0[:, ::-1, :])]
    else:
        from bench import make_batches
        batches = make_batches(0, n)
    import cct_hip
    from cct_hip import _ffi, DeviceBuffer, codec_params, encode_payload_dev
    from cct_hip.batch import payload_stride
    L = _ffi.lib()
    d_imgs = [DeviceBuffer.from_numpy(b) for b in batches]
    stride = payload_stride(w, w, 16)
    d_pay, d_sz, d_st = DeviceBuffer(n * stride), DeviceBuffer(4 * n), DeviceBuffer(4 * n)
    params = codec_params(cct_hip.default_config(), np.uint16)
    print("device:", cct_hip.device_info(), flush=True)
    L.cct_set_option(b"pipe_timing", 1)
    reps = 2 if os.environ.get("CCT_PIPE_STAMPS") else 12
    tpw3s = [int(x) for x in os.environ.get("TPW3", "0").split(",")]
    L.cct_set_option(b"debug_skip", int(os.environ.get("DBG", "0")))
    ways_list = [0]
    for tpw in [t for t in tpws for _ in ways_list]:
        L.cct_set_option(b"pipe_tpw", tpw)
        rows = []
        for it in range(reps + 2):
            encode_payload_dev(d_imgs[it % 3], n, w, w, params, d_pay, d_sz, d_st)
            v = [C.c_int(0) for _ in range(4)]
            for i in range(4):
                L.cct_get_option(f"pipe_us_k{i + 1}".encode(), C.byref(v[i]))
            rows.append([x.value / 10.0 for x in v])
        med = np.median(np.array(rows[2:]), axis=0)
        print(f"tpw {tpw}: analyse {med[0]:7.1f}  masks {med[1]:7.1f}  resolve {med[2]:7.1f}  pack {med[3]:7.1f}  sum {med.sum():7.1f} us", flush=True)
    L.cct_set_option(b"debug_skip", 0)
    L.cct_set_option(b"pipe_timing", 0)
    L.cct_set_option(b"pipe_tpw", 0)


if __name__ == "__main__":
    main()
