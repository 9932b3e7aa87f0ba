#!/usr/bin/env python3
"""Tuning harness (not part of the product): times the encode kernel alone with phase-ablation
masks (option debug_skip) on the bench workload.  Usage: python tools_tune.py [masks...]"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "2023-compact-image-compression_amd")]
import cct_hip
from cct_hip import _ffi, DeviceBuffer, Event, codec_params, encode_payload_dev
from cct_hip.batch import payload_stride
from bench import make_batches

def main():
    masks = [int(x, 0) for x in sys.argv[1:]] or [0]
    L = _ffi.lib()
    n, w, h = 256, 512, 512
    batches = make_batches(0, n)
    d_imgs = [DeviceBuffer.from_numpy(b) for b in batches]
    stride = payload_stride(w, h, 16)
    d_pay, d_sz, d_st = DeviceBuffer(n * stride), DeviceBuffer(4 * n), DeviceBuffer(4 * n)
    params = codec_params(cct_hip.default_config(), np.uint16)
    e0, e1 = Event(), Event()
    for tile in (1,):
        L.cct_set_option(b"tile_path", tile)
        for mk in masks:
            L.cct_set_option(b"debug_skip", mk)
            ts = []
            for it in range(8):
                e0.record()
                encode_payload_dev(d_imgs[it % 3], n, w, h, params, d_pay, d_sz, d_st)
                e1.record()
                ts.append(e1.elapsed_ms_since(e0) * 1e3)
            print(f"tile={tile} skip=0x{mk:02x}  us: min {min(ts[2:]):8.1f}  med {sorted(ts[2:])[3]:8.1f}", flush=True)
    L.cct_set_option(b"debug_skip", 0)

if __name__ == "__main__":
    main()
