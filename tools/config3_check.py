#!/usr/bin/env python3
"""One-off check of BASELINE configs[2] at full size on one GPU (not part of the product): 3954 synthetic 512x512
slices, encode in bounded passes + decode, exact round trip, a sample compared with the oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "2023-compact-image-compression_amd")]
import cct_hip
from cct_hip.synth import ct_phantom
from oracle import oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3954
base = [ct_phantom(200 + i, 512) for i in range(64)]
imgs = np.stack([np.roll(base[i % 64], i // 64, axis=0) for i in range(n)])
cfg = cct_hip.default_config()
t0 = time.time(); files = cct_hip.encode_batch(imgs, cfg); t1 = time.time()
print(f"encode {n} x 512^2: {t1 - t0:.2f} s, {sum(map(len, files)) / 1e6:.1f} MB", flush=True)
back = np.asarray(cct_hip.decode_batch(files, cfg)).reshape(imgs.shape); t2 = time.time()
print(f"decode: {t2 - t1:.2f} s; round trip exact: {np.array_equal(back, imgs)}", flush=True)
for i in (0, 503, 504, 505, 2000, n - 1):
    assert oracle.encode(imgs[i]) == files[i], i
print("oracle sample identical")
