#!/usr/bin/env python3
"""One-off check of BASELINE configs[3] at full size (not part of the product): 512 synthetic 1024x1024 slices,
encode (multi-pass DEFLATE workspaces) + decode, exact round trip, a sample compared with the oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "2023-compact-image-compression_amd")]
import cct_hip
from cct_hip.synth import ct_phantom
from oracle import oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
t0 = time.time()
base = [ct_phantom(100 + i, 1024) for i in range(8)]
imgs = np.stack([np.roll(base[i % 8], i, axis=1) for i in range(n)])
print(f"input {imgs.nbytes / 1e9:.2f} GB built in {time.time() - t0:.1f} s", flush=True)
cfg = cct_hip.default_config()
t0 = time.time(); files = cct_hip.encode_batch(imgs, cfg); t1 = time.time()
print(f"encode {n} x 1024^2: {t1 - t0:.2f} s, {sum(map(len, files)) / 1e6:.1f} MB", flush=True)
back = np.asarray(cct_hip.decode_batch(files, cfg)).reshape(imgs.shape); t2 = time.time()
print(f"decode: {t2 - t1:.2f} s; round trip exact: {np.array_equal(back, imgs)}", flush=True)
for i in (0, 7, n // 2, n - 1):
    assert oracle.encode(imgs[i]) == files[i], i
print("oracle sample identical")
