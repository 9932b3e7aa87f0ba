#!/usr/bin/env python3
"""Tuning / debugging: one fresh process = first use of the second encode and decode slots while the first ones are busy (what
test_concurrent_encodes_and_decodes_use_their_slots does).  Run it many times from a shell loop under `timeout`."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "2023-compact-image-compression_amd"), os.path.join(ROOT, "tests")]
import golden_inputs as gi  # noqa: E402

batches = [np.stack([gi.ct_phantom(10 * k + i, 256) for i in range(24)]) for k in range(4)]
import cct_hip as hip  # noqa: E402

cfg = hip.default_config()
serial = [hip.encode_batch(b, cfg) for b in batches]
for k, b in enumerate(batches):
    assert np.array_equal(hip.decode_batch(serial[k], cfg), b)
with ThreadPoolExecutor(4) as pool:
    fe = [pool.submit(lambda k=k: [hip.encode_batch(batches[k], cfg) for _ in range(3)]) for k in (0, 1)]
    fd = [pool.submit(lambda k=k: [hip.decode_batch(serial[k], cfg) for _ in range(3)]) for k in (2, 3)]
    for k, f in zip((0, 1), fe):
        for files in f.result():
            assert files == serial[k]
    for k, f in zip((2, 3), fd):
        for back in f.result():
            assert np.array_equal(back, batches[k])
print("ok")
