#!/usr/bin/env python3
"""Generate tests/golden/packbits.json by running the REFERENCE's PackBits (this container only).

Run:  python3 -B oracle/gen_packbits_golden.py
/root/reference/src/codec/packbits.py is imported as it is; only DATA is written: a few hundred byte strings
of several families (runs around the 127 / 128 chunk limits, alternations, ramps, random bytes of small and
large alphabets, the strings of the reference's own __main__ demo and of SURVEY Appendix C) with the bytes the
reference's encoder gives for them in both delta modes, and that its decoder gives them back.  No reference
source text is copied; the GPU box never sees /root/reference, only this fixture.
"""
import base64
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "src"))
from codec.packbits import PackBits  # noqa: E402  (the reference)


def strings():
    rng = np.random.default_rng(2023)
    out = [[3, 255, 3, 255, 3, 255, 3, 255, 20, 255], [1, 2, 3, 3, 3, 3, 4, 2, 1, 0], [7] * 300 + [1, 2], [9], [1, 1], [1, 2]]
    for n in (2, 3, 126, 127, 128, 129, 130, 254, 255, 256, 257, 381, 382, 383, 1000):
        out.append([5] * n)                                   # one run across the chunk limit
        out.append([i % 2 for i in range(n)])                 # one literal stretch across it
        out.append([5] * n + [6])
        out.append([6] + [5] * n)
        out.append([i % 251 for i in range(n)])               # a ramp: constant deltas -> one run under the delta transform
    for _ in range(120):                                      # runs and literals of random lengths, small alphabet
        parts = []
        for _ in range(int(rng.integers(1, 12))):
            if rng.random() < 0.5:
                parts += [int(rng.integers(0, 4))] * int(rng.integers(1, 300))
            else:
                parts += [int(x) for x in rng.integers(0, 4, size=int(rng.integers(1, 200)))]
        out.append(parts)
    for _ in range(60):                                       # random bytes
        out.append([int(x) for x in rng.integers(0, 256, size=int(rng.integers(2, 700)))])
    for _ in range(40):                                       # slowly varying bytes (what the delta transform is for)
        n = int(rng.integers(2, 600))
        out.append([int(x) % 256 for x in np.cumsum(rng.integers(-1, 2, size=n)) + 100])
    return out


def main():
    cases = []
    for data in strings():
        row = {"data": base64.b64encode(bytes(data)).decode()}
        for delta in (False, True):
            enc = bytes(PackBits(delta).encode(bytearray(data)))
            dec = bytes(bytearray(PackBits(delta).decode(bytearray(enc))))
            assert dec == bytes(data), (data[:8], delta)
            row["delta" if delta else "plain"] = base64.b64encode(enc).decode()
        cases.append(row)
    path = os.path.join(ROOT, "tests", "golden", "packbits.json")
    with open(path, "w") as f:
        json.dump({"source": "reference src/codec/packbits.py, PackBits(apply_delta_transform).encode / .decode, imported in the build container",
                   "cases": cases}, f, separators=(",", ":"))
    print(len(cases), "strings ->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
