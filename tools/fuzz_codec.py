#!/usr/bin/env python3
"""Fuzz run (not part of the product): random byte strings with many different statistics through the device
DEFLATE (must equal zlib.compress(x, 9) byte for byte) and through the device INFLATE (streams written by zlib
with random level / strategy / window / flush points must inflate to the input).
Usage: python tools/fuzz_codec.py [rounds] [seed]"""
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "2023-compact-image-compression_amd")]
import cct_hip  # noqa: E402


def blob(rng):
    n = int(rng.choice([0, 1, 2, 3, 5, 17, 258, 259, 300, 4096, 32768, 65535, 65536, 70000, 131072, 200000, 300000]))
    n = max(0, n + int(rng.integers(-3, 4))) if n > 8 else n
    kind = int(rng.integers(0, 9))
    if kind == 0:
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == 1:
        return rng.integers(0, int(rng.integers(1, 6)), n, dtype=np.uint8).tobytes()
    if kind == 2:  # runs of random length and byte
        parts, tot = [], 0
        while tot < n:
            k = int(rng.choice([1, 2, 3, 4, 7, 30, 257, 258, 259, 1000, 40000]))
            parts.append(bytes([int(rng.integers(0, 4))]) * k)
            tot += k
        return b"".join(parts)[:n]
    if kind == 3:  # periodic
        per = bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))
        return (per * (n // max(1, len(per)) + 1))[:n]
    if kind == 4:  # token-stream-like: mostly small bytes, zero runs
        a = np.abs(rng.normal(0, 12, n)).astype(np.uint8)
        for _ in range(int(rng.integers(0, 30))):
            s = int(rng.integers(0, max(1, n)))
            a[s:s + int(rng.integers(1, 5000))] = 0
        return a.tobytes()
    if kind == 5:  # far repeats (window edge)
        base = rng.integers(0, 256, int(rng.integers(100, 3000)), dtype=np.uint8).tobytes()
        gap = rng.integers(0, 256, int(rng.choice([32000, 32506 - 100, 32768, 33000])), dtype=np.uint8).tobytes()
        return ((base + gap) * (n // (len(base) + len(gap)) + 1))[:n]
    if kind == 6:  # skewed alphabet (long Huffman codes)
        p = 0.5 ** np.arange(1, 41)
        p = p / p.sum()
        return rng.choice(40, n, p=p).astype(np.uint8).tobytes()
    if kind == 7:  # text-like
        words = [bytes(rng.integers(97, 123, int(rng.integers(1, 10)), dtype=np.uint8)) for _ in range(200)]
        out, tot = [], 0
        while tot < n:
            w = words[int(rng.integers(0, 200))] + b" "
            out.append(w)
            tot += len(w)
        return b"".join(out)[:n]
    return bytes(n)


def stream(rng, b):
    mode = int(rng.integers(0, 5))
    if mode == 0:
        return zlib.compress(b, int(rng.integers(0, 10)))
    strat = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED][int(rng.integers(0, 5))]
    c = zlib.compressobj(int(rng.integers(1, 10)), zlib.DEFLATED, int(rng.integers(9, 16)), int(rng.integers(1, 10)), strat)
    if mode == 1 or len(b) < 10:
        return c.compress(b) + c.flush()
    cuts = sorted(int(x) for x in rng.integers(0, len(b), int(rng.integers(1, 6))))
    parts, prev = [], 0
    for cut in cuts:
        parts.append(c.compress(b[prev:cut]) + c.flush([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH][int(rng.integers(0, 2))]))
        prev = cut
    return b"".join(parts) + c.compress(b[prev:]) + c.flush()


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0, nbad = time.time(), 0
    for r in range(rounds):
        blobs = [blob(rng) for _ in range(int(rng.integers(1, 48)))]
        got = cct_hip.zlib_compress_batch(blobs)
        for i, (b, g) in enumerate(zip(blobs, got)):
            if g != zlib.compress(b, 9):
                nbad += 1
                open(f"/tmp/fuzz_deflate_{seed}_{r}_{i}.bin", "wb").write(b)
                print(f"DEFLATE MISMATCH round {r} blob {i} len {len(b)}", flush=True)
        streams = [stream(rng, b) for b in blobs]
        outs = cct_hip.zlib_decompress_batch(streams, max_out=max(16, max(len(b) for b in blobs)))
        for i, (b, o) in enumerate(zip(blobs, outs)):
            if o != b:
                nbad += 1
                open(f"/tmp/fuzz_inflate_{seed}_{r}_{i}.bin", "wb").write(streams[i])
                print(f"INFLATE MISMATCH round {r} stream {i} len {len(b)}", flush=True)
        if r % 5 == 4:
            print(f"round {r + 1}/{rounds}  {time.time() - t0:.0f} s  mismatches: {nbad}", flush=True)
    print("fuzz clean" if nbad == 0 else f"{nbad} MISMATCHES")
    sys.exit(1 if nbad else 0)


if __name__ == "__main__":
    main()
