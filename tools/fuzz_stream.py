#!/usr/bin/env python3
"""Fuzz run on the GPU box (test infrastructure): the streaming kernel (encode_stream.hip) with 1, 2 and 4 tiles per
workgroup against the generic table-gather kernel AND the CPU oracle -- payload bytes, sizes, status, statistics and
block roles -- on random images of every tiled square, including pixel values that force the exact (unpacked)
arithmetic, int16 input (segmentation sees signed values), dense noise (every block difficult: spilled lists, pairs
beyond one per lane, one island through the whole slice) and segmentation off.
Usage: python tools/fuzz_stream.py [rounds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "2023-compact-image-compression_amd"), os.path.join(ROOT, "tests")]
import cct_hip  # noqa: E402
from cct_hip import DeviceBuffer, _ffi, codec_params, encode_payload_dev  # noqa: E402
from cct_hip.batch import payload_stride  # noqa: E402
from cct_hip.synth import ct_phantom  # noqa: E402
from oracle import oracle  # noqa: E402


def image(rng, n, signed):
    kind = int(rng.integers(0, 9))
    hi = 2048
    if kind == 0:
        a = rng.integers(0, hi, (n, n))                              # dense noise
    elif kind == 1:
        a = ct_phantom(int(rng.integers(0, 10 ** 6)), n, bool(rng.integers(0, 2))).astype(np.int64)
    elif kind == 2:                                                  # noise patches on a smooth ground: islands across tile borders
        x, y = np.meshgrid(np.arange(n), np.arange(n))
        a = 900 + 300 * np.sin(x / float(rng.integers(5, 60))) * np.cos(y / float(rng.integers(5, 60)))
        for _ in range(int(rng.integers(1, 12))):
            r0, c0 = rng.integers(0, n, 2)
            h, w = rng.integers(4, max(5, n // 2), 2)
            a[r0:r0 + h, c0:c0 + w] = rng.integers(0, hi, a[r0:r0 + h, c0:c0 + w].shape)
    elif kind == 3:                                                  # pixels >= 0x4000: the exact arithmetic of the whole group
        a = rng.integers(0, 65536, (n, n)) if rng.random() < 0.3 else np.clip(ct_phantom(int(rng.integers(0, 10 ** 6)), n).astype(np.int64) * 20, 0, 65535)
    elif kind == 4:
        a = np.zeros((n, n))
        a[:64, :64] = rng.integers(0, hi, (64, 64))                  # difficult block 0 (Q4)
    elif kind == 5:                                                  # sparse spikes: many short islands
        a = np.full((n, n), int(rng.integers(0, 1500)))
        m = rng.random((n, n)) < float(rng.uniform(0.002, 0.2))
        a[m] = rng.integers(0, hi, int(m.sum()))
    elif kind == 6:                                                  # alternating texture rows of blocks: dense meshes
        a = np.where((np.arange(n)[:, None] // 4) % 2 == 0, rng.integers(0, hi, (n, n)), 1000 + rng.integers(0, 20, (n, n)))
    elif kind == 7:                                                  # values around 2048 .. 4095 (12-bit container, Q7 flags)
        a = rng.integers(0, 4096, (n, n)) if rng.random() < 0.5 else 2000 + rng.integers(0, 200, (n, n))
    else:
        a = np.clip(rng.normal(800, float(rng.uniform(5, 400)), (n, n)), 0, hi - 1)
    a = np.asarray(a).astype(np.int64)
    if signed:
        return (np.clip(a, 0, 65535) - 1000).astype(np.int16)
    return np.clip(a, 0, 65535).astype(np.uint16)


def run_path(L, d_img, n, w, params, nb, stride, tile, tpg):
    _ffi.check(L.cct_set_option(b"tile_path", tile))
    if tpg:
        _ffi.check(L.cct_set_option(b"stream_tpg", tpg))
    d_pay, d_sz, d_st = DeviceBuffer(n * stride), DeviceBuffer(4 * n), DeviceBuffer(4 * n)
    d_stats, d_roles = DeviceBuffer(16 * n), DeviceBuffer(n * nb)
    d_pay.zero()
    encode_payload_dev(d_img, n, w, w, params, d_pay, d_sz, d_st, d_stats, d_roles)
    sizes = d_sz.download(np.uint32, n)
    return (sizes, d_st.download(np.uint32, n), d_stats.download(np.uint32, 4 * n), d_roles.download(np.uint8, n * nb),
            [d_pay.download(np.uint8, int(sizes[i]), offset=i * stride).tobytes() for i in range(n)])


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    L = _ffi.lib()
    t0, nbad, ncase = time.time(), 0, 0
    for r in range(rounds):
        n_px = int(rng.choice([128, 128, 256, 256, 512, 512, 512, 1024]))
        n = int(rng.integers(1, 7)) if n_px < 1024 else int(rng.integers(1, 3))
        signed = rng.random() < 0.15
        cfg = cct_hip.default_config()
        cfg["encoder"]["transforms"]["segmentation"] = bool(rng.random() < 0.9)
        imgs = np.stack([image(rng, n_px, signed) for _ in range(n)])
        params = codec_params(cfg, imgs.dtype)
        nb = n_px * n_px // 16
        stride = payload_stride(n_px, n_px, 16)
        d_img = DeviceBuffer.from_numpy(imgs)
        ref = run_path(L, d_img, n, n_px, params, nb, stride, 0, 0)
        ncase += n
        for tpg in (4, 2, 1):
            got = run_path(L, d_img, n, n_px, params, nb, stride, 4, tpg)
            same = all(np.array_equal(a, b) for a, b in zip(got[:4], ref[:4])) and got[4] == ref[4]
            if not same:
                nbad += 1
                np.save(f"/tmp/fuzz_stream_{seed}_{r}.npy", imgs)
                which = [k for k, (a, b) in enumerate(zip(got[:4], ref[:4])) if not np.array_equal(a, b)] + ([4] if got[4] != ref[4] else [])
                print(f"MISMATCH round {r}: {n} x {n_px}^2 tpg {tpg} signed {signed} seg {cfg['encoder']['transforms']['segmentation']} fields {which}", flush=True)
        if r % 4 == 0:  # the oracle is slow on 1024^2: a sample keeps the generic kernel honest
            i = int(rng.integers(0, n))
            want = oracle.encode(imgs[i], segmentation=cfg["encoder"]["transforms"]["segmentation"], deflate=False)[13:]
            if want != ref[4][i]:
                nbad += 1
                print(f"ORACLE MISMATCH round {r}: {n_px}^2 slice {i}", flush=True)
        if r % 20 == 19:
            print(f"round {r + 1}/{rounds}  {time.time() - t0:.0f} s  slices {ncase}  mismatches: {nbad}", flush=True)
    _ffi.check(L.cct_set_option(b"tile_path", 1))
    _ffi.check(L.cct_set_option(b"stream_tpg", 4))
    print("fuzz clean" if nbad == 0 else f"{nbad} MISMATCHES", ncase, "slices")
    sys.exit(1 if nbad else 0)


if __name__ == "__main__":
    main()
