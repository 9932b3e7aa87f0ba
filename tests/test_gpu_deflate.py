"""Device DEFLATE (deflate_kernels.hip) against the system libz: byte-identical zlib streams
(zlib 1.2.11 level 9 -- the reference's zlib.compress(data, level=9), core.py:340)."""
import os
import zlib

import numpy as np
import pytest

import golden_inputs as gi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import cct_hip
    cct_hip.device_info()
    return cct_hip


def _check(hip, blobs):
    got = hip.zlib_compress_batch(blobs)
    for i, (b, g) in enumerate(zip(blobs, got)):
        want = zlib.compress(b, 9)
        assert g == want, f"blob {i} (len {len(b)}): {len(g)} vs {len(want)} bytes, first diff at " \
                          f"{next((k for k in range(min(len(g), len(want))) if g[k] != want[k]), None)}"


def test_small_inputs(hip):
    _check(hip, [b"", b"a", b"ab", b"abc", b"abcabcabcabc" * 10, bytes(1000), b"x" * 258, b"y" * 259, b"z" * 70000,
                 bytes(range(256)) * 3])


def test_random_alphabets_incl_stored_blocks_and_window_slides(hip):
    blobs = []
    for seed, alphabet, n in [(0, 256, 5000), (1, 4, 70000), (2, 16, 120000), (3, 256, 70000), (4, 2, 40000),
                              (5, 3, 65274), (6, 3, 65275), (7, 3, 65536), (8, 3, 65800), (9, 3, 98043), (10, 64, 33000)]:
        rng = np.random.default_rng(seed)
        blobs.append(rng.integers(0, alphabet, n, dtype=np.uint8).tobytes())
    _check(hip, blobs)


def test_token_payloads(hip):
    from oracle import oracle
    blobs = []
    for name in ("crop128_f1s1d0", "noise64_nodeflate", "q4_block0", "int16_signed"):
        with open(os.path.join(gi.GOLDEN, name + ".cct"), "rb") as f:
            blobs.append(f.read()[13:])
    for img in (gi.load_slice("slice0671"), gi.load_slice("slice3706"), gi.ct_phantom(0), gi.ct_phantom(5, 256)):
        blobs.append(oracle.encode(img, deflate=False)[13:])
    _check(hip, blobs)


def test_encode_batch_host_and_device_deflate_agree(hip):
    from cct_hip import _ffi
    L = _ffi.lib()
    cfg = hip.default_config()
    imgs = np.stack([gi.ct_phantom(60 + i) for i in range(6)])
    _ffi.check(L.cct_set_option(b"device_deflate", 0))
    host = hip.encode_batch(imgs, cfg)
    _ffi.check(L.cct_set_option(b"device_deflate", 1))
    dev = hip.encode_batch(imgs, cfg)
    assert host == dev


def test_packed_archive_output_equals_strided_output(hip):
    """cct_encode_batch_packed writes the same files back to back; cct_decode_batch reads that layout."""
    import ctypes as C
    from cct_hip import _ffi
    L = _ffi.lib()
    cfg = hip.default_config()
    imgs = np.stack([gi.ct_phantom(70 + i, 256) for i in range(5)])
    n, w, h = imgs.shape
    files = hip.encode_batch(imgs, cfg)
    flags, bs, eof, magic, ch, bpc = hip.codec_params(cfg, imgs.dtype)
    cap = sum(len(f) for f in files) + 64
    arch = np.zeros(cap, dtype=np.uint8)
    offs = np.zeros(n + 1, dtype=np.uint64)
    sizes = np.zeros(n, dtype=np.uint32)
    status = np.zeros(n, dtype=np.uint32)
    _ffi.check(L.cct_encode_batch_packed(imgs.ctypes.data, 0, n, w, h, bs, flags, eof, magic, ch, bpc, arch.ctypes.data, cap,
                                         offs.ctypes.data, sizes.ctypes.data, status.ctypes.data, None, None))
    assert [arch[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(n)] == files
    assert int(offs[n]) == sum(len(f) for f in files)
    out = np.empty((n, w, h), dtype=np.uint16)
    st = np.zeros(n, dtype=np.uint32)
    _ffi.check(L.cct_decode_batch(arch.ctypes.data, offs.ctypes.data, n, bs, magic, out.ctypes.data, 0, out.size, st.ctypes.data))
    assert np.array_equal(out, imgs)
