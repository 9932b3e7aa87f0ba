"""Device DEFLATE (deflate_kernels.hip) against the system libz: byte-identical zlib streams
(zlib 1.2.11 level 9 -- the reference's zlib.compress(data, level=9), core.py:340)."""
import os
import zlib

import numpy as np
import pytest

import golden_inputs as gi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[256, 512], ids=["inflate256", "inflate512"])
def hip(request):
    """Every test of this module runs with both geometries of the INFLATE kernel (256 / 512 lanes per stream; the library
    picks 512 unless an encode call is in flight, option "inflate_lanes" forces one)."""
    import cct_hip
    from cct_hip import _ffi
    cct_hip.device_info()
    _ffi.check(_ffi.lib().cct_set_option(b"inflate_lanes", request.param))
    yield cct_hip
    _ffi.check(_ffi.lib().cct_set_option(b"inflate_lanes", 0))


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


def test_both_sort_record_layouts(hip):
    """Slices below 4 MiB sort compact records (position, hash and the first five bytes of the string; the match kernel then
    reads candidates from registers), larger ones the wide records of rounds 1-2: option "deflate_compact_records" = 0 forces
    the wide layout on the same inputs, and a slice above the bound takes it by itself."""
    from cct_hip import _ffi
    from oracle import oracle
    rng = np.random.default_rng(77)
    blobs = [b"", b"abc", b"abcabcabcabc" * 10, b"y" * 259, b"z" * 70000, bytes(range(256)) * 3,
             rng.integers(0, 3, 65274, dtype=np.uint8).tobytes(), rng.integers(0, 3, 98043, dtype=np.uint8).tobytes(),
             rng.integers(0, 16, 120000, dtype=np.uint8).tobytes(), rng.integers(0, 256, 70000, dtype=np.uint8).tobytes(),
             oracle.encode(gi.ct_phantom(0), deflate=False)[13:], oracle.encode(gi.load_slice("slice0671"), deflate=False)[13:]]
    L = _ffi.lib()
    try:
        _ffi.check(L.cct_set_option(b"deflate_compact_records", 0))
        _check(hip, blobs)
    finally:
        _ffi.check(L.cct_set_option(b"deflate_compact_records", 1))
    _check(hip, blobs)
    try:  # the pass in one stream (no side branch)
        _ffi.check(L.cct_set_option(b"deflate_fork", 0))
        _check(hip, blobs)
    finally:
        _ffi.check(L.cct_set_option(b"deflate_fork", 1))
    big = np.tile(np.frombuffer(oracle.encode(gi.ct_phantom(1), deflate=False)[13:], dtype=np.uint8), 18)[: (1 << 22) + 70001]
    big = big.copy()
    big[::4099] ^= 0x55  # not one period repeated
    _check(hip, [big.tobytes(), blobs[-1]])


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
    # the same through a page-locked archive (cct_host_alloc): copied without the staging pass, same bytes
    pin = hip.PinnedArray(cap)
    offs2 = np.zeros(n + 1, dtype=np.uint64)
    _ffi.check(L.cct_encode_batch_packed(imgs.ctypes.data, 0, n, w, h, bs, flags, eof, magic, ch, bpc, pin.array.ctypes.data, cap,
                                         offs2.ctypes.data, sizes.ctypes.data, status.ctypes.data, None, None))
    assert np.array_equal(offs2, offs) and np.array_equal(pin.array[:int(offs[n])], arch[:int(offs[n])])
    out2 = np.empty((n, w, h), dtype=np.uint16)
    _ffi.check(L.cct_decode_batch(pin.array.ctypes.data, offs2.ctypes.data, n, bs, magic, out2.ctypes.data, 0, out2.size, st.ctypes.data))
    assert np.array_equal(out2, imgs)
    pin.free()


def _decode_both(hip, files, cfg):
    """decode a batch with the device INFLATE and with libz on the host; both must agree"""
    from cct_hip import _ffi
    L = _ffi.lib()
    outs = []
    for dev in (1, 0):
        _ffi.check(L.cct_set_option(b"device_inflate", dev))
        try:
            outs.append(hip.decode_batch(files, cfg))
        finally:
            _ffi.check(L.cct_set_option(b"device_inflate", 1))
    assert np.array_equal(outs[0], outs[1])
    return outs[0]


def test_device_inflate_golden_and_real_slices(hip):
    cfg = hip.default_config()
    imgs = np.stack([gi.load_slice("slice0671"), gi.load_slice("slice3706"), gi.ct_phantom(0), gi.ct_phantom(1)])
    with open(os.path.join(gi.GOLDEN, "slice0671.cct"), "rb") as f:
        ref0671 = f.read()  # the reference's own encoder output
    from oracle import oracle
    files = [ref0671] + [oracle.encode(im) for im in imgs[1:]]
    assert np.array_equal(_decode_both(hip, files, cfg), imgs)


def test_device_inflate_streams_from_other_encoders(hip):
    """Streams our DEFLATE would never write: other levels (fixed-Huffman blocks at level 1 on tiny inputs,
    different block splits), stored blocks (level 0 and incompressible data)."""
    cfg = hip.default_config()
    cfg["encoder"]["deflate_compression"] = False
    raw = hip.encode_batch(np.stack([gi.ct_phantom(5, 128), gi.ct_phantom(6, 128)]), cfg)
    rng = np.random.default_rng(3)
    noise = rng.integers(0, 2048, size=(128, 128)).astype(np.uint16)
    raw.append(hip.encode_batch(noise[None], cfg)[0])
    imgs = np.stack([gi.ct_phantom(5, 128), gi.ct_phantom(6, 128), noise])
    cfg["encoder"]["deflate_compression"] = True
    for level in (0, 1, 6, 9):
        files = [r[:12] + b"\x01" + zlib.compress(r[13:], level) for r in raw]
        assert np.array_equal(_decode_both(hip, files, cfg), imgs)
    # raw deflate features: zlib streams with trailing garbage decode fine (zlib.decompress ignores it)
    files = [r[:12] + b"\x01" + zlib.compress(r[13:], 9) + b"trailing" for r in raw]
    assert np.array_equal(_decode_both(hip, files, cfg), imgs)


def test_device_inflate_rejects_bad_streams_like_zlib(hip):
    cfg = hip.default_config()
    good = hip.encode_batch(gi.ct_phantom(9, 128)[None], cfg)[0]
    bad_adler = good[:-1] + bytes([good[-1] ^ 1])
    truncated = good[: len(good) - 9]
    bad_header = good[:13] + b"\x78\xdb" + good[15:]
    bad_block = good[:15] + b"\xff\xff\xff" + good[18:]
    for blob in (bad_adler, truncated, bad_header, bad_block):
        with pytest.raises(zlib.error):
            zlib.decompress(blob[13:])
        from cct_hip import _ffi
        for dev in (1, 0):
            _ffi.check(_ffi.lib().cct_set_option(b"device_inflate", dev))
            try:
                with pytest.raises(zlib.error):
                    hip.decode_batch([blob], cfg)
            finally:
                _ffi.check(_ffi.lib().cct_set_option(b"device_inflate", 1))


# ---------------------------------------------------------------------------------------------------------
# device INFLATE on arbitrary data (cct_zlib_decompress_batch): everything libz can write must come back
def _rng_bytes(seed, n, alphabet):
    return np.random.default_rng(seed).integers(0, alphabet, n, dtype=np.uint8).tobytes()


def _inflate_cases():
    import zlib as z
    blobs = {
        "empty": b"", "one": b"a", "zeros_1M": bytes(1 << 20), "zeros_258": bytes(258), "ab_rep": b"ab" * 50000,
        "random_64k": _rng_bytes(1, 65536, 256), "random_300k_a4": _rng_bytes(2, 300000, 4),
        "text_like": (b"the quick brown fox jumps over the lazy dog. " * 4000),
        "runs_mixed": b"".join(bytes([i % 7]) * (i % 300 + 1) for i in range(3000)),
        "far_matches": _rng_bytes(3, 32768, 256) * 6,
    }
    cases = []
    for name, b in blobs.items():
        for level in (1, 6, 9):
            cases.append((f"{name}_l{level}", z.compress(b, level), b))
        for strat, sname in ((z.Z_FIXED, "fixed"), (z.Z_RLE, "rle"), (z.Z_HUFFMAN_ONLY, "huff"), (z.Z_FILTERED, "filt")):
            c = z.compressobj(9, z.DEFLATED, 15, 9, strat)
            cases.append((f"{name}_{sname}", c.compress(b) + c.flush(), b))
        c = z.compressobj(0)
        cases.append((f"{name}_stored", c.compress(b) + c.flush(), b))
        c = z.compressobj(6, z.DEFLATED, 9)  # 512-byte window
        cases.append((f"{name}_w9", c.compress(b) + c.flush(), b))
        c = z.compressobj(6)  # several flush points: empty stored blocks and block boundaries everywhere
        parts = [c.compress(b[i:i + 7001]) + c.flush(z.Z_SYNC_FLUSH) for i in range(0, len(b), 7001)]
        cases.append((f"{name}_syncflush", b"".join(parts) + c.flush(), b))
    return cases


def test_device_inflate_alone_matches_zlib_on_many_encoders(hip):
    cases = _inflate_cases()
    outs = hip.zlib_decompress_batch([c[1] for c in cases], max_out=max(len(c[2]) for c in cases))
    for (name, _, want), got in zip(cases, outs):
        assert got == want, name


def test_device_inflate_alone_reports_errors_per_stream(hip):
    import zlib as z
    good = z.compress(b"hello world" * 1000, 9)
    bad_adler = good[:-1] + bytes([good[-1] ^ 1])
    truncated = good[: len(good) // 2]
    bad_header = b"\x78\x9b" + good[2:]
    too_long = z.compress(bytes(100000), 9)
    garbage = _rng_bytes(9, 500, 256)
    outs, status = hip.zlib_decompress_batch([good, bad_adler, truncated, bad_header, too_long, garbage, good], max_out=20000,
                                             raise_errors=False)
    assert list(status[:4]) == [0, 2, 2, 2] and status[4] == 6 and status[5] == 2 and status[6] == 0  # CCT_E_ZLIB = 2, CCT_E_CAP = 6
    assert outs[0] == outs[6] == b"hello world" * 1000
    for s_ in (bad_adler, truncated, bad_header, garbage):
        with pytest.raises(z.error):
            z.decompress(s_)


def test_device_inflate_agrees_with_zlib_on_damaged_streams(hip):
    """Error parity, not just happy-path parity: every single-bit flip of a few valid streams (dynamic, fixed and stored
    blocks) either inflates to the same bytes on both sides or is refused by both -- invalid code sets (over-subscribed,
    incomplete), bad block types, invalid distance / literal codes, distances too far back, stored-length mismatches,
    truncations of the Adler trailer all come up this way."""
    import zlib as z
    rng = np.random.default_rng(11)
    bases = [z.compress(bytes(rng.integers(0, 7, 900, dtype=np.uint8)) + b"abcabcabc" * 40, 9),
             z.compress(b"the quick brown fox jumps over the lazy dog. " * 12, 9),
             z.compressobj(9, z.DEFLATED, 15, 9, z.Z_FIXED).compress(b"fixed huffman block " * 30) + b""]
    c = z.compressobj(9, z.DEFLATED, 15, 9, z.Z_FIXED)
    bases[2] = c.compress(b"fixed huffman block " * 30) + c.flush()
    c = z.compressobj(0)
    bases.append(c.compress(bytes(range(200))) + c.flush())
    streams, expect = [], []
    for base in bases:
        nbits = len(base) * 8
        picks = range(nbits) if nbits <= 1600 else sorted(set(int(x) for x in rng.integers(0, nbits, 1600)))
        for bit in picks:
            s = bytearray(base)
            s[bit >> 3] ^= 1 << (bit & 7)
            s = bytes(s)
            try:
                d = z.decompressobj()
                out = d.decompress(s, 1 << 16)
                ok = d.eof and not d.unconsumed_tail  # zlib.decompress() semantics: the stream must be complete
                want = out if ok else None
            except z.error:
                want = None
            streams.append(s)
            expect.append(want)
    outs, status = hip.zlib_decompress_batch(streams, max_out=1 << 16, raise_errors=False)
    bad = [(i, status[i]) for i, (got, want) in enumerate(zip(outs, expect))
           if (want is None) != (got is None) or (want is not None and got != want)]
    assert not bad, f"{len(bad)} of {len(streams)} damaged streams disagree with zlib, first {bad[:5]}"
    assert sum(w is not None for w in expect) > 0 and sum(w is None for w in expect) > len(expect) // 2


def test_inflate_geometry_choice(hip):
    """Forced by the option, the kernel runs with that many lanes; left alone (0) a decode with no encode call in flight takes 512."""
    import ctypes as C
    from cct_hip import _ffi
    L = _ffi.lib()
    cfg = hip.default_config()
    files = hip.encode_batch(np.stack([gi.ct_phantom(s, 256) for s in (1, 2)]), cfg)
    forced, last = C.c_int(0), C.c_int(0)
    _ffi.check(L.cct_get_option(b"inflate_lanes", C.byref(forced)))
    hip.decode_batch(files, cfg)
    _ffi.check(L.cct_get_option(b"last_inflate_lanes", C.byref(last)))
    assert last.value == forced.value
    try:
        _ffi.check(L.cct_set_option(b"inflate_lanes", 0))
        hip.decode_batch(files, cfg)
        _ffi.check(L.cct_get_option(b"last_inflate_lanes", C.byref(last)))
        assert last.value == 512
        assert L.cct_set_option(b"inflate_lanes", 128) != 0
    finally:
        _ffi.check(L.cct_set_option(b"inflate_lanes", forced.value))
