"""GPU parity tests: the HIP path (through the C ABI / ctypes) against the reference-generated
golden fixtures and against the CPU oracle on seeded inputs.  Bit-exact everywhere: the path
is integer/byte work."""
import copy
import hashlib
import json
import os
import zlib

import numpy as np
import pytest

import golden_inputs as gi

pytestmark = pytest.mark.gpu

with open(os.path.join(gi.GOLDEN, "manifest.json")) as _f:
    _MAN = json.load(_f)
CASES = {c["name"]: c for c in _MAN["cases"]}


def _config(case):
    from cct_hip import default_config
    cfg = default_config()
    cfg["verbose"] = False
    o = case["config"]
    cfg["block_size"] = o.get("block_size", 16)
    cfg["encoder"]["transforms"]["fractal"] = o.get("fractal", True)
    cfg["encoder"]["transforms"]["segmentation"] = o.get("segmentation", True)
    cfg["encoder"]["deflate_compression"] = o.get("deflate", True)
    return cfg


@pytest.fixture(scope="module")
def hip():
    import cct_hip
    info = cct_hip.device_info()  # raises if the extension or the GPU is missing: no fallback
    assert "gfx950" in info["name"]
    return cct_hip


@pytest.mark.parametrize("name", sorted(CASES))
def test_encode_golden(hip, name):
    from codec.core import Encoder
    case = CASES[name]
    img = gi.build_input(case["input"])
    cfg = _config(case)
    if "encode_raises" in case:
        with pytest.raises(ValueError):
            Encoder(cfg, img).encode()
        return
    enc = Encoder(cfg, img)
    out = enc.encode()
    assert len(out) == case["len"]
    assert hashlib.sha1(out).hexdigest() == case["sha1"]
    if "file" in case:
        with open(os.path.join(gi.GOLDEN, case["file"]), "rb") as f:
            assert out == f.read()
    assert enc.info["delta"] == case["tokens"]["short"]
    assert enc.info["full"] == case["tokens"]["full"]
    if "jump" in case["tokens"]:
        assert enc.block_jumps_count == case["tokens"]["jump"]


@pytest.mark.parametrize("name", sorted(n for n, c in CASES.items() if "file" in c))
def test_decode_golden(hip, name):
    from codec.core import Decoder
    case = CASES[name]
    with open(os.path.join(gi.GOLDEN, case["file"]), "rb") as f:
        blob = f.read()
    cfg = _config(case)
    if "decode_raises" in case:
        exc = {"OverflowError": OverflowError}[case["decode_raises"]]
        with pytest.raises(exc):
            Decoder(cfg, blob).decode()
        return
    dec = Decoder(cfg, blob).decode()
    assert hashlib.sha1(dec).hexdigest() == case["decoded_sha1"]
    if case["roundtrip"]:
        assert dec == gi.build_input(case["input"]).tobytes()


@pytest.mark.parametrize("name", ["slice0671", "slice3706", "noise64", "q4_block0", "q4_block0_bs4",
                                  "uniform_160x96", "crop128_bs8", "int16_texture", "phantom512_s1"])
def test_block_partition_matches_reference(hip, name):
    """role[] from the kernel == BLOCK_JUMPS of the reference's BlockPartitioner (cluster.py:166)."""
    from cct_hip import DeviceBuffer, codec_params, encode_payload_dev
    from cct_hip.batch import payload_stride
    case = CASES[name]
    img = np.ascontiguousarray(gi.build_input(case["input"]))
    cfg = _config(case)
    w, h = img.shape
    bs = cfg["block_size"]
    nb = w * h // bs
    d_img = DeviceBuffer.from_numpy(img)
    d_pay = DeviceBuffer(payload_stride(w, h, bs))
    d_sz, d_st, d_roles = DeviceBuffer(4), DeviceBuffer(4), DeviceBuffer(nb)
    encode_payload_dev(d_img, 1, w, h, codec_params(cfg, img.dtype), d_pay, d_sz, d_st, None, d_roles)
    roles = d_roles.download(np.uint8, nb)
    jumps = sorted((int(b), int(b) + int(r)) for b, r in enumerate(roles) if 0 < r < 0xFF)
    assert len(jumps) == case["tokens"]["jump"]
    assert hashlib.sha1(np.array(jumps, dtype=np.int32).tobytes()).hexdigest() == case["jumps_sha1"]
    partners = {p for _, p in jumps}
    assert partners == {int(b) for b, r in enumerate(roles) if r == 0xFF}


def test_batch_matches_oracle_and_roundtrips(hip):
    """A mixed batch (phantoms + both real slices) through encode_batch/decode_batch vs the oracle."""
    from oracle import oracle
    cfg = hip.default_config()
    imgs = [gi.ct_phantom(s) for s in (3, 4, 5, 6, 7, 8)] + [gi.load_slice("slice0671"), gi.load_slice("slice3706")]
    batch = np.stack(imgs)
    files, info = hip.encode_batch(batch, cfg, return_info=True)
    for img, f, st in zip(imgs, files, info):
        ref, rst = oracle.encode(img, return_stats=True)
        assert f == ref
        assert (st["n_short"], st["n_full"], st["n_jump"], st["n_difficult"], st["payload_len"]) == \
               (rst.n_short, rst.n_full, rst.n_jump, rst.n_difficult, rst.payload_len)
        assert not st["q7"]
    dec = hip.decode_batch(files, cfg)
    assert dec.dtype == np.uint16 and dec.shape == batch.shape
    assert np.array_equal(dec, batch)


def _decode_like_oracle(hip, cfg, blob, bs=16):
    """Decode one file on the GPU and on the oracle: same bytes, or the same failure (the format
    cannot carry deltas outside [-2047, 2048] -- Q7 -- and the reference then decodes garbage or
    raises OverflowError; the HIP decoder must do exactly the same)."""
    from oracle import oracle
    try:
        want = oracle.decode(blob, block_size=bs)
    except oracle.OracleError as e:
        assert e.code == oracle.E_OVERFLOW
        with pytest.raises(OverflowError):
            hip.decode_batch([blob], cfg)
        return None
    got = hip.decode_batch([blob], cfg)[0].tobytes()
    assert got == want
    return got


@pytest.mark.parametrize("flags", [(1, 1, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)])
@pytest.mark.parametrize("bs", [4, 8, 16, 32, 64])
def test_flag_and_block_size_matrix_vs_oracle(hip, flags, bs):
    from oracle import oracle
    cfg = hip.default_config()
    cfg["block_size"] = bs
    tr = cfg["encoder"]["transforms"]
    tr["fractal"], tr["segmentation"], cfg["encoder"]["deflate_compression"] = map(bool, flags)
    rng = np.random.default_rng(bs * 8 + flags[0] * 4 + flags[1] * 2 + flags[2])
    base = gi.ct_phantom(11, 128).astype(np.int32)
    imgs = np.stack([np.clip(base + rng.integers(-40, 40, size=base.shape) * (k % 3), 0, 4095).astype(np.uint16)
                     for k in range(5)])
    files = hip.encode_batch(imgs, cfg)
    for img, f in zip(imgs, files):
        assert f == oracle.encode(img, block_size=bs, fractal=bool(flags[0]), segmentation=bool(flags[1]),
                                  deflate=bool(flags[2]))
        _decode_like_oracle(hip, cfg, f, bs)


@pytest.mark.parametrize("shape", [(16, 16), (4, 4), (20, 20), (7 * 16, 9), (48, 80), (80, 48), (100, 64), (1, 32),
                                   (32, 1), (768, 16)])
def test_odd_shapes_vs_oracle(hip, shape):
    from oracle import oracle
    cfg = hip.default_config()
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    img = rng.integers(900, 1100, size=shape).astype(np.uint16)
    f = hip.encode_batch(img[None], cfg)[0]
    assert f == oracle.encode(img)
    assert oracle.decode(f) == img.tobytes()
    assert hip.decode_batch([f], cfg)[0].tobytes() == img.tobytes()


def test_every_block_difficult_spills_lists(hip):
    """Noise slices: every block is difficult, so the difficult-block list overflows its LDS part
    (ENC_LIST_CAP) into the HBM workspace, and decode's jump list does likewise."""
    from oracle import oracle
    cfg = hip.default_config()
    rng = np.random.default_rng(99)
    imgs = rng.integers(0, 2048, size=(3, 256, 256)).astype(np.uint16)
    files, info = hip.encode_batch(imgs, cfg, return_info=True)
    for img, f, st in zip(imgs, files, info):
        assert f == oracle.encode(img)
        assert st["n_difficult"] > 3000 and st["n_jump"] > 1000
    assert np.array_equal(hip.decode_batch(files, cfg), imgs)


def test_large_slice_role_table_in_hbm(hip):
    """1024x1024 with block_size 4: 262144 blocks -> role[] lives in the HBM workspace."""
    from oracle import oracle
    cfg = hip.default_config()
    cfg["block_size"] = 4
    img = gi.ct_phantom(21, 1024)
    f = hip.encode_batch(img[None], cfg)[0]
    assert f == oracle.encode(img, block_size=4)
    assert hip.decode_batch([f], cfg)[0].tobytes() == img.tobytes()


def test_errors_mirror_reference(hip):
    from codec.core import Decoder, Encoder
    from cct_hip._ffi import CorruptStreamError
    cfg = hip.default_config()
    cfg["verbose"] = False
    img = gi.ct_phantom(1, 64)
    good = Encoder(cfg, img).encode()
    with pytest.raises(ValueError, match="valid header"):
        Decoder(cfg, b"nope" + good[4:]).decode()
    with pytest.raises(zlib.error):
        Decoder(cfg, good[:13] + b"\x00\x01\x02" + good[16:]).decode()
    with pytest.raises(ValueError):  # 100 % 16 != 0 -> reshape ValueError
        Encoder(cfg, np.zeros((10, 10), np.uint16)).encode()
    c2 = copy.deepcopy(cfg)
    c2["encoder"]["transforms"]["delta"] = False
    with pytest.raises(NotImplementedError):
        Encoder(c2, img).encode()
    c3 = copy.deepcopy(cfg)
    c3["encoder"]["transforms"]["zipper"] = True
    with pytest.raises(NotImplementedError):
        Encoder(c3, img).encode()
    with pytest.raises(TypeError):
        Encoder(cfg, img.astype(np.int32)).encode()
    # truncated raw token stream
    c4 = copy.deepcopy(cfg)
    c4["encoder"]["deflate_compression"] = False
    raw = Encoder(c4, img).encode()
    with pytest.raises(CorruptStreamError):
        Decoder(c4, raw[: len(raw) // 2]).decode()


def test_reserved_tag_bytes_repeat_the_previous_pixel(hip):
    """core.py:496-520 takes no branch for 110xxxxx / 1111xxxx: one byte is consumed and the previous pixel repeats.
    No encoder emits them; the oracle and the HIP decoder both replay the reference on them (same pixels), in raw and
    in deflated files, with and without mesh jumps around them."""
    from oracle import oracle
    from codec.core import Encoder
    cfg = hip.default_config()
    cfg["verbose"] = False
    c4 = copy.deepcopy(cfg)
    c4["encoder"]["deflate_compression"] = False
    rng = np.random.default_rng(77)
    img = gi.ct_phantom(1, 64)
    raw = Encoder(c4, img).encode()
    head, body = raw[:13], bytearray(raw[13:])
    # (1) hand-made stream: 4096 one-pixel tokens, a third of them reserved bytes
    toks = bytearray()
    reserved = [0xC0, 0xC5, 0xDF, 0xF0, 0xF3, 0xFF]
    toks += bytes([0xE3, 0x20])  # first pixel: full delta +800
    val = 800
    for k in range(1, 4096):
        r = rng.integers(0, 3)
        if r == 0:
            toks.append(reserved[rng.integers(0, len(reserved))])
        elif val < 2000 and (r == 1 or val < 500):
            d = int(rng.integers(0, 0x41))  # short delta 0..64
            toks.append(d)
            val += d
        else:
            d = int(rng.integers(0xC0, 0x100))  # full delta -64..-1
            toks += bytes([0xE0 | 0x0F, d])
            val += d - 256
    toks.append(59)
    blob = head + bytes(toks)
    want = oracle.decode(blob)
    got = hip.decode_batch([blob], c4)[0]
    assert got.tobytes() == want
    flat = np.frombuffer(want, np.uint16)
    assert len(np.unique(flat)) > 50  # the stream did something
    # (2) a real stream (with jumps) whose short-delta bytes are overwritten by reserved bytes here and there
    pos, k = [], 0
    while k < len(body) - 1:
        c = body[k]
        if (c & 0xF0) == 0xE0:
            k += 2
            continue
        if c == 0:  # a zero delta and a reserved byte both repeat the pixel: the image must not change
            pos.append(k)
        k += 1
    assert len(pos) > 20
    for k in pos[::2]:
        body[k] = reserved[k % len(reserved)]
    blob2 = head + bytes(body)
    got2 = hip.decode_batch([blob2], c4)[0]
    assert got2.tobytes() == oracle.decode(blob2)
    assert np.array_equal(got2.reshape(img.shape), img)
    # (3) the same bytes behind DEFLATE
    blob3 = head[:12] + b"\x01" + zlib.compress(bytes(body), 9)
    assert hip.decode_batch([blob3], cfg)[0].tobytes() == oracle.decode(blob3)
    # deliberate difference that stays: two jump bytes in a row
    from cct_hip._ffi import CorruptStreamError
    j = next(k for k in range(len(raw) - 13) if (raw[13 + k] & 0xC0) == 0x80 and (k == 0 or (raw[13 + k - 1] & 0xF0) != 0xE0))
    bad = bytearray(raw)
    bad.insert(13 + j, raw[13 + j])
    with pytest.raises(CorruptStreamError):
        hip.decode_batch([bytes(bad)], c4)


def test_integration_md_ctypes_stub_runs_verbatim(hip):
    """INTEGRATION.md option B shows the binding a maintainer of the reference would add.  The code block is
    executed here exactly as printed (the library is already loaded by the package; dlopen finds it by its soname)
    and must produce the reference's bytes (golden fixture) and decode them back."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "INTEGRATION.md")) as f:
        text = f.read()
    sect = text[text.index("## B."):]
    code = sect[sect.index("```python") + len("```python"):]
    code = code[: code.index("```")]
    assert "def encode(cfg, image)" in code and "def decode(cfg, file_bytes, width, height)" in code
    ns = {}
    exec(compile(code, "INTEGRATION.md:option-B", "exec"), ns)
    cfg = hip.default_config()
    done = 0
    for name in ("slice0671", "slice3706", "noise64", "q4_block0"):
        case = CASES[name]
        o = case["config"]
        if (o.get("block_size", 16), o.get("fractal", True), o.get("segmentation", True), o.get("deflate", True)) != (16, True, True, True):
            continue
        img = gi.build_input(case["input"])
        got = ns["encode"](cfg, img)
        assert len(got) == case["len"] and hashlib.sha1(got).hexdigest() == case["sha1"], name
        want = got
        done += 1
        back = ns["decode"](cfg, got, img.shape[0], img.shape[1])
        assert back == np.ascontiguousarray(img).tobytes()
    assert done >= 2
    with pytest.raises(ValueError, match="valid header"):
        ns["decode"](cfg, b"nope" + want[4:], 64, 64)


def test_fork_before_first_use_and_fork_after_init(hip):
    """scripts/evaluate.py:107 pattern: workers forked before the first device call each bind the GPU themselves and
    produce the same bytes as the parent; a child forked after the parent initialised the GPU is refused with
    DeviceError (CCT_E_DEVICE).  Runs in a fresh interpreter (tests/fork_check.py): this process already owns a context."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "fork_check.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["pool_pids"] >= 1 and not rep["parent_pid_in_pool"]
    assert rep["pool_hashes"] == rep["parent_hashes"]
    assert rep["late_child"][0] == "DeviceError", rep
    assert "forked after" in rep["late_child"][2]
    assert rep["late_exit"] == 0


def test_encoder_decoder_files_and_preview(hip, tmp_path):
    """scripts/demo.py's flow: encode to a file, decode to a PNG preview, zero error, equal SHA-1."""
    from PIL import Image
    from codec.core import Decoder, Encoder
    cfg = hip.default_config()
    cfg["verbose"] = True
    img = gi.load_slice("slice0671")
    cct = tmp_path / "testing.cct"
    out = Encoder(cfg, img, str(cct)).encode()
    assert cct.read_bytes() == out
    png = tmp_path / "decoded-testing.png"
    dec = Decoder(cfg, cct.read_bytes(), str(png))
    pixels = dec.decode()
    assert pixels.shape == (512, 512) and pixels.dtype == np.uint16
    assert np.count_nonzero(img - pixels) == 0
    assert hashlib.sha1(img.tobytes()).hexdigest() == hashlib.sha1(pixels.tobytes()).hexdigest()
    assert (dec.width, dec.height, dec.channels, dec.bytes_per_channel) == (512, 512, 1, 2)
    assert dec.fractal_transform and dec.segmentation_transform and dec.deflate_compression
    prev = np.array(Image.open(png))
    assert np.array_equal(prev.astype(np.uint32), img.astype(np.uint32) << 4)


def test_full_batch_properties(hip):
    """BASELINE config 2 size (256 x 512x512): round trip is the identity, payload length equals
    N + n_full + n_jump + 1, and a checksum of checksums matches the oracle on a sample."""
    from oracle import oracle
    from cct_hip import DeviceBuffer, codec_params, decode_payload_dev, encode_payload_dev
    from cct_hip.batch import payload_stride
    from cct_hip._ffi import SliceStats
    import ctypes as C
    cfg = hip.default_config()
    n, w, h, bs = 256, 512, 512, 16
    base = [gi.ct_phantom(s) for s in range(32)]
    sym = [lambda a: a, np.fliplr, np.flipud, lambda a: a.T, lambda a: np.fliplr(a).T, lambda a: np.flipud(a).T,
           lambda a: np.flipud(np.fliplr(a)), lambda a: np.flipud(np.fliplr(a)).T]
    imgs = np.stack([np.ascontiguousarray(sym[i // 32](base[i % 32])) for i in range(n)])
    stride = payload_stride(w, h, bs)
    d_img = DeviceBuffer.from_numpy(imgs)
    d_pay, d_sz, d_st = DeviceBuffer(n * stride), DeviceBuffer(4 * n), DeviceBuffer(4 * n)
    d_stats = DeviceBuffer(16 * n)
    d_out = DeviceBuffer(imgs.nbytes)
    params = codec_params(cfg, imgs.dtype)
    encode_payload_dev(d_img, n, w, h, params, d_pay, d_sz, d_st, d_stats)
    sizes = d_sz.download(np.uint32, n)
    est = d_st.download(np.uint32, n)
    # the phantoms live in [0, 2047]: no traversal delta can leave the format's range, so not even the informational Q7 bit
    assert not est.any()
    clean = est == 0
    stats = d_stats.download(np.uint32, 4 * n).reshape(n, 4)
    assert np.array_equal(sizes, w * h + stats[:, 1] + stats[:, 2] + 1)
    assert np.array_equal(stats[:, 0] + stats[:, 1], np.full(n, w * h))
    d_out.zero()
    decode_payload_dev(d_pay, stride, d_sz, n, w, h, bs, True, d_out, d_st)
    dst = d_st.download(np.uint32, n)
    assert not dst[clean].any()
    back = d_out.download(np.uint16, n * w * h).reshape(n, w, h)
    assert np.array_equal(back[clean], imgs[clean])
    for i in (0, 17, 100, 255):
        pay = d_pay.download(np.uint8, int(sizes[i]), offset=i * stride).tobytes()
        assert oracle.encode(imgs[i], deflate=False)[13:] == pay


def test_twelve_bit_phantoms_stay_inside_the_format(hip):
    """cct_hip.synth.ct_phantom(depth12=True) -- the bench workload: values above 2047 like the real corpus, and by
    construction no delta outside [-2047, 2048] whatever the partition: no Q7 flag, exact round trip, oracle bytes."""
    from oracle import oracle
    from cct_hip.synth import ct_phantom
    cfg = hip.default_config()
    imgs = np.stack([ct_phantom(1000 + i, 512, True) for i in range(24)])
    assert imgs.max() > 2047 and imgs.max() < 4096
    files, info = hip.encode_batch(imgs, cfg, return_info=True)
    assert not any(st["q7"] for st in info)
    assert np.array_equal(hip.decode_batch(files, cfg), imgs)
    for i in (0, 7, 23):
        assert files[i] == oracle.encode(imgs[i])
    big = ct_phantom(77, 1024, True)
    f = hip.encode_batch(big[None], cfg)[0]
    assert f == oracle.encode(big) and hip.decode_batch([f], cfg)[0].tobytes() == big.tobytes()


def test_unsupported_block_sizes_are_refused(hip):
    """The reference accepts any divisor of W*H as block_size (core.py:245); the HIP path carries 4, 8, 16, 32 and 64 (what
    config.json ships and SURVEY App. C pins) and refuses the rest with CCT_E_ARG -> ValueError, before touching the device."""
    img = gi.ct_phantom(5, 128)
    for bs in (2, 128, 1, 256, 512):
        cfg = hip.default_config()
        cfg["block_size"] = bs
        with pytest.raises(ValueError, match="block_size"):
            hip.encode_batch(img[None], cfg)
    cfg = hip.default_config()
    cfg["block_size"] = 3   # not a divisor: the reference's own error (numpy reshape, core.py:245)
    with pytest.raises(ValueError, match="cannot reshape"):
        hip.encode_batch(img[None], cfg)


@pytest.mark.parametrize("n_px", [128, 256, 512, 1024])
def test_decode_tile_tables_equal_traversal_table(hip, n_px):
    """decode_kernel maps positions to raster offsets through the LDS pattern tables when the traversal is made of
    64x64 tiles, through the HBM traversal table otherwise: both must rebuild the same rasters (= the input)."""
    from cct_hip import _ffi
    cfg = hip.default_config()
    L = _ffi.lib()
    imgs = np.stack([gi.ct_phantom(90 + i, n_px) for i in range(3)])
    files = hip.encode_batch(imgs, cfg)
    outs = []
    for tile in (1, 0):
        _ffi.check(L.cct_set_option(b"tile_path", tile))
        try:
            outs.append(hip.decode_batch(files, cfg))
        finally:
            _ffi.check(L.cct_set_option(b"tile_path", 1))
    assert np.array_equal(outs[0], outs[1])
    assert np.array_equal(np.asarray(outs[0]).reshape(imgs.shape), imgs)


def _last_path(L):
    import ctypes as C
    from cct_hip import _ffi
    v = C.c_int(-9)
    _ffi.check(L.cct_get_option(b"last_encode_path", C.byref(v)))
    return v.value


@pytest.mark.parametrize("n_px", [128, 256, 512, 1024])
def test_tile_path_equals_generic_path(hip, n_px):
    """Four implementations of stage (i) must agree byte for byte (payload, sizes, statistics, block roles) -- and
    with the oracle: the streaming kernel (encode_stream.hip, the default), the four-kernel pipeline (encode_pipe.hip),
    the one-workgroup-per-slice tile kernel and the generic LUT-gather kernel.  The test also checks that each path
    really ran (no silent fallback)."""
    from oracle import oracle
    from cct_hip import DeviceBuffer, codec_params, encode_payload_dev, _ffi
    from cct_hip.batch import payload_stride
    cfg = hip.default_config()
    L = _ffi.lib()
    n = 5
    imgs = np.stack([gi.ct_phantom(40 + i, n_px) for i in range(n)])
    rng = np.random.default_rng(n_px)
    if n_px <= 512:
        imgs[1] = rng.integers(0, 2048, size=(n_px, n_px))  # every block difficult: spilled lists, long islands
    imgs[3] = np.clip(imgs[3].astype(np.int32) * 20, 0, 65535)  # pixels >= 0x4000: the exact (unpacked) arithmetic path
    imgs[4, :64, :64] = rng.integers(0, 2048, size=(64, 64))   # difficult block 0 (Q4), islands across tile borders
    w = h = n_px
    nb = w * h // 16
    stride = payload_stride(w, h, 16)
    d_img = DeviceBuffer.from_numpy(imgs)
    res = []
    # option value, implementation that must have run, tiles per workgroup of the streaming kernel (4 is the default)
    for tile, ran, tpg in ((4, 3, 4), (4, 3, 2), (4, 3, 1), (3, 1, 4), (2, 2, 4), (0, 0, 4)):
        _ffi.check(L.cct_set_option(b"tile_path", tile))
        _ffi.check(L.cct_set_option(b"stream_tpg", tpg))
        d_pay, d_sz, d_st = DeviceBuffer(n * stride), DeviceBuffer(4 * n), DeviceBuffer(4 * n)
        d_stats, d_roles = DeviceBuffer(16 * n), DeviceBuffer(n * nb)
        d_pay.zero()
        encode_payload_dev(d_img, n, w, h, codec_params(cfg, imgs.dtype), d_pay, d_sz, d_st, d_stats, d_roles)
        assert _last_path(L) == ran
        sizes = d_sz.download(np.uint32, n)
        res.append((sizes, d_st.download(np.uint32, n), d_stats.download(np.uint32, 4 * n),
                    d_roles.download(np.uint8, n * nb),
                    [d_pay.download(np.uint8, int(sizes[i]), offset=i * stride).tobytes() for i in range(n)]))
    _ffi.check(L.cct_set_option(b"tile_path", 1))
    _ffi.check(L.cct_set_option(b"stream_tpg", 4))
    for other in res[1:]:
        for a_, b_ in zip(res[0][:4], other[:4]):
            assert np.array_equal(a_, b_)
        assert res[0][4] == other[4]
    assert not (res[0][1] & ~np.uint32(1)).any()
    for i in range(n):
        assert res[0][4][i] == oracle.encode(imgs[i], deflate=False)[13:]


def test_default_path_by_shape(hip):
    """The default choice among the tile paths: the streaming kernel wherever the traversal is made of 64x64 tiles of 4x4 blocks."""
    from cct_hip import _ffi
    L = _ffi.lib()
    cfg = hip.default_config()
    hip.encode_batch(gi.ct_phantom(3, 512)[None], cfg)
    assert _last_path(L) == 3
    hip.encode_batch(gi.ct_phantom(3, 1024)[None], cfg)
    assert _last_path(L) == 3
    hip.encode_batch(gi.ct_phantom(3, 64)[None], cfg)   # one tile only: not a tiled shape -> generic kernel
    assert _last_path(L) == 0


@pytest.mark.parametrize("tile, ran", [(4, 3), (3, 1)])
def test_pipeline_signed_and_flag_variants(hip, tile, ran):
    """The streaming kernel and the staged pipeline with int16 input (segmentation sees signed values), segmentation off
    and EOF handling, against the oracle, on 512x512."""
    from oracle import oracle
    from cct_hip import _ffi
    L = _ffi.lib()
    cfg = hip.default_config()
    rng = np.random.default_rng(5)
    base = gi.ct_phantom(61).astype(np.int32)
    signed = (base - 1000 + rng.integers(-30, 30, size=base.shape)).astype(np.int16)
    _ffi.check(L.cct_set_option(b"tile_path", tile))
    try:
        f = hip.encode_batch(signed[None], cfg)[0]
        assert _last_path(L) == ran
        assert f == oracle.encode(signed)
        cfg2 = hip.default_config()
        cfg2["encoder"]["transforms"]["segmentation"] = False
        img = gi.ct_phantom(62)
        f2 = hip.encode_batch(img[None], cfg2)[0]
        assert _last_path(L) == ran
        assert f2 == oracle.encode(img, segmentation=False)
        assert hip.decode_batch([f2], cfg2)[0].tobytes() == img.tobytes()
    finally:
        _ffi.check(L.cct_set_option(b"tile_path", 1))


def test_config4_1024_square_batch(hip):
    """BASELINE configs[3]: 1024x1024 slices, block_size 16 -> tile kernel with role[] in the HBM
    workspace (65536 blocks do not fit LDS next to the rings); checked against the oracle."""
    from oracle import oracle
    cfg = hip.default_config()
    imgs = np.stack([gi.ct_phantom(80 + i, 1024) for i in range(4)])
    files, info = hip.encode_batch(imgs, cfg, return_info=True)
    for img, f, st in zip(imgs, files, info):
        ref, rst = oracle.encode(img, return_stats=True)
        assert f == ref
        assert (st["n_jump"], st["n_difficult"], st["payload_len"]) == (rst.n_jump, rst.n_difficult, rst.payload_len)
    assert np.array_equal(hip.decode_batch(files, cfg), imgs)


def test_more_slices_than_compute_units(hip):
    """BASELINE configs[2] shape of work: many more slices than CUs in one call (here 700 small ones),
    contiguous shards as cct_hip.parallel.shard_range would hand them out; sample checked vs the oracle."""
    from oracle import oracle
    from cct_hip.parallel import shard_range
    cfg = hip.default_config()
    base = [gi.ct_phantom(90 + i, 128) for i in range(20)]
    rng = np.random.default_rng(3)
    n = 700
    imgs = np.stack([np.clip(base[i % 20].astype(np.int32) + rng.integers(-8, 9, size=(128, 128)), 0, 2047).astype(np.uint16)
                     for i in range(n)])
    files = []
    for r in range(3):  # three "ranks"
        lo, hi = shard_range(n, r, 3)
        files += hip.encode_batch(imgs[lo:hi], cfg)
    assert len(files) == n
    for i in (0, 233, 234, 466, 699):
        assert files[i] == oracle.encode(imgs[i])
    back = hip.decode_batch(files, cfg)
    assert np.array_equal(back, imgs)


@pytest.mark.timeout(600)
def test_config3_corpus_3954(hip):
    """BASELINE configs[2]: the 3954-slice corpus: phantoms of seed 0..3953 with the two real
    CT slices at #671 and #3706, encoded and decoded as the eight contiguous shards shard_range(3954, r, 8) (495/494
    slices, what the eight GPUs of a node get, scripts/evaluate.py:107-119 being the reference's fan-out) on this one
    GPU.  Exact round trip of every slice, the sizes of all 3954 files summed, files #671 / #3706 equal to the reference's
    own outputs (tests/golden), a sample of files against the oracle."""
    from oracle import oracle
    from cct_hip.parallel import shard_range
    cfg = hip.default_config()
    n = 3954
    real = {671: "slice0671", 3706: "slice3706"}
    # 64 distinct phantoms under the eight symmetries of the square and small seeded noise: distinct slices at the cost
    # of 64 phantom generations (the generator takes ~0.1 s per slice)
    base = [gi.ct_phantom(s) for s in range(64)]
    sym = [lambda a: a, np.fliplr, np.flipud, lambda a: a.T, lambda a: np.fliplr(a).T, lambda a: np.flipud(a).T,
           lambda a: np.flipud(np.fliplr(a)), lambda a: np.flipud(np.fliplr(a)).T]

    def make(i):
        if i in real:
            return gi.load_slice(real[i])
        rng = np.random.default_rng(i)
        a = sym[(i // 64) % 8](base[i % 64]).astype(np.int32)
        return np.clip(a + rng.integers(-3, 4, size=a.shape) * (a > 0), 0, 2047).astype(np.uint16)

    total = 0
    sample = {0, 494, 495, 671, 1977, 3706, 3953}
    for r in range(8):
        lo, hi = shard_range(n, r, 8)
        assert hi - lo in (494, 495)
        imgs = np.stack([make(i) for i in range(lo, hi)])
        files = hip.encode_batch(imgs, cfg)
        assert len(files) == hi - lo
        total += sum(len(f) for f in files)
        back = hip.decode_batch(files, cfg)
        assert np.array_equal(back, imgs)
        for i in sorted(sample):
            if lo <= i < hi:
                assert files[i - lo] == oracle.encode(imgs[i - lo])
        for i, name in real.items():
            if lo <= i < hi:
                with open(os.path.join(gi.GOLDEN, name + ".cct"), "rb") as f:
                    assert files[i - lo] == f.read()
    assert 3954 * 100_000 < total < 3954 * 300_000


@pytest.mark.timeout(600)
def test_config4_full_batch_512(hip):
    """BASELINE configs[3] at full size: 512 slices of 1024x1024 in one call (1.07 GB of pixels: the device DEFLATE
    runs in several bounded passes, api.cpp's chunk loop).  Exact round trip, a sample against the oracle."""
    from oracle import oracle
    cfg = hip.default_config()
    base = [gi.ct_phantom(200 + s, 1024) for s in range(8)]
    n = 512
    imgs = np.empty((n, 1024, 1024), np.uint16)
    for i in range(n):
        rng = np.random.default_rng(1000 + i)
        a = base[i % 8]
        a = a.T if (i // 8) % 2 else a
        imgs[i] = np.clip(a.astype(np.int32) + rng.integers(-2, 3, size=a.shape) * (a > 0), 0, 2047)
    files = hip.encode_batch(imgs, cfg)
    assert len(files) == n
    for i in (0, 255, 256, 511):
        assert files[i] == oracle.encode(imgs[i])
    back = hip.decode_batch(files, cfg)
    assert np.array_equal(back, imgs)


def test_tools_demo_and_evaluate(hip, tmp_path, capsys):
    """the pydicom-free counterparts of scripts/demo.py and scripts/evaluate.py on the two real slices:
    demo's checks (zero error, equal SHA-1, demo.py:85-103) and get_filename naming (demo.py:16-25); evaluate's CSV with
    the CCT sizes of results/encoder-comparisons.csv:628,3618 (207 575 and 205 179 bytes) and its ZIP column
    (270 969 and 273 262)."""
    import importlib.util
    tools = os.path.join(os.path.dirname(gi.HERE), "tools")

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(tools, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    demo, evaluate = load("demo"), load("evaluate")
    src = tmp_path / "corpus"
    src.mkdir()
    np.save(src / "1-016.npy", gi.load_slice("slice0671"))
    np.save(src / "1-55.npy", gi.load_slice("slice3706"))
    work = tmp_path / "working"
    assert demo.main([str(src / "1-016.npy"), "--workdir", str(work)]) == 0
    out = capsys.readouterr().out
    assert "Total Error: 0" in out and out.count("bc26ff59b9950f2b9857856aec4561c351cf146f") == 2
    with open(os.path.join(gi.GOLDEN, "slice0671.cct"), "rb") as f:
        assert (work / "testing.cct").read_bytes() == f.read()
    cfg = hip.default_config()
    assert demo.get_filename("data/working/testing.cct", False, cfg) == "data/working/decoded-testing.png"
    assert demo.get_filename("x/y/1-016.dcm", True, cfg) == "x/y/encoded-1-016.cct"
    assert (work / "decoded-testing.png").exists()
    csv_path = tmp_path / "evaluation.csv"
    assert evaluate.main([str(src), "--results", str(csv_path)]) == 0
    lines = csv_path.read_text().splitlines()
    assert lines[0] == "File,Raw,ZIP,PNG,RLE,JP2,CCT"
    rows = {ln.split(",")[0]: ln.split(",") for ln in lines[1:]}
    a, b = rows["(0000)-1-016.npy"], rows["(0001)-1-55.npy"]
    assert (a[1], a[2], a[6]) == ("524288", "270969", "207575")
    assert (b[1], b[2], b[6]) == ("524288", "273262", "205179")


def test_encoder_decoder_reference_attributes(hip):
    """Attributes callers of the reference read after encode()/decode(): Encoder.writer (core.py:182), .curve
    (core.py:235), .partition.block_partition() (core.py:258-268; oracle/gen_golden.py relies on it) and Decoder.fulls
    (core.py:508)."""
    from codec.core import Decoder, Encoder
    cfg = hip.default_config()
    cfg["verbose"] = False
    case = CASES["slice0671"]
    img = gi.load_slice("slice0671")
    enc = Encoder(cfg, img)
    out = enc.encode()
    assert enc.writer.output() == out and len(enc.writer.output_header()) == 13 and enc.writer.output_data() == out[13:]
    assert enc.curve.generate_all()[:4] == [0, 512, 513, 1]
    order, jumps = enc.partition.block_partition()
    assert len(jumps) == case["tokens"]["jump"]
    assert hashlib.sha1(np.array(sorted(jumps.items()), dtype=np.int32).tobytes()).hexdigest() == case["jumps_sha1"]
    assert sorted(order.tolist()) == list(range(512 * 512))
    dec = Decoder(cfg, out)
    dec.decode()
    assert len(dec.fulls) == case["tokens"]["full"]


def test_rccl_communicator_of_one(hip):
    """The C-ABI exchange step on real RCCL: a communicator of one rank (all this box has), id through the file rendezvous,
    all-gather of sizes, destroy.  The N > 1 case differs only in the number of ranks."""
    import ctypes as C
    from cct_hip import _ffi, parallel
    L = _ffi.lib()
    parallel.comm_init(0, 1)
    try:
        rk, wd = C.c_int(-5), C.c_int(-5)
        _ffi.check(L.cct_comm_info(C.byref(rk), C.byref(wd)))
        assert (rk.value, wd.value) == (0, 1)
        sizes = np.array([207575, 205179, 1, 2, 3], dtype=np.uint32)
        assert np.array_equal(parallel.gather_sizes_rccl(sizes), sizes)
        assert np.array_equal(parallel.file_offsets(parallel.gather_sizes_rccl(sizes))[-1:], [sizes.sum()])
        parallel.barrier()
    finally:
        _ffi.check(L.cct_comm_destroy())


def test_bench_two_ranks_on_one_device(hip, tmp_path):
    """bench.py's N > 1 control flow with two real processes on this box's one GPU: rank-disjoint batches, the size gather after
    every encode, barriers, max-over-ranks time, one JSON line from rank 0.  Only the collective is replaced (files instead of
    RCCL, which refuses two ranks on one device): tests/bench_rank.py."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    world, slices = 2, 32
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", CCT_TEST_EXCHANGE_DIR=str(tmp_path))
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tests", "bench_rank.py"), "--gpus", str(world),
                                       "--steps", "4", "--warmup", "1", "--slices", str(slices)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=420))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, (r, se[-2000:])
    lines0 = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines0) == 1 and not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]
    line = json.loads(lines0[0])
    assert line["n_gpus"] == world and line["verified"] is True and line["scaling"] == "weak"
    assert line["sizes_gathered"] == world * slices          # both ranks' sizes, in global slice order
    assert "cpu_baseline" not in line                        # rank 0 at N = 1 only
    px = world * slices * 512 * 512 * line["steps"]
    assert abs(line["value"] - px / (line["ms_per_step"] * 1e-3 * line["steps"]) / 1e6) / line["value"] < 0.01


def test_concurrent_encodes_and_decodes_use_their_slots(hip):
    """Two encode calls and two decode calls at a time, from four host threads (the library gives each an encode / decode
    slot with its own stream and workspaces): every result equals the serial one."""
    from concurrent.futures import ThreadPoolExecutor
    cfg = hip.default_config()
    batches = [np.stack([gi.ct_phantom(10 * k + i, 256) for i in range(24)]) for k in range(4)]
    serial = [hip.encode_batch(b, cfg) for b in batches]
    for k, b in enumerate(batches):
        assert np.array_equal(hip.decode_batch(serial[k], cfg), b)

    def enc(k):
        return [hip.encode_batch(batches[k], cfg) for _ in range(3)]

    def dec(k):
        return [hip.decode_batch(serial[k], cfg) for _ in range(3)]

    from cct_hip import _ffi
    L = _ffi.lib()
    _ffi.check(L.cct_set_option(b"encode_slots", 2))   # the library's default is one encode batch at a time
    _ffi.check(L.cct_set_option(b"decode_slots", 2))
    with ThreadPoolExecutor(4) as pool:
        fe = [pool.submit(enc, k) for k in (0, 1)]
        fd = [pool.submit(dec, k) for k in (2, 3)]
        for k, f in zip((0, 1), fe):
            for files in f.result():
                assert files == serial[k]
        for k, f in zip((2, 3), fd):
            for back in f.result():
                assert np.array_equal(back, batches[k])
    # one slot each must give the same
    try:
        _ffi.check(L.cct_set_option(b"encode_slots", 1))
        _ffi.check(L.cct_set_option(b"decode_slots", 1))
        with ThreadPoolExecutor(4) as pool:
            fe = [pool.submit(enc, k) for k in (0, 1)]
            fd = [pool.submit(dec, k) for k in (2, 3)]
            assert all(files == serial[k] for k, f in zip((0, 1), fe) for files in f.result())
            assert all(np.array_equal(back, batches[k]) for k, f in zip((2, 3), fd) for back in f.result())
    finally:
        _ffi.check(L.cct_set_option(b"encode_slots", 1))
        _ffi.check(L.cct_set_option(b"decode_slots", 2))


@pytest.mark.parametrize("scheduling", [1, 0], ids=["gate+queue_ahead", "launch_at_once"])
def test_pipelined_packed_calls_with_and_without_the_scheduling_measures(hip, scheduling):
    """The bench's pipeline in small: two threads encode into page-locked archives (cct_encode_batch_packed), one decodes the
    archive of the step before on its own stream.  With the decode gate and the queue-ahead handover and without
    them, every archive equals the serial one and every raster its input; a decode issued while the LAST encode runs
    (nothing follows: the gate gives up after its grace time) returns as well."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    from cct_hip import _ffi
    from cct_hip.batch import PinnedArray, DeviceBuffer
    L = _ffi.lib()
    cfg = hip.default_config()
    flags, bs, eof, magic, ch, bpc = hip.codec_params(cfg, np.uint16)
    n, W = 32, 256
    batches = [np.stack([gi.ct_phantom(100 * k + i, W) for i in range(n)]) for k in range(3)]
    serial = [b"".join(hip.encode_batch(b, cfg)) for b in batches]
    cap = L.cct_file_bound(W, W, bs) * n
    d_imgs = [DeviceBuffer.from_numpy(b) for b in batches]
    NSET = 3
    pins = [PinnedArray(cap) for _ in range(NSET)]
    arch = [p.array for p in pins]
    offs = [np.zeros(n + 1, dtype=np.uint64) for _ in range(NSET)]
    sizes = [np.zeros(n, dtype=np.uint32) for _ in range(NSET)]
    status = [np.zeros(n, dtype=np.uint32) for _ in range(NSET)]
    psizes = [np.zeros(n, dtype=np.uint32) for _ in range(NSET)]
    d_back = [DeviceBuffer(n * W * W * 2) for _ in range(NSET)]

    def enc(i):
        k = i % NSET
        _ffi.check(L.cct_encode_batch_packed(d_imgs[i % 3].ptr, 1, n, W, W, bs, flags, eof, magic, ch, bpc, arch[k].ctypes.data, cap,
                                             offs[k].ctypes.data, sizes[k].ctypes.data, status[k].ctypes.data,
                                             psizes[k].ctypes.data, None))
        total = int(offs[k][n])
        assert arch[k][:total].tobytes() == serial[i % 3], f"step {i}"
        return k

    def dec(fut, i):
        k = fut.result()
        st = np.zeros(n, dtype=np.uint32)
        _ffi.check(L.cct_decode_batch(arch[k].ctypes.data, offs[k].ctypes.data, n, bs, magic, d_back[k].ptr, 1, n * W * W, st.ctypes.data))
        assert np.array_equal(d_back[k].download(np.uint16, n * W * W).reshape(n, W, W), batches[i % 3]), f"step {i}"

    try:
        _ffi.check(L.cct_set_option(b"decode_yields", scheduling))
        _ffi.check(L.cct_set_option(b"queue_ahead", scheduling))
        with ThreadPoolExecutor(2) as pe, ThreadPoolExecutor(1) as pd:
            decs, prev = [], None
            for i in range(12):
                while len(decs) >= NSET:
                    decs.pop(0).result()
                e = pe.submit(enc, i)
                decs.append(pd.submit(dec, e, i))
                if prev is not None:
                    prev.result()
                prev = e
            for d in decs:
                d.result()
    finally:
        _ffi.check(L.cct_set_option(b"decode_yields", 1))
        _ffi.check(L.cct_set_option(b"queue_ahead", 1))
        for b in d_imgs + d_back:
            b.free()
        for p_ in pins:
            p_.free()
