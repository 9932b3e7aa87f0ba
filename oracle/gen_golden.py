#!/usr/bin/env python3
"""Generate tests/golden/ fixtures by running the REFERENCE codec (this container only).

Run:  python3 -B oracle/gen_golden.py
The reference (/root/reference, read-only, Python) is imported as-is; only DATA is
written to tests/golden/: input rasters recovered from PNG files the reference
ships, the reference's outputs (bytes or SHA-1 + length) and token statistics.
No reference source text is copied.  The GPU box never sees /root/reference; it
only sees these fixtures.
"""
import json
import os
import shutil
import sys
import time
import zlib

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_inputs as gi  # noqa: E402

sys.path.insert(0, os.path.join(REF, "src"))
import warnings  # noqa: E402

warnings.simplefilter("ignore")
from codec.core import Encoder, Decoder  # noqa: E402  (the reference)
from codec.curve import GeneralizedHilbertCurve  # noqa: E402
from codec.cluster import BlockPartitioner  # noqa: E402

GOLD = gi.GOLDEN


def base_config():
    cfg = json.load(open(os.path.join(REF, "src", "config.json")))
    cfg["verbose"] = False
    return cfg


def recover_slices():
    """SURVEY 8c vectors 1 and 2: 16-bit PNGs written as value<<4 -> raster = png >> 4."""
    for name, rel in (("slice0671", "data/working/decoded-testing.png"),
                      ("slice3706", "results/snapshots/0-input.png")):
        png = np.array(Image.open(os.path.join(REF, rel)))
        assert png.dtype == np.uint16 or png.dtype == np.int32, png.dtype
        ras = (png.astype(np.uint32) >> 4).astype("<u2")
        assert ras.shape == (512, 512)
        with open(os.path.join(GOLD, name + ".u16.zz"), "wb") as f:
            f.write(zlib.compress(ras.tobytes(), 9))
    # the reference's own committed encoder output for slice 0671
    shutil.copyfile(os.path.join(REF, "data/working/testing.cct"), os.path.join(GOLD, "slice0671.cct"))


def run_case(name, inp, cfg_over=None, store=False, expect_roundtrip=True):
    cfg = base_config()
    over = cfg_over or {}
    for k, v in over.items():
        if k == "block_size":
            cfg["block_size"] = v
        elif k in ("fractal", "segmentation"):
            cfg["encoder"]["transforms"][k] = v
        elif k == "deflate":
            cfg["encoder"]["deflate_compression"] = v
        else:
            raise KeyError(k)
    img = gi.build_input(inp)
    case = {"name": name, "input": inp, "input_sha1": gi.sha1(img.tobytes()),
            "shape": list(img.shape), "dtype": str(img.dtype), "config": over}
    t0 = time.time()
    try:
        enc = Encoder(cfg, img, None)
        out = enc.encode()
    except Exception as e:  # noqa: BLE001
        case["encode_raises"] = type(e).__name__
        print(f"{name:28s} encode raises {type(e).__name__}")
        return case
    case["len"] = len(out)
    case["sha1"] = gi.sha1(out)
    case["tokens"] = {"short": int(enc.info["delta"]), "full": int(enc.info["full"])}
    if cfg["encoder"]["transforms"]["segmentation"]:
        # block_partition() is deterministic; re-run it for the jump table
        _, jumps = enc.partition.block_partition()
        case["tokens"]["jump"] = len(jumps)
        case["jumps_sha1"] = gi.sha1(np.array(sorted(jumps.items()), dtype=np.int32).tobytes())
    if store:
        fn = name + ".cct"
        with open(os.path.join(GOLD, fn), "wb") as f:
            f.write(out)
        case["file"] = fn
    try:
        dec = Decoder(cfg, out, None).decode()
        case["roundtrip"] = bool(dec == img.tobytes())
        case["decoded_sha1"] = gi.sha1(dec)
    except Exception as e:  # noqa: BLE001
        case["decode_raises"] = type(e).__name__
        case["roundtrip"] = False
    if expect_roundtrip:
        assert case["roundtrip"], name
    print(f"{name:28s} len {len(out):7d} {case['tokens']} rt={case['roundtrip']} "
          f"{case.get('decode_raises', '')} ({time.time() - t0:.1f}s)")
    return case


def main():
    os.makedirs(GOLD, exist_ok=True)
    recover_slices()
    cases = []
    s0671 = {"kind": "slice", "name": "slice0671"}
    s3706 = {"kind": "slice", "name": "slice3706"}
    crop = {"kind": "slice", "name": "slice0671", "crop": [192, 320, 192, 320]}

    # real slices (SURVEY 8c vectors 1, 2)
    c = run_case("slice0671", s0671)
    ref_cct = open(os.path.join(REF, "data/working/testing.cct"), "rb").read()
    assert c["sha1"] == gi.sha1(ref_cct), "reference encoder no longer reproduces testing.cct"
    c["file"] = "slice0671.cct"
    cases.append(c)
    cases.append(run_case("slice0671_nodeflate", s0671, {"deflate": False}))
    cases.append(run_case("slice3706", s3706, store=True))
    cases.append(run_case("slice3706_nodeflate", s3706, {"deflate": False}))

    # flag matrix and block sizes on the 128x128 crop
    for fr in (1, 0):
        for sg in (1, 0):
            for df in (1, 0):
                cases.append(run_case(f"crop128_f{fr}s{sg}d{df}", crop,
                                      {"fractal": bool(fr), "segmentation": bool(sg), "deflate": bool(df)},
                                      store=True))
    for bs in (4, 8, 32, 64):
        cases.append(run_case(f"crop128_bs{bs}", crop, {"block_size": bs}, store=True))

    # shapes
    for shp in ((32, 64), (64, 32), (20, 20), (4, 4), (16, 16), (48, 80), (96, 96), (160, 96)):
        inp = {"kind": "uniform", "seed": 0, "lo": 900, "hi": 1100, "shape": list(shp)}
        cases.append(run_case(f"uniform_{shp[0]}x{shp[1]}", inp, store=True))
    cases.append(run_case("zeros_16x16", {"kind": "zeros", "shape": [16, 16]}, store=True))
    cases.append(run_case("bad_shape_10x10", {"kind": "uniform", "seed": 0, "lo": 900, "hi": 1100, "shape": [10, 10]}))

    # heavy meshing: 11-bit noise (every block difficult), both with and without deflate
    noise = {"kind": "uniform", "seed": 1, "lo": 0, "hi": 2048, "shape": [64, 64]}
    cases.append(run_case("noise64", noise, store=True))
    cases.append(run_case("noise64_nodeflate", noise, {"deflate": False}, store=True))
    cases.append(run_case("noise128x64", {"kind": "uniform", "seed": 2, "lo": 0, "hi": 2048, "shape": [128, 64]}, store=True))
    # Q4: difficult block 0 always meshes with block 1
    cases.append(run_case("q4_block0", {"kind": "q4"}, {"deflate": False}, store=True))
    cases.append(run_case("q4_block0_bs4", {"kind": "q4"}, {"deflate": False, "block_size": 4}, store=True))
    # dtype handling
    cases.append(run_case("int16_texture", {"kind": "int16_texture", "seed": 5, "shape": [64, 64]}, store=True))
    cases.append(run_case("int16_signed", {"kind": "int16_signed", "seed": 6, "shape": [32, 32]},
                          {"deflate": False}, store=True, expect_roundtrip=False))
    # Q7: out-of-range delta -> reference decoder raises
    cases.append(run_case("q7_spike", {"kind": "spike", "shape": [16, 16], "at": [5, 5], "value": 4000},
                          {"deflate": False}, store=True, expect_roundtrip=False))

    # phantoms (bench inputs): small ones stored, 512/768/1024 as SHA-1
    cases.append(run_case("phantom256_s7", {"kind": "phantom", "seed": 7, "n": 256}, store=True))
    cases.append(run_case("phantom256_s7_nodeflate", {"kind": "phantom", "seed": 7, "n": 256}, {"deflate": False}))
    cases.append(run_case("phantom_rect_192x320", {"kind": "phantom_rect", "seed": 8, "n": 320, "shape": [192, 320]}, store=True, expect_roundtrip=False))
    cases.append(run_case("phantom_rect_320x192", {"kind": "phantom_rect", "seed": 8, "n": 320, "shape": [320, 192]}, store=True, expect_roundtrip=False))
    for seed in (0, 1, 2, 255):
        cases.append(run_case(f"phantom512_s{seed}", {"kind": "phantom", "seed": seed, "n": 512}))
    cases.append(run_case("phantom512_s0_nodeflate", {"kind": "phantom", "seed": 0, "n": 512}, {"deflate": False}))
    cases.append(run_case("phantom768_s5", {"kind": "phantom", "seed": 5, "n": 768}))
    cases.append(run_case("phantom1024_s3", {"kind": "phantom", "seed": 3, "n": 1024}))

    # traversal known answers
    curves = {}
    for (w, h) in ((4, 4), (8, 4), (4, 8), (5, 3), (6, 6)):
        curves[f"{w}x{h}"] = {"list": [int(v) for v in GeneralizedHilbertCurve(w, h, get_index=True).generate_all()]}
    for (w, h) in ((16, 16), (64, 64), (20, 20), (32, 64), (64, 32), (48, 80), (96, 96), (160, 96), (192, 320), (320, 192),
                   (128, 128), (256, 256), (512, 512), (768, 768), (1024, 1024), (7, 13), (100, 37)):
        t = np.array(GeneralizedHilbertCurve(w, h, get_index=True).generate_all(), dtype="<i4")
        curves[f"{w}x{h}"] = {"sha1": gi.sha1(t.tobytes()), "head": [int(v) for v in t[:8]], "last": int(t[-1])}

    # BlockPartitioner demo (cluster.py __main__ data): known answer for Q4 at block_size 4
    data = [0, 100, 200, 300] + [50, 150, 250, 350]
    order = [0, 2, 4, 6, 1, 3, 5, 7]
    d2 = [data[i] for i in order]
    p = BlockPartitioner(d2, order=order, block_size=4)
    p.set_delta_changes_array()
    p.initial_partition()
    po, bj = p.block_partition()
    part = {"data": d2, "order": order, "block_size": 4,
            "pixel_order": [int(v) for v in po], "jumps": {str(k): int(v) for k, v in bj.items()}}

    # header error path
    bad = b"nope" + open(os.path.join(GOLD, "zeros_16x16.cct"), "rb").read()[4:]
    try:
        Decoder(base_config(), bad, None).decode()
        bad_exc = None
    except Exception as e:  # noqa: BLE001
        bad_exc = type(e).__name__

    manifest = {"generator": "oracle/gen_golden.py", "reference": "taaha-khan/2023-CompaCT-Image-Compression @ /root/reference",
                "numpy": np.__version__, "zlib": zlib.ZLIB_RUNTIME_VERSION,
                "cases": cases, "curves": curves, "partition_demo": part, "bad_magic_raises": bad_exc}
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
