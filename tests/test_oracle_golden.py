"""Pins the CPU oracle (oracle/compact_oracle.c) to fixtures generated from the
reference codec (oracle/gen_golden.py) -- SURVEY.md section 8c."""
import json
import os

import numpy as np
import pytest

import golden_inputs as gi
from oracle import oracle

with open(os.path.join(gi.GOLDEN, "manifest.json")) as _f:
    _MAN = json.load(_f)
CASES = {c["name"]: c for c in _MAN["cases"]}
BIG = {"phantom768_s5", "phantom1024_s3"}


def _cfg(case):
    o = case["config"]
    return dict(block_size=o.get("block_size", 16), fractal=o.get("fractal", True),
                segmentation=o.get("segmentation", True), deflate=o.get("deflate", True))


@pytest.mark.parametrize("name", sorted(CASES))
def test_encode_matches_reference(name):
    case = CASES[name]
    img = gi.build_input(case["input"])
    assert gi.sha1(img.tobytes()) == case["input_sha1"], "input generator drifted"
    if "encode_raises" in case:
        with pytest.raises(oracle.OracleError) as e:
            oracle.encode(img, **_cfg(case))
        assert e.value.code == oracle.E_SHAPE  # reference: ValueError from reshape
        return
    out, st = oracle.encode(img, return_stats=True, **_cfg(case))
    assert len(out) == case["len"]
    assert gi.sha1(out) == case["sha1"]
    if "file" in case:
        with open(os.path.join(gi.GOLDEN, case["file"]), "rb") as f:
            assert out == f.read()
    assert st.n_short == case["tokens"]["short"]
    assert st.n_full == case["tokens"]["full"]
    if "jump" in case["tokens"]:
        assert st.n_jump == case["tokens"]["jump"]


@pytest.mark.parametrize("name", sorted(n for n, c in CASES.items() if "file" in c))
def test_decode_matches_reference(name):
    case = CASES[name]
    with open(os.path.join(gi.GOLDEN, case["file"]), "rb") as f:
        blob = f.read()
    bs = case["config"].get("block_size", 16)
    if "decode_raises" in case:
        with pytest.raises(oracle.OracleError) as e:
            oracle.decode(blob, block_size=bs)
        assert {"OverflowError": oracle.E_OVERFLOW}.get(case["decode_raises"], oracle.E_STREAM) == e.value.code
        return
    dec = oracle.decode(blob, block_size=bs)
    assert gi.sha1(dec) == case["decoded_sha1"]
    if case["roundtrip"]:
        assert dec == gi.build_input(case["input"]).tobytes()


def test_reference_golden_artifact_testing_cct():
    """data/working/testing.cct (copied as DATA) is reproduced byte for byte from the PNG-recovered slice."""
    img = gi.load_slice("slice0671")
    assert gi.sha1(img.tobytes()) == "bc26ff59b9950f2b9857856aec4561c351cf146f"
    with open(os.path.join(gi.GOLDEN, "slice0671.cct"), "rb") as f:
        ref = f.read()
    assert len(ref) == 207575 and gi.sha1(ref) == "f9f1df755e97372186f89c501bc79e517ff97d21"
    assert oracle.encode(img) == ref
    assert oracle.decode(ref) == img.tobytes()


@pytest.mark.parametrize("key", sorted(_MAN["curves"]))
def test_curve_known_answers(key):
    w, h = map(int, key.split("x"))
    ka = _MAN["curves"][key]
    t = oracle.curve(w, h)
    if "list" in ka:
        assert t.tolist() == ka["list"]
    else:
        assert gi.sha1(t.astype("<i4").tobytes()) == ka["sha1"]
        assert t[:8].tolist() == ka["head"] and int(t[-1]) == ka["last"]
    assert sorted(t.tolist()) == list(range(w * h))


def test_partition_demo_known_answer():
    d = _MAN["partition_demo"]
    order, jumps = oracle.partition(d["data"], d["order"], d["block_size"])
    assert order.tolist() == d["pixel_order"]
    assert {str(k): v for k, v in jumps.items()} == d["jumps"]


def test_bad_magic():
    assert _MAN["bad_magic_raises"] == "ValueError"
    with open(os.path.join(gi.GOLDEN, "zeros_16x16.cct"), "rb") as f:
        blob = b"nope" + f.read()[4:]
    with pytest.raises(oracle.OracleError) as e:
        oracle.decode(blob)
    assert e.value.code == oracle.E_MAGIC
