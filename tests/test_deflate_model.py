"""Pins oracle/deflate_model.c -- the data-parallel restatement of zlib 1.2.11 deflate(level 9)
that the HIP DEFLATE kernels implement -- to the system libz (the library behind the reference's
zlib.compress(data, level=9), core.py:340): byte-identical streams."""
import ctypes as C
import os
import subprocess
import zlib

import numpy as np
import pytest

import golden_inputs as gi
from oracle import oracle

ODIR = os.path.join(gi.ROOT, "oracle")


@pytest.fixture(scope="module")
def model():
    subprocess.check_call(["make", "-s", "-C", ODIR, "libdeflate_model.so"])
    L = C.CDLL(os.path.join(ODIR, "libdeflate_model.so"))
    L.cct_model_deflate9.restype = C.c_size_t
    L.cct_model_deflate9.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]

    def run(b):
        out = np.empty(len(b) * 2 + 1024, dtype=np.uint8)
        n = L.cct_model_deflate9(b, len(b), out.ctypes.data)
        return out[:n].tobytes()
    return run


def test_zlib_version_is_the_pinned_one():
    assert zlib.ZLIB_RUNTIME_VERSION == "1.2.11"


@pytest.mark.parametrize("name,data", [
    ("empty", b""), ("one", b"a"), ("two", b"ab"), ("three", b"abc"), ("abc_rep", b"abcabcabcabc" * 10),
    ("zeros_1000", bytes(1000)), ("zeros_100k", bytes(100000)), ("run258", b"x" * 258), ("run259", b"y" * 259),
])
def test_small_known_inputs(model, name, data):
    assert model(data) == zlib.compress(data, 9)


@pytest.mark.parametrize("seed,alphabet,n", [(0, 256, 5000), (1, 4, 70000), (2, 16, 120000), (3, 256, 70000),
                                             (4, 2, 40000), (5, 3, 65274), (6, 3, 65275), (7, 3, 65536),
                                             (8, 3, 65800), (9, 3, 98043), (10, 64, 33000)])
def test_random_inputs_incl_stored_blocks_and_window_slides(model, seed, alphabet, n):
    rng = np.random.default_rng(seed)
    data = rng.integers(0, alphabet, n, dtype=np.uint8).tobytes()
    assert model(data) == zlib.compress(data, 9)


@pytest.mark.parametrize("name", ["crop128_f1s1d0", "noise64_nodeflate", "q4_block0", "int16_signed"])
def test_golden_payloads(model, name):
    with open(os.path.join(gi.GOLDEN, name + ".cct"), "rb") as f:
        payload = f.read()[13:]
    assert model(payload) == zlib.compress(payload, 9)


def test_real_slice_payload_reproduces_reference_file(model):
    """Token payload of slice 0671 -> exactly the DEFLATE body of the reference's testing.cct."""
    payload = oracle.encode(gi.load_slice("slice0671"), deflate=False)[13:]
    with open(os.path.join(gi.GOLDEN, "slice0671.cct"), "rb") as f:
        ref = f.read()
    assert model(payload) == ref[13:]
