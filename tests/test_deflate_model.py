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


# ---------------------------------------------------------------------------------------------------------
# run rule: the device evaluates positions that start a run of three equal bytes from the list of run ends
# (dfl_match_run_kernel) instead of walking the hash chain.  cct_model_check_run_rule restates that rule on
# the CPU and compares it with the chain walk (= deflate.c longest_match) on every such position.
@pytest.fixture(scope="module")
def run_rule():
    subprocess.check_call(["make", "-s", "-C", ODIR, "libdeflate_model.so"])
    L = C.CDLL(os.path.join(ODIR, "libdeflate_model.so"))
    L.cct_model_check_run_rule.restype = C.c_int64
    L.cct_model_check_run_rule.argtypes = [C.c_char_p, C.c_size_t]
    return lambda b: L.cct_model_check_run_rule(b, len(b))


def _short_runs(rng, n):
    parts, tot = [], 0
    lo, hi = [(3, 6), (3, 40), (200, 300), (3, 600)][int(rng.integers(0, 4))]
    while tot < n:
        k = int(rng.integers(lo, hi))
        sep = bytes(rng.integers(1, 3 + int(rng.integers(0, 3)), int(rng.integers(1, 3)), dtype=np.uint8))
        parts += [bytes(k), sep]
        tot += k + len(sep)
    return b"".join(parts)[:n]


def _colliding(rng, n):
    # (32,0,0) and (1,32,0) hash to the bucket of (0,0,0)
    parts, tot = [], 0
    while tot < n:
        c = rng.random()
        if c < 0.4:
            parts.append(bytes(int(rng.integers(3, 50))))
        elif c < 0.6:
            parts.append(bytes([32, 0, 0]))
        elif c < 0.8:
            parts.append(bytes([1, 32, 0]))
        else:
            parts.append(bytes(rng.integers(0, 40, int(rng.integers(1, 4)), dtype=np.uint8)))
        tot += len(parts[-1])
    return b"".join(parts)[:n]


def _far_runs(rng, n):
    # runs spaced around MAX_DIST (32506) and the window size
    parts, tot = [], 0
    while tot < n:
        k = int(rng.integers(3, 300))
        gap = max(1, int(rng.choice([32506 - k, 32505 - k, 32507 - k, 32768 - k, 100, 1000, 16000, 32506, 32503])))
        parts += [bytes([int(rng.integers(0, 2))]) * k, bytes(rng.integers(2, 256, gap, dtype=np.uint8))]
        tot += k + gap
    return b"".join(parts)[:n]


@pytest.mark.parametrize("data", [bytes(10), b"\1" + bytes(12), bytes(4) + b"\1" + bytes(5) + b"\2", bytes(258), bytes(259),
                                  bytes(3), bytes(70000), b"ab" + b"c" * 300 + b"ab" + b"c" * 300 + b"d" + b"c" * 299 + b"d"])
def test_run_rule_small_cases(run_rule, data):
    assert run_rule(data) == -1


@pytest.mark.parametrize("gen", [_short_runs, _colliding, _far_runs])
@pytest.mark.parametrize("seed", range(6))
def test_run_rule_adversarial(run_rule, gen, seed):
    rng = np.random.default_rng(1000 + seed)
    assert run_rule(gen(rng, int(rng.integers(1000, 200000)))) == -1


@pytest.mark.parametrize("name", ["slice0671", "slice3706"])
def test_run_rule_real_payloads(run_rule, name):
    payload = oracle.encode(gi.load_slice(name), deflate=False)[13:]
    assert run_rule(payload) == -1
