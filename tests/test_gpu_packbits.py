"""GPU: the PackBits kernels (csrc/packbits_kernels.hip, through cct_packbits_*_batch and the codec.packbits mirror) against
the oracle restatement of src/codec/packbits.py: known answers, segment-length edge cases around 127/128, random strings."""
import numpy as np
import pytest

from oracle import packbits_oracle as po
from test_packbits_oracle import KATS, golden_cases

pytestmark = pytest.mark.gpu


def test_known_answers_and_mirror_class():
    from codec.packbits import PackBits
    for data, delta, want in KATS:
        pk = PackBits(delta)
        enc = pk.encode(data)
        assert list(enc) == want
        assert list(PackBits(delta).decode(enc)) == data
    assert PackBits().encode([]) == []
    assert PackBits().encode([9]) == b"\x00\x09"


def test_kernels_equal_the_reference_on_the_fixture():
    """tests/golden/packbits.json: 301 strings encoded by the reference's own PackBits in both delta modes."""
    from codec import packbits
    cases = list(golden_cases())
    blobs = [c[0] for c in cases]
    for delta, col in ((False, 1), (True, 2)):
        enc = packbits.encode_batch(blobs, delta)
        assert [bytes(e) for e in enc] == [c[col] for c in cases]
        dec = packbits.decode_batch([c[col] for c in cases], delta, max_out=4096)
        assert [bytes(d) for d in dec] == blobs


def test_batches_vs_oracle():
    from codec import packbits
    rng = np.random.default_rng(7)
    blobs = []
    for L in (2, 3, 126, 127, 128, 129, 130, 253, 254, 255, 256, 257, 300, 381, 382, 1000):
        blobs.append(bytes([5]) * L)                                      # one run
        blobs.append(bytes((i * 7 + i // 3) % 256 for i in range(L)))     # mostly literals
        blobs.append(bytes([1]) + bytes([5]) * L + bytes([2, 3]))         # run inside
        blobs.append(bytes((i % 2) for i in range(L)))                    # literals ending the data
    for _ in range(200):
        n = int(rng.integers(2, 2000))
        k = int(rng.integers(1, 5))
        a = rng.integers(0, k, size=n).astype(np.uint8)                   # small alphabets: many short runs
        if rng.random() < 0.5:
            a = np.repeat(a, rng.integers(1, 200, size=n))[:n]
        blobs.append(a.tobytes())
    for delta in (False, True):
        enc = packbits.encode_batch(blobs, delta)
        for b, e in zip(blobs, enc):
            assert bytes(e) == bytes(po.encode(list(b), delta))
        dec = packbits.decode_batch([bytes(e) for e in enc], delta, max_out=4096)
        for b, d in zip(blobs, dec):
            assert bytes(d) == b


def test_decode_edge_packets():
    from codec import packbits
    # header 128 is skipped; a literal packet cut short by the end of the data comes up short, as a Python slice does
    streams = [bytes([128, 0, 9, 128]), bytes([3, 1, 2])]
    out = packbits.decode_batch(streams, False, max_out=64)
    assert [bytes(o) for o in out] == [bytes(po.decode(s)) for s in streams]
