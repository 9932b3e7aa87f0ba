"""CPU: the PackBits oracle against the reference: tests/golden/packbits.json holds 301 strings with what the reference's
own PackBits gives for them (oracle/gen_packbits_golden.py imports it in the build container), next to the known answers of
SURVEY Appendix C (the first one is what src/codec/packbits.py:166-177 prints)."""
import base64
import json
import os

from oracle import packbits_oracle as po


def golden_cases():
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "packbits.json")) as f:
        g = json.load(f)
    for c in g["cases"]:
        yield base64.b64decode(c["data"]), base64.b64decode(c["plain"]), base64.b64decode(c["delta"])


def test_oracle_equals_the_reference_on_the_fixture():
    n = 0
    for data, plain, delta in golden_cases():
        assert bytes(po.encode(list(data), False)) == plain
        assert bytes(po.encode(list(data), True)) == delta
        assert bytes(bytearray(po.decode(plain, False))) == data
        assert bytes(bytearray(po.decode(delta, True))) == data
        n += 1
    assert n >= 300

KATS = [
    ([3, 255, 3, 255, 3, 255, 3, 255, 20, 255], False, [9, 3, 255, 3, 255, 3, 255, 3, 255, 20, 255]),
    ([1, 2, 3, 3, 3, 3, 4, 2, 1, 0], False, [1, 1, 2, 253, 3, 3, 4, 2, 1, 0]),
    ([1, 2, 3, 3, 3, 3, 4, 2, 1, 0], True, [254, 1, 254, 0, 1, 1, 254, 255, 255]),
    ([7] * 300 + [1, 2], False, [130, 7, 130, 7, 211, 7, 1, 1, 2]),
]


def test_known_answers():
    for data, delta, want in KATS:
        enc = po.encode(data, delta)
        assert list(enc) == want
        assert list(po.decode(enc, delta)) == data


def test_edges():
    assert list(po.encode([], False)) == []
    assert list(po.encode([9], False)) == [0, 9]
    assert list(po.encode([5] * 128, False)) == [129, 5]           # one run may reach 128 (count bumped at the end)
    assert list(po.encode([5] * 129, False)) == [130, 5, 255, 5]
    assert list(po.encode([5] * 255, False)) == [130, 5, 129, 5]
    lit = [i % 2 for i in range(128)]
    assert list(po.encode(lit, False)) == [127] + lit              # a literal chunk that ends the data may reach 128
    lit = [i % 2 for i in range(129)]
    assert list(po.encode(lit, False)) == [126] + lit[:127] + [1] + lit[127:]
    for data in ([1, 1], [1, 2], [1, 1, 2, 2], [1, 2, 2], [2, 2, 1], [0] * 1000 + [1] * 3 + list(range(200))):
        assert list(po.decode(po.encode(data))) == data
        assert list(po.decode(po.encode(data, True), True)) == data
