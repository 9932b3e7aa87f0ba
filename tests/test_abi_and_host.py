"""CPU-side checks of the product: the C-ABI library loads and exports every symbol declared in
include/compact_hip.h, host-only entry points work (traversal table, sizes, header parsing), and
compute entry points fail loudly without a GPU (no CPU fallback)."""
import ctypes as C
import json
import os
import re
import sys

import numpy as np
import pytest

import golden_inputs as gi

ROOT = gi.ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "compact_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cct_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from cct_hip import _ffi
    L = _ffi.lib()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/compact_hip.h but not exported"
    assert sorted(_ffi.exported_symbols()) == names
    assert L.cct_version() == 1


def test_curve_table_matches_reference_known_answers(manifest):
    from codec.curve import GeneralizedHilbertCurve
    for key, ka in manifest["curves"].items():
        w, h = map(int, key.split("x"))
        t = np.asarray(GeneralizedHilbertCurve(w, h, get_index=True).generate_all(), dtype="<i4")
        if "list" in ka:
            assert t.tolist() == ka["list"]
        else:
            assert gi.sha1(t.tobytes()) == ka["sha1"], key
    assert GeneralizedHilbertCurve(4, 4).generate_all()[:3] == [(0, 0), (0, 1), (1, 1)]


def test_sizes_and_header():
    from cct_hip import _ffi
    L = _ffi.lib()
    s = L.cct_payload_stride(512, 512, 16)
    assert s % 256 == 0 and s >= 2 * 512 * 512 + 16384 // 2 + 1
    assert L.cct_file_bound(512, 512, 16) > s
    with open(os.path.join(gi.GOLDEN, "slice0671.cct"), "rb") as f:
        blob = f.read()
    hdr = _ffi.Header()
    assert L.cct_read_header(blob, len(blob), b"pact", C.byref(hdr)) == 0
    assert (hdr.width, hdr.height, hdr.channels, hdr.bytes_per_channel) == (512, 512, 1, 2)
    assert (hdr.fractal, hdr.segmentation, hdr.deflate) == (1, 1, 1)
    assert L.cct_read_header(b"nope" + blob[4:], len(blob), b"pact", C.byref(hdr)) == _ffi.E_MAGIC


def test_config_is_the_reference_default():
    from cct_hip import default_config, codec_params
    cfg = default_config()
    assert cfg["magic"] == "pact" and cfg["extension"] == "cct" and cfg["block_size"] == 16
    flags, bs, eof, magic, ch, bpc = codec_params(cfg, np.uint16)
    assert (flags, bs, eof, magic, ch, bpc) == (7, 16, 59, b"pact", 1, 2)
    assert codec_params(cfg, np.int16)[0] == 15


def test_no_cpu_fallback_without_gpu():
    """On a machine without a usable gfx950 device every compute call must raise."""
    from cct_hip import _ffi
    import cct_hip
    if _ffi.lib().cct_init(-1) == 0:
        pytest.skip("a GPU is present")
    img = np.zeros((16, 16), np.uint16)
    with pytest.raises(_ffi.DeviceError):
        cct_hip.encode_batch(img[None])
    from codec.core import Encoder
    with pytest.raises(_ffi.DeviceError):
        Encoder(cct_hip.default_config(), img).encode()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "2023-compact-image-compression_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, fn), errors="replace").read()
                # comments may cite oracle/ files; code must not import, include, link or load them
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{fn} imports the oracle"
                assert not re.search(r"#\s*include[^\n]*oracle", src), f"{fn} includes oracle code"
                assert "libcompact_oracle" not in src and "libdeflate_model" not in src, f"{fn} loads an oracle library"


def test_options_round_trip_without_a_device():
    """cct_set_option / cct_get_option are host state: they work before (and without) a device, clamp what they must and
    refuse what they do not know."""
    import ctypes as C
    from cct_hip import _ffi
    L = _ffi.lib()
    v = C.c_int(-1)
    for key, given, want in ((b"encode_slots", 2, 2), (b"encode_slots", 7, 2), (b"encode_slots", 0, 1), (b"decode_slots", 1, 1),
                             (b"decode_slots", 5, 2), (b"tile_path", 3, 3), (b"tile_path", 4, 4),
                             (b"tile_path", 9, 1), (b"device_inflate", 5, 1), (b"deflate_graph", 0, 0), (b"deflate_graph", 1, 1),
                             (b"deflate_compact_records", 0, 0), (b"deflate_compact_records", 3, 1),
                             (b"decode_yields", 0, 0), (b"decode_yields", 2, 1), (b"queue_ahead", 0, 0), (b"queue_ahead", 1, 1),
                             (b"deflate_fork", 0, 0), (b"deflate_fork", 1, 1)):
        assert L.cct_set_option(key, given) == 0
        assert L.cct_get_option(key, C.byref(v)) == 0 and v.value == want, (key, given, v.value)
    assert L.cct_set_option(b"encode_slots", 1) == 0 and L.cct_set_option(b"tile_path", 1) == 0
    assert L.cct_set_option(b"deflate_ways", 2) != 0      # removed in round 2
    assert L.cct_set_option(b"no_such_option", 1) != 0
    assert L.cct_get_option(b"no_such_option", C.byref(v)) != 0


def test_shutdown_twice_and_from_threads():
    """cct_shutdown replaces the slot objects; the slot mutexes it holds meanwhile live outside them (round 2 destroyed
    them while locked).  Twice in a row, and racing threads that poll options, must leave a usable library."""
    import threading
    import ctypes as C
    from cct_hip import _ffi
    L = _ffi.lib()
    assert L.cct_shutdown() == 0 and L.cct_shutdown() == 0
    stop = threading.Event()
    def poll():
        v = C.c_int(0)
        while not stop.is_set():
            assert L.cct_get_option(b"decode_slots", C.byref(v)) == 0
    ts = [threading.Thread(target=poll) for _ in range(3)]
    for t in ts:
        t.start()
    for _ in range(50):
        assert L.cct_shutdown() == 0
    stop.set()
    for t in ts:
        t.join()
    assert L.cct_version() == 1


def test_bench_traffic_figure_is_tied_to_the_kernel_source(tmp_path, monkeypatch):
    """bench.py prints the PMC traffic of the transform+pack stage only while the SHA-1 stored with it is the SHA-1 of
    encode_stream.hip; and the summary committed under profiles/ belongs to the source committed next to it."""
    import hashlib
    import json
    sys.path.insert(0, ROOT)
    import bench
    with open(bench.PIPE_SRC, "rb") as f:
        sha = hashlib.sha1(f.read()).hexdigest()
    with open(bench.PMC_JSON) as f:
        pmc = json.load(f)
    assert pmc["source_sha1"] == sha, "encode_stream.hip changed after the PMC passes: re-run tools/gpu_round_end.sh + collect_profiles.py"
    traffic, src = bench.pmc_traffic()
    assert traffic == pmc["traffic_bytes_per_launch"] and traffic > pmc["algorithmic_read_bytes"] and "r03_pmc_encode.json" in src
    stale = tmp_path / "pmc.json"
    stale.write_text(json.dumps(dict(pmc, source_sha1="0" * 40)))
    monkeypatch.setattr(bench, "PMC_JSON", str(stale))
    traffic, src = bench.pmc_traffic()
    assert traffic is None and src.startswith("stale")
    monkeypatch.setattr(bench, "PMC_JSON", str(tmp_path / "missing.json"))
    assert bench.pmc_traffic() == (None, None)


def test_bench_workloads_and_sharding_follow_baseline_json():
    """The bench configurations are BASELINE.json's configs[1], [3], [4]; the corpus of configs[2] shards as DESIGN 8 says."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    from cct_hip.parallel import shard_range
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    assert len(base["configs"]) >= 5
    assert bench.WORKLOADS == {2: (512, 256), 4: (1024, 512), 5: (512, 256)}
    sizes = [shard_range(3954, r, 8) for r in range(8)]
    assert [b - a for a, b in sizes] == [495, 495, 494, 494, 494, 494, 494, 494]
    assert sizes[0][0] == 0 and sizes[-1][1] == 3954 and all(sizes[i][1] == sizes[i + 1][0] for i in range(7))
