"""ctypes front-end of oracle/libcompact_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module (as the checker / the timed CPU baseline).  The product package never does.
Parity status: pinned by tests/test_oracle_golden.py against fixtures generated
from the reference (oracle/gen_golden.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcompact_oracle.so")

E_MAGIC, E_ZLIB, E_OVERFLOW, E_STREAM, E_SHAPE, E_CAP, E_NOMEM = 1, 2, 3, 4, 5, 6, 7


class Stats(C.Structure):
    _fields_ = [("n_short", C.c_uint32), ("n_full", C.c_uint32), ("n_jump", C.c_uint32),
                ("n_difficult", C.c_uint32), ("payload_len", C.c_uint32),
                ("q7_violations", C.c_uint32)]


def build(force=False):
    """Compile the C restatement (gcc, -lz) next to its source."""
    src = os.path.join(_HERE, "compact_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libcompact_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.cct_oracle_curve.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.cct_oracle_partition.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p]
        L.cct_oracle_bound.restype = C.c_size_t
        L.cct_oracle_bound.argtypes = [C.c_int64, C.c_int]
        L.cct_oracle_encode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int,
                                        C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                        C.POINTER(Stats)]
        L.cct_oracle_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_char_p, C.c_void_p,
                                        C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.cct_oracle_read_header.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p] + [C.POINTER(C.c_int)] * 7
        _lib = L
    return _lib


class OracleError(Exception):
    def __init__(self, code):
        super().__init__(f"oracle error {code}")
        self.code = code


def curve(width, height):
    """Traversal order (raster indices) of GeneralizedHilbertCurve(width, height, get_index=True)."""
    out = np.empty(width * height, dtype=np.int32)
    rc = lib().cct_oracle_curve(width, height, out.ctypes.data)
    if rc:
        raise OracleError(rc)
    return out


def partition(data, order, block_size):
    """BlockPartitioner(data, order, block_size) -> (PIXEL_ORDER, {blockA: blockB})."""
    D = np.ascontiguousarray(data, dtype=np.int32)
    O = np.ascontiguousarray(order, dtype=np.int32)
    n = D.size
    out = np.empty(n, dtype=np.int32)
    jumps = np.empty(max(n // block_size, 1), dtype=np.int32)
    nd = C.c_uint32(0)
    rc = lib().cct_oracle_partition(D.ctypes.data, O.ctypes.data, n, block_size,
                                    out.ctypes.data, jumps.ctypes.data, C.byref(nd))
    if rc:
        raise OracleError(rc)
    return out, {int(i): int(j) for i, j in enumerate(jumps[: n // block_size]) if j >= 0}


def encode(image, block_size=16, fractal=True, segmentation=True, deflate=True, eof=59,
           magic=b"pact", channels=1, bytes_per_channel=2, return_stats=False):
    """Encoder(config, image).encode() -> bytes (whole .cct file)."""
    img = np.ascontiguousarray(image)
    if img.dtype.itemsize != 2 or img.ndim != 2:
        raise TypeError("oracle.encode needs a 2-D array of a 2-byte dtype")
    signed_seg = 1 if img.dtype.kind == "i" else 0
    w, h = img.shape
    cap = lib().cct_oracle_bound(w * h, block_size)
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    st = Stats()
    rc = lib().cct_oracle_encode(img.ctypes.data, w, h, block_size, int(fractal), int(segmentation),
                                 int(deflate), -1 if eof is None else int(eof), signed_seg,
                                 magic, channels, bytes_per_channel,
                                 out.ctypes.data, cap, C.byref(n), C.byref(st))
    if rc:
        raise OracleError(rc)
    data = out[: n.value].tobytes()
    return (data, st) if return_stats else data


def decode(file_bytes, block_size=16, magic=b"pact"):
    """Decoder(config, file_bytes).decode() -> bytes (uint16 raster, native order)."""
    hdr = [C.c_int(0) for _ in range(7)]
    rc = lib().cct_oracle_read_header(file_bytes, len(file_bytes), magic, *[C.byref(x) for x in hdr])
    if rc:
        raise OracleError(rc)
    w, h = hdr[0].value, hdr[1].value
    out = np.empty(max(w * h, 1), dtype=np.uint16)
    wo, ho = C.c_int(0), C.c_int(0)
    rc = lib().cct_oracle_decode(file_bytes, len(file_bytes), block_size, magic, out.ctypes.data,
                                 out.size, C.byref(wo), C.byref(ho))
    if rc:
        raise OracleError(rc)
    return out[: w * h].tobytes()
