"""Mirror of the reference's codec.packbits (src/codec/packbits.py): PackBits run-length coding of a byte string with an
optional byte-delta pre-transform.  Dead code in the reference (nothing imports it; the run tag of the .cct format is
commented out, core.py:299-310), kept here for completeness of the "RLE pack" utility.  The work happens on
the device (csrc/packbits_kernels.hip) through cct_packbits_encode_batch / cct_packbits_decode_batch.

One difference on purpose: the reference object keeps its state machine between calls (encode() never resets
self.result / self.pos), so a second encode() on the same object returns garbage; here every call is what a fresh
PackBits(...) object would return.
"""
import ctypes as C

import numpy as np

from cct_hip import _ffi


def _batch(fn_name, blobs, delta, out_stride):
    L = _ffi.lib()
    n = len(blobs)
    offs = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum([len(b) for b in blobs], out=offs[1:])
    flat = np.frombuffer(b"".join(bytes(b) for b in blobs) or b"\0", dtype=np.uint8)
    out = np.zeros((max(n, 1), out_stride), dtype=np.uint8)
    sizes = np.zeros(max(n, 1), dtype=np.uint32)
    if fn_name == "encode":
        _ffi.check(L.cct_packbits_encode_batch(flat.ctypes.data, offs.ctypes.data, n, int(bool(delta)), out.ctypes.data,
                                               out_stride, sizes.ctypes.data))
    else:
        status = np.zeros(max(n, 1), dtype=np.uint32)
        _ffi.check(L.cct_packbits_decode_batch(flat.ctypes.data, offs.ctypes.data, n, int(bool(delta)), out.ctypes.data,
                                               out_stride, sizes.ctypes.data, status.ctypes.data))
    return [bytearray(out[i, : sizes[i]].tobytes()) for i in range(n)]


def encode_batch(blobs, apply_delta_transform=False):
    """PackBits of every byte string of `blobs` (list of bytes-like), one wave each."""
    L = _ffi.lib()
    longest = max((len(b) for b in blobs), default=0)
    return _batch("encode", blobs, apply_delta_transform, max(16, int(L.cct_packbits_bound(longest))))


def decode_batch(blobs, apply_delta_transform=False, max_out=None):
    if max_out is None:  # a run packet expands 2 bytes to at most 128
        max_out = max((64 * len(b) + 128 for b in blobs), default=16)
    return _batch("decode", blobs, apply_delta_transform, int(max_out))


class PackBits:

    MAX_LENGTH = 127  # packbits.py:30

    def __init__(self, apply_delta_transform=False):
        self.apply_delta_transform = apply_delta_transform

    def delta_transform(self, data):
        """packbits.py:43-51"""
        a = np.frombuffer(bytes(bytearray(data)), dtype=np.uint8).astype(np.int32)
        return [int(a[0])] + ((a[1:] - a[:-1]) % 256).tolist() if a.size else []

    def revert_delta_transform(self, data):
        """packbits.py:53-63"""
        return (np.cumsum(np.frombuffer(bytes(bytearray(data)), dtype=np.uint8).astype(np.int64)) % 256).tolist()

    def encode(self, data):
        """packbits.py:74-129"""
        if len(data) == 0:
            return data
        if len(data) == 1:
            return b"\x00" + bytes(bytearray(data))
        return encode_batch([bytes(bytearray(data))], self.apply_delta_transform)[0]

    def decode(self, data):
        """packbits.py:131-163 (a bytearray without, a list with the delta transform, as the reference returns them)"""
        out = decode_batch([bytes(bytearray(data))], self.apply_delta_transform)[0]
        return list(out) if self.apply_delta_transform else out
