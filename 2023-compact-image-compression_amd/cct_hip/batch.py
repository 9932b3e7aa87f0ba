"""Batch API over the C ABI: a per-slice Python call cannot feed a TB/s device, so callers
with many slices (scripts/evaluate.py's corpus loop, bench.py) hand whole batches over.
Single-slice use goes through codec.core.Encoder / Decoder, which call into here with n=1.
"""
import ctypes as C
import json
import os

import numpy as np

from . import _ffi

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def default_config():
    """The reference's src/config.json, shipped verbatim as data next to the package."""
    with open(os.path.join(_PKG_ROOT, "config.json")) as f:
        return json.load(f)


def magic_bytes(config):
    """4 header bytes of config['magic'] (core.py:188-196: big-endian write of the low 32 bits)."""
    b = bytes(map(ord, config["magic"]))
    return b[-4:].rjust(4, b"\0")


def codec_params(config, dtype=None):
    """config dict -> (flags, block_size, eof, magic, channels, bytes_per_channel)."""
    enc = config["encoder"]
    tr = enc["transforms"]
    flags = 0
    if tr["fractal"]:
        flags |= _ffi.FLAG_FRACTAL
    if tr["segmentation"]:
        flags |= _ffi.FLAG_SEGMENTATION
    if enc["deflate_compression"]:
        flags |= _ffi.FLAG_DEFLATE
    if dtype is not None and np.dtype(dtype).kind == "i":
        flags |= _ffi.FLAG_SIGNED_SEG
    eof = enc.get("end_of_file")
    return (flags, int(config["block_size"]), -1 if eof is None else int(eof) % 256, magic_bytes(config),
            int(enc["channels"]), int(enc["bytes_per_channel"]))


def device_info():
    L = _ffi.lib()
    name = C.create_string_buffer(256)
    cus = C.c_int(0)
    hbm = C.c_uint64(0)
    _ffi.check(L.cct_device_info(name, 256, C.byref(cus), C.byref(hbm)))
    return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": hbm.value}


class DeviceBuffer:
    """Plain HBM allocation owned by the library's device (hipMalloc behind cct_dev_alloc)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p(0)
        _ffi.check(_ffi.lib().cct_dev_alloc(C.byref(p), self.nbytes))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, arr):
        arr = np.ascontiguousarray(arr)
        buf = cls(arr.nbytes)
        buf.upload(arr)
        return buf

    def upload(self, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        _ffi.check(_ffi.lib().cct_h2d(self.ptr + offset, arr.ctypes.data, arr.nbytes))

    def download(self, dtype, count, offset=0):
        out = np.empty(count, dtype=dtype)
        assert offset + out.nbytes <= self.nbytes
        _ffi.check(_ffi.lib().cct_d2h(out.ctypes.data, self.ptr + offset, out.nbytes))
        return out

    def zero(self):
        _ffi.check(_ffi.lib().cct_dev_memset(self.ptr, 0, self.nbytes))

    def free(self):
        if self.ptr:
            _ffi.check(_ffi.lib().cct_dev_free(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                _ffi.lib().cct_dev_free(self.ptr)
                self.ptr = None
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


class PinnedArray:
    """Page-locked host bytes (cct_host_alloc) seen as a numpy uint8 array: archives kept here move to and from
    the device without a staging copy."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p(0)
        _ffi.check(_ffi.lib().cct_host_alloc(C.byref(p), self.nbytes))
        self.ptr = p.value
        self.array = np.ctypeslib.as_array((C.c_uint8 * self.nbytes).from_address(self.ptr))

    def free(self):
        if self.ptr:
            self.array = None
            _ffi.check(_ffi.lib().cct_host_free(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                _ffi.lib().cct_host_free(self.ptr)
                self.ptr = None
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


class Event:
    """HIP event on the library's stream (bench.py times kernels with these)."""

    def __init__(self):
        p = C.c_void_p(0)
        _ffi.check(_ffi.lib().cct_event_create(C.byref(p)))
        self.ptr = p.value

    def record(self):
        _ffi.check(_ffi.lib().cct_event_record(self.ptr))

    def elapsed_ms_since(self, start):
        ms = C.c_float(0)
        _ffi.check(_ffi.lib().cct_event_elapsed_ms(start.ptr, self.ptr, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                _ffi.lib().cct_event_destroy(self.ptr)
        except Exception:  # noqa: BLE001
            pass


def payload_stride(width, height, block_size):
    return _ffi.lib().cct_payload_stride(width, height, block_size)


def encode_payload_dev(d_images, n, width, height, config_or_params, d_payload, d_sizes, d_status,
                       d_stats=None, d_roles=None, dtype=np.uint16):
    """Stage (i) on device-resident slices; all arguments are DeviceBuffer objects."""
    params = codec_params(config_or_params, dtype) if isinstance(config_or_params, dict) else config_or_params
    flags, bs, eof = params[0], params[1], params[2]
    stride = payload_stride(width, height, bs)
    _ffi.check(_ffi.lib().cct_encode_payload_dev(
        d_images.ptr, n, width, height, bs, flags, eof, d_payload.ptr, stride, d_sizes.ptr, d_status.ptr,
        d_stats.ptr if d_stats is not None else None, d_roles.ptr if d_roles is not None else None))
    return stride


def decode_payload_dev(d_payload, stride, d_sizes, n, width, height, block_size, fractal, d_images, d_status):
    _ffi.check(_ffi.lib().cct_decode_payload_dev(d_payload.ptr, stride, d_sizes.ptr, n, width, height, block_size,
                                                  int(bool(fractal)), d_images.ptr, d_status.ptr))


def partition_roles(image, config=None):
    """Block roles of one slice (the partition of cluster.py:49-199 as the device computes it): uint8[NB], 0 = emitted
    alone, 1..63 = leader of a meshed pair (BLOCK_JUMPS[b] - b), 0xFF = partner."""
    config = config or default_config()
    image = np.ascontiguousarray(image)
    w, h = image.shape
    bs = int(config["block_size"])
    nb = w * h // bs
    d_img = DeviceBuffer.from_numpy(image)
    d_pay, d_sz, d_st, d_roles = DeviceBuffer(payload_stride(w, h, bs)), DeviceBuffer(4), DeviceBuffer(4), DeviceBuffer(max(nb, 1))
    encode_payload_dev(d_img, 1, w, h, codec_params(config, image.dtype), d_pay, d_sz, d_st, None, d_roles)
    return d_roles.download(np.uint8, nb)


def encode_batch(images, config=None, return_info=False):
    """images: (n, W, H) array of a 2-byte dtype (or a DeviceBuffer + shape via encode_batch_dev).
    Returns a list of n `bytes`, each exactly what Encoder(config, images[i]).encode() returns."""
    config = config or default_config()
    images = np.asarray(images)
    if images.ndim != 3:
        raise ValueError("encode_batch expects an array of shape (n, width, height)")
    if images.dtype.itemsize != 2:
        raise TypeError(f"2-byte pixels required (uint16/int16), got {images.dtype}")
    images = np.ascontiguousarray(images)
    n, w, h = images.shape
    return _encode(images.ctypes.data, 0, n, w, h, config, images.dtype, return_info, keep=images)


def encode_batch_dev(d_images, n, width, height, config=None, dtype=np.uint16, return_info=False):
    """Same as encode_batch for slices already resident in HBM (DeviceBuffer)."""
    return _encode(d_images.ptr, 1, n, width, height, config or default_config(), dtype, return_info)


def _encode(ptr, on_device, n, w, h, config, dtype, return_info, keep=None):
    L = _ffi.lib()
    flags, bs, eof, magic, ch, bpc = codec_params(config, dtype)
    if (w * h) % bs != 0:  # numpy's reshape message, core.py:245
        raise ValueError(f"cannot reshape array of size {w * h} into shape ({(w * h) // bs},{bs})")
    defl = bool(flags & _ffi.FLAG_DEFLATE)
    out_stride = L.cct_file_bound(w, h, bs) if defl else 13 + L.cct_payload_stride(w, h, bs)
    out = np.empty((max(n, 1), out_stride), dtype=np.uint8)
    sizes = np.zeros(max(n, 1), dtype=np.uint32)
    status = np.zeros(max(n, 1), dtype=np.uint32)
    psizes = np.zeros(max(n, 1), dtype=np.uint32)
    stats = (_ffi.SliceStats * max(n, 1))()
    dev_defl = C.c_int(0)
    L.cct_get_option(b"device_deflate", C.byref(dev_defl))
    if defl and dev_defl.value and n > 0:
        # archive layout: the files come back to back (one compact copy instead of n * out_stride strided bytes)
        arch = out.reshape(-1)
        offs = np.zeros(n + 1, dtype=np.uint64)
        _ffi.check(L.cct_encode_batch_packed(ptr, on_device, n, w, h, bs, flags, eof, magic, ch, bpc,
                                             arch.ctypes.data, arch.size, offs.ctypes.data, sizes.ctypes.data,
                                             status.ctypes.data, psizes.ctypes.data, C.cast(stats, C.c_void_p)))
        files = [arch[int(offs[i]): int(offs[i + 1])].tobytes() for i in range(n)]
    else:
        _ffi.check(L.cct_encode_batch(ptr, on_device, n, w, h, bs, flags, eof, magic, ch, bpc,
                                      out.ctypes.data, out_stride, sizes.ctypes.data, status.ctypes.data,
                                      psizes.ctypes.data, C.cast(stats, C.c_void_p)))
        files = [out[i, : sizes[i]].tobytes() for i in range(n)]
    if return_info:
        info = [{"payload_len": int(psizes[i]), "n_short": stats[i].n_short, "n_full": stats[i].n_full,
                 "n_jump": stats[i].n_jump, "n_difficult": stats[i].n_difficult,
                 "q7": bool(status[i] & _ffi.ST_Q7)} for i in range(n)]
        return files, info
    return files


def decode_batch(files, config=None, out_dev=None):
    """files: list of .cct byte strings of identical shape/flags.
    Returns an (n, W, H) uint16 array (or fills the DeviceBuffer `out_dev` and returns the shape)."""
    config = config or default_config()
    L = _ffi.lib()
    n = len(files)
    magic = magic_bytes(config)
    bs = int(config["block_size"])
    if n == 0:
        return np.zeros((0, 0, 0), dtype=np.uint16)
    hdr = _ffi.Header()
    _ffi.check(L.cct_read_header(files[0], len(files[0]), magic, C.byref(hdr)))
    w, h = hdr.width, hdr.height
    if (w * h) % bs != 0 or w * h == 0:  # core.py:429
        raise ValueError(f"cannot reshape array of size {w * h} into shape ({(w * h) // bs},{bs})")
    blob = b"".join(files)
    offs = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum([len(f) for f in files], out=offs[1:])
    status = np.zeros(n, dtype=np.uint32)
    if out_dev is not None:
        rc = L.cct_decode_batch(blob, offs.ctypes.data, n, bs, magic, out_dev.ptr, 1, out_dev.nbytes // 2,
                                status.ctypes.data)
        _ffi.check(rc)
        return (n, w, h)
    out = np.empty((n, w, h), dtype=np.uint16)
    rc = L.cct_decode_batch(blob, offs.ctypes.data, n, bs, magic, out.ctypes.data, 0, out.size, status.ctypes.data)
    _ffi.check(rc)
    return out


def zlib_compress_batch(blobs):
    """DEFLATE stage alone on the device: [bytes] -> [zlib streams], each byte-identical to
    zlib.compress(blob, level=9) (what the reference calls at core.py:340)."""
    import zlib
    L = _ffi.lib()
    n = len(blobs)
    if n == 0:
        return []
    offs = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum([len(b) for b in blobs], out=offs[1:])
    data = b"".join(blobs) or b"\0"
    longest = max(len(b) for b in blobs)
    in_stride = (longest + 16 + 255) & ~255
    out_stride = (len(zlib.compress(b"", 0)) + in_stride + (in_stride >> 12) + (in_stride >> 14) + (in_stride >> 25) + 13 + 128 + 63) & ~63
    out = np.empty((n, out_stride), dtype=np.uint8)
    sizes = np.zeros(n, dtype=np.uint32)
    _ffi.check(L.cct_zlib_compress_batch(data, offs.ctypes.data, n, out.ctypes.data, out_stride, sizes.ctypes.data))
    return [out[i, : sizes[i]].tobytes() for i in range(n)]


def zlib_decompress_batch(streams, max_out, raise_errors=True):
    """INFLATE stage alone on the device: [zlib streams] -> [bytes], what zlib.decompress returns for each
    (the reference calls it at core.py:421).  max_out bounds the inflated size of one stream.  With
    raise_errors=False returns (outputs, status) where status[i] is 0, CCT_E_ZLIB (2) or CCT_E_CAP (6)."""
    L = _ffi.lib()
    n = len(streams)
    if n == 0:
        return [] if raise_errors else ([], np.zeros(0, dtype=np.uint32))
    offs = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum([len(b) for b in streams], out=offs[1:])
    data = b"".join(streams) or b"\0"
    out_stride = (int(max_out) + 15 + 16) & ~15
    out = np.empty((n, out_stride), dtype=np.uint8)
    sizes = np.zeros(n, dtype=np.uint32)
    status = np.zeros(n, dtype=np.uint32)
    rc = L.cct_zlib_decompress_batch(data, offs.ctypes.data, n, out.ctypes.data, out_stride, sizes.ctypes.data,
                                     status.ctypes.data)
    if raise_errors:
        _ffi.check(rc)
    outs = [out[i, : sizes[i]].tobytes() if status[i] == 0 else None for i in range(n)]
    return outs if raise_errors else (outs, status)
