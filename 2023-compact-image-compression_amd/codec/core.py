"""Drop-in for the reference's codec.core (src/codec/core.py): same classes, same call
signatures, same return values and exceptions -- the work is done by hand-written HIP
kernels on an MI355X through cct_hip's ctypes binding.  There is no CPU path here.

    Encoder(config, image, out_path=None).encode()      -> bytes   (core.py:170-365)
    Decoder(config, file_bytes, out_path=None).decode() -> bytes | ndarray (core.py:367-543)

Callers that work unchanged: src/main.py:52-65, scripts/demo.py:57-75, scripts/evaluate.py:87-88.
"""
import json
import os
import sys
from collections import defaultdict

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from cct_hip import _ffi, batch  # noqa: E402


class Utils:
    """Token tag / mask constants of the byte format (core.py:40-50)."""
    TAG_DELTA, TAG_JUMP, TAG_RUN, TAG_FULL = 0x00, 0x80, 0xC0, 0xE0
    MASK_DELTA, MASK_JUMP, MASK_RUN, MASK_FULL = 0x80, 0xC0, 0xE0, 0xF0


def unsign(x, n_bits):
    """core.py:52-54"""
    return x % (1 << n_bits)


def signed(x, n_bits):
    """core.py:56-60: values strictly above half the range are negative."""
    top = 1 << n_bits
    return x - top if 2 * x > top else x


def rescale(value):
    """12-bit sample -> 16-bit preview sample (core.py:62-64)."""
    return value << 4


def unscale(value):
    """core.py:66-67"""
    return value >> 4


def _magic_int(config):
    """The MAGIC property of both classes (core.py:188-191, 380-383)."""
    return int.from_bytes(bytes(map(ord, reversed(config["magic"]))), sys.byteorder)


class ByteWriter:
    """Header and payload kept apart, as core.py:95-131 does; filled from the device results."""

    def __init__(self):
        self.header = bytearray()
        self.data = bytearray()

    def set_data(self, data):
        self.data = data

    def output_header(self):
        return bytes(self.header)

    def output_data(self):
        return bytes(self.data)

    def output(self):
        return bytes(self.header) + bytes(self.data)


def final_pixel_order(order, roles, block_size):
    """PIXEL_ORDER and BLOCK_JUMPS of BlockPartitioner.block_partition (cluster.py:49-199) from the block roles the
    device computed: role 0 = block emitted alone, 1..63 = leader meshed with block + role (its pixels interleaved,
    cluster.py:173-174), 0xFF = partner (emitted with its leader)."""
    blocks = np.asarray(order, dtype=np.int32).reshape(-1, block_size)
    out, jumps = [], {}
    for b, r in enumerate(np.asarray(roles, dtype=np.uint8).tolist()):
        if r == 0:
            out.append(blocks[b])
        elif r != 0xFF:
            jumps[b] = b + r
            pair = np.empty(2 * block_size, dtype=np.int32)
            pair[0::2] = blocks[b]
            pair[1::2] = blocks[b + r]
            out.append(pair)
    return (np.concatenate(out) if out else np.zeros(0, np.int32)), jumps


class _Partition:
    """What callers read from Encoder.partition (core.py:258): block_partition() -> (PIXEL_ORDER, BLOCK_JUMPS).  The
    partition itself is computed on the device (cct_encode_payload_dev's role table)."""

    def __init__(self, image, config, order):
        self._image, self._config, self._order = image, config, order  # order: the traversal, or a callable that makes it
        self.block_size = int(config["block_size"])
        self._result = None

    def block_partition(self):
        if self._result is None:
            roles = batch.partition_roles(self._image, self._config)
            order = self._order() if callable(self._order) else self._order
            self._result = final_pixel_order(order, roles, self.block_size)
        return self._result


class Encoder:

    def __init__(self, config, image, out_path=None):
        self.config = config
        self.image = image
        self.image_bytes = image.tobytes()
        self.width, self.height = image.shape  # NB: "width" is shape[0] (core.py:179)
        self.size = self.width * self.height
        self.out_path = out_path
        self.stats = [["Section", "Size (KB)", "Ratio (x)"]]
        self.info = defaultdict(int)
        self.writer = ByteWriter()  # core.py:182
        self._curve = None

    @property
    def MAGIC(self):
        return _magic_int(self.config)

    @property
    def curve(self):
        """core.py:234-235: the traversal object (only with the fractal transform on); built when somebody asks."""
        if not self.config["encoder"]["transforms"]["fractal"]:
            raise AttributeError("curve")
        if self._curve is None:
            from .curve import GeneralizedHilbertCurve
            self._curve = GeneralizedHilbertCurve(self.width, self.height, get_index=True)
        return self._curve

    def _traversal(self):
        """core.py:234-239: the pixel order."""
        if self.config["encoder"]["transforms"]["fractal"]:
            return np.asarray(self.curve.generate_all(), dtype=np.int32)
        return np.arange(self.size, dtype=np.int32)

    def encode(self):
        cfg = self.config
        enc = cfg["encoder"]
        self.raw_size = self.size * enc["channels"] * enc["bytes_per_channel"]
        if self.raw_size > 400_000_000_000:  # core.py:218-219
            raise MemoryError(f"Maximum byte count exceeded: {self.raw_size}")
        if not enc["transforms"]["delta"]:  # core.py:221-222
            raise NotImplementedError("Non-delta encoding not supported")
        if enc["transforms"]["zipper"]:  # core.py:224-225
            raise NotImplementedError("Zipper transform not supported or encouraged")
        if enc["channels"] * enc["bytes_per_channel"] != 2 or self.image.dtype.itemsize != 2:
            # the reference reads 2 bytes per pixel whatever the dtype and silently encodes
            # garbage for anything else; refuse instead of reproducing garbage
            raise TypeError("HIP codec path needs single-channel 2-byte pixels (uint16 / int16)")

        files, info = batch.encode_batch(np.ascontiguousarray(self.image)[None, :, :], cfg, return_info=True)
        output, st = files[0], info[0]
        self.writer.header = bytearray(output[:13])   # core.py:193-210
        self.writer.set_data(output[13:])             # core.py:337-345: the payload, DEFLATEd when configured
        if enc["transforms"]["segmentation"]:         # core.py:258-268 (evaluated lazily: one more device call when asked for)
            self.partition = _Partition(np.ascontiguousarray(self.image), cfg, self._traversal)
        self.info["delta"] = st["n_short"]
        self.info["full"] = st["n_full"]
        self.block_jumps_count = st["n_jump"]
        self.q7_violation = st["q7"]  # stream not decodable by any decoder (format limit, Q7)

        # per-stage size table (core.py:227, 332-355)
        header_len = 13
        self.stats.append(["Original", self.raw_size / 1000, 1.0])
        qoi_len = header_len + st["payload_len"]
        self.stats.append(["QOI", qoi_len / 1000, self.raw_size / qoi_len])
        if enc["deflate_compression"]:
            self.stats.append(["DEFLATE", len(output) / 1000, st["payload_len"] / (len(output) - header_len)])
        self.stats.append(["Final", len(output) / 1000, self.raw_size / len(output)])

        if cfg["verbose"]:  # core.py:325-326, 357-359
            print("\n" + json.dumps(self.info))
            try:
                from tabulate import tabulate
                print(tabulate(self.stats, headers="firstrow", tablefmt="simple_outline"))
            except ImportError:
                for row in self.stats:
                    print(row)
        if self.out_path is not None:  # core.py:361-363
            with open(self.out_path, "wb") as fout:
                fout.write(output)
        return output


class Decoder:

    def __init__(self, config, file_bytes, out_path=None):
        self.config = config
        self.file_bytes = file_bytes
        self.out_path = out_path
        self._fulls = None
        self._pixels = None

    @property
    def fulls(self):
        """core.py:508: raster indices of the pixels that arrived as full (two-byte) tokens, in stream order.  The
        reference collects them while parsing; here they are recomputed on demand from the decoded slice (the order
        the pixels were written in is the encoder's partition of that same slice).  Equal to what the reference's
        parser records for every stream an encoder can emit; for a hand-made stream (reserved tag bytes, a full token
        where a short one would do) it is what an encoder WOULD have written for the decoded pixels, not what was parsed."""
        if self._fulls is None:
            if self._pixels is None:
                return []
            cfg = self.config
            order = np.arange(self.size, dtype=np.int32)
            if self.fractal_transform:
                from .curve import GeneralizedHilbertCurve
                order = np.asarray(GeneralizedHilbertCurve(self.width, self.height, get_index=True).generate_all(), np.int32)
            # jump handling is unconditional in the decoder: a file written without segmentation has no jumps
            if self.segmentation_transform:
                roles = batch.partition_roles(self._pixels, cfg)
                order, _ = final_pixel_order(order, roles, int(cfg["block_size"]))
            vals = self._pixels.reshape(-1)[order].astype(np.int64)
            delta = np.diff(np.concatenate([[0], vals]))
            self._fulls = order[(delta < -63) | (delta > 64)].tolist()
        return self._fulls

    @property
    def MAGIC(self):
        return _magic_int(self.config)

    def read_header(self):
        """core.py:385-402"""
        fb = self.file_bytes
        if len(fb) < 13:
            raise _ffi.CorruptStreamError("file shorter than the 13-byte header")
        if int.from_bytes(fb[0:4], "big") != self.MAGIC:
            raise ValueError("Image does not contain valid header")
        self.width = (fb[4] << 8) | fb[5]
        self.height = (fb[6] << 8) | fb[7]
        self.channels = fb[8]
        self.bytes_per_channel = fb[9]
        self.fractal_transform = bool(fb[10])
        self.segmentation_transform = bool(fb[11])
        self.deflate_compression = bool(fb[12])

    def decode(self):
        self.read_header()
        self.size = self.width * self.height
        self.total_size = self.size * self.channels * self.bytes_per_channel
        pixels = batch.decode_batch([bytes(self.file_bytes)], self.config)[0]  # (width, height) uint16
        self._pixels = pixels

        if self.out_path is not None:  # core.py:522-540: 16-bit PNG preview, value << 4
            preview = (pixels.astype(np.uint32) << 4).astype(np.uint16)
            _write_png16(self.out_path, preview)
            return pixels
        return pixels.tobytes()  # core.py:543


def _write_png16(path, arr):
    """imageio.imwrite(path, uint16 array) of the reference (core.py:537-538); falls back to
    Pillow when imageio is not installed."""
    try:
        import imageio
        imageio.imwrite(path, arr)
        return
    except ImportError:
        pass
    from PIL import Image
    Image.fromarray(arr.astype(np.uint16)).save(path)
