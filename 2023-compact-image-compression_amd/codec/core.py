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

    @property
    def MAGIC(self):
        return _magic_int(self.config)

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
        self.fulls = []  # the reference collects full-token positions for debugging only

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
