"""Slice loaders shared by tools/demo.py and tools/evaluate.py: the reference callers read DICOM files through pydicom
(scripts/demo.py:51, scripts/evaluate.py:114); pydicom is not part of this environment, so the counterparts take the
pixel array itself: .npy, raw little-endian 16-bit (.u16, .raw; square or WxH given), zlib'd raw (.u16.zz, the form
tests/golden keeps the two real slices in) or a 16-bit PNG preview written as value << 4 (lib/png.py:25-31)."""
import os
import zlib

import numpy as np

EXTENSIONS = (".npy", ".u16", ".raw", ".u16.zz", ".png")


def _square(a):
    n = int(round(a.size ** 0.5))
    if n * n != a.size:
        raise ValueError(f"{a.size} samples are not a square slice; give --shape W,H")
    return a.reshape(n, n)


def load_slice(path, shape=None):
    low = path.lower()
    if low.endswith(".npy"):
        a = np.load(path, allow_pickle=False)
    elif low.endswith(".u16.zz"):
        with open(path, "rb") as f:
            a = np.frombuffer(zlib.decompress(f.read()), dtype="<u2")
    elif low.endswith((".u16", ".raw")):
        a = np.fromfile(path, dtype="<u2")
    elif low.endswith(".png"):
        from PIL import Image
        a = np.array(Image.open(path))
        if a.dtype != np.uint16:
            raise ValueError(f"{path}: expected a 16-bit PNG")
        a = a >> 4  # png_to_array, lib/png.py:33-41
    else:
        raise ValueError(f"{path}: unsupported input (one of {', '.join(EXTENSIONS)})")
    if a.ndim == 1:
        a = a.reshape(shape) if shape else _square(a)
    if a.ndim != 2 or a.dtype.itemsize != 2:
        raise ValueError(f"{path}: need a 2-D array of 2-byte samples, got {a.dtype} {a.shape}")
    return np.ascontiguousarray(a)


def list_inputs(directory):
    out = []
    for root, _, files in os.walk(directory):
        for f in sorted(files):
            if f.lower().endswith(EXTENSIONS):
                out.append(os.path.join(root, f))
    return sorted(out)
