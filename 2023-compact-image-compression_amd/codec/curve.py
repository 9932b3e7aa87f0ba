"""Traversal mirror of the reference's codec.curve (src/codec/curve.py).

The generalized Hilbert curve is produced by the library's host generator (cct_curve_table,
csrc/curve.cpp) -- the same table the kernels index in HBM -- not by Python generators.
"""
import numpy as np

from cct_hip import _ffi


class GeneralizedHilbertCurve:

    def __init__(self, width, height, get_index=False):
        self.width = width
        self.height = height
        self.get_index = get_index
        self.curve = []

    def _table(self):
        out = np.empty(self.width * self.height, dtype=np.int32)
        _ffi.check(_ffi.lib().cct_curve_table(self.width, self.height, out.ctypes.data))
        return out

    def generate_all(self):
        """curve.py:45-49: list of raster indices (get_index=True) or (row, col) tuples."""
        if not self.curve:
            t = self._table()
            if self.get_index:
                self.curve = t.tolist()
            else:
                self.curve = [(int(i) // self.width, int(i) % self.width) for i in t]
        return self.curve

    def generator(self):
        """curve.py:51-59"""
        yield from self.generate_all()

    def idx(self, p):
        """curve.py:71-74"""
        r, c = p
        return r * self.width + c

    def pos(self, p):
        """curve.py:76-78"""
        return (p // self.width, p % self.width)
