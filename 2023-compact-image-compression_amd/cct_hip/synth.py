"""Synthetic 12-bit CT slice generator.

The reference corpus (3954 QIN LUNG CT slices, scripts/evaluate.py:41) is not
available offline, so the bench and the parity tests use seeded phantoms that
exercise the same code paths as real slices: smooth tissue (short tokens),
edges (full tokens) and textured bone (difficult blocks -> mesh jumps).

Format constraint: the .cct format only carries
traversal deltas in [-2047, 2048], and the mesh step may interleave any two
blocks up to 63 apart, so adjacency arguments are not enough.  The phantom
therefore keeps every sample in [0, 2047] (bone at 1500-1750 plus texture,
instead of Appendix D's 1900-2200): then no delta can leave the range, whatever
order the encoder picks, and every slice round-trips.

`depth12=True` gives slices that use the 12-bit container like the real corpus does (real slices reach 2271,
SURVEY 8c): bone at 1900-2150 as Appendix D says, samples up to 4095 allowed -- and then every sample is
lowered to at most (smallest sample within 64 pixels) + 2047.  A mesh partner lies at most 63 blocks ahead of
its leader, which is less than 64 pixels away in either axis on the tiled traversals, so whatever order the
encoder picks no delta leaves the format's range (the Q7 assertion of Appendix D, by construction).
"""
import numpy as np


def _window_min(a, radius):
    """Minimum over the (2 radius + 1)^2 window around every sample (edges replicate), by doubling shifts."""
    def axis_min(x, axis):
        n = x.shape[axis]
        def shifted(y, k):  # y[i + k] with the last sample repeated
            idx = np.minimum(np.arange(n) + k, n - 1)
            return np.take(y, idx, axis=axis)
        m, w = x, 1                      # m[i] = min x[i .. i + w - 1]
        while 2 * w <= 2 * radius + 1:
            m = np.minimum(m, shifted(m, w))
            w *= 2
        rest = 2 * radius + 1 - w
        if rest:
            m = np.minimum(m, shifted(m, rest))
        idx = np.maximum(np.arange(n) - radius, 0)   # centre the window: out[i] = m[i - radius]
        lead = np.take(m, idx, axis=axis)
        # the first `radius` outputs must not look left of sample 0: m[0] covers [0, 2 radius], a superset -- take the
        # running minimum of the prefix instead
        pre = np.minimum.accumulate(np.take(x, np.arange(min(n, 2 * radius + 1)), axis=axis), axis=axis)
        k = min(n, radius)
        sl_out = [slice(None)] * x.ndim
        sl_out[axis] = slice(0, k)
        lead[tuple(sl_out)] = np.take(pre, np.minimum(np.arange(k) + radius, pre.shape[axis] - 1), axis=axis)
        return lead
    return axis_min(axis_min(a, 0), 1)


def ct_phantom(seed, n=512, depth12=False):
    """Return one n x n uint16 slice, deterministic in (seed, n, depth12)."""
    rng = np.random.default_rng(seed)
    h = n / 2.0
    c = (n - 1) / 2.0
    yy, xx = np.mgrid[0:n, 0:n].astype(np.float32)
    u = (xx - c) / h
    v = (yy - c) / h

    img = np.zeros((n, n), dtype=np.float32)
    fov = (u * u + v * v) <= 0.97 ** 2
    img[fov] = 24.0

    # body ellipse
    a = rng.uniform(0.70, 0.78)
    b = rng.uniform(0.50, 0.58)
    ox = rng.uniform(-0.02, 0.02)
    oy = rng.uniform(-0.02, 0.04)
    ub = (u - ox) / a
    vb = (v - oy) / b
    rb = np.sqrt(ub * ub + vb * vb)
    body = rb <= 1.0
    ph1, ph2 = rng.uniform(0, 2 * np.pi, 2)
    f1, f2 = rng.uniform(2.0, 5.0, 2)
    img[body] = (1030.0 + 30.0 * np.sin(f1 * u + ph1) * np.cos(f2 * v + ph2))[body]

    # lungs: kept inside rb < 0.70 so a soft-tissue layer separates them from the ribs
    for sgn in (-1.0, 1.0):
        lx = ox + sgn * a * rng.uniform(0.40, 0.46)
        ly = oy - b * rng.uniform(0.02, 0.10)
        la = a * rng.uniform(0.26, 0.30)
        lb = b * rng.uniform(0.46, 0.54)
        lung = (((u - lx) / la) ** 2 + ((v - ly) / lb) ** 2 <= 1.0) & (rb < 0.70)
        img[lung] = rng.uniform(160.0, 200.0)

    # spoked rib shell + spine (bone)
    theta = np.arctan2(vb, ub)
    k = int(rng.integers(9, 14))
    spokes = np.cos(k * theta + rng.uniform(0, 2 * np.pi)) > 0.15
    ribs = (rb > 0.78) & (rb < 0.90) & spokes
    sx = ox + rng.uniform(-0.02, 0.02)
    sy = oy + b * 0.55
    spine = ((u - sx) / 0.085) ** 2 + ((v - sy) / 0.10) ** 2 <= 1.0
    spine &= rb < 0.92
    bone = ribs | spine
    img[bone] = rng.uniform(1900.0, 2150.0) if depth12 else rng.uniform(1500.0, 1750.0)

    # thin table arc below the body
    rt = np.sqrt(u * u + (v - 1.75) ** 2)
    table = (np.abs(rt - 1.05) < 0.006) & fov & ~body
    img[table] = 700.0

    # acquisition noise inside the FOV
    noise = rng.normal(0.0, 14.0, size=(n, n)).astype(np.float32)
    img += np.where(fov, noise, 0.0)

    # bone texture on the 1-px dilated bone mask: makes blocks "difficult"
    dil = bone.copy()
    dil[1:, :] |= bone[:-1, :]
    dil[:-1, :] |= bone[1:, :]
    dil[:, 1:] |= bone[:, :-1]
    dil[:, :-1] |= bone[:, 1:]
    dil &= rb < 0.95
    sigma = rng.uniform(100.0, 120.0)
    tex = rng.normal(0.0, sigma, size=(n, n)).astype(np.float32)
    tex = np.clip(tex, -3.0 * sigma, 3.0 * sigma)
    img += np.where(dil, tex, 0.0)

    img[~fov] = 0.0
    if not depth12:
        return np.clip(np.rint(img), 0, 2047).astype(np.uint16)
    out = np.clip(np.rint(img), 0, 4095).astype(np.int32)
    return np.minimum(out, _window_min(out, 64) + 2047).astype(np.uint16)


def ct_batch(seeds, n=512, depth12=False):
    """Stack ct_phantom(seed, n) for every seed: (len(seeds), n, n) uint16."""
    out = np.empty((len(seeds), n, n), dtype=np.uint16)
    for i, s in enumerate(seeds):
        out[i] = ct_phantom(int(s), n, depth12)
    return out
