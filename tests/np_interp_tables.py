"""Independent NumPy restatement of cv2's fixed-point interpolation tables -- test infrastructure.

Written separately from both csrc/remap.hip and oracle/remap_ref.c (vectorised, float32 arrays, no shared text):
the remap parity tests compare kernels against the oracle, which proves the gather but not the table; this module is
the third opinion on the table itself (tests/test_boundary_cpu.py compares all three).  `fix_group_lo` is U15.
"""
import numpy as np


def lanczos4_phase_weights():
    """(32, 8) float32: taps -3..4 at phases t = 0/32 .. 31/32."""
    t = (np.arange(32, dtype=np.float32) * np.float32(1 / 32))[:, None]            # (32, 1) float32
    k = np.arange(8, dtype=np.float32)[None, :]
    d = (t + np.float32(3)) - k                                                       # float32, like `x + 3 - i`
    pi4 = np.float64(np.pi) * 0.25
    a = -(t.astype(np.float64) + 3) * pi4                                             # one angle per phase
    q = np.sqrt(0.5)
    rot = np.array([[1, 0], [-q, -q], [0, 1], [q, -q], [-1, 0], [q, q], [0, -1], [-q, q]], np.float64)
    num = rot[None, :, 0] * np.sin(a) + rot[None, :, 1] * np.cos(a)                   # (32, 8) float64
    y = -d.astype(np.float64) * pi4
    with np.errstate(divide="ignore", invalid="ignore"):
        w = (num / (y * y)).astype(np.float32)
    w = np.where(np.abs(d) < np.float32(1e-6), np.float32(1e30), w)
    total = np.zeros(32, np.float32)
    for i in range(8):                                                                # float32 running sum, in tap order
        total = total + w[:, i]
    return (w * (np.float32(1) / total)[:, None]).astype(np.float32)


def bilinear_phase_weights():
    t = np.arange(32, dtype=np.float32) * np.float32(1 / 32)
    return np.stack([np.float32(1) - t, t], axis=1).astype(np.float32)


def fixed_point_table(w1, fix_group_lo):
    """(1024, ks*ks) int16 from (32, ks) float32 phase weights: rint(wy*wx*32768) saturated, sums forced to 32768."""
    ks = w1.shape[1]
    prod = (w1[:, None, :, None] * w1[None, :, None, :]).astype(np.float32)          # (py, px, ky, kx)
    q = np.clip(np.rint(prod * np.float32(32768)), -32768, 32767).astype(np.int64).reshape(1024, ks * ks)
    excess = q.sum(1) - 32768
    lo = fix_group_lo
    group = [ky * ks + kx for ky in (lo, lo + 1) for kx in (lo, lo + 1)]              # scan order ky, kx
    for e in np.nonzero(excess)[0]:
        # the entry is corrected while the entries behind it are still zero; a window index beyond the entry reads
        # that zero region (only the 2x2 table does: its window is [1, 3)^2), beyond the whole table it is skipped
        vals = []
        for i in group:
            flat = e * ks * ks + i
            vals.append((i, None) if flat >= 1024 * ks * ks else (i, q[e, i] if i < ks * ks else 0))
        big = small = group[0]
        vb = vs = vals[0][1]
        for i, v in vals:
            if v is None:
                continue
            if v < vs:
                small, vs = i, v
            elif v > vb:
                big, vb = i, v
        at = big if excess[e] < 0 else small
        if at < ks * ks:
            q[e, at] -= excess[e]
        # (a correction that lands in the next entry's storage is overwritten when that entry is built)
    return q.astype(np.int16)


def lanczos4_itab(fix_group_lo=4):
    return fixed_point_table(lanczos4_phase_weights(), fix_group_lo)


def bilinear_itab():
    return fixed_point_table(bilinear_phase_weights(), 1)
