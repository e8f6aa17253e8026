"""Independent NumPy model of cv2.StereoSGBM in the *parallel* formulation the HIP kernels use.

Test infrastructure.  It restates SURVEY.md Appendix A in closed form -- box-summed cost volume,
every aggregation direction as an independent scan from a zero border state, order-free
right-view (disp2) construction by (min cost, then larger d) -- and is checked bit-for-bit against
the sequential C oracle (oracle/sgbm_ref.c), which keeps OpenCV's row-incremental structure.
Agreement of the two pins the reformulation the GPU path relies on.
"""
import numpy as np

MAX_COST = 32767


def normalise(p, width):
    minD = p.get("minDisparity", 0)
    D = p["numDisparities"]
    maxD = minD + D
    q = dict(
        minD=minD, maxD=maxD, D=D,
        uniq=p.get("uniquenessRatio", 0) if p.get("uniquenessRatio", 0) >= 0 else 10,
        d12=p.get("disp12MaxDiff", 0) if p.get("disp12MaxDiff", 0) > 0 else 1,
        P1=p.get("P1", 0) if p.get("P1", 0) > 0 else 2,
    )
    q["P2"] = max(p.get("P2", 0) if p.get("P2", 0) > 0 else 5, q["P1"] + 1)
    bs = p.get("blockSize", 3) if p.get("blockSize", 3) > 0 else 5
    q["SW2"] = bs // 2
    q["ftzero"] = max(p.get("preFilterCap", 0), 15) | 1
    q["minX1"] = max(maxD, 0)
    q["maxX1"] = width + min(minD, 0)
    q["width1"] = q["maxX1"] - q["minX1"]
    q["mode"] = p.get("mode", 0)
    q["speckleWindowSize"] = p.get("speckleWindowSize", 0)
    q["speckleRange"] = p.get("speckleRange", 0)
    return q


def planes(img, ftzero):
    """(2*cn, H, W) int32: cn clipped x-Sobel planes then cn raw planes; columns 0, W-1 = ftzero."""
    img = img[..., None] if img.ndim == 2 else img
    H, W, cn = img.shape
    I = img.astype(np.int32)
    up = np.concatenate([I[:1], I[:-1]], 0)
    dn = np.concatenate([I[1:], I[-1:]], 0)
    out = np.full((2 * cn, H, W), ftzero, np.int32)
    for c in range(cn):
        g = (I[:, 2:, c] - I[:, :-2, c]) * 2 + (up[:, 2:, c] - up[:, :-2, c]) + (dn[:, 2:, c] - dn[:, :-2, c])
        out[c, :, 1:-1] = np.clip(g, -ftzero, ftzero) + ftzero
        out[cn + c, :, 1:-1] = I[:, 1:-1, c]
    return out


def minmax_half(p):
    """per plane: min/max of {p, (p+left)//2, (p+right)//2}; at the image edge the missing side is p."""
    l = p.copy()
    l[..., 1:] = (p[..., 1:] + p[..., :-1]) // 2
    r = p.copy()
    r[..., :-1] = (p[..., :-1] + p[..., 1:]) // 2
    return np.minimum(np.minimum(l, r), p), np.maximum(np.maximum(l, r), p)


def pixel_cost(left, right, q):
    """BT cost summed over planes: (H, width1, D) int32."""
    cn = 1 if left.ndim == 2 else left.shape[2]
    pl, pr = planes(left, q["ftzero"]), planes(right, q["ftzero"])
    u0, u1 = minmax_half(pl)
    v0, v1 = minmax_half(pr)
    H = pl.shape[1]
    cost = np.zeros((H, q["width1"], q["D"]), np.int32)
    xs = np.arange(q["minX1"], q["maxX1"])
    for di in range(q["D"]):
        xr = xs - (di + q["minD"])
        for c in range(2 * cn):
            u, a0, a1 = pl[c][:, xs], u0[c][:, xs], u1[c][:, xs]
            v, b0, b1 = pr[c][:, xr], v0[c][:, xr], v1[c][:, xr]
            c0 = np.maximum(0, np.maximum(u - b1, b0 - u))
            c1 = np.maximum(0, np.maximum(v - a1, a0 - v))
            cost[:, :, di] += np.minimum(c0, c1) >> (0 if c < cn else 2)
    return cost


def box_cost(pix, q):
    """C = P2 + box sum with clamped (replicated) borders in cost coordinates, int16 wrap."""
    H, W1, D = pix.shape
    r = q["SW2"]
    ys = np.clip(np.arange(-r, H + r), 0, H - 1)
    xs = np.clip(np.arange(-r, W1 + r), 0, W1 - 1)
    p = pix[ys][:, xs].astype(np.int64)
    cs = np.cumsum(np.cumsum(p, 0), 1)
    cs = np.pad(cs, ((1, 0), (1, 0), (0, 0)))
    k = 2 * r + 1
    s = cs[k:, k:] - cs[:-k, k:] - cs[k:, :-k] + cs[:-k, :-k]
    return (s + q["P2"]).astype(np.int16)


def _step(C, Lp, minp, P1, P2):
    """one SGM step for a batch of pixels: C, Lp (n, D) int32; minp (n,)"""
    big = np.full((Lp.shape[0], 1), MAX_COST, np.int32)
    dm = np.concatenate([big, Lp[:, :-1]], 1) + P1
    dp = np.concatenate([Lp[:, 1:], big], 1) + P1
    delta = (minp + P2)[:, None]
    L = C + np.minimum(np.minimum(Lp, dm), np.minimum(dp, delta)) - delta
    L = L.astype(np.int16).astype(np.int32)  # (CostType) cast
    return L, L.min(1)


def aggregate_dir(C, P1, P2, dx, dy):
    """L_r for r = (dx, dy) (previous pixel = (x-dx, y-dy)); zero state outside the cost array."""
    H, W1, D = C.shape
    C = C.astype(np.int32)
    L = np.zeros((H, W1, D), np.int32)
    if dy == 0:
        Lp = np.zeros((H, D), np.int32)
        mp = np.zeros(H, np.int32)
        xs = range(W1) if dx > 0 else range(W1 - 1, -1, -1)
        for x in xs:
            Lp, mp = _step(C[:, x], Lp, mp, P1, P2)
            L[:, x] = Lp
        return L
    ys = range(H) if dy > 0 else range(H - 1, -1, -1)
    Lrow = np.zeros((W1, D), np.int32)
    mrow = np.zeros(W1, np.int32)
    for y in ys:
        Lp = np.zeros((W1, D), np.int32)
        mp = np.zeros(W1, np.int32)
        if dx == 0:
            Lp, mp = Lrow, mrow
        elif dx > 0:   # previous pixel is x-1
            Lp[1:], mp[1:] = Lrow[:-1], mrow[:-1]
        else:          # previous pixel is x+1
            Lp[:-1], mp[:-1] = Lrow[1:], mrow[1:]
        Lrow, mrow = _step(C[y], Lp, mp, P1, P2)
        L[y] = Lrow
    return L


DIRS_SGBM = [(1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0)]
DIRS_HH = DIRS_SGBM + [(1, -1), (0, -1), (-1, -1)]
DIRS_HH4 = [(1, 0), (0, 1), (-1, 0), (0, -1)]


def aggregate(C, q):
    dirs = {0: DIRS_SGBM, 1: DIRS_HH, 3: DIRS_HH4}[q["mode"]]
    S = np.zeros(C.shape, np.int64)
    for dx, dy in dirs:
        S += aggregate_dir(C, q["P1"], q["P2"], dx, dy)
    return np.minimum(S, MAX_COST).astype(np.int16)


def wta(S, q, width):
    """WTA + uniqueness + sub-pixel + right-view map + LR check -> int16 (H, width)."""
    H, W1, D = S.shape
    minD, minX1 = q["minD"], q["minX1"]
    INVALID = (minD - 1) * 16
    S32 = S.astype(np.int32)
    minS = S32.min(2)
    best = S32.argmin(2)  # first (smallest d) minimum
    d_idx = np.arange(D)[None, None, :]
    bad = (S32 * (100 - q["uniq"]) < (minS * 100)[..., None]) & (np.abs(best[..., None] - d_idx) > 1)
    unique = ~bad.any(2)
    # degenerate rows where nothing is < MAX_COST keep bestDisp = -1 in OpenCV; not modelled
    disp1 = np.full((H, width), INVALID, np.int32)
    yy, xx = np.mgrid[:H, :W1]
    dm = np.clip(best - 1, 0, D - 1)
    dp = np.clip(best + 1, 0, D - 1)
    Sm = np.take_along_axis(S32, dm[..., None], 2)[..., 0]
    Sp = np.take_along_axis(S32, dp[..., None], 2)[..., 0]
    S0 = minS
    denom2 = np.maximum(Sm + Sp - 2 * S0, 1)
    num = (Sm - Sp) * 16 + denom2
    frac = np.sign(num) * (np.abs(num) // (denom2 * 2))  # C division truncates toward zero
    interior = (best > 0) & (best < D - 1)
    d16 = best * 16 + np.where(interior, frac, 0)
    val = d16 + minD * 16
    disp1[:, minX1:minX1 + W1] = np.where(unique, val, INVALID)
    # right-view map: min cost, ties -> larger x (visited first) == larger d
    key_init = (MAX_COST << 16)
    key = np.full((H, width), key_init, np.int64)
    x2 = xx + minX1 - best - minD
    cand = (minS.astype(np.int64) << 16) | (0xFFFF - best)
    cand = np.where(unique & (minS < MAX_COST), cand, np.int64(1) << 40)
    np.minimum.at(key, (yy, x2), cand)
    disp2 = np.where(key == key_init, INVALID, (0xFFFF - (key & 0xFFFF)) + minD).astype(np.int32)
    # LR check
    out = disp1.copy()
    xs = np.arange(width)[None, :].repeat(H, 0)
    d1 = disp1
    lo = d1 >> 4
    hi = (d1 + 15) >> 4
    xa, xb = xs - lo, xs - hi
    rows = np.arange(H)[:, None].repeat(width, 1)

    def bad_at(xq, dq):
        inr = (xq >= 0) & (xq < width)
        v = disp2[rows, np.clip(xq, 0, width - 1)]
        return inr & (v >= minD) & (np.abs(v - dq) > q["d12"])

    kill = (d1 != INVALID) & bad_at(xa, lo) & bad_at(xb, hi)
    inrange = (xs >= minX1) & (xs < q["maxX1"])
    out[kill & inrange] = INVALID
    return out.astype(np.int16)


def median3(img):
    p = np.pad(img, 1, mode="edge")
    H, W = img.shape
    st = np.stack([p[dy:dy + H, dx:dx + W] for dy in range(3) for dx in range(3)], 0)
    return np.sort(st, 0)[4].astype(img.dtype)


def filter_speckles(img, new_val, max_size, max_diff):
    """4-connected components under |a-b| <= max_diff (new_val pixels are walls); small ones -> new_val."""
    H, W = img.shape
    parent = np.arange(H * W)

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    I = img.astype(np.int64)
    ok = I != new_val
    for y in range(H):
        for x in range(W):
            if not ok[y, x]:
                continue
            if x + 1 < W and ok[y, x + 1] and abs(I[y, x] - I[y, x + 1]) <= max_diff:
                a, b = find(y * W + x), find(y * W + x + 1)
                if a != b:
                    parent[max(a, b)] = min(a, b)
            if y + 1 < H and ok[y + 1, x] and abs(I[y, x] - I[y + 1, x]) <= max_diff:
                a, b = find(y * W + x), find((y + 1) * W + x)
                if a != b:
                    parent[max(a, b)] = min(a, b)
    roots = np.array([find(i) for i in range(H * W)])
    cnt = np.bincount(roots, minlength=H * W)
    out = img.copy()
    small = (cnt[roots] <= max_size).reshape(H, W) & ok
    out[small] = new_val
    return out


def sgbm_compute(left, right, raw=False, **params):
    width = left.shape[1]
    q = normalise(params, width)
    H = left.shape[0]
    if q["width1"] <= 0:
        return np.full((H, width), (q["minD"] - 1) * 16, np.int16)
    C = box_cost(pixel_cost(left, right, q), q)
    S = aggregate(C, q)
    disp = wta(S, q, width)
    if raw:
        return disp
    disp = median3(disp)
    if q["speckleWindowSize"] > 0:
        disp = filter_speckles(disp, (q["minD"] - 1) * 16, q["speckleWindowSize"], 16 * q["speckleRange"])
    return disp
