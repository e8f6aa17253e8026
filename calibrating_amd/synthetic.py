"""Deterministic synthetic inputs: rectified stereo pairs and a test rig.

NumPy only; used by bench.py, the tests and ``__graft_entry__.smoke()``.  No reference data set is
reachable offline (calibrating/utils.py:722-739 clones it from the network), so every config of
BASELINE.json is driven on these.
"""
import numpy as np


def _box3(img):
    """3x3 box mean (integer floor) with replicated border; img (H, W, cn) uint8."""
    p = np.pad(img.astype(np.int32), ((1, 1), (1, 1), (0, 0)), mode="edge")
    h, w = img.shape[:2]
    s = np.zeros(img.shape, np.int32)
    for dy in range(3):
        for dx in range(3):
            s += p[dy:dy + h, dx:dx + w]
    return (s // 9).astype(np.uint8)


def rectified_pair(seed=1234, H=1080, W=1920, D=128, cn=1):
    """Appendix C.1: random texture, smooth sinusoidal ground-truth disparity in [D/8, 7D/8].

    Returns (left, right) uint8 of shape (H, W) for cn == 1, (H, W, cn) otherwise.
    """
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (H, W + D, cn), dtype=np.uint8)
    base = _box3(base)
    left = base[:, D:].copy()
    yy, xx = np.mgrid[:H, :W]
    g = np.rint(D / 8 + (3 * D / 4) * (0.5 + 0.5 * np.sin(2 * np.pi * xx / W) * np.cos(2 * np.pi * yy / H)))
    g = g.astype(np.int64)
    right = rng.integers(0, 256, (H, W, cn), dtype=np.uint8)  # holes
    # ascending disparity order: nearer (larger g) written last, so it wins
    xr = xx - g
    order = np.argsort(g, axis=1, kind="stable")
    rows = np.arange(H)[:, None]
    xs_sorted = np.take_along_axis(xx, order, axis=1)
    xr_sorted = np.take_along_axis(xr, order, axis=1)
    ok = xr_sorted >= 0
    # sequential semantics per row: later writes override earlier ones -> process in sorted order
    # numpy fancy assignment keeps the last write for repeated indices when done column by column
    for j in range(W):
        m = ok[:, j]
        r = rows[m, 0]
        right[r, xr_sorted[m, j]] = left[r, xs_sorted[m, j]]
    noise = rng.integers(-2, 3, right.shape)
    right = np.clip(right.astype(np.int64) + noise, 0, 255).astype(np.uint8)
    if cn == 1:
        return left[..., 0], right[..., 0]
    return left, right


def drift_pair(H, W, cn=3, split=0.5, seed=0):
    """An adversarial pair for the int16 regime of the cost volume: opposite sawtooth ramps in the upper part (every
    pixel cost at its maximum: with a large block and preFilterCap the box sums overflow int16), identical random
    texture below (pixel cost 0).  The recurrence that builds C keeps what it lost in the overflow, so below the split
    C falls under P2 or turns negative -- the regime in which OpenCV's int arithmetic and packed u16 part ways."""
    x, y = np.arange(W)[None, :], np.arange(H)[:, None]
    ramp = ((x * 16 + y * 40) % 256).astype(np.uint8)
    left, right = ramp.copy(), 255 - ramp
    h0 = int(H * split)
    tex = np.random.default_rng(seed).integers(0, 256, (H, W), dtype=np.uint8)
    left[h0:] = tex[h0:]
    right[h0:] = tex[h0:]
    if cn == 3:
        left, right = left[..., None].repeat(3, 2), right[..., None].repeat(3, 2)
    return left, right


def rodrigues(r):
    """cv2.Rodrigues(vector) -> 3x3."""
    r = np.asarray(r, np.float64).reshape(3)
    theta = np.linalg.norm(r)
    if theta < np.finfo(np.float64).eps:
        return np.eye(3)
    k = r / theta
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(theta) * np.eye(3) + (1 - np.cos(theta)) * np.outer(k, k) + np.sin(theta) * K


def rig(W=1280, H=720):
    """Appendix C.2 rig as the dict ``Stereo.load`` accepts (stereo_camera.py:264-297)."""
    K1 = [[0.8 * W, 0, W / 2 + 3.3], [0, 0.8 * W, H / 2 - 2.1], [0, 0, 1]]
    K2 = [[0.8 * W * 1.01, 0, W / 2 + 3.3 - 4.7], [0, 0.8 * W * 1.01, H / 2 - 2.1], [0, 0, 1]]
    return dict(
        R=rodrigues([0.01, -0.02, 0.005]).tolist(),
        t=[[-0.12], [0.002], [-0.001]],
        cam1=dict(K=K1, D=[[-0.12, 0.05, 1e-3, -5e-4, 0.01]], xy=[W, H], name="cam1"),
        cam2=dict(K=K2, D=[[-0.10, 0.04, -8e-4, 6e-4, 0.0]], xy=[W, H], name="cam2"),
    )


def _undistort_points(xd, yd, dist, iters=40):
    """Normalised distorted -> ideal coordinates under the Brown model (k1 k2 p1 p2 k3): the fixed-point iteration of
    cv2.undistortPoints, run to convergence."""
    k1, k2, p1, p2, k3 = (list(np.asarray(dist, np.float64).reshape(-1)) + [0.0] * 5)[:5]
    x, y = xd.copy(), yd.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        radial = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x, y = (xd - dx) / radial, (yd - dy) / radial
    return x, y


def plane_texture(a, b, cn=3, seed=0):
    """Analytic band-limited texture on a plane: a, b = plane coordinates in metres.  A sum of sinusoids with
    wavelengths of 2.5 .. 25 cm in random directions (no sampling, no interpolation); uint8 (..., cn)."""
    rng = np.random.default_rng(seed)
    out = np.empty(a.shape + (cn,), np.float64)
    for c in range(cn):
        acc = np.zeros(a.shape)
        n = 14
        for k in range(n):
            wl = 0.025 * (10.0 ** rng.uniform(0, 1))
            th, ph = rng.uniform(0, 2 * np.pi), rng.uniform(0, 2 * np.pi)
            acc += np.sin(2 * np.pi / wl * (np.cos(th) * a + np.sin(th) * b) + ph)
        out[..., c] = 127.5 + acc * (120.0 / np.sqrt(n) / 2.2)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def render_plane_pair(rig_rec, normal=(0.0, 0.0, 1.0), distance=2.0, cn=3, seed=0):
    """Ground truth for the whole depth path, independent of any remap or matcher code: a textured plane
    {X : n.X = n.(0, 0, distance)} (camera-1 coordinates, metres) rendered into both cameras of the rig record by
    per-pixel ray casting through the full Brown model -- pixel -> normalised distorted coordinates -> iterative
    undistortion -> ray -> plane -> analytic texture (plane_texture); nothing is interpolated.

    Returns (img1, img2, z_true) with z_true(v, u) = the depth along camera 1's optical axis of what the IDEAL
    (undistorted, intrinsics K1) camera 1 sees at pixel (u, v): the frame Stereo.unrectify_depth reports in
    (/root/reference/calibrating/stereo_camera.py:415-428)."""
    R = np.asarray(rig_rec["R"], np.float64)
    t = np.asarray(rig_rec["t"], np.float64).reshape(3)  # X2 = R X1 + t
    n = np.asarray(normal, np.float64)
    n = n / np.linalg.norm(n)
    d0 = n[2] * distance
    e1 = np.cross([0.0, 1.0, 0.0], n)
    e1 /= np.linalg.norm(e1)
    e2 = np.cross(n, e1)

    def rays(cam, distorted=True):
        K = np.asarray(cam["K"], np.float64)
        w, h = cam["xy"]
        v, u = np.mgrid[:h, :w].astype(np.float64)
        x, y = (u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1]
        if distorted:
            x, y = _undistort_points(x, y, cam["D"])
        return np.stack([x, y, np.ones_like(x)], -1)

    def shade(X1):
        return plane_texture(X1 @ e1, X1 @ e2, cn, seed)

    r1 = rays(rig_rec["cam1"])
    img1 = shade(r1 * (d0 / (r1 @ n))[..., None])
    r2 = rays(rig_rec["cam2"]) @ R                     # R^T r2: the ray direction in camera-1 coordinates
    o2 = -R.T @ t                                       # camera 2's centre in camera-1 coordinates
    s2 = (d0 - n @ o2) / (r2 @ n)
    img2 = shade(o2 + r2 * s2[..., None])
    ri = rays(rig_rec["cam1"], distorted=False)
    z_true = d0 / (ri @ n)
    if cn == 1:
        img1, img2 = img1[..., 0], img2[..., 0]
    return img1, img2, z_true


def scene_pair(seed=7, W=1280, H=720, cn=3):
    """Unrectified-looking random textured pair for the full get_depth pipeline (content is
    arbitrary: the parity tests only need identical inputs on both sides)."""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (H, W + 64, cn), dtype=np.uint8)
    base = _box3(_box3(base))
    img1 = base[:, 64:].copy()
    img2 = base[:, 40:40 + W].copy()
    return img1, img2


def rectified_batch_torch(seed, n, H=1080, W=1920, D=128, cn=1, device="cuda"):
    """n distinct rectified pairs generated on the GPU (bench input; no host loop).

    Same family as ``rectified_pair`` -- box-smoothed random texture, sinusoidal disparity field in
    [D/8, 7D/8] whose phase differs per pair -- but built by a gather (left[x] = right[x - g(x)]) so it
    vectorises.  Returns uint8 tensors (n, H, W) for cn == 1, (n, H, W, cn) otherwise.
    """
    import torch
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    right = torch.randint(0, 256, (n, H + 2, W + 2, cn), generator=gen, device=device, dtype=torch.int32)
    acc = torch.zeros((n, H, W, cn), dtype=torch.int32, device=device)
    for dy in range(3):
        for dx in range(3):
            acc += right[:, dy:dy + H, dx:dx + W]
    right = (acc // 9)
    yy = torch.arange(H, device=device, dtype=torch.float32)[None, :, None]
    xx = torch.arange(W, device=device, dtype=torch.float32)[None, None, :]
    ph = torch.arange(n, device=device, dtype=torch.float32)[:, None, None] * 0.37
    g = torch.round(D / 8 + (3 * D / 4) * (0.5 + 0.5 * torch.sin(2 * np.pi * xx / W + ph) * torch.cos(2 * np.pi * yy / H)))
    src = (xx - g).long().clamp_(0, W - 1).expand(n, H, W)
    left = torch.gather(right, 2, src[..., None].expand(n, H, W, cn))
    noise = torch.randint(-2, 3, left.shape, generator=gen, device=device, dtype=torch.int32)
    left = (left + noise).clamp_(0, 255).to(torch.uint8)
    right = right.to(torch.uint8)
    if cn == 1:
        return left[..., 0].contiguous(), right[..., 0].contiguous()
    return left.contiguous(), right.contiguous()
