"""CPU tests of the oracle itself (no GPU): the sequential C restatement against the independent
parallel-form NumPy model, analytic known answers, SciPy cross-checks for
the median / connected components (the committed golden fixtures: tests/test_golden_cpu.py)."""
import os

import numpy as np
import pytest

import np_sgbm_model as M
from calibrating_amd import synthetic

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _p(cn, D, bs, minD=0, mode=0, **kw):
    p = dict(minDisparity=minD, numDisparities=D, blockSize=bs, P1=8 * cn * bs * bs, P2=32 * cn * bs * bs,
             disp12MaxDiff=1, uniquenessRatio=10, mode=mode)
    p.update(kw)
    return p


@pytest.mark.parametrize("H,W,D,cn,minD,bs,mode", [
    (24, 64, 16, 1, 0, 3, 0), (20, 70, 16, 3, 0, 5, 0), (18, 60, 16, 1, 2, 5, 1),
    (20, 72, 24, 3, -3, 3, 1), (16, 50, 10, 1, 0, 7, 0), (12, 90, 40, 1, 4, 11, 1),
    (22, 66, 16, 1, 0, 5, 3), (17, 75, 24, 3, 2, 3, 3),   # MODE_HH4
])
def test_oracle_matches_numpy_model_stagewise(oracle, H, W, D, cn, minD, bs, mode):
    left, right = synthetic.rectified_pair(seed=5, H=H, W=W, D=max(D, 8), cn=cn)
    p = _p(cn, D, bs, minD, mode)
    q = M.normalise(p, W)
    C = M.box_cost(M.pixel_cost(left, right, q), q)
    assert np.array_equal(oracle.sgbm_cost_volume(left, right, **p), C)
    S = M.aggregate(C, q)
    assert np.array_equal(oracle.sgbm_aggregated(left, right, **p), S)
    assert np.array_equal(oracle.sgbm_compute(left, right, raw=True, **p), M.wta(S, q, W))
    assert np.array_equal(oracle.sgbm_compute(left, right, **p), M.sgbm_compute(left, right, **p))


def test_oracle_matches_numpy_model_noise_and_speckle(oracle):
    rng = np.random.default_rng(1)
    left = rng.integers(0, 256, (22, 80, 3), dtype=np.uint8)
    right = rng.integers(0, 256, (22, 80, 3), dtype=np.uint8)
    for mode in (0, 1):
        p = _p(3, 24, 3, 1, mode, uniquenessRatio=0, disp12MaxDiff=2, speckleWindowSize=12, speckleRange=1)
        assert np.array_equal(oracle.sgbm_compute(left, right, **p), M.sgbm_compute(left, right, **p))


@pytest.mark.parametrize("mode", [0, 1, 3])
def test_known_answer_constant_shift(oracle, mode):
    rng = np.random.default_rng(0)
    H, W, D, d0 = 40, 160, 32, 11
    base = rng.integers(0, 256, (H, W + d0), dtype=np.uint8)
    left, right = base[:, :W].copy(), base[:, d0:].copy()
    d = oracle.sgbm_compute(left, right, **_p(1, D, 5, 0, mode))
    inner = d[4:-4, D + 8:-8 - d0].astype(int)
    assert (np.abs(inner - 16 * d0) <= 8).all()
    assert (d[:, :D] == -16).all()  # columns left of maxD are INVALID_DISP_SCALED = (minD-1)*16


def test_known_answer_constant_images(oracle):
    """All costs tie: bestDisp = 0, uniqueness never fails; before the LR check disp = minD*16."""
    img = np.full((20, 90), 100, np.uint8)
    for minD in (0, 3):
        p = _p(1, 16, 5, minD, 0)
        raw = oracle.sgbm_compute(img, img, raw=True, **p)
        minX1 = minD + 16
        col = raw[:, minX1:]
        assert set(np.unique(col)) <= {minD * 16, (minD - 1) * 16}
        assert (raw[:, :minX1] == (minD - 1) * 16).all()


def test_default_normalisation(oracle):
    """P1 <= 0 -> 2, P2 -> max(P2 > 0 ? P2 : 5, P1+1), uniquenessRatio < 0 -> 10, disp12MaxDiff <= 0 -> 1."""
    left, right = synthetic.rectified_pair(seed=2, H=20, W=90, D=16, cn=1)
    a = oracle.sgbm_compute(left, right, numDisparities=16, blockSize=3)
    b = oracle.sgbm_compute(left, right, numDisparities=16, blockSize=3, P1=2, P2=5, disp12MaxDiff=1)
    assert np.array_equal(a, b)
    c = oracle.sgbm_compute(left, right, numDisparities=16, blockSize=3, uniquenessRatio=-1)
    d = oracle.sgbm_compute(left, right, numDisparities=16, blockSize=3, uniquenessRatio=10)
    assert np.array_equal(c, d)
    e = oracle.sgbm_compute(left, right, numDisparities=16, blockSize=0)
    f = oracle.sgbm_compute(left, right, numDisparities=16, blockSize=5)
    assert np.array_equal(e, f)
    with pytest.raises(ValueError):
        oracle.sgbm_compute(left, right, numDisparities=0)


def test_median_against_scipy(oracle):
    ndi = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(3)
    for shape in ((31, 47), (1, 40), (40, 1), (2, 2)):
        img = rng.integers(-300, 3000, shape).astype(np.int16)
        want = ndi.median_filter(img, size=3, mode="nearest")
        assert np.array_equal(oracle.median3_s16(img), want)
        assert np.array_equal(M.median3(img), want)


def test_speckle_against_model_and_hand_pattern(oracle):
    img = np.array([[16, 16, -16, 80, 80],
                    [16, -16, -16, 80, 80],
                    [-16, -16, 48, -16, 80],
                    [160, 160, -16, 80, 80],
                    [160, 170, 180, -16, 80]], np.int16)
    out = oracle.filter_speckles_s16(img, -16, 3, 16)
    want = img.copy()
    want[0, 0:2] = -16; want[1, 0] = -16   # component of size 3 (<= 3) removed
    want[2, 2] = -16                        # single pixel removed
    # {160,160,160,170,180}: chain 160-170-180 with steps 10 -> one component of size 5: kept
    # the 80s: 9 pixels, kept
    assert np.array_equal(out, want)
    rng = np.random.default_rng(4)
    img = (rng.integers(-1, 12, (40, 50)) * 16).astype(np.int16)
    for max_size, max_diff in ((4, 16), (30, 32), (1000, 0)):
        assert np.array_equal(oracle.filter_speckles_s16(img, -16, max_size, max_diff),
                              M.filter_speckles(img, -16, max_size, max_diff))


def test_remap_tables_and_identity(oracle):
    lt = oracle.lanczos4_itab()
    bt = oracle.bilinear_itab()
    assert (lt.astype(int).sum(1) == 32768).all() and (bt.astype(int).sum(1) == 32768).all()
    # phase (0,0): the unit weight saturates to 32767 (int16) and the missing 1 goes to the
    # (ksize/2, ksize/2) tap -- cv2's table quirk; the remap of integer coordinates stays the identity
    assert lt[0, 3 * 8 + 3] == 32767 and lt[0, 4 * 8 + 4] == 1 and np.count_nonzero(lt[0]) == 2
    assert list(bt[0]) == [32767, 0, 0, 1]
    # bilinear weights are exactly (1-fy)(1-fx) * 32768
    fy, fx = 5, 9
    w = bt[fy * 32 + fx]
    assert list(w) == [(32 - fy) * (32 - fx) * 32, (32 - fy) * fx * 32, fy * (32 - fx) * 32, fy * fx * 32]
    # Lanczos separability / symmetry: table at (fy, fx) mirrored equals table at (32-fy, 32-fx) reversed
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, (30, 40, 3), dtype=np.uint8)
    yy, xx = np.mgrid[:30, :40].astype(np.float32)
    for interp in (0, 1, 4):
        assert np.array_equal(oracle.remap_u8(src, xx, yy, interp), src)
    # a constant image stays constant wherever the 8x8 footprint is inside (weights sum to 1)
    const = np.full((40, 40), 200, np.uint8)
    mx = (xx[:20, :20] * 0.7 + 10.3).astype(np.float32)
    my = (yy[:20, :20] * 0.6 + 9.7).astype(np.float32)
    assert (oracle.remap_u8(const, mx, my, 4) == 200).all()
    # fully outside -> 0 ; round-half-even for nearest
    far = np.full((2, 2), -50, np.float32)
    assert (oracle.remap_u8(const, far, far, 4) == 0).all()
    ramp = np.arange(40, dtype=np.uint8)[None].repeat(4, 0)
    mxh = np.array([[0.5, 1.5, 2.5, 3.5]], np.float32)
    assert list(oracle.remap_u8(ramp, mxh, np.zeros_like(mxh), 0)[0]) == [0, 2, 2, 4]


def test_init_undistort_rectify_map_identity_and_model(oracle):
    K = np.array([[500.0, 0, 160.5], [0, 505.0, 120.25], [0, 0, 1]])
    mx, my = oracle.init_undistort_rectify_map(K, None, None, K, (320, 240))
    yy, xx = np.mgrid[:240, :320]
    assert np.abs(mx - xx).max() < 1e-3 and np.abs(my - yy).max() < 1e-3
    # distorted: compare with a direct float64 evaluation of the Brown model
    D = np.array([-0.12, 0.05, 1e-3, -5e-4, 0.01])
    R = synthetic.rodrigues([0.01, -0.02, 0.005])
    Kn = np.array([[480.0, 0, 150.0], [0, 480.0, 118.0], [0, 0, 1]])
    mx, my = oracle.init_undistort_rectify_map(K, D, R, Kn, (320, 240))
    pts = np.stack([xx, yy, np.ones_like(xx)], -1).reshape(-1, 3).astype(np.float64)
    ray = pts @ np.linalg.inv(Kn @ R).T
    x, y = ray[:, 0] / ray[:, 2], ray[:, 1] / ray[:, 2]
    r2 = x * x + y * y
    kr = 1 + D[0] * r2 + D[1] * r2 ** 2 + D[4] * r2 ** 3
    xd = x * kr + 2 * D[2] * x * y + D[3] * (r2 + 2 * x * x)
    yd = y * kr + D[2] * (r2 + 2 * y * y) + 2 * D[3] * x * y
    assert np.abs(mx.reshape(-1) - (K[0, 0] * xd + K[0, 2])).max() < 1e-3
    assert np.abs(my.reshape(-1) - (K[1, 1] * yd + K[1, 2])).max() < 1e-3


def test_undistort_identity(oracle):
    K = np.array([[300.0, 0, 64.0], [0, 300.0, 48.0], [0, 0, 1]])
    rng = np.random.default_rng(6)
    img = rng.integers(0, 256, (96, 128, 3), dtype=np.uint8)
    assert np.array_equal(oracle.undistort_u8(img, K, None), img)
    out = oracle.undistort_u8(img, K, np.array([-0.2, 0.05, 0, 0, 0]))
    assert out.shape == img.shape and not np.array_equal(out, img)
    assert np.array_equal(out[48, 64], img[48, 64])  # the principal point does not move


def test_disp_to_depth_matches_reference_expression(oracle):
    """The reference's own NumPy lines (stereo_matching.py:63-69, stereo_camera.py:510-513,408-413)."""
    rng = np.random.default_rng(7)
    h, w = 30, 44
    disp16 = rng.integers(-16, 100 * 16, (h, w)).astype(np.int16)
    disp16[0, :4] = [0, -16, 1, 264]
    mask = rng.random((h, w)) < 0.8
    minD, bf, max_depth = 2, 0.12 * 1000.0, 3.5
    for translate, add in ((False, 0), (True, 9)):
        sd = disp16.astype(np.float32).clip(0)
        sd[sd < minD * 16] = 0
        disparity = sd / 16.0 * w / w
        if translate:
            disparity += add
        disparity = mask * disparity
        with np.errstate(divide="ignore"):
            depth = 1.0 * np.float64(bf) / disparity
        depth[depth > max_depth] = 0
        depth[depth < 0] = 0
        gd, gz = oracle.disp_to_depth(disp16, mask, minD, add, translate, bf, max_depth)
        assert disparity.dtype == np.float32 and depth.dtype == np.float64  # NumPy >= 2 promotion (F9)
        assert np.array_equal(gd, disparity) and np.array_equal(gz, depth)
    # d in {0, -1 (invalid), tiny, 16.5} -> {0, 0, 0 (> max_depth), bf/16.5}
    _, z = oracle.disp_to_depth(np.array([[0, -16, 1, 264]], np.int16), np.ones((1, 4), bool), 0, 0, False,
                                bf, 1000.0)
    assert z[0, 0] == 0 and z[0, 1] == 0 and z[0, 2] == 0 and z[0, 3] == bf / 16.5


def test_unrectify_matches_reference_expression(oracle):
    rng = np.random.default_rng(8)
    h, w = 20, 30
    depth = rng.uniform(0, 4, (h, w))
    K = np.array([[40.0, 0, 15], [0, 40, 10], [0, 0, 1]])
    R = synthetic.rodrigues([0.02, -0.01, 0.03])
    Mfull = R @ np.linalg.inv(K)
    ys, xs = np.mgrid[:h, :w]
    pts = np.array([xs.flatten(), ys.flatten(), np.ones(h * w, int)]) * depth.flatten()[None]
    want_z = (Mfull @ pts).T[:, 2].reshape(h, w)
    mx, my = xs.astype(np.float32), ys.astype(np.float32)
    got = oracle.unrectify_depth(depth, Mfull[2], mx, my)
    assert np.abs(got - want_z).max() < 1e-12


def test_oracle_refuses_undefined_narrow_case(oracle):
    """0 < width1 <= blockSize/2: cv2's first box sum reads columns it never computed -> no defined answer."""
    img = np.zeros((12, 66), np.uint8)
    with pytest.raises(ValueError):
        oracle.sgbm_compute(img, img, numDisparities=64, blockSize=5)
    assert oracle.sgbm_compute(img, img, numDisparities=64, blockSize=3).shape == (12, 66)


def test_pointcloud_oracle_roundtrip_and_zbuffer():
    """oracle/pointcloud_ref.py: depth -> cloud -> depth is the identity; the nearest point wins a pixel."""
    from oracle import pointcloud_ref as ref
    K = np.array([[200.0, 0, 31.5], [0, 200.0, 23.5], [0, 0, 1]])
    rng = np.random.default_rng(0)
    depth = 1 + rng.random((48, 64))
    depth[rng.random((48, 64)) < 0.3] = 0
    cloud = ref.depth_to_point_cloud(depth, K)
    assert cloud.shape == ((depth != 0).sum(), 3)
    assert np.allclose(ref.point_cloud_to_depth(cloud, K, (64, 48)), depth)
    two = np.array([[0.0, 0.0, 3.0], [0.0, 0.0, 2.0], [0.0, 0.0, 5.0]])
    assert ref.point_cloud_to_depth(two, K, (64, 48))[24, 32] == 2.0
    up = ref.depth_to_point_cloud(depth, K, interpolation_rate=2, return_xyzuv=True)
    assert up.shape[1] == 5 and up[:, 3].max() <= 63.5
    assert ref.resize_nearest(np.arange(6.0).reshape(2, 3), (6, 4)).tolist() == [
        [0, 0, 1, 1, 2, 2], [0, 0, 1, 1, 2, 2], [3, 3, 4, 4, 5, 5], [3, 3, 4, 4, 5, 5]]
