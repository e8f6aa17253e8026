"""GPU parity of the HIP SGBM path against the CPU oracle, through the C ABI (-m gpu).

Bar: bit-exact int16 disparity (max |disp - oracle| == 0), stage by stage (cost volume C,
aggregated volume S, raw disparity, final disparity) on identical inputs.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import calibrating_amd as ca  # noqa: E402
from calibrating_amd import StereoSGBM_create, synthetic  # noqa: E402


def _params(cn, D, bs, minD=0, mode=0, **kw):
    p = dict(minDisparity=minD, numDisparities=D, blockSize=bs, P1=8 * cn * bs * bs, P2=32 * cn * bs * bs,
             disp12MaxDiff=1, uniquenessRatio=10, speckleWindowSize=0, speckleRange=0, mode=mode)
    p.update(kw)
    return p


def _check_stages(oracle, left, right, p, stages=True, paths=(2, 1, 3)):
    """paths (CAMD_OPT_PATH): 2 = fused band-wavefront passes (throughput path; falls back to 1 where not
    instantiated), 1 = one line scan per direction, 3 = all directions concurrently (latency path)"""
    got = None
    for path in paths:
        got = _check_stages_path(oracle, left, right, p, stages, path)
    return got


def _check_stages_path(oracle, left, right, p, stages, path):
    m = StereoSGBM_create(**p)
    m.set_option("path", path).set_option("keep_S", 1)
    got = m.compute(left, right)
    ref = oracle.sgbm_compute(left, right, **p)
    if stages and m.geometry()["width1"] > 0:
        C = m.debug_volume("C").cpu().numpy()
        Cr = oracle.sgbm_cost_volume(left, right, **p)
        assert np.array_equal(C, Cr), "cost volume differs: max |d| = %d at %s" % (
            np.abs(C.astype(int) - Cr).max(), np.argwhere(C != Cr)[:5].tolist())
        if path != 3:  # the concurrent path never materialises the summed volume
            S = m.debug_volume("S").cpu().numpy()
            Sr = oracle.sgbm_aggregated(left, right, **p)
            assert np.array_equal(S, Sr), "aggregated volume differs: max |d| = %d at %s" % (
                np.abs(S.astype(int) - Sr).max(), np.argwhere(S != Sr)[:5].tolist())
        raw = m.debug_volume("raw").cpu().numpy()
        rr = oracle.sgbm_compute(left, right, raw=True, **p)
        assert np.array_equal(raw, rr), "raw disparity differs at %s" % np.argwhere(raw != rr)[:5].tolist()
    assert got.dtype == np.int16 and got.shape == ref.shape
    assert np.array_equal(got, ref), "disparity differs: %d px, max |d| = %d" % (
        (got != ref).sum(), np.abs(got.astype(int) - ref).max())
    return got


CASES = [
    # H, W, D, cn, bs, minD, mode
    (48, 200, 128, 1, 5, 0, 0),    # 16 lanes x 1 vector, reference's mode
    (48, 200, 128, 3, 5, 0, 1),    # RGB, 8 paths
    (40, 330, 256, 1, 5, 0, 0),    # D = 256 (4K config): 2 vectors per lane
    (40, 330, 218, 3, 11, 2, 0),   # the reference's hard-coded matcher: D = 218 (padded), block 11, minD 2
    (37, 150, 64, 3, 5, 0, 0),     # 8-lane groups (VGA config)
    (33, 97, 32, 1, 3, 0, 1),      # 4-lane groups
    (31, 90, 16, 1, 3, 0, 0),      # 2-lane groups
    (30, 120, 48, 1, 7, -7, 1),    # negative minDisparity, padded D
    (25, 140, 100, 1, 9, 3, 0),    # D not a multiple of 8
    (20, 64, 40, 3, 5, 5, 1),
    (48, 200, 128, 3, 5, 0, 3),    # MODE_HH4 (4 paths), two bands
    (61, 150, 64, 1, 3, 1, 3),     # MODE_HH4, 8-lane groups
    (30, 330, 256, 1, 5, 0, 3),    # MODE_HH4, 2 vectors per lane
    (25, 97, 24, 1, 5, 0, 3),      # MODE_HH4 on the scan paths only (D <= 32)
]


@pytest.mark.parametrize("H,W,D,cn,bs,minD,mode", CASES)
def test_sgbm_stagewise_bit_exact(oracle, H, W, D, cn, bs, minD, mode):
    left, right = synthetic.rectified_pair(seed=11 + D, H=H, W=W, D=max(D, 8), cn=cn)
    _check_stages(oracle, left, right, _params(cn, D, bs, minD, mode))


def test_sgbm_random_noise_images(oracle):
    """Uncorrelated noise: exercises uniqueness rejections, LR-check failures, ties."""
    rng = np.random.default_rng(3)
    left = rng.integers(0, 256, (40, 180, 3), dtype=np.uint8)
    right = rng.integers(0, 256, (40, 180, 3), dtype=np.uint8)
    for mode in (0, 1):
        _check_stages(oracle, left, right, _params(3, 64, 3, 0, mode, uniquenessRatio=0, disp12MaxDiff=2))


def test_sgbm_flat_images_ties(oracle):
    """Constant images: every cost ties; bestDisp must be the smallest d, disp2 ties keep larger x."""
    left = np.full((24, 100), 77, np.uint8)
    right = np.full((24, 100), 77, np.uint8)
    got = _check_stages(oracle, left, right, _params(1, 32, 5, 0, 0))
    assert (got[:, :32] == -16).all()


def test_sgbm_default_params_and_speckle(oracle):
    left, right = synthetic.rectified_pair(seed=5, H=60, W=320, D=64, cn=3)
    # cv2.StereoSGBM_create() defaults (P1 = P2 = 0 -> 2 / 5, uniqueness 0, disp12MaxDiff 0 -> 1)
    _check_stages(oracle, left, right, dict(numDisparities=16, blockSize=3))
    # speckle filter on
    p = _params(3, 64, 5, 0, 0, speckleWindowSize=100, speckleRange=2)
    _check_stages(oracle, left, right, p, stages=False)
    p = _params(3, 64, 5, 1, 1, speckleWindowSize=30, speckleRange=1, uniquenessRatio=5)
    _check_stages(oracle, left, right, p, stages=False)


def test_sgbm_degenerate_width(oracle):
    """numDisparities >= width: minX1 >= maxX1, everything is (minD-1)*16."""
    left = np.zeros((10, 20), np.uint8)
    got = StereoSGBM_create(numDisparities=32, blockSize=3).compute(left, left)
    assert (got == -16).all()
    assert np.array_equal(got, oracle.sgbm_compute(left, left, numDisparities=32, blockSize=3))


def test_sgbm_batch_matches_single(oracle):
    pairs = [synthetic.rectified_pair(seed=s, H=32, W=200, D=64, cn=1) for s in (1, 2, 3)]
    L = np.stack([p[0] for p in pairs])
    R = np.stack([p[1] for p in pairs])
    p = _params(1, 64, 5)
    m = StereoSGBM_create(**p)
    got = m.compute(L, R)
    assert got.shape == (3, 32, 200)
    for i in range(3):
        assert np.array_equal(got[i], oracle.sgbm_compute(L[i], R[i], **p))
    # torch tensors in -> torch tensor out, same values
    gt = m.compute(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda())
    assert gt.is_cuda and np.array_equal(gt.cpu().numpy(), got)


def test_sgbm_known_answer_constant_shift():
    """right = left shifted by d0 -> interior |disp16 - 16 d0| <= 8."""
    rng = np.random.default_rng(0)
    H, W, D, d0 = 40, 260, 128, 37
    base = rng.integers(0, 256, (H, W + d0), dtype=np.uint8)
    left, right = base[:, :W].copy(), base[:, d0:].copy()
    for mode in (0, 1):
        d = StereoSGBM_create(**_params(1, D, 5, 0, mode)).compute(left, right)
        inner = d[4:-4, D + 8:-8 - d0].astype(int)
        assert (np.abs(inner - 16 * d0) <= 8).all()
        assert (d[:, :D] == -16).all()


def test_sgbm_error_behaviour():
    m = StereoSGBM_create(numDisparities=16, blockSize=3)
    with pytest.raises(ValueError):
        m.compute(np.zeros((8, 40), np.uint8), np.zeros((8, 41), np.uint8))
    with pytest.raises(ValueError):
        m.compute(np.zeros((8, 40), np.float32), np.zeros((8, 40), np.float32))
    with pytest.raises(ValueError):
        StereoSGBM_create(numDisparities=16, mode=2).compute(np.zeros((8, 40), np.uint8), np.zeros((8, 40), np.uint8))


@pytest.mark.parametrize("mode", [0, 1])
def test_sgbm_full_size_properties(mode):
    """BASELINE config (1920x1080, D=128, blockSize 5) through size-independent properties:
    determinism, invalid left band, batch invariance, and agreement with ground truth."""
    D = 128
    left, right = synthetic.rectified_pair(seed=1234, H=1080, W=1920, D=D, cn=1)
    m = StereoSGBM_create(**_params(1, D, 5, 0, mode))
    d1 = m.compute(left, right)
    d2 = m.compute(np.stack([left, left]), np.stack([right, right]))
    assert np.array_equal(d2[0], d1) and np.array_equal(d2[1], d1)
    assert (d1[:, :D] == -16).all()
    yy, xx = np.mgrid[:1080, :1920]
    g = np.rint(D / 8 + (3 * D / 4) * (0.5 + 0.5 * np.sin(2 * np.pi * xx / 1920) * np.cos(2 * np.pi * yy / 1080)))
    v = d1 >= 0
    assert v.mean() > 0.7
    assert (np.abs(d1[v] / 16.0 - g[v]) < 1).mean() > 0.9


@pytest.mark.parametrize("mode", [0, 1])
def test_sgbm_headline_config_whole_pair_vs_oracle(oracle, mode):
    """The bench workload itself -- one whole 1920x1080 RGB pair, D=128, blockSize 5 -- bit for bit against the oracle
    (a few seconds of scalar C per mode), through the band passes the throughput runs use (every row chunk of k_cost,
    all 39 bands and their edge records) and through the AUTO path a single pair takes; then inside a batch of three
    different pairs."""
    D = 128
    left, right = synthetic.rectified_pair(seed=1234, H=1080, W=1920, D=D, cn=3)
    p = _params(3, D, 5, 0, mode)
    want = oracle.sgbm_compute(left, right, **p)
    for path in (2, 0):
        m = StereoSGBM_create(**p)
        m.set_option("path", path)
        got = m.compute(left, right)
        assert np.array_equal(got, want), "path %d: %d pixels differ" % (path, (got != want).sum())
    l2, r2 = synthetic.rectified_pair(seed=99, H=1080, W=1920, D=D, cn=3)
    m = StereoSGBM_create(**p)
    m.set_option("path", 2)
    got = m.compute(np.stack([l2, left, l2[::-1].copy()]), np.stack([r2, right, r2[::-1].copy()]))
    assert np.array_equal(got[1], want)


def test_sgbm_full_size_rows_vs_oracle(oracle):
    """Full-width strip of the 1080p config against the oracle (the oracle finishes it in seconds)."""
    D = 128
    left, right = synthetic.rectified_pair(seed=1234, H=96, W=1920, D=D, cn=3)
    for mode in (0, 1):
        p = _params(3, D, 5, 0, mode)
        got = StereoSGBM_create(**p).compute(left, right)
        assert np.array_equal(got, oracle.sgbm_compute(left, right, **p))


# ---- MODE_SGBM_3WAY (cv2's four row stripes x three directions) ------------------------------------------------
@pytest.mark.parametrize("H,W,D,cn,bs,minD,extra", [
    (64, 200, 64, 1, 5, 0, {}),
    (90, 260, 128, 3, 5, 0, {}),
    (75, 300, 50, 3, 3, 0, {}),                       # D not a multiple of 8: SIMD region + scalar tail of the tie rule
    (81, 180, 24, 1, 7, -5, {}),
    (100, 240, 96, 3, 11, 2, dict(speckleWindowSize=60, speckleRange=2, uniquenessRatio=5)),
    (68, 160, 16, 1, 0, 0, dict(uniquenessRatio=0)),   # blockSize 0 -> radius 1 in this mode; uniqueness test off
    (97, 330, 200, 1, 5, 0, {}),                      # D > 128: two 8-groups per lane
])
@pytest.mark.parametrize("path", [2, 1, 0])  # 2: two band passes where the shape allows, 1: three line scans, 0: AUTO
def test_sgbm_3way_vs_oracle(oracle, H, W, D, cn, bs, minD, extra, path):
    left, right = synthetic.rectified_pair(seed=21, H=H, W=W, D=max(D, 8), cn=cn)
    b = bs if bs > 0 else 3
    p = dict(minDisparity=minD, numDisparities=D, blockSize=bs, P1=8 * cn * b * b, P2=32 * cn * b * b, disp12MaxDiff=1,
             uniquenessRatio=10, mode=ca.MODE_SGBM_3WAY)
    p.update(extra)
    m = ca.StereoSGBM_create(**p)
    m.set_option("path", path)
    got = m.compute(left, right)
    want = oracle.sgbm_compute(left, right, **p)
    assert np.array_equal(got, want), "%d of %d pixels differ" % ((got != want).sum(), got.size)
    assert (got[:, max(minD + D, 0):] >= minD * 16).mean() > 0.2  # (sanity: the matchable columns mostly match)
    # the raw stage too, and a batch of the same pair twice
    assert np.array_equal(m.debug_volume("raw").cpu().numpy(), oracle.sgbm_compute(left, right, raw=True, **p))
    both = m.compute(np.stack([left, left]), np.stack([right, right]))
    assert np.array_equal(both[0], want) and np.array_equal(both[1], want)


@pytest.mark.parametrize("lanes", [8, 1])
def test_sgbm_3way_tie_rule(oracle, lanes):
    """An image on which EVERY total ties (constant 15 = the value the BT border columns carry, so even the image
    borders do not break the tie): the winner is decided by the tie rule alone -- cv2's SIMD lane-slot rule (default:
    the last disparity of each of the 8 slots, then the smallest of those = D - 8 for D % 8 == 0) or the scalar build's
    smallest d.  D = 50 exercises the SIMD region [0, 48) + scalar tail {48, 49}."""
    H = 72
    try:
        oracle.set_switches(way3_simd_lanes=lanes)
        for D, winner8 in ((64, 56), (50, 40), (256, 248)):  # (256: two 8-groups per lane in the band pass)
            W = D + 156
            flat = np.full((H, W), 15, np.uint8)
            p = dict(minDisparity=0, numDisparities=D, blockSize=5, P1=200, P2=800, disp12MaxDiff=1, uniquenessRatio=0,
                     mode=ca.MODE_SGBM_3WAY)
            want = oracle.sgbm_compute(flat, flat, **p)
            for path in (2, 1):  # band passes (tie rule inside the last pass where D % 8 == 0) and line scans + k_wta
                m = ca.StereoSGBM_create(**p)
                m.set_option("way3_simd_lanes", lanes).set_option("path", path)
                got = m.compute(flat, flat)
                assert np.array_equal(got, want), (lanes, D, path, (got != want).sum())
            raw = oracle.sgbm_compute(flat, flat, raw=True, **p)
            assert raw[H // 2, W - 1] == (winner8 * 16 if lanes == 8 else 0), (lanes, D, raw[H // 2, W - 1])
    finally:
        oracle.set_switches()


@pytest.mark.parametrize("D", [32, 128, 200])
def test_sgbm_3way_ties_next_to_texture(oracle, D):
    """Flat blocks (every total ties) tiled into a textured pair: waves that meet both kinds of pixel evaluate the
    tie rule for some lanes only, and plateaus inside a block tie between a few disparities rather than all."""
    H, W = 96, D + 150
    left, right = synthetic.rectified_pair(seed=5, H=H, W=W, D=D, cn=1)
    left, right = left.copy(), right.copy()
    for img in (left, right):
        img[10:40, D + 10:D + 90] = 15
        img[50:90:2, :] = 15                 # flat rows between textured ones
        img[:, D + 100:D + 130] //= 64       # 4 grey levels: short plateaus
    p = dict(minDisparity=0, numDisparities=D, blockSize=3, P1=20, P2=80, disp12MaxDiff=1, uniquenessRatio=0,
             mode=ca.MODE_SGBM_3WAY)
    want = oracle.sgbm_compute(left, right, **p)
    for path in (2, 1):
        m = ca.StereoSGBM_create(**p)
        m.set_option("path", path)
        got = m.compute(left, right)
        assert np.array_equal(got, want), (path, (got != want).sum())


def test_sgbm_3way_refuses_tiny_images():
    with pytest.raises(ValueError, match="3WAY"):
        ca.StereoSGBM_create(numDisparities=16, blockSize=11, mode=ca.MODE_SGBM_3WAY).compute(
            np.zeros((12, 64), np.uint8), np.zeros((12, 64), np.uint8))
