"""Parity at the BASELINE configs that are not the headline (-m gpu): C4 (3840x2160, D=256) on a full-width strip
the oracle finishes in seconds plus size-independent properties at full size, and C5 (640x480, D=64, LR check and
speckle filter on) end to end through get_depth."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import calibrating_amd as ca  # noqa: E402
from calibrating_amd import synthetic  # noqa: E402


@pytest.mark.parametrize("mode", [0, 1])
def test_c4_4k_d256_strip_vs_oracle(oracle, mode):
    P = dict(minDisparity=0, numDisparities=256, blockSize=5, P1=200, P2=800, disp12MaxDiff=1, uniquenessRatio=10, mode=mode)
    left, right = synthetic.rectified_pair(seed=7, H=64, W=3840, D=256, cn=1)
    got = ca.StereoSGBM_create(**P).compute(left, right)
    assert np.abs(got.astype(int) - oracle.sgbm_compute(left, right, **P)).max() == 0


def test_c4_4k_d256_full_size_properties():
    P = dict(minDisparity=0, numDisparities=256, blockSize=5, P1=200, P2=800, disp12MaxDiff=1, uniquenessRatio=10)
    dev = torch.device("cuda", 0)
    L, R = synthetic.rectified_batch_torch(7, 2, 2160, 3840, 256, 1, dev)
    m = ca.StereoSGBM_create(**P)
    two = m.compute(L, R)                      # band passes
    one = m.compute(L[:1], R[:1])              # concurrent scans
    assert torch.equal(two[0], one[0])         # the two aggregation paths agree at full size
    assert (two[:, :, :256] == -16).all()      # the left band has no match
    assert (two >= 0).float().mean() > 0.6


def test_c5_vga_get_depth_vs_oracle(oracle):
    W, H = 640, 480
    stereo = ca.Stereo.load(synthetic.rig(W, H))
    cfg = dict(max_size=W, minDisparity=0, numDisparities=64, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1,
               uniquenessRatio=10, speckleWindowSize=100, speckleRange=2)
    stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(cfg), max_depth=3.5)
    i1, i2 = synthetic.scene_pair(9, W, H, 3)
    res = stereo.get_depth(i1, i2)
    r1 = oracle.remap_u8(i1, *stereo.undistort_rectify_map1, oracle.INTER_LANCZOS4)
    r2 = oracle.remap_u8(i2, *stereo.undistort_rectify_map2, oracle.INTER_LANCZOS4)
    s = stereo.min_disparity
    r2[:, s:] = r2[:, :-s].copy()
    r2[:, :s] = 0
    assert np.array_equal(res["rectify_img1"], r1) and np.array_equal(res["rectify_img2"], r2)
    disp16 = oracle.sgbm_compute(r1, r2, **{k: v for k, v in cfg.items() if k != "max_size"})
    disparity, depth = oracle.disp_to_depth(disp16, stereo.rectify_valid_mask1, 0, s, True,
                                            1.0 * stereo.baseline * stereo.K[0, 0], 3.5)
    assert np.array_equal(res["disparity"], disparity)
    assert np.abs(res["rectify_depth"] - depth).max() <= 1e-4
