"""The matcher's ``max_size`` downsizing branch through ``Stereo.get_depth`` / ``get_depth_batch`` against the oracle
composition (-m gpu).  This is the branch every user of the reference's UNMODIFIED default plugin takes on an image
wider than 1000 px (/root/reference/calibrating/stereo_matching.py:27,60-70; stereo_camera.py:506-513): on the GPU it
is resize x2 -> SGBM -> ONE kernel (k_disp16_up_to_depth: int16 -> f32, clip / threshold, /16, cv2.resize's bilinear
back to the rectified size, ``* w / sw``, ``+= min_disparity``, mask, float64 depth)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import calibrating_amd as ca  # noqa: E402
from calibrating_amd import synthetic  # noqa: E402
from oracle_pipeline import compare, oracle_get_depth  # noqa: E402

KEYS = ("rectify_img1", "rectify_img2", "undistort_img1", "disparity", "rectify_depth", "unrectify_depth")


def _check(got, ref, what):
    bad, inexact = compare(got, ref, keys=KEYS)
    assert not bad, (what, bad)
    assert not inexact, (what, "within 1e-4 m but not the same float64 bits", inexact)  # what is actually reached


@pytest.mark.parametrize("max_depth", [None, 3.5])
def test_default_plugin_on_a_wide_pair(oracle, max_depth):
    """``SemiGlobalBlockMatching({})``: max_size 1000, D=218, block 11, minDisparity 2, speckle 200 / 2 -- on a rendered
    1280 x 720 pair (matched at 1000 x 562), one call and a batch of two different pairs."""
    W, H = 1280, 720
    rec = synthetic.rig(W, H)
    stereo = ca.Stereo.load(rec)
    stereo.set_stereo_matching(ca.SemiGlobalBlockMatching({}), max_depth=max_depth)
    a1, a2, _ = synthetic.render_plane_pair(rec, (0.3, 0.1, 1.0), 2.0)
    b1, b2, _ = synthetic.render_plane_pair(rec, (-0.2, 0.15, 1.0), 1.6, seed=5)
    got = stereo.get_depth(a1, a2)
    ref_a = oracle_get_depth(oracle, stereo, {}, a1, a2)
    _check(got, ref_a, "get_depth")
    assert (got["rectify_depth"] > 0).mean() > 0.5, "the comparison must run on real depths, not on zeros"
    gb = stereo.get_depth_batch(np.stack([b1, a1]), np.stack([b2, a2]))
    _check({k: v[1] for k, v in gb.items()}, ref_a, "get_depth_batch[1]")
    _check({k: v[0] for k, v in gb.items()}, oracle_get_depth(oracle, stereo, {}, b1, b2), "get_depth_batch[0]")
    # device tensors in -> device tensors out, same bits
    gt = stereo.get_depth(torch.from_numpy(a1).cuda(), torch.from_numpy(a2).cuda())
    _check({k: v.cpu().numpy() for k, v in gt.items()}, ref_a, "get_depth (tensors)")


@pytest.mark.parametrize("max_depth", [None, 3.5])
@pytest.mark.parametrize("frac", ["W-1", "0.37W", "W/2+1", "W/2"])
def test_max_size_ratios(oracle, frac, max_depth):
    """Ratios just below 1, an odd small one, just above and exactly one half (cv2.resize's exact-2x area path), on an
    odd-sized rig, all four SGBM modes spread over the cases; batch of three."""
    W, H = (404, 302) if frac == "W/2" else (403, 301)
    rec = synthetic.rig(W, H)
    stereo = ca.Stereo.load(rec)
    max_size = {"W-1": W - 1, "0.37W": int(0.37 * W), "W/2+1": W // 2 + 1, "W/2": W // 2}[frac]
    mode = {"W-1": 0, "0.37W": 1, "W/2+1": 3, "W/2": 0}[frac]
    cfg = dict(max_size=max_size, minDisparity=1, numDisparities=48, blockSize=5, P1=200, P2=800, disp12MaxDiff=1,
               uniquenessRatio=8, speckleWindowSize=40, speckleRange=2, mode=mode)
    stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(cfg), max_depth=max_depth)
    pairs = [synthetic.render_plane_pair(rec, (0.25, -0.1, 1.0), 2.2)[:2], synthetic.scene_pair(3, W, H, 3),
             synthetic.render_plane_pair(rec, (0.0, 0.0, 1.0), 1.5, seed=2)[:2]]
    refs = [oracle_get_depth(oracle, stereo, cfg, a, b) for a, b in pairs]
    assert any((r["rectify_depth"] > 0).mean() > 0.3 for r in refs)
    for i, (a, b) in enumerate(pairs):
        _check(stereo.get_depth(a, b), refs[i], "get_depth %d" % i)
    gb = stereo.get_depth_batch(np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs]))
    for i in range(len(pairs)):
        _check({k: v[i] for k, v in gb.items()}, refs[i], "get_depth_batch[%d]" % i)


def test_plugin_call_equals_stereo_branch(oracle):
    """The plugin called on its own (``m(left, right)``, the staged path: resize kernel + element-wise ops) and the fused
    kernel inside get_depth produce the same disparity where the mask is set and no translation is applied."""
    W, H = 1200, 270
    left, right = synthetic.rectified_pair(seed=3, H=H, W=W, D=128, cn=3)
    m = ca.SemiGlobalBlockMatching({})
    from oracle_pipeline import matcher_disparity
    want = matcher_disparity(oracle, {}, left, right)
    assert np.array_equal(m(left, right), want)
    assert np.array_equal(m.call_batch(torch.from_numpy(np.stack([right, left])).cuda(),
                                       torch.from_numpy(np.stack([left, right])).cuda())[1].cpu().numpy(), want)
