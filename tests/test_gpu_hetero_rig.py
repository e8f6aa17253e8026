"""Rigs whose two cameras differ in resolution and intrinsics (-m gpu).  The reference rectifies each camera through
ITS OWN maps into the common target frame (/root/reference/calibrating/stereo_camera.py:159-165,216-228; used by
/root/reference/example/test_different_stereo.py:38-60), so nothing requires cam2.xy == cam1.xy; here both the
one-pair call and the batched form must follow."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import calibrating_amd as ca  # noqa: E402
from calibrating_amd import synthetic  # noqa: E402
from oracle_pipeline import compare, oracle_get_depth  # noqa: E402


def hetero_rig(xy1=(640, 480), xy2=(800, 600)):
    (w1, h1), (w2, h2) = xy1, xy2
    K1 = [[0.8 * w1, 0, w1 / 2 + 3.3], [0, 0.81 * w1, h1 / 2 - 2.1], [0, 0, 1]]
    K2 = [[0.78 * w2, 0, w2 / 2 - 5.2], [0, 0.79 * w2, h2 / 2 + 4.4], [0, 0, 1]]
    return dict(R=synthetic.rodrigues([0.012, -0.018, 0.006]).tolist(), t=[[-0.12], [0.002], [-0.001]],
                cam1=dict(K=K1, D=[[-0.12, 0.05, 1e-3, -5e-4, 0.01]], xy=list(xy1), name="wide"),
                cam2=dict(K=K2, D=[[0.08, -0.03, -8e-4, 6e-4, 0.002]], xy=list(xy2), name="fine"))


@pytest.mark.parametrize("xy1,xy2,max_size,max_depth", [((640, 480), (800, 600), 640, 3.5),
                                                        ((640, 480), (800, 600), 400, None),
                                                        ((500, 375), (321, 243), 500, 3.0)])
def test_hetero_rig_get_depth_and_batch(oracle, xy1, xy2, max_size, max_depth):
    rec = hetero_rig(xy1, xy2)
    stereo = ca.Stereo.load(rec)
    assert tuple(stereo.xy) == tuple(xy1)  # the target frame defaults to camera 1's size
    cfg = dict(max_size=max_size, minDisparity=0, numDisparities=64, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1,
               uniquenessRatio=10, speckleWindowSize=60, speckleRange=2)
    stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(cfg), max_depth=max_depth)
    a1, a2, _ = synthetic.render_plane_pair(rec, (0.2, 0.1, 1.0), 2.0)
    b1, b2, _ = synthetic.render_plane_pair(rec, (-0.1, 0.0, 1.0), 1.4, seed=3)
    assert a1.shape[:2] == xy1[::-1] and a2.shape[:2] == xy2[::-1]
    ref = oracle_get_depth(oracle, stereo, cfg, a1, a2)
    assert (ref["rectify_depth"] > 0).mean() > 0.5
    got = stereo.get_depth(a1, a2)
    bad, inexact = compare(got, ref)
    assert not bad and not inexact, (bad, inexact)
    gb = stereo.get_depth_batch(np.stack([b1, a1]), np.stack([b2, a2]))
    bad, inexact = compare({k: v[1] for k, v in gb.items()}, ref)
    assert not bad and not inexact, (bad, inexact)
    bad, inexact = compare({k: v[0] for k, v in gb.items()}, oracle_get_depth(oracle, stereo, cfg, b1, b2))
    assert not bad and not inexact, (bad, inexact)
    with pytest.raises(ValueError):
        stereo.get_depth_batch(np.stack([a1, b1]), np.stack([a2]))  # pair counts must agree


def test_hetero_rig_recovers_the_plane():
    """Oracle-independent: the rendered plane's true depth comes back through the two-resolution rig."""
    import ground_truth as gt
    normal, dist, eps, frac = gt.PLANES["slanted"]
    rec = hetero_rig()
    img1, img2, z_true = synthetic.render_plane_pair(rec, normal, dist)
    stereo = ca.Stereo.load(rec)
    stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(dict(gt.CFG, max_size=640, numDisparities=64)),
                               max_depth=gt.MAX_DEPTH)
    res = stereo.get_depth(img1, img2)
    gt.check_depth(res["unrectify_depth"], z_true, float(stereo.baseline), float(stereo.K[0, 0]), eps, frac,
                   "unrectify_depth (two-resolution rig)")
