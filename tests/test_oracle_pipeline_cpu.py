"""The end-to-end checker (tests/oracle_pipeline.py: the reference's get_depth composed from oracle stages) against
the oracle's own one-pass restatement of the stages after SGBM (oracle.disp_to_depth, which follows
stereo_matching.py:63-69 + stereo_camera.py:510-513,408-413 in C): two independent spellings of the same lines must
agree bit for bit at the matcher's full resolution, with and without the min_disparity translation -- and the
downsizing branch must reduce to it when max_size is not below the image.  CPU only."""
import numpy as np
import pytest

import calibrating_amd as ca
from calibrating_amd import synthetic
from oracle_pipeline import matcher_disparity, oracle_get_depth, rectified_pair, sgbm_params


class _NoMatcher:  # Stereo only needs *a* matcher installed to fix min_disparity / translation
    pass


@pytest.mark.parametrize("max_depth", [None, 3.0])
def test_composition_equals_the_one_pass_restatement(oracle, max_depth):
    W, H = 200, 120
    rec = synthetic.rig(W, H)
    stereo = ca.Stereo.load(rec)
    stereo.set_stereo_matching(_NoMatcher(), max_depth=max_depth)
    cfg = dict(max_size=W, minDisparity=1, numDisparities=32, blockSize=5, P1=200, P2=800, disp12MaxDiff=1,
               uniquenessRatio=5, speckleWindowSize=30, speckleRange=2)
    img1, img2, _ = synthetic.render_plane_pair(rec, (0.2, 0.1, 1.0), 1.5)
    ref = oracle_get_depth(oracle, stereo, cfg, img1, img2)
    r1, r2, mask = rectified_pair(oracle, stereo, img1, img2)
    assert np.array_equal(mask, stereo.rectify_valid_mask1)
    assert np.array_equal(r1, ref["rectify_img1"]) and np.array_equal(r2, ref["rectify_img2"])
    disp16 = oracle.sgbm_compute(r1, r2, **sgbm_params(cfg))
    disparity, depth = oracle.disp_to_depth(disp16, mask, cfg["minDisparity"], stereo.min_disparity,
                                            stereo.translation_rectify_img, 1.0 * stereo.baseline * stereo.K[0, 0],
                                            stereo.get_max_depth())
    assert (depth > 0).mean() > 0.3
    assert ref["disparity"].dtype == np.float32 and np.array_equal(ref["disparity"], disparity)
    assert ref["rectify_depth"].dtype == np.float64 and np.array_equal(ref["rectify_depth"], depth)
    # max_size at or above the image: the matcher's resize calls are identities (boxx.resize)
    assert np.array_equal(matcher_disparity(oracle, dict(cfg, max_size=10 * W), r1, r2),
                          matcher_disparity(oracle, cfg, r1, r2))
    # and below it the disparity is computed on the downsized pair and scaled back by w / sw
    small = matcher_disparity(oracle, dict(cfg, max_size=W // 2), r1, r2)
    assert small.shape == (H, W) and small.dtype == np.float32
    both = (small > 0) & (ref["disparity"] > 0) if not stereo.translation_rectify_img else None
    if both is not None and both.mean() > 0.2:
        assert np.median(np.abs(small[both] - matcher_disparity(oracle, cfg, r1, r2)[both])) < 2.0


def test_sgbm_params_are_the_reference_defaults():
    p = sgbm_params({})
    assert p == dict(minDisparity=2, numDisparities=218, blockSize=11, uniquenessRatio=5, speckleWindowSize=200,
                     speckleRange=2, disp12MaxDiff=0, P1=8 * 121, P2=32 * 121)
    q = sgbm_params(dict(blockSize=5, max_size=640, numDisparities=64))
    assert q["P1"] == 200 and q["P2"] == 800 and q["numDisparities"] == 64 and "max_size" not in q
