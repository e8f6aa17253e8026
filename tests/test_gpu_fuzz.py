"""Bounded, seeded slices of the three fuzzers (tests/fuzzers.py) inside the driver's GPU suite (-m gpu): the same
generators tools/gpu_fuzz*.py run by the thousand by hand.  Each asserts zero mismatches AND that the slice reached
the branches it exists for (so a silently skipped branch fails the test instead of passing it)."""
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import fuzzers  # noqa: E402


def _run(fn, n, seeds, need):
    lines, total, inexact = [], {}, {}
    for seed in seeds:
        res = fn(n, seed, log=lambda *a: lines.append(" ".join(str(x) for x in a)))
        fuzzers.report(res)
        assert not res["mismatches"], "\n".join(lines[:10])
        for k, v in res["branches"].items():
            total[k] = total.get(k, 0) + v
        for k, v in res.get("within_tolerance_but_not_bit_identical", {}).items():
            inexact[k] = inexact.get(k, 0) + v
    for branch, at_least in need.items():
        assert total.get(branch, 0) >= at_least, (branch, total)
    assert not inexact, inexact


def test_fuzz_sgbm_slice(oracle):
    _run(fuzzers.fuzz_sgbm, 300, (401, 402, 403), dict(mode0=100, mode1=100, mode2=100, mode3=100, cost0=100, cost1=400,
                                                       cost2=400, batched=200, left_u16_regime=3, rgb_drift_input=50,
                                                       gray_drift_input=50,
                                                       # every lane shape of the volume layout (16 lanes x 3..8 / 12 registers)
                                                       Dp96=40, Dp160=40, Dp192=40, Dp224=80, Dp256=20, Dp384=20))


def test_fuzz_remap_slice(oracle):
    _run(fuzzers.fuzz_remap, 150, (411, 412), dict(lanczos4=300, linear=300, nearest=300, shifted=250, fold=70, noise=70,
                                                   magnify=70, warp=70))


def test_fuzz_pipeline_slice(oracle):
    _run(fuzzers.fuzz_pipeline, 100, (421, 422), dict(downsizing=60, full_resolution=30, hetero_rig=30, translated=60,
                                                     untranslated=30, over_25pct_valid_depth=50))


def test_fuzz_speckle_slice(oracle):
    _run(fuzzers.fuzz_speckle, 200, (431, 432), dict(smooth=100, noise=100, serpentine=100, comb=100, batched=150,
                                                    images_with_erased_pixels=150))


def test_speckle_workspace_of_a_handle_survives_many_pairs(oracle):
    """The speckle kernels keep their union-find scratch clean from call to call instead of initialising it per call:
    one StereoSGBM handle, different pairs (and batch sizes) one after the other, every result against the oracle."""
    import numpy as np
    import calibrating_amd as ca
    from calibrating_amd import synthetic
    p = dict(minDisparity=0, numDisparities=32, blockSize=3, P1=24, P2=96, disp12MaxDiff=1, uniquenessRatio=5,
             speckleWindowSize=60, speckleRange=1)
    m = ca.StereoSGBM_create(**p)
    pairs = [synthetic.rectified_pair(seed=s, H=70, W=200, D=32, cn=1) for s in (1, 2, 3)]
    pairs.append((np.random.default_rng(0).integers(0, 256, (70, 200), dtype=np.uint8),) * 2)
    want = [oracle.sgbm_compute(a, b, **p) for a, b in pairs]
    assert any((w_ != oracle.sgbm_compute(a, b, **dict(p, speckleWindowSize=0))).any() for w_, (a, b) in zip(want, pairs))
    for order in ((0, 1, 2, 3), (3, 2, 1, 0), (1, 1, 3, 0)):
        for i in order:
            assert np.array_equal(m.compute(*pairs[i]), want[i]), (order, i)
        got = m.compute(np.stack([pairs[i][0] for i in order]), np.stack([pairs[i][1] for i in order]))
        for k, i in enumerate(order):
            assert np.array_equal(got[k], want[i]), (order, k)


@pytest.mark.parametrize("h,w,nb", [(1080, 1920, 3), (2160, 3840, 1), (1081, 1923, 2)])
def test_speckle_full_size(oracle, h, w, nb):
    """The speckle filter at BASELINE's image sizes (and an odd one): piecewise-smooth disparity-like images with noise
    speckles, holes and one long thin structure, against cv2.filterSpeckles' flood fill restated in the oracle."""
    import numpy as np
    from calibrating_amd import imgproc
    rng = np.random.default_rng(h + w)
    imgs = []
    for i in range(nb):
        img = fuzzers.speckle_image(rng, 0, h, w, -16)
        img[h // 3, :] = 7000          # a one-pixel line across every segment
        img[:, w // 2] = 7000          # ... and down every strip
        img[h // 3 + 2:h // 3 + 40:2, ::3] = 9000  # a comb of isolated pixels (size-1 components)
        imgs.append(img)
    imgs = np.stack(imgs)
    got = imgproc.filterSpeckles(imgs, -16, 200, 32)
    for i in range(nb):
        want = oracle.filter_speckles_s16(imgs[i], -16, 200, 32)
        assert (want != imgs[i]).any()
        assert np.array_equal(got[i], want), (i, int((got[i] != want).sum()))
