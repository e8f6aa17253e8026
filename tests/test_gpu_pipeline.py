"""GPU parity of the remap / filter / depth kernels and of Stereo.get_depth end to end (-m gpu).

Bars: remaps, median, speckle and masks bit-exact vs the CPU oracle; depth within 1e-4 m
(BASELINE.json) -- in fact identical, both sides round every float op the same way.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import calibrating_amd as ca  # noqa: E402
from calibrating_amd import imgproc, synthetic  # noqa: E402

DEPTH_TOL = 1e-4  # metres, BASELINE.json north_star


def _maps(rng, h, w, sw, sh, spread=6.0):
    yy, xx = np.mgrid[:h, :w].astype(np.float32)
    mapx = xx * (sw / w) + rng.uniform(-spread, spread, (h, w)).astype(np.float32)
    mapy = yy * (sh / h) + rng.uniform(-spread, spread, (h, w)).astype(np.float32)
    return mapx.astype(np.float32), mapy.astype(np.float32)


@pytest.mark.parametrize("cn", [1, 3])
@pytest.mark.parametrize("interp", [imgproc.INTER_LANCZOS4, imgproc.INTER_LINEAR, imgproc.INTER_NEAREST])
def test_remap_bit_exact(oracle, cn, interp):
    rng = np.random.default_rng(cn * 10 + interp)
    sh, sw, dh, dw = 70, 93, 64, 120
    src = rng.integers(0, 256, (sh, sw, cn) if cn > 1 else (sh, sw), dtype=np.uint8)
    mapx, mapy = _maps(rng, dh, dw, sw, sh)  # includes taps crossing and fully outside the border
    # exact .5 coordinates (round-half-even) and integer hits
    mapx[0, :10] = np.arange(10) + 0.5
    mapy[0, :10] = 3.5
    mapx[1, :10] = np.arange(10)
    mapy[1, :10] = 7
    mapx[2, :4] = [-20, sw + 20, 5, 5]
    mapy[2, :4] = [5, 5, -20, sh + 20]
    got = imgproc.remap(src, mapx, mapy, interp)
    ref = oracle.remap_u8(src, mapx, mapy, interp)
    assert got.shape == ref.shape
    assert np.array_equal(got, ref), "max |d| = %d" % np.abs(got.astype(int) - ref).max()


@pytest.mark.parametrize("cn", [1, 3])
def test_remap_borders_unaligned_base_and_padded_pitch(oracle, cn):
    """Windows hanging over every edge and corner, with the image at an odd address inside a 0xFF-filled buffer and
    rows padded with 0xFF: whatever the dword fetches pick up outside a row must be masked to the border constant."""
    from calibrating_amd import _native
    rng = np.random.default_rng(40 + cn)
    sh, sw, dh, dw = 37, 53, 48, 200
    img = rng.integers(0, 256, (sh, sw, cn), dtype=np.uint8)
    pitch, lead = sw * cn + 5, 13
    buf = np.full(lead + sh * pitch + 64, 255, np.uint8)
    rows = np.lib.stride_tricks.as_strided(buf[lead:], (sh, sw * cn), (pitch, 1))
    rows[:] = img.reshape(sh, sw * cn)
    # a map that sweeps from 6 px outside to 6 px outside in both axes at fractional steps (all four corners)
    mapx = np.tile(np.linspace(-6.3, sw + 5.7, dw, dtype=np.float32), (dh, 1))
    mapy = np.tile(np.linspace(-6.6, sh + 5.4, dh, dtype=np.float32)[:, None], (1, dw))
    mapx += rng.uniform(-0.4, 0.4, mapx.shape).astype(np.float32)
    d_buf = torch.from_numpy(buf).cuda()
    mx, my = torch.from_numpy(mapx).cuda(), torch.from_numpy(mapy).cuda()
    for interp in (imgproc.INTER_LANCZOS4, imgproc.INTER_LINEAR):
        out = torch.empty((dh, dw, cn), dtype=torch.uint8, device="cuda")
        rc = _native.lib().camd_remap_u8(d_buf.data_ptr() + lead, sw, sh, cn, pitch, sh * pitch, mx.data_ptr(),
                                         my.data_ptr(), out.data_ptr(), dw, dh, dw * cn, dh * dw * cn, interp, 0, 1,
                                         _native.current_stream())
        _native.check(rc, "remap")
        ref = oracle.remap_u8(img if cn > 1 else img[..., 0], mapx, mapy, interp).reshape(dh, dw, cn)
        assert np.array_equal(out.cpu().numpy(), ref)


def test_remap_identity_and_shift(oracle):
    rng = np.random.default_rng(1)
    src = rng.integers(0, 256, (40, 60, 3), dtype=np.uint8)
    yy, xx = np.mgrid[:40, :60].astype(np.float32)
    got = imgproc.remap(src, xx, yy, imgproc.INTER_LANCZOS4)
    assert np.array_equal(got, src)  # integer coordinates: Lanczos-4 is the identity
    # x_shift == the reference's translation of rectify_img2 (stereo_camera.py:230-240)
    for shift in (7, -5):
        g = imgproc.remap(src, xx, yy, imgproc.INTER_LANCZOS4, x_shift=shift)
        want = src.copy()
        if shift > 0:
            want[:, shift:] = src[:, :-shift]
            want[:, :shift] = 0
        else:
            want[:, :shift] = src[:, -shift:]
            want[:, shift:] = 0
        assert np.array_equal(g, want)


def test_remap_batched_tensor_input():
    rng = np.random.default_rng(2)
    src = torch.from_numpy(rng.integers(0, 256, (3, 30, 50, 3), dtype=np.uint8)).cuda()
    mapx, mapy = _maps(rng, 30, 50, 50, 30, 2.0)
    out = imgproc.remap(src, torch.from_numpy(mapx).cuda(), torch.from_numpy(mapy).cuda())
    assert out.shape == (3, 30, 50, 3) and out.is_cuda
    for i in range(3):
        assert np.array_equal(out[i].cpu().numpy(), imgproc.remap(src[i].cpu().numpy(), mapx, mapy))


@pytest.mark.parametrize("cn", [1, 3])
@pytest.mark.parametrize("n", [2, 17, 35])
def test_batch_inner_kernels_against_the_oracle_per_image(oracle, cn, n):
    """The images of a batch share the rig's maps: one workgroup applies a destination pixel's map, phase and weight
    entry to up to 16 images (remap.hip / depth.hip).  Every image of a batch -- 17 and 35 leave ragged last groups --
    must equal the oracle's result for that image alone: interior waves, waves crossing the border, the zero-filled
    columns of an x-shift, distinct content per image."""
    rng = np.random.default_rng(100 * cn + n)
    sh, sw, dh, dw = 61, 333, 40, 300                      # 300 columns: a full 256-wide workgroup and a ragged one
    src = rng.integers(0, 256, (n, sh, sw, cn), dtype=np.uint8)  # (a gray batch is (n, h, w, 1): 3-D means one (h, w, c) image)
    one = (lambda a: a) if cn > 1 else (lambda a: a[..., 0])
    mapx, mapy = _maps(rng, dh, dw, sw, sh, 1.5)           # mostly interior (the batch-inner fast path) ...
    mapx[:, :20] -= 12.0                                   # ... with windows hanging over the left border,
    mapy[-3:, :] += 9.0                                    # over the bottom border,
    mapx[5, 100:110] = sw + 50.0                           # and entirely outside
    d_src = torch.from_numpy(src).cuda()
    d_mx, d_my = torch.from_numpy(mapx).cuda(), torch.from_numpy(mapy).cuda()
    for interp, shift in ((imgproc.INTER_LANCZOS4, 0), (imgproc.INTER_LANCZOS4, 7), (imgproc.INTER_LINEAR, 0)):
        got = imgproc.remap(d_src, d_mx, d_my, interp, x_shift=shift).cpu().numpy()
        for i in range(n):
            ref = oracle.remap_u8(one(src[i]), mapx, mapy, interp)
            if shift:
                ref = np.concatenate([np.zeros_like(ref[:, :shift]), ref[:, :-shift]], axis=1)
            assert np.array_equal(one(got[i]), ref), (interp, shift, i)
    # cv2.undistort's fixed-point bilinear remap, batched
    rig = synthetic.rig(sw, sh)
    K, D = np.array(rig["cam1"]["K"]), np.array(rig["cam1"]["D"])
    mxy, ma = imgproc.undistort_maps(K, D, (sw, sh))
    got = imgproc.remap_fixed_bilinear(d_src, torch.from_numpy(mxy).cuda(), torch.from_numpy(ma.view(np.int16)).cuda()).cpu().numpy()
    for i in range(0, n, 5):
        assert np.array_equal(one(got[i]), oracle.undistort_u8(one(src[i]), K, D)), i
    # unrectify_depth, batched
    depth = rng.uniform(0, 5, (n, sh, sw))
    M = np.array([0.0123, -0.0045, 0.9991])
    umx, umy = _maps(rng, dh, dw, sw, sh, 3.0)
    got = imgproc.unrectify_depth(torch.from_numpy(depth).cuda(), M, torch.from_numpy(umx).cuda(),
                                  torch.from_numpy(umy).cuda()).cpu().numpy()
    for i in range(0, n, 4):
        assert np.array_equal(got[i], oracle.unrectify_depth(depth[i], M, umx, umy)), i


def test_undistort_bit_exact(oracle):
    rig = synthetic.rig(320, 240)
    K, D = np.array(rig["cam1"]["K"]), np.array(rig["cam1"]["D"])
    img = synthetic.scene_pair(3, 320, 240, 3)[0]
    mxy, ma = imgproc.undistort_maps(K, D, (320, 240))
    got = imgproc.remap_fixed_bilinear(img, mxy, ma)
    ref = oracle.undistort_u8(img, K, D)
    assert np.array_equal(got, ref)
    gray = np.ascontiguousarray(img[..., 0])
    assert np.array_equal(imgproc.remap_fixed_bilinear(gray, mxy, ma), oracle.undistort_u8(gray, K, D))


def test_median_and_speckle_bit_exact(oracle):
    rng = np.random.default_rng(4)
    img = (rng.integers(-2, 60, (57, 83)) * 16).astype(np.int16)
    img[rng.random(img.shape) < 0.2] = -16
    assert np.array_equal(imgproc.medianBlur3_s16(img), oracle.median3_s16(img))
    one_row = img[:1].copy()
    assert np.array_equal(imgproc.medianBlur3_s16(one_row), oracle.median3_s16(one_row))
    for max_size, max_diff in ((5, 16), (40, 32), (400, 0)):
        got = imgproc.filterSpeckles(img, -16, max_size, max_diff)
        assert np.array_equal(got, oracle.filter_speckles_s16(img, -16, max_size, max_diff))
    # large smooth regions with noise speckles, full-HD rows
    big = np.full((128, 1920), 16 * 20, np.int16)
    big[:, 900:] = 16 * 40
    spk = rng.random(big.shape) < 0.02
    big[spk] = (rng.integers(0, 100, spk.sum()) * 16).astype(np.int16)
    got = imgproc.filterSpeckles(big, -16, 200, 32)
    assert np.array_equal(got, oracle.filter_speckles_s16(big, -16, 200, 32))


def test_disp_to_depth_and_unrectify(oracle):
    rng = np.random.default_rng(5)
    h, w = 50, 70
    disp16 = rng.integers(-16, 128 * 16, (h, w)).astype(np.int16)
    disp16[0, :6] = [0, -16, 1, 31, 32, 264]  # d in {0, invalid, tiny, below minD*16, minD*16, 16.5}
    mask = rng.random((h, w)) < 0.9
    for translate, add in ((False, 0), (True, 13)):
        disparity, depth = imgproc.disp_to_depth(disp16, mask, 2, add, translate, 0.12 * 1536.0, 3.5)
        rd, rz = oracle.disp_to_depth(disp16, mask, 2, add, translate, 0.12 * 1536.0, 3.5)
        assert disparity.dtype == np.float32 and depth.dtype == np.float64
        assert np.array_equal(disparity, rd)
        assert np.array_equal(depth == 0, rz == 0)
        assert np.abs(depth - rz).max() <= DEPTH_TOL
    # odd pixel counts, several images: the kernels work on pixel pairs (the last pixel of an image stands alone, the
    # second image starts on an odd element)
    d3 = rng.integers(-16, 128 * 16, (3, 51, 71)).astype(np.int16)
    m3 = rng.random((51, 71)) < 0.9
    disparity, depth = imgproc.disp_to_depth(torch.from_numpy(d3).cuda(), torch.from_numpy(m3.view(np.uint8)).cuda(), 2, 13,
                                             True, 0.12 * 1536.0, 3.5)
    for i in range(3):
        rd, rz = oracle.disp_to_depth(d3[i], m3, 2, 13, True, 0.12 * 1536.0, 3.5)
        assert np.array_equal(disparity[i].cpu().numpy(), rd) and np.array_equal(depth[i].cpu().numpy(), rz)
    odd = (rng.integers(-2, 60, (2, 33, 41)) * 16).astype(np.int16)
    got = imgproc.medianBlur3_s16(torch.from_numpy(odd).cuda()).cpu().numpy()
    for i in range(2):
        assert np.array_equal(got[i], oracle.median3_s16(odd[i]))
    depth = rng.uniform(0, 5, (h, w))
    mapx, mapy = _maps(rng, 44, 66, w, h, 3.0)
    M = np.array([0.0123, -0.0045, 0.9991])
    got = imgproc.unrectify_depth(depth, M, mapx, mapy)
    ref = oracle.unrectify_depth(depth, M, mapx, mapy)
    assert np.array_equal(got == 0, ref == 0)
    assert np.abs(got - ref).max() <= DEPTH_TOL


def _oracle_get_depth(oracle, stereo, sgbm_params, img1, img2):
    """The reference's get_depth (stereo_camera.py:492-533) composed from oracle stages at the matcher's full
    resolution (tests/oracle_pipeline.py holds the composition, incl. the max_size downsizing branch)."""
    from oracle_pipeline import oracle_get_depth
    return oracle_get_depth(oracle, stereo, dict(sgbm_params, max_size=1 << 30), img1, img2)


@pytest.mark.parametrize("W,H,max_depth", [(640, 480, None), (640, 480, 3.5), (320, 240, 3.0), (1920, 1080, 3.5)])
def test_get_depth_end_to_end(oracle, W, H, max_depth):
    """Config C5-like: full get_depth (rectify x2 + SGBM + disp_to_depth + unrectify + undistort); the last case is a
    whole 1080p pair at D=128 (about ten seconds of oracle)."""
    stereo = ca.Stereo.load(synthetic.rig(W, H))
    cfg = dict(max_size=max(W, H), minDisparity=0, numDisparities=128 if W > 1000 else 64, blockSize=5, P1=8 * 3 * 25,
               P2=32 * 3 * 25, disp12MaxDiff=1, uniquenessRatio=10, speckleWindowSize=100, speckleRange=2)
    stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(cfg), max_depth=max_depth)
    # a rendered slanted plane 2 m away (synthetic.render_plane_pair): ~90 % of the image carries a valid depth, so the
    # depth / unrectify / speckle stages are compared on real values, not on zeros
    img1, img2, _ = synthetic.render_plane_pair(synthetic.rig(W, H), (0.3, 0.1, 1.0), 2.0)
    got = stereo.get_depth(img1, img2)
    sp = {k: v for k, v in cfg.items() if k != "max_size"}
    ref = _oracle_get_depth(oracle, stereo, sp, img1, img2)
    assert set(ref) <= set(got)
    for k in ("rectify_img1", "rectify_img2", "undistort_img1"):
        assert got[k].dtype == np.uint8 and np.array_equal(got[k], ref[k]), k
    assert got["disparity"].dtype == np.float32 and np.array_equal(got["disparity"], ref["disparity"])
    for k in ("rectify_depth", "unrectify_depth"):
        assert got[k].dtype == np.float64
        assert np.array_equal(got[k] == 0, ref[k] == 0), k
        assert np.abs(got[k] - ref[k]).max() <= DEPTH_TOL, k  # the stated bar (north_star: 1e-4 m) ...
        assert np.array_equal(got[k], ref[k]), k              # ... and what is actually reached: the same float64 bits
    assert (got["rectify_depth"] > 0).mean() > 0.7 and (got["unrectify_depth"] > 0).mean() > 0.7
    # tensors in -> tensors out, same numbers
    gt = stereo.get_depth(torch.from_numpy(img1).cuda(), torch.from_numpy(img2).cuda())
    assert gt["unrectify_depth"].is_cuda
    assert np.array_equal(gt["unrectify_depth"].cpu().numpy(), got["unrectify_depth"])


@pytest.mark.parametrize("plane", ["slanted", "fronto"])
@pytest.mark.parametrize("W,H", [(640, 480), (1280, 720)])
def test_get_depth_recovers_a_rendered_plane(plane, W, H):
    """Oracle-independent: a textured plane ray-cast through the Brown model of both cameras, get_depth on the GPU,
    depth against the geometric truth in the rectified and in camera 1's ideal frame (tests/ground_truth.py; the
    reference's own accuracy check: /root/reference/example/test_depth_accuracy.py:101-108)."""
    import ground_truth as gt
    normal, dist, eps, frac = gt.PLANES[plane]
    rec = synthetic.rig(W, H)
    img1, img2, z_true = synthetic.render_plane_pair(rec, normal, dist)
    stereo = ca.Stereo.load(rec)
    cfg = dict(gt.CFG, max_size=max(W, H), numDisparities=64 if W <= 640 else 128)
    stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(cfg), max_depth=gt.MAX_DEPTH)
    res = stereo.get_depth(img1, img2)
    b, fx = float(stereo.baseline), float(stereo.K[0, 0])
    gt.check_depth(res["rectify_depth"], gt.rectified_truth(stereo.K, stereo.R1, normal, dist, stereo.xy), b, fx, eps,
                   frac, "rectify_depth (%s)" % plane)
    cov, within, bias = gt.check_depth(res["unrectify_depth"], z_true, b, fx, eps, frac, "unrectify_depth (%s)" % plane)
    if plane == "slanted":
        assert abs(bias) <= 0.05, "mean signed disparity error %.3f px: a convention is off somewhere" % bias
    # the batch entry point sees the same physics
    rb = stereo.get_depth_batch(np.stack([img1, img1]), np.stack([img2, img2]))
    assert np.array_equal(np.asarray(rb["unrectify_depth"][1]), res["unrectify_depth"])


def test_get_depth_reference_default_matcher(oracle):
    """Config C1 plumbing: reference-default matcher parameters (stereo_matching.py:30-58)."""
    W, H = 640, 360
    stereo = ca.Stereo.load(synthetic.rig(W, H))
    stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(dict(max_size=W)), max_depth=3.5)
    img1, img2 = synthetic.scene_pair(21, W, H, 3)
    got = stereo.get_depth(img1, img2)
    sp = dict(minDisparity=2, numDisparities=218, blockSize=11, uniquenessRatio=5, speckleWindowSize=200,
              speckleRange=2, disp12MaxDiff=0, P1=8 * 121, P2=32 * 121)
    ref = _oracle_get_depth(oracle, stereo, sp, img1, img2)
    assert np.array_equal(got["disparity"], ref["disparity"])
    assert np.abs(got["unrectify_depth"] - ref["unrectify_depth"]).max() <= DEPTH_TOL


def test_get_depth_requires_matcher():
    stereo = ca.Stereo.load(synthetic.rig(64, 48))
    with pytest.raises(AssertionError, match="set_stereo_matching"):
        stereo.get_depth(np.zeros((48, 64, 3), np.uint8), np.zeros((48, 64, 3), np.uint8))
    with pytest.raises(NotImplementedError):
        ca.MetaStereoMatching()(None, None)


@pytest.mark.parametrize("max_size,max_depth", [(320, 3.5), (200, 3.5), (250, None)])
def test_get_depth_batch_matches_per_pair(max_size, max_depth):
    """The batched throughput form returns, pair by pair, exactly what get_depth returns -- at the matcher's full
    resolution (fused depth kernel) and through its max_size downsizing (the reference's default: resize, match,
    resize back), with and without the min_disparity translation."""
    W, H = 320, 240
    stereo = ca.Stereo.load(synthetic.rig(W, H))
    cfg = dict(max_size=max_size, minDisparity=0, numDisparities=64, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1,
               uniquenessRatio=10, speckleWindowSize=50, speckleRange=2)
    stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(cfg), max_depth=max_depth)
    pairs = [synthetic.scene_pair(s, W, H, 3) for s in (1, 2, 3, 4, 5)]
    I1 = np.stack([p[0] for p in pairs]); I2 = np.stack([p[1] for p in pairs])
    got = stereo.get_depth_batch(I1, I2)
    for i, (a, b) in enumerate(pairs):
        ref = stereo.get_depth(a, b)
        assert set(ref) == set(got)
        for k in ref:
            assert np.array_equal(got[k][i], ref[k], equal_nan=True), (i, k)
    with pytest.raises(ValueError):
        stereo.get_depth_batch(I1[0], I2[0])


def test_tables_built_on_gpu_are_bit_identical(oracle):
    """n2: camd_init_undistort_rectify_map / camd_undistort_maps against the oracle and the host construction."""
    from calibrating_amd import geometry
    rng = np.random.default_rng(5)
    for (w, h), dist in (((320, 240), [-0.21, 0.07, 1e-3, -2e-3, 0.01]),
                         ((1300, 70), [0.1, -0.05, 5e-4, 3e-4, 0.002, 0.03, -0.01, 0.004, 1e-3, -2e-3, 5e-4, 1e-4]),
                         ((97, 33), None)):
        K = np.array([[0.9 * w, 0, w / 2 + 3.3], [0, 0.95 * w, h / 2 - 1.7], [0, 0, 1]])
        Kn = np.array([[0.8 * w, 0, w / 2], [0, 0.8 * w, h / 2], [0, 0, 1]])
        R = geometry.rodrigues(rng.normal(0, 0.03, 3))
        mx, my, mask = imgproc.init_undistort_rectify_map(K, dist, R, Kn, (w, h), valid_for=(w, h))
        ox, oy = oracle.init_undistort_rectify_map(K, dist, R, Kn, (w, h))
        assert np.array_equal(mx.cpu().numpy(), ox) and np.array_equal(my.cpu().numpy(), oy)
        gx, gy = geometry.init_undistort_rectify_map(K, dist, R, Kn, (w, h))
        assert np.array_equal(mx.cpu().numpy(), gx) and np.array_equal(my.cpu().numpy(), gy)
        ref_mask = (-0.5 < ox) & (ox < w - 0.5) & (-0.5 < oy) & (oy < h - 0.5)
        assert np.array_equal(mask.cpu().numpy().astype(bool), ref_mask)
        # R = None (identity) and the fixed-point maps of cv2.undistort
        mx, my = imgproc.init_undistort_rectify_map(K, dist, None, K, (w, h))
        ox, oy = oracle.init_undistort_rectify_map(K, dist, None, K, (w, h))
        assert np.array_equal(mx.cpu().numpy(), ox) and np.array_equal(my.cpu().numpy(), oy)
        hxy, ha = imgproc.undistort_maps(K, dist, (w, h))
        dxy, da = imgproc.undistort_maps_device(K, dist, (w, h))
        assert np.array_equal(dxy.cpu().numpy(), hxy) and np.array_equal(da.cpu().numpy().view(np.uint16), ha)


def test_stereo_device_tables_match_host_properties():
    W, H = 320, 240
    stereo = ca.Stereo.load(synthetic.rig(W, H))
    tb = stereo._tables(torch.device("cuda", 0))
    assert np.array_equal(tb["map1x"].cpu().numpy(), stereo.undistort_rectify_map1[0])
    assert np.array_equal(tb["map1y"].cpu().numpy(), stereo.undistort_rectify_map1[1])
    assert np.array_equal(tb["map2x"].cpu().numpy(), stereo.undistort_rectify_map2[0])
    assert np.array_equal(tb["map2y"].cpu().numpy(), stereo.undistort_rectify_map2[1])
    assert np.array_equal(tb["mask"].cpu().numpy().astype(bool), stereo.rectify_valid_mask1)


def test_disparity_to_depth_tensor_branch_divides_like_numpy():
    """Stereo.disparity_to_depth on a CUDA tensor (the branch behind foreign plugins and the downsizing matcher) gives
    bit for bit what the reference's NumPy line gives: one IEEE division, zeros for inf / beyond max_depth / negatives."""
    stereo = ca.Stereo.load(synthetic.rig(320, 240))
    stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(dict(max_size=320)), max_depth=3.5)
    rng = np.random.default_rng(8)
    disp = rng.uniform(-3, 120, (240, 320)).astype(np.float32)
    disp[::9, ::7] = 0
    want = stereo.disparity_to_depth(disp.copy())
    got = stereo.disparity_to_depth(torch.from_numpy(disp).cuda()).cpu().numpy()
    assert got.dtype == want.dtype == np.float64 and np.array_equal(got, want)
