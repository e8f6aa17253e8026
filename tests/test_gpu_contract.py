"""The rest of `Stereo.get_depth`'s contract (-m gpu): branches of the reference's orchestration
(/root/reference/calibrating/stereo_camera.py:492-533) that the headline configs do not reach -- foreign
MetaStereoMatching plugins (array or dict results, in-place `+= min_disparity`), rectified sizes that differ from
the source size (xy_target / K_target), image-file inputs, and the U7 saturate / wrap switch of the cost volume."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import calibrating_amd as ca  # noqa: E402
from calibrating_amd import synthetic  # noqa: E402

from test_gpu_pipeline import DEPTH_TOL, _oracle_get_depth  # noqa: E402


class _ForeignPlugin(ca.MetaStereoMatching):
    """A matcher written against the reference's plugin surface only: NumPy RGB in, float32 disparity (or a dict
    holding it) out.  Keeps what it saw and what it returned for the assertions."""

    def __init__(self, cfg=None):
        super().__init__(cfg)
        self.returned = None

    def __call__(self, img1, img2):
        assert isinstance(img1, np.ndarray) and img1.dtype == np.uint8 and img1.ndim == 3 and img1.shape == img2.shape
        self.seen = (img1.copy(), img2.copy())
        h, w = img1.shape[:2]
        disp = (5.0 + 20.0 * (np.mgrid[:h, :w][1] / w)).astype(np.float32)
        disp[::7, ::5] = 0  # holes: depth must come out 0 there only when nothing is added back
        self.returned = disp
        if self.cfg and self.cfg.get("as_dict"):
            return dict(disparity=disp, confidence=np.ones((h, w), np.float32) * 0.5, note="extra keys are merged")
        return disp


@pytest.mark.parametrize("as_dict", [False, True])
@pytest.mark.parametrize("max_depth", [None, 3.5])
def test_foreign_plugin_contract(oracle, as_dict, max_depth):
    W, H = 320, 240
    stereo = ca.Stereo.load(synthetic.rig(W, H))
    plugin = _ForeignPlugin(dict(as_dict=as_dict))
    stereo.set_stereo_matching(plugin, max_depth=max_depth)
    img1, img2 = synthetic.scene_pair(11, W, H, 3)
    got = stereo.get_depth(img1, img2)
    # the plugin saw the rectified (and, with max_depth, translated) NumPy images
    assert np.array_equal(plugin.seen[0], got["rectify_img1"]) and np.array_equal(plugin.seen[1], got["rectify_img2"])
    # reference :510-513 -- `disparity += min_disparity` happens IN PLACE on the plugin's own array, then * mask
    h, w = img1.shape[:2]
    base = (5.0 + 20.0 * (np.mgrid[:h, :w][1] / w)).astype(np.float32)
    base[::7, ::5] = 0
    shift = stereo.min_disparity if max_depth else 0
    assert stereo.translation_rectify_img == bool(max_depth)
    assert np.array_equal(plugin.returned, base + np.float32(shift)), "in-place += visible on the plugin's array"
    want_disp = stereo.rectify_valid_mask1 * (base + np.float32(shift))
    assert np.array_equal(got["disparity"], want_disp)
    with np.errstate(divide="ignore"):
        want_depth = 1.0 * stereo.baseline * stereo.K[0, 0] / want_disp
    want_depth[want_depth > stereo.get_max_depth()] = 0
    want_depth[want_depth < 0] = 0
    assert np.abs(got["rectify_depth"] - want_depth).max() <= DEPTH_TOL
    if as_dict:
        assert got["note"] == "extra keys are merged" and got["confidence"].shape == (H, W)
    else:
        assert "note" not in got
    for k in ("unrectify_depth", "undistort_img1"):
        assert k in got
    assert set(stereo.get_depth(img1, img2, return_unrectify_depth=False)) >= {"rectify_img1", "rectify_depth", "disparity",
                                                                              "rectify_img2"}


class _PostFilteringSGBM(ca.SemiGlobalBlockMatching):
    """A user's subclass of the SGBM plugin that overrides ``__call__``: returns a dict with an extra key and its own
    post-filter.  The reference always goes through ``stereo_matching(img1, img2)`` (stereo_camera.py:506-509), so the
    override must be honoured at full resolution AND in the max_size downsizing branch (no fused shortcut around it)."""

    def __call__(self, img1, img2):
        disparity = super().__call__(img1, img2)
        assert isinstance(disparity, np.ndarray), "the plugin contract is NumPy in, NumPy out"
        disparity[disparity > 15] = 0
        return dict(disparity=disparity, filtered_by="subclass")


@pytest.mark.parametrize("max_size", [320, 200])
def test_sgbm_subclass_override_is_honoured(oracle, max_size):
    from oracle_pipeline import matcher_disparity, rectified_pair
    W, H = 320, 240
    rec = synthetic.rig(W, H)
    stereo = ca.Stereo.load(rec)
    cfg = dict(max_size=max_size, minDisparity=0, numDisparities=64, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1,
               uniquenessRatio=10, speckleWindowSize=0)
    stereo.set_stereo_matching(_PostFilteringSGBM(cfg), max_depth=None)
    img1, img2, _ = synthetic.render_plane_pair(rec, (0.3, 0.1, 1.0), 2.0)
    got = stereo.get_depth(img1, img2)
    assert got["filtered_by"] == "subclass"
    r1, r2, mask = rectified_pair(oracle, stereo, img1, img2)
    want = matcher_disparity(oracle, cfg, r1, r2)
    assert (want > 15).mean() > 0.1 and (want > 0).mean() > 0.5
    want[want > 15] = 0
    assert np.array_equal(got["disparity"], mask * want)
    with pytest.raises(ValueError, match="own __call__"):
        stereo.get_depth_batch(np.stack([img1, img1]), np.stack([img2, img2]))


@pytest.mark.parametrize("xy_target,K_target", [(0.5, 0.5), ((400, 260), 1), (None, 0.8), (1.25, 1)])
def test_get_depth_with_resized_rectified_frame(oracle, xy_target, K_target):
    """Rectified size != source size (reference stereo_camera.py:125-156: xy_target / K_target), single and batched."""
    W, H = 480, 320
    rig = synthetic.rig(W, H)
    stereo = ca.Stereo(ca.Cam.load(rig["cam1"]), ca.Cam.load(rig["cam2"]), xy_target=xy_target, K_target=K_target,
                       R=np.array(rig["R"]), t=np.array(rig["t"]))
    Wt, Ht = stereo.xy
    if isinstance(xy_target, float):
        assert (Wt, Ht) == (int(round(W * xy_target)), int(round(H * xy_target)))
    elif xy_target is not None:
        assert (Wt, Ht) == tuple(xy_target)
    cfg = dict(max_size=max(Wt, Ht), minDisparity=0, numDisparities=48, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1,
               uniquenessRatio=10, speckleWindowSize=60, speckleRange=2)
    stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(cfg), max_depth=4.0)
    img1, img2 = synthetic.scene_pair(13, W, H, 3)
    got = stereo.get_depth(img1, img2)
    assert got["rectify_img1"].shape == (Ht, Wt, 3) and got["disparity"].shape == (Ht, Wt)
    assert got["unrectify_depth"].shape == (H, W) and got["undistort_img1"].shape == (H, W, 3)
    ref = _oracle_get_depth(oracle, stereo, {k: v for k, v in cfg.items() if k != "max_size"}, img1, img2)
    for k in ("rectify_img1", "rectify_img2", "undistort_img1", "disparity"):
        assert np.array_equal(got[k], ref[k]), k
    for k in ("rectify_depth", "unrectify_depth"):
        assert np.array_equal(got[k] == 0, ref[k] == 0), k
        assert np.abs(got[k] - ref[k]).max() <= DEPTH_TOL, k
    # the batched form gates max_size on the RECTIFIED size, like get_depth, and returns the same numbers
    gb = stereo.get_depth_batch(np.stack([img1, img1]), np.stack([img2, img2]))
    for k in got:
        assert np.array_equal(gb[k][1], got[k], equal_nan=True), k
    small = ca.SemiGlobalBlockMatching(dict(cfg, max_size=max(Wt, Ht) - 1))
    stereo.set_stereo_matching(small, max_depth=4.0)
    one = stereo.get_depth(img1, img2)  # one pixel too large for the matcher: both forms downsize, identically
    assert one["disparity"].shape == (Ht, Wt)
    gb = stereo.get_depth_batch(np.stack([img1]), np.stack([img2]))
    for k in one:
        assert np.array_equal(gb[k][0], one[k], equal_nan=True), k


def test_get_depth_reads_image_files(tmp_path):
    """`Stereo._get_img` accepts a path like the reference (stereo_camera.py:304-308)."""
    from PIL import Image
    W, H = 160, 120
    stereo = ca.Stereo.load(synthetic.rig(W, H))
    stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(dict(max_size=W, numDisparities=32, minDisparity=0, blockSize=5)),
                               max_depth=3.0)
    img1, img2 = synthetic.scene_pair(17, W, H, 3)
    p1, p2 = str(tmp_path / "l.png"), str(tmp_path / "r.png")
    Image.fromarray(img1).save(p1)
    Image.fromarray(img2).save(p2)
    a, b = stereo.get_depth(p1, p2), stereo.get_depth(img1, img2)
    for k in b:
        assert np.array_equal(a[k], b[k], equal_nan=True), k


def _c_volume(matcher, index=0):
    return matcher.debug_volume("C", index).cpu().numpy()


def _sawtooth_pair(H, W):
    """Opposite sawtooth ramps (slope +-16 per pixel, phase drifting with the row): every gradient plane of the left
    image sits at its upper clip, of the right image at its lower clip, and the raw planes are far apart -- the
    largest pixel costs images can produce."""
    x, y = np.arange(W)[None, :], np.arange(H)[:, None]
    ramp = ((x * 16 + y * 40) % 256).astype(np.uint8)
    left = ramp[..., None].repeat(3, 2)
    return left, 255 - left


@pytest.mark.parametrize("saturate", [1, 0])
@pytest.mark.parametrize("cost_path", [1, 2])
@pytest.mark.parametrize("cap", [31, 63])
def test_u7_saturate_switch_matches_oracle(oracle, saturate, cost_path, cap):
    """Block 11 x RGB with a raised preFilterCap on opposite sawtooth ramps drives window sums past 32767 (a few
    percent of the cells at cap 31, most at 63; at the default cap 0 even these images stay below 26000): the cost
    volume must follow OpenCV's CV_SIMD saturation (default) or the scalar build's wrap, as the switch says, in the
    fused kernel (cost_path 1) and in the split pair (2) alike."""
    H, W, D = 36, 260, 96
    left, right = _sawtooth_pair(H, W)
    p = dict(minDisparity=0, numDisparities=D, blockSize=11, P1=968, P2=3872, disp12MaxDiff=1, uniquenessRatio=5,
             preFilterCap=cap)
    try:
        oracle.set_switches(cost_saturate=saturate)
        want = oracle.sgbm_cost_volume(left, right, **p)
        m = ca.StereoSGBM_create(**p)
        m.set_option("saturate", saturate).set_option("cost", cost_path)
        disp = m.compute(left, right)
        got = _c_volume(m)
        assert np.array_equal(got, want), "C volume, saturate=%d: %d cells differ" % (saturate, (got != want).sum())
        if saturate:
            assert (want == 32767).any(), "the case must actually saturate"
            assert want.min() >= 0
            # with C inside [0, 32767] the aggregation stays in its exact regime: the disparity matches as well
            assert np.array_equal(disp, oracle.sgbm_compute(left, right, **p))
        else:
            assert (want < 0).any(), "the case must actually wrap"
    finally:
        oracle.set_switches()


def test_u7_two_stage_build_flags_only_the_saturating_pairs(oracle):
    """The saturating volume is built in two stages (sgbm_cost.hpp): row chunks in parallel with wrapping arithmetic,
    then the sequential saturating kernel for the volumes in which a value came close to 32767.  A batch that mixes
    ordinary pairs with a saturating one, tall enough for several row chunks: every pair must equal the oracle, cost
    volume and disparity."""
    H, W, D = 100, 260, 96
    saw = _sawtooth_pair(H, W)
    tex = synthetic.rectified_pair(seed=3, H=H, W=W, D=D, cn=3)
    tex2 = synthetic.rectified_pair(seed=4, H=H, W=W, D=D, cn=3)
    p = dict(minDisparity=0, numDisparities=D, blockSize=11, P1=968, P2=3872, disp12MaxDiff=1, uniquenessRatio=5,
             preFilterCap=31)
    pairs = [tex, saw, tex2, saw]
    m = ca.StereoSGBM_create(**p)
    got = m.compute(np.stack([a for a, _ in pairs]), np.stack([b for _, b in pairs]))
    vols = [oracle.sgbm_cost_volume(a, b, **p) for a, b in pairs]
    assert vols[0].max() < 32767 - 11 * 3 * 125 and (vols[1] == 32767).any()  # one kind of each
    for i, (a, b) in enumerate(pairs):
        assert np.array_equal(got[i], oracle.sgbm_compute(a, b, **p)), i
        assert np.array_equal(_c_volume(m, i), vols[i]), i


def test_cost_paths_switched_on_one_handle_with_padded_disparities(oracle):
    """ONE handle, the reference's default matcher parameters (block 11 x RGB: an int16 overflow is possible, so the
    below-P2 flags are live; D = 218 padded to 256), switched between the fused cost kernel, the split pair, the
    saturating and the wrapping build from call to call.  The padded disparities of the cost volume are expected to
    hold P2 whichever kernel wrote the volume last (k_cost's all-padding waves write nothing; k_flag_below no longer
    depends on it): no pair may be flagged, refused or come back different."""
    H, W, D = 40, 300, 218
    p = dict(minDisparity=2, numDisparities=D, blockSize=11, P1=968, P2=3872, disp12MaxDiff=0, uniquenessRatio=5,
             speckleWindowSize=200, speckleRange=2)
    a = synthetic.rectified_pair(seed=21, H=H, W=W + 60, D=64, cn=3)
    b = synthetic.rectified_pair(seed=22, H=H, W=W + 60, D=64, cn=3)
    want = [oracle.sgbm_compute(l, r, **p) for l, r in (a, b)]
    m = ca.StereoSGBM_create(**p)
    m.set_option("exact", 0)  # a wrongly raised flag must surface as a refusal, not be repaired silently
    seq = [(1, 1), (2, 1), (1, 0), (2, 0), (1, 1), (2, 1), (0, 1)]
    for i, (cost, sat) in enumerate(seq):
        m.set_option("cost", cost).set_option("saturate", sat)
        l, r = (a, b)[i & 1]
        got = m.compute(l, r)
        m.status()
        assert np.array_equal(got, want[i & 1]), (i, cost, sat)
        vol = m.debug_volume("C").cpu().numpy()
        assert vol.shape[-1] == D and vol.min() >= p["P2"]
    # and as a batch of both pairs through the band passes
    m.set_option("path", 2)
    got = m.compute(np.stack([a[0], b[0]]), np.stack([a[1], b[1]]))
    m.status()
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


def test_u7_modes_agree_without_overflow(oracle):
    """Block 11 x RGB on black/white and checkerboard images (the reference's default block size): no window sum
    reaches 32767, so saturate and wrap give the same disparity, equal to the oracle's."""
    H, W, D = 40, 300, 96
    p = dict(minDisparity=0, numDisparities=D, blockSize=11, P1=968, P2=3872, disp12MaxDiff=1, uniquenessRatio=5)
    bw = np.zeros((H, W, 3), np.uint8)
    bw[:, W // 2:] = 255
    chk = ((np.add.outer(np.arange(H), np.arange(W)) & 1) * 255).astype(np.uint8)[..., None].repeat(3, 2)
    for l, r in ((bw, 255 - bw), (chk, 255 - chk), (chk, bw)):
        want = oracle.sgbm_compute(l, r, **p)
        assert oracle.sgbm_cost_volume(l, r, **p).max() < 32767
        for sat in (1, 0):
            m = ca.StereoSGBM_create(**p)
            m.set_option("saturate", sat)
            assert np.array_equal(m.compute(l, r), want)


def test_numpy_results_are_owned_by_the_caller():
    """Ownership at the NumPy surface (every stage returns a freshly allocated array): a result keeps its
    values when later calls run, is writable, and the page-locked hand-over returns the same values as plain copies."""
    from calibrating_amd import hostio
    W, H = 320, 240
    stereo = ca.Stereo.load(synthetic.rig(W, H))
    cfg = dict(max_size=W, minDisparity=0, numDisparities=32, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1)
    stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(cfg), max_depth=3.5)
    a1, a2 = synthetic.scene_pair(3, W, H, 3)
    b1, b2 = synthetic.scene_pair(4, W, H, 3)
    first = stereo.get_depth(a1, a2)
    kept = {k: v.copy() for k, v in first.items()}
    for _ in range(3):
        other = stereo.get_depth(b1, b2)
    assert not np.array_equal(other["disparity"], first["disparity"])
    for k, v in first.items():
        assert isinstance(v, np.ndarray) and v.flags.writeable and np.array_equal(v, kept[k]), k
    first["rectify_depth"][:] = -1  # the caller may scribble on its result
    assert np.array_equal(stereo.get_depth(a1, a2)["rectify_depth"], kept["rectify_depth"])
    try:
        hostio.PINNED = False
        plain = stereo.get_depth(a1, a2)
    finally:
        hostio.PINNED = True
    assert plain.keys() == kept.keys()
    for k in kept:
        assert np.array_equal(plain[k], kept[k]), k
    many = stereo.get_depth_batch(np.stack([a1, b1]), np.stack([a2, b2]))
    for k in kept:
        assert np.array_equal(many[k][0], kept[k]), k


def test_get_depth_is_capturable_in_a_hip_graph():
    """The device-tensor path makes no hidden synchronisation, allocation outside torch's pool or blocking copy: a whole
    get_depth (rectify x2, SGBM with its memsets / ticket resets / status copy, depth, unrectify, undistort) can be
    captured in a HIP graph once its tables exist, and the replay writes the same bits.  (Replay is no faster than the
    eager call -- 0.27 ms at VGA, 2.1 ms at 1080p either way: the kernels already run back to back.)"""
    W, H = 320, 240
    stereo = ca.Stereo.load(synthetic.rig(W, H))
    cfg = dict(max_size=W, minDisparity=0, numDisparities=64, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1,
               uniquenessRatio=10, speckleWindowSize=100, speckleRange=2)
    stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(cfg), max_depth=3.5)
    i1, i2 = synthetic.scene_pair(9, W, H, 3)
    t1, t2 = torch.from_numpy(i1).cuda(), torch.from_numpy(i2).cuda()
    ref = stereo.get_depth(t1, t2)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        stereo.get_depth(t1, t2)  # tables, workspaces and allocator pool of the capture stream
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        out = stereo.get_depth(t1, t2)
    j1, j2 = synthetic.scene_pair(10, W, H, 3)  # new images into the captured input buffers
    t1.copy_(torch.from_numpy(j1)); t2.copy_(torch.from_numpy(j2))
    graph.replay()
    torch.cuda.synchronize()
    got = {k: v.clone() for k, v in out.items()}
    want = stereo.get_depth(t1, t2)
    for k in want:
        assert torch.equal(got[k], want[k]), k
    assert not torch.equal(want["disparity"], ref["disparity"])


def test_get_depth_keys_returns_the_asked_entries_only():
    """``get_depth(..., keys=...)`` (not in the reference: its dict is ~60 MB per 1080p pair) returns exactly the asked
    entries, each identical to the full dict's; entries nothing else needs are not even computed."""
    W, H = 320, 240
    rec = synthetic.rig(W, H)
    stereo = ca.Stereo.load(rec)
    cfg = dict(max_size=W, minDisparity=0, numDisparities=64, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1,
               uniquenessRatio=10, speckleWindowSize=50, speckleRange=2)
    stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(cfg), max_depth=3.5)
    img1, img2, _ = synthetic.render_plane_pair(rec, (0.2, 0.1, 1.0), 2.0)
    full = stereo.get_depth(img1, img2)
    assert sorted(full) == sorted(ca.Stereo.RESULT_KEYS)
    for keys in (("unrectify_depth",), ("rectify_depth", "disparity"), "undistort_img1", ca.Stereo.RESULT_KEYS):
        got = stereo.get_depth(img1, img2, keys=keys)
        want_keys = (keys,) if isinstance(keys, str) else keys
        assert sorted(got) == sorted(want_keys)
        assert all(isinstance(v, np.ndarray) and np.array_equal(v, full[k]) for k, v in got.items())
    t = stereo.get_depth(torch.from_numpy(img1).cuda(), torch.from_numpy(img2).cuda(), keys=("unrectify_depth",))
    assert list(t) == ["unrectify_depth"] and np.array_equal(t["unrectify_depth"].cpu().numpy(), full["unrectify_depth"])
    with pytest.raises(ValueError):
        stereo.get_depth(img1, img2, keys=("depth",))
    # the batched form takes the same argument
    gb = stereo.get_depth_batch(np.stack([img1, img1]), np.stack([img2, img2]), keys=("unrectify_depth", "disparity"))
    assert sorted(gb) == ["disparity", "unrectify_depth"]
    assert all(np.array_equal(gb[k][1], full[k]) for k in gb)
    # ONE asked key (the documented headline use): the value keeps its leading n -- a single result used to be unwrapped
    # by hostio.to_host and zipped row by row, which returned pair 0 only, shape (h, w), without a word
    for key in ("unrectify_depth", "disparity"):
        g1 = stereo.get_depth_batch(np.stack([img1, img1, img1]), np.stack([img2, img2, img2]), keys=(key,))
        assert list(g1) == [key] and isinstance(g1[key], np.ndarray) and g1[key].shape == (3,) + full[key].shape
        assert all(np.array_equal(g1[key][i], full[key]) for i in range(3))
    g1 = stereo.get_depth_batch(np.stack([img1]), np.stack([img2]), keys="rectify_depth")  # n = 1, one key
    assert g1["rectify_depth"].shape == (1,) + full["rectify_depth"].shape
    with pytest.raises(ValueError):
        stereo.get_depth_batch(np.stack([img1]), np.stack([img2]), keys=("nope",))


def test_get_depth_async_returns_the_same_dicts_in_order():
    """``get_depth_async`` (not in the reference) queues a call and returns; several calls in flight on different pairs,
    results collected later and out of order, must each equal the synchronous call's dict."""
    W, H = 320, 240
    rec = synthetic.rig(W, H)
    stereo = ca.Stereo.load(rec)
    cfg = dict(max_size=W, minDisparity=0, numDisparities=64, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1,
               uniquenessRatio=10, speckleWindowSize=50, speckleRange=2)
    stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(cfg), max_depth=3.5)
    pairs = [synthetic.render_plane_pair(rec, (0.2, 0.1, 1.0), 1.6 + 0.3 * i, seed=i)[:2] for i in range(4)]
    want = [stereo.get_depth(a, b) for a, b in pairs]
    pend = [stereo.get_depth_async(a, b) for a, b in pairs]
    for i in (2, 0, 3, 1):
        got = pend[i].result()
        assert sorted(got) == sorted(want[i])
        assert all(isinstance(got[k], np.ndarray) and np.array_equal(got[k], want[i][k]) for k in got), i
    assert pend[1].result() is pend[1].result()
    one = stereo.get_depth_async(*pairs[1], keys=("unrectify_depth",)).result()
    assert list(one) == ["unrectify_depth"] and np.array_equal(one["unrectify_depth"], want[1]["unrectify_depth"])
    t = stereo.get_depth_async(torch.from_numpy(pairs[0][0]).cuda(), torch.from_numpy(pairs[0][1]).cuda()).result()
    assert np.array_equal(t["disparity"].cpu().numpy(), want[0]["disparity"])
    stereo.stereo_matching.stereo_sgbm.status()
