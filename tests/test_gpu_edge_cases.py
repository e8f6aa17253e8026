"""Edge cases of the SGBM kernels (-m gpu): tiny and ragged images, extreme parameters, every
aggregation path; all against the CPU oracle, bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import calibrating_amd as ca  # noqa: E402
from calibrating_amd import StereoSGBM_create, synthetic  # noqa: E402


def _rand_pair(seed, H, W, cn):
    rng = np.random.default_rng(seed)
    shape = (H, W) if cn == 1 else (H, W, cn)
    base = rng.integers(0, 256, shape, dtype=np.uint8)
    right = np.roll(base, -3, axis=1)
    noise = rng.integers(-3, 4, shape)
    return base, np.clip(right.astype(int) + noise, 0, 255).astype(np.uint8)


def _check(oracle, left, right, p, paths=(2, 1, 3)):
    ref = oracle.sgbm_compute(left, right, **p)
    for path in paths:
        m = StereoSGBM_create(**p)
        m.set_option("path", path)
        got = m.compute(left, right)
        assert np.array_equal(got, ref), "path %d: %d px differ, max |d| = %d" % (
            path, (got != ref).sum(), np.abs(got.astype(int) - ref).max())


@pytest.mark.parametrize("H,W,D,cn,bs,minD,mode", [
    (1, 200, 64, 1, 5, 0, 0),      # a single row: vertical / diagonal paths see only borders
    (2, 150, 64, 3, 3, 0, 1),
    (3, 97, 40, 1, 5, 0, 1),
    (29, 70, 64, 1, 5, 0, 0),      # one row more than a band (28 rows)
    (57, 66, 64, 3, 3, 0, 1),      # width1 = 2 columns
    (30, 67, 64, 1, 5, 0, 0),      # width1 = 3 columns = blockSize/2 + 1, the narrowest defined case
    (30, 65, 64, 1, 1, 0, 1),      # width1 = 1 column (blockSize 1)
    (40, 130, 128, 1, 1, 0, 0),    # blockSize 1 (no box filter)
    (24, 300, 128, 3, 11, 0, 0),   # large block
    (20, 400, 300, 1, 5, 0, 1),    # D = 300: three vectors per lane, band path not instantiated -> scans
    (16, 700, 512, 1, 3, 0, 0),    # maximum supported D
    (12, 640, 512, 3, 11, 0, 1),   # maximum D with a large block: > 64 KB of dynamic LDS in the cost kernel
    (12, 400, 256, 1, 15, 0, 0),   # largest supported block
    (33, 180, 96, 1, 5, -40, 1),   # negative minDisparity
    (33, 180, 96, 3, 5, 17, 0),    # large positive minDisparity
    (65, 143, 72, 1, 7, 3, 1),     # everything odd
])
def test_sgbm_edge_shapes(oracle, H, W, D, cn, bs, minD, mode):
    left, right = _rand_pair(H * 1000 + W, H, W, cn)
    p = dict(minDisparity=minD, numDisparities=D, blockSize=bs, P1=8 * cn * bs * bs, P2=32 * cn * bs * bs,
             disp12MaxDiff=1, uniquenessRatio=10, mode=mode)
    _check(oracle, left, right, p)


@pytest.mark.parametrize("kw", [
    dict(uniquenessRatio=0), dict(uniquenessRatio=50), dict(uniquenessRatio=99), dict(uniquenessRatio=-5),
    dict(disp12MaxDiff=-1), dict(disp12MaxDiff=10), dict(P1=1, P2=2), dict(P1=3000, P2=9000),
    dict(preFilterCap=63), dict(preFilterCap=5), dict(speckleWindowSize=50, speckleRange=4),
    # cv2's rule of thumb P1 = 8 cn b^2, P2 = 32 cn b^2 at the largest block of an RGB pair; the library's P2 limit
    dict(blockSize=15, P1=5400, P2=21600), dict(blockSize=15, P1=5400, P2=21600, preFilterCap=63), dict(P1=10, P2=24000),
])
def test_sgbm_parameter_extremes(oracle, kw):
    left, right = synthetic.rectified_pair(seed=17, H=36, W=220, D=64, cn=3)
    p = dict(minDisparity=0, numDisparities=64, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1, uniquenessRatio=10)
    p.update(kw)
    for mode in (0, 1):
        p["mode"] = mode
        _check(oracle, left, right, p)


def test_sgbm_too_narrow_is_refused(oracle):
    """0 < width - numDisparities <= blockSize/2: cv2 reads unwritten memory; oracle and product both refuse."""
    left, right = _rand_pair(1, 30, 66, 1)
    p = dict(numDisparities=64, blockSize=5)
    with pytest.raises(ValueError):
        oracle.sgbm_compute(left, right, **p)
    with pytest.raises(ValueError, match="undefined"):
        StereoSGBM_create(**p).compute(left, right)


def test_sgbm_saturated_images(oracle):
    """All-black / all-white / checkerboard-of-extremes inputs (maximum BT costs, ties everywhere)."""
    H, W = 30, 160
    yy, xx = np.mgrid[:H, :W]
    checker = (((xx // 3 + yy // 2) % 2) * 255).astype(np.uint8)
    for left, right in ((np.zeros((H, W), np.uint8), np.full((H, W), 255, np.uint8)),
                        (checker, np.roll(checker, 2, axis=1)), (checker, 255 - checker)):
        for mode in (0, 1):
            p = dict(minDisparity=0, numDisparities=48, blockSize=5, P1=200, P2=800, disp12MaxDiff=1,
                     uniquenessRatio=5, mode=mode)
            _check(oracle, left, right, p)


def test_sgbm_pitched_and_batched_inputs(oracle):
    """Non-contiguous views are accepted (made contiguous by the wrapper); a batch larger than the first call
    re-creates the handle."""
    left, right = synthetic.rectified_pair(seed=4, H=40, W=260, D=64, cn=3)
    p = dict(minDisparity=0, numDisparities=64, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1, uniquenessRatio=10)
    m = StereoSGBM_create(**p)
    ref = oracle.sgbm_compute(left, right, **p)
    big_l = np.zeros((40, 300, 3), np.uint8); big_l[:, :260] = left
    big_r = np.zeros((40, 300, 3), np.uint8); big_r[:, :260] = right
    got = m.compute(torch.from_numpy(big_l).cuda()[:, :260], torch.from_numpy(big_r).cuda()[:, :260])
    assert np.array_equal(got.cpu().numpy(), ref)
    L = np.stack([left] * 9); R = np.stack([right] * 9)        # 9 pairs: auto path = band passes
    got9 = m.compute(L, R)
    assert all(np.array_equal(got9[i], ref) for i in range(9))
    got1 = m.compute(left, right)                                # back to one pair on the same handle
    assert np.array_equal(got1, ref)


def test_sgbm_randomised_shapes_and_parameters(oracle):
    """Seeded fuzz: random sizes, channel counts, disparity ranges, block sizes, penalties and modes; every
    aggregation path against the oracle, bit-exact."""
    rng = np.random.default_rng(20240929)
    for case in range(24):
        cn = int(rng.choice([1, 3]))
        D = int(rng.choice([16, 32, 48, 64, 96, 128, 160, 256]))
        bs = int(rng.choice([1, 3, 5, 7, 9]))
        minD = int(rng.integers(-6, 7))
        W = D + abs(minD) + int(rng.integers(bs // 2 + 2, 90))
        H = int(rng.integers(3, 70))
        mode = int(rng.choice([0, 1, 3]))
        P1 = int(rng.integers(1, 8 * cn * bs * bs + 2))
        P2 = P1 + int(rng.integers(1, 32 * cn * bs * bs + 2))
        p = dict(minDisparity=minD, numDisparities=D, blockSize=bs, P1=P1, P2=min(P2, 15000), disp12MaxDiff=int(rng.integers(-1, 4)),
                 uniquenessRatio=int(rng.integers(0, 30)), preFilterCap=int(rng.choice([0, 15, 31, 63])),
                 speckleWindowSize=int(rng.choice([0, 0, 40])), speckleRange=int(rng.integers(1, 4)), mode=mode)
        if case % 3 == 0:
            left, right = _rand_pair(case, H, W, cn)
        else:
            left, right = synthetic.rectified_pair(seed=case, H=H, W=W, D=max(min(D, W // 2), 8), cn=cn)
        try:
            _check(oracle, left, right, p)
        except AssertionError as e:
            raise AssertionError("case %d %s %s: %s" % (case, (H, W, cn), p, e))


def test_c_abi_pitched_buffers(oracle):
    """camd_sgbm_compute with row pitches / pair strides larger than the packed sizes (straight through the ABI)."""
    import ctypes
    from calibrating_amd import _native
    H, W, cn, D, nb = 37, 210, 3, 64, 3
    p = dict(minDisparity=0, numDisparities=D, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1, uniquenessRatio=10,
             speckleWindowSize=40, speckleRange=2)
    pairs = [synthetic.rectified_pair(seed=30 + i, H=H, W=W, D=D, cn=cn) for i in range(nb)]
    pitch, istride = W * cn + 29, (H + 2) * (W * cn + 29) + 64          # bytes
    dpitch, dstride = W + 11, (H + 1) * (W + 11) + 8                      # int16 elements
    L = torch.zeros(nb * istride, dtype=torch.uint8, device="cuda")
    R = torch.zeros(nb * istride, dtype=torch.uint8, device="cuda")
    for i, (l, r) in enumerate(pairs):
        for buf, img in ((L, l), (R, r)):
            view = buf[i * istride:i * istride + H * pitch].view(H, pitch)
            view[:, :W * cn] = torch.from_numpy(img.reshape(H, W * cn)).cuda()
    out = torch.full((nb * dstride,), -999, dtype=torch.int16, device="cuda")
    for path in (2, 1, 3):
        hd = ctypes.c_void_p()
        prm = _native.SgbmParams(**dict(dict(preFilterCap=0, mode=0), **p))
        lib = _native.lib()
        _native.check(lib.camd_sgbm_create(ctypes.byref(prm), W, H, cn, nb, ctypes.byref(hd)))
        _native.check(lib.camd_sgbm_set_option(hd, 0, path))
        _native.check(lib.camd_sgbm_compute(hd, L.data_ptr(), R.data_ptr(), pitch, istride, out.data_ptr(), dpitch * 2,
                                            dstride * 2, nb, _native.current_stream()))
        _native.check(lib.camd_sgbm_status(hd, _native.current_stream()))
        res = out.cpu().numpy()
        for i, (l, r) in enumerate(pairs):
            got = res[i * dstride:i * dstride + H * dpitch].reshape(H, dpitch)
            assert np.array_equal(got[:, :W], oracle.sgbm_compute(l, r, **p)), (path, i)
            assert (got[:, W:] == -999).all()                             # padding untouched
        _native.check(lib.camd_sgbm_destroy(hd))


def test_c_abi_handle_sized_for_more_pairs_than_the_first_call(oracle):
    """A handle created for max_batch pairs and first used with fewer (straight through the ABI: the Python wrapper sizes
    its handle to the call): the speckle filter's union-find scratch is cleared once for the WHOLE handle, not for the
    first call's pairs, so a later, larger call must find its part of it clean."""
    import ctypes
    from calibrating_amd import _native
    H, W, D, nb = 70, 200, 32, 4
    p = dict(minDisparity=0, numDisparities=D, blockSize=3, P1=24, P2=96, disp12MaxDiff=1, uniquenessRatio=5,
             speckleWindowSize=60, speckleRange=1)
    pairs = [synthetic.rectified_pair(seed=60 + i, H=H, W=W, D=D, cn=1) for i in range(nb)]
    want = [oracle.sgbm_compute(l, r, **p) for l, r in pairs]
    L = torch.from_numpy(np.stack([l for l, _ in pairs])).cuda()
    R = torch.from_numpy(np.stack([r for _, r in pairs])).cuda()
    lib, hd = _native.lib(), ctypes.c_void_p()
    prm = _native.SgbmParams(**dict(dict(preFilterCap=0, mode=0), **p))
    _native.check(lib.camd_sgbm_create(ctypes.byref(prm), W, H, 1, nb, ctypes.byref(hd)))
    try:
        for n in (1, nb, 2, nb):
            out = torch.full((n, H, W), -999, dtype=torch.int16, device="cuda")
            _native.check(lib.camd_sgbm_compute(hd, L.data_ptr(), R.data_ptr(), W, H * W, out.data_ptr(), W * 2, H * W * 2, n,
                                                _native.current_stream()))
            _native.check(lib.camd_sgbm_status(hd, _native.current_stream()))
            for i in range(n):
                assert np.array_equal(out[i].cpu().numpy(), want[i]), (n, i)
    finally:
        _native.check(lib.camd_sgbm_destroy(hd))


def test_matcher_keeps_a_handle_per_image_shape(oracle):
    """cv2's StereoSGBM object takes any image size from call to call (the reference's plugin feeds it whatever its
    resize produced, stereo_matching.py:60-63).  One matcher alternating between two shapes must not rebuild its device
    workspace every call: after the first call of each shape no camd_sgbm_create happens, the results stay identical,
    and the cache is bounded (least recently used out)."""
    p = dict(minDisparity=0, numDisparities=64, blockSize=5, P1=200, P2=800, disp12MaxDiff=1, uniquenessRatio=10)
    a = synthetic.rectified_pair(seed=1, H=120, W=320, D=64, cn=3)   # "640 x 480" / "800 x 600" in small
    b = synthetic.rectified_pair(seed=2, H=150, W=400, D=64, cn=3)
    want = [oracle.sgbm_compute(*a, **p), oracle.sgbm_compute(*b, **p)]
    m = ca.StereoSGBM_create(**p)
    m.compute(*a)
    m.compute(*b)
    n0 = ca.StereoSGBM.creates
    for k in range(6):
        pair, w = ((a, want[0]), (b, want[1]))[k & 1]
        assert np.array_equal(m.compute(*pair), w)
    assert ca.StereoSGBM.creates == n0, "alternating two shapes re-created a handle"
    # a larger batch of a known shape replaces that shape's handle (one create), then serves the smaller batches too
    m.compute(np.stack([a[0], a[0]]), np.stack([a[1], a[1]]))
    assert ca.StereoSGBM.creates == n0 + 1
    assert np.array_equal(m.compute(*a), want[0]) and ca.StereoSGBM.creates == n0 + 1
    # bounded: more shapes than HANDLE_CACHE evicts the least recently used, never the one in use
    for k in range(ca.StereoSGBM.HANDLE_CACHE + 1):
        c = synthetic.rectified_pair(seed=3, H=64, W=200 + 16 * k, D=64, cn=3)
        assert np.array_equal(m.compute(*c), oracle.sgbm_compute(*c, **p))
    assert len(m._cache) == ca.StereoSGBM.HANDLE_CACHE
    # a byte budget below two handles keeps only the one in use
    m2 = ca.StereoSGBM_create(**p)
    m2.HANDLE_CACHE_BYTES = 1
    m2.compute(*a)
    m2.compute(*b)
    assert len(m2._cache) == 1 and np.array_equal(m2.compute(*a), want[0])
    # a parameter change drops every handle (cv2 setters take effect on the next compute)
    m.setUniquenessRatio(5)
    assert len(m._cache) == 0
    assert np.array_equal(m.compute(*a), oracle.sgbm_compute(*a, **dict(p, uniquenessRatio=5)))


def test_cu_mask_stream_and_split_phases_give_the_same_disparity(oracle):
    """camd_stream_create_cu_mask (a HIP stream restricted to a subset of the compute units) and CAMD_OPT_PHASES (one
    compute queued as its cost half and its aggregation half, here on two different masked streams with an event in
    between): placement and splitting must not change a bit of the result."""
    import ctypes
    from calibrating_amd import _native
    p = dict(minDisparity=0, numDisparities=96, blockSize=5, P1=200, P2=800, disp12MaxDiff=1, uniquenessRatio=10)
    a = synthetic.rectified_pair(seed=4, H=96, W=400, D=96, cn=3)
    want = oracle.sgbm_compute(*a, **p)
    lib = _native.lib()
    streams, handles = [], []
    for lo, hi in ((0, 96), (96, 256)):
        words = (ctypes.c_uint32 * 8)()
        for i in range(lo, hi):
            words[i // 32] |= 1 << (i % 32)
        st = ctypes.c_void_p()
        _native.check(lib.camd_stream_create_cu_mask(words, 8, ctypes.byref(st)))
        handles.append(st)
        streams.append(torch.cuda.ExternalStream(st.value, device=torch.device("cuda", 0)))
    L, R = (torch.from_numpy(x).cuda() for x in a)
    m = ca.StereoSGBM_create(**p)
    m.set_option("path", 2)
    torch.cuda.synchronize()
    with torch.cuda.stream(streams[0]):           # everything on 96 CUs
        whole = m.compute(L, R)
    streams[0].synchronize()
    assert np.array_equal(whole.cpu().numpy(), want)
    out = torch.empty_like(whole)
    with torch.cuda.stream(streams[0]):           # the cost volume on 96 CUs ...
        m.set_option("phases", 1)
        m.compute(L, R, out=out)
        ev = streams[0].record_event()
    with torch.cuda.stream(streams[1]):           # ... aggregation and post on the other 160
        streams[1].wait_event(ev)
        m.set_option("phases", 2)             # (first aggregation pass)
        m.compute(L, R, out=out)
        m.set_option("phases", 4)             # (last pass, winner-take-all, post filters)
        m.compute(L, R, out=out)
    streams[1].synchronize()
    m.set_option("phases", 7)
    m.status()
    assert np.array_equal(out.cpu().numpy(), want)
    with pytest.raises(ValueError):
        m.set_option("phases", 0)
    del streams
    torch.cuda.synchronize()
    for st in handles:
        _native.check(lib.camd_stream_destroy(st))


def test_resident_launches_give_the_same_disparity(oracle):
    """CAMD_OPT_RESIDENT (experimental header): the cost kernel and the row-parallel last pass as a fixed number of
    persistent workgroups per CU that take their items by ticket -- the launch form of the co-residency experiment
    (profiles/r06_resident.txt).  Same bits as the ordinary launches, for several batches through one handle."""
    p = dict(minDisparity=0, numDisparities=128, blockSize=5, P1=200, P2=800, disp12MaxDiff=1, uniquenessRatio=10)
    pairs = [synthetic.rectified_pair(seed=40 + i, H=70, W=333, D=128, cn=3) for i in range(3)]
    L = torch.stack([torch.from_numpy(a[0]) for a in pairs]).cuda()
    R = torch.stack([torch.from_numpy(a[1]) for a in pairs]).cuda()
    want = np.stack([oracle.sgbm_compute(*a, **p) for a in pairs])
    m = ca.StereoSGBM_create(**p)
    m.set_option("path", 2)
    for value in (0x21, 0x31, 0x10, 0x02, 0):
        m.set_option("resident", value)
        for _ in range(2):
            assert np.array_equal(m.compute(L, R).cpu().numpy(), want), hex(value)
    m.status()
    with pytest.raises(ValueError):
        m.set_option("resident", 0x55)
    # gray, block 3: the other instantiations
    p1 = dict(p, blockSize=3, P1=72, P2=288)
    g = synthetic.rectified_pair(seed=44, H=64, W=300, D=128, cn=1)
    m1 = ca.StereoSGBM_create(**p1)
    m1.set_option("path", 2)
    m1.set_option("resident", 0x21)
    assert np.array_equal(m1.compute(*g), oracle.sgbm_compute(*g, **p1))
