"""CPU tests of the drop-in boundary and the host logic (no GPU, no compute calls):
the C-ABI library loads and exports every symbol include/calibrating_amd.h declares, the product fails
loudly without a device, and the Python mirror of Stereo reproduces the reference's scalar logic and
table construction (checked against the oracle and against geometric self-consistency)."""
import os
import re

import numpy as np
import pytest

import calibrating_amd as ca
from calibrating_amd import _native, geometry, synthetic
from calibrating_amd.parallel_pairs import owner_of, shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols(name="calibrating_amd.h"):
    src = open(os.path.join(ROOT, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(camd_[A-Za-z0-9_]+)\s*\(", src)))


def _exported_symbols():
    import shutil
    import subprocess
    nm = shutil.which("nm") or shutil.which("llvm-nm")
    if nm is None:
        pytest.skip("no nm on this machine")
    out = subprocess.run([nm, "-D", "--defined-only", _native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    return sorted({l.split()[-1] for l in out.splitlines() if l.split() and l.split()[-1].startswith("camd_")})


def test_library_exports_every_header_symbol():
    syms = _header_symbols()
    assert len(syms) >= 20
    lib = _native.lib()
    for s in syms:
        assert hasattr(lib, s), "libcalibrating_amd.so does not export %s" % s
    # the binding table IS the public header: same names, nothing more, nothing less
    assert sorted(syms) == sorted(_native.SIGNATURES)
    # what has no counterpart in the reference's interface lives in the experimental header, bound separately
    exp = _header_symbols("calibrating_amd_experimental.h")
    assert sorted(exp) == sorted(_native.EXPERIMENTAL_SIGNATURES) and not set(exp) & set(syms)
    # and the shared object exports exactly the two headers' C symbols
    assert _exported_symbols() == sorted(syms + exp)
    assert lib.camd_version() >= 100
    assert lib.camd_sgbm_num_stages() > 0 and lib.camd_sgbm_stage_name(0)


def test_the_product_does_not_call_the_experimental_entry_points():
    """camd_stream_create_cu_mask / camd_stream_destroy / CAMD_OPT_PHASES are measurement hooks: no module of the
    package may call them (tools/ and tests/ may)."""
    pkg = os.path.join(ROOT, "calibrating_amd")
    for f in sorted(os.listdir(pkg)):
        if f.endswith(".py") and f != "_native.py":
            src = open(os.path.join(pkg, f)).read()
            assert "camd_stream_" not in src, f
            assert not re.search(r"set_option\(\s*[\"']phases", src), f


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert _native.lib().camd_device_ok() == _native.CAMD_ERR_NO_DEVICE
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ca.StereoSGBM_create(numDisparities=16).compute(np.zeros((8, 40), np.uint8), np.zeros((8, 40), np.uint8))
    with pytest.raises(RuntimeError):
        from calibrating_amd import imgproc
        imgproc.remap(np.zeros((4, 4), np.uint8), np.zeros((4, 4), np.float32), np.zeros((4, 4), np.float32))


def test_argument_validation_without_gpu():
    import ctypes
    lib = _native.lib()
    p = _native.SgbmParams(numDisparities=128, blockSize=5, P1=200, P2=800)
    # C2 workspace: C and S (495 MB each) + the five per-direction volumes of the one-pair latency path + edge
    # records, keys, raw disparity; with 64 pairs per call the latency path holds at most 4 pairs' worth
    V = 1080 * 1792 * 128 * 2
    ws = lib.camd_sgbm_workspace_bytes(ctypes.byref(p), 1920, 1080, 1, 1)
    assert 7 * V < ws < 7.4 * V
    ws64 = lib.camd_sgbm_workspace_bytes(ctypes.byref(p), 1920, 1080, 1, 64)
    assert (2 * 64 + 5 * 4) * V < ws64 < (2.4 * 64 + 5 * 4) * V
    assert lib.camd_sgbm_workspace_bytes(ctypes.byref(p), 1920, 1080, 2, 1) == 0  # 2 channels: cv2 error too
    bad = _native.SgbmParams(numDisparities=0)
    h = ctypes.c_void_p()
    assert lib.camd_sgbm_create(ctypes.byref(bad), 64, 64, 1, 1, ctypes.byref(h)) == _native.CAMD_ERR_BAD_ARG
    assert "numDisparities" in _native.last_error()
    m3 = _native.SgbmParams(numDisparities=16, mode=5)
    assert lib.camd_sgbm_create(ctypes.byref(m3), 64, 64, 1, 1, ctypes.byref(h)) == _native.CAMD_ERR_BAD_ARG
    # MODE_SGBM_3WAY needs room for its stripes' warm-up rows
    small = _native.SgbmParams(numDisparities=16, blockSize=11, mode=2)
    assert lib.camd_sgbm_workspace_bytes(ctypes.byref(small), 64, 12, 1, 1) == 0
    assert lib.camd_sgbm_workspace_bytes(ctypes.byref(small), 64, 64, 1, 1) > 0


def test_interpolation_tables_three_independent_builders():
    """The fixed-point Lanczos-4 / bilinear tables of the product (csrc/remap.hip), of the oracle (oracle/remap_ref.c)
    and of an independently written NumPy model (tests/np_interp_tables.py) agree bit for bit -- for both settings of
    the U15 correction-group switch, which the product and the oracle share (one flip moves both)."""
    import oracle
    import np_interp_tables as model
    lib = _native.lib()

    def product_tables():
        tab, tb = np.empty((1024, 64), np.int16), np.empty((1024, 4), np.int16)
        assert lib.camd_lanczos4_table_host(tab.ctypes.data) == 0
        assert lib.camd_bilinear_table_host(tb.ctypes.data) == 0
        return tab, tb

    try:
        seen = []
        for lo in (4, 3):
            assert lib.camd_set_global_option(0, lo) == 0
            oracle.set_switches(lanczos_fix_group_lo=lo)
            tab, tb = product_tables()
            assert np.array_equal(tab, model.lanczos4_itab(lo)), "product vs NumPy model, group %d" % lo
            assert np.array_equal(oracle.lanczos4_itab(), model.lanczos4_itab(lo)), "oracle vs NumPy model, group %d" % lo
            assert np.array_equal(tb, model.bilinear_itab()) and np.array_equal(oracle.bilinear_itab(), tb)
            sums = tab.astype(np.int64).sum(1)
            assert (tb.astype(np.int64).sum(1) == 32768).all()
            if lo == 4:
                assert (sums == 32768).all()
            else:
                # evidence for U15 = 4: with the group {3,4} the phase-(0,0) entry would add the missing 1 to its
                # centre tap, which already holds the saturated 32767 -> (short)32768 = -32768: integer-coordinate
                # remaps would negate the image, which cv2 visibly does not do
                assert sums[0] == -32768 and (sums[1:] == 32768).all()
            seen.append(tab.copy())
        assert not np.array_equal(seen[0], seen[1]), "the switch must change some entries"
        assert lib.camd_set_global_option(0, 7) != 0 and lib.camd_set_global_option(99, 0) != 0
    finally:
        lib.camd_set_global_option(0, 4)
        oracle.set_switches()


def test_plugin_surface_matches_reference_names():
    assert issubclass(ca.SemiGlobalBlockMatching, ca.MetaStereoMatching)
    m = ca.SemiGlobalBlockMatching()
    assert m.max_size == 1000 and m.cfg == {}
    # the reference's hard-coded cv2.StereoSGBM_create arguments (stereo_matching.py:30-58)
    assert m.stereo_sgbm.params == dict(minDisparity=2, numDisparities=218, blockSize=11, P1=968, P2=3872,
                                        disp12MaxDiff=0, preFilterCap=0, uniquenessRatio=5,
                                        speckleWindowSize=200, speckleRange=2, mode=0)
    assert m.stereo_sgbm.getMinDisparity() == 2
    with pytest.raises(NotImplementedError):
        ca.MetaStereoMatching()(None, None)
    with pytest.raises(TypeError):
        ca.StereoSGBM_create(numDisparities=16, bogus=1)
    m.stereo_sgbm.setNumDisparities(64)
    assert m.stereo_sgbm.getNumDisparities() == 64


def test_stereo_load_dump_roundtrip_and_scalars(tmp_path):
    rig = synthetic.rig(320, 240)
    s = ca.Stereo.load(rig)
    assert s.xy == (320, 240) and s.undistort_rectify_map1[0].shape == (240, 320)
    assert s.undistort_rectify_map1[0].dtype == np.float32 and s.rectify_valid_mask1.dtype == bool
    assert abs(s.baseline - np.linalg.norm([0.12, 0.002, 0.001])) < 1e-12
    path = str(tmp_path / "stereo.yaml")
    s.dump(path)
    s2 = ca.Stereo.load(path)
    assert np.array_equal(s2.undistort_rectify_map2[1], s.undistort_rectify_map2[1])
    s3 = ca.Stereo.load(s.dump())  # yaml string
    assert np.allclose(s3.R1, s.R1)
    # r (Rodrigues) and T (4x4) inputs (stereo_camera.py:287-291)
    d = dict(rig)
    d.pop("R")
    d["r"] = [0.01, -0.02, 0.005]
    assert np.allclose(ca.Stereo.load(d).R, s.R)
    T = np.eye(4)
    T[:3, :3] = s.R
    T[:3, 3:] = s.t
    d2 = {k: v for k, v in rig.items() if k not in ("R", "t")}
    d2["T"] = T.tolist()
    s4 = ca.Stereo.load(d2)
    assert np.allclose(s4.R, s.R, atol=1e-9) and np.allclose(s4.t, s.t)
    # set_stereo_matching scalar logic (stereo_camera.py:466-489), incl. quirk Q1 (cam1.K, not self.K)
    s.set_stereo_matching(ca.SemiGlobalBlockMatching({}), max_depth=3.5)
    assert s.translation_rectify_img is True and s.max_depth == 3.5
    assert s.min_disparity == int(s.cam1.K[0, 0] * s.baseline / 3.5)
    s.set_stereo_matching(ca.SemiGlobalBlockMatching({}))
    assert s.translation_rectify_img is False and s.max_depth == 1000 and s.min_disparity == 0
    s.set_stereo_matching(ca.SemiGlobalBlockMatching({}), max_depth=2.0, translation_rectify_img=False)
    assert s.translation_rectify_img is False and s.min_disparity == int(s.cam1.K[0, 0] * s.baseline / 2.0)
    # disparity_to_depth on the host (NumPy): reference edge cases (stereo_camera.py:408-413)
    z = s.disparity_to_depth(np.array([[0.0, -1.0, 1e-9, 16.5]], np.float32))
    bf = s.baseline * s.K[0, 0]
    assert z.dtype == np.float64 and z[0, 0] == 0 and z[0, 1] == 0 and z[0, 2] == 0 and z[0, 3] == bf / 16.5
    with pytest.raises(NotImplementedError):
        ca.Stereo(s.cam1, s.cam2)  # extrinsic calibration is out of scope: needs R, t
    s5 = ca.Stereo(s.cam1, s.cam2, R=s.R, t=s.t)
    assert np.array_equal(s5.undistort_rectify_map1[0], s.undistort_rectify_map1[0])


def test_rectification_is_self_consistent(oracle):
    """R1 = R2 R; projecting a 3-D point through both rectified cameras gives equal rows;
    maps equal the oracle's initUndistortRectifyMap; valid mask follows stereo_camera.py:167-176."""
    s = ca.Stereo.load(synthetic.rig(640, 480))
    assert np.allclose(s.R1, s.R2 @ s.R)
    assert np.allclose(s.R1 @ s.R1.T, np.eye(3), atol=1e-12) and np.allclose(s.R2 @ s.R2.T, np.eye(3), atol=1e-12)
    # cam2 = R cam1 + t (stereoCalibrate convention): rectified coords X1r = R1 X1, X2r = R2 X2
    rng = np.random.default_rng(0)
    X1 = rng.uniform([-1, -1, 1.5], [1, 1, 4], (50, 3))
    X2 = X1 @ s.R.T + s.t.reshape(1, 3)
    p1 = (X1 @ s.R1.T) @ s.K.T
    p2 = (X2 @ s.R2.T) @ s.K.T
    v1, v2 = p1[:, 1] / p1[:, 2], p2[:, 1] / p2[:, 2]
    # epipolar lines are rows (residual ~3e-6 px comes from the reference's own eps = 1e-8 in
    # rotate_shortest_of_two_vecs, utils.py:146)
    assert np.abs(v1 - v2).max() < 1e-4
    u1, u2 = p1[:, 0] / p1[:, 2], p2[:, 0] / p2[:, 2]
    disp = u1 - u2
    assert (disp > 0).all()                       # positive disparity, = baseline * fx / z
    assert np.allclose(disp, s.baseline * s.K[0, 0] / (X1 @ s.R1.T)[:, 2], rtol=1e-6)
    for cam, R, maps in ((s.cam1, s.R1, s.undistort_rectify_map1), (s.cam2, s.R2, s.undistort_rectify_map2)):
        mx, my = oracle.init_undistort_rectify_map(cam.K, cam.D, R, s.K, s.xy)
        assert np.array_equal(mx, maps[0]) and np.array_equal(my, maps[1])
    mx, my = s.undistort_rectify_map1
    W, H = s.cam1.xy
    want = (-0.5 < mx) & (mx < W - 0.5) & (-0.5 < my) & (my < H - 0.5)
    assert np.array_equal(s.rectify_valid_mask1, want)
    # xy_target / K_target variants run and keep the principal ray near the image centre
    s2 = ca.Stereo.load(dict(synthetic.rig(640, 480)))
    s2.xy_target, s2.K_target = 0.5, 0.5
    s2._get_undistort_rectify_map()
    assert s2.xy == (320, 240) and abs(s2.K[0, 0] - 0.5 * s2.cam1.K[0, 0]) < 1e-9


def test_geometry_helpers():
    r = np.array([0.3, -0.2, 0.1])
    R = geometry.rodrigues(r)
    assert np.allclose(R @ R.T, np.eye(3)) and abs(np.linalg.det(R) - 1) < 1e-12
    assert np.allclose(geometry.rodrigues(R).reshape(3), r)
    assert np.allclose(geometry.rodrigues(np.zeros(3)), np.eye(3))
    v1, v2 = np.array([-1.0, 0, 0]), np.array([-0.12, 0.002, -0.001])
    Rs = geometry.rotate_shortest_of_two_vecs(v1, v2)
    assert np.allclose(Rs @ v1, v2 / np.linalg.norm(v2), atol=1e-6)
    assert np.allclose(geometry.project_vec_on_plane(np.array([1.0, 2, 3]), np.array([0, 0, 2.0])), [1, 2, 0])
    m = np.array([[2.0, 1, 0], [0, 3, 1], [1, 0, 4]])
    assert np.allclose(geometry.inv3(m) @ m, np.eye(3))


def test_shard_ranges_cover_the_pair_list():
    for n, g in ((512, 8), (10, 4), (7, 8), (64, 1)):
        ranges = [shard_range(n, g, r) for r in range(g)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        for i in range(n):
            r = owner_of(i, n, g)
            assert ranges[r][0] <= i < ranges[r][1]
    assert [shard_range(512, 8, r) for r in range(8)] == [(64 * r, 64 * r + 64) for r in range(8)]


def test_a_bundle_rig_refuses_to_rebuild_tables_it_was_not_given():
    """A rig made from a broadcast table bundle has no camera 2 intrinsics (cam2.K is NaN): asked for tables on a
    device the bundle was not installed on, it must raise -- a silent rebuild would hand back NaN maps and garbage depth.
    The device key has one spelling (torch.device('cpu') == 'cpu')."""
    import torch
    import calibrating_amd as ca
    from calibrating_amd import synthetic
    src = ca.Stereo.load(synthetic.rig(96, 64))
    tabs = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in src.table_bundle().items()}
    st = ca.Stereo.from_bundle(tabs, "cpu")
    assert st._tables(torch.device("cpu"))["map2x"] is tabs["map2x"]
    assert st._unrectify_tables("cpu")[1] is tabs["unrect_mapy"]
    with pytest.raises(RuntimeError, match="table bundle"):
        st._tables(torch.device("cuda", 0))  # (no CUDA call is made: the key is compared first)
    with pytest.raises(ValueError, match="live on"):
        ca.Stereo.from_bundle(tabs, torch.device("cuda", 1))
    # the host views come from the installed tensors, whatever else sits in the device cache
    assert np.array_equal(st.undistort_rectify_map2[0], tabs["map2x"].numpy())
    assert np.array_equal(st.rectify_valid_mask1, tabs["mask"].numpy().astype(bool))
