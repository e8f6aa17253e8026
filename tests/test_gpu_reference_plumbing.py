"""The REFERENCE's own Python as the checker of the HIP path (-m gpu).

tests/golden/reference_plumbing.npz = what /root/reference/calibrating (unmodified; oracle-backed stand-ins for its
cv2 / boxx imports; made in the build container by tests/golden/make_reference_golden.py) returned from
``Stereo.load -> set_stereo_matching -> get_depth`` and from its depth post-ops on the catalogue
tests/golden/reference_cases.py.  ``calibrating_amd`` gets the same records, arguments and images and must return the
same dict: same keys, dtypes and shapes, images and disparity bit for bit, depths with the same invalid set within
1e-4 m (BASELINE.json) -- what is reached is the same float64 bits, and the tests say so.
/root/reference is not read here (it does not exist on the GPU box); the fixture is data.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import calibrating_amd as ca  # noqa: E402
from calibrating_amd import pointcloud  # noqa: E402
import reference_fixture as rf  # noqa: E402
from reference_fixture import rc  # noqa: E402


@pytest.fixture(scope="module")
def fx():
    return rf.fixture()


def _stereo(fx, case):
    rec, kw = rf.load_args(fx, case)
    st = ca.Stereo(**kw).load(rec)
    kind, cfg = case["plugin"]
    st.set_stereo_matching(rc.make_plugin(kind, cfg, ca.MetaStereoMatching, ca.SemiGlobalBlockMatching), **case["setm"])
    return st


def _assert_same(fx, case, res, what):
    bad, inexact = rf.check_result(fx, case, res)
    assert not bad, (case["name"], what, bad)
    assert not inexact, (case["name"], what, "within 1e-4 m but not the reference's float64 bits", inexact)


@pytest.mark.parametrize("name", [c["name"] for c in rc.CASES])
def test_get_depth_equals_the_references(fx, name):
    case = rc.CASE_BY_NAME[name]
    st = _stereo(fx, case)
    img1, img2 = rf.images(fx, case)
    res = st.get_depth(img1, img2, **case.get("call", {}))
    _assert_same(fx, case, res, "get_depth(ndarray)")
    assert all(isinstance(v, np.ndarray) for k, v in res.items() if k != "note")
    # the rig state the GPU tables were built from, and the GPU-built tables themselves
    bad = rf.check_rig(fx, case, st, maps=False)
    assert not bad, bad
    tb = st._tables(torch.device("cuda", torch.cuda.current_device()))
    h, w = int(st.xy[1]), int(st.xy[0])
    mask = np.unpackbits(fx[name + "/mask_bits"])[:h * w].reshape(h, w)
    assert np.array_equal(tb["mask"].cpu().numpy().astype(np.uint8), mask)
    for key in ("map1x", "map1y", "map2x", "map2y"):
        assert rc.sha(tb[key].cpu().numpy()) == str(fx["%s/%s_sha" % (name, key)]), key
    # device tensors in -> device tensors out, the same bits (foreign plugins get ndarrays either way)
    t1, t2 = torch.from_numpy(img1).cuda(), torch.from_numpy(img2).cuda()
    rt = st.get_depth(t1, t2, **case.get("call", {}))
    _assert_same(fx, case, {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in rt.items()},
                 "get_depth(tensors)")


@pytest.mark.parametrize("name", ["c1_default_720p", "translate_default", "hetero", "max_size_300", "xytarget_scalar",
                                  "ktarget_matrix"])
def test_get_depth_batch_equals_the_references(fx, name):
    """The batched form (not in the reference): pair i of the batch == the reference's one-pair call."""
    case = rc.CASE_BY_NAME[name]
    st = _stereo(fx, case)
    img1, img2 = rf.images(fx, case)
    other1, other2 = np.ascontiguousarray(img1[::-1]), np.ascontiguousarray(img2[::-1])  # a different pair beside it
    gb = st.get_depth_batch(np.stack([other1, img1, img1]), np.stack([other2, img2, img2]))
    for i in (1, 2):
        _assert_same(fx, case, {k: v[i] for k, v in gb.items()}, "get_depth_batch[%d]" % i)


def test_unrectify_maps_equal_the_references(fx):
    """utils.rotate_depth_by_remap's memoised maps (utils.py:183-191) as the GPU builds them."""
    for name in ("ktarget_scalar", "xytarget_tuple", "hetero"):
        case = rc.CASE_BY_NAME[name]
        st = _stereo(fx, case)
        mx, my = st._unrectify_tables(torch.device("cuda", torch.cuda.current_device()))
        assert rc.sha(mx.cpu().numpy()) == str(fx[name + "/unrect_mapx_sha"])
        assert rc.sha(my.cpu().numpy()) == str(fx[name + "/unrect_mapy_sha"])


# ---- depth post-ops (n4): the reference's NumPy (utils.py:152-161,201-318, camera.py:298-309) ------------------------
@pytest.mark.parametrize("rate", rc.POST_RATES)
def test_depth_to_point_cloud_equals_the_references(fx, rate):
    depth = rc.post_depth(1, rc.POST_XY1[1], rc.POST_XY1[0])
    got = pointcloud.depth_to_point_cloud(depth, rc.POST_K1, interpolation_rate=rate, return_xyzuv=True)
    assert len(got) == int(fx["post/cloud_rate%s_n" % rate])
    assert rc.sha(got[:, 3:]) == str(fx["post/cloud_rate%s_uv_sha" % rate])
    assert np.array_equal(got[::rc.CLOUD_ROWS], fx["post/cloud_rate%s" % rate])  # the same float64 bits


def test_scatter_and_projection_equal_the_references(fx):
    mm = (np.arange(12, dtype=np.uint16).reshape(3, 4) * 250)
    assert np.array_equal(pointcloud.depth_to_point_cloud(mm, rc.POST_K1), fx["post/cloud_mm"])
    depth2 = rc.post_depth(2, rc.POST_XY1[1], rc.POST_XY1[0])
    cloud = pointcloud.depth_to_point_cloud(depth2, rc.POST_K1)
    moved = pointcloud.apply_T_to_point_cloud(rc.post_T(), cloud)
    assert np.array_equal(moved[::rc.CLOUD_ROWS], fx["post/moved"])
    s = rc.POST_SAMPLE
    assert np.array_equal(rc.sample(pointcloud.point_cloud_to_depth(cloud, rc.POST_K1, rc.POST_XY1), s), fx["post/back"])
    assert np.array_equal(rc.sample(pointcloud.point_cloud_to_depth(moved, rc.POST_K1, rc.POST_XY1), s),
                          fx["post/moved_depth"])
    cam1 = ca.Cam.init_by_K_D(rc.POST_K1, None, rc.POST_XY1)
    cam2 = ca.Cam.init_by_K_D(rc.POST_K2, None, rc.POST_XY2)
    depth3 = rc.post_depth(3, rc.POST_XY2[1], rc.POST_XY2[0])
    for interp in rc.POST_INTERPOLATIONS:
        got = cam1.project_cam2_depth(cam2, depth3, T=fx["post/T2"], interpolation=interp)
        assert np.array_equal(rc.sample(got, s), fx["post/project_%s" % interp]), interp
