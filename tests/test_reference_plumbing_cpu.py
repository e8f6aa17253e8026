"""The REFERENCE's own Python as the checker of everything it owns on the stereo-depth path -- CPU half.

tests/golden/reference_plumbing.npz holds what /root/reference/calibrating (unmodified, imported from where it lies in
the build container, with oracle-backed stand-ins for cv2 / boxx: tests/golden/make_reference_golden.py) produced for
the catalogue tests/golden/reference_cases.py.  Here, without a GPU:
  * ``calibrating_amd.Stereo`` (geometry.py: load from R / r / T records, rectifying rotations, target intrinsics incl.
    better_cx_cy, xy_target / K_target forms, valid mask, maps, set_stereo_matching's rules) reproduces the rig state
    the reference derived -- stereo_camera.py:125-185,199-214,264-297,466-489, utils.py:139-149;
  * the end-to-end checker of the GPU suite, tests/oracle_pipeline.py (the reference's get_depth composed from oracle
    stages), reproduces the reference's result dicts -- stereo_camera.py:216-242,408-431,492-533, utils.py:173-200,
    stereo_matching.py:60-70 -- so that "HIP == oracle_pipeline" elsewhere in the suite means "HIP == the reference's
    Python around the oracle's cv2 entry points", not "HIP == a second reading by the same author";
  * oracle/pointcloud_ref.py reproduces the reference's depth post-ops -- utils.py:152-161,201-318, camera.py:298-309.
What this does NOT pin: cv2's arithmetic (behind the stand-ins sits the oracle).
"""
import numpy as np
import pytest

import calibrating_amd as ca
from oracle import pointcloud_ref
from oracle_pipeline import oracle_get_depth
import reference_fixture as rf
from reference_fixture import rc


@pytest.fixture(scope="module")
def fx():
    return rf.fixture()


class _Installed:  # Stereo only needs *a* matcher installed to fix min_disparity / translation
    pass


def _stereo(fx, case):
    rec, kw = rf.load_args(fx, case)
    st = ca.Stereo(**kw).load(rec)
    st.set_stereo_matching(_Installed(), **case["setm"])
    return st


@pytest.mark.parametrize("name", [c["name"] for c in rc.CASES])
def test_rig_state_equals_the_references(fx, name):
    case = rc.CASE_BY_NAME[name]
    st = _stereo(fx, case)
    bad = rf.check_rig(fx, case, st)
    assert not bad, bad
    if case.get("record") == "T":  # the 4x4 spelling itself (see reference_fixture.load_args)
        rec, kw = rf.load_args(fx, case, as_given=True)
        given = ca.Stereo(**kw).load(rec)
        assert np.abs(given.R - fx[name + "/R"]).max() < 4e-15 and np.array_equal(given.t, fx[name + "/t"])
        assert np.abs(given.R1 - fx[name + "/R1"]).max() < 4e-15


@pytest.mark.parametrize("name", [c["name"] for c in rc.CASES])
def test_oracle_pipeline_reproduces_the_references_get_depth(fx, oracle, name):
    case = rc.CASE_BY_NAME[name]
    st = _stereo(fx, case)
    img1, img2 = rf.images(fx, case)
    kind, cfg = case["plugin"]
    plugin = None if kind == "sgbm" else rc.make_plugin(kind, cfg, ca.MetaStereoMatching, None)
    got = oracle_get_depth(oracle, st, cfg, img1, img2, plugin=plugin, **case.get("call", {}))
    bad, inexact = rf.check_result(fx, case, got)
    assert not bad, bad
    assert not inexact, ("same within 1e-4 m but not the reference's float64 bits", inexact)


@pytest.mark.parametrize("name", [c["name"] for c in rc.CASES])
def test_dump_and_yaml_round_trip_equal_the_references(fx, name):
    """``Stereo.dump(return_dict=True)`` writes the record the reference writes (same keys, same numbers: R, t, per camera
    fx / fy / cx / cy, D, xy, name), and loading the YAML text of a dump arrives where the reference's own load of ITS
    text arrives (stereo_camera.py:246-297, camera.py:407-448)."""
    import json
    case = rc.CASE_BY_NAME[name]
    st = _stereo(fx, case)
    assert json.dumps(st.dump(return_dict=True), sort_keys=True) == str(fx[name + "/dump_json"])
    assert np.array_equal(np.asarray(st.T, np.float64), fx[name + "/T"])  # R rounded through float32
    assert np.array_equal(st.depth_to_disparity(np.float64([0.5, 1.0, 2.5, 80.0])), fx[name + "/depth_to_disparity"])
    assert np.array_equal(st.D, np.zeros((1, 5)))
    text = st.dump()
    assert "\n" in text and "_calibrating_version" in text
    again = ca.Stereo(**case.get("stereo", {})).load(text)
    got = np.concatenate([np.asarray(getattr(again, k), np.float64).reshape(-1) for k in ("R", "t", "R1", "R2", "K")])
    assert np.array_equal(got, fx[name + "/yaml_roundtrip"])


def test_fixture_is_complete(fx):
    assert str(fx["reference_version"]) == "0.8.7"
    for c in rc.CASES:
        keys = [str(k) for k in fx[c["name"] + "/result_keys"]]
        want = set(rc.RESULT_KEYS) if c.get("call", {}).get("return_unrectify_depth", True) else \
            {"rectify_img1", "rectify_img2", "disparity", "rectify_depth"}
        assert want <= set(keys), (c["name"], keys)
    assert "confidence" in [str(k) for k in fx["foreign_dict/result_keys"]]
    # the cases exist for their branches: translation on and off, a non-trivial min_disparity, a downsizing ratio
    sc = {c["name"]: fx[c["name"] + "/scalars"] for c in rc.CASES}
    assert sc["translate_default"][1] == 1 and sc["translate_off"][1] == 0 and sc["translate_default"][0] == 14
    assert sc["translate_on_no_depth"][:2].tolist() == [0, 1] and sc["ktarget_scalar"][:2].tolist() == [0, 0]
    assert tuple(fx["xytarget_scalar/xy"]) == (403, 302) and tuple(fx["ktarget_matrix/xy"]) == (480, 352)


# ---- depth post-ops (n4) -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("rate", rc.POST_RATES)
def test_pointcloud_ref_depth_to_point_cloud(fx, rate):
    depth = rc.post_depth(1, rc.POST_XY1[1], rc.POST_XY1[0])
    got = pointcloud_ref.depth_to_point_cloud(depth, rc.POST_K1, interpolation_rate=rate, return_xyzuv=True)
    assert len(got) == int(fx["post/cloud_rate%s_n" % rate])
    assert rc.sha(got[:, 3:]) == str(fx["post/cloud_rate%s_uv_sha" % rate])
    assert np.array_equal(got[::rc.CLOUD_ROWS], fx["post/cloud_rate%s" % rate])


def test_pointcloud_ref_scatter_and_projection(fx):
    mm = (np.arange(12, dtype=np.uint16).reshape(3, 4) * 250)
    assert np.array_equal(pointcloud_ref.depth_to_point_cloud(mm, rc.POST_K1), fx["post/cloud_mm"])
    depth2 = rc.post_depth(2, rc.POST_XY1[1], rc.POST_XY1[0])
    cloud = pointcloud_ref.depth_to_point_cloud(depth2, rc.POST_K1)
    moved = pointcloud_ref.apply_T_to_point_cloud(rc.post_T(), cloud)
    assert np.array_equal(moved[::rc.CLOUD_ROWS], fx["post/moved"])
    s = rc.POST_SAMPLE
    assert np.array_equal(rc.sample(pointcloud_ref.point_cloud_to_depth(cloud, rc.POST_K1, rc.POST_XY1), s), fx["post/back"])
    assert np.array_equal(rc.sample(pointcloud_ref.point_cloud_to_depth(moved, rc.POST_K1, rc.POST_XY1), s),
                          fx["post/moved_depth"])
    depth3 = rc.post_depth(3, rc.POST_XY2[1], rc.POST_XY2[0])
    for interp in rc.POST_INTERPOLATIONS:
        got = pointcloud_ref.project_cam2_depth(rc.POST_K1, rc.POST_XY1, rc.POST_K2, depth3, fx["post/T2"], interp)
        assert np.array_equal(rc.sample(got, s), fx["post/project_%s" % interp]), interp
