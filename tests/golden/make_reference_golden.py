#!/usr/bin/env python3
"""Runs the REFERENCE's own Python on the stereo-depth path and writes tests/golden/reference_plumbing.npz.

BUILD CONTAINER ONLY (it needs /root/reference, which does not travel to the GPU box; only the .npz does).
    python tests/golden/make_reference_golden.py

What is executed.  ``/root/reference/calibrating`` is imported FROM WHERE IT LIES, unmodified -- nothing of it is copied
into this repository -- and its own code runs every case of tests/golden/reference_cases.py:
    Stereo.load(record)  [Cam.load, _get_undistort_rectify_map incl. better_cx_cy, stereo_recitfy, utils.project_vec_on_plane,
                          utils.rotate_shortest_of_two_vecs, utils.T_to_r_t]                 stereo_camera.py:125-185,199-214,264-297
    Stereo.set_stereo_matching(SemiGlobalBlockMatching(cfg) | foreign plugin, max_depth, translation_rectify_img)   :466-489
    Stereo.get_depth(img1, img2[, return_unrectify_depth])  [rectify, the plugin's __call__, += min_disparity, mask,
                          disparity_to_depth, unrectify_depth -> utils.rotate_depth_by_remap, undistort_img]      :492-533,
                          :216-242, :408-431, stereo_matching.py:22-70, utils.py:173-200
    utils.depth_to_point_cloud / apply_T_to_point_cloud / point_cloud_to_depth, Cam.project_cam2_depth        utils.py:152-161,
                          :201-318, camera.py:298-309

What stands in for the reference's two missing dependencies.  The reference imports ``cv2`` (opencv-contrib-python
>= 4.7.0.72, requirements.txt:2) and ``boxx`` (>= 0.10.6, requirements.txt:1) at module top; neither is installable
here (no wheel, no network: SURVEY.md F3).  This script puts stand-in modules of those names into ``sys.modules``
whose entry points ON THE PATH are backed by this repo's CPU oracle (oracle/*.c, oracle/pointcloud_ref.resize_nearest):
    cv2.initUndistortRectifyMap, cv2.remap (u8 Lanczos-4 / f64 nearest), cv2.undistort, cv2.resize,
    cv2.StereoSGBM_create(...).compute / .getMinDisparity, cv2.Rodrigues (written here from OpenCV's published formula,
    cross-checked against scipy.spatial.transform.Rotation), boxx.resize (recollection: SURVEY A.13, flag U9), boxx.npa,
    boxx.inpkg (no-op context), boxx.increase.  Everything else of cv2 / boxx / tqdm is an inert placeholder that
    raises when called (none is reached on the path).

WHAT THIS PINS AND WHAT IT DOES NOT.  It pins every line of the reference's OWN Python between those entry points:
the rig geometry (R1, R2, K, xy incl. better_cx_cy), the valid mask, min_disparity / translation rules, the order and
dtype of every NumPy operation in get_depth / disparity_to_depth / rotate_depth_by_remap / the matcher's post-processing
and downsizing branch, the result-dict contract, the quirks Q1-Q9 of SURVEY.md section 3.6, and the depth post-ops
(pure NumPy but for one cv2.resize).  It does NOT pin cv2's arithmetic: behind the entry points sits the same oracle
the other tests use, so "parity with cv2" stays UNPINNED.
"""
import contextlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import oracle  # noqa: E402  (test infrastructure; this script lives under tests/)
from oracle import pointcloud_ref  # noqa: E402
import reference_cases as rc  # noqa: E402

REFERENCE = "/root/reference"


# ---- stand-in modules ------------------------------------------------------------------------------------------
class _Inert:
    """Placeholder for every cv2 / boxx name the path never reaches: attribute access works (module-level code of the
    reference touches e.g. cv2.aruco.DICT_*), calling it raises."""

    def __init__(self, name):
        self._name = name

    def __getattr__(self, attr):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return _Inert(self._name + "." + attr)

    def __call__(self, *a, **k):
        raise RuntimeError("stand-in %s was called: it is not on the stereo-depth path" % self._name)

    def __mro_entries__(self, bases):
        return (object,)


class _StandIn(types.ModuleType):
    def __getattr__(self, attr):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return _Inert(self.__name__ + "." + attr)


def rodrigues(src):
    """cv2.Rodrigues(src) -> (dst, jacobian=None).  Vector -> matrix in OpenCV's operation order (calib3d, cvRodrigues2:
    theta = |r|; itheta = 1 / theta; r *= itheta; R = c I + (1 - c) r r^T + s [r]x); matrix -> vector through SciPy."""
    a = np.asarray(src, np.float64)
    if a.size == 3:
        r = a.reshape(3).copy()
        theta = float(np.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]))
        if theta < np.finfo(np.float64).eps:
            return np.eye(3), None
        c, s = np.cos(theta), np.sin(theta)
        r *= 1.0 / theta
        rrt = np.outer(r, r)
        rx = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
        return c * np.eye(3) + (1.0 - c) * rrt + s * rx, None
    from scipy.spatial.transform import Rotation
    return Rotation.from_matrix(a.reshape(3, 3)).as_rotvec().reshape(3, 1), None


class _StereoSGBM:
    def __init__(self, **p):
        self.p = dict(p)

    def compute(self, left, right):
        return oracle.sgbm_compute(left, right, **self.p)

    def getMinDisparity(self):
        return self.p.get("minDisparity", 0)


def _cv2_remap(src, map1, map2, interpolation, *a, **k):
    assert not a and not k, "the path passes no border arguments"
    if src.dtype == np.float64 and interpolation == oracle.INTER_NEAREST:
        return oracle.remap_nearest_f64(src, map1, map2)
    assert src.dtype == np.uint8
    return oracle.remap_u8(src, map1, map2, interpolation)


def _cv2_resize(src, dsize, interpolation=oracle.INTER_LINEAR):
    if interpolation == oracle.INTER_NEAREST:
        return pointcloud_ref.resize_nearest(src, dsize)
    assert interpolation == oracle.INTER_LINEAR
    if tuple(src.shape[:2]) == (dsize[1], dsize[0]):
        return src.copy()
    return oracle.resize_linear(src, (dsize[1], dsize[0]))


def _boxx_resize(img, arg2, interpolation=None):
    """boxx.resize as recalled: a number scales both sides (1 -> the image itself), a pair is (h, w)."""
    if isinstance(arg2, (int, float)):
        if arg2 == 1:
            return img
        hw = (int(round(img.shape[0] * arg2)), int(round(img.shape[1] * arg2)))
    elif hasattr(arg2, "shape"):
        hw = tuple(arg2.shape[:2])
    else:
        hw = (int(arg2[0]), int(arg2[1]))
    return _cv2_resize(img, (hw[1], hw[0]), oracle.INTER_LINEAR if interpolation is None else interpolation)


_counters = {}


def _increase(name):
    _counters[name] = _counters.get(name, -1) + 1
    return _counters[name]


def install_stand_ins():
    cv2 = _StandIn("cv2")
    cv2.INTER_NEAREST, cv2.INTER_LINEAR, cv2.INTER_LANCZOS4 = oracle.INTER_NEAREST, oracle.INTER_LINEAR, oracle.INTER_LANCZOS4
    cv2.CV_32FC1 = 5
    cv2.Rodrigues = rodrigues
    cv2.initUndistortRectifyMap = lambda K, D, R, Knew, size, m1type: oracle.init_undistort_rectify_map(
        np.asarray(K, np.float64), None if D is None else np.asarray(D, np.float64), R, Knew, size)
    cv2.remap = _cv2_remap
    cv2.undistort = lambda img, K, D: oracle.undistort_u8(img, K, D)
    cv2.resize = _cv2_resize
    cv2.StereoSGBM_create = lambda **p: _StereoSGBM(**p)
    boxx = _StandIn("boxx")
    boxx.np = np
    boxx.pi = np.pi
    boxx.inpkg = contextlib.nullcontext
    boxx.npa = np.array
    boxx.resize = _boxx_resize
    boxx.increase = _increase
    tqdm = _StandIn("tqdm")
    tqdm.tqdm = lambda it, *a, **k: it
    for m in (cv2, boxx, tqdm):
        sys.modules[m.__name__] = m
    sys.modules["cv2.aruco"] = _StandIn("cv2.aruco")


def import_reference():
    install_stand_ins()
    sys.path.insert(0, REFERENCE)
    import calibrating
    assert os.path.dirname(os.path.abspath(calibrating.__file__)).startswith(REFERENCE)
    return calibrating


# ---- running the cases -----------------------------------------------------------------------------------------
def _np_record(case, rec):
    """The record in the form the reference's loaders need it (ndarrays for ``T``; lists elsewhere are fine)."""
    form = case.get("record", "R")
    if form != "R":
        R = np.array(rec.pop("R"), np.float64)
        if form == "r":
            rec["r"] = rodrigues(R)[0].reshape(3).tolist()
        else:
            T = np.eye(4)
            T[:3, :3] = R
            T[:3, 3] = np.array(rec.pop("t"), np.float64).reshape(3)
            rec["T"] = T
    return rec


def record_for(case, as_arrays=True):
    """(record handed to ``Stereo.load``, extra arrays to store so that the consumer loads the very same numbers)."""
    rec = _np_record(case, rc.rig_record(case))
    return rec


def run_case(cal, case, out):
    name = case["name"]
    rec = record_for(case)
    for k in ("r", "T"):  # what the consumers must feed (the Rodrigues vector is made here)
        if k in rec:
            out["%s/record_%s" % (name, k)] = np.array(rec[k], np.float64)
    st = cal.Stereo(**case.get("stereo", {}))
    st.load({k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in rec.items()})
    kind, cfg = case["plugin"]
    st.set_stereo_matching(rc.make_plugin(kind, cfg, cal.MetaStereoMatching, cal.SemiGlobalBlockMatching), **case["setm"])
    img1, img2 = rc.images(case)
    res = st.get_depth(img1.copy(), img2.copy(), **case.get("call", {}))
    # -- rig state the reference derived
    for k in ("R", "t", "R1", "R2", "K"):
        out["%s/%s" % (name, k)] = np.array(getattr(st, k), np.float64)
    out[name + "/xy"] = np.array(st.xy, np.int64)
    # -- the record the reference writes back (Stereo.dump / Cam.dump, stereo_camera.py:246-262, camera.py:407-422), and
    #    what its own load makes of that YAML text (:264-297)
    out[name + "/T"] = np.asarray(st.T, np.float64)                       # utils.R_t_to_T: R through float32 (Q9)
    out[name + "/depth_to_disparity"] = np.asarray(st.depth_to_disparity(np.float64([0.5, 1.0, 2.5, 80.0])), np.float64)
    import json
    out[name + "/dump_json"] = np.array(json.dumps(st.dump(return_dict=True), sort_keys=True))
    again = cal.Stereo(**case.get("stereo", {})).load(st.dump())
    out[name + "/yaml_roundtrip"] = np.concatenate([np.asarray(getattr(again, k), np.float64).reshape(-1) for k in ("R", "t", "R1", "R2", "K")])
    out[name + "/scalars"] = np.array([st.min_disparity, float(st.translation_rectify_img), st.max_depth, st.baseline,
                                       st.get_max_depth()], np.float64)
    out[name + "/mask_bits"] = np.packbits(st.rectify_valid_mask1)
    for i, maps in ((1, st.undistort_rectify_map1), (2, st.undistort_rectify_map2)):
        for ax, m in zip("xy", maps):
            out["%s/map%d%s_sha" % (name, i, ax)] = np.array(rc.sha(m))
            out["%s/map%d%s" % (name, i, ax)] = rc.sample(m, rc.MAP_SAMPLE)
    if hasattr(st, "_unrectify_depth_maps"):
        for ax, m in zip("xy", st._unrectify_depth_maps):
            out["%s/unrect_map%s_sha" % (name, ax)] = np.array(rc.sha(m))
            out["%s/unrect_map%s" % (name, ax)] = rc.sample(m, rc.MAP_SAMPLE)
    # -- inputs (hash only: regenerated by the consumer) and the result dict
    out[name + "/img1_sha"], out[name + "/img2_sha"] = np.array(rc.sha(img1)), np.array(rc.sha(img2))
    keys = sorted(res)
    out[name + "/result_keys"] = np.array(keys)
    for k in keys:
        v = res[k]
        if isinstance(v, np.ndarray):
            out["%s/out/%s_sha" % (name, k)] = np.array(rc.sha(v))
            out["%s/out/%s_dtype_shape" % (name, k)] = np.array([v.dtype.str] + [str(s) for s in v.shape])
            out["%s/out/%s" % (name, k)] = rc.sample(v)
        else:
            out["%s/out/%s_value" % (name, k)] = np.array(repr(v))
    valid = float((res["rectify_depth"] > 0).mean())
    print("%-24s xy=%s min_disp=%d translate=%d  valid depth %.2f  keys=%s" % (
        name, tuple(st.xy), st.min_disparity, st.translation_rectify_img, valid, ",".join(keys)))
    return st, res


def run_post_ops(cal, out):
    u = cal.utils
    depth = rc.post_depth(1, rc.POST_XY1[1], rc.POST_XY1[0])
    for rate in rc.POST_RATES:
        cloud = u.depth_to_point_cloud(depth, rc.POST_K1, interpolation_rate=rate, return_xyzuv=True)
        out["post/cloud_rate%s_n" % rate] = np.array(len(cloud))
        out["post/cloud_rate%s" % rate] = cloud[::rc.CLOUD_ROWS]
        out["post/cloud_rate%s_uv_sha" % rate] = np.array(rc.sha(cloud[:, 3:]))
    mm = (np.arange(12, dtype=np.uint16).reshape(3, 4) * 250)
    out["post/cloud_mm"] = u.depth_to_point_cloud(mm, rc.POST_K1)
    depth2 = rc.post_depth(2, rc.POST_XY1[1], rc.POST_XY1[0])
    cloud = u.depth_to_point_cloud(depth2, rc.POST_K1)
    T = rc.post_T()
    moved = u.apply_T_to_point_cloud(T, cloud)
    out["post/moved"] = moved[::rc.CLOUD_ROWS]
    out["post/back"] = rc.sample(u.point_cloud_to_depth(cloud, rc.POST_K1, rc.POST_XY1), rc.POST_SAMPLE)
    out["post/moved_depth"] = rc.sample(u.point_cloud_to_depth(moved, rc.POST_K1, rc.POST_XY1), rc.POST_SAMPLE)
    cam1 = cal.Cam.init_by_K_D(rc.POST_K1, np.zeros((1, 5)), rc.POST_XY1, name="a")
    cam2 = cal.Cam.init_by_K_D(rc.POST_K2, np.zeros((1, 5)), rc.POST_XY2, name="b")
    depth3 = rc.post_depth(3, rc.POST_XY2[1], rc.POST_XY2[0])
    T2 = np.eye(4)
    T2[:3, :3] = rodrigues(np.array([0.01, 0.03, -0.02]))[0]
    T2[:3, 3] = [-0.05, 0.0, 0.01]
    out["post/T2"] = T2
    for interp in rc.POST_INTERPOLATIONS:
        out["post/project_%s" % interp] = rc.sample(cam1.project_cam2_depth(cam2, depth3, T=T2, interpolation=interp),
                                                     rc.POST_SAMPLE)
    print("post-ops: %d clouds, project_cam2_depth x%d" % (len(rc.POST_RATES) + 2, len(rc.POST_INTERPOLATIONS)))


def main():
    oracle.build()
    cal = import_reference()
    out = {"reference_version": np.array(cal.__version__), "sample": np.array(rc.SAMPLE)}
    # the stand-in Rodrigues against an independent implementation
    from scipy.spatial.transform import Rotation
    for r in ([0.01, -0.02, 0.005], [1.2, 0.4, -2.0], [3.1, 0.0, 0.2]):
        assert np.abs(rodrigues(np.array(r))[0] - Rotation.from_rotvec(r).as_matrix()).max() < 1e-15
    for case in rc.CASES:
        run_case(cal, case, out)
    run_post_ops(cal, out)
    np.savez_compressed(rc.FIXTURE, **out)
    print("wrote %s (%d KB, %d arrays)" % (os.path.relpath(rc.FIXTURE, ROOT), os.path.getsize(rc.FIXTURE) // 1024, len(out)))


if __name__ == "__main__":
    main()
