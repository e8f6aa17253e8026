"""Reading side of tests/golden/reference_plumbing.npz -- outputs of the REFERENCE's own Python (made in the build
container by tests/golden/make_reference_golden.py; see its header for what that pins and what it does not).

Test infrastructure.  ``check_rig`` / ``check_result`` compare a ``calibrating_amd.Stereo`` (or any object with the same
attributes) and a result dict with what the reference produced for the same case: the whole array through its SHA-256,
and -- so that a mismatch says where -- every ``SAMPLE``-th row and column value by value.  Images and the disparity
must be bit-identical; depths must have the same zero (= invalid) set and agree within 1e-4 m (BASELINE.json's
tolerance; what is actually reached is reported by ``inexact``).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import reference_cases as rc  # noqa: E402

DEPTH_TOL = 1e-4  # metres, BASELINE.json north_star


def fixture():
    fx = rc.load_fixture()
    assert fx is not None, "tests/golden/reference_plumbing.npz is missing (python tests/golden/make_reference_golden.py)"
    return fx


def load_args(fx, case, as_given=False):
    """(record for ``Stereo.load``, Stereo kwargs): the very numbers the reference was given.

    A ``T`` record goes matrix -> Rodrigues vector -> matrix inside ``load`` (stereo_camera.py:287-291), and the
    matrix -> vector direction of cv2.Rodrigues is a log map that two correct implementations round differently (the
    generator's stand-in: SciPy; here: geometry.rodrigues; ~1e-15).  So unless ``as_given``, a T case is loaded with the
    ``R`` the reference ended up with -- everything downstream of that can then be compared bit for bit -- and
    ``as_given=True`` is used once, to check that the T spelling arrives at that R within rounding."""
    rec = rc.rig_record(case)
    form = case.get("record", "R")
    if form == "T" and not as_given:
        rec["R"], rec["t"] = fx[case["name"] + "/R"].tolist(), fx[case["name"] + "/t"].tolist()
        form = "R"
    if form == "r":
        rec.pop("R")
        rec["r"] = fx[case["name"] + "/record_r"].tolist()
    elif form == "T":
        rec.pop("R")
        rec.pop("t")
        rec["T"] = fx[case["name"] + "/record_T"]
    return rec, dict(case.get("stereo", {}))


def images(fx, case):
    img1, img2 = rc.images(case)
    name = case["name"]
    assert rc.sha(img1) == str(fx[name + "/img1_sha"]) and rc.sha(img2) == str(fx[name + "/img2_sha"]), \
        "the regenerated input scene differs from the one the reference was run on (synthetic.render_plane_pair changed?)"
    return img1, img2


def check_rig(fx, case, st, maps=True):
    """Rig state derived at load / set_stereo_matching time.  Returns a list of problems (empty = identical)."""
    name, bad = case["name"], []
    for k, tol in (("R", 0), ("t", 0), ("R1", 0), ("R2", 0), ("K", 0)):  # the same float64 bits
        want = fx["%s/%s" % (name, k)]
        got = np.asarray(getattr(st, k), np.float64).reshape(want.shape)
        if np.abs(got - want).max() > tol:
            bad.append("%s differs by %.3g" % (k, np.abs(got - want).max()))
    if tuple(int(v) for v in st.xy) != tuple(fx[name + "/xy"]):
        bad.append("xy %s != %s" % (tuple(st.xy), tuple(fx[name + "/xy"])))
    min_disp, translate, max_depth, baseline, get_max = fx[name + "/scalars"]
    if (st.min_disparity, float(bool(st.translation_rectify_img)), float(st.max_depth), float(st.get_max_depth())) != \
            (min_disp, translate, max_depth, get_max):
        bad.append("min_disparity / translation / max_depth: %s" % (
            (st.min_disparity, st.translation_rectify_img, st.max_depth, st.get_max_depth()),))
    if abs(float(st.baseline) - baseline) > 1e-16:
        bad.append("baseline")
    if maps:
        h, w = int(st.xy[1]), int(st.xy[0])
        mask = np.unpackbits(fx[name + "/mask_bits"])[:h * w].reshape(h, w).astype(bool)
        if not np.array_equal(np.asarray(st.rectify_valid_mask1, bool), mask):
            bad.append("rectify_valid_mask1: %d pixels" % (np.asarray(st.rectify_valid_mask1, bool) != mask).sum())
        for i, m in ((1, st.undistort_rectify_map1), (2, st.undistort_rectify_map2)):
            for ax, got in zip("xy", m):
                bad += _cmp("map%d%s" % (i, ax), np.asarray(got), str(fx["%s/map%d%s_sha" % (name, i, ax)]),
                            fx["%s/map%d%s" % (name, i, ax)], rc.MAP_SAMPLE, exact=True)[0]
    return bad


def _cmp(key, got, want_sha, want_sample, step, exact):
    """-> (problems, inexact): ``exact`` entries must reproduce the hash; depth entries may differ within DEPTH_TOL."""
    if rc.sha(got) == want_sha:
        return [], []
    g = rc.sample(got, step)
    if g.shape != want_sample.shape or g.dtype != want_sample.dtype:
        return ["%s: shape / dtype %s %s, the reference returned %s %s (sampled)" % (
            key, g.shape, g.dtype, want_sample.shape, want_sample.dtype)], []
    diff = g != want_sample
    if exact:
        return ["%s: not the reference's bits (%d of %d sampled values differ, first at %s)" % (
            key, diff.sum(), diff.size, tuple(np.argwhere(diff)[0]) if diff.any() else "an unsampled position")], []
    if not np.array_equal(g == 0, want_sample == 0):
        return ["%s: the invalid (zero) sets differ at %d sampled pixels" % (key, ((g == 0) != (want_sample == 0)).sum())], []
    err = float(np.abs(g - want_sample).max())
    if err > DEPTH_TOL:
        return ["%s: %.3g m off (tolerance %g)" % (key, err, DEPTH_TOL)], []
    return [], ["%s (max %.3g m on the samples)" % (key, err)]


def check_result(fx, case, res, keys=None):
    """A ``get_depth`` result dict against the reference's.  -> (problems, inexact depth entries)."""
    name, bad, inexact = case["name"], [], []
    want_keys = [str(k) for k in fx[name + "/result_keys"]]
    if keys is None:
        if sorted(res) != want_keys:
            bad.append("result keys %s != the reference's %s" % (sorted(res), want_keys))
        keys = [k for k in want_keys if k in res]
    for k in keys:
        pre = "%s/out/%s" % (name, k)
        if pre + "_value" in fx:  # non-array extras of a plugin's dict
            if repr(res[k]) != str(fx[pre + "_value"]):
                bad.append("%s: %r != %s" % (k, res[k], fx[pre + "_value"]))
            continue
        got = np.asarray(res[k])
        ds = [str(s) for s in fx[pre + "_dtype_shape"]]
        if [got.dtype.str] + [str(s) for s in got.shape] != ds:
            bad.append("%s: dtype / shape %s %s, the reference returned %s" % (k, got.dtype.str, got.shape, ds))
            continue
        b, i = _cmp(k, got, str(fx[pre + "_sha"]), fx[pre], rc.SAMPLE, exact=not k.endswith("depth"))
        bad += b
        inexact += i
    return bad, inexact
