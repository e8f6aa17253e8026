"""CPU oracle against the golden fixtures of tests/golden/ (catalogue: tests/golden/cases.py).

`oracle_<stage>.npz` (made by this repo's oracle, always present) must reproduce bit for bit: it freezes the
oracle.  `cv2_<stage>.npz` (made by tools/export_cv2_golden.py on a machine with OpenCV) is the parity pin: when
present the oracle must match it within the stage's stated tolerance (0 for SGBM, the BASELINE metric); when absent
the test is skipped and parity stays "unpinned" (DESIGN.md section 2)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import cases  # noqa: E402


@pytest.mark.parametrize("stage", cases.STAGES)
def test_oracle_reproduces_its_own_fixtures(oracle, stage):
    data = cases.load("oracle", stage)
    assert data is not None, "run python tests/golden/make_golden.py"
    assert [n for n, _, _ in data] == [c["name"] for c in cases.all_cases() if c["stage"] == stage], \
        "catalogue and fixture file disagree: regenerate with python tests/golden/make_golden.py"
    for name, ins, outs in data:
        got = cases.run("oracle", dict(name=name, stage=stage, inputs=ins))
        for k, want in outs.items():
            assert np.array_equal(np.asarray(got[k]), want), (stage, name, k)


def test_fixture_inputs_match_the_catalogue():
    """The committed inputs are the ones the catalogue generates today (NumPy generator drift would show here)."""
    data = {n: ins for st in cases.STAGES for n, ins, _ in cases.load("oracle", st)}
    for c in cases.all_cases():
        for k, v in c["inputs"].items():
            assert np.array_equal(np.asarray(v), data[c["name"]][k]), (c["name"], k)


@pytest.mark.parametrize("stage", cases.STAGES)
def test_oracle_matches_cv2_fixtures(oracle, stage):
    data = cases.load("cv2", stage)
    if data is None:
        pytest.skip("tests/golden/cv2_%s.npz absent: run `python tools/export_cv2_golden.py` where cv2 is installed and "
                    "commit what it writes; it pins: %s" % (stage, cases.WHAT_IT_PINS[stage]))
    for name, ins, outs in data:
        got = cases.run("oracle", dict(name=name, stage=stage, inputs=ins))
        for k, want in outs.items():
            d = np.abs(np.asarray(got[k], np.float64) - np.asarray(want, np.float64)).max() if want.size else 0.0
            assert d <= cases.tolerance(stage, k, want.dtype), "%s/%s/%s: max |oracle - cv2| = %g.  %s" % (
                stage, name, k, d, cases.WHAT_IT_PINS[stage])


def test_export_tool_plumbing(tmp_path, monkeypatch, oracle):
    """tools/export_cv2_golden.py end to end with a stand-in `cv2` module that answers from the oracle: checks the
    tool's plumbing (every catalogue case is exported and re-loadable), not OpenCV."""
    import types
    fake = types.ModuleType("cv2")
    fake.__version__, fake.getNumThreads = "stand-in", lambda: 1
    fake.INTER_NEAREST, fake.INTER_LINEAR, fake.INTER_LANCZOS4, fake.CV_32FC1 = 0, 1, 4, 5

    class _M:
        def __init__(self, **p):
            self.p = p

        def compute(self, l, r):
            return oracle.sgbm_compute(l, r, **self.p)

    fake.StereoSGBM_create = lambda **p: _M(**p)
    fake.remap = lambda s, mx, my, i: (oracle.remap_nearest_f64(s, mx, my) if s.dtype == np.float64
                                       else oracle.remap_u8(s, mx, my, i))
    fake.initUndistortRectifyMap = lambda A, d, R, An, size, t: oracle.init_undistort_rectify_map(A, d, R, An, size)
    fake.undistort = lambda s, K, d: oracle.undistort_u8(s, K, d)
    fake.resize = lambda s, wh, interpolation=1: oracle.resize_linear(s, wh[::-1])
    fake.medianBlur = lambda a, k: oracle.median3_s16(a)

    def _fs(a, nv, ms, md):
        a[...] = oracle.filter_speckles_s16(a, nv, ms, md)
    fake.filterSpeckles = _fs
    from calibrating_amd import geometry
    fake.Rodrigues = lambda r: (geometry.rodrigues(r), None)
    monkeypatch.setitem(sys.modules, "cv2", fake)
    monkeypatch.setattr(cases, "HERE", str(tmp_path))
    sys.path.insert(0, os.path.join(cases.ROOT, "tools"))
    import export_cv2_golden
    export_cv2_golden.main()
    for st in cases.STAGES:
        got = cases.load("cv2", st)
        assert got is not None and len(got) == sum(1 for c in cases.all_cases() if c["stage"] == st)
