"""The HIP path against the committed golden fixtures (-m gpu): every stage of tests/golden/cases.py, through the
same Python surface a user calls.  `oracle_<stage>.npz` is always there; `cv2_<stage>.npz` -- OpenCV's own outputs,
written by tools/export_cv2_golden.py -- is compared as soon as it is committed (that comparison is the BASELINE
metric max |disparity - cv2.SGBM|; skipped until then)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import cases  # noqa: E402


def _check(producer, stage, exact):
    data = cases.load(producer, stage)
    if data is None:
        pytest.skip("tests/golden/%s_%s.npz absent -- run `python tools/export_cv2_golden.py` where cv2 is installed and "
                    "commit what it writes; it pins: %s" % (producer, stage, cases.WHAT_IT_PINS[stage]))
    for name, ins, outs in data:
        got = cases.run("gpu", dict(name=name, stage=stage, inputs=ins))
        for k, want in outs.items():
            g = got[k].cpu().numpy() if hasattr(got[k], "cpu") else np.asarray(got[k])
            assert g.shape == want.shape, (stage, name, k, g.shape, want.shape)
            d = np.abs(np.asarray(g, np.float64) - np.asarray(want, np.float64)).max() if want.size else 0.0
            tol = 0 if exact else cases.tolerance(stage, k, want.dtype)
            if stage in ("maps", "rodrigues", "resize") and not np.issubdtype(want.dtype, np.integer):
                tol = max(tol, 1e-4 if stage != "rodrigues" else 1e-12)  # float stages: SURVEY 8(d) tolerances
            # (a committed fixture that differs FAILS -- it never skips; the message names the switch to look at)
            assert d <= tol, "%s/%s/%s: max |gpu - %s| = %g (tolerance %g).  %s" % (
                stage, name, k, producer, d, tol, cases.WHAT_IT_PINS[stage] if producer == "cv2" else "")


@pytest.mark.parametrize("stage", cases.STAGES)
def test_gpu_matches_oracle_fixtures(stage):
    # integer stages bit-exact; float32 maps / float resizes to 1e-4 (same formulas, FMA-free, but f32 rounding of
    # intermediate products may differ between gcc and hipcc)
    _check("oracle", stage, exact=stage in ("sgbm", "remap", "undistort", "post"))


@pytest.mark.parametrize("stage", cases.STAGES)
def test_gpu_matches_cv2_fixtures(stage):
    _check("cv2", stage, exact=False)
