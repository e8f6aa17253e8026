#!/usr/bin/env python3
"""One command that pins parity with OpenCV: run it on ANY machine where `import cv2` works.

    python tools/export_cv2_golden.py            # writes tests/golden/cv2_<stage>.npz and prints a report

For every case of the golden catalogue (tests/golden/cases.py: SGBM at the BASELINE configs and at every
uncertainty flag of SURVEY.md Appendix A.14, Lanczos / bilinear / nearest remaps incl. a read-back of the whole
fixed-point weight table, initUndistortRectifyMap, undistort, resize, medianBlur, filterSpeckles, Rodrigues) it
stores the inputs together with what cv2 itself returns -- the calls the reference makes on its depth path
(calibrating/stereo_matching.py:48-63, stereo_camera.py:159-165,217-228,431, utils.py:184-199).
Commit the files: from then on `pytest tests/ -m "not gpu"` compares the CPU oracle with them
(tests/test_golden_cpu.py) and `pytest tests/ -m gpu` the HIP kernels (tests/test_gpu_golden.py); the BASELINE
metric max |disparity - cv2.SGBM| is the `sgbm` stage.  The report printed here already says how the repo's CPU
oracle compares (needs gcc for oracle/, not a GPU).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    try:
        import cv2
    except ImportError:
        sys.exit("export_cv2_golden.py needs OpenCV (pip install opencv-contrib-python>=4.7.0.72, the reference's pin)")
    import cases
    print("cv2 %s, %d threads" % (cv2.__version__, cv2.getNumThreads()))
    written = []
    for p in cases.write("cv2"):
        written.append(os.path.relpath(p, ROOT))
        print("wrote %s (%d KB)" % (written[-1], os.path.getsize(p) // 1024))
    # report: the repo's oracle against what was just written
    try:
        worst = {}
        for st in cases.STAGES:
            for name, ins, outs in cases.load("cv2", st):
                got = cases.run("oracle", dict(name=name, stage=st, inputs=ins))
                for k, want in outs.items():
                    d = float(np.abs(np.asarray(got[k], np.float64) - np.asarray(want, np.float64)).max()) if want.size else 0.0
                    tol = cases.tolerance(st, k, want.dtype)
                    flag = "ok " if d <= tol else "DIFF"
                    print("  %s %-9s %-32s %-8s max|oracle - cv2| = %g (tolerance %g)" % (flag, st, name, k, d, tol))
                    worst[st] = max(worst.get(st, 0.0), d)
        print("worst per stage:", worst)
        for st in cases.STAGES:
            if worst.get(st, 0.0) > 0:
                print("  %-9s differs somewhere -> %s" % (st, cases.WHAT_IT_PINS[st]))
    except Exception as e:  # the export itself succeeded; the comparison needs gcc + this repo's oracle
        print("oracle comparison skipped:", e)
    print("\nCommit exactly these files (data only: inputs and what cv2 %s returned):\n  git add %s\n"
          "then `python -m pytest tests -m 'not gpu'` checks the CPU oracle against them (tests/test_golden_cpu.py) and\n"
          "`python -m pytest tests -m gpu` the HIP kernels (tests/test_gpu_golden.py; the sgbm stage IS the BASELINE metric\n"
          "max |disparity - cv2.SGBM|).  A fixture that differs fails, it does not skip." % (cv2.__version__, " ".join(written)))


if __name__ == "__main__":
    main()
