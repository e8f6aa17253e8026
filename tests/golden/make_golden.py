#!/usr/bin/env python3
"""Writes the golden fixtures under tests/golden/.

Provenance: cv2 is not importable where this repo was authored and the reference holds no golden
vectors for the stereo path, so these vectors are outputs of the repo's own
CPU oracle (oracle/*.c) on seeded synthetic inputs -- they freeze the oracle's behaviour (so an
accidental change to it is caught) and let the GPU parity tests run against data as well as against
the live oracle.  They are NOT cv2 outputs: parity with cv2 itself stays unpinned.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle  # noqa: E402
from calibrating_amd import synthetic  # noqa: E402

NAMES = ["minDisparity", "numDisparities", "blockSize", "P1", "P2", "disp12MaxDiff", "preFilterCap",
         "uniquenessRatio", "speckleWindowSize", "speckleRange", "mode"]


def main():
    oracle.build()
    cases = [
        # H, W, D, cn, params
        (48, 256, 128, 1, dict(minDisparity=0, numDisparities=128, blockSize=5, P1=200, P2=800,
                               disp12MaxDiff=1, uniquenessRatio=10, mode=0)),
        (48, 256, 128, 3, dict(minDisparity=0, numDisparities=128, blockSize=5, P1=600, P2=2400,
                               disp12MaxDiff=1, uniquenessRatio=10, mode=1)),
        (40, 320, 218, 3, dict(minDisparity=2, numDisparities=218, blockSize=11, P1=968, P2=3872,
                               disp12MaxDiff=0, uniquenessRatio=5, speckleWindowSize=200, speckleRange=2)),
        (36, 160, 64, 3, dict(minDisparity=0, numDisparities=64, blockSize=5, P1=600, P2=2400,
                              disp12MaxDiff=1, uniquenessRatio=10, speckleWindowSize=100, speckleRange=2)),
        (30, 128, 48, 1, dict(minDisparity=-7, numDisparities=48, blockSize=7, P1=392, P2=1568,
                              disp12MaxDiff=1, uniquenessRatio=10, mode=1)),
    ]
    out = dict(n=len(cases), param_names=np.array(NAMES))
    for i, (H, W, D, cn, p) in enumerate(cases):
        left, right = synthetic.rectified_pair(seed=100 + i, H=H, W=W, D=D, cn=cn)
        full = {k: 0 for k in NAMES}
        full.update(p)
        out["left_%d" % i] = left
        out["right_%d" % i] = right
        out["params_%d" % i] = np.array([full[k] for k in NAMES], np.int32)
        out["disp_%d" % i] = oracle.sgbm_compute(left, right, **full)
    np.savez_compressed(os.path.join(HERE, "sgbm_small.npz"), **out)

    rng = np.random.default_rng(42)
    src = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    yy, xx = np.mgrid[:40, :56].astype(np.float32)
    mapx = (xx * 1.1 + rng.uniform(-4, 4, xx.shape)).astype(np.float32)
    mapy = (yy * 1.15 + rng.uniform(-4, 4, yy.shape)).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "remap_small.npz"), src=src, mapx=mapx, mapy=mapy,
                        lanczos4=oracle.remap_u8(src, mapx, mapy, 4), linear=oracle.remap_u8(src, mapx, mapy, 1),
                        lanczos4_itab=oracle.lanczos4_itab())
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
