#!/usr/bin/env python3
"""Writes the ORACLE-made golden fixtures tests/golden/oracle_<stage>.npz (catalogue: tests/golden/cases.py).

Provenance: cv2 is not importable where this repo was authored and the reference holds no golden vectors for the
stereo path, so these vectors are outputs of the repo's own CPU oracle (oracle/*.c) on the
catalogue's inputs -- they freeze the oracle's behaviour (an accidental change to it is caught) and give the GPU
parity tests data to run against.  They are NOT cv2 outputs: `python tools/export_cv2_golden.py` writes the
cv2-made twins (cv2_<stage>.npz) on any machine that has cv2, and the same tests then pin parity with cv2.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import cases  # noqa: E402

if __name__ == "__main__":
    for p in cases.write("oracle"):
        print("wrote %s (%d KB)" % (os.path.relpath(p, cases.ROOT), os.path.getsize(p) // 1024))
