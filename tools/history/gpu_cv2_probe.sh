#!/bin/bash
# Passive probe for OpenCV on the GPU box.  The reference pins opencv-contrib-python>=4.7.0.72
# (/root/reference/requirements.txt:2).  Installing packages is outside what this build is allowed to do (no network,
# no pip install), so the probe only looks for an existing cv2; if one is ever present the golden files are exported.
mkdir -p gpurun_out
{
  echo "== python -c 'import cv2'"; python -c 'import cv2; print(cv2.__version__)' 2>&1 | tail -1
  echo "== files named cv2* / *opencv* on the box"
  find / \( -iname 'cv2*' -o -iname '*opencv*' \) -not -path '/proc/*' -not -path '*/gpurun_out/*' -not -path '*/tests/*' -not -path '*/tools/*' 2>/dev/null | head -5
  echo "== pip config (index disabled by the image)"; pip config list 2>&1
  if python -c 'import cv2' 2>/dev/null; then
    python tools/export_cv2_golden.py 2>&1 | tail -60
    mkdir -p gpurun_out/cv2_golden && cp tests/golden/cv2_*.npz gpurun_out/cv2_golden/ 2>/dev/null
  else
    echo "cv2 is not present on the GPU box; package installs are not permitted in this environment -> parity stays unpinned"
  fi
} > gpurun_out/cv2_probe.log 2>&1
cat gpurun_out/cv2_probe.log
