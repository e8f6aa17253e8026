"""Shared by the ground-truth tests: what counts as "the depth of a rendered plane was recovered".

The scene is rendered by per-pixel ray casting through the Brown model (calibrating_amd.synthetic.render_plane_pair:
no remap, no interpolation, no matcher), so the checks below hold the whole path -- rectifying rotations, map
conventions, the min_disparity shift of the right image, disparity -> depth, M = R1^T K^-1 and the unrectify maps --
against geometry, not against this repo's restatement of OpenCV.  The reference checks its own depth the same way
against measured patches (/root/reference/example/test_depth_accuracy.py:101-108).

Error budget, in pixels of disparity (depth error = z^2 / (b fx) * disparity error):
  * cv2's fixed point contributes 1/16;
  * the Birchfield-Tomasi cost is by design insensitive to shifts of up to half a pixel, so on a fronto-parallel
    plane (ONE disparity everywhere) the parabola fit locks towards the integer and the error is systematic, up to
    0.5 px; on a slanted plane the fractional part sweeps and the error averages out.
Hence: slanted plane -> 90 % of the valid pixels within 1/16 + 3/16 px and |mean signed error| <= 0.05 px (a sign or
half-pixel convention error anywhere on the path shows as >= 0.5 px here); fronto-parallel -> 99 % within
1/16 + 7/16 px.  Coverage: >= 70 % of the image valid.
"""
import numpy as np

PLANES = {"slanted": ((0.3, 0.1, 1.0), 2.0, 3.0 / 16, 0.90), "fronto": ((0.0, 0.0, 1.0), 2.0, 7.0 / 16, 0.99)}
CFG = dict(minDisparity=0, numDisparities=64, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1, uniquenessRatio=10,
           speckleWindowSize=100, speckleRange=2)
MAX_DEPTH = 3.5


def check_depth(depth, z_true, baseline, fx, eps_px, frac, what):
    valid = depth > 0
    assert valid.mean() >= 0.70, "%s: only %.1f %% of the image has a depth" % (what, 100 * valid.mean())
    bound = z_true[valid] ** 2 / (baseline * fx) * (1.0 / 16 + eps_px)
    within = (np.abs(depth[valid] - z_true[valid]) <= bound).mean()
    assert within >= frac, "%s: %.2f %% of the valid pixels within (1/16 + %.4f) px of the true depth, need %.0f %%" % (
        what, 100 * within, eps_px, 100 * frac)
    signed_px = baseline * fx / depth[valid] - baseline * fx / z_true[valid]
    return float(valid.mean()), float(within), float(signed_px.mean())


def rectified_truth(K, R1, normal, distance, wh):
    """Depth along the RECTIFIED camera's optical axis of the plane n.X = n.(0, 0, distance) (camera-1 coordinates),
    per rectified pixel: the frame of get_depth's `rectify_depth`."""
    w, h = wh
    n = np.asarray(normal, np.float64)
    n = n / np.linalg.norm(n)
    v, u = np.mgrid[:h, :w].astype(np.float64)
    rays = np.stack([(u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], np.ones_like(u)], -1)
    return n[2] * distance / ((rays @ R1) @ n)  # X1 = R1^T X_rect
