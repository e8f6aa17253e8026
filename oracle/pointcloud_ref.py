"""CPU oracle of the depth post-ops -- TEST INFRASTRUCTURE ONLY (imported by tests/ alone).

NumPy restatement of the reference's own NumPy code, /root/reference/calibrating/utils.py:152-161
(apply_T_to_point_cloud), :201-210 (interpolation rate), :213-246 (depth_to_point_cloud), :249-318
(point_cloud_to_depth -> point_cloud_to_arr2d -> uvzs_to_arr2d) and camera.py:298-309
(Cam.project_cam2_depth).  cv2.resize(INTER_NEAREST) is restated from OpenCV's resizeNN
(sx = min(floor(x * (1 / (dst_w / src_w))), src_w - 1)): PARITY UNPINNED against cv2 itself.
"""
import numpy as np


def resize_nearest(src, dsize_wh):
    """cv2.resize(src, (w, h), interpolation=cv2.INTER_NEAREST)"""
    dw, dh = int(dsize_wh[0]), int(dsize_wh[1])
    sh, sw = src.shape[:2]
    ifx, ify = 1.0 / (dw / sw), 1.0 / (dh / sh)
    xs = np.minimum(np.floor(np.arange(dw) * ifx).astype(np.int64), sw - 1)
    ys = np.minimum(np.floor(np.arange(dh) * ify).astype(np.int64), sh - 1)
    return src[ys[:, None], xs[None, :]]


def depth_to_point_cloud(depth, K, interpolation_rate=1, return_xyzuv=False):
    assert depth.ndim == 2
    rows, cols = depth.shape
    if depth.dtype == np.uint16:
        depth = np.float32(depth / 1000.0)
    if interpolation_rate == 1:
        nz = depth != 0
        vs, us = np.mgrid[:rows, :cols][:, nz]
        zs = depth[nz]
    else:
        rows_, cols_ = int(round(rows * interpolation_rate)), int(round(cols * interpolation_rate))
        up = resize_nearest(depth, (cols_, rows_))
        nz = up != 0
        vs, us = np.mgrid[:rows_, :cols_][:, nz] / interpolation_rate
        zs = up[nz]
    homog = (np.array([us, vs, np.ones_like(us)]) * zs).T
    cloud = (np.linalg.inv(K) @ homog.T).T
    if return_xyzuv:
        return np.concatenate([cloud, us[:, None], vs[:, None]], -1)
    return cloud


def apply_T_to_point_cloud(T, cloud):
    n4 = np.ones((len(cloud), 4))
    n4[:, :3] = cloud[:, :3]
    moved = (T @ n4.T).T[:, :3]
    if cloud.shape[1] > 3:
        moved = np.concatenate((moved, cloud[:, 3:]), -1)
    return moved


def point_cloud_to_depth(points, K, xy, bg_value=0):
    proj = points[:, :3] @ K.T
    proj[:, :2] /= proj[:, 2:]
    far_first = proj[np.argsort(-proj[:, 2])]
    h, w = xy[1], xy[0]
    img = np.ones((h, w), far_first.dtype) * bg_value
    xs, ys = np.int32(far_first[:, :2].round()).T
    ok = (xs >= 0) & (xs < w) & (ys >= 0) & (ys < h)
    img[ys[ok], xs[ok]] = far_first[ok][:, 2]
    return img


def interpolation_rate(K1, K2, interpolation=1.5):
    if not interpolation:
        return 1
    rate = K1[0, 0] / K2[0, 0] * interpolation
    return max(rate, 1) if interpolation >= 1 else rate


def project_cam2_depth(K1, xy1, K2, depth2, T, interpolation=1.5):
    cloud2 = depth_to_point_cloud(depth2, K2, interpolation_rate=interpolation_rate(K1, K2, interpolation))
    return point_cloud_to_depth(apply_T_to_point_cloud(T, cloud2), K1, xy1)
