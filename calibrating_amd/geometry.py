"""Init-time geometry of the stereo rig, host side, float64 NumPy.

Restates the helpers ``Stereo`` needs once per rig:
  /root/reference/calibrating/utils.py:139-140   project_vec_on_plane
  /root/reference/calibrating/utils.py:143-149   rotate_shortest_of_two_vecs (-> cv2.Rodrigues)
  cv2.Rodrigues (vector <-> matrix), cv2.initUndistortRectifyMap(..., CV_32FC1)
      called at stereo_camera.py:159-165 and utils.py:184-191.
These run once per calibration; per-pair work is on the GPU.
"""
import numpy as np

eps = 1e-8


def project_vec_on_plane(v, plane_v):
    return v - np.dot(v, plane_v) / (np.linalg.norm(plane_v) ** 2) * plane_v


def rodrigues(r):
    """cv2.Rodrigues(r)[0]: rotation vector (3,) -> matrix, or matrix (3,3) -> vector (3,1)."""
    r = np.asarray(r, np.float64)
    if r.size == 3:
        r = r.reshape(3)
        theta = np.linalg.norm(r)
        if theta < np.finfo(np.float64).eps:
            return np.eye(3)
        k = r / theta
        c, s = np.cos(theta), np.sin(theta)
        kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        return c * np.eye(3) + (1 - c) * np.outer(k, k) + s * kx
    R = r.reshape(3, 3)
    # project onto SO(3) like cv2 does, then log map
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    rv = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.sqrt(np.sum(rv ** 2) * 0.25)
    c = np.clip((np.trace(R) - 1) * 0.5, -1, 1)
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            out = np.zeros(3)
        else:
            t = (R + np.eye(3)) * 0.5
            v = np.sqrt(np.maximum(np.diag(t), 0))
            v[1] *= -1 if R[0, 1] < 0 else 1
            v[2] *= -1 if R[0, 2] < 0 else 1
            if abs(v[0]) < abs(v[1]) and abs(v[0]) < abs(v[2]) and (R[1, 2] > 0) != (v[1] * v[2] > 0):
                v[2] = -v[2]
            out = v * (theta / max(np.linalg.norm(v), 1e-300))
    else:
        out = rv * (0.5 / s) * theta
    return out.reshape(3, 1)


def rotate_shortest_of_two_vecs(v1, v2, return_rodrigues=False):
    cross = np.cross(v1, v2)
    rad = np.arccos((v1 * v2).sum() / np.linalg.norm(v1) / np.linalg.norm(v2))
    r = rad * cross / (np.linalg.norm(cross) + eps)
    if return_rodrigues:
        return r
    return rodrigues(r)


def inv3(m):
    """3x3 inverse by the adjugate, the form cv::Mat::inv takes for 3x3 double matrices."""
    m = np.asarray(m, np.float64).reshape(9)
    d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6])
    d = 1.0 / d if d != 0 else 0.0
    return np.array([
        (m[4] * m[8] - m[5] * m[7]) * d, (m[2] * m[7] - m[1] * m[8]) * d, (m[1] * m[5] - m[2] * m[4]) * d,
        (m[5] * m[6] - m[3] * m[8]) * d, (m[0] * m[8] - m[2] * m[6]) * d, (m[2] * m[3] - m[0] * m[5]) * d,
        (m[3] * m[7] - m[4] * m[6]) * d, (m[1] * m[6] - m[0] * m[7]) * d, (m[0] * m[4] - m[1] * m[3]) * d,
    ]).reshape(3, 3)


def init_undistort_rectify_map(A, dist, R, Anew, size):
    """cv2.initUndistortRectifyMap(A, dist, R, Anew, size, CV_32FC1) -> (mapx, mapy) float32 (h, w).

    float64 internally; X, Y, W are accumulated column by column like OpenCV's scalar loop."""
    w, h = int(size[0]), int(size[1])
    A = np.asarray(A, np.float64)
    Anew = np.asarray(Anew, np.float64)
    R = np.eye(3) if R is None else np.asarray(R, np.float64)
    k = np.zeros(14)
    if dist is not None:
        d = np.asarray(dist, np.float64).reshape(-1)
        k[:d.size] = d
    if k[12] != 0 or k[13] != 0:
        raise NotImplementedError("tilted sensor model (tauX, tauY) is not on this path")
    k1, k2, p1, p2, k3, k4, k5, k6, s1, s2, s3, s4 = k[:12]
    ir = inv3(Anew[:, :3] @ R).reshape(9)
    i = np.arange(h, dtype=np.float64)[:, None]

    def accumulate(start, step):
        a = np.empty((h, w))
        a[:, :1] = start
        a[:, 1:] = step
        return np.cumsum(a, axis=1)  # sequential: ((x0 + s) + s) + ...

    _x = accumulate(i * ir[1] + ir[2], ir[0])
    _y = accumulate(i * ir[4] + ir[5], ir[3])
    _w = accumulate(i * ir[7] + ir[8], ir[6])
    ww = 1.0 / _w
    x, y = _x * ww, _y * ww
    x2, y2 = x * x, y * y
    r2, _2xy = x2 + y2, 2 * x * y
    kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2)
    xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + s1 * r2 + s2 * r2 * r2
    yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + s3 * r2 + s4 * r2 * r2
    u = A[0, 0] * xd + A[0, 2]
    v = A[1, 1] * yd + A[1, 2]
    return u.astype(np.float32), v.astype(np.float32)


def R_t_to_T(R, t=None):
    if t is None:
        t = np.zeros((3,))
    R = np.float32(R)  # the reference rounds R through float32 here (utils.py:18)
    if R.size == 3:
        R = rodrigues(np.float64(R).reshape(3))
    T = np.zeros((4, 4))
    T[:3, :3] = R
    T[:3, -1] = np.array(t).squeeze()
    T[3, 3] = 1
    return T


def T_to_r_t(T):
    T = np.asarray(T, np.float64)
    rvec = rodrigues(T[:3, :3])
    tvec = T[:3, 3:]
    if not tvec.size:
        tvec = np.zeros((3, 1))
    return rvec, tvec
