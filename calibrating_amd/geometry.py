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
    """Component of ``v`` inside the plane whose normal is ``plane_v`` (utils.py:139-140)."""
    n = np.asarray(plane_v, np.float64)
    v = np.asarray(v, np.float64)
    return v - n * (np.dot(v, n) / np.linalg.norm(n) ** 2)


def rodrigues(r):
    """cv2.Rodrigues(r)[0]: rotation vector (3,) -> matrix, or matrix (3,3) -> vector (3,1)."""
    r = np.asarray(r, np.float64)
    if r.size == 3:
        r = r.reshape(3)
        theta = float(np.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]))
        if theta < np.finfo(np.float64).eps:
            return np.eye(3)
        k = r * (1.0 / theta)  # OpenCV multiplies by the reciprocal (cvRodrigues2: itheta = 1 / theta; r *= itheta)
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
    """Rotation taking the direction of ``v1`` onto ``v2`` about their common normal (utils.py:143-149):
    axis = v1 x v2 normalised with the reference's ``+ eps`` in the denominator, angle = the angle between."""
    a, b = np.asarray(v1, np.float64), np.asarray(v2, np.float64)
    axis = np.cross(a, b)
    angle = np.arccos(np.sum(a * b) / np.linalg.norm(a) / np.linalg.norm(b))
    rvec = angle * axis / (np.linalg.norm(axis) + eps)  # this order of operations: the reference's rounding
    return rvec if return_rodrigues else rodrigues(rvec)


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
    """4x4 pose from a rotation (matrix, or a 3-vector = Rodrigues) and a translation.  The reference
    rounds the rotation argument through float32 first (utils.py:15-31); a 3-vector is converted after
    that rounding and the resulting matrix rounded to float32 again, like cv2.Rodrigues on a float32 input."""
    rot = np.asarray(R, np.float32)
    if rot.size == 3:
        rot = rodrigues(rot.astype(np.float64).reshape(3)).astype(np.float32)
    T = np.eye(4)
    T[:3, :3] = rot
    if t is not None:
        T[:3, 3] = np.asarray(t, np.float64).reshape(3)
    return T


def T_to_r_t(T):
    """(rvec (3,1), tvec (3,1)) of a 4x4 pose or a 3x3 rotation."""
    T = np.asarray(T, np.float64)
    tvec = T[:3, 3:4] if T.shape[1] > 3 else np.zeros((3, 1))
    return rodrigues(T[:3, :3]), tvec


# ---- rig-level pure functions (what Stereo's methods of the reference's names are thin wrappers of) -------------
def rectifying_rotations(R, t):
    """(R1, R2): rotations of camera 1 / camera 2 into the common rectified frame.

    SURVEY.md section 3.2 (reference stereo_camera.py:199-214).  Frame of camera 2; ``R, t`` map camera-1
    coordinates into it.  The rectified x axis is the baseline (-x onto t), the rectified z axis is the
    bisector of the two optical axes after each has been projected into the plane normal to the baseline:
        Rx = shortest rotation of -x onto t
        Rz = shortest rotation of Rx.z onto that bisector
        R2 = (Rz Rx)^T ,  R1 = R2 R
    """
    R = np.asarray(R, np.float64)[:3, :3]
    base = np.asarray(t, np.float64).reshape(3)
    ez = np.array([0.0, 0.0, 1.0])
    in_plane = [project_vec_on_plane(axis, base) for axis in (ez, R @ ez)]  # camera 2's, camera 1's optical axis
    bisector = sum(v / np.linalg.norm(v) for v in in_plane)
    Rx = rotate_shortest_of_two_vecs(np.array([-1.0, 0.0, 0.0]), base)
    Rz = rotate_shortest_of_two_vecs(Rx @ ez, bisector)
    R2 = (Rz @ Rx).T
    return R2 @ R, R2


def _image_centre_shift(xy_cam, K_cam, R_rect, K_new):
    """Where the centroid of a camera's four image corners lands in the rectified image, relative to the
    rectified principal point (the ``get_center`` closure of stereo_camera.py:137-152)."""
    w, h = xy_cam
    corners = np.array([[0, 0, 1], [w, 0, 1], [w, h, 1], [0, h, 1]], np.float64)
    rays = corners @ np.linalg.inv(K_cam).T @ R_rect.T
    uvw = rays @ K_new.T
    return (uvw[:, :2] / uvw[:, 2:]).mean(0) - K_new[:2, 2]


def target_intrinsics(cam1_K, cam1_xy, cam2_K, cam2_xy, R1, R2, xy_target=None, K_target=1):
    """(xy, K) of the rectified image pair (reference stereo_camera.py:125-156).

    xy_target  None -> camera 1's size; a number scales it (each side rounded); else taken as (w, h).
    K_target   a number s -> camera 1's K with the upper-left 2x2 times s and the principal point moved by
               half the size change, then re-centred: the principal point is placed so that the mean of the
               two cameras' projected image-corner centroids sits at the image centre ("better cx cy").
               An ndarray is used as is (the same object, no re-centring).
    """
    if xy_target is None:
        xy_target = cam1_xy
    if isinstance(xy_target, (int, float)):
        xy_target = [int(round(side * xy_target)) for side in cam1_xy]
    xy = tuple(xy_target)
    if isinstance(K_target, np.ndarray):
        return xy, K_target
    K = K_target
    if isinstance(K_target, (int, float)):
        K = np.array(cam1_K, np.float64)
        K[:2, :2] *= K_target
        K[:2, 2] += (np.array(xy) - cam1_xy) / 2
    shift = (_image_centre_shift(cam1_xy, cam1_K, R1, K) + _image_centre_shift(cam2_xy, cam2_K, R2, K)) / 2
    K[:2, 2] = np.array(xy) / 2 - shift
    return xy, K


def valid_mask_from_maps(mapx, mapy, xy_src):
    """True where a remap source coordinate falls inside the source image (stereo_camera.py:166-176)."""
    w, h = xy_src
    return (mapx > -0.5) & (mapx < w - 0.5) & (mapy > -0.5) & (mapy < h - 0.5)


def rig_rotation_from_record(rec):
    """The rotation of a Stereo record: ``R`` (3x3), else ``T`` (4x4, also supplies ``t``), else ``r``
    (Rodrigues vector), else identity (stereo_camera.py:287-292).  Mutates and returns ``rec``."""
    if "R" not in rec and "T" in rec:
        rec["r"], rec["t"] = T_to_r_t(np.asarray(rec.pop("T"), np.float64))
    if "R" not in rec and "r" in rec:
        rec["R"] = rodrigues(np.asarray(rec.pop("r"), np.float64).reshape(3))
    rec.setdefault("R", np.eye(3))
    return rec
