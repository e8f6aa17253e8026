"""Depth post-ops around ``get_depth`` on the GPU -- same names and argument meaning as the reference's
``utils.depth_to_point_cloud`` / ``apply_T_to_point_cloud`` / ``point_cloud_to_depth``
(/root/reference/calibrating/utils.py:152-161, 213-318) and the interpolation-rate rule of
``utils._get_appropriate_interpolation_rate`` (:201-210).  float64 like the reference's NumPy.
NumPy in -> NumPy out, torch CUDA tensors in -> tensors out.
"""
import ctypes

import numpy as np

from . import _native, hostio


def _to_dev(a, dtype):
    import torch
    if isinstance(a, np.ndarray):
        _native.require_device()
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda(), True
    if not a.is_cuda:
        raise ValueError("tensor inputs must live on the GPU")
    return a.to(dtype).contiguous(), False


def _mat(m, n):
    a = np.ascontiguousarray(m, np.float64).reshape(-1)
    if a.size != n:
        raise ValueError("expected %d matrix entries, got %d" % (n, a.size))
    return a


def get_appropriate_interpolation_rate(cam1, cam2, interpolation=1.5):
    """utils._get_appropriate_interpolation_rate (utils.py:201-210)."""
    if interpolation:
        rate = cam1.K[0, 0] / cam2.K[0, 0] * interpolation
        if interpolation >= 1:
            rate = max(rate, 1)
    else:
        rate = 1
    return rate


def depth_to_point_cloud(depth, K, interpolation_rate=1, return_xyzuv=False):
    """(N, 3) points of the non-zero depths in row-major order, or (N, 5) ``xyzuv`` (utils.py:213-246).
    uint16 depth is millimetres, as in the reference."""
    import torch
    if isinstance(depth, np.ndarray) and depth.dtype == np.uint16:
        depth = np.float32(depth / 1000.0)
    d, was_np = _to_dev(depth, torch.float64)
    if d.dim() != 2:
        raise AssertionError("depth.ndim == 2")
    h, w = d.shape
    lib = _native.lib()
    rate = float(interpolation_rate)
    gw, gh = ctypes.c_int(), ctypes.c_int()
    _native.check(lib.camd_point_cloud_grid(w, h, rate, ctypes.byref(gw), ctypes.byref(gh)), "depth_to_point_cloud")
    cap = gw.value * gh.value
    Kinv = np.ascontiguousarray(np.linalg.inv(np.asarray(K, np.float64)[:3, :3])).reshape(9)
    with torch.cuda.device(d.device):
        pts = torch.empty((cap, 3), dtype=torch.float64, device=d.device)
        uv = torch.empty((cap, 2), dtype=torch.float64, device=d.device) if return_xyzuv else None
        count = torch.zeros(1, dtype=torch.int64, device=d.device)
        ws = torch.empty(lib.camd_point_cloud_workspace_bytes(w, h, rate), dtype=torch.uint8, device=d.device)
        rc = lib.camd_depth_to_point_cloud(d.data_ptr(), w, h, Kinv.ctypes.data, rate, pts.data_ptr(),
                                           None if uv is None else uv.data_ptr(), cap, count.data_ptr(), ws.data_ptr(),
                                           _native.current_stream())
    _native.check(rc, "depth_to_point_cloud")
    n = int(count.item())  # synchronises: the output length is data dependent
    out = torch.cat([pts[:n], uv[:n]], dim=1) if return_xyzuv else pts[:n]
    return hostio.to_host(out) if was_np else out


def apply_T_to_point_cloud(T, point_cloud):
    """(T @ [p, 1])[:3] for every row; extra columns are carried over (utils.py:152-161)."""
    import torch
    p, was_np = _to_dev(point_cloud, torch.float64)
    xyz = p[:, :3].contiguous()
    out = torch.empty_like(xyz)
    Tm = _mat(T, 16)
    with torch.cuda.device(p.device):
        rc = _native.lib().camd_apply_T_to_point_cloud(xyz.data_ptr(), xyz.shape[0], Tm.ctypes.data, out.data_ptr(),
                                                       _native.current_stream())
    _native.check(rc, "apply_T_to_point_cloud")
    if p.shape[1] > 3:
        out = torch.cat([out, p[:, 3:]], dim=1)
    return hostio.to_host(out) if was_np else out


def point_cloud_to_depth(points, K, xy, bg_value=0):
    """Depth image (xy[1], xy[0]) float64 of a point cloud: nearest z per pixel (utils.py:249-318)."""
    import torch
    p, was_np = _to_dev(points, torch.float64)
    if p.dim() != 2 or p.shape[1] < 3:
        raise ValueError("points must be (N, >=3)")
    w, h = int(xy[0]), int(xy[1])
    Km = _mat(np.asarray(K, np.float64)[:3, :3], 9)
    with torch.cuda.device(p.device):
        depth = torch.empty((h, w), dtype=torch.float64, device=p.device)
        keys = torch.empty((h, w), dtype=torch.int64, device=p.device)
        rc = _native.lib().camd_point_cloud_to_depth(p.data_ptr(), p.shape[0], p.shape[1], Km.ctypes.data, w, h,
                                                     float(bg_value), depth.data_ptr(), keys.data_ptr(),
                                                     _native.current_stream())
    _native.check(rc, "point_cloud_to_depth")
    return hostio.to_host(depth) if was_np else depth


def project_depth(depth2, K2, T_2in1, K1, xy1, interpolation_rate=1):
    """depth image of camera 2 seen from camera 1: depth_to_point_cloud -> apply_T -> point_cloud_to_depth
    (camera.py:298-309) as one scatter pass over the sampling grid."""
    import torch
    d, was_np = _to_dev(depth2, torch.float64)
    h2, w2 = d.shape
    w1, h1 = int(xy1[0]), int(xy1[1])
    K2inv = np.ascontiguousarray(np.linalg.inv(np.asarray(K2, np.float64)[:3, :3])).reshape(9)
    Tm, K1m = _mat(T_2in1, 16), _mat(np.asarray(K1, np.float64)[:3, :3], 9)
    with torch.cuda.device(d.device):
        depth1 = torch.empty((h1, w1), dtype=torch.float64, device=d.device)
        keys = torch.empty((h1, w1), dtype=torch.int64, device=d.device)
        rc = _native.lib().camd_project_depth(d.data_ptr(), w2, h2, K2inv.ctypes.data, Tm.ctypes.data, K1m.ctypes.data,
                                              float(interpolation_rate), w1, h1, depth1.data_ptr(), keys.data_ptr(),
                                              _native.current_stream())
    _native.check(rc, "project_depth")
    return hostio.to_host(depth1) if was_np else depth1
