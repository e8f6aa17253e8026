"""cv2-shaped front ends of the remap / filter / depth kernels (host side: torch tensors in HBM).

Each function names the cv2 / NumPy call of the reference it stands in for; all of them work on
torch CUDA tensors (zero-copy) or NumPy arrays (copied to the GPU and back).
"""
import ctypes

import numpy as np

from . import _native, hostio
from ._native import INTER_LANCZOS4, INTER_LINEAR, INTER_NEAREST


def _to_dev(a, dtype=None):
    """(tensor on GPU, was_numpy)"""
    import torch
    if isinstance(a, np.ndarray):
        _native.require_device()
        t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
        was_np = True
    else:
        t, was_np = a, False
    if not t.is_cuda:
        raise ValueError("tensor inputs must live on the GPU")
    if dtype is not None and t.dtype != dtype:
        raise ValueError("expected dtype %s, got %s" % (dtype, t.dtype))
    return t.contiguous(), was_np


def _img_dims(t):
    """(batch, h, w, cn, batched) of a (h,w) | (h,w,c) | (n,h,w,c) uint8 image tensor."""
    if t.dim() == 2:
        return 1, t.shape[0], t.shape[1], 1, False
    if t.dim() == 3:
        return 1, t.shape[0], t.shape[1], t.shape[2], False
    if t.dim() == 4:
        return t.shape[0], t.shape[1], t.shape[2], t.shape[3], True
    raise ValueError("unsupported image shape %s" % (tuple(t.shape),))


def remap(src, mapx, mapy, interpolation=INTER_LANCZOS4, x_shift=0):
    """cv2.remap(src, mapx, mapy, interpolation) for uint8 images, CV_32FC1 maps, BORDER_CONSTANT 0
    (stereo_camera.py:217-228).  ``x_shift`` fuses stereo_camera.py:230-240."""
    import torch
    s, was_np = _to_dev(src, torch.uint8)
    mx, _ = _to_dev(mapx, torch.float32)
    my, _ = _to_dev(mapy, torch.float32)
    if mx.shape != my.shape or mx.dim() != 2:
        raise ValueError("mapx / mapy must be 2-D float32 arrays of equal shape")
    n, sh, sw, cn, batched = _img_dims(s)
    dh, dw = mx.shape
    shape = (n, dh, dw) + ((cn,) if s.dim() > 2 else ())
    dst = torch.empty(shape, dtype=torch.uint8, device=s.device)
    with torch.cuda.device(s.device):
        rc = _native.lib().camd_remap_u8(s.data_ptr(), sw, sh, cn, sw * cn, sh * sw * cn, mx.data_ptr(),
                                         my.data_ptr(), dst.data_ptr(), dw, dh, dw * cn, dh * dw * cn,
                                         int(interpolation), int(x_shift), n, _native.current_stream())
    _native.check(rc, "remap")
    dst = dst if batched else dst[0]
    return hostio.to_host(dst) if was_np else dst


def _dist_args(D):
    D = np.zeros(0) if D is None else np.ascontiguousarray(D, np.float64).reshape(-1)
    return D, (D.ctypes.data if D.size else None), int(D.size)


def init_undistort_rectify_map(A, dist, R, Anew, size, valid_for=None, device=None):
    """cv2.initUndistortRectifyMap(A, dist, R, Anew, size, CV_32FC1) built on the GPU
    (stereo_camera.py:159-165, utils.py:184-191): returns (mapx, mapy) float32 CUDA tensors (h, w), plus
    the uint8 valid mask of stereo_camera.py:167-176 when ``valid_for=(src_w, src_h)`` is given."""
    import torch
    _native.require_device()
    w, h = int(size[0]), int(size[1])
    A = np.ascontiguousarray(A, np.float64).reshape(9)
    Anew = np.ascontiguousarray(np.asarray(Anew, np.float64)[:, :3]).reshape(9)
    Rm = None if R is None else np.ascontiguousarray(R, np.float64).reshape(9)
    D, dptr, nd = _dist_args(dist)
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    mapx = torch.empty((h, w), dtype=torch.float32, device=dev)
    mapy = torch.empty((h, w), dtype=torch.float32, device=dev)
    mask = torch.empty((h, w), dtype=torch.uint8, device=dev) if valid_for is not None else None
    sw, sh = (int(valid_for[0]), int(valid_for[1])) if valid_for is not None else (0, 0)
    with torch.cuda.device(dev):
        rc = _native.lib().camd_init_undistort_rectify_map(
            A.ctypes.data, dptr, nd, None if Rm is None else Rm.ctypes.data, Anew.ctypes.data, w, h,
            mapx.data_ptr(), mapy.data_ptr(), None if mask is None else mask.data_ptr(), sw, sh,
            _native.current_stream())
    _native.check(rc, "init_undistort_rectify_map")
    return (mapx, mapy) if mask is None else (mapx, mapy, mask)


def undistort_maps_device(K, D, size, device=None):
    """The CV_16SC2 + CV_16UC1 maps of cv2.undistort, built on the GPU: (mapxy int16 (h,w,2), mapa int16 view)."""
    import torch
    _native.require_device()
    w, h = int(size[0]), int(size[1])
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    D, dptr, nd = _dist_args(D)
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    mxy = torch.empty((h, w, 2), dtype=torch.int16, device=dev)
    ma = torch.empty((h, w), dtype=torch.int16, device=dev)  # uint16 bit patterns (torch has no uint16 arithmetic)
    with torch.cuda.device(dev):
        rc = _native.lib().camd_undistort_maps(K.ctypes.data, dptr, nd, w, h, mxy.data_ptr(), ma.data_ptr(),
                                               _native.current_stream())
    _native.check(rc, "undistort_maps")
    return mxy, ma


def undistort_maps(K, D, size):
    """The CV_16SC2 + CV_16UC1 maps cv2.undistort(img, K, D) builds internally (host, init time)."""
    w, h = int(size[0]), int(size[1])
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    D = np.zeros(0) if D is None else np.ascontiguousarray(D, np.float64).reshape(-1)
    mxy = np.empty((h, w, 2), np.int16)
    ma = np.empty((h, w), np.uint16)
    rc = _native.lib().camd_undistort_maps_host(K.ctypes.data, D.ctypes.data if D.size else None, D.size, w, h,
                                                mxy.ctypes.data, ma.ctypes.data)
    _native.check(rc, "undistort_maps")
    return mxy, ma


def remap_fixed_bilinear(src, mapxy, mapa):
    """cv2.remap(src, map16SC2, map16UC1, INTER_LINEAR): the second half of cv2.undistort
    (stereo_camera.py:430-431)."""
    import torch
    s, was_np = _to_dev(src, torch.uint8)
    mxy, _ = _to_dev(mapxy, torch.int16)
    ma, _ = _to_dev(mapa.view(np.int16) if isinstance(mapa, np.ndarray) else mapa, torch.int16)
    n, sh, sw, cn, batched = _img_dims(s)
    dh, dw = ma.shape
    shape = (n, dh, dw) + ((cn,) if s.dim() > 2 else ())
    dst = torch.empty(shape, dtype=torch.uint8, device=s.device)
    with torch.cuda.device(s.device):
        rc = _native.lib().camd_remap_fixed_bilinear_u8(
            s.data_ptr(), sw, sh, cn, sw * cn, sh * sw * cn, mxy.data_ptr(), ma.data_ptr(), dst.data_ptr(), dw,
            dh, dw * cn, dh * dw * cn, n, _native.current_stream())
    _native.check(rc, "remap_fixed_bilinear")
    dst = dst if batched else dst[0]
    return hostio.to_host(dst) if was_np else dst


def medianBlur3_s16(disp):
    """cv2.medianBlur(disp, 3) on int16 (the unconditional tail of StereoSGBM.compute)."""
    import torch
    s, was_np = _to_dev(disp, torch.int16)
    h, w = s.shape[-2:]
    n = s.numel() // (h * w)
    dst = torch.empty_like(s)
    with torch.cuda.device(s.device):
        rc = _native.lib().camd_median3_s16(s.data_ptr(), dst.data_ptr(), w, h, n, _native.current_stream())
    _native.check(rc, "medianBlur3_s16")
    return hostio.to_host(dst) if was_np else dst


def filterSpeckles(disp, newVal, maxSpeckleSize, maxDiff):
    """cv2.filterSpeckles(disp, newVal, maxSpeckleSize, maxDiff) on int16; returns a new array."""
    import torch
    s, was_np = _to_dev(disp, torch.int16)
    s = s.clone()
    h, w = s.shape[-2:]
    n = s.numel() // (h * w)
    ws = torch.empty(_native.lib().camd_speckle_workspace_bytes(w, h, n), dtype=torch.uint8, device=s.device)
    with torch.cuda.device(s.device):
        rc = _native.lib().camd_filter_speckles_s16(s.data_ptr(), w, h, int(newVal), int(maxSpeckleSize),
                                                    int(maxDiff), ws.data_ptr(), n, _native.current_stream())
    _native.check(rc, "filterSpeckles")
    return hostio.to_host(s) if was_np else s


def disp_to_depth(disp16, valid_mask, sgbm_min_disparity, add_min_disparity, translate, baseline_fx,
                  max_depth):
    """stereo_matching.py:63-69 + stereo_camera.py:510-513 in one pass.
    Returns (disparity float32, rectify_depth float64)."""
    import torch
    d, was_np = _to_dev(disp16, torch.int16)
    m, _ = _to_dev(valid_mask.view(np.uint8) if isinstance(valid_mask, np.ndarray) and valid_mask.dtype == bool
                   else valid_mask)
    if m.dtype == torch.bool:
        m = m.view(torch.uint8)
    h, w = d.shape[-2:]
    n = d.numel() // (h * w)
    disparity = torch.empty(d.shape, dtype=torch.float32, device=d.device)
    depth = torch.empty(d.shape, dtype=torch.float64, device=d.device)
    with torch.cuda.device(d.device):
        rc = _native.lib().camd_disp_to_depth(d.data_ptr(), m.data_ptr(), w, h, int(sgbm_min_disparity),
                                              int(add_min_disparity), int(bool(translate)),
                                              ctypes.c_double(baseline_fx), ctypes.c_double(max_depth),
                                              disparity.data_ptr(), depth.data_ptr(), n,
                                              _native.current_stream())
    _native.check(rc, "disp_to_depth")
    if was_np:
        return tuple(hostio.to_host(disparity, depth))
    return disparity, depth


def disp16_resized_to_depth(sdisp16, hw, valid_mask, sgbm_min_disparity, add_min_disparity, translate, baseline_fx,
                            max_depth):
    """The matcher's downsizing branch (stereo_matching.py:63-69 with max_size < image) + stereo_camera.py:510-513 +
    :408-413 in one pass: ``sdisp16`` is the int16 disparity of the downsized pair(s), ``hw`` the rectified size.
    Returns (disparity float32, rectify_depth float64) at ``hw``.  CUDA tensors only."""
    import torch
    d, _ = _to_dev(sdisp16, torch.int16)
    m, _ = _to_dev(valid_mask)
    if m.dtype == torch.bool:
        m = m.view(torch.uint8)
    sh, sw = d.shape[-2:]
    n = d.numel() // (sh * sw)
    h, w = int(hw[0]), int(hw[1])
    disparity = torch.empty(d.shape[:-2] + (h, w), dtype=torch.float32, device=d.device)
    depth = torch.empty(d.shape[:-2] + (h, w), dtype=torch.float64, device=d.device)
    with torch.cuda.device(d.device):
        rc = _native.lib().camd_disp16_resized_to_depth(d.data_ptr(), sw, sh, m.data_ptr(), w, h,
                                                        int(sgbm_min_disparity), int(add_min_disparity),
                                                        int(bool(translate)), ctypes.c_double(baseline_fx),
                                                        ctypes.c_double(max_depth), disparity.data_ptr(),
                                                        depth.data_ptr(), n, _native.current_stream())
    _native.check(rc, "disp16_resized_to_depth")
    return disparity, depth


def unrectify_depth(depth, M_row2, mapx, mapy):
    """utils.rotate_depth_by_remap (utils.py:192-199): z-rescale + INTER_NEAREST remap, float64."""
    import torch
    z, was_np = _to_dev(depth, torch.float64)
    mx, _ = _to_dev(mapx, torch.float32)
    my, _ = _to_dev(mapy, torch.float32)
    h, w = z.shape[-2:]
    n = z.numel() // (h * w)
    oh, ow = mx.shape
    M = (ctypes.c_double * 3)(*[float(v) for v in np.asarray(M_row2).reshape(3)])
    out = torch.empty(z.shape[:-2] + (oh, ow), dtype=torch.float64, device=z.device)
    with torch.cuda.device(z.device):
        rc = _native.lib().camd_unrectify_depth(z.data_ptr(), w, h, M, mx.data_ptr(), my.data_ptr(),
                                                out.data_ptr(), ow, oh, n, _native.current_stream())
    _native.check(rc, "unrectify_depth")
    return hostio.to_host(out) if was_np else out
