"""ctypes front-end of the CPU oracle (oracle/*.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of bench.py -- never from ``calibrating_amd``.  PARITY UNPINNED against cv2 (cv2 is not
available where this was written and the reference holds no golden vectors; see oracle/oracle.h); the restatements of
the reference's own NumPy are pinned by the reference's own run (tests/golden/reference_plumbing.npz).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

MODE_SGBM = 0
MODE_HH = 1
INTER_NEAREST = 0
INTER_LINEAR = 1
INTER_LANCZOS4 = 4


class SgbmParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        "minDisparity", "numDisparities", "blockSize", "P1", "P2", "disp12MaxDiff",
        "preFilterCap", "uniquenessRatio", "speckleWindowSize", "speckleRange", "mode")]


class Switches(ctypes.Structure):
    _fields_ = [("lanczos_fix_group_lo", ctypes.c_int), ("bt_border_raw_tab0", ctypes.c_int),
                ("cost_saturate", ctypes.c_int), ("way3_stripes", ctypes.c_int), ("way3_simd_lanes", ctypes.c_int)]


def build(force=False):
    """Compile oracle/*.c with gcc (Makefile in this directory)."""
    srcs = [os.path.join(_HERE, f) for f in ("sgbm_ref.c", "remap_ref.c", "depth_ref.c", "resize_ref.c", "oracle.h")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _params(**kw):
    d = dict(minDisparity=0, numDisparities=16, blockSize=3, P1=0, P2=0, disp12MaxDiff=0,
             preFilterCap=0, uniquenessRatio=0, speckleWindowSize=0, speckleRange=0, mode=0)
    for k, v in kw.items():
        if k not in d:
            raise TypeError("unknown SGBM parameter %r" % k)
        d[k] = int(v)
    return SgbmParams(**d)


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def _prep_pair(left, right):
    left = np.ascontiguousarray(left, dtype=np.uint8)
    right = np.ascontiguousarray(right, dtype=np.uint8)
    assert left.shape == right.shape and left.ndim in (2, 3)
    h, w = left.shape[:2]
    cn = 1 if left.ndim == 2 else left.shape[2]
    return left, right, h, w, cn


def cost_dims(w, **kw):
    p = _params(**kw)
    minD, maxD = p.minDisparity, p.minDisparity + p.numDisparities
    minX1, maxX1 = max(maxD, 0), w + min(minD, 0)
    return minX1, maxX1 - minX1, p.numDisparities


def sgbm_compute(left, right, raw=False, **kw):
    """cv2.StereoSGBM_create(**kw).compute(left, right) -> int16 (h, w), disparity*16."""
    left, right, h, w, cn = _prep_pair(left, right)
    p = _params(**kw)
    disp = np.empty((h, w), np.int16)
    fn = lib().oracle_sgbm_raw if raw else lib().oracle_sgbm_compute
    rc = fn(ctypes.byref(p), _p(left, ctypes.c_uint8), _p(right, ctypes.c_uint8), w, h, cn,
            ctypes.c_size_t(w * cn), _p(disp, ctypes.c_int16))
    if rc:
        raise ValueError("oracle_sgbm_compute: bad arguments")
    return disp


def sgbm_compute_batch(lefts, rights, nthreads=1, **kw):
    lefts = np.ascontiguousarray(lefts, dtype=np.uint8)
    rights = np.ascontiguousarray(rights, dtype=np.uint8)
    n, h, w = lefts.shape[:3]
    cn = 1 if lefts.ndim == 3 else lefts.shape[3]
    p = _params(**kw)
    disp = np.empty((n, h, w), np.int16)
    rc = lib().oracle_sgbm_compute_batch(ctypes.byref(p), _p(lefts, ctypes.c_uint8),
                                         _p(rights, ctypes.c_uint8), w, h, cn, n, int(nthreads),
                                         _p(disp, ctypes.c_int16))
    if rc:
        raise ValueError("oracle_sgbm_compute_batch: bad arguments")
    return disp


def _volume(fn, left, right, **kw):
    left, right, h, w, cn = _prep_pair(left, right)
    p = _params(**kw)
    _, width1, D = cost_dims(w, **kw)
    vol = np.empty((h, max(width1, 0), D), np.int16)
    rc = fn(ctypes.byref(p), _p(left, ctypes.c_uint8), _p(right, ctypes.c_uint8), w, h, cn,
            ctypes.c_size_t(w * cn), _p(vol, ctypes.c_int16))
    if rc:
        raise ValueError("oracle: bad arguments")
    return vol


def sgbm_cost_volume(left, right, **kw):
    return _volume(lib().oracle_sgbm_cost_volume, left, right, **kw)


def sgbm_aggregated(left, right, **kw):
    return _volume(lib().oracle_sgbm_aggregated, left, right, **kw)


def median3_s16(img):
    img = np.ascontiguousarray(img, dtype=np.int16)
    out = np.empty_like(img)
    h, w = img.shape
    lib().oracle_median3_s16(_p(img, ctypes.c_int16), _p(out, ctypes.c_int16), w, h)
    return out


def filter_speckles_s16(img, new_val, max_speckle_size, max_diff):
    img = np.array(img, dtype=np.int16, order="C", copy=True)
    h, w = img.shape
    lib().oracle_filter_speckles_s16(_p(img, ctypes.c_int16), w, h, int(new_val),
                                     int(max_speckle_size), int(max_diff))
    return img


def remap_u8(src, mapx, mapy, interp=INTER_LANCZOS4):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    mapx = np.ascontiguousarray(mapx, dtype=np.float32)
    mapy = np.ascontiguousarray(mapy, dtype=np.float32)
    sh, sw = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    dh, dw = mapx.shape
    dst = np.empty((dh, dw) + src.shape[2:], np.uint8)
    rc = lib().oracle_remap_u8(_p(src, ctypes.c_uint8), sw, sh, cn, _p(mapx, ctypes.c_float),
                               _p(mapy, ctypes.c_float), _p(dst, ctypes.c_uint8), dw, dh, int(interp))
    if rc:
        raise ValueError("oracle_remap_u8: unsupported interpolation")
    return dst


def remap_nearest_f64(src, mapx, mapy):
    src = np.ascontiguousarray(src, dtype=np.float64)
    mapx = np.ascontiguousarray(mapx, dtype=np.float32)
    mapy = np.ascontiguousarray(mapy, dtype=np.float32)
    sh, sw = src.shape
    dh, dw = mapx.shape
    dst = np.empty((dh, dw), np.float64)
    lib().oracle_remap_nearest_f64(_p(src, ctypes.c_double), sw, sh, _p(mapx, ctypes.c_float),
                                   _p(mapy, ctypes.c_float), _p(dst, ctypes.c_double), dw, dh)
    return dst


def init_undistort_rectify_map(A, dist, R, Anew, size):
    w, h = int(size[0]), int(size[1])
    A = np.ascontiguousarray(A, np.float64).reshape(9)
    Anew = np.ascontiguousarray(Anew, np.float64).reshape(9)
    dist = None if dist is None else np.ascontiguousarray(dist, np.float64).reshape(-1)
    R = None if R is None else np.ascontiguousarray(R, np.float64).reshape(9)
    mapx = np.empty((h, w), np.float32)
    mapy = np.empty((h, w), np.float32)
    lib().oracle_init_undistort_rectify_map(
        _p(A, ctypes.c_double), None if dist is None else _p(dist, ctypes.c_double),
        0 if dist is None else dist.size, None if R is None else _p(R, ctypes.c_double),
        _p(Anew, ctypes.c_double), w, h, _p(mapx, ctypes.c_float), _p(mapy, ctypes.c_float))
    return mapx, mapy


def undistort_u8(src, K, dist):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    h, w = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    dist = None if dist is None else np.ascontiguousarray(dist, np.float64).reshape(-1)
    dst = np.empty_like(src)
    lib().oracle_undistort_u8(_p(src, ctypes.c_uint8), w, h, cn, _p(K, ctypes.c_double),
                              None if dist is None else _p(dist, ctypes.c_double),
                              0 if dist is None else dist.size, _p(dst, ctypes.c_uint8))
    return dst


def resize_linear(src, dsize_hw):
    """cv2.resize(src, (w, h), interpolation=cv2.INTER_LINEAR) for uint8 HWC / HW or float32 HW."""
    dh, dw = int(dsize_hw[0]), int(dsize_hw[1])
    if src.dtype == np.uint8:
        src = np.ascontiguousarray(src)
        sh, sw = src.shape[:2]
        cn = 1 if src.ndim == 2 else src.shape[2]
        dst = np.empty((dh, dw) + src.shape[2:], np.uint8)
        rc = lib().oracle_resize_linear_u8(_p(src, ctypes.c_uint8), sw, sh, cn, _p(dst, ctypes.c_uint8), dw, dh)
    else:
        src = np.ascontiguousarray(src, np.float32)
        sh, sw = src.shape
        dst = np.empty((dh, dw), np.float32)
        rc = lib().oracle_resize_linear_f32(_p(src, ctypes.c_float), sw, sh, _p(dst, ctypes.c_float), dw, dh)
    if rc:
        raise ValueError("oracle_resize_linear: bad arguments")
    return dst


def lanczos4_itab():
    t = np.empty((1024, 64), np.int16)
    lib().oracle_lanczos4_itab(_p(t, ctypes.c_int16))
    return t


def bilinear_itab():
    t = np.empty((1024, 4), np.int16)
    lib().oracle_bilinear_itab(_p(t, ctypes.c_int16))
    return t


def disp_to_depth(disp16, valid_mask, sgbm_min_disparity, add_min_disparity, translate,
                  baseline_fx, max_depth):
    disp16 = np.ascontiguousarray(disp16, np.int16)
    h, w = disp16.shape
    mask = np.ascontiguousarray(valid_mask, np.uint8)
    disparity = np.empty((h, w), np.float32)
    depth = np.empty((h, w), np.float64)
    lib().oracle_disp_to_depth(_p(disp16, ctypes.c_int16), _p(mask, ctypes.c_uint8), w, h,
                               int(sgbm_min_disparity), int(add_min_disparity), int(bool(translate)),
                               ctypes.c_double(baseline_fx), ctypes.c_double(max_depth),
                               _p(disparity, ctypes.c_float), _p(depth, ctypes.c_double))
    return disparity, depth


def unrectify_depth(depth, M_row2, mapx, mapy):
    depth = np.ascontiguousarray(depth, np.float64)
    h, w = depth.shape
    mapx = np.ascontiguousarray(mapx, np.float32)
    mapy = np.ascontiguousarray(mapy, np.float32)
    oh, ow = mapx.shape
    M = np.ascontiguousarray(M_row2, np.float64).reshape(3)
    out = np.empty((oh, ow), np.float64)
    lib().oracle_unrectify_depth(_p(depth, ctypes.c_double), w, h, _p(M, ctypes.c_double),
                                 _p(mapx, ctypes.c_float), _p(mapy, ctypes.c_float),
                                 _p(out, ctypes.c_double), ow, oh)
    return out


def set_switches(lanczos_fix_group_lo=4, bt_border_raw_tab0=1, cost_saturate=1, way3_stripes=4, way3_simd_lanes=8):
    s = Switches(int(lanczos_fix_group_lo), int(bt_border_raw_tab0), int(cost_saturate), int(way3_stripes),
                 int(way3_simd_lanes))
    lib().oracle_set_switches(ctypes.byref(s))
