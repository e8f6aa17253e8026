"""``StereoSGBM`` -- the object ``cv2.StereoSGBM_create`` returns, backed by the gfx950 kernels.

Mirrors the part of cv2's interface the reference touches
(/root/reference/calibrating/stereo_matching.py:48-58 ``StereoSGBM_create``, :63 ``compute``,
:64 ``getMinDisparity``) plus the remaining getters/setters of cv2.StereoSGBM.
``compute`` takes NumPy arrays (host path: one H2D + one D2H copy) or torch CUDA tensors
(zero-copy) and also accepts a leading batch dimension, which cv2 does not.
"""
import collections
import ctypes

import numpy as np

from . import _native, hostio

MODE_SGBM = _native.MODE_SGBM
MODE_HH = _native.MODE_HH
MODE_HH4 = _native.MODE_HH4
MODE_SGBM_3WAY = _native.MODE_SGBM_3WAY
STEREO_SGBM_MODE_SGBM = MODE_SGBM
STEREO_SGBM_MODE_HH = MODE_HH
STEREO_SGBM_MODE_HH4 = MODE_HH4
DISP_SHIFT = 4
DISP_SCALE = 16

_DEFAULTS = dict(minDisparity=0, numDisparities=16, blockSize=3, P1=0, P2=0, disp12MaxDiff=0,
                 preFilterCap=0, uniquenessRatio=0, speckleWindowSize=0, speckleRange=0, mode=MODE_SGBM)


class StereoSGBM:
    # A handle owns the device workspace for one (width, height, channels, device) and a maximum batch.  cv2's object
    # takes any image size from call to call (stereo_matching.py:63 feeds it whatever the resize produced), so the
    # handles of the last few shapes are kept (least recently used first out) instead of paying a multi-GB
    # hipFree + hipMalloc whenever two sizes alternate.  Bounded by a count and by a byte budget
    # (camd_sgbm_workspace_bytes; None = half of the device's memory); the handle in use is never evicted.
    HANDLE_CACHE = 4
    HANDLE_CACHE_BYTES = None
    creates = 0  # camd_sgbm_create calls of this process (tests / latency measurements read it)

    def __init__(self, **kw):
        self._p = dict(_DEFAULTS)
        for k, v in kw.items():
            if k not in self._p:
                raise TypeError("StereoSGBM_create() got an unexpected keyword argument %r" % k)
            self._p[k] = int(v)
        self._handle = None
        self._key = None
        self._max_batch = 0
        self._cache = collections.OrderedDict()  # key -> (handle, max_batch, workspace bytes)
        self.profiling = False
        self._options = {}

    def set_option(self, option, value):
        """'path': 0 = auto (default), 1 = one line-scan launch per direction, 2 = fused band-wavefront
        passes (throughput), 3 = all directions concurrently (latency, small batches);
        'keep_S': 1 = keep the aggregated volume for debug_volume('S') on the band path;
        'cost': cost-volume kernels, 0 = auto, 1 = fused k_cost (BT + KxK box sum -> C in one pass),
        2 = k_hsum + k_vsum (two passes through an intermediate volume).  All choices are bit-identical;
        'saturate': int16 overflow of the cost-volume sums, 1 = saturate like cv2's SIMD build (default),
        0 = wrap like OpenCV's scalar build (differs only for blockSize >= 7 on extreme images);
        'way3_simd_lanes': MODE_SGBM_3WAY's tie rule, 8 = cv2's SSE / NEON builds (default), 1 = scalar build;
        'exact': a pair whose cost volume left the int16 regime of the fast kernels (only after an overflow of the box
        sums on adversarial images) is 1 = aggregated again in plain int arithmetic (default), 0 = refused (written
        as invalid, status() / the next compute raise);
        'phases' / 'resident': measurement hooks (include/calibrating_amd_experimental.h; tools/ only) -- which of
        {1 cost volume, 2 first aggregation pass, 4 last pass + post} a compute() queues (default 7), and 16 a + b
        persistent workgroups per CU for the cost kernel / the row-parallel last pass (default 0 = ordinary launches).
        Every cached workspace takes the value or none does: a refusal rolls the others back before it is raised."""
        names = {"path": 0, "keep_S": 1, "cost": 2, "saturate": 3, "way3_simd_lanes": 4, "exact": 5, "phases": 6, "resident": 7}
        defaults = {0: 0, 1: 0, 2: 0, 3: 1, 4: 8, 5: 1, 6: 7, 7: 0}
        opt, value = names[option], int(value)
        lib, applied = _native.lib(), []
        for hd, _, _ in self._cache.values():
            rc = lib.camd_sgbm_set_option(hd, opt, value)
            if rc != _native.CAMD_OK:
                msg = _native.last_error()
                for done in applied:  # (the previous value was accepted once, so this cannot be refused)
                    lib.camd_sgbm_set_option(done, opt, self._options.get(opt, defaults[opt]))
                raise ValueError("StereoSGBM.set_option(%r, %d): %s" % (option, value, msg))
            applied.append(hd)
        self._options[opt] = value
        return self

    def status(self):
        """Synchronise and raise if a device-side bounded wait timed out (fused path)."""
        for hd, _, _ in list(self._cache.values()):
            _native.check(_native.lib().camd_sgbm_status(hd, _native.current_stream()), "StereoSGBM")

    # -- cv2-style accessors ---------------------------------------------------------------------
    def _get(self, k):
        return self._p[k]

    def _set(self, k, v):
        self._p[k] = int(v)
        self._release()

    def getMinDisparity(self): return self._get("minDisparity")
    def setMinDisparity(self, v): self._set("minDisparity", v)
    def getNumDisparities(self): return self._get("numDisparities")
    def setNumDisparities(self, v): self._set("numDisparities", v)
    def getBlockSize(self): return self._get("blockSize")
    def setBlockSize(self, v): self._set("blockSize", v)
    def getP1(self): return self._get("P1")
    def setP1(self, v): self._set("P1", v)
    def getP2(self): return self._get("P2")
    def setP2(self, v): self._set("P2", v)
    def getDisp12MaxDiff(self): return self._get("disp12MaxDiff")
    def setDisp12MaxDiff(self, v): self._set("disp12MaxDiff", v)
    def getPreFilterCap(self): return self._get("preFilterCap")
    def setPreFilterCap(self, v): self._set("preFilterCap", v)
    def getUniquenessRatio(self): return self._get("uniquenessRatio")
    def setUniquenessRatio(self, v): self._set("uniquenessRatio", v)
    def getSpeckleWindowSize(self): return self._get("speckleWindowSize")
    def setSpeckleWindowSize(self, v): self._set("speckleWindowSize", v)
    def getSpeckleRange(self): return self._get("speckleRange")
    def setSpeckleRange(self, v): self._set("speckleRange", v)
    def getMode(self): return self._get("mode")
    def setMode(self, v): self._set("mode", v)

    @property
    def params(self):
        return dict(self._p)

    # -- handle management -----------------------------------------------------------------------
    def _release(self):
        for hd, _, _ in self._cache.values():
            _native.lib().camd_sgbm_destroy(hd)
        self._cache.clear()
        self._handle = None
        self._key = None
        self._max_batch = 0

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _cparams(self):
        return _native.SgbmParams(**self._p)

    def _budget(self, device_index):
        if self.HANDLE_CACHE_BYTES is not None:
            return self.HANDLE_CACHE_BYTES
        import torch
        return torch.cuda.get_device_properties(device_index).total_memory // 2

    def _ensure(self, w, h, cn, batch, device_index):
        key = (w, h, cn, device_index)
        hit = self._cache.get(key)
        if hit is not None and hit[1] >= batch:
            self._cache.move_to_end(key)
            self._handle, self._key, self._max_batch = hit[0], key, hit[1]
            return
        lib = _native.lib()
        if hit is not None:  # same shape, larger batch: the smaller workspace goes first
            lib.camd_sgbm_destroy(self._cache.pop(key)[0])
            self._handle = None
        p = self._cparams()
        need = int(lib.camd_sgbm_workspace_bytes(ctypes.byref(p), w, h, cn, batch))
        # make room BEFORE allocating: least recently used first, until count and bytes fit with the new handle
        budget = self._budget(device_index)
        while self._cache and (len(self._cache) >= self.HANDLE_CACHE
                               or sum(v[2] for v in self._cache.values()) + need > budget):
            _, (old, _, _) = self._cache.popitem(last=False)
            lib.camd_sgbm_destroy(old)
        self._handle = None
        hd = ctypes.c_void_p()
        _native.check(lib.camd_sgbm_create(ctypes.byref(p), w, h, cn, batch, ctypes.byref(hd)), "StereoSGBM")
        StereoSGBM.creates += 1
        self._cache[key] = (hd, batch, need)
        self._handle, self._key, self._max_batch = hd, key, batch
        if self.profiling:
            _native.check(lib.camd_sgbm_set_profiling(hd, 1))
        for opt, val in self._options.items():
            _native.check(lib.camd_sgbm_set_option(hd, opt, val))

    def workspace_bytes(self, w, h, cn=1, batch=1):
        p = self._cparams()
        return int(_native.lib().camd_sgbm_workspace_bytes(ctypes.byref(p), w, h, cn, batch))

    # -- compute ---------------------------------------------------------------------------------
    def compute(self, left, right, out=None):
        """disparity * 16 as int16, shape (h, w) (or (n, h, w) for batched input)."""
        import torch
        is_np = isinstance(left, np.ndarray)
        if is_np != isinstance(right, np.ndarray):
            raise ValueError("left and right must both be NumPy arrays or both torch tensors")
        if is_np:
            _native.require_device()
            left_t = torch.from_numpy(np.ascontiguousarray(left)).cuda()
            right_t = torch.from_numpy(np.ascontiguousarray(right)).cuda()
        else:
            left_t, right_t = left, right
        if left_t.shape != right_t.shape or left_t.dtype != torch.uint8 or right_t.dtype != torch.uint8:
            # cv2: (-215:Assertion failed) left.size() == right.size() && left.type() == right.type() && depth == CV_8U
            raise ValueError("left and right must be uint8 images of identical shape, got %s %s and %s %s"
                             % (tuple(left_t.shape), left_t.dtype, tuple(right_t.shape), right_t.dtype))
        if not left_t.is_cuda:
            raise ValueError("tensor inputs must live on the GPU")
        nd = left_t.dim()
        # (h,w) | (h,w,c) | (n,h,w) with c not in (1,3) ... disambiguate by the last dimension
        if nd == 2:
            batched, cn = False, 1
        elif nd == 3 and left_t.shape[-1] in (1, 3):
            batched, cn = False, left_t.shape[-1]
        elif nd == 3:
            batched, cn = True, 1
        elif nd == 4 and left_t.shape[-1] in (1, 3):
            batched, cn = True, left_t.shape[-1]
        else:
            raise ValueError("unsupported image shape %s" % (tuple(left_t.shape),))
        left_t, right_t = left_t.contiguous(), right_t.contiguous()
        n = left_t.shape[0] if batched else 1
        h, w = (left_t.shape[1], left_t.shape[2]) if batched else (left_t.shape[0], left_t.shape[1])
        dev = left_t.device.index or 0
        with torch.cuda.device(dev):
            self._ensure(w, h, cn, n, dev)
            if out is None:
                out = torch.empty((n, h, w), dtype=torch.int16, device=left_t.device)
            elif out.dtype != torch.int16 or out.numel() != n * h * w or not out.is_contiguous():
                raise ValueError("out must be a contiguous int16 tensor of %d elements" % (n * h * w))
            rc = _native.lib().camd_sgbm_compute(
                self._handle, left_t.data_ptr(), right_t.data_ptr(), w * cn, h * w * cn, out.data_ptr(),
                w * 2, h * w * 2, n, _native.current_stream())
            _native.check(rc, "StereoSGBM.compute")
        if not self._options.get(6, 7) & 4:
            # (measurement hook 'phases' without the last part: nothing was written to `out`; hand back nothing rather than
            # an uninitialised buffer -- or, for ndarray input, a host copy of one)
            return None
        res = out.view(n, h, w) if batched else out.view(h, w)
        if is_np:
            res = hostio.to_host(res)
            self.status()  # the copy synchronised: surface device-side timeouts here at no extra cost
        return res

    # -- parity / measurement hooks ----------------------------------------------------------------
    def geometry(self):
        v = [ctypes.c_int() for _ in range(4)]
        _native.check(_native.lib().camd_sgbm_query(self._handle, *[ctypes.byref(x) for x in v]))
        return dict(width1=v[0].value, D=v[1].value, Dp=v[2].value, minX1=v[3].value)

    def debug_volume(self, which, index=0):
        """'C' (matching cost incl. +P2), 'S' (aggregated) as (h, width1, D) int16, or 'raw' (h, w)."""
        import torch
        if self._handle is None:
            raise RuntimeError("call compute() first")
        g = self.geometry()
        w, h = self._key[0], self._key[1]
        code = {"C": 0, "S": 1, "raw": 2}[which]
        shape = (h, w) if code == 2 else (h, max(g["width1"], 0), g["Dp"])
        buf = torch.empty(shape, dtype=torch.int16, device="cuda:%d" % self._key[3])
        _native.check(_native.lib().camd_sgbm_debug_copy(self._handle, code, index, buf.data_ptr(),
                                                         _native.current_stream()))
        return buf if code == 2 else buf[..., :g["D"]]

    def set_profiling(self, enable=True):
        self.profiling = bool(enable)
        for hd, _, _ in self._cache.values():
            _native.check(_native.lib().camd_sgbm_set_profiling(hd, int(self.profiling)))

    def stage_times_ms(self):
        """{stage: ms} of the last compute (hipEvents on the compute stream)."""
        l = _native.lib()
        n = l.camd_sgbm_num_stages()
        ms = (ctypes.c_float * n)()
        _native.check(l.camd_sgbm_get_profile(self._handle, ms, n))
        return {l.camd_sgbm_stage_name(i).decode(): float(ms[i]) for i in range(n)}


def StereoSGBM_create(minDisparity=0, numDisparities=16, blockSize=3, P1=0, P2=0, disp12MaxDiff=0,
                      preFilterCap=0, uniquenessRatio=0, speckleWindowSize=0, speckleRange=0,
                      mode=MODE_SGBM):
    """Same signature and defaults as cv2.StereoSGBM_create."""
    return StereoSGBM(minDisparity=minDisparity, numDisparities=numDisparities, blockSize=blockSize,
                      P1=P1, P2=P2, disp12MaxDiff=disp12MaxDiff, preFilterCap=preFilterCap,
                      uniquenessRatio=uniquenessRatio, speckleWindowSize=speckleWindowSize,
                      speckleRange=speckleRange, mode=mode)
