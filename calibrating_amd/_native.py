"""ctypes binding of libcalibrating_amd.so (include/calibrating_amd.h).

The library holds the hand-written gfx950 kernels; there is no CPU fallback.  ``lib()`` raises
``RuntimeError`` when the shared object is missing (run ``python -c "import __graft_entry__ as g;
g.build()"`` or ``make -C calibrating_amd/csrc``) and every compute entry point returns
``CAMD_ERR_NO_DEVICE`` -> ``RuntimeError`` when no MI355X is visible.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (bench.py --lib, or CAMD_LIB in the environment of a test / measurement run, points this at an experimental build of
# the same ABI before the first call: A/B of kernel variants built by tools/build_dbg.sh.  Still a HIP library: there
# is no CPU fallback either way)
LIB_PATH = os.environ.get("CAMD_LIB") or os.path.join(_HERE, "lib", "libcalibrating_amd.so")
_lib = None

c_void_p, c_int, c_size_t, c_double = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_double

CAMD_OK = 0
CAMD_ERR_BAD_ARG = -1
CAMD_ERR_UNSUPPORTED = -2
CAMD_ERR_NO_DEVICE = -3
CAMD_ERR_HIP = -4
CAMD_ERR_NOMEM = -5

MODE_SGBM = 0
MODE_HH = 1
MODE_SGBM_3WAY = 2  # cv2's four-stripe, three-direction variant -- UNPINNED vs cv2: stripes, warm-up overlap and the SIMD
                    # tie rule are restated from recollection; tools/export_cv2_golden.py settles them
MODE_HH4 = 3
INTER_NEAREST = 0
INTER_LINEAR = 1
INTER_LANCZOS4 = 4


class SgbmParams(ctypes.Structure):
    FIELDS = ("minDisparity", "numDisparities", "blockSize", "P1", "P2", "disp12MaxDiff",
              "preFilterCap", "uniquenessRatio", "speckleWindowSize", "speckleRange", "mode")
    _fields_ = [(n, c_int) for n in FIELDS]


# name -> (restype, argtypes); every symbol include/calibrating_amd.h declares
SIGNATURES = {
    "camd_last_error": (ctypes.c_char_p, []),
    "camd_version": (c_int, []),
    "camd_device_ok": (c_int, []),
    "camd_sgbm_workspace_bytes": (c_size_t, [ctypes.POINTER(SgbmParams), c_int, c_int, c_int, c_int]),
    "camd_sgbm_create": (c_int, [ctypes.POINTER(SgbmParams), c_int, c_int, c_int, c_int,
                                 ctypes.POINTER(c_void_p)]),
    "camd_sgbm_destroy": (c_int, [c_void_p]),
    "camd_sgbm_compute": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, c_size_t,
                                  c_size_t, c_int, c_void_p]),
    "camd_sgbm_query": (c_int, [c_void_p] + [ctypes.POINTER(c_int)] * 4),
    "camd_sgbm_debug_copy": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "camd_sgbm_set_option": (c_int, [c_void_p, c_int, c_int]),
    "camd_sgbm_status": (c_int, [c_void_p, c_void_p]),
    "camd_sgbm_set_profiling": (c_int, [c_void_p, c_int]),
    "camd_sgbm_num_stages": (c_int, []),
    "camd_sgbm_stage_name": (ctypes.c_char_p, [c_int]),
    "camd_sgbm_get_profile": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_float), c_int]),
    "camd_median3_s16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "camd_speckle_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "camd_filter_speckles_s16": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                         c_void_p]),
    "camd_remap_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_size_t, c_size_t, c_void_p, c_void_p,
                              c_void_p, c_int, c_int, c_size_t, c_size_t, c_int, c_int, c_int, c_void_p]),
    "camd_remap_fixed_bilinear_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_size_t, c_size_t, c_void_p,
                                             c_void_p, c_void_p, c_int, c_int, c_size_t, c_size_t, c_int,
                                             c_void_p]),
    "camd_undistort_maps_host": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "camd_init_undistort_rectify_map": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                                c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "camd_undistort_maps": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "camd_point_cloud_grid": (c_int, [c_int, c_int, c_double, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "camd_point_cloud_workspace_bytes": (c_size_t, [c_int, c_int, c_double]),
    "camd_depth_to_point_cloud": (c_int, [c_void_p, c_int, c_int, c_void_p, c_double, c_void_p, c_void_p, c_size_t,
                                          c_void_p, c_void_p, c_void_p]),
    "camd_apply_T_to_point_cloud": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "camd_point_cloud_to_depth": (c_int, [c_void_p, c_size_t, c_int, c_void_p, c_int, c_int, c_double, c_void_p,
                                          c_void_p, c_void_p]),
    "camd_project_depth": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_double, c_int, c_int,
                                   c_void_p, c_void_p, c_void_p]),
    "camd_set_global_option": (c_int, [c_int, c_int]),
    "camd_lanczos4_table_host": (c_int, [c_void_p]),
    "camd_bilinear_table_host": (c_int, [c_void_p]),
    "camd_resize_linear_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "camd_resize_linear_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "camd_disp_to_depth": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_double,
                                   c_double, c_void_p, c_void_p, c_int, c_void_p]),
    "camd_disp16_resized_to_depth": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                             c_double, c_double, c_void_p, c_void_p, c_int, c_void_p]),
    "camd_unrectify_depth": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_int, c_int, c_int, c_void_p]),
}

# include/calibrating_amd_experimental.h: measurement hooks without a counterpart in the reference's interface (CU-masked
# streams; with them goes set_option's CAMD_OPT_PHASES = 6).  Bound so that tools/ and one parity test can reach them;
# nothing in this package calls them.
EXPERIMENTAL_SIGNATURES = {
    "camd_stream_create_cu_mask": (c_int, [c_void_p, c_int, ctypes.POINTER(c_void_p)]),
    "camd_stream_destroy": (c_int, [c_void_p]),
}


def lib():
    """The loaded library (declares every signature on first use)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "calibrating_amd: %s is missing -- build the HIP extension first "
                "(make -C calibrating_amd/csrc). There is no CPU fallback." % LIB_PATH)
        # torch first: it bundles its own libamdhip64.so.7; loading ours afterwards binds to that same
        # HIP runtime instance, so device pointers and streams are shared with torch
        import torch  # noqa: F401
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in list(SIGNATURES.items()) + list(EXPERIMENTAL_SIGNATURES.items()):
            fn = getattr(l, name)  # AttributeError here = header and library disagree
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def last_error():
    msg = lib().camd_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc, what=""):
    """Map a camd_status to the exception the reference would raise at that point."""
    if rc == CAMD_OK:
        return
    msg = "%s%s" % (what + ": " if what else "", last_error())
    if rc in (CAMD_ERR_BAD_ARG, CAMD_ERR_UNSUPPORTED):
        raise ValueError(msg)
    if rc == CAMD_ERR_NOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def require_device():
    check(lib().camd_device_ok(), "calibrating_amd")


def current_stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
