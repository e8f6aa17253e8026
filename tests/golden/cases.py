"""The golden-vector catalogue of the stereo path: ONE list of stage-wise cases, two producers, three consumers.

Producers (each writes ``tests/golden/<producer>_<stage>.npz``):
  * ``python tests/golden/make_golden.py``      producer "oracle": this repo's CPU oracle (oracle/*.c).  Always
                                                committed; freezes the oracle and feeds the GPU tests with data.
  * ``python tools/export_cv2_golden.py``       producer "cv2": the real OpenCV, on any machine where ``import cv2``
                                                works.  Its files PIN parity; they are absent
                                                from this repository only because cv2 is not installable where it
                                                was written.
Consumers: tests/test_golden_cpu.py (oracle vs both producers' files), tests/test_gpu_golden.py (the HIP path vs
both producers' files), and the producers themselves.

A case is ``dict(name, stage, inputs={...arrays / scalars...})``; ``run(backend, case)`` returns the dict of output
arrays of that stage computed by ``backend`` in {"oracle", "cv2", "gpu"}.  Fixtures are data (inputs + expected
outputs); nothing of /root/reference is read here.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from calibrating_amd import synthetic  # noqa: E402

SGBM_NAMES = ["minDisparity", "numDisparities", "blockSize", "P1", "P2", "disp12MaxDiff", "preFilterCap",
              "uniquenessRatio", "speckleWindowSize", "speckleRange", "mode"]
STAGES = ("sgbm", "remap", "maps", "undistort", "resize", "post", "rodrigues")


def _sgbm(name, seed, H, W, D, cn, **p):
    left, right = synthetic.rectified_pair(seed=seed, H=H, W=W, D=max(D, 8), cn=cn)
    full = {k: 0 for k in SGBM_NAMES}
    full.update(p)
    return dict(name=name, stage="sgbm", inputs=dict(left=left, right=right,
                                                     params=np.array([full[k] for k in SGBM_NAMES], np.int32)))


def _std(cn, D, bs, **kw):
    p = dict(minDisparity=0, numDisparities=D, blockSize=bs, P1=8 * cn * bs * bs, P2=32 * cn * bs * bs,
             disp12MaxDiff=1, uniquenessRatio=10)
    p.update(kw)
    return p


def sgbm_cases():
    c = []
    # the BASELINE configs on strips the oracle finishes in seconds
    c.append(_sgbm("c2_rgb_sgbm", 1234, 32, 1920, 128, 3, **_std(3, 128, 5, mode=0)))
    c.append(_sgbm("c2_rgb_hh", 1234, 32, 1920, 128, 3, **_std(3, 128, 5, mode=1)))
    c.append(_sgbm("c2_gray_sgbm", 1234, 32, 1920, 128, 1, **_std(1, 128, 5, mode=0)))
    c.append(_sgbm("c4_4k_d256", 7, 24, 3840, 256, 1, **_std(1, 256, 5, mode=0)))
    c.append(_sgbm("c5_vga_speckle", 9, 60, 640, 64, 3, **_std(3, 64, 5, speckleWindowSize=100, speckleRange=2)))
    # the matcher the reference hard-codes (stereo_matching.py:30-58) at its default max_size
    c.append(_sgbm("reference_default", 5, 48, 1000, 218, 3, minDisparity=2, numDisparities=218, blockSize=11,
                   uniquenessRatio=5, speckleWindowSize=200, speckleRange=2, disp12MaxDiff=0, P1=968, P2=3872))
    # small shapes / modes / parameter normalisation (SURVEY Appendix A.14 U-flags)
    c.append(_sgbm("small_gray", 100, 48, 256, 128, 1, **_std(1, 128, 5)))
    c.append(_sgbm("small_hh4", 101, 40, 200, 64, 3, **_std(3, 64, 5, mode=3)))
    c.append(_sgbm("neg_mind_hh", 104, 30, 128, 48, 1, **_std(1, 48, 7, minDisparity=-7, mode=1)))
    # MODE_SGBM_3WAY (four row stripes x three directions): U16-U20
    c.append(_sgbm("way3_rgb", 110, 96, 260, 64, 3, **_std(3, 64, 5, mode=2)))
    c.append(_sgbm("way3_gray_d50_block3", 111, 80, 200, 50, 1, **_std(1, 50, 3, mode=2)))
    c.append(_sgbm("way3_default_block", 112, 64, 160, 16, 1, numDisparities=16, mode=2))
    c.append(_sgbm("u1_disp12_zero", 105, 30, 160, 32, 1, **_std(1, 32, 5, disp12MaxDiff=0)))
    c.append(_sgbm("u1_disp12_negative", 105, 30, 160, 32, 1, **_std(1, 32, 5, disp12MaxDiff=-1)))
    c.append(_sgbm("u2_defaults_all_zero", 106, 30, 160, 16, 1, numDisparities=16))
    c.append(_sgbm("u6_d_not_multiple_of_16", 107, 30, 200, 50, 3, **_std(3, 50, 3)))
    c.append(_sgbm("u10_mind_positive", 108, 36, 220, 64, 3, **_std(3, 64, 5, minDisparity=3)))
    c.append(_sgbm("u4_prefiltercap_63", 109, 30, 200, 64, 1, **_std(1, 64, 5, preFilterCap=63)))
    # U7: block 11 x RGB on images that push the window sums towards int16 overflow
    H, W, D = 40, 300, 96
    p11 = dict(minDisparity=0, numDisparities=D, blockSize=11, P1=968, P2=3872, disp12MaxDiff=1, uniquenessRatio=5)
    bw_l = np.zeros((H, W, 3), np.uint8); bw_l[:, W // 2:] = 255
    bw_r = 255 - bw_l
    chk = ((np.add.outer(np.arange(H), np.arange(W)) & 1) * 255).astype(np.uint8)[..., None].repeat(3, 2)
    rng = np.random.default_rng(77)
    noise_l = rng.integers(0, 2, (H, W, 3), dtype=np.uint8) * 255
    noise_r = rng.integers(0, 2, (H, W, 3), dtype=np.uint8) * 255
    for nm, l, r, extra in (("u7_black_white", bw_l, bw_r, {}), ("u7_checker", chk, 255 - chk, {}),
                            ("u7_binary_noise_cap63", noise_l, noise_r, dict(preFilterCap=63))):
        full = {k: 0 for k in SGBM_NAMES}
        full.update(p11)
        full.update(extra)
        c.append(dict(name=nm, stage="sgbm", inputs=dict(left=l, right=r,
                                                         params=np.array([full[k] for k in SGBM_NAMES], np.int32))))
    # U20: every total ties (constant 15 = what the BT border columns carry): the winner shows the tie rule of the build
    flat = np.full((72, 220), 15, np.uint8)
    full = {k: 0 for k in SGBM_NAMES}
    full.update(dict(numDisparities=64, blockSize=5, P1=200, P2=800, disp12MaxDiff=1, mode=2))
    c.append(dict(name="way3_u20_all_ties", stage="sgbm",
                  inputs=dict(left=flat, right=flat, params=np.array([full[k] for k in SGBM_NAMES], np.int32))))
    # U11: texture only in the two border columns
    bl = np.full((24, 120, 1), 90, np.uint8); br = bl.copy()
    bl[:, 0] = 255; bl[:, -1] = 0; br[:, 0] = 0; br[:, -1] = 255
    full = {k: 0 for k in SGBM_NAMES}
    full.update(_std(1, 32, 3))
    c.append(dict(name="u11_border_columns", stage="sgbm",
                  inputs=dict(left=bl[..., 0], right=br[..., 0], params=np.array([full[k] for k in SGBM_NAMES], np.int32))))
    return c


def remap_cases():
    rng = np.random.default_rng(42)
    src = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    yy, xx = np.mgrid[:40, :56].astype(np.float32)
    mapx = (xx * 1.1 + rng.uniform(-4, 4, xx.shape)).astype(np.float32)
    mapy = (yy * 1.15 + rng.uniform(-4, 4, yy.shape)).astype(np.float32)
    c = [dict(name="random_lanczos4", stage="remap", inputs=dict(src=src, mapx=mapx, mapy=mapy, interp=np.int32(4))),
         dict(name="random_linear", stage="remap", inputs=dict(src=src, mapx=mapx, mapy=mapy, interp=np.int32(1)))]
    # U8 / U15: a delta image sampled at every one of the 32 x 32 sub-pixel phases and all 8 x 8 tap offsets reads the
    # whole fixed-point Lanczos table back (value = weight of one tap, scaled by 255 / 32768 and rounded)
    delta = np.zeros((24, 24), np.uint8)
    delta[12, 12] = 255
    ph = np.arange(32, dtype=np.float32) / 32
    offs = np.arange(-4, 4, dtype=np.float32)
    mx = (12 + offs[None, :, None, None] + ph[None, None, None, :]) + np.zeros((8, 8, 32, 32), np.float32)
    my = (12 + offs[:, None, None, None] + ph[None, None, :, None]) + np.zeros((8, 8, 32, 32), np.float32)
    # output image (8*32, 8*32): block (oy, ox), phase (fy, fx)
    mapx_d = mx.transpose(0, 2, 1, 3).reshape(256, 256).astype(np.float32)
    mapy_d = my.transpose(0, 2, 1, 3).reshape(256, 256).astype(np.float32)
    c.append(dict(name="delta_all_phases_lanczos4", stage="remap",
                  inputs=dict(src=delta, mapx=mapx_d, mapy=mapy_d, interp=np.int32(4))))
    c.append(dict(name="delta_all_phases_linear", stage="remap",
                  inputs=dict(src=delta, mapx=mapx_d, mapy=mapy_d, interp=np.int32(1))))
    # U13: nearest-neighbour rounding at exact .5 coordinates (float64 depth image, utils.py:199)
    d64 = rng.random((20, 30)) * 5
    hx = (np.mgrid[:20, :30][1] * 0.5 + 0.5).astype(np.float32)
    hy = (np.mgrid[:20, :30][0] * 0.5 + 0.5).astype(np.float32)
    c.append(dict(name="nearest_half_coordinates_f64", stage="remap", inputs=dict(src=d64, mapx=hx, mapy=hy, interp=np.int32(0))))
    return c


def _rig_arrays(W, H):
    r = synthetic.rig(W, H)
    return (np.array(r["cam1"]["K"], np.float64), np.array(r["cam1"]["D"], np.float64).reshape(-1),
            np.array(r["cam2"]["K"], np.float64), np.array(r["cam2"]["D"], np.float64).reshape(-1),
            np.array(r["R"], np.float64), np.array(r["t"], np.float64))


def maps_cases():
    K1, D1, K2, D2, R, t = _rig_arrays(160, 120)
    Kn = K1.copy(); Kn[:2, :2] *= 0.9
    c = [dict(name="cam1_distorted_rotated", stage="maps", inputs=dict(A=K1, dist=D1, R=R, Anew=Kn, size=np.array([160, 120]))),
         dict(name="cam2_distorted_rotated", stage="maps", inputs=dict(A=K2, dist=D2, R=R.T, Anew=Kn, size=np.array([150, 100]))),
         dict(name="no_distortion_unrectify", stage="maps", inputs=dict(A=Kn, dist=np.zeros(5), R=R.T, Anew=K1, size=np.array([160, 120])))]
    return c


def undistort_cases():
    K1, D1 = _rig_arrays(160, 120)[:2]
    src = np.random.default_rng(3).integers(0, 256, (120, 160, 3), dtype=np.uint8)
    return [dict(name="rig_cam1", stage="undistort", inputs=dict(src=src, K=K1, dist=D1))]


def resize_cases():
    rng = np.random.default_rng(4)
    src = rng.integers(0, 256, (90, 120, 3), dtype=np.uint8)
    f = (rng.random((54, 100)) * 100).astype(np.float32)
    c = []
    for hw in ((45, 60), (75, 100), (180, 240), (67, 91)):
        c.append(dict(name="u8_to_%dx%d" % hw, stage="resize", inputs=dict(src=src, dsize_hw=np.array(hw))))
    for hw in ((108, 200), (135, 250)):
        c.append(dict(name="f32_to_%dx%d" % hw, stage="resize", inputs=dict(src=f, dsize_hw=np.array(hw))))
    return c


def post_cases():
    rng = np.random.default_rng(0)
    img = (rng.integers(0, 40, (70, 90)) * 16).astype(np.int16)
    img[rng.random(img.shape) < 0.1] = -16
    return [dict(name="median3_and_speckles", stage="post",
                 inputs=dict(img=img, new_val=np.int32(-16), max_size=np.int32(30), max_diff=np.int32(32)))]


def rodrigues_cases():
    return [dict(name="vectors", stage="rodrigues",
                 inputs=dict(r=np.array([[0.01, -0.02, 0.005], [0.3, 0.2, -0.7], [0, 0, 0], [3.1, 0.01, 0.0]], np.float64)))]


def all_cases():
    return (sgbm_cases() + remap_cases() + maps_cases() + undistort_cases() + resize_cases() + post_cases()
            + rodrigues_cases())


# ---------------------------------------------------------------------------------------------------------------
def run(backend, case):
    """Outputs of ``case`` computed by ``backend`` ("oracle" | "cv2" | "gpu") as a dict of NumPy arrays."""
    st, x = case["stage"], case["inputs"]
    if backend == "cv2":
        import cv2
    elif backend == "oracle":
        import oracle
        oracle.build()
    else:
        import torch  # noqa: F401
        import calibrating_amd as ca
        from calibrating_amd import imgproc, resize as _resize
    if st == "sgbm":
        p = {k: int(v) for k, v in zip(SGBM_NAMES, x["params"])}
        if backend == "cv2":
            return dict(disp=cv2.StereoSGBM_create(**p).compute(x["left"], x["right"]))
        if backend == "oracle":
            return dict(disp=oracle.sgbm_compute(x["left"], x["right"], **p))
        return dict(disp=ca.StereoSGBM_create(**p).compute(x["left"], x["right"]))
    if st == "remap":
        interp = int(x["interp"])
        if x["src"].dtype == np.float64:
            if backend == "cv2":
                return dict(dst=cv2.remap(x["src"], x["mapx"], x["mapy"], cv2.INTER_NEAREST))
            if backend == "oracle":
                return dict(dst=oracle.remap_nearest_f64(x["src"], x["mapx"], x["mapy"]))
            # the GPU applies this remap inside unrectify_depth (depth * 1 with M = (0, 0, 1))
            return dict(dst=imgproc.unrectify_depth(x["src"], np.array([0.0, 0.0, 1.0]), x["mapx"], x["mapy"]))
        if backend == "cv2":
            return dict(dst=cv2.remap(x["src"], x["mapx"], x["mapy"], interp))
        if backend == "oracle":
            return dict(dst=oracle.remap_u8(x["src"], x["mapx"], x["mapy"], interp))
        return dict(dst=imgproc.remap(x["src"], x["mapx"], x["mapy"], interp))
    if st == "maps":
        size = (int(x["size"][0]), int(x["size"][1]))
        if backend == "cv2":
            mx, my = cv2.initUndistortRectifyMap(x["A"], x["dist"], x["R"], x["Anew"], size, cv2.CV_32FC1)
        elif backend == "oracle":
            mx, my = oracle.init_undistort_rectify_map(x["A"], x["dist"], x["R"], x["Anew"], size)
        else:
            mx, my = [m.cpu().numpy() for m in imgproc.init_undistort_rectify_map(x["A"], x["dist"], x["R"], x["Anew"], size)]
        return dict(mapx=mx, mapy=my)
    if st == "undistort":
        if backend == "cv2":
            return dict(dst=cv2.undistort(x["src"], x["K"], x["dist"]))
        if backend == "oracle":
            return dict(dst=oracle.undistort_u8(x["src"], x["K"], x["dist"]))
        h, w = x["src"].shape[:2]
        mxy, ma = imgproc.undistort_maps_device(x["K"], x["dist"], (w, h))
        return dict(dst=imgproc.remap_fixed_bilinear(x["src"], mxy, ma))
    if st == "resize":
        hw = (int(x["dsize_hw"][0]), int(x["dsize_hw"][1]))
        if backend == "cv2":
            return dict(dst=cv2.resize(x["src"], hw[::-1], interpolation=cv2.INTER_LINEAR))
        if backend == "oracle":
            return dict(dst=oracle.resize_linear(x["src"], hw))
        return dict(dst=_resize.resize(torch.from_numpy(np.ascontiguousarray(x["src"])).cuda(), hw))
    if st == "post":
        nv, ms, md = int(x["new_val"]), int(x["max_size"]), int(x["max_diff"])
        if backend == "cv2":
            sp = x["img"].copy()
            cv2.filterSpeckles(sp, nv, ms, md)
            return dict(median=cv2.medianBlur(x["img"], 3), speckles=sp)
        if backend == "oracle":
            return dict(median=oracle.median3_s16(x["img"]), speckles=oracle.filter_speckles_s16(x["img"], nv, ms, md))
        return dict(median=imgproc.medianBlur3_s16(x["img"]), speckles=imgproc.filterSpeckles(x["img"], nv, ms, md))
    if st == "rodrigues":
        if backend == "cv2":
            return dict(R=np.stack([cv2.Rodrigues(r)[0] for r in x["r"]]))
        from calibrating_amd import geometry  # host-side product code on every other backend
        return dict(R=np.stack([geometry.rodrigues(r) for r in x["r"]]))
    raise ValueError(st)


def _np(v):
    return v.cpu().numpy() if hasattr(v, "cpu") else np.asarray(v)


def write(producer, backend=None, stages=STAGES):
    """Run every case on ``backend`` (default: ``producer``) and write tests/golden/<producer>_<stage>.npz."""
    backend = backend or producer
    written = []
    for st in stages:
        out = {}
        names = []
        for case in all_cases():
            if case["stage"] != st:
                continue
            names.append(case["name"])
            for k, v in case["inputs"].items():
                out["%s/in/%s" % (case["name"], k)] = np.asarray(v)
            for k, v in run(backend, case).items():
                out["%s/out/%s" % (case["name"], k)] = _np(v)
        out["names"] = np.array(names)
        path = os.path.join(HERE, "%s_%s.npz" % (producer, st))
        np.savez_compressed(path, **out)
        written.append(path)
    return written


def load(producer, stage):
    """[(name, inputs, outputs)] of tests/golden/<producer>_<stage>.npz, or None when the file is absent."""
    path = os.path.join(HERE, "%s_%s.npz" % (producer, stage))
    if not os.path.exists(path):
        return None
    z = np.load(path)
    res = []
    for name in z["names"]:
        name = str(name)
        ins = {k.split("/", 2)[2]: z[k] for k in z.files if k.startswith(name + "/in/")}
        outs = {k.split("/", 2)[2]: z[k] for k in z.files if k.startswith(name + "/out/")}
        res.append((name, ins, outs))
    return res


# tolerance of a stage against a cv2-made file, as SURVEY.md section 8(d) states it: SGBM / post filters / tables are
# bit-exact, remaps and resizes of u8 may differ by 1 LSB (table rounding U8/U15), float maps / resizes by float eps
# What a cv2 fixture of each stage decides, and which switch moves the oracle AND the HIP path together if it disagrees
# (oracle.set_switches / oracle_set_switches on the CPU side, camd_sgbm_set_option /
# camd_set_global_option on the GPU side).  Quoted by the skip and failure messages of the golden tests and by
# tools/export_cv2_golden.py, so that whoever first runs them with a real cv2 knows where to look.
WHAT_IT_PINS = {
    "sgbm": "BASELINE metric max |disparity - cv2.SGBM|; decides U1-U6, U10-U12, U14 (no switch: restated semantics), "
            "U7 (oracle cost_saturate / CAMD_OPT_SATURATE), U11 (oracle bt_border_raw_tab0), U16-U20 MODE_SGBM_3WAY "
            "(oracle way3_stripes, way3_simd_lanes / CAMD_OPT_3WAY_SIMD_LANES)",
    "remap": "cv2.remap INTER_LANCZOS4 / INTER_NEAREST; decides U8 / U15 (oracle lanczos_fix_group_lo / "
             "CAMD_GOPT_LANCZOS_FIX_GROUP_LO) and U13 (round-half-to-even of the nearest remap)",
    "maps": "cv2.initUndistortRectifyMap (float64 inside, CV_32FC1 out)",
    "undistort": "cv2.undistort = fixed-point bilinear remap through its internal CV_16SC2 maps",
    "resize": "cv2.resize INTER_LINEAR as boxx.resize calls it; decides U9",
    "post": "cv2.medianBlur(3) on int16 and cv2.filterSpeckles; decides U5",
    "rodrigues": "cv2.Rodrigues (vector -> matrix)",
}


def tolerance(stage, key, dtype):
    if stage in ("sgbm", "post"):
        return 0
    if stage in ("remap", "undistort", "resize") and np.issubdtype(dtype, np.integer):
        return 1
    if stage == "remap":
        return 0
    if stage == "maps":
        return 1e-3
    if stage == "rodrigues":
        return 1e-12
    return 1e-4
