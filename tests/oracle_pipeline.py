"""The reference's ``Stereo.get_depth`` composed from CPU oracle stages -- the checker of the end-to-end GPU tests and
of the pipeline fuzzer (test infrastructure; nothing of the product path is called here except the rig's host-side
3x3 geometry: R1, R2, K, xy, baseline).

Follows /root/reference/calibrating/stereo_camera.py:492-533 (get_depth), :216-242 (rectify), :408-428
(disparity_to_depth, unrectify_depth) and /root/reference/calibrating/stereo_matching.py:60-70 (the matcher's
``max_size`` downsizing: resize -> SGBM -> clip / threshold -> /16 -> resize back -> ``* w / sw``)."""
import numpy as np

SGBM_DEFAULTS = dict(minDisparity=2, numDisparities=218, blockSize=11, uniquenessRatio=5, speckleWindowSize=200,
                     speckleRange=2, disp12MaxDiff=0, P1=8 * 121, P2=32 * 121)  # stereo_matching.py:30-58


def sgbm_params(cfg):
    """StereoSGBM parameters the reference's plugin builds from ``cfg`` (stereo_matching.py:28-58)."""
    cfg = dict(cfg or {})
    cfg.pop("max_size", None)
    bs = int(cfg.get("blockSize", 11))
    p = dict(SGBM_DEFAULTS, P1=8 * bs * bs, P2=32 * bs * bs)
    p.update(cfg)
    return p


def rectified_pair(oracle, stereo, img1, img2):
    """stereo_camera.py:216-242: Lanczos-4 remap of each camera through ITS OWN maps (cam2 may have another size),
    then the right image translated by min_disparity with zero fill."""
    m1 = oracle.init_undistort_rectify_map(stereo.cam1.K, stereo.cam1.D, stereo.R1, stereo.K, stereo.xy)
    m2 = oracle.init_undistort_rectify_map(stereo.cam2.K, stereo.cam2.D, stereo.R2, stereo.K, stereo.xy)
    r1 = oracle.remap_u8(img1, *m1, oracle.INTER_LANCZOS4)
    r2 = oracle.remap_u8(img2, *m2, oracle.INTER_LANCZOS4)
    shift = stereo.min_disparity if stereo.translation_rectify_img else 0
    if shift > 0:
        r2[:, shift:] = r2[:, :-shift].copy()
        r2[:, :shift] = 0
    w1, h1 = stereo.cam1.xy
    mask = (-0.5 < m1[0]) & (m1[0] < w1 - 0.5) & (-0.5 < m1[1]) & (m1[1] < h1 - 0.5)  # :178-183
    return r1, r2, mask


def matcher_disparity(oracle, cfg, r1, r2):
    """``SemiGlobalBlockMatching(cfg)(r1, r2)``: float32 disparity at the input resolution (stereo_matching.py:60-70)."""
    p = sgbm_params(cfg)
    max_size = (cfg or {}).get("max_size", 1000)
    h, w = r1.shape[:2]
    ratio = min(max_size / max(h, w), 1)
    hw = (h, w) if ratio == 1 else (int(round(h * ratio)), int(round(w * ratio)))  # boxx.resize
    s1, s2 = (r1, r2) if hw == (h, w) else (oracle.resize_linear(r1, hw), oracle.resize_linear(r2, hw))
    sd = oracle.sgbm_compute(s1, s2, **p).astype(np.float32).clip(0)
    sd[sd < p["minDisparity"] * 16] = 0
    sd = sd / np.float32(16.0)
    if hw != (h, w):
        sd = oracle.resize_linear(sd, (h, w))
    return sd * w / hw[1]


def oracle_get_depth(oracle, stereo, cfg, img1, img2, plugin=None, return_unrectify_depth=True):
    """Every entry of get_depth's result dict for the SGBM plugin built from ``cfg`` (incl. ``max_size``), or for a
    foreign ``plugin`` (a callable on the rectified ndarrays returning a disparity array or a dict, :506-509)."""
    r1, r2, mask = rectified_pair(oracle, stereo, img1, img2)
    extra = {}
    if plugin is None:
        disparity = matcher_disparity(oracle, cfg, r1, r2)
    else:
        disparity = plugin(r1, r2)
        if isinstance(disparity, dict):
            extra = {k: v for k, v in disparity.items() if k != "disparity"}
            disparity = disparity["disparity"]
    if stereo.translation_rectify_img:
        disparity += stereo.min_disparity
    disparity = mask * disparity
    with np.errstate(divide="ignore"):
        depth = 1.0 * stereo.baseline * stereo.K[0, 0] / disparity  # float64 scalar / float32 array -> float64
    depth[depth > stereo.get_max_depth()] = 0
    depth[depth < 0] = 0
    result = dict(extra, rectify_img1=r1, rectify_img2=r2, disparity=disparity, rectify_depth=depth)
    if return_unrectify_depth:
        maps = oracle.init_undistort_rectify_map(stereo.K, None, stereo.R1.T, stereo.cam1.K, stereo.cam1.xy)
        M = stereo.R1.T @ np.linalg.inv(stereo.K)
        result.update(unrectify_depth=oracle.unrectify_depth(depth, M[2], *maps),
                      undistort_img1=oracle.undistort_u8(img1, stereo.cam1.K, stereo.cam1.D))
    return result


def compare(got, ref, depth_tol=1e-4, keys=None):
    """Names of the result entries that differ: images / disparity bit for bit, depths within ``depth_tol`` metres with
    identical zero (= invalid) sets; second value: the depth entries that are inside the tolerance but not identical."""
    bad, inexact = [], []
    for k in (keys or ref):
        g, r = np.asarray(got[k]), ref[k]
        if g.shape != r.shape or g.dtype != r.dtype:
            bad.append(k + ":shape/dtype")
        elif k.endswith("depth"):
            if not np.array_equal(g == 0, r == 0) or np.abs(g - r).max() > depth_tol:
                bad.append(k)
            elif not np.array_equal(g, r):
                inexact.append(k)
        elif not np.array_equal(g, r):
            bad.append(k)
    return bad, inexact
