"""Catalogue of the "reference plumbing" fixtures: what tests/golden/make_reference_golden.py feeds to the REFERENCE's own,
unmodified Python (``calibrating.Stereo.load -> set_stereo_matching -> get_depth`` and the depth post-ops of
``calibrating.utils`` / ``Cam.project_cam2_depth``) and what the consumers (tests/test_reference_plumbing_cpu.py,
tests/test_gpu_reference_plumbing.py) feed to ``calibrating_amd``.

Only DATA lives here: rig records, constructor / set_stereo_matching arguments, the seeds of the rendered input
scenes and two toy plugins written for this catalogue (a foreign plugin has to exist on both sides of the comparison).
Nothing of /root/reference is read or restated in this file; the inputs are regenerated from
``calibrating_amd.synthetic`` (deterministic NumPy) and checked against the hashes the fixture file carries.

What each case is there for (SURVEY.md section 3.6 quirks in brackets):
  c1_default_720p          BASELINE configs[0] as far as it can be driven offline: 1280x720 rig, the reference's
                           default plugin ``SemiGlobalBlockMatching({})`` (max_size 1000 -> the downsizing branch),
                           ``max_depth=3.5`` -- the call sequence of example/checkboard_example.py:41-57
  ktarget_scalar           K_target = 1.15 (scales fx, fy; better_cx_cy; [Q1] cam1.K vs self.K)
  ktarget_matrix           K_target = a 3x3 ndarray (used as is, no re-centring), xy_target tuple
  xytarget_scalar          xy_target = 0.9 (rounded sides), K_target 0.95 ([Q4] mask on source size)
  xytarget_tuple           xy_target = (512, 320)
  translate_default        max_depth = 3.0 -> translation_rectify_img defaults to True ([Q2] [Q3] [Q5] [Q6])
  translate_off            max_depth = 3.0, translation_rectify_img=False
  translate_on_no_depth    max_depth None, translation_rectify_img=True (min_disparity = int(fx b / 1000) = 0)
  record_r / record_T      the rig given as a Rodrigues vector / as a 4x4 pose (stereo_camera.py:287-292)
  hetero                   second camera of another resolution and intrinsics
  max_size_300             cfg max_size below the image (non-default downsizing ratio)
  foreign_dict             a foreign plugin returning dict(disparity=float32, confidence=...) (stereo_camera.py:506-509)
  foreign_f64              a foreign plugin returning a float64 ndarray ([Q5] in-place +=, dtype propagation)
  no_unrectify             get_depth(..., return_unrectify_depth=False)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from calibrating_amd import synthetic  # noqa: E402

FIXTURE = os.path.join(HERE, "reference_plumbing.npz")
RESULT_KEYS = ("rectify_img1", "rectify_img2", "disparity", "rectify_depth", "unrectify_depth", "undistort_img1")
SAMPLE = 4  # fixtures keep every SAMPLE-th row and column of an output (and the SHA-256 of the whole array)


MAP_SAMPLE = 16  # the (smooth) remap tables: every 16th row and column


def sample(a, step=SAMPLE):
    return np.ascontiguousarray(a[::step, ::step])


def sha(a):
    import hashlib
    a = np.ascontiguousarray(a)
    return hashlib.sha256(str((a.dtype.str, a.shape)).encode() + a.tobytes()).hexdigest()


def _hetero_rig():
    (w1, h1), (w2, h2) = (448, 336), (560, 420)
    K1 = [[0.8 * w1, 0, w1 / 2 + 3.3], [0, 0.81 * w1, h1 / 2 - 2.1], [0, 0, 1]]
    K2 = [[0.78 * w2, 0, w2 / 2 - 5.2], [0, 0.79 * w2, h2 / 2 + 4.4], [0, 0, 1]]
    return dict(R=synthetic.rodrigues([0.012, -0.018, 0.006]).tolist(), t=[[-0.12], [0.002], [-0.001]],
                cam1=dict(K=K1, D=[[-0.12, 0.05, 1e-3, -5e-4, 0.01]], xy=[w1, h1], name="wide"),
                cam2=dict(K=K2, D=[[0.08, -0.03, -8e-4, 6e-4, 0.002]], xy=[w2, h2], name="fine"))


_K_MATRIX = np.array([[372.0, 0, 247.5], [0, 369.0, 170.25], [0, 0, 1]])

# name, rig size (None: the two-resolution rig), Stereo kwargs, plugin, set_stereo_matching kwargs, scene, extras
CASES = [
    dict(name="c1_default_720p", wh=(1280, 720), plugin=("sgbm", {}), setm=dict(max_depth=3.5),
         scene=((0.3, 0.1, 1.0), 2.0, 0)),
    dict(name="ktarget_scalar", wh=(448, 336), stereo=dict(K_target=1.15), plugin=("sgbm", {}), setm={},
         scene=((0.2, 0.1, 1.0), 1.8, 1)),
    dict(name="ktarget_matrix", wh=(448, 336), stereo=dict(K_target=_K_MATRIX, xy_target=(480, 352)),
         plugin=("sgbm", {}), setm=dict(max_depth=4.0), scene=((-0.2, 0.05, 1.0), 1.7, 2)),
    dict(name="xytarget_scalar", wh=(448, 336), stereo=dict(xy_target=0.9, K_target=0.95), plugin=("sgbm", {}),
         setm=dict(max_depth=3.5), scene=((0.1, -0.15, 1.0), 1.9, 3)),
    dict(name="xytarget_tuple", wh=(448, 336), stereo=dict(xy_target=(512, 320)), plugin=("sgbm", {}), setm={},
         scene=((0.0, 0.0, 1.0), 1.6, 4)),
    dict(name="translate_default", wh=(448, 336), plugin=("sgbm", {}), setm=dict(max_depth=3.0),
         scene=((0.25, -0.1, 1.0), 2.2, 5)),
    dict(name="translate_off", wh=(448, 336), plugin=("sgbm", {}),
         setm=dict(max_depth=3.0, translation_rectify_img=False), scene=((0.25, -0.1, 1.0), 2.2, 5)),
    dict(name="translate_on_no_depth", wh=(448, 336), plugin=("sgbm", {}), setm=dict(translation_rectify_img=True),
         scene=((0.15, 0.2, 1.0), 1.5, 6)),
    dict(name="record_r", wh=(448, 336), record="r", plugin=("sgbm", {}), setm=dict(max_depth=3.5),
         scene=((0.3, 0.0, 1.0), 2.0, 7)),
    dict(name="record_T", wh=(448, 336), record="T", plugin=("sgbm", {}), setm=dict(max_depth=3.5),
         scene=((0.3, 0.0, 1.0), 2.0, 7)),
    dict(name="hetero", wh=None, plugin=("sgbm", {}), setm=dict(max_depth=3.5), scene=((0.2, 0.1, 1.0), 2.0, 8)),
    dict(name="max_size_300", wh=(448, 336), plugin=("sgbm", dict(max_size=300)), setm=dict(max_depth=3.5),
         scene=((-0.1, 0.1, 1.0), 1.2, 9)),
    dict(name="foreign_dict", wh=(320, 240), plugin=("foreign_dict", None), setm=dict(max_depth=3.0),
         scene=((0.2, 0.1, 1.0), 1.8, 10)),
    dict(name="foreign_f64", wh=(320, 240), plugin=("foreign_f64", None), setm=dict(max_depth=2.5),
         scene=((0.2, 0.1, 1.0), 1.8, 10)),
    dict(name="no_unrectify", wh=(448, 336), plugin=("sgbm", {}), setm=dict(max_depth=3.5),
         scene=((0.3, 0.1, 1.0), 2.0, 11), call=dict(return_unrectify_depth=False)),
]
CASE_BY_NAME = {c["name"]: c for c in CASES}


def rig_record(case):
    """The rig as plain lists, in the spelling ``Cam.dump`` writes (fx, fy, cx, cy) -- the one both packages load."""
    rec = _hetero_rig() if case["wh"] is None else synthetic.rig(*case["wh"])
    for cam in (rec["cam1"], rec["cam2"]):
        K = cam.pop("K")
        cam.update(fx=K[0][0], fy=K[1][1], cx=K[0][2], cy=K[1][2])
    return rec


def render_record(case):
    """The same rig with K matrices, as ``synthetic.render_plane_pair`` wants it."""
    return _hetero_rig() if case["wh"] is None else synthetic.rig(*case["wh"])


def images(case):
    normal, distance, seed = case["scene"]
    img1, img2, _ = synthetic.render_plane_pair(render_record(case), normal, distance, seed=seed)
    return img1, img2


def foreign_disparity(img1, img2):
    """What the two toy plugins compute: a deterministic float32 'disparity' from the rectified pair (content is
    irrelevant; the plumbing after the plugin is what the case pins).  Exact in float32: small integers / 8."""
    a = img1[..., 0].astype(np.int32) + img1[..., 2]
    b = img2[..., 1].astype(np.int32)
    return ((a - b) % 160).astype(np.float32) / np.float32(8.0) + np.float32(12.0)


def make_plugin(kind, cfg, base, sgbm_cls):
    """The plugin of a case for either package: ``base`` = that package's MetaStereoMatching, ``sgbm_cls`` = its
    SemiGlobalBlockMatching."""
    if kind == "sgbm":
        return sgbm_cls(dict(cfg))

    class ForeignDict(base):
        def __call__(self, img1, img2):
            d = foreign_disparity(np.asarray(img1), np.asarray(img2))
            return dict(disparity=d, confidence=(d > 20).astype(np.uint8), note="toy")

    class ForeignF64(base):
        def __call__(self, img1, img2):
            return foreign_disparity(np.asarray(img1), np.asarray(img2)).astype(np.float64)

    return {"foreign_dict": ForeignDict, "foreign_f64": ForeignF64}[kind]({})


# ---- the depth post-ops -----------------------------------------------------------------
POST_K1 = np.array([[420.0, 0, 161.3], [0, 424.0, 118.9], [0, 0, 1]])
POST_K2 = np.array([[380.0, 0, 150.0], [0, 380.0, 110.0], [0, 0, 1]])
POST_XY1, POST_XY2 = (320, 240), (300, 220)
POST_RATES = (1, 1.5, 2, 0.75, 1.37)
POST_INTERPOLATIONS = (1.5, 1, 0)
CLOUD_ROWS = 64  # fixtures keep every CLOUD_ROWS-th point of a cloud
POST_SAMPLE = 2  # ... and every second row and column of a post-op depth image


def post_depth(seed, h, w, holes=0.2):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[:h, :w]
    z = 1.5 + 0.5 * np.sin(xx / 37.0) * np.cos(yy / 23.0) + 0.3 * (xx > w // 2)
    z[rng.random((h, w)) < holes] = 0
    return z


def post_T():
    T = np.eye(4)
    T[:3, :3] = synthetic.rodrigues([0.02, -0.05, 0.01])
    T[:3, 3] = [0.06, -0.01, 0.02]
    return T


def load_fixture():
    if not os.path.exists(FIXTURE):
        return None
    with np.load(FIXTURE, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}
