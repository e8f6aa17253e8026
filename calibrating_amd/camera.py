"""``Cam``: the per-camera record ``Stereo`` needs -- K, D, xy, name.

Only the record format of the reference's ``Cam`` is mirrored (``Cam.load`` / ``Cam.dump``,
/root/reference/calibrating/camera.py:407-448): a YAML / dict record with either ``K`` (3x3) or
``fx, fy, cx, cy``, optional ``D`` (default five zeros), ``xy`` = (width, height) (required), and the
free-form keys ``name``, ``T_in_main_cam``, ``retval``.  Intrinsic calibration itself
(cv2.calibrateCamera, boards, caches) is outside the stereo-depth hot path.
"""
import copy

import numpy as np
import yaml

_INTRINSIC_KEYS = ("fx", "fy", "cx", "cy")
_RECORD_KEYS = ("D", "xy", "name", "T_in_main_cam", "retval")  # what a dump carries besides the intrinsics


def intrinsic_format_conversion(K_or_dic):
    """K (3x3) <-> dict(fx, fy, cx, cy): the two spellings of the intrinsics in a camera record."""
    if isinstance(K_or_dic, dict):
        fx, fy, cx, cy = (K_or_dic[k] for k in _INTRINSIC_KEYS)
        return np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float64)
    K = np.asarray(K_or_dic)
    return {k: float(v) for k, v in zip(_INTRINSIC_KEYS, (K[0, 0], K[1, 1], K[0, 2], K[1, 2]))}


def read_record(source):
    """A record from a dict (deep-copied), a YAML document (a string containing a newline) or a YAML file
    path -- the three input forms ``Cam.load`` / ``Stereo.load`` of the reference accept."""
    if isinstance(source, (dict, list)):
        return copy.deepcopy(source)
    if "\n" in source:
        return yaml.safe_load(source)
    with open(source) as f:
        return yaml.safe_load(f)


def write_record(record, path=""):
    """YAML text of ``record`` stamped with the package version; also written to ``path`` when given."""
    from .__info__ import __version__
    text = yaml.safe_dump(dict(record, _calibrating_version=__version__))
    if path:
        with open(path, "w") as f:
            f.write(text)
    return text


def _plain(value):
    if isinstance(value, np.ndarray):
        return value.tolist()
    return list(value) if isinstance(value, tuple) else value


class Cam(dict):
    def __init__(self, K=None, D=None, xy=None, name=None):
        super().__init__()
        if K is not None:
            self.K = np.float64(K)
            self.D = np.zeros((1, 5)) if D is None else np.float64(D)
            self.xy = tuple(xy)
            self.name = name

    @classmethod
    def init_by_K_D(cls, K, D=None, xy=None, name=None):
        return cls(K, D, xy, name)

    def load(self, path_or_str_or_dict=None):
        """``Cam.load(record)`` (called on the class: builds a new Cam) or ``cam.load(record)`` (in place)."""
        if path_or_str_or_dict is None:  # Cam.load(x) binds x to `self`
            return Cam().load(self)
        if isinstance(path_or_str_or_dict, Cam):
            return path_or_str_or_dict.copy()
        rec = read_record(path_or_str_or_dict)
        rec.pop("_calibrating_version", None)
        if "K" in rec:
            K = rec.pop("K")
        else:
            K = intrinsic_format_conversion({k: rec.pop(k) for k in _INTRINSIC_KEYS})
        xy = tuple(rec.pop("xy", getattr(self, "xy", ())))
        assert len(xy), "Need xy"
        self.K = np.float64(K)
        self.D = np.float64(rec.pop("D")) if "D" in rec else np.zeros((1, 5))
        self.xy = xy
        self.__dict__.update(rec)
        return self

    def copy(self):
        return type(self)().load(self.dump(return_dict=True))

    def dump(self, path="", return_dict=False):
        rec = {k: _plain(self.__dict__[k]) for k in _RECORD_KEYS if k in self.__dict__}
        rec.update(intrinsic_format_conversion(self.K))
        return rec if return_dict else write_record(rec, path)

    # intrinsics by name and the fields of view in DEGREES (reference camera.py:500-530; its boxx trig works in degrees)
    fx = property(lambda self: self.K[0, 0])
    fy = property(lambda self: self.K[1, 1])
    cx = property(lambda self: self.K[0, 2])
    cy = property(lambda self: self.K[1, 2])

    @property
    def fovs(self):
        half_x, half_y = self.xy[0] / 2 / self.fx, self.xy[1] / 2 / self.fy  # tangents of the half angles
        deg = lambda t: float(np.degrees(2 * np.arctan(t)))  # noqa: E731
        return dict(fov=deg(np.hypot(half_x, half_y)), fovx=deg(half_x), fovy=deg(half_y))

    def project_cam2_depth(cam1, cam2, depth2, T=None, interpolation=1.5):
        """Depth image of ``cam2`` re-projected into this camera (camera.py:298-309), on the GPU.
        ``T`` = pose of cam2 in this camera (4x4); the reference's fallback that derives it from calibration
        board detections (``get_T_cam2_in_self``) is outside the MI355X path, so ``T`` is required."""
        if T is None:
            raise NotImplementedError("pass T (cam2 in cam1): board-based extrinsics are outside the MI355X path")
        from . import pointcloud
        rate = pointcloud.get_appropriate_interpolation_rate(cam1, cam2, interpolation)
        return pointcloud.project_depth(depth2, cam2.K, T, cam1.K, cam1.xy, interpolation_rate=rate)
