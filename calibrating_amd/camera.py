"""``Cam``: the per-camera record ``Stereo`` needs -- K, D, xy, name.

Only the ``Cam.load`` path of the reference (/root/reference/calibrating/camera.py:424-448) is
mirrored: intrinsic calibration itself (cv2.calibrateCamera, boards, caches) is outside the
stereo-depth hot path.
"""
import copy

import numpy as np
import yaml


def intrinsic_format_conversion(K_or_dic):
    """K (3x3) <-> dict(fx, fy, cx, cy)  (camera.py dump/load format)."""
    if isinstance(K_or_dic, dict):
        d = K_or_dic
        return np.array([[d["fx"], 0, d["cx"]], [0, d["fy"], d["cy"]], [0, 0, 1]], np.float64)
    K = np.asarray(K_or_dic)
    return dict(fx=float(K[0, 0]), fy=float(K[1, 1]), cx=float(K[0, 2]), cy=float(K[1, 2]))


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
        if path_or_str_or_dict is None:
            path_or_str_or_dict = self
            self = Cam()
        if isinstance(path_or_str_or_dict, Cam):
            return path_or_str_or_dict.copy()
        if not isinstance(path_or_str_or_dict, (list, dict)):
            path_or_str = path_or_str_or_dict
            if "\n" in path_or_str:
                dic = yaml.safe_load(path_or_str)
            else:
                with open(path_or_str) as f:
                    dic = yaml.safe_load(f)
        else:
            dic = copy.deepcopy(path_or_str_or_dict)
        if "K" not in dic:
            dic["K"] = intrinsic_format_conversion(dic)
            [dic.pop(k) for k in ("fx", "fy", "cx", "cy")]
        dic["K"] = np.float64(dic["K"])
        dic["D"] = np.float64(dic["D"]) if "D" in dic else np.zeros((1, 5))
        dic["xy"] = tuple(dic.get("xy", getattr(self, "xy", "")))
        assert len(dic["xy"]), "Need xy"
        dic.pop("_calibrating_version", None)
        self.__dict__.update(dic)
        return self

    def copy(self):
        new = type(self)()
        new.load(self.dump(return_dict=True))
        return new

    def dump(self, path="", return_dict=False):
        dic = {k: v.tolist() if isinstance(v, np.ndarray) else (list(v) if isinstance(v, tuple) else v)
               for k, v in self.__dict__.items() if k in ["D", "xy", "name", "T_in_main_cam", "retval"]}
        dic.update(intrinsic_format_conversion(self.K))
        if return_dict:
            return dic
        from .__info__ import __version__
        dic["_calibrating_version"] = __version__
        yamlstr = yaml.safe_dump(dic)
        if path:
            with open(path, "w") as f:
                f.write(yamlstr)
        return yamlstr

    def project_cam2_depth(cam1, cam2, depth2, T=None, interpolation=1.5):
        """Depth image of ``cam2`` re-projected into this camera (camera.py:298-309), on the GPU.
        ``T`` = pose of cam2 in this camera (4x4); the reference's fallback that derives it from calibration
        board detections (``get_T_cam2_in_self``) is outside the MI355X path, so ``T`` is required."""
        if T is None:
            raise NotImplementedError("pass T (cam2 in cam1): board-based extrinsics are outside the MI355X path")
        from . import pointcloud
        rate = pointcloud.get_appropriate_interpolation_rate(cam1, cam2, interpolation)
        return pointcloud.project_depth(depth2, cam2.K, T, cam1.K, cam1.xy, interpolation_rate=rate)
