"""``Stereo`` -- the stereo pipeline object of the reference with its per-pair work on the MI355X.

Mirrors /root/reference/calibrating/stereo_camera.py for the depth path:
  load / dump / copy                  :244-302
  _get_undistort_rectify_map          :125-185   (init time, host NumPy)
  stereo_recitfy                      :199-214   (init time, host NumPy; spelling is the reference's)
  rectify                             :216-242   -> imgproc.remap (Lanczos-4 kernel, x-shift fused)
  set_stereo_matching                 :466-489
  get_depth                           :492-533   -> kernels, every intermediate stays in HBM
  disparity_to_depth                  :408-413
  unrectify_depth                     :415-428   -> imgproc.unrectify_depth
  undistort_img                       :430-431   -> imgproc.remap_fixed_bilinear
Extrinsic calibration (cv2.stereoCalibrate, :95-123) and ``distort_depth`` (:433-464) are outside
the hot path: construct the rig from known {K, D, R, t} via ``Stereo.load`` or the R/t keywords.

NumPy in -> NumPy out (one H2D copy of the pair, one D2H copy per result entry); torch CUDA tensors
in -> torch tensors out, zero-copy.
"""
import numpy as np
import yaml

from . import geometry, imgproc
from .__info__ import __version__
from .camera import Cam
from .stereo_matching import SemiGlobalBlockMatching


def _npa(v):
    return np.array(v)


class Stereo:
    def __init__(self, cam1=None, cam2=None, xy_target=None, K_target=1, R=None, t=None):
        """K_target: float or np.array(3, 3): the new camera intrinsic; a float multiplies fx, fy."""
        self.xy_target = xy_target
        self.K_target = K_target
        if cam1 is None:
            return
        self.cam1 = cam1
        self.cam2 = cam2
        if R is None or t is None:
            raise NotImplementedError(
                "extrinsic calibration from board detections (cv2.stereoCalibrate) is outside the MI355X "
                "stereo-depth path; pass R= and t=, or use Stereo.load(dict(R=..., t=..., cam1=..., cam2=...))")
        self.R = np.float64(R)
        self.t = np.float64(t).reshape(3, 1)
        self._get_undistort_rectify_map()

    # ---- init-time tables (host) -----------------------------------------------------------------
    def _get_undistort_rectify_map(self):
        self.stereo_recitfy()
        if self.xy_target is None:
            self.xy_target = self.cam1.xy
        if isinstance(self.xy_target, (int, float)):
            self.xy_target = [int(round(i * self.xy_target)) for i in self.cam1.xy]
        self.xy = xy = tuple(self.xy_target)
        self.K = self.K_target
        if isinstance(self.K_target, (int, float)):
            self.K = self.cam1.K.copy()
            self.K[:2, :2] *= self.K_target
            self.K[:2, 2] += (np.array(xy) - self.cam1.xy) / 2
        if not isinstance(self.K_target, np.ndarray):  # "better_cx_cy"

            def get_center(xy, K, R):
                corner_uvs_real = [[0, 0, 1], [xy[0], 0, 1], list(xy) + [1], [0, xy[1], 1]]
                corner_xyz_old = np.array(corner_uvs_real) @ np.linalg.inv(K).T
                corner_xyz = corner_xyz_old @ R.T
                corner_uvs = corner_xyz @ self.K.T
                corner_uvs = corner_uvs[:, :2] / corner_uvs[:, 2:]
                center_uv = corner_uvs.mean(0)
                return center_uv - self.K[:2, 2]

            center1 = get_center(self.cam1.xy, self.cam1.K, self.R1)
            center2 = get_center(self.cam2.xy, self.cam2.K, self.R2)
            center = (center1 + center2) / 2
            self.K[:2, 2] = np.array(xy) / 2 - center

        # The maps themselves are built on the GPU when a stage first needs them (_tables); the host copies
        # (undistort_rectify_map1/2, rectify_valid_mask1: public attributes in the reference) are lazy.
        for k in ("_map1", "_map2", "_mask1", "_unrectify_depth_maps"):
            self.__dict__.pop(k, None)
        self._dev = {}  # device tables, built lazily per device

    # host views of the tables, computed on first access (NumPy, float64 internally)
    @property
    def undistort_rectify_map1(self):
        if "_map1" not in self.__dict__:
            self._map1 = geometry.init_undistort_rectify_map(self.cam1.K, self.cam1.D, self.R1, self.K, self.xy)
        return self._map1

    @property
    def undistort_rectify_map2(self):
        if "_map2" not in self.__dict__:
            self._map2 = geometry.init_undistort_rectify_map(self.cam2.K, self.cam2.D, self.R2, self.K, self.xy)
        return self._map2

    @property
    def rectify_valid_mask1(self):
        if "_mask1" not in self.__dict__:
            mapx, mapy = self.undistort_rectify_map1
            x, y = self.cam1.xy
            self._mask1 = (-0.5 < mapx) & (mapx < x - 0.5) & (-0.5 < mapy) & (mapy < y - 0.5)
        return self._mask1

    def stereo_recitfy(self):
        axes_z = np.array([0, 0, 1.0])
        axes_nx = np.array([-1.0, 0, 0])
        t = self.t.squeeze()
        plane_v = t
        z_on_plane2 = geometry.project_vec_on_plane(axes_z, plane_v)
        z_on_plane1 = geometry.project_vec_on_plane(self.R @ axes_z, plane_v)
        z_on_plane = z_on_plane2 / np.linalg.norm(z_on_plane2) + z_on_plane1 / np.linalg.norm(z_on_plane1)
        R_align_x = geometry.rotate_shortest_of_two_vecs(axes_nx, t)
        R_align_z = geometry.rotate_shortest_of_two_vecs(R_align_x @ axes_z, z_on_plane)
        self.R2 = (R_align_z @ R_align_x).T
        self.R1 = self.R2 @ self.R[:3, :3]

    def _tables(self, device):
        """Device-resident tables of this rig: {map1x, map1y, map2x, map2y, mask}, built by the GPU kernel
        (camd_init_undistort_rectify_map; bit-identical to the host properties above)."""
        key = str(device)
        if key not in self._dev:
            m1x, m1y, mask = imgproc.init_undistort_rectify_map(
                self.cam1.K, self.cam1.D, self.R1, self.K, self.xy, valid_for=self.cam1.xy, device=device)
            m2x, m2y = imgproc.init_undistort_rectify_map(
                self.cam2.K, self.cam2.D, self.R2, self.K, self.xy, device=device)
            self._dev[key] = dict(map1x=m1x, map1y=m1y, map2x=m2x, map2y=m2y, mask=mask)
        return self._dev[key]

    def table_bundle(self):
        """Host copies of everything a worker rank needs (bench.py broadcasts this over RCCL)."""
        return dict(map1x=self.undistort_rectify_map1[0], map1y=self.undistort_rectify_map1[1],
                    map2x=self.undistort_rectify_map2[0], map2y=self.undistort_rectify_map2[1],
                    mask=self.rectify_valid_mask1.view(np.uint8))

    # ---- per-pair stages ---------------------------------------------------------------------------
    @staticmethod
    def _get_img(path_or_np):
        if isinstance(path_or_np, str):
            raise NotImplementedError("image files are read by the caller on this path (no cv2.imread)")
        return path_or_np

    @staticmethod
    def _to_dev(img):
        import torch
        if isinstance(img, np.ndarray):
            return torch.from_numpy(np.ascontiguousarray(img)).cuda(), True
        return img, False

    def rectify(self, img1, img2):
        i1, np1 = self._to_dev(self._get_img(img1))
        i2, _ = self._to_dev(self._get_img(img2))
        tb = self._tables(i1.device)
        shift = self.min_disparity if getattr(self, "translation_rectify_img", None) else 0
        rectify_img1 = imgproc.remap(i1, tb["map1x"], tb["map1y"], imgproc.INTER_LANCZOS4)
        rectify_img2 = imgproc.remap(i2, tb["map2x"], tb["map2y"], imgproc.INTER_LANCZOS4, x_shift=shift)
        if np1:
            return [rectify_img1.cpu().numpy(), rectify_img2.cpu().numpy()]
        return [rectify_img1, rectify_img2]

    DUMP_ATTRS = ["R", "t", "retval"]

    def dump(self, path="", return_dict=False):
        dic = {k: v.tolist() if isinstance(v, np.ndarray) else v
               for k, v in self.__dict__.items() if k in self.DUMP_ATTRS}
        dic["cam1"] = self.cam1.dump(return_dict=True)
        dic["cam2"] = self.cam2.dump(return_dict=True)
        if return_dict:
            return dic
        dic["_calibrating_version"] = __version__
        yamlstr = yaml.safe_dump(dic)
        if path:
            with open(path, "w") as f:
                f.write(yamlstr)
        return yamlstr

    def load(self, path_or_str_or_dict=None):
        if path_or_str_or_dict is None:
            path_or_str_or_dict = self
            self = Stereo()
        if isinstance(path_or_str_or_dict, Stereo):
            return path_or_str_or_dict.copy()
        if not isinstance(path_or_str_or_dict, (list, dict)):
            path_or_str = path_or_str_or_dict
            if "\n" in path_or_str:
                dic = yaml.safe_load(path_or_str)
            else:
                with open(path_or_str) as f:
                    dic = yaml.safe_load(f)
        else:
            dic = dict(path_or_str_or_dict)
        dic.pop("_calibrating_version", None)
        if hasattr(self, "cam1"):
            self.cam1.load(dic.pop("cam1"))
            self.cam2.load(dic.pop("cam2"))
        else:
            self.cam1 = Cam.load(dic.pop("cam1"))
            self.cam2 = Cam.load(dic.pop("cam2"))
        if "R" not in dic and "T" in dic:
            dic["r"], dic["t"] = geometry.T_to_r_t(np.asarray(dic.pop("T"), np.float64))
        if "R" not in dic and "r" in dic:
            dic["R"] = geometry.rodrigues(np.asarray(dic.pop("r"), np.float64).reshape(3))
        dic.setdefault("R", np.eye(3))
        self.__dict__.update({k: _npa(v) if k in self.DUMP_ATTRS else v for k, v in dic.items()})
        self.R = np.float64(self.R)
        self.t = np.float64(self.t).reshape(3, 1)
        self._get_undistort_rectify_map()
        return self

    def copy(self):
        new = type(self)()
        new.load(self.dump())
        return new

    def __str__(self):
        r = geometry.rodrigues(self.R).squeeze()
        du = np.linalg.norm(r) * 180 / np.pi
        strr = "Stereo(cam1='%s', cam2='%s'):\n" % (self.cam1.name, self.cam2.name)
        strr += "\t xy: %s\n" % str(list(self.cam1.xy))[1:-1]
        strr += "\t baseline: %.2fcm\n" % (100 * self.baseline)
        strr += "\t t(cm): [%s]\n" % (" ".join([str(i) for i in (self.t.squeeze() * 100).round(2)]))
        strr += "\t r(rodrigues): [%s] %.2f deg\n" % (" ".join([str(i) for i in r.round(3)]), du)
        if hasattr(self, "retval"):
            strr += "\t retval: %s\n" % self.retval
        return strr

    __repr__ = __str__

    MAX_DEPTH = 1000

    def get_max_depth(self):
        return getattr(self, "max_depth", self.MAX_DEPTH)

    @property
    def D(self):
        return np.zeros((1, 5))

    @property
    def T(self):
        return geometry.R_t_to_T(self.R, self.t)

    @property
    def baseline(self):
        return np.sum(self.t ** 2) ** 0.5

    def depth_to_disparity(self, depth):
        fx = self.K[0, 0]
        return 1.0 * self.baseline * fx / depth

    def disparity_to_depth(self, disparity):
        """NumPy or torch ``disparity`` -> depth, same dtype rules and edge cases as :408-413."""
        fx = self.K[0, 0]
        bf = 1.0 * self.baseline * fx
        if isinstance(disparity, np.ndarray):
            with np.errstate(divide="ignore"):
                depth = bf / disparity
            depth[depth > self.get_max_depth()] = 0
            depth[depth < 0] = 0
            return depth
        import torch
        depth = float(bf) / disparity.to(torch.float64)
        depth[depth > self.get_max_depth()] = 0
        depth[depth < 0] = 0
        return depth

    def _unrectify_tables(self, device):
        # utils.py:183-191: initUndistortRectifyMap(K, None, R1.T, cam1.K, cam1.xy), memoised per device
        key = "unrect:" + str(device)
        if key not in self._dev:
            self._dev[key] = imgproc.init_undistort_rectify_map(self.K, None, self.R1.T, self.cam1.K, self.cam1.xy,
                                                                device=device)
        return self._dev[key]

    def unrectify_depth(self, depth):
        d, was_np = self._to_dev(depth)
        mx, my = self._unrectify_tables(d.device)
        M = self.R1.T @ np.linalg.inv(self.K)
        out = imgproc.unrectify_depth(d, M[2], mx, my)
        return out.cpu().numpy() if was_np else out

    def undistort_img(self, img1):
        i1, was_np = self._to_dev(img1)
        key = "undist:" + str(i1.device)
        if key not in self._dev:
            self._dev[key] = imgproc.undistort_maps_device(self.cam1.K, self.cam1.D, self.cam1.xy, device=i1.device)
        mxy, ma = self._dev[key]
        out = imgproc.remap_fixed_bilinear(i1, mxy, ma)
        return out.cpu().numpy() if was_np else out

    def distort_depth(self, depth):
        raise NotImplementedError("Stereo.distort_depth ('OOM warning and very slow' in the reference, "
                                  "stereo_camera.py:433-464) is outside the MI355X hot path")

    def set_stereo_matching(self, stereo_matching, max_depth=None, translation_rectify_img=None):
        """Same semantics as :466-489 (note: min_disparity uses cam1.K, disparity_to_depth self.K)."""
        self.stereo_matching = stereo_matching
        self.translation_rectify_img = (bool(max_depth) if translation_rectify_img is None
                                        else translation_rectify_img)
        self.max_depth = max_depth or self.MAX_DEPTH
        self.min_disparity = int(self.cam1.K[0, 0] * self.baseline / self.max_depth)
        return self

    def get_depth(self, img1, img2, return_unrectify_depth=True, return_distort_depth=False):
        """Return dict: rectify_img1, rectify_depth, disparity, rectify_img2 (+ unrectify_depth,
        undistort_img1). Depth unit is m; 0 = invalid."""
        import torch
        result = {}
        assert hasattr(self, "stereo_matching"), "Please stereo.set_stereo_matching(stereo_matching)"
        if return_distort_depth:
            self.distort_depth(None)
        i1, was_np = self._to_dev(self._get_img(img1))
        i2, _ = self._to_dev(self._get_img(img2))
        rectify_img1, rectify_img2 = self.rectify(i1, i2)
        tb = self._tables(i1.device)
        sm = self.stereo_matching
        translate = bool(getattr(self, "translation_rectify_img"))
        fused = isinstance(sm, SemiGlobalBlockMatching) and \
            min(sm.max_size / max(rectify_img1.shape[:2]), 1) == 1
        if fused:
            # matcher post-processing, += min_disparity, * mask and disparity_to_depth in one kernel
            disp16, _ = sm.compute_disp16(rectify_img1, rectify_img2)
            disparity, rectify_depth = imgproc.disp_to_depth(
                disp16, tb["mask"], sm.stereo_sgbm.getMinDisparity(), self.min_disparity, translate,
                1.0 * self.baseline * self.K[0, 0], self.get_max_depth())
        else:
            if isinstance(sm, SemiGlobalBlockMatching):
                disparity = sm(rectify_img1, rectify_img2)
            else:  # foreign plugin: reference contract is NumPy in / NumPy (or dict) out
                disparity = sm(rectify_img1.cpu().numpy(), rectify_img2.cpu().numpy())
            if isinstance(disparity, dict):
                result.update(disparity)
                disparity = disparity["disparity"]
            if isinstance(disparity, np.ndarray):
                disparity = torch.from_numpy(np.ascontiguousarray(disparity)).to(i1.device)
            if translate:
                disparity += self.min_disparity
            disparity = tb["mask"].to(torch.bool) * disparity
            rectify_depth = self.disparity_to_depth(disparity)

        result.update(rectify_img1=rectify_img1, rectify_depth=rectify_depth, disparity=disparity,
                      rectify_img2=rectify_img2)
        if return_unrectify_depth:
            result.update(unrectify_depth=self.unrectify_depth(rectify_depth),
                          undistort_img1=self.undistort_img(i1))
        if was_np:
            result = {k: v.cpu().numpy() if isinstance(v, torch.Tensor) else v for k, v in result.items()}
        return result

    def get_depth_batch(self, imgs1, imgs2, return_unrectify_depth=True):
        """``get_depth`` for ``n`` pairs of the same rig at once: ``imgs1`` / ``imgs2`` are ``(n, h, w, 3)``
        uint8 (NumPy or torch CUDA), every value of the returned dict carries the leading ``n``.

        Not in the reference (its ``get_depth`` takes one pair, stereo_camera.py:491-533); this is the
        throughput form of the same stages -- each kernel is launched once for the whole batch, which is
        what keeps small images (VGA) from being launch-bound.  Pair ``i`` of the result is bit-identical
        to ``get_depth(imgs1[i], imgs2[i])``.  Requires the SGBM plugin without downsizing.
        """
        import torch
        assert hasattr(self, "stereo_matching"), "Please stereo.set_stereo_matching(stereo_matching)"
        i1, was_np = self._to_dev(imgs1)
        i2, _ = self._to_dev(imgs2)
        if i1.dim() != 4 or i1.shape != i2.shape:
            raise ValueError("imgs1 / imgs2 must be (n, h, w, c) arrays of equal shape")
        sm = self.stereo_matching
        if not (isinstance(sm, SemiGlobalBlockMatching) and min(sm.max_size / max(i1.shape[1:3]), 1) == 1):
            raise ValueError("get_depth_batch needs a SemiGlobalBlockMatching plugin with max_size >= image size")
        rectify_img1, rectify_img2 = self.rectify(i1, i2)
        tb = self._tables(i1.device)
        disp16 = sm.stereo_sgbm.compute(rectify_img1, rectify_img2)
        disparity, rectify_depth = imgproc.disp_to_depth(
            disp16, tb["mask"], sm.stereo_sgbm.getMinDisparity(), self.min_disparity,
            bool(getattr(self, "translation_rectify_img")), 1.0 * self.baseline * self.K[0, 0], self.get_max_depth())
        result = dict(rectify_img1=rectify_img1, rectify_depth=rectify_depth, disparity=disparity,
                      rectify_img2=rectify_img2)
        if return_unrectify_depth:
            result.update(unrectify_depth=self.unrectify_depth(rectify_depth), undistort_img1=self.undistort_img(i1))
        if was_np:
            result = {k: v.cpu().numpy() if isinstance(v, torch.Tensor) else v for k, v in result.items()}
        return result
