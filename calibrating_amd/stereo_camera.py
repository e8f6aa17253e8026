"""``Stereo`` -- the stereo pipeline object of the reference with its per-pair work on the MI355X.

Public names and semantics follow /root/reference/calibrating/stereo_camera.py for the depth path;
the bodies are this package's own: rig geometry lives in ``geometry.py`` as pure functions
(``rectifying_rotations``, ``target_intrinsics``), the per-pair stages call the gfx950 kernels.
  load / dump / copy                  :244-302   record = {R | T | r, t, cam1, cam2[, retval]}
  _get_undistort_rectify_map          :125-185   (init time; maps built lazily, on the GPU)
  stereo_recitfy                      :199-214   (spelling is the reference's)
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

from . import geometry, hostio, imgproc
from .camera import Cam, read_record, write_record
from .stereo_matching import SemiGlobalBlockMatching

_TABLE_KEYS = ("map1x", "map1y", "map2x", "map2y", "mask")
# A broadcast bundle: the six float32 maps, the validity mask and a block of 64 doubles
# from which a worker rank builds its rig without ever seeing the rig record.
_BUNDLE_MAGIC = 20250930.0
_P = dict(magic=(0, 1), K=(1, 10), R1=(10, 19), R2=(19, 28), cam1_K=(28, 37), cam1_D=(37, 51), cam1_xy=(51, 53),
          cam2_xy=(53, 55), xy=(55, 57), t=(57, 60), nD=(60, 61))


class PendingDepth:
    """What ``Stereo.get_depth_async`` returns: ``result()`` -> the dict of ``get_depth`` (waits for this call's copies;
    device-tensor calls have nothing to wait for).  A device-side failure of the matcher (a bounded wait that expired)
    is not checked here -- that would wait for every later call too; the matcher's next ``compute`` / ``status()`` raise."""

    def __init__(self, result, sink):
        self._result, self._sink = result, sink

    def result(self):
        if self._sink is not None:
            self._result.update(self._sink.collect(own_copies_only=True))
            self._sink = None
        return self._result


class Stereo:
    DUMP_ATTRS = ["R", "t", "retval"]
    MAX_DEPTH = 1000

    def __init__(self, cam1=None, cam2=None, xy_target=None, K_target=1, R=None, t=None):
        """K_target: float or np.array(3, 3): the new camera intrinsic; a float multiplies fx, fy."""
        self.xy_target = xy_target
        self.K_target = K_target
        if cam1 is None:
            return
        if R is None or t is None:
            raise NotImplementedError(
                "extrinsic calibration from board detections (cv2.stereoCalibrate) is outside the MI355X "
                "stereo-depth path; pass R= and t=, or use Stereo.load(dict(R=..., t=..., cam1=..., cam2=...))")
        self.cam1, self.cam2 = cam1, cam2
        self._set_pose(R, t)
        self._get_undistort_rectify_map()

    def _set_pose(self, R, t):
        self.R = np.float64(R)
        self.t = np.float64(t).reshape(3, 1)

    # ---- init-time geometry (host, 3x3 algebra only) -----------------------------------------------
    def stereo_recitfy(self):
        self.R1, self.R2 = geometry.rectifying_rotations(self.R, self.t)

    def _get_undistort_rectify_map(self):
        self.stereo_recitfy()
        self.xy, self.K = geometry.target_intrinsics(self.cam1.K, self.cam1.xy, self.cam2.K, self.cam2.xy,
                                                     self.R1, self.R2, self.xy_target, self.K_target)
        if self.xy_target is None:  # the attribute ends up resolved, as in the reference
            self.xy_target = self.cam1.xy
        elif isinstance(self.xy_target, (int, float)):
            self.xy_target = list(self.xy)
        # The four maps + valid mask are built on the GPU when a stage first needs them (_tables); the host
        # copies (undistort_rectify_map1/2, rectify_valid_mask1: public attributes in the reference) are lazy.
        self._host, self._dev = {}, {}

    @staticmethod
    def _dev_key(device):
        """One spelling per device: torch.device("cuda") and "cuda:0" name the same tables (the current device's)."""
        import torch
        d = torch.device(device)
        if d.type == "cuda" and d.index is None:
            d = torch.device("cuda", torch.cuda.current_device())
        return str(d)

    def _host_table(self, name, build):
        if name not in self._host:
            if getattr(self, "_bundle_only", False):  # no camera 2 intrinsics to rebuild from: the installed tensors
                tables = [v for k, v in self._dev.items() if not k.startswith("unrect:")]
                if not tables:
                    raise RuntimeError("this rig was built from a table bundle but holds no installed tables")
                tb = tables[0]
                host = lambda t: t.cpu().numpy()  # noqa: E731
                self._host.update(map1=(host(tb["map1x"]), host(tb["map1y"])), map2=(host(tb["map2x"]), host(tb["map2y"])),
                                  mask1=host(tb["mask"]).astype(bool))
                un = [v for k, v in self._dev.items() if k.startswith("unrect:")]
                if un:
                    self._host["unrect"] = (host(un[0][0]), host(un[0][1]))
            if name not in self._host:
                if getattr(self, "_bundle_only", False) and name != "unrect":
                    raise RuntimeError("table %r cannot be rebuilt on a rig made from a table bundle" % name)
                self._host[name] = build()  # ("unrect" needs K, R1 and camera 1 only: a bundle rig has them)
        return self._host[name]

    @property
    def undistort_rectify_map1(self):
        return self._host_table("map1", lambda: geometry.init_undistort_rectify_map(
            self.cam1.K, self.cam1.D, self.R1, self.K, self.xy))

    @property
    def undistort_rectify_map2(self):
        return self._host_table("map2", lambda: geometry.init_undistort_rectify_map(
            self.cam2.K, self.cam2.D, self.R2, self.K, self.xy))

    @property
    def rectify_valid_mask1(self):
        return self._host_table("mask1", lambda: geometry.valid_mask_from_maps(
            *self.undistort_rectify_map1, self.cam1.xy))

    # ---- device tables -----------------------------------------------------------------------------
    def _tables(self, device):
        """Device-resident tables of this rig: {map1x, map1y, map2x, map2y, mask}, built by the GPU kernel
        (camd_init_undistort_rectify_map; bit-identical to the host properties above) unless a broadcast
        bundle was installed for ``device`` (install_tables)."""
        key = self._dev_key(device)
        if key not in self._dev:
            if getattr(self, "_bundle_only", False):
                # camera 2's intrinsics are not part of a bundle (cam2.K is NaN): rebuilding here would hand back NaN
                # maps and garbage depth without a word
                raise RuntimeError("this rig was built from a table bundle installed on %s; it has no tables for %s and "
                                   "cannot rebuild them (a bundle carries no camera 2 intrinsics) -- install_tables() the "
                                   "bundle on that device, or load the rig's record there"
                                   % (sorted(k for k in self._dev if not k.startswith("unrect:")), key))
            m1x, m1y, mask = imgproc.init_undistort_rectify_map(
                self.cam1.K, self.cam1.D, self.R1, self.K, self.xy, valid_for=self.cam1.xy, device=device)
            m2x, m2y = imgproc.init_undistort_rectify_map(
                self.cam2.K, self.cam2.D, self.R2, self.K, self.xy, device=device)
            self._dev[key] = dict(map1x=m1x, map1y=m1y, map2x=m2x, map2y=m2y, mask=mask)
        return self._dev[key]

    def table_bundle(self):
        """Host copies of everything a worker rank needs (parallel_pairs.broadcast_tables sends this): the rectify
        maps of both cameras, the validity mask, the unrectify maps (utils.py:183-191, memoised by the reference in
        ``_unrectify_depth_maps``) and ``params``: 64 float64 -- K, R1, R2, cam1.K, cam1.D, the three image sizes and t
        (what :159-176,408-431,466-489 read).  ~52 MB at 1080p.  ``Stereo.from_bundle`` is the receiving side."""
        (m1x, m1y), (m2x, m2y) = self.undistort_rectify_map1, self.undistort_rectify_map2
        ux, uy = self._host_table("unrect", lambda: geometry.init_undistort_rectify_map(
            self.K, None, self.R1.T, self.cam1.K, self.cam1.xy))
        p = np.zeros(64)
        d = np.asarray(self.cam1.D, np.float64).reshape(-1)
        if d.size > 14:
            raise ValueError("camera 1 has %d distortion coefficients; the bundle carries up to 14" % d.size)
        for name, val in (("magic", _BUNDLE_MAGIC), ("K", self.K), ("R1", self.R1), ("R2", self.R2), ("cam1_K", self.cam1.K),
                          ("cam1_xy", self.cam1.xy), ("cam2_xy", self.cam2.xy), ("xy", self.xy), ("t", self.t),
                          ("nD", d.size)):
            lo, hi = _P[name]
            p[lo:hi] = np.asarray(val, np.float64).reshape(-1)
        p[_P["cam1_D"][0]:_P["cam1_D"][0] + d.size] = d
        return dict(map1x=m1x, map1y=m1y, map2x=m2x, map2y=m2y, mask=self.rectify_valid_mask1.view(np.uint8),
                    unrect_mapx=ux, unrect_mapy=uy, params=p)

    def install_tables(self, bundle, device=None):
        """Use the tensors of a broadcast table bundle (parallel_pairs.broadcast_tables) as this rig's device
        tables on ``device`` instead of rebuilding them: the worker-rank side of the one-time RCCL broadcast."""
        missing = [k for k in _TABLE_KEYS if k not in bundle]
        if missing:
            raise ValueError("table bundle lacks %s" % missing)
        h, w = bundle["map1x"].shape
        if (w, h) != tuple(self.xy):
            raise ValueError("bundle is for a %dx%d rectified image, this rig rectifies to %dx%d" % (w, h, *self.xy))
        device = bundle["map1x"].device if device is None else device
        key = self._dev_key(device)
        if self._dev_key(bundle["map1x"].device) != key:
            raise ValueError("the bundle's tensors live on %s, not on %s" % (bundle["map1x"].device, key))
        self._dev[key] = {k: bundle[k] for k in _TABLE_KEYS}
        if "unrect_mapx" in bundle:
            self._dev["unrect:" + key] = (bundle["unrect_mapx"], bundle["unrect_mapy"])
        return self

    @classmethod
    def from_bundle(cls, bundle, device=None):
        """A rig built from a broadcast bundle ALONE -- no record, no rebuild: what a worker rank of a multi-GPU job
        holds.  It serves everything ``get_depth`` / ``get_depth_batch`` touch (tables from the bundle; K, R1, t,
        cam1.K / D / xy from ``params``); camera 2's intrinsics and the rig's R are not part of a bundle (nothing on
        the depth path reads them once the maps exist), so ``dump`` and a rebuild for another target frame need the
        record."""
        p = bundle["params"]
        p = np.asarray(p.cpu() if hasattr(p, "cpu") else p, np.float64).reshape(-1)
        if p.size != 64 or p[0] != _BUNDLE_MAGIC:
            raise ValueError("not a table bundle of this version (params block: %d doubles, magic %r)" % (p.size, p[:1]))
        get = lambda name: p[_P[name][0]:_P[name][1]]  # noqa: E731
        nD = int(get("nD")[0])
        xy_of = lambda name: tuple(int(v) for v in get(name))  # noqa: E731
        st = cls(xy_target=list(xy_of("xy")), K_target=get("K").reshape(3, 3).copy())
        st.cam1 = Cam(get("cam1_K").reshape(3, 3).copy(), get("cam1_D")[:nD].reshape(1, -1).copy() if nD else None,
                      xy_of("cam1_xy"), name="cam1 (from a table bundle)")
        st.cam2 = Cam(np.full((3, 3), np.nan), None, xy_of("cam2_xy"), name="cam2 (maps only: from a table bundle)")
        st.t = get("t").reshape(3, 1).copy()
        st.R1, st.R2 = get("R1").reshape(3, 3).copy(), get("R2").reshape(3, 3).copy()
        st.K, st.xy = get("K").reshape(3, 3).copy(), xy_of("xy")
        st._host, st._dev = {}, {}
        st._bundle_only = True
        return st.install_tables(bundle, device)

    # ---- record I/O --------------------------------------------------------------------------------
    def dump(self, path="", return_dict=False):
        if getattr(self, "_bundle_only", False):
            raise ValueError("this rig was built from a broadcast table bundle (Stereo.from_bundle): it holds neither the "
                             "rig's R nor camera 2's intrinsics, so there is no record to dump -- dump the rig on the rank "
                             "that loaded it")
        rec = {k: (v.tolist() if isinstance(v, np.ndarray) else v)
               for k, v in vars(self).items() if k in self.DUMP_ATTRS}
        rec.update(cam1=self.cam1.dump(return_dict=True), cam2=self.cam2.dump(return_dict=True))
        return rec if return_dict else write_record(rec, path)

    def load(self, path_or_str_or_dict=None):
        """``Stereo.load(record)`` (on the class: new object) or ``stereo.load(record)`` (in place; xy_target /
        K_target of the object are kept).  record: dict, YAML text or YAML file path."""
        if path_or_str_or_dict is None:  # Stereo.load(x) binds x to `self`
            return Stereo().load(self)
        if isinstance(path_or_str_or_dict, Stereo):
            return path_or_str_or_dict.copy()
        rec = read_record(path_or_str_or_dict) if not isinstance(path_or_str_or_dict, dict) \
            else dict(path_or_str_or_dict)
        rec.pop("_calibrating_version", None)
        for name in ("cam1", "cam2"):
            if hasattr(self, name):
                getattr(self, name).load(rec.pop(name))
            else:
                setattr(self, name, Cam.load(rec.pop(name)))
        geometry.rig_rotation_from_record(rec)
        for k, v in rec.items():
            setattr(self, k, np.array(v) if k in self.DUMP_ATTRS else v)
        self._set_pose(self.R, self.t)
        self._get_undistort_rectify_map()
        return self

    def copy(self):
        return type(self)().load(self.dump())

    def __str__(self):
        # like the reference's (stereo_camera.py:363-382) printing never raises: a rig loaded from a record without
        # camera names / sizes still prints, with the failure appended
        lines = []
        try:
            rvec = geometry.rodrigues(self.R).reshape(3)
            lines.append("Stereo(cam1='%s', cam2='%s'):" % (getattr(self.cam1, "name", None), getattr(self.cam2, "name", None)))
            lines += ["\t xy: %s" % ", ".join(str(v) for v in self.cam1.xy),
                      "\t baseline: %.2fcm" % (100 * self.baseline),
                      "\t t(cm): [%s]" % " ".join(str(v) for v in (self.t.reshape(3) * 100).round(2)),
                      "\t r(rodrigues): [%s] %.2f\u00b0" % (" ".join(str(v) for v in rvec.round(3)),
                                                          np.degrees(np.linalg.norm(rvec)))]
            lines.append("\t cam1.fovs: %s" % ", ".join("%s=%s\u00b0" % (k, round(v, 2)) for k, v in self.cam1.fovs.items()))
            if hasattr(self, "retval"):
                lines.append("\t retval: %s" % self.retval)
        except Exception as e:
            lines.append("\t Exception(%s) in Stereo.__str__()" % e)
        return "\n".join(lines) + "\n"

    __repr__ = __str__

    # ---- small properties (names of the reference, :386-406) ----------------------------------------
    def get_max_depth(self):
        return getattr(self, "max_depth", self.MAX_DEPTH)

    @property
    def D(self):
        return np.zeros((1, 5))  # the rectified camera has no distortion

    @property
    def T(self):
        return geometry.R_t_to_T(self.R, self.t)

    @property
    def baseline(self):
        return float(np.sqrt(np.sum(np.square(self.t))))

    def depth_to_disparity(self, depth):
        bf = 1.0 * self.baseline * self.K[0, 0]
        if isinstance(depth, np.ndarray) or np.isscalar(depth):
            return bf / depth
        import torch  # float64 like NumPy's float64-scalar / array, and an IEEE division (see disparity_to_depth)
        return torch.full((), float(bf), dtype=torch.float64, device=depth.device) / depth.to(torch.float64)

    def disparity_to_depth(self, disparity):
        """NumPy or torch ``disparity`` -> depth, same dtype rules and edge cases as :408-413: inf (d = 0)
        and everything beyond max_depth become 0, then negatives become 0."""
        bf = 1.0 * self.baseline * self.K[0, 0]
        if isinstance(disparity, np.ndarray):
            with np.errstate(divide="ignore"):
                depth = bf / disparity
        else:
            import torch
            # (a tensor numerator: `number / tensor` would be computed as reciprocal(tensor) * number, two roundings)
            depth = torch.full((), float(bf), dtype=torch.float64, device=disparity.device) / disparity.to(torch.float64)
        depth[depth > self.get_max_depth()] = 0
        depth[depth < 0] = 0
        return depth

    # ---- per-pair stages ---------------------------------------------------------------------------
    @staticmethod
    def _get_img(path_or_np):
        """A file path is read as RGB uint8 (the reference: ``cv2.imread(path)[..., ::-1]``, :304-308; here
        through Pillow, which decodes PNG/BMP/PPM to the same bytes -- JPEG decoders may differ by 1 LSB)."""
        if isinstance(path_or_np, str):
            from PIL import Image
            with Image.open(path_or_np) as im:
                return np.array(im.convert("RGB"))
        return path_or_np

    @staticmethod
    def _to_dev(img):
        import torch
        if isinstance(img, np.ndarray):
            # (a plain pageable copy: 0.12 ms per 1080p image on the GPU box; staging through a page-locked block
            # was measured 10x slower, see hostio)
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
            return hostio.to_host(rectify_img1, rectify_img2)
        return [rectify_img1, rectify_img2]

    def _unrectify_tables(self, device):
        # utils.py:183-191: initUndistortRectifyMap(K, None, R1.T, cam1.K, cam1.xy), memoised per device
        key = "unrect:" + self._dev_key(device)
        if key not in self._dev:
            self._dev[key] = imgproc.init_undistort_rectify_map(self.K, None, self.R1.T, self.cam1.K, self.cam1.xy,
                                                                device=device)
        return self._dev[key]

    def unrectify_depth(self, depth):
        d, was_np = self._to_dev(depth)
        mx, my = self._unrectify_tables(d.device)
        M = self.R1.T @ np.linalg.inv(self.K)
        out = imgproc.unrectify_depth(d, M[2], mx, my)
        return hostio.to_host(out) if was_np else out

    def undistort_img(self, img1):
        i1, was_np = self._to_dev(self._get_img(img1))
        key = "undist:" + str(i1.device)
        if key not in self._dev:
            self._dev[key] = imgproc.undistort_maps_device(self.cam1.K, self.cam1.D, self.cam1.xy, device=i1.device)
        mxy, ma = self._dev[key]
        out = imgproc.remap_fixed_bilinear(i1, mxy, ma)
        return hostio.to_host(out) if was_np else out

    def distort_depth(self, depth):
        raise NotImplementedError("Stereo.distort_depth ('OOM warning and very slow' in the reference, "
                                  "stereo_camera.py:433-464) is outside the MI355X hot path")

    def set_stereo_matching(self, stereo_matching, max_depth=None, translation_rectify_img=None):
        """Install the matcher plugin (:466-489).  ``max_depth`` (default MAX_DEPTH) fixes
        ``min_disparity = int(cam1.fx * baseline / max_depth)`` -- note cam1's fx here, the rectified K in
        disparity_to_depth; ``translation_rectify_img`` defaults to ``bool(max_depth)``."""
        self.stereo_matching = stereo_matching
        self.translation_rectify_img = bool(max_depth) if translation_rectify_img is None \
            else translation_rectify_img
        self.max_depth = max_depth or self.MAX_DEPTH
        self.min_disparity = int(self.cam1.K[0, 0] * self.baseline / self.max_depth)
        return self

    def _native_sgbm(self):
        """The installed plugin if it is this package's SGBM plugin with ITS OWN ``__call__`` -- only then may the stages
        after the match be fused on the device.  A subclass that overrides ``__call__`` (extra result keys, its own
        post-filter) goes through ``plugin(img1, img2)`` like any foreign plugin, as in the reference
        (stereo_camera.py:506-509)."""
        sm = self.stereo_matching
        if isinstance(sm, SemiGlobalBlockMatching) and type(sm).__call__ is SemiGlobalBlockMatching.__call__:
            return sm
        return None

    def _sgbm_full_res(self, rectified_hw):
        """The SGBM plugin when it runs at the rectified resolution (no max_size downsizing), else None."""
        sm = self._native_sgbm()
        if sm is not None and min(sm.max_size / max(rectified_hw), 1) == 1:
            return sm
        return None

    def _fused_depth(self, sm, disp16, tables, hw=None):
        """Matcher post-processing, += min_disparity, * mask and disparity_to_depth in one kernel.  ``hw``: the rectified
        size when ``disp16`` belongs to a DOWNSIZED pair (cfg["max_size"] below the image: the reference's default):
        the resize back (stereo_matching.py:66) is then part of the same pass."""
        args = (tables["mask"], sm.stereo_sgbm.getMinDisparity(), self.min_disparity, bool(self.translation_rectify_img),
                1.0 * self.baseline * self.K[0, 0], self.get_max_depth())
        if hw is not None and tuple(hw) != tuple(disp16.shape[-2:]):
            return imgproc.disp16_resized_to_depth(disp16, hw, *args)
        return imgproc.disp_to_depth(disp16, *args)

    RESULT_KEYS = ("rectify_img1", "rectify_img2", "disparity", "rectify_depth", "unrectify_depth", "undistort_img1")

    def get_depth_async(self, img1, img2, return_unrectify_depth=True, keys=None):
        """``get_depth`` without its final wait (not in the reference, whose calls are synchronous): everything is
        queued -- upload, kernels, the results' way back to page-locked host blocks -- and a ``PendingDepth`` is returned;
        ``.result()`` waits for THIS call's copies only and hands over the same dict ``get_depth`` returns.  A caller
        that keeps two or three calls in flight overlaps the ~1 ms a 1080p result dict spends on PCIe with the next
        call's kernels: one pair per call then runs at the rate of the kernels alone (tools/gpu_numpy_latency.py)."""
        return self._get_depth(img1, img2, return_unrectify_depth, False, keys, defer=True)

    def get_depth(self, img1, img2, return_unrectify_depth=True, return_distort_depth=False, keys=None):
        """Return dict: rectify_img1, rectify_depth, disparity, rectify_img2 (+ unrectify_depth,
        undistort_img1). Depth unit is m; 0 = invalid.  ndarray inputs give ndarray results (each result starts
        its way to the host as soon as its kernel is queued, hostio.Sink); device tensors stay on the device.

        ``keys`` (not in the reference): the entries the caller wants, e.g. ``keys=("unrectify_depth",)``.  The full
        dict of a 1080p pair is ~60 MB -- two float64 depth maps among them -- and its way back over PCIe costs as much
        as a third of the kernels; entries that are not asked for are neither copied nor, where nothing else needs
        them (undistort_img1, unrectify_depth), computed.  ``None`` = the reference's dict."""
        return self._get_depth(img1, img2, return_unrectify_depth, return_distort_depth, keys, defer=False)

    def _get_depth(self, img1, img2, return_unrectify_depth, return_distort_depth, keys, defer):
        """The body of get_depth / get_depth_async (``defer``: hand back a PendingDepth instead of waiting)."""
        import torch
        assert hasattr(self, "stereo_matching"), "Please stereo.set_stereo_matching(stereo_matching)"
        if return_distort_depth:
            self.distort_depth(None)
        if keys is not None:
            keys = (keys,) if isinstance(keys, str) else tuple(keys)
            unknown = [k for k in keys if k not in self.RESULT_KEYS]
            if unknown:
                raise ValueError("get_depth(keys=...): unknown result entries %s (known: %s)" % (unknown, list(self.RESULT_KEYS)))
            return_unrectify_depth = "unrectify_depth" in keys or "undistort_img1" in keys
        want = (lambda k: True) if keys is None else (lambda k: k in keys)
        img1, img2 = self._get_img(img1), self._get_img(img2)
        was_np = isinstance(img1, np.ndarray)
        sink, result = None, {}

        def emit(**tensors):
            for k, t in tensors.items():
                if want(k):
                    result[k] = t
                    if sink is not None:
                        sink.send(k, t)

        # camera 1 first: its rectification and undistortion are queued before camera 2's pixels are copied
        i1, _ = self._to_dev(img1)
        if was_np:
            sink = hostio.Sink(i1.device)
        tb = self._tables(i1.device)
        rectify_img1 = imgproc.remap(i1, tb["map1x"], tb["map1y"], imgproc.INTER_LANCZOS4)
        emit(rectify_img1=rectify_img1)
        if return_unrectify_depth and want("undistort_img1"):
            emit(undistort_img1=self.undistort_img(i1))  # independent of the matcher: its copy hides under SGBM
        i2, _ = self._to_dev(img2)
        shift = self.min_disparity if getattr(self, "translation_rectify_img", None) else 0
        rectify_img2 = imgproc.remap(i2, tb["map2x"], tb["map2y"], imgproc.INTER_LANCZOS4, x_shift=shift)
        emit(rectify_img2=rectify_img2)
        plugin = self.stereo_matching
        sm = self._sgbm_full_res(rectify_img1.shape[:2])
        if sm is not None:
            disp16, _ = sm.compute_disp16(rectify_img1, rectify_img2)
            disparity, rectify_depth = self._fused_depth(sm, disp16, tb)
        elif self._native_sgbm() is not None:  # the downsizing matcher: resize, match, resize back + post
            sdisp16, _ = plugin.compute_disp16(rectify_img1, rectify_img2)
            disparity, rectify_depth = self._fused_depth(plugin, sdisp16, tb, rectify_img1.shape[:2])
        else:
            # foreign plugin: the reference's contract is NumPy in, NumPy (or dict) out
            disparity = plugin(*hostio.to_host(rectify_img1, rectify_img2))
            if isinstance(disparity, dict):
                result.update({k: v for k, v in disparity.items() if k != "disparity"})
                disparity = disparity["disparity"]
            if self.translation_rectify_img:
                disparity += self.min_disparity  # in place, on the plugin's own array like :510-511
            if isinstance(disparity, np.ndarray):
                disparity = torch.from_numpy(np.ascontiguousarray(disparity)).to(i1.device)
            disparity = tb["mask"].to(torch.bool) * disparity
            rectify_depth = self.disparity_to_depth(disparity)
        if return_unrectify_depth and want("unrectify_depth"):
            # queued before the copies of disparity / rectify_depth start: the caller usually waits for this one
            unrect = self.unrectify_depth(rectify_depth)
            emit(unrectify_depth=unrect)
        emit(disparity=disparity, rectify_depth=rectify_depth)
        if defer:
            return PendingDepth(result, sink)
        if sink is not None:
            result.update(sink.collect())
            if isinstance(plugin, SemiGlobalBlockMatching):
                plugin.stereo_sgbm.status()  # collect() synchronised: surface device-side timeouts at no extra cost
        return result

    def get_depth_batch(self, imgs1, imgs2, return_unrectify_depth=True, keys=None):
        """``get_depth`` for ``n`` pairs of the same rig at once: ``imgs1`` / ``imgs2`` are ``(n, h, w, 3)``
        uint8 (NumPy or torch CUDA), every value of the returned dict carries the leading ``n``.

        Not in the reference (its ``get_depth`` takes one pair, stereo_camera.py:491-533); this is the
        throughput form of the same stages -- each kernel is launched once for the whole batch, which is
        what keeps small images (VGA) from being launch-bound.  Pair ``i`` of the result is bit-identical
        to ``get_depth(imgs1[i], imgs2[i])``.  Requires the SGBM plugin; a ``max_size`` below the rectified image
        size downsizes the whole batch first, as the matcher does for one pair.  ``keys``: as in ``get_depth`` -- only
        the asked entries are returned (and copied to the host for ndarray input; a batch of 64 1080p pairs is 3.8 GB).
        """
        assert hasattr(self, "stereo_matching"), "Please stereo.set_stereo_matching(stereo_matching)"
        if keys is not None:
            keys = (keys,) if isinstance(keys, str) else tuple(keys)
            unknown = [k for k in keys if k not in self.RESULT_KEYS]
            if unknown:
                raise ValueError("get_depth_batch(keys=...): unknown result entries %s (known: %s)" % (unknown, list(self.RESULT_KEYS)))
            return_unrectify_depth = "unrectify_depth" in keys or "undistort_img1" in keys
        want = (lambda k: True) if keys is None else (lambda k: k in keys)
        i1, was_np = self._to_dev(imgs1)
        i2, _ = self._to_dev(imgs2)
        # (the two cameras of a rig may differ in resolution -- each is rectified through its own maps,
        # stereo_camera.py:159-165,216-228 -- so only the pair count and the channel count have to agree)
        if i1.dim() != 4 or i2.dim() != 4 or i1.shape[0] != i2.shape[0] or i1.shape[3] != i2.shape[3]:
            raise ValueError("imgs1 / imgs2 must be (n, h, w, c) stacks of the same number of pairs and channels")
        if self._native_sgbm() is None:
            raise ValueError("get_depth_batch needs the SemiGlobalBlockMatching plugin (with its own __call__)")
        rectify_img1, rectify_img2 = self.rectify(i1, i2)
        sm = self._sgbm_full_res(rectify_img1.shape[1:3])
        tb = self._tables(i1.device)
        if sm is not None:
            disparity, rectify_depth = self._fused_depth(sm, sm.stereo_sgbm.compute(rectify_img1, rectify_img2), tb)
        else:  # the downsizing matcher: batched resize, match, resize back + post-processing in one pass
            sm = self.stereo_matching
            sdisp16, _ = sm.compute_disp16(rectify_img1, rectify_img2, batched=True)
            disparity, rectify_depth = self._fused_depth(sm, sdisp16, tb, rectify_img1.shape[1:3])
        result = dict(rectify_img1=rectify_img1, rectify_depth=rectify_depth, disparity=disparity,
                      rectify_img2=rectify_img2)
        if return_unrectify_depth:
            if want("unrectify_depth"):
                result.update(unrectify_depth=self.unrectify_depth(rectify_depth))
            if want("undistort_img1"):
                result.update(undistort_img1=self.undistort_img(i1))
        result = {k: v for k, v in result.items() if want(k)}
        if was_np and result:
            # (to_host_list: with a single asked key, to_host's bare ndarray would be zipped row by row)
            result = dict(zip(result, hostio.to_host_list(*result.values())))
            sm.stereo_sgbm.status()  # to_host synchronised: surface device-side timeouts at no extra cost
        return result
