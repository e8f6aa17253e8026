"""StereoMatching plugin surface -- same names and contract as
/root/reference/calibrating/stereo_matching.py:10-70, with the SGBM plugin running on the MI355X.

``MetaStereoMatching``: ``__init__(cfg)``, ``__call__(img1, img2)`` with rectified RGB uint8
``(h, w, 3)`` images; returns ``disparity (h, w)`` in pixels of the input resolution or a dict
containing ``"disparity"`` (extra keys are merged into ``Stereo.get_depth``'s result,
stereo_camera.py:506-509).
"""
import numpy as np

from . import hostio, resize as _resize
from .sgbm import MODE_SGBM, StereoSGBM_create


class MetaStereoMatching:
    def __init__(self, cfg=None):
        self.cfg = cfg

    def __call__(self, img1, img2):
        # input: RGB uint8 (h, w, 3)uint8
        raise NotImplementedError()
        # output: float disparity (h, w), or dict(disparity=disparity)


class SemiGlobalBlockMatching(MetaStereoMatching):
    """cv2.StereoSGBM wrapper of the reference (stereo_matching.py:22-70) on the gfx950 kernels.

    ``cfg`` keys: ``max_size`` (reference key, default 1000: inputs are downsized so that
    max(h, w) <= max_size, :61-62) plus the StereoSGBM parameters, whose defaults are the values the
    reference hard-codes (:30-58): block 11, minDisparity 2, numDisparities 218, uniquenessRatio 5,
    speckleWindowSize 200, speckleRange 2, disp12MaxDiff 0, P1 = 8*121, P2 = 32*121, default mode.
    """

    def __init__(self, cfg=None):
        if cfg is None:
            cfg = {}
        self.cfg = cfg
        self.max_size = self.cfg.get("max_size", 1000)
        block_size = int(self.cfg.get("blockSize", 11))
        min_disp = int(self.cfg.get("minDisparity", 2))
        num_disp = int(self.cfg.get("numDisparities", 220 - 2))
        self.stereo_sgbm = StereoSGBM_create(
            minDisparity=min_disp,
            numDisparities=num_disp,
            blockSize=block_size,
            uniquenessRatio=self.cfg.get("uniquenessRatio", 5),
            speckleWindowSize=self.cfg.get("speckleWindowSize", 200),
            speckleRange=self.cfg.get("speckleRange", 2),
            disp12MaxDiff=self.cfg.get("disp12MaxDiff", 0),
            P1=self.cfg.get("P1", 8 * 1 * block_size * block_size),
            P2=self.cfg.get("P2", 32 * 1 * block_size * block_size),
            preFilterCap=self.cfg.get("preFilterCap", 0),
            mode=self.cfg.get("mode", MODE_SGBM),
        )

    def compute_disp16(self, img1, img2, batched=False):
        """Device-resident stage used by ``Stereo.get_depth``: int16 disparity*16 of the (possibly
        downsized) pair plus the width it was computed at; torch tensors in, torch tensor out.
        ``batched``: (n, h, w, c) stacks of pairs, one launch per stage."""
        lead = 1 if batched else 0
        h, w = img1.shape[lead:lead + 2]
        resize_ratio = min(self.max_size / max(h, w), 1)
        simg1, simg2 = _resize.resize(img1, resize_ratio, batched), _resize.resize(img2, resize_ratio, batched)
        return self.stereo_sgbm.compute(simg1, simg2), simg1.shape[lead + 1]

    def _disparity_from_disp16(self, sdisp16, hw, sw, batched=False):
        # stereo_matching.py:63-69: float32, clip at 0, below minDisparity -> 0, /16, back to the input size, x w/sw
        import torch
        sdisparity = sdisp16.to(torch.float32).clamp_(min=0)
        sdisparity[sdisparity < self.stereo_sgbm.getMinDisparity() * 16] = 0
        up = _resize.resize(sdisparity / 16.0, hw, batched) * hw[1]  # (/16 is exact in any form)
        # NumPy's `x * w / sw` divides; torch turns a division by a Python number into a multiplication by its
        # reciprocal (one ulp off now and then), a tensor divisor gets the IEEE division
        return up / torch.full((), float(sw), dtype=torch.float32, device=up.device)

    def call_batch(self, imgs1, imgs2):
        """``__call__`` for (n, h, w, 3) CUDA stacks of rectified pairs -> (n, h, w) float32 disparities; pair i equals
        ``self(imgs1[i], imgs2[i])``.  Not in the reference (one pair per call); used by ``Stereo.get_depth_batch``."""
        hw = tuple(imgs1.shape[1:3])
        sdisp16, sw = self.compute_disp16(imgs1, imgs2, batched=True)
        return self._disparity_from_disp16(sdisp16, hw, sw, batched=True)

    def __call__(self, img1, img2):
        import torch
        is_np = isinstance(img1, np.ndarray)
        if is_np:
            img1, img2 = torch.from_numpy(np.ascontiguousarray(img1)).cuda(), \
                torch.from_numpy(np.ascontiguousarray(img2)).cuda()
        sdisp16, sw = self.compute_disp16(img1, img2)
        disparity = self._disparity_from_disp16(sdisp16, tuple(img1.shape[:2]), sw)
        return hostio.to_host(disparity) if is_np else disparity
