"""calibrating_amd -- MI355X-native stereo depth path of DIYer22/calibrating.

Drop-in for the ``Stereo.get_depth`` hot path of the reference behind its own plugin surface
(calibrating/__init__.py:3-14 exports these names): ``MetaStereoMatching``,
``SemiGlobalBlockMatching``, ``Stereo``, ``Cam``.  The per-pair work runs in hand-written gfx950
kernels (calibrating_amd/csrc, C ABI in include/calibrating_amd.h); importing the package needs no
GPU, computing does.
"""
from .__info__ import __version__
from .camera import Cam
from .sgbm import MODE_HH, MODE_HH4, MODE_SGBM, MODE_SGBM_3WAY, StereoSGBM, StereoSGBM_create
from .stereo_matching import MetaStereoMatching, SemiGlobalBlockMatching
from .stereo_camera import Stereo

__all__ = ["Cam", "Stereo", "MetaStereoMatching", "SemiGlobalBlockMatching", "StereoSGBM",
           "StereoSGBM_create", "MODE_SGBM", "MODE_HH", "MODE_SGBM_3WAY", "MODE_HH4", "__version__"]
