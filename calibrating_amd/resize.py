"""``boxx.resize`` as the reference's matcher uses it (stereo_matching.py:62,66).

``resize(img, ratio)`` with ``ratio == 1`` and ``resize(img, (h, w))`` to the current size return
the input unchanged -- the only cases reached when ``cfg["max_size"] >= max(h, w)``.  A real
down-/up-scale (cv2.resize INTER_LINEAR) is the "next" row n1 of SURVEY.md section 8f and is not
built yet: it raises instead of silently substituting another interpolation.
"""
import numbers


def target_hw(shape_hw, arg):
    if isinstance(arg, numbers.Number):
        return (int(round(shape_hw[0] * arg)), int(round(shape_hw[1] * arg)))
    if hasattr(arg, "shape"):
        return tuple(arg.shape[:2])
    return (int(arg[0]), int(arg[1]))


def resize(img, arg):
    if isinstance(arg, numbers.Number) and arg == 1:
        return img
    hw = target_hw(tuple(img.shape[:2]), arg)
    if hw == tuple(img.shape[:2]):
        return img
    raise NotImplementedError(
        "calibrating_amd: resizing %s -> %s is not implemented on the GPU path yet; construct the matcher "
        "with cfg['max_size'] >= max(h, w) (SURVEY.md F8)" % (tuple(img.shape[:2]), hw))
