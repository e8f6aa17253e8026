"""``boxx.resize`` as the reference's matcher uses it (stereo_matching.py:62,66), on the GPU.

``resize(img, ratio)`` with ``ratio == 1`` and ``resize(img, (h, w))`` to the current size return the
input unchanged (the cases reached when ``cfg["max_size"] >= max(h, w)``); anything else runs the
cv2.resize(..., INTER_LINEAR) kernels of csrc/resize.hip (uint8 HWC images, float32 HW maps).
torch CUDA tensors in, tensors out.
"""
import numbers

from . import _native


def target_hw(shape_hw, arg):
    if isinstance(arg, numbers.Number):
        return (int(round(shape_hw[0] * arg)), int(round(shape_hw[1] * arg)))
    if hasattr(arg, "shape"):
        return tuple(arg.shape[:2])
    return (int(arg[0]), int(arg[1]))


def resize(img, arg, batched=False):
    """``batched``: ``img`` carries a leading image index, (n, h, w[, c]); one launch resizes all n images."""
    import torch
    if isinstance(arg, numbers.Number) and arg == 1:
        return img
    lead = 1 if batched else 0
    hw = target_hw(tuple(img.shape[lead:lead + 2]), arg)
    if hw == tuple(img.shape[lead:lead + 2]):
        return img
    if not isinstance(img, torch.Tensor) or not img.is_cuda:
        raise ValueError("resize expects a torch CUDA tensor")
    img = img.contiguous()
    n = img.shape[0] if batched else 1
    sh, sw = img.shape[lead:lead + 2]
    dh, dw = hw
    head = tuple(img.shape[:lead])
    with torch.cuda.device(img.device):
        if img.dtype == torch.uint8:
            cn = 1 if img.dim() == lead + 2 else img.shape[lead + 2]
            out = torch.empty(head + (dh, dw) + tuple(img.shape[lead + 2:]), dtype=torch.uint8, device=img.device)
            rc = _native.lib().camd_resize_linear_u8(img.data_ptr(), sw, sh, cn, out.data_ptr(), dw, dh, n,
                                                     _native.current_stream())
        elif img.dtype == torch.float32 and img.dim() == lead + 2:
            out = torch.empty(head + (dh, dw), dtype=torch.float32, device=img.device)
            rc = _native.lib().camd_resize_linear_f32(img.data_ptr(), sw, sh, out.data_ptr(), dw, dh, n,
                                                      _native.current_stream())
        else:
            raise ValueError("resize: unsupported dtype/shape %s %s" % (img.dtype, tuple(img.shape)))
    _native.check(rc, "resize")
    return out
