"""Seeded fuzz of the remap kernels against the CPU oracle (run by hand on a GPU box: python tools/gpu_fuzz_remap.py N).
Random source sizes down to 1x1 (narrower than the 8x8 window), odd base addresses and padded pitches, maps that
wander far outside the image, magnify, fold back and hit exact half-phase coordinates; all three interpolations,
both channel counts, x-shifts."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle
from calibrating_amd import _native, imgproc

oracle.build()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad = 0
for case in range(n):
    cn = int(rng.choice([1, 3]))
    sh, sw = int(rng.choice([1, 2, 3, 7, 8, 9, 20, 61])), int(rng.choice([1, 2, 5, 8, 9, 17, 64, 130]))
    dh, dw = int(rng.integers(1, 40)), int(rng.choice([1, 3, 63, 64, 65, 255, 256, 257, 300]))
    img = rng.integers(0, 256, (sh, sw, cn), dtype=np.uint8)
    pad, lead = int(rng.integers(0, 7)), int(rng.integers(0, 9))
    pitch = sw * cn + pad
    buf = np.full(lead + sh * pitch + 64, 255, np.uint8)
    np.lib.stride_tricks.as_strided(buf[lead:], (sh, sw * cn), (pitch, 1))[:] = img.reshape(sh, sw * cn)
    kind = case % 4
    yy, xx = np.mgrid[:dh, :dw].astype(np.float32)
    if kind == 0:    # smooth warp crossing every edge
        mapx = xx * ((sw + 12) / dw) - 6 + rng.uniform(-0.5, 0.5, (dh, dw))
        mapy = yy * ((sh + 12) / dh) - 6 + rng.uniform(-0.5, 0.5, (dh, dw))
    elif kind == 1:  # pure noise, mostly outside
        mapx = rng.uniform(-40, sw + 40, (dh, dw))
        mapy = rng.uniform(-40, sh + 40, (dh, dw))
    elif kind == 2:  # strong magnification on the 1/32 grid (exact phases, half-way rounding cases)
        mapx = np.round(xx * 0.07 * 64) / 64 + rng.integers(-2, max(sw, 2))
        mapy = np.round(yy * 0.11 * 64) / 64 + rng.integers(-2, max(sh, 2))
    else:            # fold-over with huge excursions (the short-range clamp of the cell index)
        mapx = np.where(rng.random((dh, dw)) < 0.1, rng.choice([-1e6, 1e6, 40000.3, -40000.7]), (dw - xx) * sw / dw)
        mapy = np.where(rng.random((dh, dw)) < 0.1, rng.choice([-1e6, 1e6, 32767.5, -32768.5]), (dh - yy) * sh / dh)
    mapx, mapy = mapx.astype(np.float32), mapy.astype(np.float32)
    d_buf, mx, my = torch.from_numpy(buf).cuda(), torch.from_numpy(mapx).cuda(), torch.from_numpy(mapy).cuda()
    for interp in (imgproc.INTER_LANCZOS4, imgproc.INTER_LINEAR, imgproc.INTER_NEAREST):
        shift = int(rng.choice([0, 0, 3, -2]))
        out = torch.full((dh, dw, cn), 77, dtype=torch.uint8, device="cuda")
        rc = _native.lib().camd_remap_u8(d_buf.data_ptr() + lead, sw, sh, cn, pitch, sh * pitch, mx.data_ptr(), my.data_ptr(),
                                         out.data_ptr(), dw, dh, dw * cn, dh * dw * cn, interp, shift, 1, _native.current_stream())
        _native.check(rc, "remap")
        ref = oracle.remap_u8(img if cn > 1 else img[..., 0], mapx, mapy, interp).reshape(dh, dw, cn)
        if shift:  # stereo_camera.py:230-240: translate the remapped image, zero fill
            sref = np.zeros_like(ref)
            if shift > 0: sref[:, shift:] = ref[:, :-shift] if shift < dw else 0
            else: sref[:, :shift] = ref[:, -shift:] if -shift < dw else 0
            ref = sref
        if not np.array_equal(out.cpu().numpy(), ref):
            bad += 1
            d = np.argwhere(out.cpu().numpy() != ref)
            print("MISMATCH case", case, dict(cn=cn, src=(sh, sw), dst=(dh, dw), pad=pad, lead=lead, kind=kind, interp=interp, shift=shift),
                  len(d), "px, first", d[0], flush=True)
print("cases", n, "mismatches", bad)
