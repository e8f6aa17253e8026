"""Seeded fuzzers of the HIP path against the CPU oracle.  One implementation, two users: bounded slices run in the
driver's GPU suite (tests/test_gpu_fuzz.py), and tools/gpu_fuzz*.py run thousands of cases by hand and keep the log.

Every run returns a dict {fuzzer, seed, cases, branches: {name: count}, mismatches: [...]}; ``stamp()`` adds what
identifies the code that was tested (the git SHA handed in through CAMD_GIT_SHA -- .git does not travel to the GPU box --
and a hash of the shipped library)."""
import hashlib
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stamp():
    so = os.path.join(ROOT, "calibrating_amd", "lib", "libcalibrating_amd.so")
    h = hashlib.sha256(open(so, "rb").read()).hexdigest()[:16] if os.path.exists(so) else None
    return dict(git_sha=os.environ.get("CAMD_GIT_SHA", "unknown"), lib_sha256_16=h)


def _count(d, key):
    d[key] = d.get(key, 0) + 1


# ------------------------------------------------------------------------------------------------------------------
def fuzz_sgbm(n, seed=77, log=print):
    """SGBM against oracle.sgbm_compute: random sizes (incl. widths that leave partial strips / single columns), channel
    counts, disparity ranges, block sizes up to 11, penalties up to the library's P2 limit, preFilterCap up to 63 (the
    saturating regime), all four modes, batches, both cost-kernel paths; every fifth case is built to drift out of the
    int16 regime (synthetic.drift_pair) and must still match bit for bit."""
    import calibrating_amd as ca
    import oracle
    from calibrating_amd import synthetic
    rng = np.random.default_rng(seed)
    br, bad = {}, []
    for case in range(n):
        cn = int(rng.choice([1, 3]))
        D = int(rng.choice([8, 16, 24, 32, 48, 50, 64, 80, 96, 128, 144, 160, 176, 192, 200, 218, 224, 256, 300]))  # every lane shape
        bs = int(rng.choice([0, 1, 3, 5, 7, 9, 11]))
        minD = int(rng.integers(-9, 10))
        mode = int(rng.choice([0, 1, 2, 3]))
        force_band = bool(rng.integers(0, 2))  # (AUTO sends small 3WAY calls down the scan path)
        b = max(bs, 1)
        W = D + abs(minD) + int(rng.integers(b // 2 + 2, 140))
        H = int(rng.integers(3, 90)) if mode != 2 else int(rng.integers(40, 110))
        P1 = int(rng.integers(1, 8 * cn * b * b + 2))
        P2 = P1 + int(rng.integers(1, 32 * cn * b * b + 2))
        if rng.random() < 0.3:  # up to the library's limit (cv2's rule of thumb 32*cn*b*b is 21600 at block 15 RGB)
            P2 = int(rng.integers(max(P1 + 1, 12000), 24001))
        p = dict(minDisparity=minD, numDisparities=D, blockSize=bs, P1=P1, P2=P2, disp12MaxDiff=int(rng.integers(-1, 4)),
                 uniquenessRatio=int(rng.integers(0, 30)), preFilterCap=int(rng.choice([0, 15, 31, 63])),
                 speckleWindowSize=int(rng.choice([0, 0, 40])), speckleRange=int(rng.integers(1, 4)), mode=mode)
        kind = case % 5
        if kind == 4:  # saturation in the upper part, none below: C drifts under P2 / negative -> the exact int path
            left, right = synthetic.drift_pair(H, W, cn, split=float(rng.uniform(0.2, 0.8)), seed=seed * 100003 + case)
            if cn == 1:
                left, right = np.ascontiguousarray(left), np.ascontiguousarray(right)
        elif kind == 0:
            r2 = np.random.default_rng(seed * 100003 + case)
            shape = (H, W) if cn == 1 else (H, W, cn)
            left, right = r2.integers(0, 256, shape, dtype=np.uint8), r2.integers(0, 256, shape, dtype=np.uint8)
        elif kind == 1:  # opposite sawtooth ramps: drives the window sums into saturation when preFilterCap is raised
            x, y = np.arange(W)[None, :], np.arange(H)[:, None]
            ramp = ((x * 16 + y * 40) % 256).astype(np.uint8)
            left = ramp if cn == 1 else ramp[..., None].repeat(3, 2)
            right = 255 - left
        else:
            left, right = synthetic.rectified_pair(seed=seed * 100003 + case, H=H, W=W, D=max(min(D, W // 2), 8), cn=cn)
        try:
            want = oracle.sgbm_compute(left, right, **p)
        except ValueError:
            _count(br, "refused_by_oracle")
            try:  # the product must refuse too
                ca.StereoSGBM_create(**p).compute(left, right)
                bad.append(dict(case=case, why="oracle refuses, product does not", shape=(H, W, cn), params=p))
            except ValueError:
                pass
            continue
        _count(br, "mode%d" % mode)
        _count(br, ("gray" if cn == 1 else "rgb") + ("_drift_input" if kind == 4 else ""))
        try:
            for cost in ((1, 2) if mode != 2 and bs <= 11 else (0,)):
                m = ca.StereoSGBM_create(**p)
                m.set_option("cost", cost)
                if mode == 2 and force_band:
                    m.set_option("path", 2)
                nb = int(rng.choice([1, 1, 3]))
                got = m.compute(np.stack([left] * nb), np.stack([right] * nb)) if nb > 1 else m.compute(left, right)[None]
                _count(br, "cost%d" % cost)
                _count(br, "Dp%d" % m.geometry()["Dp"])  # which lane shape (16 lanes x Dp/32 registers from 96 on)
                if nb > 1:
                    _count(br, "batched")
                if cost != 2 and nb == 1 and mode != 2:
                    # how many cases leave the packed-u16 regime (C < P2 after an int16 overflow): the exact int path
                    P2n = max(p["P2"] if p["P2"] > 0 else 5, (p["P1"] if p["P1"] > 0 else 2) + 1)
                    if int(m.debug_volume("C").min()) < P2n:
                        _count(br, "left_u16_regime")
                for i in range(nb):
                    if not np.array_equal(got[i], want):
                        bad.append(dict(case=case, cost=cost, batch="%d/%d" % (i, nb), shape=(H, W, cn), params=p,
                                        pixels=int((got[i] != want).sum())))
                        log("MISMATCH", bad[-1])
                        break
        except ValueError as e:
            bad.append(dict(case=case, why="refused by the product only: %s" % e, shape=(H, W, cn), params=p))
            log("MISMATCH", bad[-1])
    return dict(fuzzer="sgbm", seed=seed, cases=n, branches=br, mismatches=bad)


# ------------------------------------------------------------------------------------------------------------------
def fuzz_remap(n, seed=5, log=print):
    """The remap kernels through the raw C ABI against oracle.remap_u8: source sizes down to 1x1 (narrower than the 8x8
    window), odd base addresses and padded pitches, maps that wander far outside the image, magnify, fold back and hit
    exact half-phase coordinates; all three interpolations, both channel counts, x-shifts."""
    import torch
    import oracle
    from calibrating_amd import _native, imgproc
    rng = np.random.default_rng(seed)
    br, bad = {}, []
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
        _count(br, ("warp", "noise", "magnify", "fold")[kind])
        for interp, iname in ((imgproc.INTER_LANCZOS4, "lanczos4"), (imgproc.INTER_LINEAR, "linear"),
                              (imgproc.INTER_NEAREST, "nearest")):
            shift = int(rng.choice([0, 0, 3, -2]))
            out = torch.full((dh, dw, cn), 77, dtype=torch.uint8, device="cuda")
            rc = _native.lib().camd_remap_u8(d_buf.data_ptr() + lead, sw, sh, cn, pitch, sh * pitch, mx.data_ptr(),
                                             my.data_ptr(), out.data_ptr(), dw, dh, dw * cn, dh * dw * cn, interp, shift, 1,
                                             _native.current_stream())
            _native.check(rc, "remap")
            ref = oracle.remap_u8(img if cn > 1 else img[..., 0], mapx, mapy, interp).reshape(dh, dw, cn)
            if shift:  # stereo_camera.py:230-240: translate the remapped image, zero fill
                sref = np.zeros_like(ref)
                if shift > 0:
                    sref[:, shift:] = ref[:, :-shift] if shift < dw else 0
                else:
                    sref[:, :shift] = ref[:, -shift:] if -shift < dw else 0
                ref = sref
                _count(br, "shifted")
            _count(br, iname)
            got = out.cpu().numpy()
            if not np.array_equal(got, ref):
                d = np.argwhere(got != ref)
                bad.append(dict(case=case, cn=cn, src=(sh, sw), dst=(dh, dw), pad=pad, lead=lead, kind=kind, interp=iname,
                                shift=shift, pixels=len(d), first=d[0].tolist()))
                log("MISMATCH", bad[-1])
    return dict(fuzzer="remap", seed=seed, cases=n, branches=br, mismatches=bad)


# ------------------------------------------------------------------------------------------------------------------
def fuzz_pipeline(n, seed=11, log=print):
    """The whole Stereo.get_depth against tests/oracle_pipeline.oracle_get_depth: random rigs (rotation, baseline,
    distortion, focal lengths; every fourth with a second camera of ANOTHER resolution), xy_target / K_target,
    max_depth (and with it the min_disparity translation), matcher parameters incl. max_size downsizing; and the batched
    form against the same oracle result."""
    import calibrating_amd as ca
    import oracle
    from calibrating_amd import synthetic
    from oracle_pipeline import compare, oracle_get_depth
    rng = np.random.default_rng(seed)
    br, bad, inexact_total = {}, [], {}
    for case in range(n):
        W, H = int(rng.choice([160, 200, 256, 320])), int(rng.choice([96, 120, 150, 200]))
        hetero = case % 4 == 3
        W2, H2 = (int(rng.choice([128, 240, 333])), int(rng.choice([100, 144, 180]))) if hetero else (W, H)
        f = W * rng.uniform(0.6, 1.1)
        f2 = f * W2 / W
        K1 = [[f, 0, W / 2 + rng.uniform(-6, 6)], [0, f * rng.uniform(0.98, 1.02), H / 2 + rng.uniform(-5, 5)], [0, 0, 1]]
        K2 = [[f2 * rng.uniform(0.97, 1.03), 0, W2 / 2 + rng.uniform(-6, 6)],
              [0, f2 * rng.uniform(0.97, 1.03), H2 / 2 + rng.uniform(-5, 5)], [0, 0, 1]]
        dscale = [0.2, 0.08, 2e-3, 2e-3, 0.02]
        rig = dict(R=synthetic.rodrigues(rng.uniform(-0.04, 0.04, 3)).tolist(),
                   t=[[-rng.uniform(0.05, 0.3)], [rng.uniform(-0.01, 0.01)], [rng.uniform(-0.01, 0.01)]],
                   cam1=dict(K=K1, D=[(rng.uniform(-1, 1, 5) * dscale).tolist()], xy=[W, H], name="a"),
                   cam2=dict(K=K2, D=[(rng.uniform(-1, 1, 5) * dscale).tolist()], xy=[W2, H2], name="b"))
        xy_target = [None, None, 0.75, (W + 16, H - 8)][int(rng.integers(0, 4))]
        K_target = float(rng.choice([1, 1, 0.8, 1.2]))
        try:
            stereo = ca.Stereo(ca.Cam.load(rig["cam1"]), ca.Cam.load(rig["cam2"]), xy_target=xy_target, K_target=K_target,
                               R=np.array(rig["R"]), t=np.array(rig["t"]))
        except Exception as e:  # (a degenerate random rig)
            _count(br, "rig_refused")
            log("case", case, "rig refused:", str(e)[:80])
            continue
        Wt, Ht = stereo.xy
        D = int(rng.choice([16, 32, 48, 64]))
        bs = int(rng.choice([3, 5, 7, 11]))
        if Wt - D < 24:
            _count(br, "too_narrow")
            continue
        big = max(Wt, Ht)
        cfg = dict(max_size=int(rng.choice([big, big, int(big * 0.7), big - 1, big // 2])), minDisparity=int(rng.integers(0, 4)),
                   numDisparities=D, blockSize=bs, P1=8 * 3 * bs * bs, P2=32 * 3 * bs * bs,
                   disp12MaxDiff=int(rng.integers(0, 3)), uniquenessRatio=int(rng.integers(0, 15)),
                   speckleWindowSize=int(rng.choice([0, 60])), speckleRange=2, mode=int(rng.choice([0, 1, 3])))
        max_depth = [None, 3.0, 8.0][int(rng.integers(0, 3))]
        stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(cfg), max_depth=max_depth)
        if case % 2:
            img1, img2 = synthetic.render_plane_pair(rig, (rng.uniform(-0.3, 0.3), rng.uniform(-0.2, 0.2), 1.0),
                                                     float(rng.uniform(1.0, 2.5)), seed=case)[:2]
        else:
            img1 = synthetic.scene_pair(seed * 100003 + case, W, H, 3)[0]
            img2 = synthetic.scene_pair(seed * 100003 + case, W2, H2, 3)[1]
        downsized = cfg["max_size"] < big
        _count(br, "downsizing" if downsized else "full_resolution")
        _count(br, "hetero_rig" if hetero else "same_size_rig")
        _count(br, "translated" if max_depth else "untranslated")
        ref = oracle_get_depth(oracle, stereo, cfg, img1, img2)
        if (ref["rectify_depth"] > 0).mean() > 0.25:
            _count(br, "over_25pct_valid_depth")
        problems, inexact = compare(stereo.get_depth(img1, img2), ref)
        for k in inexact:
            _count(inexact_total, k)
        # batched form against the same oracle result (second slot; the first holds another pair)
        o1 = synthetic.scene_pair(case + 1000, W, H, 3)[0]
        o2 = synthetic.scene_pair(case + 1000, W2, H2, 3)[1]
        gb = stereo.get_depth_batch(np.stack([o1, img1]), np.stack([o2, img2]))
        pb, ib = compare({k: v[1] for k, v in gb.items()}, ref)
        problems += ["batch:" + k for k in pb]
        if problems:
            bad.append(dict(case=case, cam1=(W, H), cam2=(W2, H2), target=(int(Wt), int(Ht)), cfg=cfg, max_depth=max_depth,
                            problems=problems))
            log("MISMATCH", bad[-1])
    return dict(fuzzer="pipeline", seed=seed, cases=n, branches=br, mismatches=bad,
                within_tolerance_but_not_bit_identical=inexact_total)


def speckle_image(rng, kind, h, w, new_val):
    """Disparity-like int16 images for the speckle filter: smooth regions with noise and holes, few-level noise (many
    tiny components), one-pixel-wide serpentines (a long component through many strips and segments) and combs (teeth
    that only meet at the bottom: every tooth is born as its own tree and merged late)."""
    if kind == 0:
        yy, xx = np.mgrid[:h, :w]
        img = (16 * 20 + 3 * xx + 5 * yy).astype(np.int16)
        for _ in range(int(rng.integers(1, 5))):
            y0, x0 = int(rng.integers(0, h)), int(rng.integers(0, w))
            img[y0:y0 + int(rng.integers(1, h + 1)), x0:x0 + int(rng.integers(1, w + 1))] += np.int16(rng.integers(-400, 400))
        spk = rng.random((h, w)) < 0.03
        img[spk] = (rng.integers(0, 100, int(spk.sum())) * 16).astype(np.int16)
        img[rng.random((h, w)) < 0.05] = new_val
    elif kind == 1:
        img = (rng.integers(0, int(rng.integers(2, 6)), (h, w)) * 40).astype(np.int16)
        img[rng.random((h, w)) < 0.2] = new_val
    elif kind == 2:
        tr = rng.random() < 0.5  # (vertical serpentine: built lying down, then transposed)
        hh, ww = (w, h) if tr else (h, w)
        img = np.full((hh, ww), new_val, np.int16)
        step = int(rng.integers(2, 5))
        for r, y in enumerate(range(0, hh, step)):
            img[y, :] = 500
            img[y:min(y + step, hh), ww - 1 if r % 2 == 0 else 0] = 500  # connector alternately right / left
        if tr:
            img = np.ascontiguousarray(img.T)
    else:
        img = np.full((h, w), new_val, np.int16)
        pitch = int(rng.integers(2, 7))
        img[:, ::pitch] = 300
        img[h - 1 if rng.random() < 0.5 else 0, :] = 300
        if rng.random() < 0.3:
            img[(np.indices((h, w)).sum(0) % 2 == 0) & (img == new_val)] = 1000
    return img


def fuzz_speckle(n, seed=9, log=print):
    """camd_filter_speckles_s16 against oracle.filter_speckles_s16 (= cv2.filterSpeckles' flood fill): sizes around the
    64-column segments and 16-row strips of the kernels, batches, thresholds from 0 to "everything", and -- through one
    StereoSGBM handle used for several different pairs -- the handle-owned workspace that is never re-initialised."""
    import oracle
    from calibrating_amd import imgproc
    rng = np.random.default_rng(seed)
    br, bad = {}, []
    for case in range(n):
        kind = case % 4
        h = int(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 48, 70, 130]))
        w = int(rng.choice([1, 2, 63, 64, 65, 127, 128, 129, 200, 256, 257, 300, 520]))
        new_val = int(rng.choice([-16, 16, 0]))
        nb = int(rng.choice([1, 1, 2, 5]))
        imgs = np.stack([speckle_image(rng, kind, h, w, new_val) for _ in range(nb)])
        max_size = int(rng.choice([0, 1, 5, 40, 200, 5000, h * w]))
        max_diff = int(rng.choice([0, 16, 32, 200]))
        got = imgproc.filterSpeckles(imgs if nb > 1 else imgs[0], new_val, max_size, max_diff)
        got = got if nb > 1 else got[None]
        _count(br, ("smooth", "noise", "serpentine", "comb")[kind])
        if nb > 1:
            _count(br, "batched")
        for i in range(nb):
            want = oracle.filter_speckles_s16(imgs[i], new_val, max_size, max_diff)
            if (want != imgs[i]).any():
                _count(br, "images_with_erased_pixels")
            if not np.array_equal(got[i], want):
                bad.append(dict(case=case, kind=kind, shape=(h, w), image="%d/%d" % (i, nb), new_val=new_val,
                                max_size=max_size, max_diff=max_diff, pixels=int((got[i] != want).sum())))
                log("MISMATCH", bad[-1])
                break
    return dict(fuzzer="speckle", seed=seed, cases=n, branches=br, mismatches=bad)


def report(res, log=print):
    """One greppable summary line per run (what profiles/*_fuzz_*.log keep)."""
    st = stamp()
    log("FUZZ %s git=%s lib=%s seed=%d cases=%d mismatches=%d branches=%s%s" % (
        res["fuzzer"], st["git_sha"], st["lib_sha256_16"], res["seed"], res["cases"], len(res["mismatches"]),
        dict(sorted(res["branches"].items())),
        (" inexact=%s" % res["within_tolerance_but_not_bit_identical"])
        if "within_tolerance_but_not_bit_identical" in res else ""))
