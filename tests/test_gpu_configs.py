"""Parity at the BASELINE configs that are not the headline (-m gpu): C4 (3840x2160, D=256) on a full-width strip
the oracle finishes in seconds plus size-independent properties at full size, and C5 (640x480, D=64, LR check and
speckle filter on) end to end through get_depth."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import calibrating_amd as ca  # noqa: E402
from calibrating_amd import synthetic  # noqa: E402


@pytest.mark.parametrize("mode", [0, 1])
def test_c4_4k_d256_strip_vs_oracle(oracle, mode):
    P = dict(minDisparity=0, numDisparities=256, blockSize=5, P1=200, P2=800, disp12MaxDiff=1, uniquenessRatio=10, mode=mode)
    left, right = synthetic.rectified_pair(seed=7, H=64, W=3840, D=256, cn=1)
    got = ca.StereoSGBM_create(**P).compute(left, right)
    assert np.abs(got.astype(int) - oracle.sgbm_compute(left, right, **P)).max() == 0


def test_c4_4k_d256_full_size_properties():
    P = dict(minDisparity=0, numDisparities=256, blockSize=5, P1=200, P2=800, disp12MaxDiff=1, uniquenessRatio=10)
    dev = torch.device("cuda", 0)
    L, R = synthetic.rectified_batch_torch(7, 2, 2160, 3840, 256, 1, dev)
    m = ca.StereoSGBM_create(**P)
    m.set_option("path", 2)
    two = m.compute(L, R)                      # band passes
    # (a 4K / D=256 pair is 8 units of work: AUTO would send the single pair down the band passes as well, so the
    # second aggregation path is forced -- one line scan per direction, a different kernel and a different order)
    m.set_option("path", 1)
    one = m.compute(L[:1], R[:1])              # sequential line scans
    assert torch.equal(two[0], one[0])         # the two aggregation paths agree at full size
    m8 = ca.StereoSGBM_create(**dict(P, mode=1))
    m8.set_option("path", 2)
    hh = m8.compute(L[:1], R[:1])
    m8.set_option("path", 1)
    assert torch.equal(hh, m8.compute(L[:1], R[:1]))  # ... and in the 8-path mode
    assert (two[:, :, :256] == -16).all()      # the left band has no match
    assert (two >= 0).float().mean() > 0.6


def test_c5_vga_get_depth_vs_oracle(oracle):
    W, H = 640, 480
    stereo = ca.Stereo.load(synthetic.rig(W, H))
    cfg = dict(max_size=W, minDisparity=0, numDisparities=64, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1,
               uniquenessRatio=10, speckleWindowSize=100, speckleRange=2)
    stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(cfg), max_depth=3.5)
    i1, i2 = synthetic.scene_pair(9, W, H, 3)
    res = stereo.get_depth(i1, i2)
    r1 = oracle.remap_u8(i1, *stereo.undistort_rectify_map1, oracle.INTER_LANCZOS4)
    r2 = oracle.remap_u8(i2, *stereo.undistort_rectify_map2, oracle.INTER_LANCZOS4)
    s = stereo.min_disparity
    r2[:, s:] = r2[:, :-s].copy()
    r2[:, :s] = 0
    assert np.array_equal(res["rectify_img1"], r1) and np.array_equal(res["rectify_img2"], r2)
    disp16 = oracle.sgbm_compute(r1, r2, **{k: v for k, v in cfg.items() if k != "max_size"})
    disparity, depth = oracle.disp_to_depth(disp16, stereo.rectify_valid_mask1, 0, s, True,
                                            1.0 * stereo.baseline * stereo.K[0, 0], 3.5)
    assert np.array_equal(res["disparity"], disparity)
    assert np.abs(res["rectify_depth"] - depth).max() <= 1e-4


def test_c3_sharding_invariance():
    """Config C3 (512 pairs sharded over 8 GPUs) at reduced size: a pair's disparity does not depend on which other
    pairs share its launch, so rank r computing shard [lo, hi) of the global list (parallel_pairs.shard_range) gives
    exactly the rows of the one-launch result -- the property the no-collective data path rests on."""
    from calibrating_amd.parallel_pairs import owner_of, shard_range
    n, world = 20, 8
    dev = torch.device("cuda", 0)
    P = dict(minDisparity=0, numDisparities=128, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1, uniquenessRatio=10)
    L, R = synthetic.rectified_batch_torch(1234, n, 96, 640, 128, 3, dev)
    whole = ca.StereoSGBM_create(**P).compute(L, R)
    covered = []
    for rank in range(world):
        lo, hi = shard_range(n, world, rank)
        covered += list(range(lo, hi))
        assert all(owner_of(i, n, world) == rank for i in range(lo, hi))
        if hi > lo:
            part = ca.StereoSGBM_create(**P).compute(L[lo:hi], R[lo:hi])
            assert torch.equal(part, whole[lo:hi]), rank
    assert covered == list(range(n))
    assert [shard_range(512, 8, r) for r in range(8)] == [(64 * r, 64 * r + 64) for r in range(8)]


def test_rccl_path_with_one_forced_rank():
    """bench.py's distributed branch on the GPU under torch.distributed.run with ONE rank: init_process_group("nccl")
    (= RCCL), table broadcast, Stereo.from_bundle, barrier-bracketed timing, all_gather of the results and the
    cross-rank checksum agreement -- everything the multi-GPU run does except having a second device."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CAMD_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1",
           "--batch", "8", "--width", "640", "--height", "360", "--no-also", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["rccl"]["backend"].startswith("nccl") and d["rccl"]["ranks_agree"] is True
    assert d["rccl"]["table_bytes"] == 6 * 4 * 640 * 360 + 640 * 360 + 64 * 8  # 6 maps + mask + 64 doubles
    assert d["value"] > 0 and d["rccl"]["per_rank"][0]["pairs"] == 16


def test_rig_from_a_table_bundle_alone_equals_the_rig_from_the_record():
    """What a worker rank of a multi-GPU job holds: ``Stereo.from_bundle`` of the broadcast tensors (6 maps + mask +
    64 doubles), never the rig record.  Its get_depth / get_depth_batch -- incl. undistort_img1, whose maps it builds
    from cam1.K / cam1.D of the parameter block -- must return the record-built rig's bits."""
    W, H = 448, 336
    rec = synthetic.rig(W, H)
    dev = torch.device("cuda", 0)
    cfg = dict(max_size=W, minDisparity=0, numDisparities=64, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1,
               uniquenessRatio=10, speckleWindowSize=60, speckleRange=2)
    full = ca.Stereo.load(rec)
    bundle = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in full.table_bundle().items()}
    assert sum(t.numel() * t.element_size() for t in bundle.values()) == W * H * 25 + 512
    worker = ca.Stereo.from_bundle(bundle, dev)
    for st in (full, worker):
        st.set_stereo_matching(ca.SemiGlobalBlockMatching(cfg), max_depth=3.5)
    img1, img2, _ = synthetic.render_plane_pair(rec, (0.2, 0.1, 1.0), 2.0)
    want, got = full.get_depth(img1, img2), worker.get_depth(img1, img2)
    assert sorted(want) == sorted(got)
    for k in want:
        assert np.array_equal(want[k], got[k]), k
    assert (got["unrectify_depth"] > 0).mean() > 0.3
    gb = worker.get_depth_batch(np.stack([img1, img1]), np.stack([img2, img2]))
    assert all(np.array_equal(gb[k][1], want[k]) for k in want)
    with pytest.raises(ValueError, match="table bundle"):
        worker.dump()  # no R, no camera 2 intrinsics: nothing to write, and it says so
    # the device key has one spelling: a bundle installed as torch.device("cuda") serves images on "cuda:0" ...
    w2 = ca.Stereo.from_bundle(bundle, torch.device("cuda"))
    w2.set_stereo_matching(ca.SemiGlobalBlockMatching(cfg), max_depth=3.5)
    got2 = w2.get_depth(img1, img2, keys=("rectify_img2", "unrectify_depth"))
    assert all(np.array_equal(got2[k], want[k]) for k in got2)
    # ... and a device the bundle was never installed on is an error, not a rebuild from camera 2's NaN intrinsics
    w2._dev = {k: v for k, v in w2._dev.items() if k.startswith("unrect:")}
    with pytest.raises(RuntimeError, match="table bundle"):
        w2.get_depth(img1, img2)
