"""Bounded, seeded slices of the three fuzzers (tests/fuzzers.py) inside the driver's GPU suite (-m gpu): the same
generators tools/gpu_fuzz*.py run by the thousand by hand.  Each asserts zero mismatches AND that the slice reached
the branches it exists for (so a silently skipped branch fails the test instead of passing it)."""
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import fuzzers  # noqa: E402


def _run(fn, n, seed, need):
    lines = []
    res = fn(n, seed, log=lambda *a: lines.append(" ".join(str(x) for x in a)))
    fuzzers.report(res)
    assert not res["mismatches"], "\n".join(lines[:10])
    for branch, at_least in need.items():
        assert res["branches"].get(branch, 0) >= at_least, (branch, res["branches"])
    return res


@pytest.mark.parametrize("seed", [401, 402, 403])
def test_fuzz_sgbm_slice(oracle, seed):
    _run(fuzzers.fuzz_sgbm, 100, seed, dict(mode0=10, mode1=10, mode2=5, mode3=10, cost1=40, cost2=40, batched=15,
                                            left_u16_regime=1))


@pytest.mark.parametrize("seed", [411, 412])
def test_fuzz_remap_slice(oracle, seed):
    _run(fuzzers.fuzz_remap, 100, seed, dict(lanczos4=100, linear=100, nearest=100, shifted=80, fold=25))


@pytest.mark.parametrize("seed", [421, 422])
def test_fuzz_pipeline_slice(oracle, seed):
    res = _run(fuzzers.fuzz_pipeline, 30, seed, dict(downsizing=8, full_resolution=5, hetero_rig=4, translated=8,
                                                     over_25pct_valid_depth=6))
    assert not res["within_tolerance_but_not_bit_identical"], res["within_tolerance_but_not_bit_identical"]
