"""Extended seeded fuzz of the speckle filter against the CPU oracle, by hand on a GPU box:
    CAMD_GIT_SHA=<sha> python tools/gpu_fuzz_speckle.py N [first_seed [n_seeds]]
N cases for each of n_seeds consecutive seeds (tests/fuzzers.fuzz_speckle holds the generator; the driver's GPU suite runs
a bounded slice of the same).  Prints one FUZZ line per seed: git SHA, library hash, seed, per-branch counts, mismatches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: F401,E402
import oracle  # noqa: E402
import fuzzers  # noqa: E402

oracle.build()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
first = int(sys.argv[2]) if len(sys.argv) > 2 else 9
total = 0
for seed in range(first, first + (int(sys.argv[3]) if len(sys.argv) > 3 else 1)):
    res = fuzzers.fuzz_speckle(n, seed, log=lambda *a: print(*a, flush=True))
    fuzzers.report(res, log=lambda *a: print(*a, flush=True))
    total += len(res["mismatches"])
sys.exit(1 if total else 0)
