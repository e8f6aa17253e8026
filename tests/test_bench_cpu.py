"""bench.py's in-run counter collection without a GPU: the parsing of rocprofv3's counter files, the gfx950 correction
(HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) KB), the per-launch averaging, and the fall-backs that keep the bench line
coming when rocprofv3 is missing or fails (the line then says so in roofline.counters)."""
import argparse
import csv
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _args():
    return argparse.Namespace(width=1920, height=1080, disparities=128, block=5, channels=3, mode="sgbm", path=0, cost=0, lib="")


def _fake_rocprof(monkeypatch, values, rc=0):
    """subprocess.run stand-in: writes what `rocprofv3 --pmc <ctrs> ... -d <dir>` would (two launches per kernel)."""
    def run(cmd, **kw):
        d = cmd[cmd.index("-d") + 1]
        ctrs = cmd[cmd.index("--pmc") + 1:cmd.index("--kernel-trace")]
        os.makedirs(os.path.join(d, "host", "123"), exist_ok=True)
        with open(os.path.join(d, "host", "123", "p_counter_collection.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
            for kernel, per in values.items():
                for launch in (0, 1):
                    for c in ctrs:
                        if c in per:
                            w.writerow([launch, kernel, c, per[c] * (1.0 if launch else 3.0)])  # average = 2 x
            w.writerow([9, "void at::native::something(int)", ctrs[0], 1e9])  # not ours: ignored
        assert kw.get("cwd") == "/tmp" and "--no-pmc" in cmd and "RANK" not in kw["env"]
        return subprocess.CompletedProcess(cmd, rc, stdout=b"", stderr=b"")
    monkeypatch.setattr(bench.subprocess, "run", run)
    monkeypatch.setattr("shutil.which", lambda name: "/opt/rocm/bin/rocprofv3")


def test_pmc_in_run_parses_and_corrects(monkeypatch):
    k1 = "void camd::k_cost<3, 5, false, 8>(unsigned char const*, int)"
    k2 = "void camd::k_band<16, 4, true, 0, false, true, false>(camd::BandArgs, camd::Geom)"
    _fake_rocprof(monkeypatch, {k1: dict(FETCH_SIZE=100.0, WRITE_SIZE=50.0, SQ_INSTS_VALU=1e6, SQ_WAVE_CYCLES=4e6, SQ_WAIT_ANY=1e6),
                                k2: dict(FETCH_SIZE=10.0, WRITE_SIZE=5.0, SQ_INSTS_VALU=2e6, SQ_WAVE_CYCLES=8e6, SQ_WAIT_ANY=6e6)})
    doc, why = bench.pmc_in_run(_args(), 64, 60.0)
    assert why is None and doc["pairs_per_launch"] == 64
    c = doc["kernels"]["camd::k_cost<3, 5, false, 8>"]
    # per-launch average of the two launches (x 2 here), then (2 * FETCH + WRITE) KB
    assert c["FETCH_SIZE"] == 200.0 and c["WRITE_SIZE"] == 100.0
    assert c["hbm_bytes_per_launch"] == (2 * 200.0 + 100.0) * 1024 and c["hbm_bytes_per_pair"] == c["hbm_bytes_per_launch"] / 64
    assert c["SQ_INSTS_VALU"] == 2e6 and abs(c["derived"]["wait_any_frac"] - 0.25) < 1e-12
    b = doc["kernels"]["camd::k_band<16, 4, true, 0, false, true, false>"]
    assert abs(b["derived"]["wait_any_frac"] - 0.75) < 1e-12
    assert not any("at::native" in k for k in doc["kernels"])


def test_pmc_in_run_falls_back_with_a_reason(monkeypatch):
    monkeypatch.setattr("shutil.which", lambda name: None)
    doc, why = bench.pmc_in_run(_args(), 64, 60.0)
    assert doc is None and "PATH" in why
    _fake_rocprof(monkeypatch, {}, rc=1)
    doc, why = bench.pmc_in_run(_args(), 64, 60.0)
    assert doc is None and "failed" in why
    _fake_rocprof(monkeypatch, {"void camd::k_x()": dict(FETCH_SIZE=1.0, WRITE_SIZE=1.0)})
    doc, why = bench.pmc_in_run(_args(), 64, 5.0)  # less than one pass's worth of budget left
    assert doc is None and "budget" in why

    def boom(cmd, **kw):
        raise subprocess.TimeoutExpired(cmd, 1)
    monkeypatch.setattr(bench.subprocess, "run", boom)
    monkeypatch.setattr("shutil.which", lambda name: "/x/rocprofv3")
    doc, why = bench.pmc_in_run(_args(), 64, 60.0)
    assert doc is None and "did not finish" in why


def test_valu_class_mix_is_committed_and_sane():
    rel, mix = bench.valu_class_mix()
    assert rel and rel.startswith("profiles/r") and rel.endswith("_isa_valu_mix.json")
    assert any("k_cost<3, 5," in k for k in mix) and any("k_band<16, 4, true, 0," in k for k in mix)
    for k, (ff, fs) in mix.items():
        assert 0.0 < ff < 1.0 and abs(ff + fs - 1.0) < 1e-9, k
    # a kernel made of the plain class alone would issue at 1.55, of the packed class alone at 0.9
    assert bench.RATE_FAST > bench.RATE_SLOW > 0
