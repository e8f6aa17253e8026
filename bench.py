#!/usr/bin/env python3
"""bench.py -- stereo pairs/s at 1920x1080, numDisparities=128 on MI355X (BASELINE.json metric).

A step = one pass of the SGBM hot path (BT cost volume -> path aggregation -> WTA / uniqueness /
LR check -> median) over one batch of synthetic rectified pairs that are already resident in HBM.
Consecutive steps alternate over --in-flight (default 2) independent sets of {inputs, handle, output, HIP stream}, the
way a streaming deployment double-buffers: the VALU-bound cost kernel of one batch then overlaps the HBM-bound last
pass of the previous one (+8 % over one batch at a time, which `also.single_stream_pairs_per_s` reports).  The
timed region is exactly --steps steps between barrier + synchronize.
Independent pairs shard across ranks with no data-path collective ("scaling": "weak": every rank
processes its own batch of --batch pairs); the only RCCL traffic is the one-time broadcast of the rig's
remap tables and the end-of-run reduction of timings / checksums.

    python bench.py [--gpus N --steps K --warmup W]

With --gpus N > 1 and no RANK in the environment the script launches itself under
`python -m torch.distributed.run --nproc-per-node N` (one rank per GPU, RCCL); it exits non-zero with
a message when fewer than N GPUs are visible.  Under an external torchrun it reads RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* from the environment.

Default workload = BASELINE.json configs[1]/[2]: 1920x1080 RGB pairs (the reference feeds RGB), D=128,
blockSize=5, cv2's default MODE_SGBM (5 paths: the mode the reference's cv2.StereoSGBM_create call
selects), 64 pairs per GPU per step (512 pairs / 8 GPUs).  --mode hh runs the 8-path MODE_HH; its
throughput is also reported in "also" on every default run.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
CLOCK_HZ, N_CUS = 2.4e9, 256  # MI355X peak engine clock and CUs: one VALU wave-instruction per CU and cycle (4 SIMD16s)


def _latest_profile(pattern, pairs_per_launch=None, exclude=None):
    """The committed rocprofv3 --pmc summary of the most recent round that has one (profiles/rNN_<pattern>); with
    `pairs_per_launch`, the most recent one taken at exactly that many pairs per launch is preferred (per-pair traffic
    depends a little on it: the band passes' edge records and tails)."""
    import glob
    hits = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + pattern))
                  if not (exclude and exclude in os.path.basename(f)))
    if pairs_per_launch is not None:
        same = []
        for f in hits:
            try:
                if json.load(open(f)).get("pairs_per_launch") == pairs_per_launch:
                    same.append(f)
            except Exception:
                pass
        hits = same or hits
    return os.path.relpath(hits[-1], ROOT) if hits else None


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=80, help="timed steps (default: >= 3 s of GPU work)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="pairs per GPU per step")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="batches in flight per GPU: consecutive steps alternate over this many handle + stream sets, so "
                         "the VALU-bound cost kernel of one batch overlaps the HBM-bound last pass of the previous one")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--disparities", type=int, default=128)
    ap.add_argument("--block", type=int, default=5)
    ap.add_argument("--channels", type=int, default=3, help="3 = RGB as the reference feeds SGBM, 1 = gray")
    ap.add_argument("--mode", default="sgbm", choices=["sgbm", "hh"],
                    help="sgbm = 5-path MODE_SGBM (the reference's call), hh = 8-path MODE_HH")
    ap.add_argument("--path", type=int, default=0, help="aggregation path: 0 auto, 1 scans, 2 band passes, 3 concurrent")
    ap.add_argument("--cost", type=int, default=0, help="cost-volume kernels: 0 auto (fused), 1 fused k_cost, 2 k_hsum + k_vsum")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary measurements (other mode, gray, PCIe)")
    ap.add_argument("--cpu-pairs", type=int, default=0, help="strips in the CPU baseline sample (0 = two per thread)")
    ap.add_argument("--lib", default="", help="measurement only: load this build of libcalibrating_amd.so (A/B kernels)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the in-run rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ counters of one step in a child "
                         "process); roofline.traffic / valu_frac then come from the committed profiles/ summaries")
    ap.add_argument("--pmc-budget-s", type=float, default=240.0, help="wall-time budget of the in-run counter passes")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the N > 1 run: nccl (= RCCL, the product) or gloo (CPU; with --stub-compute)")
    ap.add_argument("--stub-compute", action="store_true",
                    help="TEST HOOK, no GPU: CPU tensors and a no-op matcher that returns a deterministic tensor, so that "
                         "everything around the kernels -- sharding, process group, table broadcast / install, the timed "
                         "region, the reductions, the JSON line -- runs end to end with N ranks on a CPU box "
                         "(tests/test_parallel_cpu.py).  The line it prints is marked \"data\": \"stub\" and is not a measurement")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-budget-s", type=float, default=25.0, help="time budget of the CPU baseline's thread sweep")
    return ap.parse_args(argv)


PMC_SETS = (("FETCH_SIZE", ["FETCH_SIZE"]), ("WRITE_SIZE", ["WRITE_SIZE"]),
            ("SQ", ["SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_BUSY_CU_CYCLES"]))


def pmc_in_run(a, nb, budget_s):
    """HBM traffic and SQ counters of THIS run's kernels: one step of the same workload in a child process under
    `rocprofv3 --pmc <set> --kernel-trace`, one pass per counter set (FETCH_SIZE and WRITE_SIZE do not fit one pass;
    MI355X_MICROARCH.md, HBM / rocprofv3 section), no other trace domain.  HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) KB:
    FETCH_SIZE counts wide coalesced reads at half their size on gfx950 (same guide).  Returns
    ({"pairs_per_launch", "kernels": {name: {...per-launch averages...}}}, None) or (None, reason)."""
    import collections
    import csv
    import glob
    import re
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 is not on PATH"
    child = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--in-flight", "1",
             "--batch", str(nb), "--no-also", "--no-cpu-baseline", "--no-pmc", "--width", str(a.width),
             "--height", str(a.height), "--disparities", str(a.disparities), "--block", str(a.block),
             "--channels", str(a.channels), "--mode", a.mode, "--path", str(a.path), "--cost", str(a.cost)]
    if a.lib:
        child += ["--lib", os.path.abspath(a.lib)]
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    t_end = time.time() + budget_s
    per = collections.OrderedDict()
    tmp = tempfile.mkdtemp(prefix="camd_pmc_", dir="/tmp")
    try:
        for tag, ctrs in PMC_SETS:
            left = t_end - time.time()
            if left < 20:
                return None, "the counter passes ran out of their %.0f s budget" % budget_s
            d = os.path.join(tmp, tag)
            try:
                r = subprocess.run([exe, "--pmc"] + ctrs + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p",
                                                           "--"] + child, cwd="/tmp", env=env, timeout=left,
                                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            except subprocess.TimeoutExpired:
                return None, "rocprofv3 --pmc %s did not finish within the budget" % tag
            files = glob.glob(os.path.join(d, "**", "*counter_collection*.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s failed (exit %d, %d counter files)" % (tag, r.returncode, len(files))
            agg, cnt = collections.defaultdict(float), collections.Counter()
            for f in files:
                for row in csv.DictReader(open(f)):
                    k = row.get("Kernel_Name", "")
                    if "camd::" not in k:
                        continue
                    k = re.sub(r"^void ", "", k.split("(")[0])
                    agg[(k, row["Counter_Name"])] += float(row["Counter_Value"])
                    cnt[(k, row["Counter_Name"])] += 1
            for (k, c), v in agg.items():
                per.setdefault(k, {})[c] = v / cnt[(k, c)]  # average per launch
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for k, v in per.items():
        f, w = v.get("FETCH_SIZE"), v.get("WRITE_SIZE")
        if f is not None and w is not None:
            v["hbm_bytes_per_launch"] = (2 * f + w) * 1024
            v["hbm_bytes_per_pair"] = v["hbm_bytes_per_launch"] / nb
        if v.get("SQ_WAVE_CYCLES"):
            v["derived"] = {"wait_any_frac": v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"]}
    if not any("hbm_bytes_per_pair" in v for v in per.values()):
        return None, "no camd:: kernel in the counter files"
    return {"pairs_per_launch": nb, "kernels": per,
            "command": "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES> "
                       "--kernel-trace -- python bench.py --steps 1 --warmup 1 --in-flight 1 --batch %d (same workload)" % nb}, None


def valu_class_mix():
    """{kernel substring: (fraction of its VALU instructions in the plain 32-bit class, fraction in the packed / DPP /
    three-operand class)} from the committed listing statistics (profiles/rNN_isa_valu_mix.json, tools/isa_cost.py)."""
    rel = _latest_profile("isa_valu_mix.json")
    if not rel:
        return None, {}
    doc = json.load(open(os.path.join(ROOT, rel)))
    return rel, {k: (v["fast_frac"], v["slow_frac"]) for k, v in doc["kernels"].items()}


# measured issue rates of the two VALU classes (tools/microtests/valu_rate.hip), wave-instructions per cycle and CU
RATE_FAST, RATE_SLOW = 1.55, 0.90


def sgbm_params(a, mode=None, channels=None):
    cn, bs = (a.channels if channels is None else channels), a.block
    mode = a.mode if mode is None else mode
    return dict(minDisparity=0, numDisparities=a.disparities, blockSize=bs, P1=8 * cn * bs * bs,
                P2=32 * cn * bs * bs, disp12MaxDiff=1, preFilterCap=0, uniquenessRatio=10,
                speckleWindowSize=0, speckleRange=0, mode=1 if mode == "hh" else 0)


def algorithmic_bytes_per_pair(W, H, D, cn, minD=0):
    """B_alg = 2*V + IO, V = H*W1*D*2, IO = 2*H*W*cn + 2*H*W."""
    maxD = minD + D
    W1 = (W + min(minD, 0)) - max(maxD, 0)
    V = H * W1 * D * 2
    return 2 * V + 2 * H * W * cn + 2 * H * W, V


# stage (hipEvent bracket inside the library) -> (kernel it times, algorithmic HBM bytes per pair as f(V, HW, cn),
# substrings that identify the kernel in a rocprofv3 summary)
def stage_table(V, HW, cn, mode):
    last = "k_band last pass (%s + WTA: C and S in)" % ("4 directions" if mode == "hh" else "1 direction")
    return {
        "cost": ("k_cost (BT cost + KxK box sum -> C, fused)", V + 2 * HW * cn, ("k_cost",)),
        "hsum": ("k_hsum (BT cost + horizontal box sum)", V + 2 * HW * cn, ("k_hsum",)),
        "vsum": ("k_vsum (vertical box sum + P2 -> C)", 2 * V, ("k_vsum",)),
        "scan": ("k_band first pass (4 directions: C in, S out)", 2 * V, ("k_band", ", true, 0,")),
        "scan_last": (last, 2 * V, ("k_band", ", 2,")),
        "wta": ("k_lrcheck", 12 * HW, ("k_lrcheck",)),
        "median": ("k_median3", 4 * HW, ("k_median3",)),
        "speckle": ("k_cc_* (filterSpeckles: label, borders, count, apply)", 4 * HW, ("k_cc_",)),
    }


def _host_topology():
    """What bounds a host-side baseline on this box: usable CPUs, the cgroup CPU quota, sockets / NUMA nodes."""
    import glob
    info = {"os_cpu_count": os.cpu_count(), "affinity_cpus": len(os.sched_getaffinity(0))}
    quota = None
    try:  # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        info["cgroup_cpu_max"] = "%s %s" % (q, per)
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        try:  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            info["cgroup_cpu_max"] = "%d %d" % (q, per)
            quota = None if q <= 0 else q / per
        except Exception:
            info["cgroup_cpu_max"] = None
    info["cgroup_quota_cpus"] = quota
    info["numa_nodes"] = len(glob.glob("/sys/devices/system/node/node[0-9]*")) or None
    try:
        pk = {open(f).read().strip() for f in glob.glob("/sys/devices/system/cpu/cpu[0-9]*/topology/physical_package_id")}
        cores = {(open(f.replace("core_id", "physical_package_id")).read().strip(), open(f).read().strip())
                 for f in glob.glob("/sys/devices/system/cpu/cpu[0-9]*/topology/core_id")}
        info["sockets"], info["physical_cores"] = len(pk) or None, len(cores) or None
    except Exception:
        info["sockets"] = info["physical_cores"] = None
    return info


def cpu_baseline_worker(a):
    """Runs in a process of its own (no torch, no GPU binding, its own OpenMP pool): the scalar C oracle on the host
    cores, swept over thread counts.  Prints one JSON object."""
    try:
        os.sched_setaffinity(0, range(os.cpu_count() or 1))  # before the first OpenMP region creates its pool
    except OSError:
        pass
    import oracle
    from calibrating_amd import synthetic
    oracle.build()
    params = sgbm_params(a)
    topo = _host_topology()
    usable = topo["affinity_cpus"]
    # bounded sample: full-width strips of a quarter of the rows (SGBM cost is linear in rows), ONE strip per
    # thread, distinct data per strip; rates are scaled back to whole pairs.  Inputs and outputs exist (and are
    # touched) before any clock starts; the oracle's own scratch is a few rows per thread in MODE_SGBM
    hs = max(a.height // 4, 16)
    cap = usable
    if a.mode == "hh":  # the two-pass mode keeps two whole strip volumes per thread: bound the host memory (32 GB)
        cap = max(1, min(cap, int(32e9 // (2 * 2 * hs * max(a.width - a.disparities, 1) * a.disparities))))
    counts = sorted({t for t in (1, 8, 16, 32, 64, 128, 256, usable) if t <= cap})
    nmax = counts[-1]
    base_l, base_r = synthetic.rectified_pair(seed=1234, H=hs, W=a.width, D=a.disparities, cn=a.channels)
    L = np.stack([np.roll(base_l, 17 * i, axis=0) for i in range(nmax)])
    R = np.stack([np.roll(base_r, 17 * i, axis=0) for i in range(nmax)])
    oracle.sgbm_compute_batch(L[:1, :32], R[:1, :32], nthreads=1, **params)  # (library paged in)
    t_start = time.perf_counter()
    sweep, secs = {}, {}
    # one whole pair on one thread: the latency of the reference's one-pair-per-call surface
    full_l, full_r = synthetic.rectified_pair(seed=1234, H=a.height, W=a.width, D=a.disparities, cn=a.channels)
    t0 = time.perf_counter()
    oracle.sgbm_compute(full_l, full_r, **params)
    single_call_ms = (time.perf_counter() - t0) * 1e3
    for t in counts:
        if t > 1 and time.perf_counter() - t_start > a.cpu_budget_s:
            break
        t0 = time.perf_counter()
        oracle.sgbm_compute_batch(L[:t], R[:t], nthreads=t, **params)
        secs[t] = time.perf_counter() - t0
        sweep[t] = (t * hs / a.height) / secs[t]
    best = max(sweep, key=sweep.get)
    one = sweep.get(1) or (1e3 / single_call_ms)
    eff = sweep[best] / best / one
    quota = topo["cgroup_quota_cpus"]
    if eff >= 0.5:
        why = "per-thread rate at the best point is %.2f of the single-thread rate" % eff
    else:
        why = ("per-thread rate at the best point is %.2f of the single-thread rate: " % eff) + (
            "the container's cgroup CPU quota is %.1f CPUs (cpu.max = %s), far fewer than the %d CPUs it may be scheduled on"
            % (quota, topo["cgroup_cpu_max"], usable) if quota and quota < best else
            "%d hardware threads share %s physical cores (SMT) and the port's per-thread working set (a row of the "
            "volume per buffer) competes for the shared caches" % (usable, topo.get("physical_cores")))
    print(json.dumps(dict(
        value=sweep[best], unit="pairs/s", cores=best, kind="port", single_call_ms=single_call_ms,
        single_thread_pairs_per_s=one, per_thread_efficiency_at_best=eff, efficiency_note=why,
        thread_sweep_pairs_per_s={str(k): v for k, v in sweep.items()},
        thread_sweep_seconds={str(k): v for k, v in secs.items()}, host=topo,
        sample="thread sweep %s, per point one 1920x%d strip per thread (= %.2f pairs each; rates scaled to whole %dx%d "
               "pairs), D=%d cn=%d mode=%s, scalar C port oracle/sgbm_ref.c with OpenMP across strips, run in a "
               "process of its own with the affinity mask lifted to all %d CPUs; best point = `value`; single_call_ms "
               "= one whole pair on one thread; %.1f s in all; cv2 itself is not installed on this box"
               % (sorted(sweep), hs, hs / a.height, a.width, a.height, a.disparities, a.channels, a.mode, usable,
                  time.perf_counter() - t_start))))


def cpu_baseline(a, params):
    """The CPU oracle (scalar C port of cv2.StereoSGBM, oracle/sgbm_ref.c) on this host's cores, in a subprocess:
    this process is bound to the CPUs next to its GPU and shares an OpenMP pool with torch, neither of which a fair
    host-side baseline should inherit."""
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--width", str(a.width), "--height",
           str(a.height), "--disparities", str(a.disparities), "--block", str(a.block), "--channels", str(a.channels),
           "--mode", a.mode, "--cpu-budget-s", str(a.cpu_budget_s)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "OMP_NUM_THREADS")}
    env["OMP_WAIT_POLICY"] = "passive"  # a finished team must not spin on CPUs the next, larger team needs
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    if p.returncode != 0 or not p.stdout.strip():
        return dict(value=None, unit="pairs/s", cores=0, kind="port", sample="CPU baseline failed: " + p.stderr[-300:])
    return json.loads(p.stdout.strip().splitlines()[-1])


def gpu_steps(matcher, left, right, out, steps, warmup, distributed=False, sync=None):
    """(seconds, {stage: ms summed over the timed steps}) -- parallel_pairs.timed_steps around compute()."""
    import torch
    from calibrating_amd.parallel_pairs import timed_steps
    stage_ms = {}
    sync = sync or torch.cuda.synchronize

    def step():
        matcher.compute(left, right, out=out)

    def timed_step():
        matcher.compute(left, right, out=out)
        for k, v in matcher.stage_times_ms().items():  # hipEvents on the compute stream, read after the fact
            stage_ms[k] = stage_ms.get(k, 0.0) + v

    timed_steps(step, 0, warmup, sync, False)
    dt = timed_steps(timed_step, steps, 0, sync, distributed)
    return dt, stage_ms


def pipelined_steps(matchers, streams, lefts, rights, outs, steps, warmup, distributed=False, sync=None):
    """seconds for `steps` steps, step k on handle / stream k % len(matchers); nothing synchronises between steps."""
    import contextlib
    import torch
    from calibrating_amd.parallel_pairs import timed_steps
    n = len(matchers)
    k = [0]
    sync = sync or torch.cuda.synchronize
    on = lambda st: torch.cuda.stream(st) if st is not None else contextlib.nullcontext()  # noqa: E731
    # set-up, not a step: the first compute on a handle touches its freshly allocated workspace (seconds for 80 GB),
    # so every set runs once before the W warm-up steps -- whatever W is
    for i in range(n):
        with on(streams[i]):
            matchers[i].compute(lefts[i], rights[i], out=outs[i])
    sync()

    def step():
        i = k[0] % n
        k[0] += 1
        with on(streams[i]):
            matchers[i].compute(lefts[i], rights[i], out=outs[i])

    return timed_steps(step, steps, warmup, sync, distributed)


def pcie_inclusive(matcher, left, right, out, steps, streams=()):
    """pairs/s when every step's inputs come from (pinned) host memory and its disparities go back to it:
    H2D of step k+1 and D2H of step k-1 overlap the compute of step k on three streams, two device buffers."""
    import torch
    nb = left.shape[0]
    hl, hr = left.cpu().pin_memory(), right.cpu().pin_memory()
    ho = torch.empty(out.shape, dtype=out.dtype).pin_memory()
    dl = [left, torch.empty_like(left)]
    dr = [right, torch.empty_like(right)]
    do = [out, torch.empty_like(out)]
    # the copy streams REUSE the side streams of the timed region where they exist: HIP maps streams onto a handful of
    # hardware queues, and a fifth live stream would share one with (and serialise against) the compute stream
    pool = list(streams) + [torch.cuda.Stream() for _ in range(max(0, 2 - len(streams)))]
    s_in, s_out, s_cmp = pool[0], pool[1], torch.cuda.current_stream()
    ev_in = [torch.cuda.Event() for _ in range(2)]
    ev_cmp = [torch.cuda.Event() for _ in range(2)]
    ev_out = [torch.cuda.Event() for _ in range(2)]

    def run(k):
        b = k & 1
        with torch.cuda.stream(s_in):
            s_in.wait_event(ev_cmp[b])        # buffer b's previous compute has consumed its inputs
            dl[b].copy_(hl, non_blocking=True)
            dr[b].copy_(hr, non_blocking=True)
            ev_in[b].record(s_in)
        s_cmp.wait_event(ev_in[b])
        s_cmp.wait_event(ev_out[b])           # buffer b's previous result has left
        matcher.compute(dl[b], dr[b], out=do[b])
        ev_cmp[b].record(s_cmp)
        with torch.cuda.stream(s_out):
            s_out.wait_event(ev_cmp[b])
            ho.copy_(do[b], non_blocking=True)
            ev_out[b].record(s_out)

    for k in range(2):
        run(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        run(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bytes_per_pair = (hl[0].numel() + hr[0].numel()) + ho[0].numel() * 2
    return dict(pairs_per_s=nb * steps / dt, host_bytes_per_pair=bytes_per_pair,
                host_link_GBs=bytes_per_pair * nb * steps / dt / 1e9,
                note="pinned host buffers, H2D + compute + D2H overlapped on 3 streams, %d steps of %d pairs" % (steps, nb))


def depth_path_rate(ca, synthetic, params, imgs1, imgs2, streams, W, H, D, cn, max_depth, reps):
    """Stereo.get_depth_batch (rectify x2 -> SGBM -> disp_to_depth -> unrectify -> undistort) on `imgs1` / `imgs2`:
    pairs/s with one batch at a time, and with two batches in flight (two rigs' worth of handles on two streams, as
    the headline runs the matcher), plus the roofline of SURVEY.md 8(d): SGBM bytes + 81*H*W per pair."""
    import torch
    nb = imgs1.shape[0]
    stereos = []
    for _ in range(2):
        st = ca.Stereo.load(synthetic.rig(W, H))
        st.set_stereo_matching(ca.SemiGlobalBlockMatching(dict(params, max_size=max(W, H))), max_depth=max_depth)
        stereos.append(st)
    pool = list(streams)[:2] + [torch.cuda.Stream() for _ in range(max(0, 2 - len(streams)))]
    for i in range(2):  # set-up: tables, workspaces, allocator pools of both streams
        with torch.cuda.stream(pool[i]):
            stereos[i].get_depth_batch(imgs1, imgs2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        stereos[0].get_depth_batch(imgs1, imgs2)
    torch.cuda.synchronize()
    single = nb * reps / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    for k in range(2 * reps):
        with torch.cuda.stream(pool[k & 1]):
            stereos[k & 1].get_depth_batch(imgs1, imgs2)
    torch.cuda.synchronize()
    double = nb * 2 * reps / (time.perf_counter() - t0)
    for st in stereos:
        st.stereo_matching.stereo_sgbm.status()
    b_sgbm, _ = algorithmic_bytes_per_pair(W, H, D, cn)
    b_alg = b_sgbm + 81 * W * H
    return dict(pairs_per_s=double, single_stream_pairs_per_s=single, batches_in_flight=2, pairs_per_call=nb,
                algorithmic_bytes_per_pair=b_alg, achieved_GBs=b_alg * double / 1e9,
                frac=b_alg * double / 1e9 / HBM_PEAK_GBS,
                note="rectify x2 (Lanczos-4) + SGBM + disp_to_depth + unrectify + undistort, device-resident RGB pairs; "
                     "B_alg = 2V + IO + 81*H*W (SURVEY.md 8d)")


def config_c4(ca, synthetic, dev, nb=16, reps=3):
    """BASELINE.json configs[3]: 3840x2160 gray pairs, D=256, blockSize 5, MODE_SGBM, 16 pairs per call."""
    import torch
    W, H, D = 3840, 2160, 256
    P = dict(minDisparity=0, numDisparities=D, blockSize=5, P1=8 * 25, P2=32 * 25, disp12MaxDiff=1, preFilterCap=0,
             uniquenessRatio=10, speckleWindowSize=0, speckleRange=0, mode=0)
    L, R = synthetic.rectified_batch_torch(7, nb, H, W, D, 1, dev)
    m = ca.StereoSGBM_create(**P)
    out = torch.empty((nb, H, W), dtype=torch.int16, device=dev)
    m.compute(L, R, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        m.compute(L, R, out=out)
    torch.cuda.synchronize()
    rate = nb * reps / (time.perf_counter() - t0)
    m.status()
    b_alg, V = algorithmic_bytes_per_pair(W, H, D, 1)
    # per-kernel table (hipEvents inside the library, each kernel alone on the GPU), like C5's per-stage table
    m.set_profiling(True)
    m.compute(L, R, out=out)
    torch.cuda.synchronize()
    table = stage_table(V, W * H, 1, "sgbm")
    kernels = []
    for st, ms in m.stage_times_ms().items():
        if st in table and ms > 0.02:  # (an empty stage bracket still measures a few microseconds)
            name, bpp, _ = table[st]
            kernels.append(dict(stage=st, kernel=name, ms_per_call=ms, algorithmic_bytes_per_pair=bpp,
                                hbm_frac=bpp * nb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS))
    del m, L, R, out
    torch.cuda.empty_cache()
    return dict(workload="3840x2160 gray rectified pairs, numDisparities=256 blockSize=5 MODE_SGBM, %d pairs per call, "
                         "one batch at a time" % nb, pairs_per_s=rate, algorithmic_bytes_per_pair=b_alg,
                achieved_GBs=b_alg * rate / 1e9, frac=b_alg * rate / 1e9 / HBM_PEAK_GBS, per_kernel=kernels,
                note="throughput-bound at the same per-byte efficiency as the 1080p workload, not by the fill of its 78-band "
                     "wavefront: 24 pairs per call, 12 x 2 or 8 x 3 in flight give the same rate (profiles/r05_c4_batching.txt). "
                     "At 8 registers per lane (D = 256) the band passes fit ONE 7 + 1-wave workgroup per CU (167 VGPRs), which "
                     "is what holds their fractions at 0.53 / 0.66; k_cost for gray writes its volume at ~2.9 TB/s, the rate "
                     "at which 16-byte store pieces of 16 waves combine in the L2")


def depth_path_stages(st, imgs1, imgs2, W, H, D, cn, reps=5):
    """Per-stage GPU time of one Stereo.get_depth_batch call (each stage alone on the GPU, torch events on the current
    stream, which is the stream the library launches on; the SGBM kernels from the library's own hipEvent brackets), with
    each stage's algorithmic HBM bytes per pair and the fraction of the HBM peak it reaches."""
    import torch
    sm = st.stereo_matching
    sg = sm.stereo_sgbm
    nb, HW = imgs1.shape[0], W * H
    _, V = algorithmic_bytes_per_pair(W, H, D, cn)

    def timed(fn):
        out = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps, out

    rows = []  # (stage, ms per call, algorithmic bytes per pair)
    ms, (r1, r2) = timed(lambda: st.rectify(imgs1, imgs2))
    rows.append(("rectify x2 (k_remap_f32 Lanczos-4, right image translated)", ms, 4 * HW * cn))
    sg.compute(r1, r2)
    sg.set_profiling(True)
    acc = {}
    for _ in range(reps):
        disp16 = sg.compute(r1, r2)
        for k, v in sg.stage_times_ms().items():
            acc[k] = acc.get(k, 0.0) + v / reps
    sg.set_profiling(False)
    table = stage_table(V, HW, cn, "sgbm")
    for k, v in acc.items():
        if k in table and v >= 0.02:  # (an empty stage bracket still measures a few microseconds)
            rows.append(("SGBM " + table[k][0], v, table[k][1]))
    tb = st._tables(imgs1.device)
    ms, (_, depth) = timed(lambda: st._fused_depth(sm, disp16, tb))
    rows.append(("disp_to_depth (k_disp_to_depth: int16 -> f32 disparity + f64 depth)", ms, 14 * HW))
    ms, _ = timed(lambda: st.unrectify_depth(depth))
    rows.append(("unrectify_depth (k_unrectify: f64 gather)", ms, 16 * HW))
    ms, _ = timed(lambda: st.undistort_img(imgs1))
    rows.append(("undistort_img1 (k_remap_fixed_bilinear)", ms, 2 * HW * cn))
    total = sum(r[1] for r in rows)
    return {"pairs_per_call": nb, "sum_ms_per_call": total,
            "stages": [{"stage": n, "ms_per_call": m, "share": m / total, "algorithmic_bytes_per_pair": b,
                        "hbm_frac": b * nb / (m * 1e-3) / 1e9 / HBM_PEAK_GBS} for n, m, b in rows]}


def config_c5(ca, synthetic, dev, streams, nb=128):
    """BASELINE.json configs[4]: 640x480 RGB, D=64, LR check on, speckle 100 / 2, the FULL get_depth path, batched.

    The primary figure stays on the workload of rounds 1-3 so that rounds compare like for like: unrelated random
    textures in the two cameras (every SGBM kernel costs the same whatever the content, but the matcher's output is all
    speckles -- the worst case of the data-dependent speckle filter), 128 pairs per call.  A real stereo scene (textured
    planes ray-cast through the rig's Brown models: smooth disparities, what the LR check and the speckle filter are
    for) and 256 pairs per call are reported beside it under ``also_measured``."""
    import torch
    W, H, D = 640, 480, 64
    P = dict(minDisparity=0, numDisparities=D, blockSize=5, P1=8 * 3 * 25, P2=32 * 3 * 25, disp12MaxDiff=1, preFilterCap=0,
             uniquenessRatio=10, speckleWindowSize=100, speckleRange=2, mode=0)
    rec = synthetic.rig(W, H)
    planes = [((0.3, 0.1, 1.0), 2.0), ((-0.2, 0.15, 1.0), 1.6), ((0.0, 0.0, 1.0), 2.5), ((0.1, -0.25, 1.0), 1.3)]
    scene = [synthetic.render_plane_pair(rec, n_, d_, seed=i)[:2] for i, (n_, d_) in enumerate(planes)]
    noise = [synthetic.scene_pair(100 + i, W, H, 3) for i in range(8)]

    def batch(n, src):
        return (torch.from_numpy(np.stack([src[i % len(src)][0] for i in range(n)])).to(dev),
                torch.from_numpy(np.stack([src[i % len(src)][1] for i in range(n)])).to(dev))

    def rate(n, src):
        b1, b2 = batch(n, src)
        out = depth_path_rate(ca, synthetic, P, b1, b2, streams, W, H, D, 3, max_depth=3.5, reps=6)
        del b1, b2
        return out
    B1, B2 = batch(nb, noise)
    r = depth_path_rate(ca, synthetic, P, B1, B2, streams, W, H, D, 3, max_depth=3.5, reps=6)
    r["workload"] = ("640x480 RGB pairs (unrelated random textures in the two cameras: the input of rounds 1-3) through the "
                     "whole get_depth path (rectify x2, SGBM numDisparities=64 blockSize=5 LR check on speckle 100/2, "
                     "disp_to_depth, unrectify, undistort), %d pairs per call" % nb)
    st = ca.Stereo.load(synthetic.rig(W, H))
    st.set_stereo_matching(ca.SemiGlobalBlockMatching(dict(P, max_size=max(W, H))), max_depth=3.5)
    r["per_stage"] = depth_path_stages(st, B1, B2, W, H, D, 3)
    del B1, B2
    pick = lambda d: {k: d[k] for k in ("pairs_per_s", "single_stream_pairs_per_s", "frac")}  # noqa: E731
    r["also_measured"] = {
        "random_textures_at_256_pairs_per_call": pick(rate(256, noise)),
        "rendered_scene_at_128_pairs_per_call": pick(rate(128, scene)),
        "rendered_scene_at_256_pairs_per_call": pick(rate(256, scene)),
        "note": "an image this small needs a deep batch to fill the band passes' last round of workgroups (9 bands x pairs "
                "over 512 places); on a rendered scene the matcher's output is smooth and the speckle filter cheaper",
    }
    return r


class StubMatcher:
    """--stub-compute: stands in for StereoSGBM where there is no GPU.  compute() writes a deterministic function of
    its inputs (so the checksum of checksums depends on every rank's shard) and does no stereo matching."""
    def set_option(self, *_):
        return self

    def set_profiling(self, *_):
        pass

    def status(self):
        pass

    def stage_times_ms(self):
        return {}

    def compute(self, left, right, out=None):
        import torch
        v = (left[..., 0].to(torch.int16) - right[..., 0].to(torch.int16)) if left.dim() == 4 else \
            (left.to(torch.int16) - right.to(torch.int16))
        if out is None:
            return v
        out.copy_(v)
        return out


def stub_inputs(seed, n, H, W, cn):
    import torch
    g = torch.Generator().manual_seed(int(seed))
    shape = (n, H, W, cn) if cn > 1 else (n, H, W)
    return (torch.randint(0, 256, shape, generator=g, dtype=torch.uint8),
            torch.randint(0, 256, shape, generator=g, dtype=torch.uint8))


def self_launch(a):
    """--gpus N > 1 without a launcher: run N ranks of this script under torch.distributed.run."""
    import torch
    have = a.gpus if a.stub_compute else torch.cuda.device_count()
    if have < a.gpus:
        sys.stderr.write("bench.py: --gpus %d requested but only %d GPU(s) visible; refusing to report a "
                         "multi-GPU number from fewer devices\n" % (a.gpus, have))
        sys.exit(2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.call(cmd, env=env))


def main():
    a = parse()
    if a.cpu_baseline_worker:
        return cpu_baseline_worker(a)
    if a.gpus > 1 and "RANK" not in os.environ:
        self_launch(a)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "RANK" in os.environ and a.gpus not in (1, world):
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (a.gpus, world))
    stub = a.stub_compute
    if stub and a.backend != "gloo":
        sys.exit("bench.py: --stub-compute is the CPU test hook and goes with --backend gloo")
    if not stub and a.backend != "nccl":
        sys.exit("bench.py: the product runs over RCCL (--backend nccl); gloo is for --stub-compute")
    if stub:
        dev, cpus_near_gpu = torch.device("cpu"), None
        sync = lambda: None  # noqa: E731
    else:
        assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback"
        if local_rank >= torch.cuda.device_count():
            sys.exit("bench.py: rank %d has no GPU (LOCAL_RANK %d, %d visible)" % (rank, local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        from calibrating_amd import hostio
        cpus_near_gpu = hostio.bind_near_gpu(local_rank)
        sync = torch.cuda.synchronize
    # CAMD_BENCH_FORCE_DIST=1 exercises the RCCL path (init, table broadcast, barrier, reductions) with a
    # single rank, e.g. under `python -m torch.distributed.run --nproc-per-node 1`
    distributed = world > 1 or (os.environ.get("CAMD_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ["NCCL_DEBUG"] = os.environ.get("CAMD_NCCL_DEBUG", "WARN")  # keep RCCL's banner off stdout
        if stub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if a.lib:
        from calibrating_amd import _native
        _native.LIB_PATH = os.path.abspath(a.lib)
    import calibrating_amd as ca
    from calibrating_amd import synthetic
    from calibrating_amd.parallel_pairs import aggregate, broadcast_tables, ranks_agree, shard_range

    params = sgbm_params(a)
    # this rank's shard of the global pair list: pairs [lo, hi) of world*batch
    lo, hi = shard_range(world * a.batch, world, rank)
    nb = hi - lo
    nfl = max(1, min(a.in_flight, a.steps))
    lefts, rights, outs, matchers, streams = [], [], [], [], []
    for i in range(nfl):  # every batch in flight has its own inputs, handle (workspace), output and stream
        if stub:
            l, r = stub_inputs(1234 + rank + 1000 * i, nb, a.height, a.width, a.channels)
            m = StubMatcher()
        else:
            l, r = synthetic.rectified_batch_torch(1234 + rank + 1000 * i, nb, a.height, a.width, a.disparities,
                                                   a.channels, dev)
            m = ca.StereoSGBM_create(**params)
        m.set_option("path", a.path)
        m.set_option("cost", a.cost)
        lefts.append(l); rights.append(r); matchers.append(m)
        outs.append(torch.empty((nb, a.height, a.width), dtype=torch.int16, device=dev))
        streams.append(None if stub else torch.cuda.Stream(device=dev))
    left, right, out, matcher = lefts[0], rights[0], outs[0], matchers[0]

    # (1) kernel characterisation: a few steps on ONE stream with the library's hipEvents around every kernel
    matcher.set_profiling(True)
    kprof = max(1, min(a.steps, 5))
    dt1, stage_ms = gpu_steps(matcher, left, right, out, kprof, a.warmup, sync=sync)
    matcher.set_profiling(False)
    prof_steps = kprof
    # (2) the timed region: exactly --steps steps, `nfl` batches in flight, barrier + synchronize on both sides
    dt = pipelined_steps(matchers, streams, lefts, rights, outs, a.steps, a.warmup, distributed, sync=sync)
    for m in matchers:
        m.status()  # raises if a device-side bounded wait timed out
    checksum = sum(int(o.to(torch.int64).sum().item()) for o in outs)
    agg = aggregate(nb * a.steps, dt, checksum, dev, distributed)
    value = agg["total_pairs"] / agg["seconds"]
    single_stream = nb * kprof / dt1
    del matchers[1:], lefts[1:], rights[1:], outs[1:], m, l, r  # (the loop variables hold the last set alive)
    if not stub:
        torch.cuda.empty_cache()  # (the host-link rate measured below drops when much more HBM is allocated)

    rccl = None
    if distributed:
        # one-time table broadcast (rank 0 owns the rig), installed into every rank's Stereo; then every rank runs
        # the full get_depth on the SAME two pairs through the broadcast tables and the ranks compare checksums
        t0 = time.perf_counter()
        # ONLY rank 0 ever sees the rig record; every other rank builds its Stereo from the broadcast alone
        bundle = ca.Stereo.load(synthetic.rig(a.width, a.height)).table_bundle() if rank == 0 else None
        tables = broadcast_tables(bundle, dev, src=0)
        sync()
        t_bcast = time.perf_counter() - t0
        stereo = ca.Stereo.from_bundle(tables, dev)
        if stub:
            # no kernels to run the installed tables through: the checksum is taken over the tables as this rank's
            # Stereo now holds them (what get_depth_batch would read)
            held = stereo._tables(dev)
            dsum = int(sum(torch.nan_to_num(held[k].to(torch.float64)).mul(16).round().to(torch.int64).sum().item()
                           for k in sorted(held)))
            stereo.set_stereo_matching(StubMatcher(), max_depth=20.0)  # needs cam1.K and t from the parameter block
            dsum += int(stereo.min_disparity) + int(round(1e6 * stereo.baseline))
            res = imgs = None
        else:
            stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(dict(params, max_size=max(a.width, a.height))),
                                       max_depth=20.0)
            imgs = torch.stack([torch.from_numpy(np.ascontiguousarray(im)) for im in
                                synthetic.scene_pair(77, a.width, a.height, 3)]).to(dev)
            res = stereo.get_depth_batch(imgs[:1].repeat(2, 1, 1, 1), imgs[1:].repeat(2, 1, 1, 1))
            dsum = int(torch.nan_to_num(res["unrectify_depth"]).mul(1e4).round().to(torch.int64).sum().item())
        rccl = dict(backend="gloo (CPU stub)" if stub else "nccl (RCCL)", world_size=world,
                    table_bytes=int(sum(t.numel() * t.element_size() for t in tables.values())),
                    broadcast_s=t_bcast, get_depth_checksum=dsum, ranks_agree=bool(ranks_agree(dsum, dev)),
                    per_rank=[dict(pairs=p, seconds=s, disparity_checksum=c) for p, s, c in agg["per_rank"]])
        del stereo, tables, res, imgs

    also = {}
    if rank == 0 and world == 1 and not a.no_also and not stub:
        k2 = max(3, min(a.steps, 10))
        also["note"] = ("figures named *_pairs_per_s are measured with ONE batch in flight on one stream; get_depth_batch, "
                        "c4 and c5 say how they were run")
        also["single_stream_pairs_per_s"] = single_stream
        if os.environ.get("CAMD_BENCH_DEBUG"):
            free, total = torch.cuda.mem_get_info()
            sys.stderr.write("before pcie_inclusive: %.1f GB of %.1f GB free\n" % (free / 1e9, total / 1e9))
        also["pcie_inclusive"] = pcie_inclusive(matcher, left, right, out, k2, streams)
        del matcher
        matchers.clear()
        # the other aggregation mode on the same inputs, and the gray variant of the headline mode
        other = "hh" if a.mode == "sgbm" else "sgbm"
        m2 = ca.StereoSGBM_create(**sgbm_params(a, other))
        m2.set_profiling(True)
        m2.set_option("path", a.path)
        m2.set_option("cost", a.cost)
        d2, st2 = gpu_steps(m2, left, right, out, k2, 1)
        also["%s_%s_pairs_per_s" % ("rgb" if a.channels == 3 else "gray", other)] = nb * k2 / d2
        # the other mode's kernels one by one (hipEvents inside the library), priced like roofline.kernels: algorithmic
        # bytes per launch / launch duration against the HBM peak
        _, V2 = algorithmic_bytes_per_pair(a.width, a.height, a.disparities, a.channels)
        t2 = stage_table(V2, a.width * a.height, a.channels, other)
        also[other] = {"pairs_per_s": nb * k2 / d2, "batches_in_flight": 1, "per_kernel": {
            st: {"kernel": t2[st][0], "avg_ms_per_launch": ms / k2, "algorithmic_bytes_per_launch": float(t2[st][1] * nb),
                 "achieved_GBs": t2[st][1] * nb / (ms / k2 * 1e-3) / 1e9,
                 "frac": t2[st][1] * nb / (ms / k2 * 1e-3) / 1e9 / HBM_PEAK_GBS}
            for st, ms in st2.items() if st in t2 and ms / k2 >= 0.02}}
        del m2
        if a.channels == 3:
            m3 = ca.StereoSGBM_create(**sgbm_params(a, channels=1))
            m3.set_profiling(True)
            m3.set_option("path", a.path)
            m3.set_option("cost", a.cost)
            gl, gr = left[..., 1].contiguous(), right[..., 1].contiguous()
            d3, _ = gpu_steps(m3, gl, gr, out, k2, 1)
            also["gray_%s_pairs_per_s" % a.mode] = nb * k2 / d3
            del m3, gl, gr
            # the whole Stereo.get_depth path around the same matcher (rectify x2 -> SGBM -> disp_to_depth -> unrectify
            # -> undistort) on a synthetic rig of the same size: one batch at a time, and double-buffered like the headline
            torch.cuda.empty_cache()
            also["get_depth_batch"] = depth_path_rate(ca, synthetic, params, left, right, streams, a.width, a.height,
                                                      a.disparities, 3, max_depth=20.0, reps=4)
            also["get_depth_batch_pairs_per_s"] = also["get_depth_batch"]["pairs_per_s"]
            del left, right, out, lefts, rights, outs
            torch.cuda.empty_cache()
            also["c4"] = config_c4(ca, synthetic, dev)
            also["c5"] = config_c5(ca, synthetic, dev, streams)

    if rank == 0:
        b_alg, V = algorithmic_bytes_per_pair(a.width, a.height, a.disparities, a.channels)
        table = stage_table(V, a.width * a.height, a.channels, a.mode)
        kernel_ms_step = sum(stage_ms.values()) / prof_steps   # one batch alone on the GPU
        gpu_ms_step = agg["seconds"] / a.steps * 1e3             # the timed region (batches in flight overlap)
        # HBM traffic per kernel from the committed PMC profile of the same kernels (separate rocprofv3 --pmc
        # passes, 2*FETCH_SIZE + WRITE_SIZE per the gfx950 correction); null when no profile matches this workload
        suffix = "_hh" if a.mode == "hh" else ""
        # (rNN_pmc_traffic[_hh][_b<pairs per launch>].json; the one taken at this run's pairs per launch is preferred)
        pmc_rel = _latest_profile("pmc_traffic%s*.json" % suffix, nb, exclude=None if suffix else "_hh")
        pmc_other = _latest_profile("pmc_traffic%s.json" % suffix)
        sq_rel = _latest_profile("pmc_sq%s.json" % suffix)
        pmc = sq = None
        pmc_note = "committed profile (--no-pmc)" if a.no_pmc else None
        if world == 1 and not stub and not a.no_pmc:
            # counters of THIS run's kernels, taken now (the timed region is over; its buffers are still allocated,
            # the child needs its own ~1.3 GB per pair)
            torch.cuda.empty_cache()
            inrun, why = pmc_in_run(a, nb, a.pmc_budget_s)
            if inrun:
                pmc = sq = inrun
                pmc_rel = sq_rel = "in-run: " + inrun["command"]
                pmc_other = None
            else:
                pmc_note = "committed profile (in-run counters unavailable: %s)" % why
        if pmc is None and a.channels == 3 and (a.width, a.height, a.disparities, a.block) == (1920, 1080, 128, 5):
            pmc = json.load(open(os.path.join(ROOT, pmc_rel))) if pmc_rel else None
            sq = json.load(open(os.path.join(ROOT, sq_rel))) if sq_rel else None
        mix_rel, mix = valu_class_mix()

        def pmc_bytes_per_pair(keys):
            if not pmc:
                return None
            hit = [v["hbm_bytes_per_pair"] for k, v in pmc["kernels"].items()
                   if all(s in k for s in keys) and "hbm_bytes_per_pair" in v]
            return hit[0] if len(hit) == 1 else None

        def sq_counters(keys):
            """(SQ_INSTS_VALU wave-instructions per pair, fraction of wave cycles spent in s_waitcnt / s_barrier) from the
            committed SQ counter profile of the same kernel."""
            if not sq:
                return None, None
            hit = [v for k, v in sq["kernels"].items() if all(s_ in k for s_ in keys)]
            if len(hit) != 1 or not hit[0].get("SQ_INSTS_VALU"):
                return None, None
            return hit[0]["SQ_INSTS_VALU"] / sq["pairs_per_launch"], hit[0].get("derived", {}).get("wait_any_frac")

        kernels = {}
        for st, ms_sum in stage_ms.items():
            ms = ms_sum / prof_steps
            if st not in table or ms < 0.02:  # (an empty stage bracket still measures a few microseconds)
                continue
            name, per_pair, keys = table[st]
            tr = pmc_bytes_per_pair(keys)
            ach = per_pair * nb / (ms * 1e-3) / 1e9
            kernels[st] = {"kernel": name, "avg_ms_per_launch": ms, "algorithmic_bytes_per_launch": float(per_pair * nb),
                           "achieved": ach, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                           "traffic": (tr * nb) if tr is not None else None}
            vi, waitf = sq_counters(keys)
            if vi is not None:
                # VALU issue roofline: one (packed-form) wave-instruction per CU and cycle; the instruction count is a
                # property of the kernel (counted once, committed), the duration is this run's.  A kernel that reaches
                # neither 0.8 of that nor 0.6 of the HBM peak is bound by its dependency chain (barrier per step, waits):
                # `wave_wait_frac` = share of wave cycles in s_waitcnt / s_barrier
                vf = vi * nb / (N_CUS * CLOCK_HZ) / (ms * 1e-3)
                hf = ach / HBM_PEAK_GBS
                # the kernel's own issue floor: its instructions at the measured rates of their two classes (the plain
                # 32-bit forms issue at ~1.55 per cycle and CU, packed / DPP / three-operand forms at ~0.9)
                shape = ("k_cost<%d, %d," % (a.channels, a.block),) if st == "cost" else \
                    ("k_band<16, %d," % ((a.disparities + 31) // 32),) if st in ("scan", "scan_last") else ()
                ff, fs = next((m for k, m in mix.items() if all(s_ in k for s_ in keys + shape)), (0.0, 1.0))
                floor_ms = vi * nb * (ff / RATE_FAST + fs / RATE_SLOW) / (N_CUS * CLOCK_HZ) * 1e3
                kernels[st].update(valu_wave_insts_per_launch=vi * nb, valu_frac=vf, wave_wait_frac=waitf,
                                   valu_plain_class_frac=ff, valu_floor_ms=floor_ms,
                                   hbm_floor_ms=(tr * nb / (HBM_PEAK_GBS * 1e9) * 1e3) if tr is not None else None,
                                   bound="valu" if floor_ms >= 0.8 * ms else "hbm" if hf >= 0.6 else "latency")
        dom = max(kernels, key=lambda k: kernels[k]["avg_ms_per_launch"]) if kernels else None
        # whole-step traffic = every kernel of the committed PMC profile (incl. the small init / check kernels)
        traffic_total = sum(v.get("hbm_bytes_per_pair", 0.0) for v in pmc["kernels"].values()) * nb if pmc else None
        # the step's two floors: every VALU instruction of its kernels at the issue rate of its class, and every HBM byte
        # the counters saw at the 8 TB/s peak -- what the step would take if one of the two were the only limit
        floors = [k.get("valu_floor_ms") for k in kernels.values()]
        valu_floor_ms = sum(f for f in floors if f is not None) if any(f is not None for f in floors) else None
        hbm_floor_ms = traffic_total / (HBM_PEAK_GBS * 1e9) * 1e3 if traffic_total else None
        achieved = b_alg * nb / (gpu_ms_step * 1e-3) / 1e9
        # the same ratio from the profile taken at another number of pairs per launch, when it differs by more than 2 %
        other_ratio = None
        if pmc and pmc_other and pmc_other != pmc_rel:
            po = json.load(open(os.path.join(ROOT, pmc_other)))
            ro = sum(v["hbm_bytes_per_pair"] for v in po["kernels"].values()) / b_alg
            if abs(ro / (traffic_total / (b_alg * nb)) - 1) > 0.02:
                other_ratio = {"profile": pmc_other, "pairs_per_launch": po.get("pairs_per_launch"), "traffic_ratio": ro}
        line = {
            "metric": "stereo pairs/s at 1920x1080 numDisparities=128",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": agg["seconds"] / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16", "data": "stub" if stub else "synthetic",
            "config": {"workload": "cv2.StereoSGBM-equivalent disparity of %dx%d rectified %s pairs, "
                                   "numDisparities=%d blockSize=%d mode=%s, median3 on, speckle off"
                                   % (a.width, a.height, "RGB" if a.channels == 3 else "gray", a.disparities,
                                      a.block, "MODE_HH(8 paths)" if a.mode == "hh" else "MODE_SGBM(5 paths)"),
                       "pairs_per_gpu_per_step": a.batch, "global_pairs_per_step": world * a.batch,
                       "batches_in_flight_per_gpu": nfl, "host_cpus_bound": cpus_near_gpu,
                       "parallelism": "pairs sharded over %d GPU(s), no data-path collective" % world},
            # Headline = SURVEY section 8(d): B_alg x pairs per step / time of one step in the timed region, against
            # the 8 TB/s HBM peak.  `kernels` / `dominant_kernel` characterise every kernel ALONE on the GPU (hipEvents
            # on the compute stream, a few single-stream steps before the timed region; `kernel_ms_per_step` = their
            # sum): the kernel with the largest launch duration, priced with ITS algorithmic bytes.  Every number can
            # be recomputed from the latest profiles/rNN_*kernel_stats.csv, rNN_pmc_traffic*.json and rNN_pmc_sq*.json.
            "roofline": {
                "bound": "hbm", "scope": "whole step (all kernels of one batch)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "algorithmic_bytes_per_pair": b_alg, "algorithmic_bytes_per_launch": float(b_alg * nb),
                "gpu_ms_per_step": gpu_ms_step, "kernel_ms_per_step": kernel_ms_step,
                "traffic": traffic_total,
                "traffic_ratio": (traffic_total / (b_alg * nb)) if traffic_total else None,
                "traffic_source": (pmc_rel + " (bytes per pair per launch x pairs per launch; taken at %s pairs per "
                                             "launch)" % pmc.get("pairs_per_launch")) if pmc else None,
                "counters": "in-run" if (pmc and str(pmc_rel).startswith("in-run")) else pmc_note,
                # what the step would take if VALU issue / HBM bytes were its only limit, next to what it took
                "valu_floor_ms": valu_floor_ms, "hbm_floor_ms": hbm_floor_ms,
                "frac_of_valu_floor": (valu_floor_ms / gpu_ms_step) if valu_floor_ms else None,
                "frac_of_hbm_floor": (hbm_floor_ms / gpu_ms_step) if hbm_floor_ms else None,
                "valu_class_mix_source": mix_rel,
                "traffic_ratio_other_profile": other_ratio,
                "valu_source": (sq_rel + " (SQ_INSTS_VALU per pair x pairs per launch / (%d CUs x %.1f GHz) / this run's "
                                         "kernel time)" % (N_CUS, CLOCK_HZ / 1e9)) if sq else None,
                "dominant_kernel": dict(kernels[dom], stage=dom) if dom else None,
                "kernels": kernels,
            },
        }
        if also:
            line["also"] = also
        if rccl:
            line["rccl"] = rccl
        if stub:
            line["stub"] = "TEST HOOK: CPU tensors and a no-op matcher; this line exercises the harness, it measures nothing"
        if world == 1 and not a.no_cpu_baseline and not stub:
            line["cpu_baseline"] = cpu_baseline(a, params)
        print(json.dumps(line))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
