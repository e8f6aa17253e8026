"""Issue-cost estimate of the largest loop of selected kernels in a hipcc -S listing (gfx950).

The two VALU classes are the measured ones of tools/microtests/valu_rate.hip: the plain 32-bit VOP1/VOP2 forms
(v_add/sub/subrev_u32, v_and/or/xor/not_b32, v_lshrrev_b32, v_ashrrev_i32, v_mov_b32, unpacked v_min/add/sub_u16,
v_add/mul/fmac/fma_f32) and v_bitop3_b32 issue at ~1.55 wave-instructions per cycle and CU ("fast": 4 / 1.55 = 2.58 SIMD
cycles each), everything else -- v_pk_*, v_perm, dot, DPP / SDWA forms, min/max, three-operand integer forms -- at ~0.9
("slow": 4.44 cycles).
usage: isa_cost.py file.s name_substring [name_substring ...] [--rows N]   (N = loop iterations' worth of rows, default 1)
"""
import collections, re, sys

FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_lshrrev_b32",
        "v_ashrrev_i32", "v_mov_b32", "v_min_u16", "v_max_u16", "v_add_u16", "v_sub_u16", "v_add_f32", "v_mul_f32",
        "v_fmac_f32", "v_fma_f32", "v_bitop3_b32"}
CYC_FAST, CYC_SLOW = 4 / 1.55, 4 / 0.9


def classify(op):
    if not op.startswith("v_"):
        return None
    if op.endswith(("_dpp", "_sdwa")):
        return "slow"
    base = re.sub(r"_e(32|64)$", "", op)
    if base in FAST and not op.endswith("_e64"):
        return "fast"
    if base == "v_bitop3_b32" or base == "v_fma_f32":
        return "fast"
    return "slow"


def loops_of(txt, names):
    starts = [(i, l.split(":")[0]) for i, l in enumerate(txt) if l.startswith("_Z") and ": " in l and "@" in l]
    for n, (i, name) in enumerate(starts):
        if not any(k in name for k in names):
            continue
        end = starts[n + 1][0] if n + 1 < len(starts) else len(txt)
        seq, labels = [], {}
        for l in txt[i + 1:end]:
            l = l.strip()
            m0 = re.match(r"(\.LBB\w+):", l)
            if m0:
                labels[m0.group(1)] = len(seq)
            elif l and not l.startswith((".", ";")) and not l.endswith(":"):
                seq.append(l)
        loops = []
        for j, l in enumerate(seq):
            m = re.match(r"s_c?branch\w* (\.LBB\S+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] <= j:
                loops.append((labels[m.group(1)], j))
        loops.sort(key=lambda s: s[0] - s[1])
        meta = {}
        for l in txt[i:end]:
            m = re.match(r"\s*;\s*(NumVgprs|ScratchSize|Occupancy|LDSByteSize|NumSgprs):\s*(\d+)", l)
            if m:
                meta[m.group(1)] = int(m.group(2))
        yield name, seq, loops, meta


def write_mix(listing, out_path, names):
    """profiles/rNN_isa_valu_mix.json: per kernel, the share of the main loop's VALU instructions in each issue class --
    what bench.py prices a kernel's SQ_INSTS_VALU with (roofline.kernels.*.valu_floor_ms)."""
    import json
    import subprocess
    txt = open(listing).read().split("\n")
    doc = {"source": "hipcc -O3 --offload-arch=gfx950 -S of calibrating_amd/csrc/sgbm.hip; largest loop of each kernel; "
                     "classes and rates: tools/microtests/valu_rate.hip (plain 32-bit VOP1/VOP2 + v_bitop3 ~1.55 per cycle and "
                     "CU, packed / DPP / SDWA / three-operand / min-max ~0.9)", "kernels": {}}
    for name, seq, loops, meta in loops_of(txt, names):
        if not loops:
            continue
        lo, hi = loops[0]
        c = collections.Counter(classify(l.split()[0]) for l in seq[lo:hi + 1])
        n = c["fast"] + c["slow"]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"^void ", "", dem.split("(")[0])
        doc["kernels"][dem] = {"valu_in_loop": n, "fast_frac": c["fast"] / n, "slow_frac": c["slow"] / n,
                               "vgprs": meta.get("NumVgprs"), "scratch_bytes": meta.get("ScratchSize")}
    json.dump(doc, open(out_path, "w"), indent=1)
    for k, v in doc["kernels"].items():
        print("%-60s plain %.3f  packed-class %.3f  (%d VALU in the loop, %s VGPRs)" % (k, v["fast_frac"], v["slow_frac"],
                                                                                      v["valu_in_loop"], v["vgprs"]))


if __name__ == "__main__":
    args = sys.argv[1:]
    if "--mix" in args:  # isa_cost.py listing.s --mix out.json name_substring...
        k = args.index("--mix")
        out = args[k + 1]
        del args[k:k + 2]
        write_mix(args[0], out, args[1:])
        sys.exit(0)
    rows = 1
    if "--rows" in args:
        k = args.index("--rows")
        rows = int(args[k + 1])
        del args[k:k + 2]
    txt = open(args[0]).read().split("\n")
    for name, seq, loops, meta in loops_of(txt, args[1:]):
        print(name[:90], meta)
        if not loops:
            continue
        lo, hi = loops[0]
        body = seq[lo:hi + 1]
        c = collections.Counter()
        ops = {"fast": collections.Counter(), "slow": collections.Counter()}
        other = collections.Counter()
        for l in body:
            op = l.split()[0]
            cl = classify(op)
            if cl:
                c[cl] += 1
                ops[cl][op] += 1
            else:
                other["lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_"))
                      else "s_nop" if op == "s_nop" else "s_waitcnt" if op == "s_waitcnt" else "salu"] += 1
        cyc = c["fast"] * CYC_FAST + c["slow"] * CYC_SLOW
        print("  loop of %d instr / %d rows: per row  VALU %.1f (fast %.1f, slow %.1f) = %.0f issue cycles;  %s" % (
            len(body), rows, (c["fast"] + c["slow"]) / rows, c["fast"] / rows, c["slow"] / rows, cyc / rows,
            {k: round(v / rows, 1) for k, v in other.items()}))
        for cl in ("slow", "fast"):
            print("   %s:" % cl, ", ".join("%s %.1f" % (o, n / rows) for o, n in ops[cl].most_common(14)))
