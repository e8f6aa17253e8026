"""Instruction mix of the largest loop of selected kernels in a hipcc -S listing.
usage: isa_loop_stats.py file.s name_substring [name_substring ...]"""
import collections, re, sys
txt = open(sys.argv[1]).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(txt) if l.startswith("_Z") and ": " in l and "@" in l]
for n, (i, name) in enumerate(starts):
    if not any(k in name for k in sys.argv[2:]):
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
        if l.startswith("s_endpgm"):
            pass
    loops = []
    for j, l in enumerate(seq):
        m = re.match(r"s_c?branch\w* (\.LBB\S+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] <= j:
            loops.append((labels[m.group(1)], j))
    loops.sort(key=lambda s: s[0] - s[1])
    print(name[:70], "instrs", len(seq))
    for lo, hi in loops[:2]:
        body = seq[lo:hi + 1]
        c = collections.Counter()
        for l in body:
            op = l.split()[0]
            c["valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_")
              else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "other"] += 1
        ops = collections.Counter(l.split()[0] for l in body if l.startswith("v_"))
        print("  loop", hi - lo + 1, dict(c))
        print("   ", ops.most_common(12))
