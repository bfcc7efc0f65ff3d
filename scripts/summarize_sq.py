#!/usr/bin/env python3
"""Per-kernel averages of the SQ / LDS counter passes of scripts/prof_sq.sh <tag>: one row per pipeline kernel and counter,
plus a few ratios (instructions per key, LDS-pipe and VALU-pipe activity against busy cycles).  usage: summarize_sq.py <tag> [out.txt]"""
import collections, csv, glob, os, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in "abc":
    fs = glob.glob(f"gpurun_out/{tag}_sq_{p}/*/*counter_collection.csv")
    if not fs:
        continue
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        name = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if name.startswith(("k_op_scatter1", "k_part_scatter2", "k_seg_insert")):
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = []
KEYS = 1.2e8
for name, cs in agg.items():
    a = {c: sum(v) / len(v) for c, v in cs.items()}
    lines.append(name)
    lines.append("   " + "  ".join(f"{c}={a[c]:.4g}" for c in sorted(a)))
    r = []
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
        if c in a:
            r.append(f"{c[9:]} per 64 keys = {a[c] / (KEYS / 64):.1f}")
    if "SQ_BUSY_CYCLES" in a:
        for c in ("SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_ANY"):
            if c in a and a.get("SQ_WAVE_CYCLES"):
                r.append(f"{c[3:]} / WAVE_CYCLES = {a[c] / a['SQ_WAVE_CYCLES']:.3f}")
    if "SQ_LDS_BANK_CONFLICT" in a and a.get("SQ_LDS_IDX_ACTIVE"):
        r.append(f"LDS_BANK_CONFLICT / LDS_IDX_ACTIVE = {a['SQ_LDS_BANK_CONFLICT'] / a['SQ_LDS_IDX_ACTIVE']:.3f}")
    lines.append("   " + "; ".join(r))
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(f"# SQ / LDS counters per launch of the headline pipeline's kernels (scripts/prof_sq.sh {tag}; averages over the timed launches)\n" + out + "\n")
