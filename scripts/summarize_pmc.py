#!/usr/bin/env python3
"""Turn the PMC passes of scripts/profile_round.sh <tag> (gpurun_out/<tag>_pmc_{fetch,write,req}) into
profiles/r02/pmc_pipeline_<tag>.json (the file bench.py reads `roofline.traffic` from) and copy the raw
counter CSVs, the kernel stats and the bench line next to it.  Corrections: FETCH_SIZE x2 (gfx950 tallies
128-B read requests at 64 B; calibrated on k_part_hist2 in v3/v4), WRITE_SIZE as reported; both are
counted in KiB-like units of 1024 B by rocprofv3 (`*_SIZE` counters are KB)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
note = sys.argv[2] if len(sys.argv) > 2 else ""
out_dir = sys.argv[3] if len(sys.argv) > 3 else "profiles/r02"
PIPE = ("k_op_scatter1_reads", "k_part_hist1_reads", "k_part_scatter1_reads", "k_part_hist2", "k_part_scatter2", "k_seg_insert")


def load(kind):
    f = max(glob.glob(f"gpurun_out/{tag}_pmc_{kind}/*/*counter_collection.csv"), key=os.path.getmtime)      # newest run of that tag
    shutil.copy(f, f"{out_dir}/pmc_{tag}_{kind}_counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void ", "").split("(")[0]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


fetch, write, req = load("fetch"), load("write"), load("req")
per_kernel, total = {}, 0.0
for name in sorted(set(fetch) | set(write)):
    if not name.startswith(PIPE):
        continue
    f_raw = fetch.get(name, {}).get("FETCH_SIZE", 0.0) * 1024.0
    w = write.get(name, {}).get("WRITE_SIZE", 0.0) * 1024.0
    rq = req.get(name, {})
    per_kernel[name] = {"FETCH_SIZE_bytes_raw": f_raw, "fetch_bytes_corrected_x2": 2 * f_raw, "WRITE_SIZE_bytes": w,
                        "rd_requests": rq.get("TCC_EA0_RDREQ_sum"), "wr_requests": rq.get("TCC_EA0_WRREQ_sum"),
                        "atomic_requests": rq.get("TCC_EA0_ATOMIC_sum")}
    total += 2 * f_raw + w
doc = {"command": f"scripts/profile_round.sh {tag} auto  (rocprofv3 --pmc <group> --kernel-trace, one pass per group: FETCH_SIZE | "
                  "WRITE_SIZE | TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum; bench.py --steps 3 --warmup 1)",
       "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B read requests at 64 B for coalesced streams); WRITE_SIZE exact",
       "per_kernel": per_kernel, "hbm_bytes_per_launch": total, "note": note}
json.dump(doc, open(f"{out_dir}/pmc_pipeline_{tag}.json", "w"), indent=1)
ks = glob.glob(f"gpurun_out/{tag}_stats/*/*kernel_stats.csv")
if ks:
    shutil.copy(max(ks, key=os.path.getmtime), f"{out_dir}/bench_n1_{tag}_kernel_stats.csv")
shutil.copy(f"gpurun_out/{tag}_bench.json", f"{out_dir}/bench_n1_{tag}.json")
print(json.dumps({k: {a: b for a, b in v.items() if "bytes" in a} for k, v in per_kernel.items()}, indent=1), total)
