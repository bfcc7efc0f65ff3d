#!/usr/bin/env python3
"""Per-kernel memory traffic of the C3 flow (scripts/prof_c3_pmc.sh <tag>) against each kernel's algorithmic bytes:
profiles/r02/pmc_c3_<tag>.json.  FETCH_SIZE x2 (gfx950 correction, see summarize_pmc.py), *_SIZE in KiB; totals are per
kernel NAME over the whole flow (three count batches, one graph build)."""
import collections, csv, glob, json, os, sys
tag = sys.argv[1]
out_dir = sys.argv[2] if len(sys.argv) > 2 else "profiles/r03"
res = json.load(open(f"gpurun_out/{tag}_c3prof.json"))


def short(name):
    """k_classify<1, gk::Table<1, gk::Slot<1> > > -> k_classify<1>: the kernel and its key width (round 3 templated the kernels on the
    table / slot type as well; the C3 flow runs one instantiation of each)"""
    name = name.replace("void ", "").split("(")[0]
    if "<" not in name:
        return name
    base, args = name.split("<", 1)
    return f"{base}<{args.split(',')[0].split('>')[0].strip()}>"



def counters(kind):
    f = max(glob.glob(f"gpurun_out/{tag}_c3pmc_{kind}/*/*counter_collection.csv"), key=os.path.getmtime)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    return agg


stats = {}
f = max(glob.glob(f"gpurun_out/{tag}_c3prof/*/*kernel_stats.csv"), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    nm = short(r["Name"])
    c0, m0 = stats.get(nm, (0, 0.0))
    stats[nm] = (c0 + int(r["Calls"]), m0 + float(r["TotalDurationNs"]) / 1e6)
fetch, write, req = counters("fetch"), counters("write"), counters("req")
try:
    rdsz = counters("rdsz")          # optional fourth pass: TCC_EA0_RDREQ_{32B,64B,128B}_sum (scripts/runs/gpu_run33.sh)
except ValueError:
    rdsz = {}
occ, distinct, good = res["occurrences"], res["distinct_in_table"], res["good_kmers"]
slots_count = res["table"]["slots"]
walked = res["build_stats"]["walked_bases"]
# SURVEY.md §8(d) algorithmic bytes of the graph-phase kernels; the count pipeline's are in bench.py's c3 object
algo = {"k_classify<1>": ("80 B per live key", 80.0 * good), "k_walk_q<1>": ("9.25 B per walked base", 9.25 * walked),
        "k_filter_lt<1>": ("20 B per slot (12 scan + 8 tombstone)", 20.0 * slots_count),
        "k_rehash<1>": ("16 B read per old slot + 16 B written per survivor", 16.0 * slots_count + 16.0 * good),
        "k_compact_seg<1>": ("20 B per slot (SURVEY's figure for scan + tombstone; this pass reads 12 B per old slot and writes 16 B per new slot)", 20.0 * slots_count)}
doc = {"command": f"scripts/prof_c3_pmc.sh {tag}: rocprofv3 --kernel-trace --stats, then three separate --pmc passes, over scripts/run_c3.py (C3, resident reads)",
       "corrections": "FETCH_SIZE x2 (gfx950), FETCH/WRITE_SIZE in units of 1024 B; sums over all launches of a kernel in the flow",
       "reading": "every read request of every kernel here is a 128-byte line (rd_request_sizes), also for the random probes of k_classify / k_walk_q: "
                  "they are bound by the BYTES of those lines (moved_frac_of_hbm_peak), which is why their algorithmic fraction (16-byte slots) looks low; "
                  "allocating the table as uncached or fine-grained memory (option graph_mem) changes nothing",
       "random_load_ceiling_per_s": 48e9, "hbm_peak_GBs": 8000.0, "flow": {k: res[k] for k in ("occurrences", "distinct_in_table", "good_kmers", "times", "graph_built")},
       "kernels": {}}
for name, (calls, ms) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    if ms < 0.3 or name.startswith("__amd") or name.startswith("k_synth"):
        continue
    fb = 2048.0 * fetch.get(name, {}).get("FETCH_SIZE", 0.0)
    wb = 1024.0 * write.get(name, {}).get("WRITE_SIZE", 0.0)
    rq = req.get(name, {})
    k = {"calls": calls, "total_ms": ms, "fetch_bytes": fb, "write_bytes": wb, "traffic_GBs": (fb + wb) / ms / 1e6 if ms else 0,
         "rd_requests": rq.get("TCC_EA0_RDREQ_sum"), "wr_requests": rq.get("TCC_EA0_WRREQ_sum"), "atomic_requests": rq.get("TCC_EA0_ATOMIC_sum")}
    if rq.get("TCC_EA0_RDREQ_sum"):
        k["rd_requests_per_s"] = rq["TCC_EA0_RDREQ_sum"] / (ms * 1e-3)
    k["moved_frac_of_hbm_peak"] = (fb + wb) / (ms * 1e-3) / 8e12          # what the memory system actually carried
    z = rdsz.get(name, {})
    if z.get("TCC_EA0_RDREQ_sum"):
        k["rd_request_sizes"] = {s_: z.get(f"TCC_EA0_RDREQ_{s_}_sum", 0.0) / z["TCC_EA0_RDREQ_sum"] for s_ in ("32B", "64B", "128B")}
    if name in algo:
        k["algorithmic"] = algo[name][0]
        k["algorithmic_bytes"] = algo[name][1]
        k["roofline_frac_of_hbm_peak"] = algo[name][1] / (ms * 1e-3) / 8e12
        k["traffic_over_algorithmic"] = (fb + wb) / algo[name][1]
    doc["kernels"][name] = k
json.dump(doc, open(f"{out_dir}/pmc_c3_{tag}.json", "w"), indent=1)
for n, k in doc["kernels"].items():
    print(f"{n[:44]:44s} {k['calls']:3d}x {k['total_ms']:8.2f} ms  fetch {k['fetch_bytes'] / 1e9:7.2f} GB  write {k['write_bytes'] / 1e9:7.2f} GB  {k['traffic_GBs']:7.0f} GB/s"
          + (f"  rd {k['rd_requests_per_s'] / 1e9:5.1f} G/s" if k.get("rd_requests_per_s") else "") + (f"  frac {k['roofline_frac_of_hbm_peak']:.3f}" if "roofline_frac_of_hbm_peak" in k else ""))
