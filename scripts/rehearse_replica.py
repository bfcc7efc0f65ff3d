"""Rehearsal of an 8-GPU configuration's REPLICA at full size on one MI355X: >= 1.5e9 solid 55-mers (C4's whole k-mer set;
BASELINE.json configs[3]) counted from error-free synthetic reads, gathered through the real collective (a communicator of
one rank: chunked export -> insert into a table sized by graph_table_load; in its classified form, so the first buildGraph takes the
degree masks that came with the keys and the second one classifies by itself), then Graph.buildGraph -> removeBubbles ->
simplifyGraph -> retainLargest, with the context's device-memory high-water mark per stage.

    python scripts/rehearse_replica.py [--k 55] [--genome 1500000000] [--coverage 6] [--out gpurun_out/replica_c4.json]

What it shows: DESIGN.md section 6's bytes-per-live-key table holds on a table of the real size (the pointer-jumping state
by live-key rank, node ids in the slot annotation, 24-byte slots, bounded gather staging).  Not a parity test: the graph's
invariants are checked on a sample, the two unitig constructions must agree (gk_graph_checksum).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GK_LIB_PATH", os.path.join(sys.path[0], "genome_amd", "libgenome_amd_test.so"))      # (graph_unitigs is a test-build switch)

from genome_amd import synth                                   # noqa: E402
from genome_amd.dist import DistDNAMap, HipDist, unique_id     # noqa: E402
from genome_amd.dnamap import Context                          # noqa: E402
from genome_amd.graph import buildGraph                        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=55)
    ap.add_argument("--genome", type=int, default=1_500_000_000)
    ap.add_argument("--coverage", type=float, default=6.0)
    ap.add_argument("--chunk", type=int, default=10_000_000)
    ap.add_argument("--both", type=int, default=1, help="1: build the graph in both unitig constructions and compare")
    ap.add_argument("--classified", type=int, default=1, help="1: the gather in its classified form — the FIRST build then takes the owners' masks, "
                    "the second classifies by itself, and the two graphs must be the same")
    ap.add_argument("--out", default="gpurun_out/replica.json")
    a = ap.parse_args()
    k, G, L_ = a.k, a.genome, 150
    n = int(G * a.coverage / L_)
    ctx = Context(0)
    stride = synth.record_stride(L_)
    d = ctx.alloc(a.chunk * stride + 64)
    rep = {"k": k, "genome": G, "reads": n, "read_len": L_}
    dist = HipDist(ctx, 0, 1, unique_id())
    dm = DistDNAMap(dist, k, int(G * 1.02))
    t0 = time.perf_counter()
    occ = 0
    for first in range(0, n, a.chunk):
        c = min(a.chunk, n - first)
        ctx.synth_reads(d, c, L_, "G", 44, first, G, 0.0)
        sent, owned = dm.count_reads_dev(d, c, L_)
        occ += owned
        st = dm.local.stats()
        print(f"counted {first + c} reads, {dm.local.size()} keys", {k_: st[k_] for k_ in ("slots", "partitioned_launches", "direct_launches", "spilled_keys",
              "failed_segments", "retries_direct", "repeat_heavy", "grows")}, dm.local.last_phase_ms(), flush=True)
    rep["count_s"] = time.perf_counter() - t0
    rep["windows"], rep["keys"] = occ, dm.size()
    ctx.free(d)
    local_bytes = dm.local.slots() * dm.local.stats()["slot_bytes"]
    ctx.trim()
    base = ctx.mem_stats(reset_peak=True)["live"] - local_bytes
    t0 = time.perf_counter()
    full = dm.gathered(classified=bool(a.classified))      # (one rank: every neighbour is local; the masks travel into the replica)
    rep["gather_s"] = time.perf_counter() - t0
    rep["gather_classified"] = bool(a.classified)
    rep["gather_peak_bytes"] = ctx.mem_stats()["peak"] - base          # local partition + replica + staging
    rep["local_table_bytes"] = local_bytes
    assert full.verify_checksum() == dm.local.verify_checksum()
    rep["replica_slots"], rep["replica_load"] = full.slots(), rep["keys"] / full.slots()
    dm.close(); dist.close(); ctx.trim()
    base = ctx.mem_stats(reset_peak=True)["live"] - full.slots() * full.stats()["slot_bytes"]
    print("gathered", rep, flush=True)
    out = {}
    modes = ((1, "walk"), (2, "pj")) if a.both else ((0, "auto"),)
    g = None
    for mode, name in modes:
        ctx.set_option("graph_unitigs", mode)
        t0 = time.perf_counter()
        g = buildGraph(k, full)
        rep[name + "_build_s"] = time.perf_counter() - t0
        rep[name + "_build"] = g.buildStats()
        if a.both and a.classified:
            assert rep[name + "_build"]["classified_by_owners"] == (mode == 1)      # the masks serve the first build only
        out[name] = (g.counts(), g.checksum())
        rep[name + "_build_peak_bytes"] = ctx.mem_stats(reset_peak=True)["peak"] - base
        print(name, out[name], rep[name + "_build_s"], flush=True)
        if mode == 1:
            g.close()
            ctx.mem_stats(reset_peak=True)
    if a.both:
        assert out["walk"] == out["pj"], out
    rep["nodes"], rep["edges"], rep["edge_bases"] = list(out.values())[-1][0]
    t0 = time.perf_counter()
    g.removeBubbles(); g.simplifyGraph()
    kept, comps = g.retainLargest()
    rep["simplify_retain_s"] = time.perf_counter() - t0
    rep["components"], rep["kept_nodes"], rep["final"] = comps, kept, g.counts()
    rep["simplify_retain_peak_bytes"] = ctx.mem_stats()["peak"] - base
    peaks = [v for key, v in rep.items() if key.endswith("_peak_bytes")]
    rep["peak_bytes"] = max(peaks)
    rep["peak_bytes_per_key"] = max(peaks) / rep["keys"]
    g.close(); full.close(); ctx.close()
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
