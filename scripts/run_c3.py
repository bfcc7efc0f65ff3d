"""BASELINE.json configs[2] rehearsal: N x 150 bp reads over an E. coli-scale genome, k=31, one GPU:
count (chunked) -> deleteAll(<3) -> buildGraph -> removeBubbles -> simplifyGraph -> retainLargest, timed.
usage: python scripts/run_c3.py [reads=5000000] [genome=4600000] [err=0.005] [chunk_reads=2000000] [capacity_hint=0] [prefilter_distinct=0] [k=31]
prefilter_distinct > 0: two passes over the (regenerated) chunks through the exact singleton pre-filter."""
import sys, time, json
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from genome_amd import synth
from genome_amd.dnamap import Context, HipDNAMap
from genome_amd.graph import buildGraph

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4_600_000
err = float(sys.argv[3]) if len(sys.argv) > 3 else 0.005
L = 150
k = int(sys.argv[7]) if len(sys.argv) > 7 else 31
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 2_000_000
hint = int(sys.argv[5]) if len(sys.argv) > 5 else 0
pfd = int(sys.argv[6]) if len(sys.argv) > 6 else 0
ctx = Context(0)
stride = synth.record_stride(L)
d = ctx.alloc(chunk * stride + 64)
m = HipDNAMap(ctx, k, hint)
t = {}
t0 = time.perf_counter(); occ = 0; tgen = 0.0
pf = None
if pfd:
    from genome_amd.prefilter import HipPrefilter
    pf = HipPrefilter(ctx, k, pfd)
    for first in range(0, n, chunk):
        c = min(chunk, n - first)
        g0 = time.perf_counter()
        ctx.synth_reads(d, c, L, "G", 3, first, G, err)
        tgen += time.perf_counter() - g0
        pf.add_reads_dev(d, c, L)
    t["prefilter_pass1_s"] = time.perf_counter() - t0 - tgen
    t0 = time.perf_counter(); tgen = 0.0
admitted = 0
for first in range(0, n, chunk):
    c = min(chunk, n - first)
    g0 = time.perf_counter()
    ctx.synth_reads(d, c, L, "G", 3, first, G, err)
    tgen += time.perf_counter() - g0
    if pf:
        o, a = pf.count_reads_dev(m, d, c, L); occ += o; admitted += a
    else:
        occ += m.count_reads_dev(d, c, L)
t["count_s"] = time.perf_counter() - t0 - tgen
distinct = m.size(); st = m.stats()
t0 = time.perf_counter(); m.deleteAll_lt(3); t["filter_s"] = time.perf_counter() - t0
good = m.size()
t0 = time.perf_counter(); g = buildGraph(k, m); t["build_s"] = time.perf_counter() - t0
c0 = g.counts()
t0 = time.perf_counter(); g.removeBubbles(); t["bubbles_s"] = time.perf_counter() - t0
c1 = g.counts()
t0 = time.perf_counter(); g.simplifyGraph(); t["simplify_s"] = time.perf_counter() - t0
c2 = g.counts()
t0 = time.perf_counter(); kept, comps = g.retainLargest(); t["retain_s"] = time.perf_counter() - t0
c3 = g.counts()
print(json.dumps({"reads": n, "genome": G, "err": err, "occurrences": occ, "distinct_in_table": distinct, "good_kmers": good,
                  "prefilter": (dict(pf.stats(), admitted=admitted) if pf else None),
                  "table": {k_: st[k_] for k_ in ("slots", "grows", "partitioned_launches", "direct_launches")},
                  "occ_per_s_count": occ / t["count_s"], "graph_built": c0, "after_bubbles": c1, "after_simplify": c2,
                  "components": comps, "largest": c3, "times": {k_: round(v, 4) for k_, v in t.items()}}))
