"""BASELINE.json configs[2] (C3): N x 150 bp reads over an E. coli-scale genome, k=31, one GPU:
count -> deleteAll(<3) -> buildGraph -> removeBubbles -> simplifyGraph -> retainLargest, timed.
usage: python scripts/run_c3.py [reads=5000000] [genome=4600000] [err=0.005] [chunk_reads=0] [capacity_hint=0] [prefilter_distinct=0] [k=31] [path=auto] [opts]
chunk_reads = 0: ALL reads resident in HBM (39 B each), ONE gk_map_count_reads_dev call — the library cuts it into
batches itself; > 0: the round-1 form, reads regenerated chunk by chunk into one small buffer, one call per chunk.
prefilter_distinct > 0: two passes over the reads through the exact singleton pre-filter."""
import sys, time, json
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
if len(sys.argv) > 9 and sys.argv[9]:      # option switches live in the test build of the library
    __import__('os').environ.setdefault("GK_LIB_PATH", __import__('os').path.join(sys.path[0], "genome_amd", "libgenome_amd_test.so"))
from genome_amd import synth
from genome_amd.dnamap import Context, HipDNAMap
from genome_amd.graph import buildGraph

arg = lambda i, d, f=int: f(sys.argv[i]) if len(sys.argv) > i else d
n, G, err = arg(1, 5_000_000), arg(2, 4_600_000), arg(3, 0.005, float)
chunk, hint, pfd, k = arg(4, 0), arg(5, 0), arg(6, 0), arg(7, 31)
path = arg(8, "auto", str)
opts = arg(9, "", str)          # gk_ctx_set_option pairs: name=value,name=value (A/B switches of the insert pipeline)
L = 150
ctx = Context(0)
for kv in filter(None, opts.split(",")):
    name, val = kv.split("=")
    ctx.set_option(name, int(val))
stride = synth.record_stride(L)
resident = chunk == 0
if resident:
    chunk = n
d = ctx.alloc(chunk * stride + 64)
m = HipDNAMap(ctx, k, hint)
m.set_insert_path(path)
t = {}
occ = 0; tgen = 0.0
pf = None
phases = np.zeros(5)
kernel_ms = 0.0


def chunks():
    global tgen
    for first in range(0, n, chunk):
        c = min(chunk, n - first)
        if not resident or first == 0 and not chunks.done:
            g0 = time.perf_counter()
            ctx.synth_reads(d, c, L, "G", 3, first, G, err)
            ctx.sync()
            tgen += time.perf_counter() - g0
            chunks.done = resident
        yield c


chunks.done = False
if pfd:
    from genome_amd.prefilter import HipPrefilter
    pf = HipPrefilter(ctx, k, pfd)
    t0 = time.perf_counter(); tgen = 0.0
    for c in chunks():
        pf.add_reads_dev(d, c, L)
    t["prefilter_pass1_s"] = time.perf_counter() - t0 - tgen
admitted = 0
t0 = time.perf_counter(); tgen = 0.0
for c in chunks():
    if pf:
        o, a = pf.count_reads_dev(m, d, c, L); occ += o; admitted += a
    else:
        occ += m.count_reads_dev(d, c, L)
        phases += np.array(m.last_phase_ms()); kernel_ms += m.last_count_kernel()[0]
t["count_s"] = time.perf_counter() - t0 - tgen
distinct = m.size(); st = m.stats()
t0 = time.perf_counter(); m.deleteAll_lt(3); t["filter_s"] = time.perf_counter() - t0
good = m.size()
t0 = time.perf_counter(); g = buildGraph(k, m); t["build_s"] = time.perf_counter() - t0
c0 = g.counts(); bstats = g.buildStats()
t0 = time.perf_counter(); g.removeBubbles(); t["bubbles_s"] = time.perf_counter() - t0
c1 = g.counts()
t0 = time.perf_counter(); g.simplifyGraph(); t["simplify_s"] = time.perf_counter() - t0
c2 = g.counts()
t0 = time.perf_counter(); kept, comps = g.retainLargest(); t["retain_s"] = time.perf_counter() - t0
c3 = g.counts()
print(json.dumps({"reads": n, "genome": G, "err": err, "resident": resident, "occurrences": occ, "distinct_in_table": distinct, "good_kmers": good,
                  "prefilter": (dict(pf.stats(), admitted=admitted) if pf else None),
                  "table": {k_: st[k_] for k_ in ("slots", "grows", "partitioned_launches", "direct_launches", "spilled_keys", "failed_segments",
                                                   "retries_direct", "est_new_distinct_last_batch", "repeat_heavy")},
                  "count_kernel_ms": kernel_ms, "count_phases_ms": [round(float(x), 3) for x in phases],
                  "occ_per_s_count": occ / t["count_s"], "graph_built": c0, "build_stats": bstats, "after_bubbles": c1, "after_simplify": c2,
                  "components": comps, "largest": c3, "times": {k_: round(v, 4) for k_, v in t.items()}}))
# release everything explicitly: a profiler attached to this process waits for the streams to go away
g.close(); m.close()
if pf:
    pf.close()
ctx.free(d); ctx.close()
