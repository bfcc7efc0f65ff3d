"""A table beyond 34 GB (512 / 1024 L1 buckets): one big batch of distinct k-mers through the partitioned pipeline and through
the direct path.  usage: time_big_table.py [reads (default 16e6)] [k (31)]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genome_amd import synth
from genome_amd.dnamap import Context, HipDNAMap

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 16_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
L = 150
nk = L - k + 1
ctx = Context(0)
d = ctx.alloc(n * synth.record_stride(L) + 64)
ctx.synth_reads(d, n, L, "U", 9, 0, 0, 0.0)
out = {"reads": n, "k": k, "windows": n * nk}
# "partitioned_one_batch": gk_map_set_max_batch_keys lifts the default cap on a batch's key scratch (16 GiB per buffer) so that the
# whole call is ONE batch — the table is then streamed once (written from empty) instead of once per batch (in and out again)
for path in ("partitioned", "partitioned_one_batch", "direct"):
    m = HipDNAMap(ctx, k, int(n * nk * 1.05))
    st = m.stats()
    m.set_insert_path("partitioned" if path.startswith("partitioned") else path)
    if path == "partitioned_one_batch":
        if n * nk * (8 if k <= 31 else 16) <= (16 << 30):
            m.close()
            continue                                    # (already one batch)
        m.set_max_batch_keys(n * nk)
    res = []
    for rep in range(2):
        m.clear()
        ctx.sync()
        t0 = time.perf_counter()
        occ = m.count_reads_dev(d, n, L)
        ctx.sync()
        res.append((time.perf_counter() - t0) * 1e3)
    s2 = m.stats()
    out[path] = {"table_GB": st["slots"] * st["slot_bytes"] / 1e9, "ms": [round(x, 1) for x in res], "ps_per_window": round(min(res) * 1e9 / (n * nk), 1),
                 "phases_ms": [round(x, 2) for x in m.last_phase_ms()], "partitioned_launches": s2["partitioned_launches"], "direct_launches": s2["direct_launches"],
                 "size": m.size(), "verify": m.verify()}
    m.close()
    ctx.trim()
print(json.dumps(out))
