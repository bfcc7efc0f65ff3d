import sys
sys.path.insert(0, '.')
from genome_amd import synth
from genome_amd.dnamap import Context, HipDNAMap
ctx = Context(0)
for n in (20000, 1000000):
    L, k = 150, 31
    d = ctx.alloc(n * synth.record_stride(L) + 64)
    ctx.synth_reads(d, n, L, "U", 2, 0, 0, 0.0)
    m = HipDNAMap(ctx, k, int(n * 120 * 1.05))
    m.set_insert_path("partitioned")
    occ = m.count_reads_dev(d, n, L)
    st = m.stats()
    print(n, occ, m.size(), {k_: st[k_] for k_ in ("slots", "spilled_keys", "failed_segments", "retries_direct", "partitioned_launches", "direct_launches")}, flush=True)
    m.close(); ctx.free(d)
