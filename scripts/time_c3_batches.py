"""C3's count (6e9 windows, device-resident reads) against the size of a partitioned batch: every batch after the first streams the
whole table in and out again.  usage: python scripts/time_c3_batches.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genome_amd import synth
from genome_amd.dnamap import Context, HipDNAMap

N, L, k, G, err = 50_000_000, 150, 31, 4_600_000, 0.005
ctx = Context(0)
d = ctx.alloc(N * synth.record_stride(L) + 64)
ctx.synth_reads(d, N, L, "G", 3, 0, G, err)
m = HipDNAMap(ctx, k, 0)
for cap in (0, 1 << 31, 3_300_000_000, 6_100_000_000, 1 << 31):
    if cap:
        m.set_max_batch_keys(cap)
    best = None
    for rep in range(3):
        m.clear(); ctx.sync()
        t0 = time.perf_counter(); m.count_reads_dev(d, N, L); ctx.sync(); ms = (time.perf_counter() - t0) * 1e3
        if best is None or ms < best[0]:
            best = (ms, [round(x, 2) for x in m.last_phase_ms()])
    print(json.dumps({"max_batch_keys": cap or "default", "count_ms": round(best[0], 2), "phases_ms": best[1], "mem": ctx.mem_stats()}), flush=True)
