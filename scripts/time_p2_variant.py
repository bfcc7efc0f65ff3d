"""P2 phase time of a variant library (GK_LIB_PATH) whose results may be wrong downstream: a fresh map per step, the
L1-scatter phase read from the first (partitioned) attempt."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genome_amd import synth
from genome_amd.dnamap import Context, HipDNAMap
n, L, k = 1_000_000, 150, 31
ctx = Context(0)
d = ctx.alloc(n * synth.record_stride(L) + 64)
ctx.synth_reads(d, n, L, "U", 2, 0, 0, 0.0)
for i in range(6):
    m = HipDNAMap(ctx, k, int(n * 120 * 1.05))
    m.set_insert_path("partitioned")
    try:
        m.count_reads_dev(d, n, L)
    except Exception as e:
        print("count failed (expected for a timing-only build):", str(e)[:80])
    print("phases ms", [round(x, 3) for x in m.last_phase_ms()], m.stats()["retries_direct"])
    m.close()
