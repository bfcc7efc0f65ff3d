"""Which insert path runs, and with what side effects (growth, spills, failed segments, hand-backs), for one batch of
device-resident reads at several capacity hints; both paths must give the same table."""
import sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from genome_amd import synth
from genome_amd.dnamap import Context, HipDNAMap
ctx = Context(0)
n, L, k = 50000, 150, 31
d = ctx.alloc(n * synth.record_stride(L) + 64)
ctx.synth_reads(d, n, L, "U", 9, 0, 0, 0.0)
ref = None
for path in ("direct", "partitioned"):
    for hint in (64, 100000, 3000000):
        m = HipDNAMap(ctx, k, hint); m.set_insert_path(path)
        m.count_reads_dev(d, n, L)
        st = m.stats()
        it = m.sorted_items()
        if ref is None: ref = it
        ok = all(np.array_equal(a, b) for a, b in zip(ref, it))
        print(path, hint, "ok" if ok else "MISMATCH", {k_: st[k_] for k_ in ("slots", "grows", "partitioned_launches", "direct_launches", "spilled_keys", "failed_segments", "retries_direct")})
        m.close()
