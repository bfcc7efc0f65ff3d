"""Partitioned vs direct insert at batch sizes whose key buffer is large enough (>= 1.5 GB) to be interleaved (l1_slot)."""
import sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from genome_amd import synth
from genome_amd.dnamap import Context, HipDNAMap
ctx = Context(0)
for (k, n, L, mode, G, e) in ((31, 2500000, 150, "U", 0, 0.0), (55, 1300000, 150, "G", 3000000, 0.01), (31, 3000000, 120, "G", 50000000, 0.005), (64, 1500000, 160, "U", 0, 0.0)):
    d = ctx.alloc(n * synth.record_stride(L) + 64)
    ctx.synth_reads(d, n, L, mode, 21, 0, G, e)
    res = []
    for path in ("direct", "partitioned"):
        m = HipDNAMap(ctx, k, n * (L - k + 1)); m.set_insert_path(path)
        m.count_reads_dev(d, n, L)
        st = m.stats()
        lo, hi, cnt = m.sorted_items()
        res.append((lo, hi, cnt))
        print(k, n, mode, path, "distinct", len(lo), "sum", int(cnt.astype(np.int64).sum()), {x: st[x] for x in ("partitioned_launches", "direct_launches", "spilled_keys", "retries_direct")}, flush=True)
        m.close()
    assert all(np.array_equal(a, b) for a, b in zip(res[0], res[1])), "MISMATCH"
    ctx.free(d)
print("striped check ok")
