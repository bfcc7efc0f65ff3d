"""What one device allocation costs by size on this card: gk_dev_alloc (= hipMalloc behind the context's block pool, which is
trimmed before every measurement) and the free that follows, GiB by GiB.  The default batch of the partitioned insert is bounded by
its key scratch because of this curve.  usage: python scripts/time_alloc.py [sizes in GiB ...]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genome_amd.dnamap import Context

sizes = [float(x) for x in sys.argv[1:]] or [1, 4, 8, 16, 24, 32, 48, 64, 96]
ctx = Context(0)
out = []
for rep in range(2):
    for g in sizes:
        ctx.trim(); ctx.sync()
        t0 = time.perf_counter(); p = ctx.alloc(int(g * (1 << 30))); ctx.sync(); t1 = time.perf_counter()
        ctx.free(p); ctx.trim(); ctx.sync(); t2 = time.perf_counter()
        out.append({"rep": rep, "GiB": g, "alloc_ms": round((t1 - t0) * 1e3, 2), "free_ms": round((t2 - t1) * 1e3, 2)})
        print(json.dumps(out[-1]), flush=True)
# two buffers of 31 GiB next to a 70 GiB block (the k = 55 one-batch case)
ctx.trim()
big = ctx.alloc(70 << 30)
t0 = time.perf_counter(); a = ctx.alloc(31 << 30); b = ctx.alloc(31 << 30); ctx.sync(); t1 = time.perf_counter()
print(json.dumps({"what": "2 x 31 GiB beside a live 70 GiB block", "alloc_ms": round((t1 - t0) * 1e3, 2)}), flush=True)
ctx.free(a); ctx.free(b); ctx.free(big); ctx.trim()
