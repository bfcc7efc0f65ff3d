"""The host-fed count (gk_map_count_reads from a pinned `.bin` buffer) a few times: run under
rocprofv3 --kernel-trace --memory-copy-trace to see how the sub-chunk uploads and the L1 scatter interleave."""
import sys, time, ctypes as C
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from genome_amd import synth
from genome_amd.dnamap import Context, HipDNAMap
ctx = Context(0)
n, L, k = 1_000_000, 150, 31
stride = synth.record_stride(L)
d = ctx.alloc(n * stride + 64)
ctx.synth_reads(d, n, L, "U", 2, 0, 0, 0.0)
hb = ctx.host_alloc(n * stride)
hb[:] = ctx.download(d, n * stride)
m = HipDNAMap(ctx, k, int(n * (L - k + 1) * 1.05))
for it in range(6):
    m.clear(); ctx.sync(); t0 = time.perf_counter(); m.count_reads(hb, n); ctx.sync()
    print(f"step {it}: {(time.perf_counter() - t0) * 1e3:.3f} ms, phases {[round(x, 3) for x in m.last_phase_ms()]}")
