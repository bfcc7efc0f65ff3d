"""PCIe-inclusive rate at C2: the same 1e6 x 150 bp reads handed over as a HOST `.bin` buffer (gk_map_count_reads:
framing walk on the host, H2D copy, exact (histogram) pipeline) against the device-resident call."""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from genome_amd import synth
from genome_amd.dnamap import Context, HipDNAMap
ctx = Context(0)
n, L, k = 1_000_000, 150, 31
stride = synth.record_stride(L)
d = ctx.alloc(n * stride + 64)
ctx.synth_reads(d, n, L, "U", 2, 0, 0, 0.0)
host = ctx.download(d, n * stride).tobytes()
m = HipDNAMap(ctx, k, int(n * (L - k + 1) * 1.05))
for name, f in (("device-resident", lambda: m.count_reads_dev(d, n, L)), ("host buffer", lambda: m.count_reads(host, n))):
    best = 1e9
    for _ in range(5):
        m.clear(); ctx.sync(); t0 = time.perf_counter(); occ = f(); ctx.sync(); best = min(best, time.perf_counter() - t0)
    print(f"{name}: {best * 1e3:.2f} ms per {occ} windows = {m.size() / best:.3e} distinct k-mers/s; device pipeline phases {[round(x, 3) for x in m.last_phase_ms()]} ms")
