"""Phase timers of a -DGK_TIMERS build (scripts/sweep_variants.sh style): C2 count pass x steps, then the
per-phase wall-clock ticks (100 MHz) summed over workgroups, as microseconds per workgroup-tile."""
import ctypes as C, sys
sys.path.insert(0, '.')
from genome_amd import synth, _lib
from genome_amd.dnamap import Context, HipDNAMap
n, L, k, steps = 1_000_000, 150, 31, 5
ctx = Context(0)
d = ctx.alloc(n * synth.record_stride(L) + 64)
ctx.synth_reads(d, n, L, "U", 2, 0, 0, 0.0)
m = HipDNAMap(ctx, k, int(n * (L - k + 1) * 1.05))
lib = _lib.lib()
lib.gk_debug_timers.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 16)()
for it in range(2):
    m.clear(); m.count_reads_dev(d, n, L)
lib.gk_debug_timers(buf, 1)
for it in range(steps):
    m.clear(); m.count_reads_dev(d, n, L)
print("phase_ms", m.last_phase_ms())
lib.gk_debug_timers(buf, 1)
t = [buf[i] / steps for i in range(16)]
tot = sum(t[:8])
names = ["zero+stage+barrier", "extract (wave 0)", "barrier after extract", "reserve+barrier", "write-out (wave 0)", "loop-top barrier"]
print("ticks per step summed over WGs:", [int(x) for x in t[:8]])
for i, nm in enumerate(names):
    print(f"  {nm:24s} {100 * t[i] / tot:5.1f} %")
