"""Phase timers of a -DGK_TIMERS build of the library (csrc/gk_internal.h): C2 reads, k=31.
  P2 (k_op_scatter1_reads, via count_reads_dev) and k_skm_route (via shard_superkmers, P=8).
Prints each phase's share of thread 0's wall clock, summed over workgroups.
usage (GPU box): build a timers variant into genome_amd/variants/ and run scripts/sweep_timers.sh <name>"""
import ctypes as C, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from genome_amd import synth, _lib
from genome_amd.dnamap import Context, HipDNAMap, skm_slot_bytes
n, L, k, steps = 1_000_000, 150, 31, 5
ctx = Context(0)
for o in sys.argv[1:]:                      # name=value context options (gk_ctx_set_option), e.g. p2_sorted=1
    nm, val = o.split("=")
    ctx.set_option(nm, int(val))
print("options:", sys.argv[1:])
d = ctx.alloc(n * synth.record_stride(L) + 64)
ctx.synth_reads(d, n, L, "U", 2, 0, 0, 0.0)
lib = _lib.lib()
buf = (C.c_ulonglong * 16)()


def report(fn, names, base=0, reset=1):
    fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    fn(buf, reset)
    t = [buf[base + i] / steps for i in range(8)]
    tot = sum(t) or 1
    for i, nm in enumerate(names):
        print(f"  {nm:28s} {100 * t[i] / tot:5.1f} %   ({t[i] / 100:.0f} us summed over workgroups per step)")


m = HipDNAMap(ctx, k, int(n * (L - k + 1) * 1.05))
for it in range(2):
    m.clear(); m.count_reads_dev(d, n, L)
lib.gk_debug_timers_partition(buf, 1)
for it in range(steps):
    m.clear(); m.count_reads_dev(d, n, L)
print("P2 k_op_scatter1_reads; phase_ms", [round(x, 3) for x in m.last_phase_ms()])
report(lib.gk_debug_timers_partition, ["zero+stage+barrier", "extract (wave 0)", "barrier after extract", "reserve+barrier", "write-out (wave 0)", "loop-top barrier"], 0, 0)
print("P4 k_part_scatter2")
report(lib.gk_debug_timers_partition, ["locate+zero bins+barrier", "load keys + LDS ranks", "barrier", "reserve atomics + barrier", "scan", "place + barrier", "write-out (wave 0)", "final barrier"], 8, 1)

P = 8
cap = n * 24 // P * P
out = ctx.alloc(cap * skm_slot_bytes(k))
for it in range(2):
    ctx.shard_superkmers(k, d, n, L, P, out, cap)
lib.gk_debug_timers_skm(buf, 1)
for it in range(steps):
    ctx.shard_superkmers(k, d, n, L, P, out, cap)
print("k_skm_route<16>, P=8")
report(lib.gk_debug_timers_skm, ["stage+barrier", "ph.1 window min + runs", "barrier+reserve", "phase 2 (records)", "loop-top barrier", "ph.1 m-mer scores", "ph.1 min of 8 per position"])
