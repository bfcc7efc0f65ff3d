"""C3's count phase twice on the same map (the way bench.py's c3 object does: a warm-up pass, then the timed one), with the
library's phase timers and host-gap timer of each pass — why is the second pass slower than a cold single pass?"""
import sys, time, json
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from genome_amd import synth
from genome_amd.dnamap import Context, HipDNAMap
N, L, k, G, err = 50_000_000, 150, 31, 4_600_000, 0.005
ctx = Context(0)
d = ctx.alloc(N * synth.record_stride(L) + 64)
ctx.synth_reads(d, N, L, "G", 3, 0, G, err); ctx.sync()
m = HipDNAMap(ctx, k, 0)
for p in range(3):
    m.clear()
    t0 = time.perf_counter(); occ = m.count_reads_dev(d, N, L); dt = time.perf_counter() - t0
    st = m.stats()
    print(f"pass {p}: {dt * 1e3:.1f} ms wall, kernel {m.last_count_kernel()[0]:.1f} ms, phases {[round(x, 1) for x in m.last_phase_ms()]}, "
          f"slots {st['slots']}, grows {st['grows']}, launches {st['partitioned_launches']}, host gap {st.get('last_count_host_gap_ms')}")
    m.deleteAll_lt(3)
m.close(); ctx.free(d); ctx.close()
