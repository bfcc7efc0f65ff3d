"""C2 mode G (1e6 x 150 bp over 5 Mbp, 1 % error), k=31: plain count + filter_lt(3) vs the exact two-pass
singleton pre-filter; table sizes and wall times."""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from genome_amd import synth
from genome_amd.dnamap import Context, HipDNAMap
from genome_amd.prefilter import HipPrefilter
ctx = Context(0)
n, L, k, G, e = 1_000_000, 150, 31, 5_000_000, 0.01
d = ctx.alloc(n * synth.record_stride(L) + 64)
ctx.synth_reads(d, n, L, "G", 2, 0, G, e)
def t(f, reps=3):
    best = 1e9
    for _ in range(reps):
        ctx.sync(); t0 = time.perf_counter(); r = f(); ctx.sync(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, r
m = HipDNAMap(ctx, k, n * (L - k + 1))
def plain():
    m.clear(); m.count_reads_dev(d, n, L); return m.size()
ms_plain, distinct = t(plain)
slots_plain = m.slots()
m.deleteAll_lt(3); kept = m.size()
print(f"plain: {ms_plain:.2f} ms, distinct {distinct}, table slots {slots_plain} ({slots_plain*16/1e9:.2f} GB), kept after filter_lt(3): {kept}")
for mult in (1.0, 0.25):
    pf = HipPrefilter(ctx, k, int(distinct * mult))
    ms1, _ = t(lambda: pf.add_reads_dev(d, n, L), reps=1)
    m2 = HipDNAMap(ctx, k, int(distinct * (0.45 if mult >= 1 else 0.8)))
    def p2():
        m2.clear(); return pf.count_reads_dev(m2, d, n, L)
    ms2, (looked, adm) = t(p2)
    size2, slots2 = m2.size(), m2.slots()
    m2.deleteAll_lt(3)
    st = pf.stats()
    print(f"prefilter x{mult}: filter {st['bytes']/1e6:.0f} MB, pass1 {ms1:.2f} ms, pass2 {ms2:.2f} ms, admitted {adm}/{looked}, "
          f"table {size2} keys in {slots2} slots ({slots2*16/1e9:.2f} GB), kept {m2.size()} (same: {m2.size()==kept})")
    m2.close(); pf.close()
