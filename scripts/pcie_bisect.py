"""Why does bench.py's pcie_inclusive object run its uploads at half the speed of scripts/trace_host.py?  Same calls, different
orders: which preceding step makes the difference."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from genome_amd import synth
from genome_amd.dnamap import Context, HipDNAMap
ctx = Context(0)
n, L, k = 1_000_000, 150, 31
stride = synth.record_stride(L)
rec = ctx.alloc(n * stride + 64)
ctx.synth_reads(rec, n, L, "U", 2, 0, 5_000_000, 0.01)
what = sys.argv[1:]


def pcie(tag):
    hb = ctx.host_alloc(n * stride)
    hb[:] = ctx.download(rec, n * stride)
    r = bench.c2_variant(ctx, None, n, L, k, "U", 10, 2, host_buf=hb)
    print(f"{tag}: pcie {r['ms_per_step']:.3f} ms/step, P2 phase {r['phases_ms']['k_part_scatter1']:.3f}")
    ctx.host_free(hb)


if "first" in what:
    pcie("before anything")
if "prime_pin" in what:                      # a pinned allocation early, nothing else
    x = ctx.host_alloc(n * stride); x[:1] = 0; ctx.host_free(x)
if "prime_copy" in what:                     # the copy stream used early: a small host-fed count
    tiny = HipDNAMap(ctx, k, 1 << 16)
    small = ctx.host_alloc(2000 * stride); small[:] = ctx.download(rec, 2000 * stride)
    tiny.count_reads(small, 2000); tiny.close(); ctx.host_free(small)
if "prime_stage" in what:                    # a device block of the staging area's size parked in the pool early
    x = ctx.alloc(n * stride + 64); ctx.free(x)
m = HipDNAMap(ctx, k, int(n * (L - k + 1) * 1.05))
for _ in range(12):
    m.clear(); m.count_reads_dev(rec, n, L)
ctx.sync()
if "after_headline" in what:
    pcie("after the headline steps (map kept)")
if "modeG" in what:
    recg = ctx.alloc(n * stride + 64)
    ctx.synth_reads(recg, n, L, "G", 2, 0, 5_000_000, 0.01)
    r = bench.c2_variant(ctx, recg, n, L, k, "G", 5, 2)
    print(f"mode G {r['ms_per_step']:.3f}")
    ctx.free(recg)
    pcie("after mode G")
if "stats" in what:
    print(m.stats()["slots"])
    pcie("after stats")
