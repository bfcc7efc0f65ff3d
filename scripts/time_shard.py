"""Time the two routing kernels of the multi-GPU path (super-k-mer records vs 8-byte keys) at C2 for P = 1, 2, 8."""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from genome_amd import synth
from genome_amd.dnamap import Context, skm_slot_bytes
ctx = Context(0)
n, L, k = 1_000_000, 150, 31
d = ctx.alloc(n * synth.record_stride(L) + 64)
ctx.synth_reads(d, n, L, "U", 2, 0, 0, 0.0)
for P in (1, 2, 8):
    cap = n * 40 * P
    out = ctx.alloc(cap * skm_slot_bytes(k))
    for it in range(3):
        t0 = time.perf_counter()
        recs, kmers = ctx.shard_superkmers(k, d, n, L, P, out, cap)
        dt = time.perf_counter() - t0
    print("P", P, "superkmers ms", round(dt * 1e3, 2), "records", int(recs.sum()), "kmers/rec", round(float(kmers.sum()) / float(recs.sum()), 2), flush=True)
    ctx.free(out)
    keys = ctx.alloc(n * 120 * 8)
    for it in range(3):
        t0 = time.perf_counter()
        c = ctx.shard_reads(k, d, n, L, P, keys, n * 120)
        dt = time.perf_counter() - t0
    print("P", P, "keys ms", round(dt * 1e3, 2), flush=True)
    ctx.free(keys)
