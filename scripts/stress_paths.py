"""Randomised stress: the partitioned insert pipeline against the direct kernels on device-resident reads at sizes
the CPU oracle would take minutes for.  Random k (both key widths), read length, mode, genome size, error rate,
capacity hint (incl. far too small: growth, failed segments), 1-3 batches, clear in between or not.
usage: python scripts/stress_paths.py [seconds=240] [seed=1]"""
import random, sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from genome_amd import synth
from genome_amd.dnamap import Context, HipDNAMap

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = Context(0)
t_end = time.time() + budget
cases = 0
while time.time() < t_end:
    k = rnd.choice([11, 15, 21, 27, 31, 34, 41, 55, 63, 64])
    L = rnd.randint(k, 255)
    n = int(10 ** rnd.uniform(3.5, 5.6))
    n = min(n, 40_000_000 // (L - k + 1) + 1)
    mode = rnd.choice("UG")
    G = int(10 ** rnd.uniform(3.5, 6.5)); G = max(G, L + 1)
    err = rnd.choice([0.0, 0.002, 0.02])
    nb = rnd.randint(1, 3)
    occ = n * (L - k + 1)
    hint = rnd.choice([0, 64, occ // 50, occ, 2 * occ])
    stride = synth.record_stride(L)
    d = ctx.alloc(n * stride + 64)
    tables = []
    for path in ("direct", "partitioned", "auto"):
        m = HipDNAMap(ctx, k, hint)
        m.set_insert_path(path)
        for b in range(nb):
            ctx.synth_reads(d, n, L, mode, 100 + cases, b * n, G, err)
            got = m.count_reads_dev(d, n, L)
            assert got == occ, (got, occ)
            if b == 0 and nb == 3:
                m.clear()           # exercise the deferred clear followed by a rebuild
        tables.append(m.sorted_items())
        st = m.stats()
        m.close()
    for other in tables[1:]:
        for a, b in zip(tables[0], other):
            assert a.shape == b.shape and np.array_equal(a, b), ("MISMATCH", k, L, n, mode, G, err, nb, hint)
    ctx.free(d)
    cases += 1
    print(f"case {cases}: k={k} L={L} n={n} mode={mode} G={G} err={err} batches={nb} hint={hint} distinct={len(tables[0][0])} "
          f"last stats: part={st['partitioned_launches']} direct={st['direct_launches']} spilled={st['spilled_keys']} failed_seg={st['failed_segments']} retries={st['retries_direct']} grows={st['grows']}", flush=True)
print("stress ok,", cases, "cases")
