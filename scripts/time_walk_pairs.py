"""Throughput of the paired-end stage (gk_graph_walk_pairs) on real pairs: a 1 Mbp genome, 30x coverage of 150 bp mates with
insert sizes 230..300 (range 180..250 applies to insert - k), 0.5 % error, k = 31: count -> filter -> buildGraph -> retain ->
getGraphMap -> walkPairs -> splitBySupport -> simplify.  Prints wall times and the graph before / after."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from genome_amd import dna
from genome_amd.dnamap import Context, HipDNAMap
from genome_amd.graph import Support, buildGraph

G, L, k, cov, err = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 150, 31, 30, 0.005
rng = np.random.default_rng(7)
genome = rng.integers(0, 4, G, dtype=np.uint8)
npairs = G * cov // (2 * L)
ins = rng.integers(230, 301, npairs)
start = rng.integers(0, G - 300, npairs)
flip = rng.random(npairs) < 0.5
comp = np.array([3, 2, 1, 0], np.uint8)            # A0 G1 C2 T3: complement = 3 - b
reads = np.empty((2 * npairs, L), np.uint8)
idx = np.arange(L)
for i in range(0, npairs, 100000):
    j = min(npairs, i + 100000)
    a = genome[start[i:j, None] + idx[None, :]]                                   # left mate, forward
    b = comp[genome[(start[i:j] + ins[i:j])[:, None] - 1 - idx[None, :]]]         # right mate, reverse complement of the fragment's end
    f = flip[i:j, None]
    reads[2 * i:2 * j:2] = np.where(f, b, a)
    reads[2 * i + 1:2 * j:2] = np.where(f, a, b)
e = rng.random(reads.shape) < err
reads = np.where(e, (reads + rng.integers(1, 4, reads.shape, dtype=np.uint8)) & 3, reads).astype(np.uint8)
# pack: [len][ceil(len/4) bytes], 2 bits per base LSB first
pad = np.zeros((2 * npairs, 152), np.uint8); pad[:, :L] = reads
packed = (pad[:, 0::4] | (pad[:, 1::4] << 2) | (pad[:, 2::4] << 4) | (pad[:, 3::4] << 6)).astype(np.uint8)
rec = np.concatenate([np.full((2 * npairs, 1), L, np.uint8), packed], axis=1)
binb = rec.tobytes()
ctx = Context(0)
m = HipDNAMap(ctx, k)
t0 = time.perf_counter(); m.count_reads(binb, 2 * npairs); m.deleteAll_lt(3); t_count = time.perf_counter() - t0
t0 = time.perf_counter(); g = buildGraph(k, m); g.retainLargest(); t_build = time.perf_counter() - t0
before = g.counts()
t0 = time.perf_counter(); vm = g.getGraphMap(); t_map = time.perf_counter() - t0
sup = Support(ctx)
t0 = time.perf_counter(); g.walkPairs(vm, sup, binb, npairs, 180, 250); t_walk = time.perf_counter() - t0
pairs, bad, walked = sup.sizes()
print('walkPairs phases ms:', {k_: round(v, 2) for k_, v in sup.last_ms().items()})
t0 = time.perf_counter(); rm, nn = g.splitBySupport(sup, 3); t_only_split = time.perf_counter() - t0
g.simplifyGraph(); t_split = time.perf_counter() - t0
print(f"split alone {t_only_split * 1e3:.1f} ms, simplify {(t_split - t_only_split) * 1e3:.1f} ms")
after = g.counts()
e_len = None
print(f"genome {G}, {npairs} pairs: count+filter {t_count * 1e3:.1f} ms, build+retain {t_build * 1e3:.1f} ms, getGraphMap {t_map * 1e3:.1f} ms ({vm.size()} entries), "
      f"walkPairs {t_walk * 1e3:.1f} ms = {npairs / t_walk:.3e} pairs/s ({walked} orientations walked, {bad} bad, {pairs} supported edge pairs), "
      f"split+simplify {t_split * 1e3:.1f} ms (removed {rm} edges, {nn} new nodes); graph {before} -> {after}")
