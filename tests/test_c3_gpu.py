"""BASELINE.json configs[2] (C3) at full size on one MI355X (-m gpu): 50 M x 150 bp reads over a 4.6 Mbp genome
(E. coli scale, ~1600x), k = 31, 0.5 % error: FreqFilter.extractFilteredKmers(data, k, 3) -> Graph.buildGraph ->
removeBubbles -> simplifyGraph -> components/retain — GraphBuilder.startup (S/scripts/GraphBuilder.scala:18-59)
plus the two structural simplifications.

6e9 windows are beyond the oracle (minutes per million reads), so this test checks what does not depend on size:
the window count, the table's invariants (no key twice, sum of counts == windows), containment of an oracle-counted
sample, the reference's own graph invariants (every node has its reverse complement as a node,
(start.seq ++ edge.seq) ends with end.seq, edge/node cross-consistency through the counts), and that the run through
the exact singleton pre-filter ends in the SAME survivors and the SAME graph (counts and content checksums).
Parity proper (bit-exact against the oracle) is at oracle-size inputs: test_table_gpu.py, test_graph_gpu.py,
test_configs_gpu.py::test_c1_full_size_table_and_graph_exact.
"""
import numpy as np
import pytest

from genome_amd import dna, synth
from genome_amd.dnamap import Context, HipDNAMap
from genome_amd.graph import buildGraph
from genome_amd.prefilter import HipPrefilter
from oracle import oracle as O

pytestmark = pytest.mark.gpu

N, L_, K, G, ERR = 50_000_000, 150, 31, 4_600_000, 0.005


def revcomp31(lo: np.ndarray) -> np.ndarray:
    """reverse complement of packed 31-mers (A0 G1 C2 T3, base i at bits 2i), vectorised"""
    v = ~lo
    out = np.zeros_like(lo)
    for i in range(K):
        out |= ((v >> np.uint64(2 * i)) & np.uint64(3)) << np.uint64(2 * (K - 1 - i))
    return out


def test_c3_full_pipeline_properties():
    ctx = Context(0)
    stride = synth.record_stride(L_)
    d = ctx.alloc(N * stride + 64)
    ctx.synth_reads(d, N, L_, "G", 3, 0, G, ERR)                  # all 50 M reads resident: 1.95 GB
    occ_want = N * (L_ - K + 1)

    # ---- plain: one call, the library batches and sizes the table itself (no hint)
    m = HipDNAMap(ctx, K, 0)
    assert m.count_reads_dev(d, N, L_) == occ_want
    st = m.stats()
    assert st["partitioned_launches"] >= 1, st
    live, bad, total, _ = m.verify_checksum()
    assert live == m.size() and bad == 0 and total == occ_want
    assert 4e8 < live < 4.6e8                                       # 4.6e6 x 3 x 31 single-error k-mers + the genome's, nearly saturated
    assert st["slots"] < 4 * live, st                               # sized for the distinct keys, not for 6e9 windows
    head = ctx.download(d, 300 * stride)
    ref = O.PMap(K, 1)
    ref.count_reads(head.tobytes(), 300)
    rlo, rhi, rcnt = ref.export_sorted()
    assert (m.apply_batch((rlo, rhi)) >= rcnt).all()
    m.deleteAll_lt(3)                                               # GraphBuilder.scala:30
    survivors = m.verify_checksum()
    assert survivors[0] == m.size() and survivors[1] == 0
    g = buildGraph(K, m)
    nodes, edges, total_len = built = g.counts()
    assert nodes > 1e7 and edges > nodes                            # error k-mers seen >= 3 times make a bushy graph at 1600x
    chk_built = g.checksum()
    # node set closed under reverse complement (termKmers = T ++ T.map(revComplement), Graph.scala:330-333)
    nlo, _ = g.getNodes()
    assert len(nlo) == nodes
    a = np.sort(nlo)
    assert len(np.unique(a)) == nodes
    assert np.array_equal(a, np.sort(revcomp31(nlo)))
    # (start.seq ++ edge.seq) ends with end.seq; edge.seq(0) names the out-edge (Graph.scala:180, application.conf:73)
    e = g.getEdges()
    assert len(e["len"]) == edges and int(e["len"].sum()) == total_len
    rnd = np.random.default_rng(3)
    for i in rnd.choice(edges, 3000, replace=False):
        o, ln = int(e["off"][i]), int(e["len"][i])
        seq = dna.unpack_2bit(e["seq"][o:o + (ln + 3) // 4], ln)
        s, t = dna.unpack(int(e["slo"][i]), 0, K), dna.unpack(int(e["elo"][i]), 0, K)
        assert (s + seq).endswith(t), (s, seq[:40], t)
    # start nodes exist, (start, first base) is unique: at most 4 out-edges per node
    starts = np.sort(e["slo"])
    assert np.isin(starts[:: max(1, edges // 100000)], a).all()
    first = np.array([int(e["seq"][int(o)]) & 3 for o in e["off"][:200000]], np.uint64)
    key = (e["slo"][:200000] << np.uint64(2)) | first
    assert len(np.unique(key)) == len(key)
    del e, nlo, a, starts
    g.removeBubbles()
    after_bubbles = g.counts()
    assert after_bubbles[0] == nodes and after_bubbles[1] <= edges
    g.simplifyGraph()
    after_simplify = g.counts()
    h1, h2 = g.componentHistograms()
    kept, comps = g.retainLargest()
    assert sum(c for _, c in h1) == comps == sum(c for _, c in h2)
    assert kept == h1[-1][0] and g.counts()[0] == kept
    chk_final = g.checksum()
    final = g.counts()
    g.close(); m.close()

    # ---- the same through the exact singleton pre-filter: same survivors, same graph
    pf = HipPrefilter(ctx, K, 450_000_000)
    pf.add_reads_dev(d, N, L_)
    m2 = HipDNAMap(ctx, K, 0)
    looked, admitted = pf.count_reads_dev(m2, d, N, L_)
    assert looked == occ_want and admitted < looked
    assert m2.size() <= live
    m2.deleteAll_lt(3)
    assert m2.verify_checksum() == survivors
    pf.close()
    g2 = buildGraph(K, m2)
    assert g2.counts() == built and g2.checksum() == chk_built
    g2.removeBubbles()
    assert g2.counts() == after_bubbles
    g2.simplifyGraph()
    assert g2.counts() == after_simplify
    kept2, comps2 = g2.retainLargest()
    assert (kept2, comps2) == (kept, comps) and g2.counts() == final and g2.checksum() == chk_final
    g2.close(); m2.close(); ctx.free(d); ctx.close()
