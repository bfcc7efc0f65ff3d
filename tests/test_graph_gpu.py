"""GPU parity tests of Graph.buildGraph / simplifyGraph / removeBubbles / removeEdge / retain
against the CPU oracle, bit-exact on the canonical node and edge serialisations."""
import json
import os
import random

import numpy as np
import pytest

from genome_amd import dna, synth
from genome_amd.dnamap import Context, HipDNAMap
from genome_amd.graph import buildGraph
from genome_amd.partitioned import PartitionedDNAMap
from oracle import oracle as O
from oracle import pyref as R

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


@pytest.fixture
def unitig_mode(ctx):
    """Force one of the two unitig constructions for the duration of a test (gk_ctx_set_option)."""
    def _set(mode):
        # "walk" = lanes fed from a queue (k_walk_q, the default walk), "walk1" = one edge per lane (k_walk pass 0), "pj" = pointer jumping
        ctx.set_option("graph_unitigs", {"auto": 0, "walk": 1, "walk1": 1, "pj": 2}[mode])
        ctx.set_option("graph_walk_queue", 0 if mode == "walk1" else -1)
    yield _set
    ctx.set_option("graph_unitigs", 0)
    ctx.set_option("graph_walk_queue", -1)


def oracle_canonical(og):
    k = og.k
    nlo, nhi = og.nodes()
    nodes = [dna.unpack(int(a), int(b), k) for a, b in zip(nlo, nhi)]
    e = og.edges()
    edges = []
    for i in range(len(e["len"])):
        seq = synth.bases_to_str(e["bases"][e["off"][i]:e["off"][i] + e["len"][i]])
        edges.append((dna.unpack(int(e["slo"][i]), int(e["shi"][i]), k),
                      dna.unpack(int(e["elo"][i]), int(e["ehi"][i]), k), seq))
    return nodes, edges


@pytest.mark.parametrize("name", sorted(f[:-5] for f in os.listdir(GOLDEN) if f.endswith(".json") and f != "hash_ties.json"))
def test_golden_graphs(ctx, name):
    fx = json.load(open(os.path.join(GOLDEN, name + ".json")))
    k, P, rounds = fx["k"], fx["P"], fx["rounds"]
    m = PartitionedDNAMap(ctx, k, P) if P > 1 else HipDNAMap(ctx, k)
    m.count_reads(bytes.fromhex(fx["bin_hex"]), fx["nreads"])
    m.deleteAll_lt(rounds)
    g = buildGraph(k, m)
    nodes, edges = g.canonical()
    assert nodes == fx["nodes"]
    assert [list(e) for e in edges] == fx["edges"]
    assert g.counts() == (len(nodes), len(edges), sum(len(e[2]) for e in edges))
    g.removeBubbles()
    assert [list(e) for e in g.canonical()[1]] == fx["edges_after_bubbles"]
    g.simplifyGraph()
    nodes, edges = g.canonical()
    assert nodes == fx["nodes_after_simplify"]
    assert [list(e) for e in edges] == fx["edges_after_simplify"]
    g.close(); m.close()


def _reads(rnd, n, lmin, lmax, glen, err, haplotypes=1):
    g = "".join(rnd.choice("AGCT") for _ in range(glen))
    haps = [g]
    for _ in range(haplotypes - 1):
        h = list(g)
        for p in range(30, glen, 57):
            h[p] = rnd.choice([c for c in "AGCT" if c != h[p]])
        haps.append("".join(h))
    out = []
    for _ in range(n):
        h = rnd.choice(haps)
        ln = rnd.randint(lmin, lmax)
        st = rnd.randrange(0, glen - ln + 1)
        r = h[st:st + ln]
        if rnd.random() < 0.5:
            r = R.rev_comp(r)
        out.append("".join(c if rnd.random() >= err else rnd.choice([x for x in "AGCT" if x != c]) for c in r))
    return out


@pytest.mark.parametrize("k,seed,hap", [(7, 1, 1), (11, 2, 2), (15, 3, 2), (21, 4, 1), (31, 5, 2), (35, 6, 2), (47, 7, 1), (63, 8, 2), (64, 9, 2)])
def test_build_bubbles_simplify_remove_retain_vs_oracle(ctx, k, seed, hap):
    rnd = random.Random(seed)
    reads = _reads(rnd, 900, k + 5, min(255, k + 90), 1500, 0.01, hap)
    binb = dna.reads_to_bin(reads)
    m = HipDNAMap(ctx, k)
    ref = O.PMap(k, 1)
    m.count_reads(binb, len(reads)); ref.count_reads(binb, len(reads))
    m.deleteAll_lt(2); ref.delete_lt(2)
    g, og = buildGraph(k, m), O.Graph(ref)
    assert g.canonical() == oracle_canonical(og)
    n0, e0, l0 = g.counts()
    assert (n0, e0, l0) == (og.num_nodes(), og.num_edges(), og.total_edge_len()) and n0 > 0 and e0 > 0
    # invariants of SURVEY §4 on the GPU result itself
    nodes, edges = g.canonical()
    nodeset = set(nodes)
    assert {R.rev_comp(x) for x in nodes} == nodeset
    for s, t, q in edges:
        assert (s + q).endswith(t)
    # out-edge insertion order of a fresh graph is A,G,C,T (Graph.scala:351 over Base.fromInt)
    for s in nodes[:40]:
        assert g.out_order(s) == og.out_order(*dna.pack(s))
    g.removeBubbles(); og.remove_bubbles()
    assert g.canonical() == oracle_canonical(og)
    g.simplifyGraph(); og.simplify()
    assert g.canonical() == oracle_canonical(og)
    for s in g.canonical()[0][:60]:      # insertion order after the merges (removeEdge/addEdge :180,:192)
        assert g.out_order(s) == og.out_order(*dna.pack(s)), s
    # removeEdge of every third edge, then simplify again: long (1,1) chains, self-loops, dead ends
    _, edges = g.canonical()
    victims = [(s, q[0]) for i, (s, _, q) in enumerate(edges) if i % 3 == 0]
    assert g.removeEdges(victims) == len(victims)
    assert g.removeEdges(victims[:5]) == 0          # already gone
    for s, b in victims:
        assert og.remove_edge(*dna.pack(s), "AGCT".index(b))
    assert g.canonical() == oracle_canonical(og)
    g.simplifyGraph(); og.simplify()
    assert g.canonical() == oracle_canonical(og)
    for s in g.canonical()[0][:60]:
        assert g.out_order(s) == og.out_order(*dna.pack(s)), s
    g.removeBubbles(); og.remove_bubbles()
    assert g.canonical() == oracle_canonical(og)
    kept, comps = g.retainLargest()
    assert comps == og.num_components()
    assert kept == og.retain_largest()
    assert g.canonical() == oracle_canonical(og)
    assert g.counts()[0] == kept
    g.close(); m.close()


def test_two_components_and_perfect_cycle(ctx):
    """A circular genome alone is an all-(1,1) cycle: buildGraph yields nothing (Graph.scala:375
    'perfect cycles are ignored'); next to a linear one only the linear component appears."""
    rnd = random.Random(5)
    k = 15
    circ = "".join(rnd.choice("AGCT") for _ in range(120))
    lin = "".join(rnd.choice("AGCT") for _ in range(200))
    cc = circ + circ[:60]
    reads = [cc[i:i + 50] for i in range(0, 121)] * 2
    m = HipDNAMap(ctx, k); ref = O.PMap(k, 1)
    b = dna.reads_to_bin(reads)
    m.count_reads(b, len(reads)); ref.count_reads(b, len(reads))
    g, og = buildGraph(k, m), O.Graph(ref)
    assert g.counts() == (0, 0, 0) and og.num_nodes() == 0
    assert g.canonical() == ([], [])
    g.simplifyGraph(); g.removeBubbles()
    assert g.retainLargest() == (0, 0)
    g.close()
    reads2 = [lin[i:i + 50] for i in range(0, 151)] * 2
    b2 = dna.reads_to_bin(reads2)
    m.count_reads(b2, len(reads2)); ref.count_reads(b2, len(reads2))
    g, og = buildGraph(k, m), O.Graph(ref)
    assert g.canonical() == oracle_canonical(og)
    assert g.counts()[0] == 4
    g.close(); m.close()


def test_even_k_palindromes(ctx):
    """Even k admits x == rc(x): a tie filed under rcx == x (FreqFilter.scala:32), one node not two."""
    k = 6
    rnd = random.Random(9)
    core = "AGGCCT"                      # its own reverse complement
    assert R.rev_comp(core) == core
    left = "".join(rnd.choice("AGCT") for _ in range(40))
    reads = []
    for tail in ("".join(rnd.choice("AGCT") for _ in range(40)) for _ in range(3)):
        s = left + core + tail
        reads += [s[i:i + 30] for i in range(0, len(s) - 29)]
    b = dna.reads_to_bin(reads)
    m = HipDNAMap(ctx, k); ref = O.PMap(k, 1)
    m.count_reads(b, len(reads)); ref.count_reads(b, len(reads))
    for a, c in zip(m.sorted_items(), ref.export_sorted()):
        assert np.array_equal(a, c)
    g, og = buildGraph(k, m), O.Graph(ref)
    assert g.canonical() == oracle_canonical(og)
    g.removeBubbles(); og.remove_bubbles(); g.simplifyGraph(); og.simplify()
    assert g.canonical() == oracle_canonical(og)
    g.close(); m.close()


def test_graph_at_scale_properties(ctx):
    """200k x 150 bp reads over a 300 kbp genome (k=31): too slow for the oracle's literal graph
    build at full tilt, so check properties: strand symmetry of nodes and edges, edge/node
    consistency, every genome k-mer covered by the graph (CheckGraph.scala:48-55)."""
    n, L_, k, G = 200_000, 150, 31, 300_000
    d = ctx.alloc(n * synth.record_stride(L_) + 64)
    ctx.synth_reads(d, n, L_, "G", 11, 0, G, 0.005)
    m = HipDNAMap(ctx, k, n * 40)
    m.count_reads_dev(d, n, L_)
    m.deleteAll_lt(3)
    g = buildGraph(k, m)
    nn, ne, ln = g.counts()
    assert nn > 0 and ne > 0
    lo, hi = g.getNodes()
    e = g.getEdges()
    nodes = set(int(x) for x in lo)
    assert len(nodes) == nn
    assert set(int(x) for x in e["slo"]) <= nodes and set(int(x) for x in e["elo"]) <= nodes
    # reverse-complement closure of the node set
    sample = list(nodes)[:2000]
    for x in sample:
        assert O.revcomp(x, 0, k)[0] in nodes
    # total walked length is strand symmetric: every edge has a mirror of the same length
    assert ln % 2 == 0 and ne % 2 == 0
    g.simplifyGraph()
    assert g.counts()[:2] == (nn, ne)            # a fresh graph has no (1,1)/(0,0) node
    g.close(); m.close(); ctx.free(d)


@pytest.mark.parametrize("mode", ["walk", "walk1", "pj"])
@pytest.mark.parametrize("k,seed,hap", [(11, 2, 2), (31, 5, 2), (35, 6, 2), (64, 9, 1)])
def test_unitig_construction_modes_agree_with_oracle(ctx, unitig_mode, mode, k, seed, hap):
    """Both unitig constructions — one lane walking each edge (k_walk) and pointer jumping
    (k_pj_*) — must give the oracle's graph, on bushy graphs and on a long clean unitig."""
    unitig_mode(mode)
    rnd = random.Random(seed)
    for err, nreads in ((0.01, 700), (0.0, 400)):
        reads = _reads(rnd, nreads, k + 5, min(255, k + 90), 1500, err, hap)
        binb = dna.reads_to_bin(reads)
        m = HipDNAMap(ctx, k); ref = O.PMap(k, 1)
        m.count_reads(binb, len(reads)); ref.count_reads(binb, len(reads))
        m.deleteAll_lt(2); ref.delete_lt(2)
        g, og = buildGraph(k, m), O.Graph(ref)
        assert g.canonical() == oracle_canonical(og)
        g.removeBubbles(); og.remove_bubbles(); g.simplifyGraph(); og.simplify()
        assert g.canonical() == oracle_canonical(og)
        g.close(); m.close()


def test_long_unitig_is_fast_and_exact(ctx):
    """A 300 kbp error-free genome is two mirrored unitigs of ~3e5 bases: pointer jumping builds them
    in milliseconds (one lane walking would take ~0.6 s); the sequence must BE the genome."""
    import time
    n, L_, k, G = 60000, 150, 31, 300_000
    d = ctx.alloc(n * synth.record_stride(L_) + 64)
    ctx.synth_reads(d, n, L_, "G", 12, 0, G, 0.0)
    m = HipDNAMap(ctx, k, G * 2)
    m.count_reads_dev(d, n, L_)
    m.deleteAll_lt(1)
    t0 = time.perf_counter()
    g = buildGraph(k, m)
    dt = time.perf_counter() - t0
    nodes, edges = g.canonical()
    genome = synth.bases_to_str(synth.genome_bases(G, 12))
    total = sum(len(q) for _, _, q in edges)
    assert len(nodes) % 2 == 0 and total == g.counts()[2]
    # every edge, prefixed by its start node, is a substring of the genome or of its reverse complement
    rc = R.rev_comp(genome)
    for s, t_, q in edges:
        w = s + q
        assert (w in genome) or (w in rc)
        assert w.endswith(t_)
    longest = max(len(q) for _, _, q in edges)
    assert longest > 20000
    assert dt < 5.0
    g.close(); m.close(); ctx.free(d)


@pytest.mark.parametrize("idx", range(9))
@pytest.mark.parametrize("rounds", [1, 3])
def test_hash_tie_kmers_table_and_graph(ctx, idx, rounds):
    """The hash-rule tie (FreqFilter.scala:31-32): a k-mer x != rc(x) with hashCode(x) == hashCode(rc x)
    is stored under TWO keys (seen as x -> filed under rc x and vice versa).  Table and graph must
    match the oracle when both keys survive (rounds=1) and when the filter drops one of them
    (rounds=3: 4 reads from one strand, 2 from the other) — `contains` then has to find the k-mer
    through the surviving orientation (Graph.scala:270)."""
    tie = json.load(open(os.path.join(GOLDEN, "hash_ties.json")))["ties"][idx]
    k, x = tie["k"], tie["kmer"]
    rnd = random.Random(idx)
    flank = lambda n: "".join(rnd.choice("AGCT") for _ in range(n))
    genome = flank(60) + x + flank(60)
    branch = genome[:60 + k - 1] + ("A" if genome[60 + k - 1] != "A" else "G") + flank(40)   # a fork right after x
    reads = []
    for g_, plus, minus in ((genome, 4, 2), (branch, 3, 3)):
        L_ = min(len(g_), k + 40)
        for i in range(0, len(g_) - L_ + 1, 3):
            reads += [g_[i:i + L_]] * plus + [R.rev_comp(g_[i:i + L_])] * minus
    binb = dna.reads_to_bin(reads)
    m = HipDNAMap(ctx, k); ref = O.PMap(k, 1)
    assert m.count_reads(binb, len(reads)) == ref.count_reads(binb, len(reads))
    lo, hi = dna.pack(x); rlo, rhi = dna.pack(R.rev_comp(x))
    assert ref.get(lo, hi) is not None and ref.get(rlo, rhi) is not None          # both keys exist
    assert m.apply(x) == ref.get(lo, hi) and m.apply(R.rev_comp(x)) == ref.get(rlo, rhi)
    for a, b in zip(m.sorted_items(), ref.export_sorted()):
        assert np.array_equal(a, b)
    m.deleteAll_lt(rounds); ref.delete_lt(rounds)
    for a, b in zip(m.sorted_items(), ref.export_sorted()):
        assert np.array_equal(a, b)
    for mode in ("walk", "walk1", "pj"):
        ctx.set_option("graph_unitigs", {"walk": 1, "walk1": 1, "pj": 2}[mode])
        ctx.set_option("graph_walk_queue", 0 if mode == "walk1" else -1)
        try:
            g, og = buildGraph(k, m), O.Graph(ref)
        finally:
            ctx.set_option("graph_unitigs", 0)
            ctx.set_option("graph_walk_queue", -1)
        assert g.canonical() == oracle_canonical(og)
        g.removeBubbles(); og.remove_bubbles(); g.simplifyGraph(); og.simplify()
        assert g.canonical() == oracle_canonical(og)
        g.close()
    m.close()


def test_graphbuilder_flow_through_the_mirror_api(ctx):
    """GraphBuilder.startup (S/scripts/GraphBuilder.scala:28-54) spelled with the Python mirror:
    extractFilteredKmers(data, k, rounds=3) -> size -> buildGraph -> components/retain, for one
    table and for 4 logical partitions, against the oracle."""
    from genome_amd.freqfilter import PairedEndData, extractFilteredKmers
    k, rounds = 21, 3
    rec = synth.reads_mode_g(6000, 100, 12000, 0.01, config_id=31)
    data = PairedEndData(count=3000, bin_bytes=rec.tobytes())
    ref = O.PMap(k, 1)
    ref.count_reads(rec[:5000].tobytes(), 5000)          # take_first = 2500 pairs
    ref.delete_lt(rounds)
    og = O.Graph(ref)
    want = oracle_canonical(og)
    for parts in (1, 4):
        kmers = extractFilteredKmers(data, k, rounds, ctx=ctx, take_first=2500, partitions=parts)
        assert kmers.size() == ref.size()
        g = buildGraph(k, kmers)
        assert g.canonical() == want
        kept, comps = g.retainLargest()
        og2 = O.Graph(ref)
        assert comps == og2.num_components() and kept == og2.retain_largest()
        assert g.canonical() == oracle_canonical(og2)
        g.close(); kmers.close()


@pytest.mark.parametrize("k,seed", [(21, 3), (31, 4), (41, 5)])
def test_simplify_merges_long_edges_like_the_oracle(ctx, k, seed):
    """A contig-like graph: a 24 kbp genome with a handful of SNPs between two haplotypes — edges of thousands of bases.
    removeBubbles + simplifyGraph then concatenates LONG pieces (k_copy_long: 16 bases per thread, ORed into place, next to
    short pieces copied by the chain's own lane at every 2-bit alignment) — the result must be the oracle's, base for base."""
    rnd = random.Random(seed)
    glen = 24000
    g0 = [rnd.choice("AGCT") for _ in range(glen)]
    h1 = list(g0)
    # SNP sites at irregular distances: long stretches (> 1024 bases) and short ones (a few dozen bases) between them
    sites = [700, 760, 3100, 3133, 5200, 9050, 9051 + k + 3, 14000, 14007 + 2 * k, 20500, 23000]
    for p in sites:
        h1[p] = rnd.choice([c for c in "AGCT" if c != h1[p]])
    haps = ["".join(g0), "".join(h1)]
    reads = []
    for _ in range(9000):
        h = rnd.choice(haps)
        ln = rnd.randint(k + 20, 150)
        st = rnd.randrange(0, glen - ln + 1)
        r = h[st:st + ln]
        reads.append(R.rev_comp(r) if rnd.random() < 0.5 else r)
    binb = dna.reads_to_bin(reads)
    m, ref = HipDNAMap(ctx, k), O.PMap(k, 1)
    m.count_reads(binb, len(reads)); ref.count_reads(binb, len(reads))
    m.deleteAll_lt(2); ref.delete_lt(2)
    g, og = buildGraph(k, m), O.Graph(ref)
    assert g.canonical() == oracle_canonical(og)
    assert max(len(e[2]) for e in g.canonical()[1]) > 1024
    g.removeBubbles(); og.remove_bubbles()
    g.simplifyGraph(); og.simplify()
    nodes, edges = g.canonical()
    assert (nodes, edges) == oracle_canonical(og)
    assert max(len(e[2]) for e in edges) > 4000           # long pieces were concatenated
    g.close(); m.close()
