"""CPU tests of the oracle itself: the C restatement (oracle/gk_oracle.c) against the known answers
of SURVEY.md §8a-9/§8c, against the independent Python restatement (oracle/pyref.py), against the
reference's in-source invariants and against the committed golden fixtures.

Parity status: the reference ships no tests or fixtures (parity unpinned); these tests are the pin
the build can offer.  S/ = /root/reference/src/main/scala/ru/ifmo/genome/.
"""
import json
import os
import random

import numpy as np
import pytest

from genome_amd import synth
from oracle import oracle as O
from oracle import pyref as R

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")

KATS = [  # (k, forward, lo, hi, rc_lo, rc_hi, hash, hash_rc)   SURVEY.md §8c
    (21, "GACATTTTGATATTATCGACA", 0x86CF31FF21, 0, 0x2DC02CC31B7, 0, -818806873, 46936939),
    (31, "AAATGTAGTTGCGGTCATAGCACTGCCTTAT", 0x33E9E24CB59F4DC0, 0, 0x3F238268739D250C, 0,
     -2039042164, 1287563108),
    (55, "GTTGCCATAGTCTATACTGGGAGGTCCTCATGGTATCGTTCTCACTAGGGACAAA", 0x72EB515E33B4CA7D, 0x854E2EF6CD,
     0x851CA31811D3AB7F, 0x2097384CD2AE, 1178810458, -1369388079),
    (63, "ATACGCAGCCAGGCGCCTCTTAACTCTGCACGTAGATGAGTTGTGATCTACGGACAACCGTCG", 0x627B83EE994A498C,
     0x1B682163B1DF4713, 0x8ECB822C4DADF586, 0x336797A6510F449D, -729402787, 1078901399),
]


@pytest.mark.parametrize("k,s,lo,hi,rclo,rchi,h,hrc", KATS)
def test_known_answers(k, s, lo, hi, rclo, rchi, h, hrc):
    assert R.pack(s) == (lo, hi)
    assert O.revcomp(lo, hi, k) == (rclo, rchi)
    assert R.pack(R.rev_comp(s)) == (rclo, rchi)
    assert O.hash_code(lo, hi, k) == h == R.hash_code(s)
    assert O.hash_code(rclo, rchi, k) == hrc == R.hash_code(R.rev_comp(s))
    want = (lo, hi) if h < hrc else (rclo, rchi)
    assert O.canon(lo, hi, k) == want == R.pack(R.canon(s))


def test_improve_known_answers():
    for x, e in [(0, -8130816), (1, -8139033), (42, -106205), (-1, 8662),
                 (2147483647, -2147341994), (123456789, 1272491941)]:
        assert O.improve(x) == e == R.improve(x)


def test_first_slot_known_answers():  # first slot in a 16-bin table
    assert O.improve(-818806873) & 15 == 1
    assert O.improve(-2039042164) & 15 == 9


def test_base_invariants():  # S/dna/Base.scala:22-23
    L = O.lib()
    for b in range(4):
        assert L.gko_base_complement(L.gko_base_complement(b)) == b
        assert L.gko_base_from_char(L.gko_base_to_char(b)) == b
    assert [L.gko_base_complement(b) for b in range(4)] == [3, 2, 1, 0]


def test_supported_k():
    L = O.lib()
    assert [k for k in range(0, 70) if L.gko_k_supported(k)] == list(range(2, 32)) + list(range(34, 65))


@pytest.mark.parametrize("k", [2, 5, 11, 21, 31, 34, 35, 47, 55, 63, 64])
def test_kmer_ops_vs_pyref(k):
    rnd = random.Random(k)
    L = O.lib()
    for _ in range(200):
        s = "".join(rnd.choice("AGCT") for _ in range(k))
        lo, hi = R.pack(s)
        assert O.revcomp(lo, hi, k) == R.pack(R.rev_comp(s))
        assert O.hash_code(lo, hi, k) == R.hash_code(s)
        assert O.canon(lo, hi, k) == R.pack(R.canon(s))
        for P in (1, 2, 7, 14):
            assert O.partition(lo, hi, k, P) == R.partition(s, P)
        b = rnd.randrange(4)
        r = L.gko_append(O.km(lo, hi), b, k)
        assert (r.lo, r.hi) == R.pack(s[1:] + "AGCT"[b])
        r = L.gko_prepend(b, O.km(lo, hi), k)
        assert (r.lo, r.hi) == R.pack("AGCT"[b] + s[:k - 1])
        # double reverse complement is the identity for every supported k
        assert O.revcomp(*O.revcomp(lo, hi, k), k) == (lo, hi)


def test_tie_goes_to_reverse_complement():
    """FreqFilter.scala:31-32: `if (x.hashCode < rcx.hashCode) x else rcx` — a palindrome (even k) is
    the simplest tie; both restatements must file it under rcx (== x)."""
    s = "AGCT" * 3 + "AGCT"[::-1] * 0  # not necessarily palindromic; construct one explicitly
    half = "AGGCTA"
    pal = half + R.rev_comp(half)
    assert R.rev_comp(pal) == pal
    lo, hi = R.pack(pal)
    assert O.canon(lo, hi, len(pal)) == (lo, hi)
    del s


def _random_reads(rnd, n, lmin, lmax, genome_len, err):
    g = "".join(rnd.choice("AGCT") for _ in range(genome_len))
    reads = []
    for _ in range(n):
        ln = rnd.randint(lmin, lmax)
        st = rnd.randrange(0, genome_len - ln + 1)
        r = g[st:st + ln]
        if rnd.random() < 0.5:
            r = R.rev_comp(r)
        r = "".join(c if rnd.random() >= err else rnd.choice([x for x in "AGCT" if x != c]) for c in r)
        reads.append(r)
    return reads


def _c_table(reads, k, P, rounds=None):
    pm = O.PMap(k, P)
    occ = pm.count_reads(R.reads_to_bin(reads), len(reads))
    assert occ == sum(max(0, len(r) - k + 1) for r in reads)
    if rounds is not None:
        pm.delete_lt(rounds)
    return pm


def _py_sorted(m):
    items = m.sorted_items()
    lo = np.array([R.pack(s)[0] for s, _ in items], np.uint64)
    hi = np.array([R.pack(s)[1] for s, _ in items], np.uint64)
    cnt = np.array([c for _, c in items], np.int32)
    return lo, hi, cnt


@pytest.mark.parametrize("k,P", [(5, 1), (11, 1), (11, 3), (21, 2), (31, 1), (35, 1), (35, 4), (63, 2)])
def test_count_and_filter_vs_pyref(k, P):
    rnd = random.Random(1000 * k + P)
    reads = _random_reads(rnd, 40, max(2, k - 3), min(255, k + 40), 260, 0.02)
    reads += ["", "A", "ACGT"[:min(4, k - 1)]]       # ragged / empty records are skipped (FreqFilter.scala:29)
    for rounds in (None, 2, 3):
        pm = _c_table(reads, k, P, rounds)
        pr = R.extract_filtered_kmers(reads, k, rounds if rounds else 0, P, do_filter=rounds is not None)
        a, b = pm.export_sorted(), _py_sorted(pr)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        assert pm.size() == pr.size()
        for p in range(P):   # literal container state: size, bins and number of rescales agree
            assert pm.part_stats(p) == (pr.parts[p].size, pr.parts[p].bins, pr.parts[p].rescales)


def test_partition_count_is_unobservable():
    """PartitionedDNAMap only routes (PartitionedDNAMap.scala:60-63): sorted content is P-independent."""
    rnd = random.Random(7)
    reads = _random_reads(rnd, 60, 30, 60, 300, 0.01)
    ref = _c_table(reads, 21, 1, 3).export_sorted()
    for P in (2, 5, 14):
        got = _c_table(reads, 21, P, 3).export_sorted()
        for x, y in zip(ref, got):
            assert np.array_equal(x, y)


def test_container_tombstones_and_rescale():
    """ArrayDNAMap.scala:129-150/164-173/217-230: tombstone reuse keeps `set`, rescale thresholds."""
    L = O.lib()
    m = L.gko_map_new(5)
    keys = [O.km(i * 37 + 1) for i in range(12)]
    for kk in keys:
        L.gko_map_update_inc(m, kk)
    assert L.gko_map_size(m) == 12 and L.gko_map_bins(m) == 32   # 16*0.7 < 12 -> 32
    L.gko_map_update_inc(m, keys[0])
    L.gko_map_update_inc(m, keys[0])
    L.gko_map_delete_lt(m, 3)          # only keys[0] (count 3) survives; 1 < 0.3*32 -> shrink to 16
    assert L.gko_map_size(m) == 1 and L.gko_map_bins(m) == 16
    import ctypes as C
    v = C.c_int32()
    assert L.gko_map_get(m, keys[0], C.byref(v)) == 1 and v.value == 3
    assert L.gko_map_get(m, keys[1], C.byref(v)) == 0
    # putNew is a multimap insert (:152-162); getAll returns most recently probed first (:103-113)
    L.gko_map_put_new(m, keys[0], 9)
    out = (C.c_int32 * 4)()
    assert L.gko_map_get_all(m, keys[0], out, 4) == 2 and list(out)[:2] == [9, 3]
    L.gko_map_free(m)


def _graph_pair(reads, k, P=1, rounds=2):
    pm = _c_table(reads, k, P, rounds)
    pr = R.extract_filtered_kmers(reads, k, rounds, P)
    return pm, O.Graph(pm), R.build_graph(k, pr)


def _c_graph_canonical(g: O.Graph):
    k = g.k
    nlo, nhi = g.nodes()
    nodes = [R.unpack(int(a), int(b), k) for a, b in zip(nlo, nhi)]
    e = g.edges()
    edges = []
    for i in range(len(e["len"])):
        seq = synth.bases_to_str(e["bases"][e["off"][i]:e["off"][i] + e["len"][i]])
        edges.append((R.unpack(int(e["slo"][i]), int(e["shi"][i]), k),
                      R.unpack(int(e["elo"][i]), int(e["ehi"][i]), k), seq))
    return nodes, edges


@pytest.mark.parametrize("k,P,seed", [(7, 1, 1), (9, 2, 2), (11, 1, 3), (15, 3, 4), (35, 1, 5), (35, 2, 6)])
def test_graph_build_simplify_bubbles_vs_pyref(k, P, seed):
    rnd = random.Random(seed)
    reads = _random_reads(rnd, 80, k + 2, k + 30, 220, 0.02)
    pm, cg, pg = _graph_pair(reads, k, P)
    assert _c_graph_canonical(cg) == pg.canonical()
    nodes, edges = pg.canonical()
    assert len(nodes) > 0 and len(edges) > 0
    # SURVEY §4 invariants: (start.seq + edge.seq) ends with end.seq; interior k-mers are non-terminal
    nodeset = set(nodes)
    for s, t, q in edges:
        assert (s + q).endswith(t)
        walk = s + q
        for i in range(1, len(q)):
            assert walk[i:i + k] not in nodeset
    assert set(R.rev_comp(n) for n in nodes) == nodeset      # both strands are nodes (Graph.scala:330-333)
    # removeBubbles then simplifyGraph, as GraphSimplifier.scala:317-318 would
    cg.remove_bubbles(); pg.remove_bubbles()
    assert _c_graph_canonical(cg) == pg.canonical()
    cg.simplify(); pg.simplify()
    assert _c_graph_canonical(cg) == pg.canonical()
    # explicit removeEdge of every third edge, then simplify: exercises the (1,1) merge path
    _, edges = pg.canonical()
    for i, (s, _, q) in enumerate(edges):
        if i % 3 == 0:
            lo, hi = R.pack(s)
            assert cg.remove_edge(lo, hi, "AGCT".index(q[0]))
            eid = next(e for (b, e) in next(n for n in pg.nodes.values() if n["seq"] == s)["outs"] if b == q[0])
            pg.remove_edge(eid)
    cg.simplify(); pg.simplify()
    assert _c_graph_canonical(cg) == pg.canonical()
    for nid, n in pg.nodes.items():   # no (1,1)/(0,0) node survives a simplify pass
        assert not (len(n["ins"]) == 1 and len(n["outs"]) == 1)
        assert len(n["ins"]) + len(n["outs"]) > 0


def test_application_conf_data_point():
    """The one real data point the reference holds (application.conf:73): k=19, the edge sequence
    is the bases appended after the start k-mer and ends in the end node's k-mer."""
    edge = ("CGCGGTAAAGACGTGCAATACCTTGCCCATGACTGCCGACCTACCGTACCCCGGAAGGAACCATCACTGCTTTTATGCATATGGTGGAG"
            "TACCGGCGTAATCAGAAGCAACTACGCGAAACGCCGGCGTTGCCCAGCAATCTGACTTCCAATACCG")
    start, end = "TGCGGCGAGCACTCCTCGC", "AATCTGACTTCCAATACCG"
    assert len(start) == len(end) == 19
    assert (start + edge).endswith(end) and edge.endswith(end)
    # rebuild the same unitig with the oracle: a linear genome yields exactly this edge shape
    genome = start + edge
    reads = [genome[i:i + 60] for i in range(0, len(genome) - 59)] * 3
    pm = _c_table(reads, 19, 1, 3)
    g = O.Graph(pm)
    nodes, edges = _c_graph_canonical(g)
    assert start in nodes and end in nodes
    assert (start, end, edge) in edges


def test_components_and_retain():
    rnd = random.Random(11)
    a = _random_reads(rnd, 60, 30, 50, 200, 0.0)
    b = _random_reads(rnd, 20, 30, 50, 80, 0.0)
    pm = _c_table(a + b, 15, 1, 1)
    g = O.Graph(pm)
    nc = g.num_components()
    assert nc >= 2 and nc % 2 == 0          # each component has a disjoint reverse-complement mirror
    before = g.num_nodes()
    kept = g.retain_largest()
    assert kept == g.num_nodes() <= before
    assert g.num_components() == 1


def test_golden_fixtures():
    """tests/golden/*.json were emitted by oracle/pyref.py (tests/golden/make_golden.py); the C
    oracle must reproduce them bit for bit."""
    files = sorted(f for f in os.listdir(GOLDEN) if f.endswith(".json") and f != "hash_ties.json")
    assert files, "no golden fixtures"
    for f in files:
        fx = json.load(open(os.path.join(GOLDEN, f)))
        k, P, rounds = fx["k"], fx["P"], fx["rounds"]
        binb = bytes.fromhex(fx["bin_hex"])
        pm = O.PMap(k, P)
        assert pm.count_reads(binb, fx["nreads"]) == fx["occurrences"]
        lo, hi, cnt = pm.export_sorted()
        assert [[int(a), int(b), int(c)] for a, b, c in zip(lo, hi, cnt)] == fx["table"]
        pm.delete_lt(rounds)
        lo, hi, cnt = pm.export_sorted()
        assert [[int(a), int(b), int(c)] for a, b, c in zip(lo, hi, cnt)] == fx["table_filtered"]
        g = O.Graph(pm)
        nodes, edges = _c_graph_canonical(g)
        assert nodes == fx["nodes"]
        assert [list(e) for e in edges] == fx["edges"]
        g.remove_bubbles()
        assert [list(e) for e in _c_graph_canonical(g)[1]] == fx["edges_after_bubbles"]
        g.simplify()
        nodes, edges = _c_graph_canonical(g)
        assert nodes == fx["nodes_after_simplify"]
        assert [list(e) for e in edges] == fx["edges_after_simplify"]


def test_synth_is_chunk_invariant():
    a = synth.reads_mode_u(64, 37, config_id=3)
    b = np.concatenate([synth.reads_mode_u(20, 37, 3, 0), synth.reads_mode_u(44, 37, 3, 20)])
    assert np.array_equal(a, b) and a.shape == (64, 1 + 10) and (a[:, 0] == 37).all()
    g = synth.reads_mode_g(50, 40, 500, 0.05, config_id=1)
    h = np.concatenate([synth.reads_mode_g(13, 40, 500, 0.05, 1, 0), synth.reads_mode_g(37, 40, 500, 0.05, 1, 13)])
    assert np.array_equal(g, h)
    # error-free reads are substrings of the genome or its reverse complement
    gen = synth.bases_to_str(synth.genome_bases(500, 1))
    rd = R.reads_from_bin(synth.reads_mode_g(20, 40, 500, 0.0, 1).tobytes(), 20)
    for r in rd:
        assert r in gen or R.rev_comp(r) in gen


def test_multithreaded_baseline_equals_sequential():
    """gko_count_reads_mt (the CPU baseline's multi-core form) == the sequential PartitionedDNAMap."""
    rnd = random.Random(21)
    reads = _random_reads(rnd, 400, 20, 120, 900, 0.02) + ["", "AG"]
    binb = R.reads_to_bin(reads)
    for k, P, T in [(21, 4, 3), (35, 3, 5), (11, 1, 2)]:
        seq = O.PMap(k, P); seq.count_reads(binb, len(reads))
        mt = O.PMap(k, P)
        assert mt.count_reads_mt(binb, len(reads), T) == sum(max(0, len(r) - k + 1) for r in reads)
        for x, y in zip(seq.export_sorted(), mt.export_sorted()):
            assert np.array_equal(x, y)
        for p in range(P):
            assert seq.part_stats(p)[0] == mt.part_stats(p)[0]


def test_hash_tie_vectors():
    """tests/golden/hash_ties.json: x != rc(x) with equal hashCode.  Both restatements must send the
    tie to the reverse complement (FreqFilter.scala:32), so occurrences seen as x are filed under
    rc(x) and vice versa — two keys for one k-mer."""
    ties = json.load(open(os.path.join(GOLDEN, "hash_ties.json")))["ties"]
    assert len(ties) >= 8
    for t in ties:
        k, s = t["k"], t["kmer"]
        rc = R.rev_comp(s)
        lo, hi = R.pack(s)
        assert s != rc and R.hash_code(s) == R.hash_code(rc) == O.hash_code(lo, hi, k) == t["hash"]
        assert O.canon(lo, hi, k) == R.pack(rc) and R.canon(s) == rc and R.canon(rc) == s
        pm = O.PMap(k, 1)
        pm.count_reads(R.reads_to_bin([s, s, rc]), 3)
        assert pm.get(*R.pack(rc)) == 2 and pm.get(lo, hi) == 1 and pm.size() == 2


@pytest.mark.parametrize("k,seed", [(15, 5), (31, 6), (47, 7)])
def test_oracle_graph_edits_and_graph_map_invariants(k, seed):
    """gko_graph_add_node / replace_start / replace_end (Graph.scala:172-176, 197-209) and gko_graph_get_graph_map (:90-119):
    the reference's own invariants hold before and after a node split — (start.seq ++ edge.seq) ends with end.seq
    (application.conf:73), the map holds sum(len) + nodes - edges entries (:96, :117), every node k-mer is a key."""
    import random
    from genome_amd import dna, synth
    rnd = random.Random(seed)
    g = "".join(rnd.choice("AGCT") for _ in range(1800))
    reads = []
    for _ in range(400):
        ln = rnd.randint(k + 5, min(255, k + 90))
        s = rnd.randrange(0, len(g) - ln + 1)
        r = g[s:s + ln]
        r = "".join(c if rnd.random() >= 0.02 else rnd.choice([x for x in "AGCT" if x != c]) for c in r)
        reads.append(R.rev_comp(r) if rnd.random() < 0.5 else r)
    binb = dna.reads_to_bin(reads)
    ref = O.PMap(k, 1)
    ref.count_reads(binb, len(reads)); ref.delete_lt(2)
    og = O.Graph(ref)

    def edges_of():
        e = og.edges()
        out = []
        for i in range(len(e["len"])):
            seq = synth.bases_to_str(e["bases"][e["off"][i]:e["off"][i] + e["len"][i]])
            out.append((dna.unpack(int(e["slo"][i]), int(e["shi"][i]), k), dna.unpack(int(e["elo"][i]), int(e["ehi"][i]), k), seq))
        return out

    def check():
        es = edges_of()
        assert all((s + q).endswith(t) for s, t, q in es)
        lo, hi, ie, ident, dist = og.graph_map_calls()
        assert len(lo) == og.total_edge_len() + og.num_nodes() - og.num_edges()
        assert int((ie == 0).sum()) == og.num_nodes()
        # an edge position at distance d names the window d bases into start.seq ++ edge.seq
        for i in list(range(0, len(lo), max(1, len(lo) // 200))):
            if ie[i]:
                info = og.edge_info(int(ident[i]))
                s = dna.unpack(*og.node_seq(info["start"]), k)
                q = next(x[2] for x in es if x[0] == s and x[2][0] == dna.BASES[info["first"]])
                assert (s + q)[int(dist[i]):int(dist[i]) + k] == dna.unpack(int(lo[i]), int(hi[i]), k)
        return es

    es = check()
    n0, e0 = og.num_nodes(), og.num_edges()
    by_start = {}
    for s, t, q in es:
        by_start.setdefault(s, []).append((s, t, q))
    targets = [s for s, v in by_start.items() if len(v) >= 2][:6]
    assert targets
    for j, s in enumerate(targets):
        oid = og.find_node(*dna.pack(s))
        new = og.add_node(*dna.pack(s))
        assert new == n0 + j + 1 and og.num_nodes() == n0 + j + 1
        q = sorted(by_start[s], key=lambda x: dna.BASES.index(x[2][0]))[0][2]
        oe = og.find_out_edge(oid, dna.BASES.index(q[0]))
        og.replace_start(oe, new)
        assert og.edge_info(oe)["start"] == new and og.find_out_edge(oid, dna.BASES.index(q[0])) == 0
        assert og.find_out_edge(new, dna.BASES.index(q[0])) == oe
        inc = [x for x in es if x[1] == s and x[0] != s]
        if inc:
            a, _, c = inc[0]
            oae = og.find_out_edge(og.find_node(*dna.pack(a)), dna.BASES.index(c[0]))
            if oae:
                og.replace_end(oae, new)
                assert og.edge_info(oae)["end"] == new
        assert og.num_edges() == e0
        check()
    og.simplify()
    check()


FORMULA = "A"      # scala-library 2.9.1 primitive Long.##: "A" = (int)(v ^ (v >>> 32)); see tests/golden/kat/make_hash_bit31.py


def test_long_hash_formula_is_one_visible_choice():
    """k <= 31 k-mers with bit 31 set and a non-zero high word: the two candidate readings of `Long.##` give different hash
    VALUES (every vector here) — the canonical orientation, a comparison of two such hashes, almost never differs (see the
    generator's note).  Both restatements implement FORMULA; the vectors hold the other formula's answers too, so a switch
    is one line per side plus this constant."""
    vec = json.load(open(os.path.join(GOLDEN, "kat", "hash_bit31.json")))["vectors"]
    assert len(vec) >= 20 and all(v["A"]["h_x"] != v["B"]["h_x"] or v["A"]["h_rc"] != v["B"]["h_rc"] for v in vec)
    for v in vec:
        k, want = v["k"], v[FORMULA]
        assert O.hash_code(v["x_lo"], 0, k) == want["h_x"] == R.hash_code(v["x"])
        assert O.hash_code(v["rc_lo"], 0, k) == want["h_rc"] == R.hash_code(v["rc"])
        lo, _ = O.canon(v["x_lo"], 0, k)
        assert R.canon(v["x"]) == want["canonical"] and lo == R.pack(want["canonical"])[0]


def test_oracle_paired_end_walking_resolves_a_repeated_kmer():
    """GraphSimplifier.scala:188-318 in the oracle: a k-mer that occurs twice in different contexts gives a node with two in-
    and two out-edges; read pairs that span it support exactly the two true (in, out) combinations; the split + simplify
    leaves longer edges and the same total sequence."""
    import random
    from oracle import oracle as O
    from oracle import pyref as R
    from genome_amd import dna
    k, glen = 21, 1600
    rnd = random.Random(5)
    g = [rnd.choice("AGCT") for _ in range(glen)]
    g[1100:1100 + k] = g[400:400 + k]
    g = "".join(g)
    reads = []
    for _ in range(3000):
        ins = rnd.randint(80, 100)
        s = rnd.randrange(0, glen - ins)
        frag = g[s:s + ins]
        if rnd.random() < 0.5:
            frag = R.rev_comp(frag)
        reads += [frag[:40], R.rev_comp(frag)[:40]]
    binb = dna.reads_to_bin(reads)
    ref = O.PMap(k, 1)
    ref.count_reads(binb, len(reads)); ref.delete_lt(2)
    og = O.Graph(ref)
    e0 = og.edges()
    sup = O.Support()
    walked = og.walk_pairs(sup, binb, len(reads) // 2, 60, 95)
    e1, e2, cnt = sup.items()
    assert walked > 0 and len(e1) > 0
    # every supported pair is a pair of CONSECUTIVE edges: e1 ends where e2 starts
    for a, b in zip(e1, e2):
        assert og.edge_info(int(a))["end"] == og.edge_info(int(b))["start"]
    removed, added = og.split_by_support(sup, 3)
    assert added >= 2                                  # the repeated k-mer's node and its reverse complement are split
    e = og.edges()
    assert max(e["len"]) > max(e0["len"])              # contigs grew through the resolved repeat
