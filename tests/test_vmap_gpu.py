"""The multimap half of `trait DNAMap` (putNew / getAll / update(key, v) / apply, S/ds/ArrayDNAMap.scala:90-162), the position map
Graph.getGraphMap builds on it (S/data/graph/Graph.scala:90-119) and the by-id graph edits of the simplifier's node split
(addNode / replaceStart / replaceEnd, :172-176, :197-209) — GPU against the oracle (-m gpu).

getAll's list order is the reference's probe order (its own hash and resize history): callers use the values as a set
(GraphSimplifier.scala:192-217), so parity is on the multiset of values per key.  Node / edge ids are arbitrary on both sides
(SURVEY.md §8c): positions are compared after translation to content — a node position to the node's k-mer, an edge position to
(start k-mer, first base, distance).
"""
import random
from collections import Counter

import numpy as np
import pytest

from genome_amd import _lib as L
from genome_amd import dna, synth
from genome_amd.dnamap import Context, HipDNAMap, HipValueMap, pos_decode
from genome_amd.graph import buildGraph
from oracle import oracle as O
from oracle import pyref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def rand_kmer(rnd, k):
    return "".join(rnd.choice("AGCT") for _ in range(k))


@pytest.mark.parametrize("k", [5, 21, 31, 34, 55, 64])
def test_put_new_get_all_update_apply_vs_oracle(ctx, k):
    rnd = random.Random(k)
    pool = [rand_kmer(rnd, k) for _ in range(700)]
    vm = HipValueMap(ctx, k, 16)                      # tiny hint: growth (rehash of a multimap) is exercised
    om = O.lib().gko_map_new(k)
    # putNew: many duplicates, in several batches
    seq = [rnd.choice(pool) for _ in range(5000)]
    vals = list(range(1, 5001))
    for a in range(0, 5000, 1700):
        vm.putNew_batch(seq[a:a + 1700], vals[a:a + 1700])
    for s, v in zip(seq, vals):
        O.lib().gko_map_put_new(om, O.km(*dna.pack(s)), v)
    assert vm.size() == 5000 == O.lib().gko_map_size(om)
    probe = pool + [rand_kmer(rnd, k) for _ in range(50)]
    got = vm.getAll_batch(probe)
    buf = (O.C.c_int32 * 64)()
    for s, g in zip(probe, got):
        n = O.lib().gko_map_get_all(om, O.km(*dna.pack(s)), buf, 64)
        assert Counter(int(x) for x in g) == Counter(buf[i] for i in range(n)), s
    # apply = some value stored under the key (the reference returns the first in ITS probe order)
    v, f = vm.apply_batch(probe)
    for s, vv, ff, g in zip(probe, v, f, got):
        assert bool(ff) == (len(g) > 0)
        if ff:
            assert int(vv) in set(int(x) for x in g)
    # update(key, v): on a map WITHOUT duplicates it is insert-or-overwrite; the last occurrence in a batch wins
    um = HipValueMap(ctx, k)
    ou = O.lib().gko_map_new(k)
    useq = [rnd.choice(pool[:200]) for _ in range(3000)]
    uvals = [rnd.randrange(1, 1 << 30) for _ in range(3000)]
    um.update_batch(useq[:2000], uvals[:2000])
    um.update_batch(useq[2000:], uvals[2000:])
    for s, x in zip(useq, uvals):
        O.lib().gko_map_update_set(ou, O.km(*dna.pack(s)), x)
    assert um.size() == O.lib().gko_map_size(ou) == len(set(useq))
    one = O.C.c_int32()
    uv, uf = um.apply_batch(pool[:250])
    for s, vv, ff in zip(pool[:250], uv, uf):
        has = O.lib().gko_map_get(ou, O.km(*dna.pack(s)), O.C.byref(one))
        assert bool(ff) == bool(has)
        if has:
            assert int(vv) == one.value, s
    lo, hi, val = um.items()
    assert len(lo) == um.size() and len(set(zip(lo.tolist(), hi.tolist()))) == len(lo)
    # key length is checked like everywhere else (ArrayDNAMap.scala:182)
    with pytest.raises(AssertionError):
        vm.putNew("T" * (k + 1), 1)
    O.lib().gko_map_free(om); O.lib().gko_map_free(ou)
    vm.close(); um.close()


def _graph_pair(ctx, k, seed, nreads=500, err=0.01):
    rnd = random.Random(seed)
    g = "".join(rnd.choice("AGCT") for _ in range(1800))
    reads = []
    for _ in range(nreads):
        ln = rnd.randint(k + 5, min(255, k + 90))
        s = rnd.randrange(0, len(g) - ln + 1)
        r = g[s:s + ln]
        r = "".join(c if rnd.random() >= err else rnd.choice([x for x in "AGCT" if x != c]) for c in r)
        reads.append(R.rev_comp(r) if rnd.random() < 0.5 else r)
    binb = dna.reads_to_bin(reads)
    m, ref = HipDNAMap(ctx, k), O.PMap(k, 1)
    m.count_reads(binb, len(reads)); ref.count_reads(binb, len(reads))
    m.deleteAll_lt(2); ref.delete_lt(2)
    return m, ref, buildGraph(k, m), O.Graph(ref)


def _gpu_positions(g, k, values):
    """translate position values to content: ('N', node k-mer) | ('E', start k-mer, first base, dist)"""
    dec = [pos_decode(v) for v in values]
    eids = sorted({i for t, i, _ in dec if t == "E"})
    einfo = g.edgesById(eids) if eids else None
    emap = {e: j for j, e in enumerate(eids)}
    nids = sorted({i for t, i, _ in dec if t == "N"} | ({int(x) for x in einfo["start"]} if eids else set()))
    ninfo = g.nodesById(nids) if nids else None
    nmap = {n: j for j, n in enumerate(nids)}
    out = []
    for t, i, d in dec:
        if t == "N":
            j = nmap[i]
            out.append(("N", dna.unpack(int(ninfo["lo"][j]), int(ninfo["hi"][j]), k)))
        else:
            j = emap[i]
            assert einfo["alive"][j]
            s = nmap[int(einfo["start"][j])]
            out.append(("E", dna.unpack(int(ninfo["lo"][s]), int(ninfo["hi"][s]), k), int(einfo["first"][j]), d))
    return out


def _oracle_positions(og, k, calls, idx):
    lo, hi, ie, ident, dist = calls
    out = []
    for i in idx:
        if ie[i]:
            info = og.edge_info(int(ident[i]))
            slo, shi = og.node_seq(info["start"])
            out.append(("E", dna.unpack(slo, shi, k), info["first"], int(dist[i])))
        else:
            out.append(("N", dna.unpack(int(lo[i]), int(hi[i]), k)))
    return out


@pytest.mark.parametrize("k,seed", [(11, 1), (31, 2), (35, 3), (63, 4)])
@pytest.mark.parametrize("simplified", [False, True])
def test_get_graph_map_vs_oracle(ctx, k, seed, simplified):
    m, ref, g, og = _graph_pair(ctx, k, seed)
    if simplified:                                   # longer edges: interior k-mers at distances up to hundreds
        g.removeBubbles(); og.remove_bubbles(); g.simplifyGraph(); og.simplify()
    vm = g.getGraphMap()
    calls = og.graph_map_calls()
    n, e, ln = g.counts()
    assert vm.size() == len(calls[0]) == ln + n - e                     # Graph.scala:96,117
    want = {}
    for i in range(len(calls[0])):
        want.setdefault((int(calls[0][i]), int(calls[1][i])), []).append(i)
    keys = list(want)
    random.Random(seed).shuffle(keys)
    keys = keys[:1500]
    absent = [dna.pack(rand_kmer(random.Random(9), k)) for _ in range(20)]
    lo = np.array([a for a, _ in keys + absent], np.uint64)
    hi = np.array([b for _, b in keys + absent], np.uint64)
    got = vm.getAll_batch((lo, hi))
    for key, vals in zip(keys, got[:len(keys)]):
        assert Counter(_gpu_positions(g, k, vals.tolist())) == Counter(_oracle_positions(og, k, calls, want[key])), key
    assert all(len(v) == 0 or (a, b) in want for (a, b), v in zip(absent, got[len(keys):]))
    # CheckGraph.scala:48-55: every k-mer of every read-supported unitig is in the map -> here: every node k-mer is found
    nlo, nhi = g.getNodes()
    _, found = vm.apply_batch((nlo, nhi))
    assert found.all()
    vm.close(); g.close(); m.close()


@pytest.mark.parametrize("k,seed", [(15, 5), (31, 6), (47, 7)])
def test_add_node_replace_start_end_vs_oracle(ctx, k, seed):
    """The node split of GraphSimplifier.scala:296-309: a copy of a node takes over some of its edges."""
    m, ref, g, og = _graph_pair(ctx, k, seed, nreads=400, err=0.02)

    def canon_gpu():
        nodes, edges = g.canonical()
        return sorted(nodes), sorted(edges)

    def canon_oracle():
        nlo, nhi = og.nodes()
        nodes = sorted(dna.unpack(int(a), int(b), k) for a, b in zip(nlo, nhi))
        e = og.edges()
        edges = []
        for i in range(len(e["len"])):
            seq = synth.bases_to_str(e["bases"][e["off"][i]:e["off"][i] + e["len"][i]])
            edges.append((dna.unpack(int(e["slo"][i]), int(e["shi"][i]), k), dna.unpack(int(e["elo"][i]), int(e["ehi"][i]), k), seq))
        return nodes, sorted(edges)

    assert canon_gpu() == canon_oracle()
    nodes, edges = g.canonical()
    rnd = random.Random(seed)
    # pick nodes with at least two out-edges and at least one in-edge: split them
    by_start = {}
    for s, t, q in edges:
        by_start.setdefault(s, []).append((s, t, q))
    targets = [s for s, es in by_start.items() if len(es) >= 2][:6]
    assert targets
    for s in targets:
        gid, _ = g.nodeId(s)
        oid = og.find_node(*dna.pack(s))
        assert gid is not None and oid
        new_g, new_o = g.addNode(s), og.add_node(*dna.pack(s))               # val newNode = graph.addNode(node.seq)  :297
        # move the node's first out-edge (by base) to the copy
        s_, t_, q_ = sorted(by_start[s], key=lambda x: dna.BASES.index(x[2][0]))[0]
        _, ge = g.nodeId(s, q_[0])
        oe = og.find_out_edge(oid, dna.BASES.index(q_[0]))
        assert ge is not None and oe
        g.replaceStart(ge, new_g); og.replace_start(oe, new_o)               # :306
        # and one edge that ENDS in the node, if any, to the copy as well
        incoming = [x for x in edges if x[1] == s and x[0] != s]
        if incoming:
            a, b, c = incoming[0]
            ga, gae = g.nodeId(a, c[0])
            oa = og.find_node(*dna.pack(a))
            oae = og.find_out_edge(oa, dna.BASES.index(c[0]))
            if gae is not None and oae:
                g.replaceEnd(gae, new_g); og.replace_end(oae, new_o)         # :307
        assert canon_gpu() == canon_oracle(), s
        info = g.nodesById([gid, new_g])
        assert info["alive"].all()
        assert int(info["out_deg"][1]) == 1
    # the edited graph still simplifies like the oracle's
    g.simplifyGraph(); og.simplify()
    assert canon_gpu() == canon_oracle()
    # errors: ids that do not exist
    with pytest.raises(L.GkError):
        g.replaceStart(0xfffffff0, 0)
    with pytest.raises(L.GkError):
        g.replaceEnd(0, 0xfffffff0)
    with pytest.raises(L.GkError):
        g.addNode("T" * (k + 1))
    g.close(); m.close()


@pytest.mark.parametrize("k", [21, 31, 47])
def test_checkgraph_invariant_every_kmer_of_the_genome_is_in_the_graph_map(ctx, k):
    """The reference's own validation script (S/scripts/CheckGraph.scala:48-55): every k-window of the genome the reads came
    from must be found in Graph.getGraphMap (`graphMap.contains(read)`), and its contig statistics (:37-41: edges longer than a
    cutoff, their count, summed length, median, maximum).  A 60 kbp genome with two repeated 300-base stretches, error-free
    150-base mates tiled every 25 bases on both strands (every k-mer covered), count -> buildGraph -> getGraphMap: all
    genome k-mers found (both strands — the graph holds a node per strand), k-mers of a sequence that is not in the genome
    are not, and the contig statistics equal the oracle's."""
    rnd = random.Random(5 + k)
    G = 60000
    g = [rnd.choice("AGCT") for _ in range(G)]
    for _ in range(2):
        a, b = rnd.randrange(1000, G // 2 - 1000), rnd.randrange(G // 2 + 1000, G - 1000)
        g[b:b + 300] = g[a:a + 300]
    g = "".join(g)
    reads = []
    for s in range(0, G - 150 + 1, 25):
        frag = g[s:s + 150]
        reads += [frag, R.rev_comp(frag)]
    reads += [g[G - 150:], R.rev_comp(g[G - 150:])]
    binb = dna.reads_to_bin(reads)
    m, ref = HipDNAMap(ctx, k), O.PMap(k, 1)
    m.count_reads(binb, len(reads)); ref.count_reads(binb, len(reads))
    graph, og = buildGraph(k, m), O.Graph(ref)
    vm = graph.getGraphMap()
    windows = [g[i:i + k] for i in range(G - k + 1)]
    for chunk in (windows, [R.rev_comp(w) for w in windows[::7]]):
        _, found = vm.apply_batch(chunk)
        assert found.all(), f"{(~found).sum()} genome k-mers are not in the graph map"
    alien = "".join(rnd.choice("AGCT") for _ in range(3000))
    in_genome = set(windows) | {R.rev_comp(w) for w in windows}
    aw = [alien[i:i + k] for i in range(len(alien) - k + 1)]
    _, found = vm.apply_batch(aw)
    assert [bool(f) for f in found] == [w in in_genome for w in aw]
    # :37-41 contigs = edge sequences longer than the cutoff (200 there; 100 here so that the small graph has some)
    lens = sorted(len(e[2]) for e in graph.canonical()[1] if len(e[2]) > 100)
    oe = og.edges()
    olens = sorted(int(x) for x in oe["len"] if int(x) > 100)
    assert lens == olens and len(lens) > 0
    assert (len(lens), sum(lens), lens[len(lens) // 2], lens[-1]) == (len(olens), sum(olens), olens[len(olens) // 2], olens[-1])
    vm.close(); graph.close(); m.close()


def test_put_new_of_one_key_beyond_a_segment_is_refused_with_a_clear_message(ctx):
    """The reference's putNew probes the whole table; here every copy of a key lives in the one segment its hash names (include/
    genome_amd.h states the bound): 2048 copies of an 8-byte key fit, many more do not — GK_E_CAPACITY with a message that names
    the bound, nothing aborts, and the entries that did fit are there (a failed batch is not rolled back)."""
    k = 21
    vm = HipValueMap(ctx, k, 1 << 14)
    key = "ACGT" * 5 + "A"
    other = "TTGCA" * 4 + "G"
    vm.putNew_batch([key] * 1500 + [other] * 3, list(range(1503)))
    assert len(vm.getAll(key)) == 1500 and sorted(vm.getAll(other)) == [1500, 1501, 1502]
    with pytest.raises(L.GkError) as e:
        vm.putNew_batch([key] * 1000, list(range(2000, 3000)))
    assert e.value.code == L.GK_E_CAPACITY and "segment" in str(e.value) and "2048" in str(e.value)
    n = len(vm.getAll(key))
    assert 1500 < n <= 2048 and vm.size() == n + 3
    assert sorted(vm.getAll(other)) == [1500, 1501, 1502]
    vm.close()
