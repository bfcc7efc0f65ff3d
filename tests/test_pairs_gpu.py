"""Paired-end walking — GraphSimplifier.startup (S/scripts/GraphSimplifier.scala:188-318): positions of the mates' first
k-mers through the position multimap, `annotate`, the bounded walks of WalkingActor (:33-127), the support matrix and the
node split — the library (device lookups + host-thread walks + device edits) against the oracle's LITERAL restatement
(priority-queue `reachable`, recursive `dfs` with its memo).  -m gpu.

Edge ids are arbitrary on both sides: support counts are compared after translation to content, (start k-mer, first base) of
both edges — unique on a freshly built graph.  The graphs after split + removeEdge + simplifyGraph are compared as canonical
node / edge lists (copies of a node share a sequence: lists, not sets)."""
import random
from collections import Counter

import numpy as np
import pytest

from genome_amd import dna, synth
from genome_amd.dnamap import Context, HipDNAMap
from genome_amd.graph import Support, buildGraph
from oracle import oracle as O
from oracle import pyref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def make_pairs(seed, k, glen=2400, nrep=3, L=40, npairs=5000, ins=(80, 100), err=0.0):
    """A random genome in which `nrep` k-mers occur twice in different contexts (a node with two in- and two out-edges each,
    which only read pairs can resolve), and read pairs of insert size `ins` from both strands."""
    rnd = random.Random(seed)
    g = [rnd.choice("AGCT") for _ in range(glen)]
    for _ in range(nrep):
        a = rnd.randrange(100, glen // 2 - 100)
        b = rnd.randrange(glen // 2 + 100, glen - 100)
        g[b:b + k] = g[a:a + k]
    g = "".join(g)
    reads = []
    for _ in range(npairs):
        ins_len = rnd.randint(*ins)
        s = rnd.randrange(0, glen - ins_len)
        frag = g[s:s + ins_len]
        if rnd.random() < 0.5:
            frag = R.rev_comp(frag)
        m1, m2 = frag[:L], R.rev_comp(frag)[:L]
        if err:
            m1 = "".join(c if rnd.random() >= err else rnd.choice([x for x in "AGCT" if x != c]) for c in m1)
            m2 = "".join(c if rnd.random() >= err else rnd.choice([x for x in "AGCT" if x != c]) for c in m2)
        reads += [m1, m2]
    return reads


def oracle_canonical(og):
    k = og.k
    nlo, nhi = og.nodes()
    nodes = sorted(dna.unpack(int(a), int(b), k) for a, b in zip(nlo, nhi))
    e = og.edges()
    edges = []
    for i in range(len(e["len"])):
        seq = synth.bases_to_str(e["bases"][e["off"][i]:e["off"][i] + e["len"][i]])
        edges.append((dna.unpack(int(e["slo"][i]), int(e["shi"][i]), k), dna.unpack(int(e["elo"][i]), int(e["ehi"][i]), k), seq))
    return nodes, sorted(edges)


def gpu_canonical(g):
    nodes, edges = g.canonical()
    return sorted(nodes), sorted(edges)


def gpu_support_by_content(g, k, sup):
    e1, e2, cnt = sup.items()
    ids = sorted(set(e1.tolist()) | set(e2.tolist()))
    if not ids:
        return Counter()
    info = g.edgesById(ids)
    nid = sorted({int(x) for x in info["start"]})
    ninfo = g.nodesById(nid)
    nkmer = {n: dna.unpack(int(ninfo["lo"][j]), int(ninfo["hi"][j]), k) for j, n in enumerate(nid)}
    key = {e: (nkmer[int(info["start"][j])], int(info["first"][j])) for j, e in enumerate(ids)}
    return Counter({(key[int(a)], key[int(b)]): int(c) for a, b, c in zip(e1, e2, cnt)})


def oracle_support_by_content(og, k, osup):
    e1, e2, cnt = osup.items()
    def key(e):
        info = og.edge_info(int(e))
        return (dna.unpack(*og.node_seq(info["start"]), k), info["first"])
    return Counter({(key(a), key(b)): int(c) for a, b, c in zip(e1, e2, cnt)})


@pytest.mark.parametrize("k,seed,err,rng", [(21, 1, 0.0, (60, 95)), (21, 2, 0.004, (55, 90)), (31, 3, 0.0, (50, 85)), (35, 4, 0.0, (50, 80))])
def test_walk_pairs_support_and_split_vs_oracle(ctx, k, seed, err, rng):
    reads = make_pairs(seed, k, err=err)
    binb = dna.reads_to_bin(reads)
    npairs = len(reads) // 2
    m, ref = HipDNAMap(ctx, k), O.PMap(k, 1)
    m.count_reads(binb, len(reads)); ref.count_reads(binb, len(reads))
    m.deleteAll_lt(2); ref.delete_lt(2)
    g, og = buildGraph(k, m), O.Graph(ref)
    assert gpu_canonical(g) == oracle_canonical(og)
    vm = g.getGraphMap()
    sup, osup = Support(ctx), O.Support()
    # two batches: the support accumulates (the reference streams the pairs through one pathsMap)
    half = npairs // 2
    cut = sum(1 + (len(r) + 3) // 4 for r in reads[:2 * half])
    g.walkPairs(vm, sup, binb[:cut], half, *rng)
    g.walkPairs(vm, sup, binb[cut:], npairs - half, *rng)
    walked = og.walk_pairs(osup, binb, npairs, *rng)
    pairs, bad, w = sup.sizes()
    assert w == walked and bad == osup.bad_pairs()
    want = oracle_support_by_content(og, k, osup)
    assert len(want) > 0 and max(want.values()) >= 3          # the fixture does produce support (and a split below)
    assert gpu_support_by_content(g, k, sup) == want
    # a mate shorter than k, a stream that ends inside a pair
    rm, nn = g.splitBySupport(sup, 3)
    orm, onn = og.split_by_support(osup, 3)                   # (the oracle's includes simplifyGraph :318)
    assert (rm, nn) == (orm, onn) and nn > 0
    g.simplifyGraph()
    assert gpu_canonical(g) == oracle_canonical(og)
    n, e, ln = g.counts()
    assert (n, e, ln) == (len(oracle_canonical(og)[0]), len(oracle_canonical(og)[1]), sum(len(x[2]) for x in oracle_canonical(og)[1]))
    # the position map of the OLD graph no longer describes this one
    with pytest.raises(Exception):
        g.walkPairs(vm, Support(ctx), binb, 10, *rng)
    vm.close(); g.close(); m.close()


def test_walk_pairs_edge_cases(ctx):
    k = 21
    reads = make_pairs(9, k, npairs=600)
    binb = dna.reads_to_bin(reads)
    m = HipDNAMap(ctx, k)
    m.count_reads(binb, len(reads)); m.deleteAll_lt(2)
    g = buildGraph(k, m)
    vm = g.getGraphMap()
    sup = Support(ctx)
    g.walkPairs(vm, sup, b"", 0)                               # nothing to do
    assert sup.sizes() == (0, 0, 0)
    short = dna.reads_to_bin(["ACGTACGT", "ACGTACGTAC"])         # mates shorter than k are skipped (:213)
    g.walkPairs(vm, sup, short, 1)
    assert sup.sizes() == (0, 0, 0)
    from genome_amd import _lib as L
    with pytest.raises(L.GkError):
        g.walkPairs(vm, sup, binb[:7], 1)                       # the stream ends inside a record
    with pytest.raises(L.GkError):
        g.walkPairs(vm, sup, binb, 10, 90, 50)                  # empty range
    assert g.removeEdgesById([]) == 0
    n_ids, e_ids = g.idBounds()
    ne = g.counts()[1]
    assert g.removeEdgesById([0, 0, e_ids + 5]) == 1 and g.counts()[1] == ne - 1      # each id once; unknown ids ignored
    vm.close(); g.close(); m.close()


def test_graph_builder_cli_walk_pairs_stage(ctx, tmp_path):
    """The C++ host side (genome.hpp PositionMap / Support / Graph::walkPairs / splitBySupport, graph_builder --walk-pairs) gives
    the oracle's graph and counters."""
    import json, os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "genome_amd", "host", "graph_builder")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "genome_amd", "csrc"), "host"])
    k, rng = 21, (60, 95)
    reads = make_pairs(11, k)
    binb = dna.reads_to_bin(reads)
    binf = tmp_path / "pairs.bin"
    binf.write_bytes(binb)
    out = tmp_path / "g"
    res = subprocess.run([exe, str(binf), str(len(reads) // 2), str(k), "--rounds", "2", "--no-retain", "--walk-pairs", "3", str(rng[0]), str(rng[1]),
                          "--out", str(out)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    stats = json.loads(res.stdout)
    ref = O.PMap(k, 1)
    ref.count_reads(binb, len(reads)); ref.delete_lt(2)
    og, osup = O.Graph(ref), O.Support()
    walked = og.walk_pairs(osup, binb, len(reads) // 2, *rng)
    orm, onn = og.split_by_support(osup, 3)
    w = stats["walk_pairs"]
    assert (w["orientations_walked"], w["bad_pairs"], w["supported_edge_pairs"]) == (walked, osup.bad_pairs(), len(osup.items()[0]))
    assert (w["removed_edges"], w["new_nodes"]) == (orm, onn)
    nodes, edges = oracle_canonical(og)
    assert sorted(open(str(out) + ".nodes.txt").read().split()) == nodes
    assert sorted(tuple(line.split()) for line in open(str(out) + ".edges.txt").read().splitlines()) == edges


@pytest.mark.parametrize("seed", range(8))
def test_walk_pairs_fuzz_small_cyclic_graphs(ctx, seed):
    """Small genomes with tandem and inverted repeats (cycles, self-loops, palindromic neighbourhoods in the graph), small k,
    short inserts: the walk's state space folds back on itself here, which is where a state-by-distance formulation and the
    reference's memoised recursion could part ways.  Support, bad pairs and the split graph must still be the oracle's."""
    rnd = random.Random(1000 + seed)
    k = rnd.choice([11, 13, 15])
    unit = "".join(rnd.choice("AGCT") for _ in range(rnd.randint(k + 2, 3 * k)))
    parts = []
    for _ in range(rnd.randint(4, 7)):
        t = rnd.random()
        if t < 0.35:
            parts.append(unit * rnd.randint(1, 3))                                   # tandem copies: cycles
        elif t < 0.5:
            parts.append(R.rev_comp(unit))                                           # inverted copy
        else:
            parts.append("".join(rnd.choice("AGCT") for _ in range(rnd.randint(30, 120))))
    g = "".join(parts)
    L, reads = 30, []
    for _ in range(2500):
        ins = rnd.randint(45, 80)
        if ins > len(g):
            continue
        s = rnd.randrange(0, len(g) - ins + 1)
        frag = g[s:s + ins]
        if rnd.random() < 0.5:
            frag = R.rev_comp(frag)
        reads += [frag[:L], R.rev_comp(frag)[:L]]
    binb = dna.reads_to_bin(reads)
    m, ref = HipDNAMap(ctx, k), O.PMap(k, 1)
    m.count_reads(binb, len(reads)); ref.count_reads(binb, len(reads))
    m.deleteAll_lt(2); ref.delete_lt(2)
    g_, og = buildGraph(k, m), O.Graph(ref)
    assert gpu_canonical(g_) == oracle_canonical(og)
    lo, hi = rnd.choice([(20, 60), (30, 70), (10, 45)])
    vm = g_.getGraphMap()
    sup, osup = Support(ctx), O.Support()
    g_.walkPairs(vm, sup, binb, len(reads) // 2, lo, hi)
    walked = og.walk_pairs(osup, binb, len(reads) // 2, lo, hi)
    assert sup.sizes()[1:] == (osup.bad_pairs(), walked)
    assert gpu_support_by_content(g_, k, sup) == oracle_support_by_content(og, k, osup)
    cutoff = rnd.choice([1, 2, 5])
    assert g_.splitBySupport(sup, cutoff) == og.split_by_support(osup, cutoff)
    g_.simplifyGraph()
    assert gpu_canonical(g_) == oracle_canonical(og)
    vm.close(); g_.close(); m.close()


@pytest.mark.parametrize("ragged", [False, True])
def test_walk_pairs_large_uniform_stream_is_cut_by_threads(ctx, ragged):
    """From 65 536 pairs up a stream of equal-length records is cut into its k-mers by several host threads, each checking the
    length bytes of its share; one that turns out ragged (here: one mate in the middle a base shorter) is walked serially.
    Both give the oracle's support counts."""
    k, rng = 21, (60, 95)
    reads = make_pairs(11, k, npairs=70000)
    if ragged:
        reads[70001] = reads[70001][:-1]
    binb = dna.reads_to_bin(reads)
    npairs = len(reads) // 2
    m, ref = HipDNAMap(ctx, k), O.PMap(k, 1)
    m.count_reads(binb, len(reads)); ref.count_reads(binb, len(reads))
    m.deleteAll_lt(2); ref.delete_lt(2)
    g, og = buildGraph(k, m), O.Graph(ref)
    assert gpu_canonical(g) == oracle_canonical(og)
    vm = g.getGraphMap()
    sup, osup = Support(ctx), O.Support()
    g.walkPairs(vm, sup, binb, npairs, *rng)
    walked = og.walk_pairs(osup, binb, npairs, *rng)
    pairs, bad, w = sup.sizes()
    assert w == walked and bad == osup.bad_pairs()
    want = oracle_support_by_content(og, k, osup)
    assert len(want) > 0
    assert gpu_support_by_content(g, k, sup) == want
    vm.close(); g.close(); m.close()


@pytest.mark.parametrize("mode", ["pairs_host", "test_pairs_small_sets"])
@pytest.mark.parametrize("k,seed,err,rng", [(21, 2, 0.004, (55, 90)), (35, 4, 0.0, (50, 80))])
def test_walk_pairs_host_form_and_overflow_fallback_give_the_same_support(ctx, mode, k, seed, err, rng):
    """The walks run one wave per pair orientation on the device; an orientation whose sets outgrow the wave's LDS goes to the
    host walker instead (exact either way), and `pairs_host` = 1 sends ALL of them there (the round-2 form, kept as an A/B
    switch).  With the device's sets made tiny (test_pairs_small_sets) most orientations overflow: the mix of device-walked and
    host-walked orientations must still add up to the oracle's support counts, bad pairs and walked orientations."""
    reads = make_pairs(seed, k, err=err)
    binb = dna.reads_to_bin(reads)
    npairs = len(reads) // 2
    m, ref = HipDNAMap(ctx, k), O.PMap(k, 1)
    m.count_reads(binb, len(reads)); ref.count_reads(binb, len(reads))
    m.deleteAll_lt(2); ref.delete_lt(2)
    g, og = buildGraph(k, m), O.Graph(ref)
    vm = g.getGraphMap()
    sup, osup = Support(ctx), O.Support()
    ctx.set_option(mode, 1)
    try:
        g.walkPairs(vm, sup, binb, npairs, *rng)
    finally:
        ctx.set_option(mode, 0)
    walked = og.walk_pairs(osup, binb, npairs, *rng)
    assert sup.sizes()[1:] == (osup.bad_pairs(), walked)
    assert gpu_support_by_content(g, k, sup) == oracle_support_by_content(og, k, osup)
    assert g.splitBySupport(sup, 3) == og.split_by_support(osup, 3)
    vm.close(); g.close(); m.close()
