"""Property-based cross-check of the two independent CPU restatements (C oracle vs Python pyref):
random read sets, random k over every supported width, random partition counts and thresholds —
tables (incl. literal container state), graphs, bubble removal and simplification must agree."""
import random

import numpy as np
from hypothesis import HealthCheck, given, settings, strategies as st

from genome_amd import synth
from oracle import oracle as O
from oracle import pyref as R

KS = st.sampled_from([2, 3, 7, 12, 16, 21, 31, 34, 40, 63, 64])


def _reads(seed, n, k, glen, err, ragged):
    rnd = random.Random(seed)
    g = "".join(rnd.choice("AGCT") for _ in range(glen))
    out = []
    for _ in range(n):
        ln = rnd.randint(max(1, k - 2), min(255, k + 25)) if ragged else min(255, k + 12)
        ln = min(ln, glen)
        s = rnd.randrange(0, glen - ln + 1)
        r = g[s:s + ln]
        if rnd.random() < 0.5:
            r = R.rev_comp(r)
        out.append("".join(c if rnd.random() >= err else rnd.choice([x for x in "AGCT" if x != c]) for c in r))
    return out


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(seed=st.integers(0, 10**6), k=KS, P=st.integers(1, 5), rounds=st.integers(1, 4), ragged=st.booleans(),
       err=st.sampled_from([0.0, 0.01, 0.05]))
def test_tables_agree(seed, k, P, rounds, ragged, err):
    reads = _reads(seed, 25, k, 150, err, ragged)
    pm = O.PMap(k, P)
    pm.count_reads(R.reads_to_bin(reads), len(reads))
    pr = R.extract_filtered_kmers(reads, k, rounds, P, do_filter=False)
    for stage in range(2):
        items = pr.sorted_items()
        lo, hi, cnt = pm.export_sorted()
        assert [R.pack(s) for s, _ in items] == list(zip(lo.tolist(), hi.tolist()))
        assert [c for _, c in items] == cnt.tolist()
        for p in range(P):
            assert pm.part_stats(p) == (pr.parts[p].size, pr.parts[p].bins, pr.parts[p].rescales)
        pm.delete_lt(rounds); pr.delete_lt(rounds)


@settings(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(seed=st.integers(0, 10**6), k=st.sampled_from([5, 8, 11, 15, 34, 36]), P=st.integers(1, 3), hap=st.booleans())
def test_graphs_agree(seed, k, P, hap):
    rnd = random.Random(seed)
    g = "".join(rnd.choice("AGCT") for _ in range(120))
    haps = [g]
    if hap:
        h = list(g); h[60] = "AGCT"[("AGCT".index(h[60]) + 1) % 4]; haps.append("".join(h))
    reads = []
    for hp in haps:
        for i in range(0, 120 - (k + 10) + 1, 2):
            reads += [hp[i:i + k + 10], R.rev_comp(hp[i:i + k + 10])]
    pm = O.PMap(k, P); pm.count_reads(R.reads_to_bin(reads), len(reads)); pm.delete_lt(2)
    pr = R.extract_filtered_kmers(reads, k, 2, P)
    cg, pg = O.Graph(pm), R.build_graph(k, pr)

    def canon(cgraph):
        nlo, nhi = cgraph.nodes()
        nodes = [R.unpack(int(a), int(b), k) for a, b in zip(nlo, nhi)]
        e = cgraph.edges()
        edges = [(R.unpack(int(e["slo"][i]), int(e["shi"][i]), k), R.unpack(int(e["elo"][i]), int(e["ehi"][i]), k),
                  synth.bases_to_str(e["bases"][e["off"][i]:e["off"][i] + e["len"][i]])) for i in range(len(e["len"]))]
        return nodes, edges

    assert canon(cg) == pg.canonical()
    cg.remove_bubbles(); pg.remove_bubbles()
    assert canon(cg) == pg.canonical()
    cg.simplify(); pg.simplify()
    assert canon(cg) == pg.canonical()
    for nid, n in pg.nodes.items():       # out-edge insertion order (Map1..Map4) agrees too
        lo, hi = R.pack(n["seq"])
        assert cg.out_order(lo, hi) == ["AGCT".index(b) for b, _ in n["outs"]]
