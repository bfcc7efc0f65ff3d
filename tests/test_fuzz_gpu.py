"""Property-based GPU-vs-oracle check (-m gpu): random k over every supported width, ragged random reads with
errors, one or several batches, any insert path, optional singleton pre-filter, random filter threshold; the table
(sorted content) and the graph after build / removeBubbles / simplifyGraph must equal the C oracle's bit for bit."""
import os
import random

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from genome_amd.dnamap import Context, HipDNAMap
from genome_amd.graph import buildGraph
from genome_amd.prefilter import HipPrefilter
from oracle import oracle as O
from oracle import pyref as R

pytestmark = pytest.mark.gpu
_CTX = None


def ctx():
    global _CTX
    if _CTX is None:
        _CTX = Context(0)
    return _CTX


def _reads(rnd, n, k, glen, err, haplotypes):
    g = "".join(rnd.choice("AGCT") for _ in range(glen))
    haps = [g]
    for _ in range(haplotypes):
        h = list(g)
        for _ in range(3):
            i = rnd.randrange(glen)
            h[i] = rnd.choice([c for c in "AGCT" if c != h[i]])
        haps.append("".join(h))
    out = []
    for _ in range(n):
        hp = rnd.choice(haps)
        ln = min(glen, rnd.randint(max(1, k - 2), min(255, k + 60)))
        s = rnd.randrange(0, glen - ln + 1)
        r = hp[s:s + ln]
        if rnd.random() < 0.5:
            r = R.rev_comp(r)
        out.append("".join(c if rnd.random() >= err else rnd.choice([x for x in "AGCT" if x != c]) for c in r))
    return out


def _oracle_graph(og, k):
    from genome_amd import dna, synth
    nlo, nhi = og.nodes()
    nodes = [dna.unpack(int(a), int(b), k) for a, b in zip(nlo, nhi)]
    e = og.edges()
    edges = [(dna.unpack(int(e["slo"][i]), int(e["shi"][i]), k), dna.unpack(int(e["elo"][i]), int(e["ehi"][i]), k),
              synth.bases_to_str(e["bases"][e["off"][i]:e["off"][i] + e["len"][i]])) for i in range(len(e["len"]))]
    return nodes, edges


@settings(max_examples=int(os.environ.get("GK_FUZZ_EXAMPLES", "150")), deadline=None, derandomize="GK_FUZZ_EXAMPLES" not in os.environ, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 10**6), k=st.sampled_from([2, 4, 9, 15, 21, 27, 31, 34, 35, 48, 62, 63, 64]),
       path=st.sampled_from(["auto", "direct", "partitioned"]), batches=st.integers(1, 3), rounds=st.integers(1, 4),
       err=st.sampled_from([0.0, 0.01, 0.04]), haplotypes=st.integers(0, 2), prefilter=st.booleans(), hint=st.sampled_from([0, 64, 20000]))
def test_count_filter_graph_match_the_oracle(seed, k, path, batches, rounds, err, haplotypes, prefilter, hint):
    rnd = random.Random(seed)
    glen = rnd.randint(max(k + 5, 40), 400)
    reads = _reads(rnd, rnd.randint(1, 160), k, glen, err, haplotypes)
    cut = sorted(rnd.randrange(len(reads) + 1) for _ in range(batches - 1))
    parts = [reads[a:b] for a, b in zip([0] + cut, cut + [len(reads)])]
    ref = O.PMap(k, 1)
    m = HipDNAMap(ctx(), k, hint)
    m.set_insert_path(path)
    use_pf = prefilter and rounds >= 2
    pf = None
    if use_pf:
        pf = HipPrefilter(ctx(), k, rnd.choice([1, 500, 100000]))
        for p in parts:
            pf.add_reads(R.reads_to_bin(p), len(p))
    for p in parts:
        b = R.reads_to_bin(p)
        occ = ref.count_reads(b, len(p))
        if pf:
            looked, _ = pf.count_reads(m, b, len(p))
            assert looked == occ
        else:
            assert m.count_reads(b, len(p)) == occ
    if not pf:
        for a, b in zip(m.sorted_items(), ref.export_sorted()):
            assert np.array_equal(a, b)
    m.deleteAll_lt(rounds); ref.delete_lt(rounds)
    for a, b in zip(m.sorted_items(), ref.export_sorted()):
        assert np.array_equal(a, b)
    assert m.size() == ref.size()
    g = buildGraph(k, m)
    og = O.Graph(ref)
    assert g.canonical() == _oracle_graph(og, k)
    g.removeBubbles(); og.remove_bubbles()
    assert g.canonical() == _oracle_graph(og, k)
    g.simplifyGraph(); og.simplify()
    assert g.canonical() == _oracle_graph(og, k)
    g.close(); m.close()
    if pf:
        pf.close()
