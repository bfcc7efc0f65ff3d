"""Exact two-pass singleton pre-filter against the CPU oracle (-m gpu).

The reference has no such stage on its hot path (its BloomFilter, S/ds/BloomFilter.scala, is unused), so the
checker is the property the design promises: for rounds >= 2 the table after deleteAll(v < rounds)
(FreqFilter.scala:55) is bit-identical to the plain path's — i.e. to the oracle's — and before the filter
every k-mer seen at least twice already holds its exact count while every other entry has count 1.
"""
import random

import numpy as np
import pytest

from genome_amd import dna, synth
from genome_amd.dnamap import Context, HipDNAMap
from genome_amd.freqfilter import PairedEndData, extractFilteredKmers
from genome_amd.prefilter import HipPrefilter
from oracle import oracle as O
from oracle import pyref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def same(got, want):
    for a, b in zip(got, want):
        assert a.shape == b.shape and np.array_equal(a, b)


def check_prefiltered_table(m, ref):
    """m = table after pass 2, ref = oracle PMap after plain counting."""
    lo, hi, cnt = ref.export_sorted()
    want = {(int(a), int(b)): int(c) for a, b, c in zip(lo, hi, cnt)}
    glo, ghi, gcnt = m.sorted_items()
    got = {(int(a), int(b)): int(c) for a, b, c in zip(glo, ghi, gcnt)}
    assert set(got) <= set(want)
    for key, c in want.items():
        if c >= 2:
            assert got.get(key) == c, (key, c, got.get(key))       # exact count, never missing
        else:
            assert got.get(key, 1) == 1                             # absent, or a false positive with count 1
    return sum(1 for c in want.values() if c == 1), sum(1 for key, c in want.items() if c == 1 and key in got)


@pytest.mark.parametrize("k", [5, 21, 31, 47, 64])
def test_ragged_host_stream(ctx, k):
    rnd = random.Random(100 + k)
    g = "".join(rnd.choice("AGCT") for _ in range(3000))
    reads = []
    for _ in range(600):
        ln = rnd.randint(max(1, k - 3), min(255, k + 90))
        st = rnd.randrange(0, len(g) - ln + 1)
        r = g[st:st + ln]
        if rnd.random() < 0.5:
            r = R.rev_comp(r)
        reads.append("".join(c if rnd.random() >= 0.03 else rnd.choice([x for x in "AGCT" if x != c]) for c in r))
    reads += ["", "AG", "".join(rnd.choice("AGCT") for _ in range(255))]
    binb = dna.reads_to_bin(reads)
    ref = O.PMap(k, 1)
    occ = ref.count_reads(binb, len(reads))
    for expected in (1, ref.size(), 50 * ref.size()):          # absurdly small, right, generous filter
        pf = HipPrefilter(ctx, k, expected)
        pf.add_reads(binb, len(reads))
        m = HipDNAMap(ctx, k, 64)
        looked, admitted = pf.count_reads(m, binb, len(reads))
        assert looked == occ and admitted <= occ
        st = pf.stats()
        assert st["windows_added"] == occ and st["seen_once"] + st["seen_twice_or_more"] <= st["buckets"]
        singles, fp = check_prefiltered_table(m, ref)
        if expected >= 50 * ref.size() and singles > 200 and k >= 21:
            assert fp < singles // 4, "a generous filter must keep most singletons out"
        for rounds in (2, 3, 5):
            m.deleteAll_lt(rounds)
            r2 = O.PMap(k, 1); r2.count_reads(binb, len(reads)); r2.delete_lt(rounds)
            same(m.sorted_items(), r2.export_sorted())
        m.close(); pf.close()


@pytest.mark.parametrize("k,L_", [(21, 100), (31, 150), (55, 150)])
def test_device_records_in_chunks(ctx, k, L_):
    """Pass 1 fed in two chunks, pass 2 in one call: only the union matters."""
    n, G, e, cid = 4000, 30000, 0.02, 11
    rec = synth.reads_mode_g(n, L_, G, e, cid)
    d = ctx.alloc(rec.size + 64)
    ctx.synth_reads(d, n, L_, "G", cid, 0, G, e)
    ref = O.PMap(k, 1)
    occ = ref.count_reads(rec.tobytes(), n)
    pf = HipPrefilter(ctx, k, ref.size())
    half = n // 3
    pf.add_reads_dev(d, half, L_)
    pf.add_reads_dev(d + half * rec.shape[1], n - half, L_)
    m = HipDNAMap(ctx, k, 1000)
    looked, admitted = pf.count_reads_dev(m, d, n, L_)
    assert looked == occ == n * (L_ - k + 1)
    singles, fp = check_prefiltered_table(m, ref)
    assert m.size() < ref.size(), "the filter kept nothing out"
    m.deleteAll_lt(3)
    ref.delete_lt(3)
    same(m.sorted_items(), ref.export_sorted())
    m.close(); pf.close(); ctx.free(d)


def test_freqfilter_entry_point_and_errors(ctx):
    rnd = random.Random(5)
    g = "".join(rnd.choice("AGCT") for _ in range(2000))
    reads = [g[s:s + 80] for s in (rnd.randrange(0, len(g) - 80) for _ in range(400))]
    data = PairedEndData(len(reads) // 2, dna.reads_to_bin(reads))
    a = extractFilteredKmers(data, 21, 3, ctx=ctx)
    b = extractFilteredKmers(data, 21, 3, ctx=ctx, prefilter_distinct=5000)
    same(a.sorted_items(), b.sorted_items())
    a.close(); b.close()
    with pytest.raises(ValueError):
        extractFilteredKmers(data, 21, 1, ctx=ctx, prefilter_distinct=5000)
    # k mismatch between filter and table
    from genome_amd._lib import GkError
    pf = HipPrefilter(ctx, 21, 100)
    m = HipDNAMap(ctx, 31)
    with pytest.raises(GkError):
        pf.count_reads(m, data.bin, 10)
    with pytest.raises(GkError):
        HipPrefilter(ctx, 33, 100)
    m.close(); pf.close()
