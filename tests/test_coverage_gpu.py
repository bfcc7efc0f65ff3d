"""GPU parity tests of the insert pipeline on REPEAT-HEAVY input (sequencing coverage) and of the boundary's
hardening (-m gpu): a call that brings far more windows than the table has room for must size the table for
the DISTINCT keys it really holds (the distinct-key sample), the exact fine level (range matrix, no global
atomics) and the heavy-segment loop of k_seg_insert must give the oracle's table bit for bit, a corrupt device
record must be refused, and a table filled with verbatim non-canonical keys must still build the reference's
graph (Graph.scala:270 probes both strands).
"""
import os
import random

import numpy as np
import pytest

from genome_amd import _lib as L
from genome_amd import dna, synth
from genome_amd.dnamap import Context, HipDNAMap
from genome_amd.graph import buildGraph
from oracle import oracle as O
from oracle import pyref as R

pytestmark = pytest.mark.gpu
# GK_MIN_LNB1=9|10 runs the whole suite over 512 / 1024 L1 buckets (the fan-out of tables beyond 34 GB) on small tables
FORCED_FANOUT = bool(os.environ.get("GK_MIN_LNB1"))


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def assert_same_table(got, want):
    for name, a, b in zip(("lo", "hi", "count"), got, want):
        assert a.shape == b.shape, f"{name}: {a.shape} vs {b.shape}"
        assert np.array_equal(a, b), name


def oracle_canonical(og):
    k = og.k
    nlo, nhi = og.nodes()
    nodes = [dna.unpack(int(a), int(b), k) for a, b in zip(nlo, nhi)]
    e = og.edges()
    edges = []
    for i in range(len(e["len"])):
        seq = synth.bases_to_str(e["bases"][e["off"][i]:e["off"][i] + e["len"][i]])
        edges.append((dna.unpack(int(e["slo"][i]), int(e["shi"][i]), k), dna.unpack(int(e["elo"][i]), int(e["ehi"][i]), k), seq))
    return nodes, edges


@pytest.mark.parametrize("k,L_", [(21, 100), (31, 150), (47, 150), (63, 150)])
@pytest.mark.parametrize("path", ["auto", "partitioned"])
def test_high_coverage_batch_sizes_table_for_distinct_keys(ctx, k, L_, path):
    """60 000 reads over a 3 kbp genome (coverage in the thousands): 5-8 million windows, a few 10^4 distinct k-mers,
    into a table created with no hint.  One call; the table must end up sized for the distinct keys (not for the
    windows), through the estimator, the exact fine level and segments that receive more keys than they have slots."""
    if FORCED_FANOUT:
        pytest.skip("sizing behaviour of 256 L1 buckets: a 3 kbp genome leaves ~30 heavy k-mers per bucket of 1024, whose sizes no region bound covers")
    n, G = 60000, 3000
    rec = synth.reads_mode_g(n, L_, G, 0.001, config_id=900 + k)
    d = ctx.alloc(rec.size + 64)
    ctx.upload(d, rec)
    ref = O.PMap(k, 1)
    occ = ref.count_reads(rec.tobytes(), n)
    m = HipDNAMap(ctx, k, 0)
    m.set_insert_path(path)
    assert m.count_reads_dev(d, n, L_) == occ
    st = m.stats()
    assert st["partitioned_launches"] >= 1, st
    assert st["size"] == ref.size()
    # sized for what it holds: the estimate may be generous, the windows are 50-100x the distinct keys
    assert st["slots"] < 8 * max(ref.size(), 1 << 19), st
    assert 0 < st["est_new_distinct_last_batch"] < occ // 4, st
    live, bad, total = m.verify()
    assert (live, bad, total) == (ref.size(), 0, occ)
    assert_same_table(m.sorted_items(), ref.export_sorted())
    # a second call on top of the content (table read back into LDS, estimate of NEW keys ~ 0): counts double
    assert m.count_reads_dev(d, n, L_) == occ
    lo, hi, cnt = ref.export_sorted()
    assert_same_table(m.sorted_items(), (lo, hi, cnt * 2))
    assert m.stats()["slots"] == st["slots"], "no new keys: the table must not have grown"
    m.close(); ctx.free(d)


@pytest.mark.parametrize("k", [31, 55])
def test_batches_bounded_by_scratch(ctx, k):
    """gk_map_set_max_batch_keys: the same call cut into several partitioned batches (table rebuilt from empty by the
    first, merged into by the others) gives the same table."""
    n, L_, G = 40000, 120, 20000
    rec = synth.reads_mode_g(n, L_, G, 0.01, config_id=77 + k)
    d = ctx.alloc(rec.size + 64)
    ctx.upload(d, rec)
    ref = O.PMap(k, 1)
    occ = ref.count_reads(rec.tobytes(), n)
    m = HipDNAMap(ctx, k, 1 << 16)
    m.set_insert_path("partitioned")
    m.set_max_batch_keys(1 << 20)
    assert m.count_reads_dev(d, n, L_) == occ
    st = m.stats()
    assert st["partitioned_launches"] >= 3, st
    assert m.verify() == (ref.size(), 0, occ)
    assert_same_table(m.sorted_items(), ref.export_sorted())
    m.close(); ctx.free(d)


@pytest.mark.parametrize("k", [21, 47])
def test_exact_fine_level_on_ragged_stream_many_ranges(ctx, k):
    """Ragged host stream (exact pipeline: P1, P2, range matrix P3, atomics-free P4) big enough for several ranges per
    L1 bucket and with repeats."""
    rnd = random.Random(5 + k)
    g = "".join(rnd.choice("AGCT") for _ in range(40000))
    reads = []
    for _ in range(30000):
        ln = rnd.randint(k, min(255, k + 100))
        s = rnd.randrange(0, len(g) - ln + 1)
        r = g[s:s + ln]
        reads.append(R.rev_comp(r) if rnd.random() < 0.5 else r)
    binb = dna.reads_to_bin(reads)
    ref = O.PMap(k, 1)
    occ = ref.count_reads(binb, len(reads))
    m = HipDNAMap(ctx, k, occ)
    m.set_insert_path("partitioned")
    assert m.count_reads(binb, len(reads)) == occ
    assert m.stats()["partitioned_launches"] >= 1
    assert m.verify() == (ref.size(), 0, occ)
    assert_same_table(m.sorted_items(), ref.export_sorted())
    m.close()


@pytest.mark.parametrize("k,declared,actual", [(31, 149, 152), (31, 100, 255), (21, 60, 61), (47, 149, 150)])
def test_oversized_length_byte_is_refused_not_followed(ctx, k, declared, actual):
    """A device record whose length byte exceeds the declared read_len (same stride or not) used to index past LDS
    arrays sized from read_len.  Now: clamped in the kernels, GK_E_FORMAT on return — on every path that streams
    device records."""
    n = 5000
    rec = synth.reads_mode_u(n, declared, 3)
    stride = rec.shape[1]
    bad = rec.copy()
    bad[n // 2, 0] = actual
    d = ctx.alloc(bad.size + 64)
    ctx.upload(d, bad)
    for path in ("direct", "partitioned"):
        m = HipDNAMap(ctx, k, n * 200)
        m.set_insert_path(path)
        with pytest.raises(L.GkError) as e:
            m.count_reads_dev(d, n, declared)
        assert e.value.code == L.GK_E_FORMAT, (path, e.value)
        # the handle stays usable: clear, then the clean stream counts as usual
        m.clear()
        ctx.upload(d, rec)
        assert m.count_reads_dev(d, n, declared) == n * max(0, declared - k + 1)
        ctx.upload(d, bad)
        m.close()
    from genome_amd.prefilter import HipPrefilter
    pf = HipPrefilter(ctx, k, n * 100)
    with pytest.raises(L.GkError) as e:
        pf.add_reads_dev(d, n, declared)
    assert e.value.code == L.GK_E_FORMAT
    pf.close()
    out = ctx.alloc(n * 40 * 32)
    with pytest.raises(L.GkError) as e:
        ctx.shard_superkmers(k, d, n, declared, 4, out, n * 40)
    assert e.value.code == L.GK_E_FORMAT
    ctx.free(out); ctx.free(d)
    assert stride == synth.record_stride(declared)


@pytest.mark.parametrize("k", [15, 31, 34, 55])
def test_verbatim_noncanonical_keys_still_build_the_reference_graph(ctx, k):
    """The ABI takes keys verbatim (gk_map_update_inc / add_counts).  Fill a table with the REVERSE COMPLEMENT of every
    canonical key (and, for a few k-mers, both orientations): the reference's `contains` probes both strands
    (Graph.scala:270), so its graph is the same as for the canonical table — and so must ours be."""
    rnd = random.Random(k)
    g = "".join(rnd.choice("AGCT") for _ in range(1500))
    reads = []
    for _ in range(500):
        ln = rnd.randint(k + 5, min(255, k + 80))
        s = rnd.randrange(0, len(g) - ln + 1)
        r = g[s:s + ln]
        r = "".join(c if rnd.random() >= 0.01 else rnd.choice([x for x in "AGCT" if x != c]) for c in r)
        reads.append(R.rev_comp(r) if rnd.random() < 0.5 else r)
    binb = dna.reads_to_bin(reads)
    ref = O.PMap(k, 1)
    ref.count_reads(binb, len(reads))
    ref.delete_lt(2)
    lo, hi, cnt = ref.export_sorted()
    want = oracle_canonical(O.Graph(ref))
    # the oracle's own graph from a reverse-complemented table (its contains() is the reference's, both strands)
    ref_rc = O.PMap(k, 1)
    keys = [dna.unpack(int(a), int(b), k) for a, b in zip(lo, hi)]
    flipped = [R.rev_comp(s) for s in keys]
    both = set(range(0, len(keys), 17))
    m = HipDNAMap(ctx, k, len(keys) * 2)
    ins = [flipped[i] for i in range(len(keys))] + [keys[i] for i in both]
    for s in ins:
        plo, phi = dna.pack(s)
        ref_rc.update_inc(plo, phi)
    m.update_inc(ins)
    assert m.stats()["noncanonical_keys"] is True
    og = O.Graph(ref_rc)
    got = buildGraph(k, m)
    assert got.canonical() == oracle_canonical(og)
    # and it is the canonical table's graph (strand symmetry of the reference's construction)
    assert got.canonical() == want
    got.removeBubbles(); og.remove_bubbles(); got.simplifyGraph(); og.simplify()
    assert got.canonical() == oracle_canonical(og)
    got.close(); m.close()
    # a canonically filled table does not pay for it
    m2 = HipDNAMap(ctx, k)
    m2.count_reads(binb, len(reads))
    assert m2.stats()["noncanonical_keys"] is False
    m2.update_inc(keys[:50])                      # verbatim but canonical keys: still clean
    assert m2.stats()["noncanonical_keys"] is False
    m2.close()


@pytest.mark.parametrize("k", [21, 47])
def test_host_stream_that_only_looks_uniform(ctx, k):
    """gk_map_count_reads takes a stream whose byte count is exactly nreads x (first record's size) for uniform WITHOUT walking
    its framing; the L1 scatter checks every length byte on the device.  Here one record in the middle is a base shorter (same
    record size): the check must notice, the chunk must be taken again through the host walk, and the table must be exact —
    from an empty map and on top of existing content."""
    n, L_ = 9000, 100
    rec = synth.reads_mode_g(n, L_, 30000, 0.01, config_id=k)
    odd = rec.copy()
    odd[n // 3, 0] = L_ - 1                       # 99 bases: still 26 bytes
    odd[n // 3, -1] &= 0x3f                       # the dropped base's bits are padding now: zero (DNASeq.scala:277)
    for stream in (rec, odd):
        ref = O.PMap(k, 1)
        occ = ref.count_reads(stream.tobytes(), n)
        m = HipDNAMap(ctx, k, occ)
        m.set_insert_path("partitioned")
        assert m.count_reads(stream.tobytes(), n) == occ
        assert_same_table(m.sorted_items(), ref.export_sorted())
        assert m.count_reads(stream.tobytes(), n) == occ          # second pass: table holds data
        lo, hi, cnt = ref.export_sorted()
        assert_same_table(m.sorted_items(), (lo, hi, cnt * 2))
        m.close()



def test_block_pool_reuse_and_trim(ctx):
    """Freed device buffers are parked in the context and handed out again (gk_ctx_trim gives them back): a map built, destroyed
    and built again over recycled blocks — with stale contents in them — must still be the oracle's table, twice, and so must
    one built right after a trim; user buffers (gk_dev_alloc / gk_dev_free) go the same way."""
    k, n, L_ = 31, 40000, 150
    rec = synth.reads_mode_g(n, L_, 300000, 0.01, config_id=77)
    ref = O.PMap(k, 1)
    occ = ref.count_reads(rec.tobytes(), n)
    want = ref.export_sorted()
    for cycle in range(3):
        d = ctx.alloc(rec.size + 64)             # >= 1 MiB: pooled
        ctx.upload(d, rec)
        m = HipDNAMap(ctx, k, occ)
        m.set_insert_path("partitioned")
        assert m.count_reads_dev(d, n, L_) == occ
        assert_same_table(m.sorted_items(), want)
        assert m.verify()[1] == 0
        m.deleteAll_lt(2)                        # compaction: another table comes out of the pool, the old one goes back
        g = buildGraph(k, m)
        assert g.counts()[0] > 0
        g.close(); m.close(); ctx.free(d)
        if cycle == 1:
            ctx.trim()


@pytest.mark.parametrize("k,mode", [(31, "U"), (31, "G"), (47, "U")])
def test_pipelined_pieces_build_the_same_table(ctx, k, mode):
    """A batch whose table geometry is final up front is cut into pieces: P4 takes the slice of every L1 region that piece j's
    scatter filled while piece j+1 is being scattered on the other stream ("p24_pieces").  Device-resident and host-fed input,
    one piece / the default / eight pieces: the same (key, count) set each time (order-independent checksum over the table),
    every key at its own slot, and — on a sample small enough for the oracle — the oracle's table."""
    n, L_ = 400000, 100
    stride = synth.record_stride(L_)
    d = ctx.alloc(n * stride + 64)
    ctx.synth_reads(d, n, L_, mode, 40 + k, 0, 200000 if mode == "G" else 0, 0.01 if mode == "G" else 0.0)
    host = ctx.download(d, n * stride)
    occ = n * (L_ - k + 1)
    sums = set()
    try:
        for pieces in (0, -1, 8):
            ctx.set_option("p24_pieces", pieces)
            for src in ("dev", "host"):
                m = HipDNAMap(ctx, k, occ)
                m.set_insert_path("partitioned")
                got = m.count_reads_dev(d, n, L_) if src == "dev" else m.count_reads(host, n)
                assert got == occ
                live, bad, total, chk = m.verify_checksum()
                assert bad == 0 and total == occ and live == m.size()
                sums.add((live, chk))
                m.close()
        assert len(sums) == 1, sums
        # the oracle on the first 30 000 reads, pieces forced although the batch is small
        ns = 30000
        ref = O.PMap(k, 1)
        assert ref.count_reads(host[:ns * stride].tobytes(), ns) == ns * (L_ - k + 1)
        ctx.set_option("p24_pieces", 8)
        m = HipDNAMap(ctx, k, ns * (L_ - k + 1))
        m.set_insert_path("partitioned")
        assert m.count_reads_dev(d, ns, L_) == ns * (L_ - k + 1)
        assert_same_table(m.sorted_items(), ref.export_sorted())
        m.close()
    finally:
        ctx.set_option("p24_pieces", -1)
        ctx.free(d)


@pytest.mark.parametrize("k", [31, 47])
def test_batch_that_outgrows_the_l1_fan_out_between_its_levels(ctx, k):
    """A table with no capacity hint is grown BETWEEN the two levels of a batch, for the distinct keys the batch turned out to
    hold, and must keep its L1 fan-out there (the batch is already cut by L1 bucket).  If that leaves more fine buckets per L1
    bucket than the fine level sorts (4096: a table that passes 34 GB in one step), the batch must fall back to the direct path
    — not fail — and the table must be the oracle's.  Staged with a small table through the `test_max_nb2` hook."""
    n, L_ = 30000, 100
    rec = synth.reads_mode_g(n, L_, 400000, 0.01, config_id=33 + k)
    d = ctx.alloc(rec.size + 64)
    ctx.upload(d, rec)
    ref = O.PMap(k, 1)
    occ = ref.count_reads(rec.tobytes(), n)
    ctx.set_option("test_max_nb2", 2)
    try:
        m = HipDNAMap(ctx, k, 0)
        m.set_insert_path("partitioned")
        assert m.count_reads_dev(d, n, L_) == occ
        st = m.stats()
        assert st["partitioned_launches"] == 0 and st["direct_launches"] >= 1, st       # the pipeline started, gave the batch back
        assert_same_table(m.sorted_items(), ref.export_sorted())
        assert m.verify()[1] == 0
        assert m.count_reads_dev(d, n, L_) == occ                                       # and the map keeps working
        lo, hi, cnt = ref.export_sorted()
        assert_same_table(m.sorted_items(), (lo, hi, cnt * 2))
        m.close()
    finally:
        ctx.set_option("test_max_nb2", 0)
        ctx.free(d)


@pytest.mark.parametrize("k", [31, 55])
def test_growth_across_a_change_of_the_l1_fan_out(ctx, k):
    """A table that grows past 34 GB changes its L1 fan-out (256 -> 512 / 1024 L1 buckets) and with it the hash bits its fine
    buckets use: that growth is a k_rehash into the new geometry, not the segment-local streaming rebuild.  Staged small: a
    table built with 256 L1 buckets, then `min_lnb1` raised so that the next growth picks 1024 — direct inserts and a
    partitioned batch on both sides of the change, against the oracle."""
    n, L_ = 24000, 100
    rec = synth.reads_mode_g(n, L_, 500000, 0.01, config_id=91 + k)
    d = ctx.alloc(rec.size + 64)
    ctx.upload(d, rec)
    stride = rec.shape[1]
    ref = O.PMap(k, 1)
    m = HipDNAMap(ctx, k, 360000)                     # ~270 segments: 256 L1 buckets, and too small for two thirds of the keys
    try:
        third = n // 3
        m.set_insert_path("partitioned")
        assert m.count_reads_dev(d, third, L_) == ref.count_reads(rec[:third].tobytes(), third)
        slots0 = m.stats()["slots"]
        ctx.set_option("min_lnb1", 10)
        m.set_insert_path("direct")                   # grows by map_reserve -> k_rehash into a 1024-bucket table
        assert m.count_reads_dev(d + third * stride, third, L_) == ref.count_reads(rec[third:2 * third].tobytes(), third)
        assert m.stats()["slots"] > slots0
        assert_same_table(m.sorted_items(), ref.export_sorted())
        assert m.verify()[1] == 0
        m.set_insert_path("partitioned")              # the pipeline over the new geometry
        rest = n - 2 * third
        assert m.count_reads_dev(d + 2 * third * stride, rest, L_) == ref.count_reads(rec[2 * third:].tobytes(), rest)
        assert m.stats()["partitioned_launches"] >= 1
        assert_same_table(m.sorted_items(), ref.export_sorted())
        assert m.verify()[1] == 0
        m.deleteAll_lt(2); ref.delete_lt(2)           # and the filter + rebuild on the 1024-bucket table
        assert_same_table(m.sorted_items(), ref.export_sorted())
    finally:
        ctx.set_option("min_lnb1", 0)
        m.close(); ctx.free(d)


@pytest.mark.parametrize("k", [21, 47])
@pytest.mark.parametrize("shape", ["uniform", "odd_record", "ragged"])
@pytest.mark.parametrize("path", ["auto", "partitioned", "direct"])
def test_host_stream_in_many_chunks_with_the_next_chunk_uploaded_ahead(ctx, k, shape, path):
    """A long host stream is cut into chunks of one staging area each; while the pipeline works on one chunk the copy stream
    uploads the NEXT one into the second staging area, and that chunk is then scattered in one launch behind the upload's event
    (gk_table.hip, two staging areas).  Here the staging size is shrunk (test_max_stage) so that a 26 000-read stream takes ~10
    chunks: uniform records, a stream that only LOOKS uniform (one record a base shorter: the device check sends that chunk back
    through the host walk while the chunk after it is already on its way), and a ragged stream (offset tables).  The table must
    be the oracle's, with the look-ahead switched off too, and pinned as well as pageable host memory must work."""
    n, L_ = 26000, 100
    rec = synth.reads_mode_g(n, L_, 40000, 0.01, config_id=100 + k)
    if shape == "odd_record":
        rec = rec.copy()
        rec[n // 2 + 7, 0] = L_ - 1
        rec[n // 2 + 7, -1] &= 0x3f
        stream = rec.tobytes()
    elif shape == "ragged":
        rnd = random.Random(k)
        reads = [dna.unpack_2bit(rec[i, 1:], L_)[:rnd.choice([L_, L_ - 3, k, k - 1, 60])] for i in range(n)]
        stream = dna.reads_to_bin(reads)
    else:
        stream = rec.tobytes()
    ref = O.PMap(k, 1)
    occ = ref.count_reads(stream, n)
    want = ref.export_sorted()
    ctx.set_option("test_max_stage", 64 << 10)
    try:
        for ahead in (1, 0):
            ctx.set_option("host_prefetch", ahead)
            m = HipDNAMap(ctx, k, occ)
            m.set_insert_path(path)
            assert m.count_reads(stream, n) == occ
            assert_same_table(m.sorted_items(), want)
            m.close()
        # pinned memory (asynchronous copies for real), and the caller's own prefetch of the stream's head
        hb = ctx.host_alloc(len(stream))
        hb[:] = np.frombuffer(stream, np.uint8)
        m = HipDNAMap(ctx, k, occ)
        m.set_insert_path(path)
        m.prefetch_reads(hb, n)
        assert m.count_reads(hb, n) == occ
        assert_same_table(m.sorted_items(), want)
        m.prefetch_reads(hb, n)                           # a prefetch nobody consumes: the next count is of ANOTHER buffer
        m.clear()
        assert m.count_reads(stream, n) == occ
        assert_same_table(m.sorted_items(), want)
        m.prefetch_reads(hb, n)                           # ... and one that is still pending when the map goes away
        m.close()
        ctx.host_free(hb)
    finally:
        ctx.set_option("test_max_stage", 0)
        ctx.set_option("host_prefetch", -1)
