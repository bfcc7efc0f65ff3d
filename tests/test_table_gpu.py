"""GPU parity tests of the DNAMap / FreqFilter hot path: the HIP library, called through the C-ABI,
against the CPU oracle on the same inputs — bit-exact on the canonical (sorted) table
serialisation — plus size-independent properties at BASELINE.json's full C2 size.
"""
import json
import os
import random

import numpy as np
import pytest

from genome_amd import _lib as L
from genome_amd import dna, synth
from genome_amd.dnamap import Context, HipDNAMap
from genome_amd.partitioned import PartitionedDNAMap, owner_of
from oracle import oracle as O
from oracle import pyref as R

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def assert_same_table(got, want):
    for name, a, b in zip(("lo", "hi", "count"), got, want):
        assert a.shape == b.shape, f"{name}: {a.shape} vs {b.shape}"
        assert np.array_equal(a, b), name


def test_device_is_gfx950(ctx):
    assert L.device_count() >= 1


@pytest.mark.parametrize("name", sorted(f[:-5] for f in os.listdir(GOLDEN) if f.endswith(".json") and f != "hash_ties.json"))
def test_golden_tables(ctx, name):
    fx = json.load(open(os.path.join(GOLDEN, name + ".json")))
    k, P, rounds = fx["k"], fx["P"], fx["rounds"]
    binb = bytes.fromhex(fx["bin_hex"])
    for m in (HipDNAMap(ctx, k), PartitionedDNAMap(ctx, k, P) if P > 1 else HipDNAMap(ctx, k, 100000)):
        assert m.count_reads(binb, fx["nreads"]) == fx["occurrences"]
        lo, hi, cnt = m.sorted_items()
        assert [[int(a), int(b), int(c)] for a, b, c in zip(lo, hi, cnt)] == fx["table"]
        m.deleteAll_lt(rounds)
        lo, hi, cnt = m.sorted_items()
        assert [[int(a), int(b), int(c)] for a, b, c in zip(lo, hi, cnt)] == fx["table_filtered"]
        assert m.size() == len(fx["table_filtered"])
        m.close()


def _random_reads(rnd, n, lmin, lmax, genome_len, err):
    g = "".join(rnd.choice("AGCT") for _ in range(genome_len))
    reads = []
    for _ in range(n):
        ln = rnd.randint(lmin, lmax)
        st = rnd.randrange(0, genome_len - ln + 1)
        r = g[st:st + ln]
        if rnd.random() < 0.5:
            r = R.rev_comp(r)
        reads.append("".join(c if rnd.random() >= err else rnd.choice([x for x in "AGCT" if x != c]) for c in r))
    return reads


@pytest.mark.parametrize("k", [2, 5, 11, 21, 30, 31, 34, 35, 47, 55, 62, 63, 64])
def test_ragged_reads_vs_oracle(ctx, k):
    """Ragged / empty / shorter-than-k records (FreqFilter.scala:29), every supported key width."""
    rnd = random.Random(k)
    reads = _random_reads(rnd, 300, max(1, k - 5), min(255, k + 120), 1200, 0.02)
    reads += ["", "A", "AG", "AGCT"[:max(1, min(4, k - 1))]] + ["".join(rnd.choice("AGCT") for _ in range(255))] * 2
    rnd.shuffle(reads)
    binb = dna.reads_to_bin(reads)
    m = HipDNAMap(ctx, k, 64)            # tiny hint: forces the rehash/grow path
    ref = O.PMap(k, 1)
    occ = ref.count_reads(binb, len(reads))
    assert m.count_reads(binb, len(reads)) == occ
    assert_same_table(m.sorted_items(), ref.export_sorted())
    assert m.size() == ref.size()
    if ref.size() > 4096:
        assert m.stats()["grows"] >= 1
    # a second pass over the same reads doubles every count (update(y, 1, _+1) is additive)
    m.count_reads(binb, len(reads))
    lo, hi, cnt = ref.export_sorted()
    assert_same_table(m.sorted_items(), (lo, hi, cnt * 2))
    for rounds in (3, 5):
        m.deleteAll_lt(rounds)
        ref2 = O.PMap(k, 1)
        ref2.count_reads(binb, len(reads)); ref2.count_reads(binb, len(reads))
        ref2.delete_lt(rounds)
        assert_same_table(m.sorted_items(), ref2.export_sorted())
    m.close()


@pytest.mark.parametrize("k", [21, 47])
def test_host_stream_with_long_uniform_runs_and_ragged_stretches(ctx, k):
    """gk_map_count_reads walks the framing on the host: runs of >= 4096 equal-length records become fixed-stride
    chunks (no offset table), anything else goes through the ragged form.  Mix both, in both orders, plus records
    shorter than k and a short uniform tail."""
    rnd = random.Random(1000 + k)
    g = "".join(rnd.choice("AGCT") for _ in range(5000))

    def reads(n, ln):
        return [g[s:s + ln] for s in (rnd.randrange(0, len(g) - ln + 1) for _ in range(n))]
    stream = reads(5000, 100) + reads(7, 63) + ["", "ACG"] + reads(4500, 120) + reads(300, k + 3) + reads(4096, 90) + reads(5, k - 1 if k > 2 else 1)
    binb = dna.reads_to_bin(stream)
    ref = O.PMap(k, 1)
    occ = ref.count_reads(binb, len(stream))
    for path in ("auto", "partitioned", "direct"):
        m = HipDNAMap(ctx, k, 1 << 16)
        m.set_insert_path(path)
        assert m.count_reads(binb, len(stream)) == occ
        assert_same_table(m.sorted_items(), ref.export_sorted())
        m.close()
    # the singleton pre-filter walks host streams the same way
    from genome_amd.prefilter import HipPrefilter
    pf = HipPrefilter(ctx, k, ref.size())
    pf.add_reads(binb, len(stream))
    m = HipDNAMap(ctx, k, 1 << 16)
    assert pf.count_reads(m, binb, len(stream))[0] == occ
    m.deleteAll_lt(2)
    ref.delete_lt(2)
    assert_same_table(m.sorted_items(), ref.export_sorted())
    m.close(); pf.close()
    # truncation inside a uniform run is still reported
    m = HipDNAMap(ctx, k)
    with pytest.raises(L.GkError) as e:
        m.count_reads(binb[:5000 * 26 - 3], 5000)
    assert e.value.code == L.GK_E_FORMAT
    m.close()


def test_truncated_stream_is_a_format_error(ctx):
    m = HipDNAMap(ctx, 11)
    binb = dna.reads_to_bin(["AGCTAGCTAGCTAGCT"] * 3)
    with pytest.raises(L.GkError) as e:
        m.count_reads(binb[:-2], 3)
    assert e.value.code == L.GK_E_FORMAT
    with pytest.raises(L.GkError):
        m.count_reads(binb, 4)
    assert m.size() <= 6
    m.close()


def test_key_length_and_k_errors(ctx):
    """`assert(key.length == k)` (ArrayDNAMap.scala:182,199) -> GK_E_KLEN; k=32/33/>64 unsupported."""
    m = HipDNAMap(ctx, 11)
    with pytest.raises(AssertionError):
        m.apply("AGCT")
    with pytest.raises(L.KeyLengthError):
        m.update_inc([(1 << 22, 0)])          # bit above 2k
    with pytest.raises(L.KeyLengthError):
        m.apply_batch([(1, 1)])
    m.close()
    for k in (0, 1, 32, 33, 65, 100):
        with pytest.raises(L.GkError) as e:
            HipDNAMap(ctx, k)
        assert e.value.code == L.GK_E_UNSUPPORTED_K


@pytest.mark.parametrize("k", [21, 55, 64])
def test_apply_contains_update(ctx, k):
    rnd = random.Random(k)
    keys = ["".join(rnd.choice("AGCT") for _ in range(k)) for _ in range(500)]
    m = HipDNAMap(ctx, k)
    assert m.size() == 0 and m.apply(keys[0]) is None and not m.contains(keys[0])
    m.update_inc(keys[:300])
    m.update_inc(keys[:100])
    want = {}
    for s in keys[:300] + keys[:100]:
        want[s] = want.get(s, 0) + 1
    assert m.size() == len(want)
    got = m.apply_batch(keys)
    for s, v in zip(keys, got):
        assert int(v) == want.get(s, -1)
    # verbatim keys: no canonicalisation happens in update (that is FreqFilter's job)
    rc = R.rev_comp(keys[0])
    if rc not in want:
        assert not m.contains(rc)
    # mapReduce / foreach / getAll of the trait (ArrayDNAMap.scala:52,58-59), closures on the host
    assert m.mapReduce(lambda kv: kv[1] if kv[1] > 1 else None, sum) == sum(v for v in want.values() if v > 1)
    seen = {}
    m.foreach(lambda kv: seen.__setitem__(kv[0], kv[1]))
    assert seen == want
    assert m.getAll(keys[0]) == [want[keys[0]]] and m.getAll(keys[-1]) == []
    m.clear()
    assert m.size() == 0 and m.apply(keys[0]) is None
    m.close()


@pytest.mark.parametrize("k,L_,mode", [(21, 100, "G"), (31, 150, "U"), (31, 150, "G"), (55, 150, "G"), (63, 150, "U"), (64, 150, "G")])
def test_device_resident_reads_and_device_synth(ctx, k, L_, mode):
    """count_reads_dev on records generated on the device == oracle on the numpy generator's bytes."""
    n, G, e, cid = 3000, 20000, 0.01, 7
    rec = synth.reads_mode_u(n, L_, cid) if mode == "U" else synth.reads_mode_g(n, L_, G, e, cid)
    d = ctx.alloc(rec.size + 64)
    ctx.synth_reads(d, n, L_, mode, cid, 0, G, e)
    assert np.array_equal(ctx.download(d, rec.size), rec.reshape(-1)), "device generator != numpy generator"
    # chunked generation with first_read offsets gives the same bytes
    half = n // 2
    ctx.synth_reads(d + half * rec.shape[1], n - half, L_, mode, cid, half, G, e)
    assert np.array_equal(ctx.download(d, rec.size), rec.reshape(-1))
    m = HipDNAMap(ctx, k, n * (L_ - k + 1))
    occ = m.count_reads_dev(d, n, L_)
    ref = O.PMap(k, 1)
    assert occ == ref.count_reads(rec.tobytes(), n) == n * (L_ - k + 1)
    assert_same_table(m.sorted_items(), ref.export_sorted())
    ms, kocc = m.last_count_kernel()
    assert kocc == occ and ms > 0
    m.close()
    ctx.free(d)


@pytest.mark.parametrize("k,P", [(21, 2), (31, 4), (31, 8), (55, 8), (63, 3), (64, 4)])
def test_logical_partitions_vs_oracle(ctx, k, P):
    """PartitionedDNAMap: sorted content is independent of P and of the partition function."""
    n, L_ = 2000, 120
    rec = synth.reads_mode_g(n, L_, 9000, 0.01, config_id=k + P)
    d = ctx.alloc(rec.size + 64)
    ctx.upload(d, rec)
    pm = PartitionedDNAMap(ctx, k, P)
    occ = pm.count_reads_dev(d, n, L_)
    ref = O.PMap(k, P)
    assert occ == ref.count_reads(rec.tobytes(), n)
    assert_same_table(pm.sorted_items(), ref.export_sorted())
    assert pm.size() == ref.size()
    # every key sits in the partition the owner function names, and so would its reverse complement
    sizes = []
    for p, part in enumerate(pm.parts):
        lo, hi, _ = part.items()
        sizes.append(len(lo))
        for a, b in list(zip(lo, hi))[:200]:
            assert owner_of(k, int(a), int(b), P) == p
            s = dna.unpack(int(a), int(b), k)
            rlo, rhi = dna.pack(R.rev_comp(s))
            assert owner_of(k, rlo, rhi, P) == p
    assert min(sizes) > 0
    pm.deleteAll_lt(3); ref.delete_lt(3)
    assert_same_table(pm.sorted_items(), ref.export_sorted())
    merged = pm.merged()
    assert_same_table(merged.sorted_items(), ref.export_sorted())
    keys = (ref.export_sorted()[0][:50], ref.export_sorted()[1][:50])
    assert np.array_equal(pm.apply_batch(keys), ref.export_sorted()[2][:50])
    merged.close(); pm.close(); ctx.free(d)


def test_full_size_c2_properties(ctx):
    """BASELINE.json configs[1]: 1M x 150 bp, k=31, one GPU.  The oracle cannot finish this in
    seconds, so check size-independent properties: window count, sum of counts == occurrences,
    additivity of a second pass, filter monotonicity, and a sampled exact comparison."""
    n, L_, k = 1_000_000, 150, 31
    stride = synth.record_stride(L_)
    d = ctx.alloc(n * stride + 64)
    ctx.synth_reads(d, n, L_, "G", 2, 0, 5_000_000, 0.01)
    m = HipDNAMap(ctx, k, n * (L_ - k + 1))
    occ = m.count_reads_dev(d, n, L_)
    assert occ == n * (L_ - k + 1)
    lo, hi, cnt = m.items()
    assert len(lo) == m.size() and int(cnt.astype(np.int64).sum()) == occ
    assert len(np.unique(lo)) == len(lo)                      # a key never lands in two slots
    # sampled exactness: the first 300 reads through the oracle, every key's count must be >= and the
    # key set must be contained
    head = ctx.download(d, 300 * stride)
    ref = O.PMap(k, 1)
    ref.count_reads(head.tobytes(), 300)
    rlo, rhi, rcnt = ref.export_sorted()
    got = m.apply_batch((rlo, rhi))
    assert (got >= rcnt).all()
    size1 = m.size()
    m.count_reads_dev(d, n, L_)
    assert m.size() == size1
    lo2, _, cnt2 = m.items()
    o1, o2 = np.argsort(lo), np.argsort(lo2)
    assert np.array_equal(lo[o1], lo2[o2]) and np.array_equal(cnt[o1] * 2, cnt2[o2])
    ge3 = int((cnt2 >= 3).sum())
    m.deleteAll_lt(3)
    assert m.size() == ge3
    m.close(); ctx.free(d)


@pytest.mark.parametrize("k,L_,n,hint", [(31, 150, 40000, 0), (31, 150, 40000, 6_000_000), (21, 100, 30000, 100), (55, 150, 30000, 0),
                                          (63, 120, 20000, 3_000_000), (11, 60, 50000, 0), (64, 150, 20000, 0)])
def test_partitioned_path_equals_direct_and_oracle(ctx, k, L_, n, hint):
    """The LDS segment-build path (gk_partition.hip) and the global-atomic path must produce the
    same table as the oracle: from empty, on top of existing content, after a deferred clear, and
    for routed key arrays (owner side of the all-to-all)."""
    rec = synth.reads_mode_g(n, L_, 60000, 0.01, config_id=k)
    d = ctx.alloc(rec.size + 64)
    ctx.upload(d, rec)
    ref = O.PMap(k, 1)
    occ = ref.count_reads(rec.tobytes(), n)
    want1 = ref.export_sorted()
    tables = {}
    for path in ("direct", "partitioned"):
        m = HipDNAMap(ctx, k, hint)
        m.set_insert_path(path)
        assert m.count_reads_dev(d, n, L_) == occ
        assert_same_table(m.sorted_items(), want1)
        assert m.size() == ref.size()
        m.count_reads_dev(d, n // 2, L_)                       # second batch on top of a non-empty table
        tables[path] = m
    ref.count_reads(rec[:n // 2].tobytes(), n // 2)
    want2 = ref.export_sorted()
    for path, m in tables.items():
        assert_same_table(m.sorted_items(), want2)
        st = m.stats()
        assert (st["partitioned_launches"] > 0) == (path == "partitioned"), st
        m.clear()                                              # deferred clear, then rebuild from empty
        assert m.size() == 0
        assert m.count_reads(rec.tobytes(), n) == occ          # host stream entry point
        assert_same_table(m.sorted_items(), want1)
        m.clear()
        assert m.apply_batch((want1[0][:5], want1[1][:5])).tolist() == [-1] * 5   # a lookup materialises the clear
    # routed keys: shard into 1 partition = plain canonical key stream, insert through both paths
    W = 1 if k <= 32 else 2
    nk = n * (L_ - k + 1)
    dk = ctx.alloc(nk * 8 * W)
    counts = ctx.shard_reads(k, d, n, L_, 1, dk, nk)
    assert int(counts[0]) == occ
    for path, m in tables.items():
        m.clear()
        m.update_inc_dev(dk, nk)
        assert_same_table(m.sorted_items(), want1)
        m.deleteAll_lt(2)
        m.close()
    ctx.free(dk); ctx.free(d)


def test_partitioned_path_heavy_hitters_and_tiny_batches(ctx):
    """Skew: one k-mer repeated 300k times lands in ONE segment bucket; empty and 1-read batches."""
    k = 21
    reads = ["A" * 150] * 3000 + ["AG" * 75] * 2000
    rnd = random.Random(4)
    reads += ["".join(rnd.choice("AGCT") for _ in range(150)) for _ in range(3000)]
    binb = dna.reads_to_bin(reads)
    ref = O.PMap(k, 1)
    ref.count_reads(binb, len(reads))
    m = HipDNAMap(ctx, k)
    m.set_insert_path("partitioned")
    assert m.count_reads(binb, len(reads)) == len(reads) * 130
    assert_same_table(m.sorted_items(), ref.export_sorted())
    assert m.count_reads(b"", 0) == 0
    one = dna.reads_to_bin(["AGCT" * 10])
    m.count_reads(one, 1); ref.count_reads(one, 1)
    assert_same_table(m.sorted_items(), ref.export_sorted())
    m.close()


@pytest.mark.parametrize("k,n,mode", [(55, 1100000, "G")])
def test_interleaved_key_buffer(ctx, k, n, mode):
    """Batches whose level-1 key buffer reaches 1.5 GB interleave the 256 regions in chunk-sized blocks (l1_slot in
    gk_partition.hip) so that P2 does not thrash the TLB.  Too big for the CPU oracle in test time: the partitioned
    result must equal the direct kernels' (which the oracle tests pin) and the window count must be exact."""
    L_ = 150
    d = ctx.alloc(n * synth.record_stride(L_) + 64)
    ctx.synth_reads(d, n, L_, mode, 31, 0, 3_000_000, 0.0005)    # ~1e7 distinct k-mers: the export stays small
    tables = []
    for path in ("direct", "partitioned"):
        m = HipDNAMap(ctx, k, n * (L_ - k + 1))
        m.set_insert_path(path)
        assert m.count_reads_dev(d, n, L_) == n * (L_ - k + 1)
        assert (m.stats()["partitioned_launches"] > 0) == (path == "partitioned")
        tables.append(m.sorted_items())
        m.close()
    assert_same_table(tables[1], tables[0])
    assert int(tables[0][2].astype(np.int64).sum()) == n * (L_ - k + 1)
    ctx.free(d)


@pytest.mark.parametrize("k", [31, 47])
def test_failed_segments_are_replayed(ctx, k):
    """P5 hands a segment that fills up back to the host (grow + replay through the direct path).  Hashed keys do not
    fill a segment of a properly reserved table, so the test hook "test_no_reserve" leaves the table too small: both
    calls push segments past their 2048 slots."""
    ctx.set_option("test_no_reserve", 1)
    try:
        n, L_ = 40000, 150
        rec = synth.reads_mode_u(2 * n, L_, 77)
        d = ctx.alloc(rec.size + 64)
        ctx.upload(d, rec)
        distinct = n * (L_ - k + 1)
        ref = O.PMap(k, 1)
        m = HipDNAMap(ctx, k, int(distinct * 0.5))           # ~0.77 x distinct slots: the batch cannot fit
        m.set_insert_path("partitioned")
        assert m.count_reads_dev(d, n, L_) == distinct
        ref.count_reads(rec[:n].tobytes(), n)
        st = m.stats()
        assert st["failed_segments"] > 0 and st["partitioned_launches"] >= 1, st
        assert_same_table(m.sorted_items(), ref.export_sorted())
        first = st["failed_segments"]
        assert m.count_reads_dev(d + n * rec.shape[1], n, L_) == distinct
        ref.count_reads(rec[n:].tobytes(), n)
        assert m.stats()["failed_segments"] > first
        assert_same_table(m.sorted_items(), ref.export_sorted())
        m.close(); ctx.free(d)
    finally:
        ctx.set_option("test_no_reserve", 0)


@pytest.mark.parametrize("k", [21, 55])
def test_partitioned_path_gives_the_batch_back_under_extreme_skew(ctx, k):
    """Device-resident fixed-length reads that are all the same sequence: every window falls into ONE over-provisioned
    region, the spill list overflows too, and the batch has to take the direct path — from an empty table (the
    overflow is only noticed after P5 has run: the table is cleared again) and on top of existing content (noticed
    before P5: nothing but scratch was touched).  Counts must come out exact both times."""
    n, L_ = 30000, 150
    rec = np.tile(dna.reads_to_bin_array(["AC" * 75], L_), (n, 1)) if hasattr(dna, "reads_to_bin_array") else None
    if rec is None:
        one = np.frombuffer(dna.reads_to_bin(["AC" * 75]), np.uint8)
        rec = np.tile(one, (n, 1))
    d = ctx.alloc(rec.size + 64)
    ctx.upload(d, rec)
    ref = O.PMap(k, 1)
    ref.count_reads(rec.tobytes(), n)
    m = HipDNAMap(ctx, k, 1 << 20)
    m.set_insert_path("partitioned")
    assert m.count_reads_dev(d, n, L_) == n * (L_ - k + 1)
    st = m.stats()
    assert st["retries_direct"] == 1, st
    assert_same_table(m.sorted_items(), ref.export_sorted())
    # second batch on the now non-empty table: same skew, same answer, counts doubled
    assert m.count_reads_dev(d, n, L_) == n * (L_ - k + 1)
    assert m.stats()["retries_direct"] == 2
    lo, hi, cnt = ref.export_sorted()
    assert_same_table(m.sorted_items(), (lo, hi, cnt * 2))
    # and after a clear the deferred-clear rebuild path is hit again
    m.clear()
    assert m.size() == 0
    assert m.count_reads_dev(d, n, L_) == n * (L_ - k + 1)
    assert_same_table(m.sorted_items(), ref.export_sorted())
    m.close(); ctx.free(d)


@pytest.mark.parametrize("k,L_,P", [(31, 150, 8), (21, 100, 3), (11, 60, 2), (55, 150, 8), (63, 200, 4), (34, 255, 5), (64, 150, 8),
                                    (7, 255, 16)])     # k < m: a run per window, the descriptor list overflows and the tile is regrouped
def test_superkmer_records(ctx, k, L_, P):
    """gk_shard_superkmers_dev: every record is a run of same-owner windows of one read, in the `.bin`
    framing; together the records hold every window exactly once (multiset of canonical k-mers ==
    the oracle's table) and every window sits with the owner gk_owner_of names."""
    from genome_amd.dnamap import skm_slot_bytes
    n = 1500
    rec = synth.reads_mode_g(n, L_, 20000, 0.02, config_id=k * 7 + P)
    d = ctx.alloc(rec.size + 64)
    ctx.upload(d, rec)
    slot = skm_slot_bytes(k)
    assert slot == (16 if k <= 31 else 32)
    with pytest.raises(L.GkError) as e:                       # too small: counts are still reported
        ctx.shard_superkmers(k, d, n, L_, P, ctx.alloc(64), 2)
    assert e.value.code == L.GK_E_CAPACITY
    cap = n * (L_ - k + 1) * P
    region = cap // P
    d_out = ctx.alloc(cap * slot)
    recs, kmers = ctx.shard_superkmers(k, d, n, L_, P, d_out, cap)
    assert int(kmers.sum()) == n * (L_ - k + 1)
    if k >= 21:
        assert int(recs.sum()) < int(kmers.sum()) / 3         # runs, not single windows (k == m has no shared minimizers)
    ref = O.PMap(k, 1)
    ref.count_reads(rec.tobytes(), n)
    got = O.PMap(k, 1)
    max_bases = (slot - 1) * 4
    packed = []
    for p in range(P):
        part = ctx.download(d_out + p * region * slot, int(recs[p]) * slot).reshape(-1, slot)   # owner p's region
        packed.append(part)
        lens = part[:, 0].astype(int)
        assert (lens >= k).all() and (lens <= max_bases).all()
        assert int((lens - k + 1).sum()) == int(kmers[p])
        # padding after the last base is zero; the record parses as a `.bin` record of its own
        for row in part[:: max(1, len(part) // 60)]:
            ln = int(row[0]); nb = (ln + 3) // 4
            assert not row[1 + nb:].any()
            if ln % 4:
                assert int(row[nb]) >> (2 * (ln % 4)) == 0
            s = R.reads_from_bin(bytes(row[:1 + nb]), 1)[0]
            for i in range(0, ln - k + 1, 7):
                lo, hi = dna.pack(s[i:i + k])
                assert owner_of(k, lo, hi, P) == p
        # the `.bin` framing lets the oracle count the records directly (strip the slot padding)
        stream = b"".join(bytes(row[:1 + (int(row[0]) + 3) // 4]) for row in part)
        got.count_reads(stream, len(part))
    assert_same_table(got.export_sorted(), ref.export_sorted())
    # and the GPU owner-side count of the records gives the same table through both insert paths
    allrec = np.concatenate(packed)
    d_all = ctx.alloc(allrec.size + 64)
    ctx.upload(d_all, allrec)
    for path in ("direct", "partitioned"):
        m = HipDNAMap(ctx, k)
        m.set_insert_path(path)
        assert m.count_superkmers_dev(d_all, int(recs.sum()), int(kmers.sum())) == int(kmers.sum())
        assert_same_table(m.sorted_items(), ref.export_sorted())
        with pytest.raises(L.GkError):
            m.count_superkmers_dev(d_all, int(recs.sum()), int(kmers.sum()) - 1)   # announced count must match
        m.close()
    ctx.free(d_out); ctx.free(d); ctx.free(d_all)


def test_stream_bench_reports_plausible_rates(ctx):
    """gk_dev_stream_bench (the measured yardstick bench.py quotes next to the nominal HBM peak): three positive rates in a
    plausible band for an HBM part, and the argument checks of the boundary."""
    r = ctx.stream_bench(256 << 20, 3)
    for name in ("copy_GBps", "fill_GBps", "sum_GBps"):
        assert 200.0 < r[name] < 20000.0, r
    with pytest.raises(L.GkError):
        ctx.stream_bench(1024, 3)
    with pytest.raises(L.GkError):
        ctx.stream_bench(1 << 20, 0)
