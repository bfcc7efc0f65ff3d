"""BASELINE.json's configurations on one MI355X (-m gpu).

C1 (10k x 100 bp, k=21) at its stated size: table and graph bit-exact against the oracle.
C2 (1M x 150 bp, k=31) at full size, mode U (the bench's insert-stress reading) and mode G: size-independent
   properties + sampled exactness, through the partitioned pipeline the bench runs.
C4 (500M x 150 bp, k=55, 8 GPUs) and C5 (2B x 150 bp, k=63, 8 GPUs): ONE RANK'S SHARE — an eighth of the reads
   over an eighth of the genome (so coverage and the table a rank ends up with are those of the real run) —
   routed to P = 8 logical partitions on one device, through the exact singleton pre-filter so that the tables
   fit.  The oracle cannot follow at this size (parity at oracle-size inputs: test_table_gpu.py,
   test_prefilter_gpu.py); checked here: the window count, owner placement of every stored key, no key stored twice,
   sum of counts == windows admitted, and — where the unfiltered table fits — survivors of deleteAll(v < 3)
   identical with and without the pre-filter (live, sum, checksum over all partitions).
The 8-GPU exchange itself (RCCL) cannot run on this box: `configs_untested` keeps "C4/C5 at 8 ranks".
"""
import numpy as np
import pytest

from genome_amd import dna, synth
from genome_amd.dnamap import Context, HipDNAMap
from genome_amd.graph import buildGraph
from genome_amd.partitioned import PartitionedDNAMap
from oracle import oracle as O

pytestmark = pytest.mark.gpu


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


def test_c1_full_size_table_and_graph_exact(ctx):
    """configs[0]: 10k x 100 bp, k = 21, 20 kbp genome (50x), 1 % error — SURVEY.md §8d C1, at its stated size."""
    n, L_, k, G, e = 10_000, 100, 21, 20_000, 0.01
    rec = synth.reads_mode_g(n, L_, G, e, config_id=1)
    ref = O.PMap(k, 1)
    occ = ref.count_reads(rec.tobytes(), n)
    assert occ == n * (L_ - k + 1) == 800_000
    want = ref.export_sorted()
    d = ctx.alloc(rec.size + 64)
    ctx.upload(d, rec)
    maps = []
    for path, entry in (("auto", "host"), ("direct", "dev"), ("partitioned", "dev")):
        m = HipDNAMap(ctx, k)
        m.set_insert_path(path)
        got = m.count_reads(rec.tobytes(), n) if entry == "host" else m.count_reads_dev(d, n, L_)
        assert got == occ
        assert_same_table(m.sorted_items(), want)
        assert m.verify() == (ref.size(), 0, occ)
        maps.append(m)
    ref.delete_lt(3)                                   # GraphBuilder.scala:30 rounds = 3
    for m in maps:
        m.deleteAll_lt(3)
        assert_same_table(m.sorted_items(), ref.export_sorted())
    og = O.Graph(ref)
    g = buildGraph(k, maps[0])
    assert g.canonical() == oracle_canonical(og)
    assert g.counts() == (og.num_nodes(), og.num_edges(), og.total_edge_len())
    # GraphBuilder.scala:37-54: components, the two histograms, retain(max component)
    nodes_pc, len_pc = g.componentStats()
    assert len(nodes_pc) == og.num_components()
    assert int(nodes_pc.sum()) == og.num_nodes() and int(len_pc.sum()) == og.total_edge_len()
    h1, h2 = g.componentHistograms()
    assert sum(c for _, c in h1) == sum(c for _, c in h2) == og.num_components()
    kept, comps = g.retainLargest()
    assert comps == og.num_components() and kept == og.retain_largest() == int(nodes_pc.max())
    assert g.canonical() == oracle_canonical(og)
    g.removeBubbles(); og.remove_bubbles(); g.simplifyGraph(); og.simplify()
    assert g.canonical() == oracle_canonical(og)
    g.close()
    for m in maps:
        m.close()
    ctx.free(d)


@pytest.mark.parametrize("mode", ["U", "G"])
def test_c2_full_size_properties(ctx, mode):
    """configs[1]: 1M x 150 bp, k = 31 — the bench's workload, both synthetic modes, the path the bench takes."""
    n, L_, k = 1_000_000, 150, 31
    stride = synth.record_stride(L_)
    d = ctx.alloc(n * stride + 64)
    ctx.synth_reads(d, n, L_, mode, 2, 0, 5_000_000, 0.01)
    occ = n * (L_ - k + 1)
    m = HipDNAMap(ctx, k, int(occ * 1.05))             # bench.py's hint
    assert m.count_reads_dev(d, n, L_) == occ
    st = m.stats()
    assert st["partitioned_launches"] >= 1 and st["direct_launches"] == 0, st
    live, bad, total, chk = m.verify_checksum()
    assert live == m.size() and bad == 0 and total == occ
    if mode == "U":
        assert live > occ - 100                        # uniform 31-mers: all distinct but for birthday collisions
    # sampled exactness: the first 300 reads through the oracle
    head = ctx.download(d, 300 * stride)
    ref = O.PMap(k, 1)
    ref.count_reads(head.tobytes(), 300)
    rlo, rhi, rcnt = ref.export_sorted()
    got = m.apply_batch((rlo, rhi))
    assert (got >= rcnt).all()
    if mode == "U":
        assert (got == rcnt).sum() > len(rcnt) - 5      # distinct reads: the sample's counts are the table's
    # the direct path builds the same table (live, sum, checksum)
    m2 = HipDNAMap(ctx, k, int(occ * 1.05))
    m2.set_insert_path("direct")
    assert m2.count_reads_dev(d, n, L_) == occ
    assert m2.verify_checksum() == (live, 0, total, chk)
    m2.close()
    # additivity of a second pass, filter monotonicity
    m.count_reads_dev(d, n, L_)
    assert m.size() == live and m.verify()[2] == 2 * occ
    m.deleteAll_lt(3)
    assert m.verify()[0] == m.size() <= live
    m.close(); ctx.free(d)


def _graph_stage(ctx, pm, k, budget, tag):
    """The graph stage of an 8-GPU configuration as ONE rank sees it, at an eighth of the size: the 8 logical partitions stand
    for the 8 ranks' tables (partition 0 = "this rank's", the other seven live on other GPUs in the real run), the merged table
    for the replica every rank gathers (gk_dist_gather_map; here gk_map_add_map, partition by partition — the same export /
    insert chunks without the exchange), then Graph.buildGraph in both unitig constructions, removeBubbles, simplifyGraph and
    GraphBuilder's retain (Graph.scala:269-382, 125-149, 211-230; GraphBuilder.scala:32-54).  `budget` = an eighth of a
    288 GB GPU: the library plans against it (gk_ctx_set_mem_budget) and the high-water mark of what this rank would hold
    must stay inside it, i.e. 8 x peak < 288 GB at full size."""
    import json, os, time
    from genome_amd.dist import DistDNAMap, HipDist, unique_id
    W = 2 if k > 32 else 1
    slot = 16 if W == 1 else 24
    total = pm.size()
    want = pm.verify()
    ctx.trim()
    # what is on this GPU but would not be on rank 0's in the real run: the reads buffer and the other seven partitions' tables
    base0 = ctx.mem_stats(reset_peak=True)["live"] - pm.parts[0].slots() * slot
    ctx.set_mem_budget(budget + base0)
    report = {"config": tag, "k": k, "keys": total, "budget_bytes": budget}
    try:
        # ---- this rank's own partition through the real collective (a communicator of one rank): chunked export + insert
        dist = HipDist(ctx, 0, 1, unique_id())
        dm = DistDNAMap.__new__(DistDNAMap)
        dm.dist, dm.ctx, dm.k, dm.local = dist, ctx, k, pm.parts[0]
        #      in its CLASSIFIED form: the partition's keys are classified by their owner (here: all by this rank) and the masks travel
        #      with them; the graph buildGraph derives from the masks must be the graph it derives by its own neighbour lookups
        own = dm.gathered(classified=True)
        assert own.verify_checksum() == pm.parts[0].verify_checksum()
        g1 = buildGraph(k, own)
        assert g1.buildStats()["classified_by_owners"]
        by_masks = (g1.counts(), g1.checksum())
        g1.close()
        g2 = buildGraph(k, own)                                   # (the masks served one build)
        assert not g2.buildStats()["classified_by_owners"]
        assert by_masks == (g2.counts(), g2.checksum()) and by_masks[0][0] > 0
        g2.close()
        own.close(); dist.close(); ctx.trim()
        ctx.mem_stats(reset_peak=True)
        # ---- the replica: every partition's survivors in one table sized for the graph phase
        t0 = time.perf_counter()
        full = pm.merged()
        report["gather_s"] = time.perf_counter() - t0
        assert full.size() == total and full.verify_checksum() == want
        peak_gather = ctx.mem_stats()["peak"] - base0                 # (base0 holds the reads AND the seven partitions that are elsewhere in the real run)
        report["gather_peak_bytes"] = peak_gather
        report["replica_slots"], report["replica_load"] = full.slots(), total / full.slots()
        pm.close()
        ctx.trim()
        ctx.mem_stats(reset_peak=True)
        base1 = ctx.mem_stats()["live"] - full.slots() * slot
        # ---- buildGraph, both constructions: the same graph
        out = {}
        for mode, name in ((1, "walk"), (2, "pj")):
            ctx.set_option("graph_unitigs", mode)
            t0 = time.perf_counter()
            g = buildGraph(k, full)
            out[name] = (g.counts(), g.checksum())
            report[name + "_build_s"] = time.perf_counter() - t0
            report[name + "_build_peak_bytes"] = ctx.mem_stats(reset_peak=True)["peak"] - base1
            if mode == 1:
                g.close()
                ctx.mem_stats(reset_peak=True)
        ctx.set_option("graph_unitigs", 0)
        assert out["walk"] == out["pj"], out
        (nodes, edges, total_len), _ = out["pj"]
        report.update(nodes=nodes, edges=edges, edge_bases=total_len)
        assert nodes > 0 and nodes % 2 == 0 and edges >= nodes // 2
        assert total_len >= total - nodes // 2                      # every interior k-mer lies on an edge (both strands walk it)
        # the reference's own invariants on a sample
        e = g.getEdges()
        rnd = np.random.default_rng(5)
        for i in rnd.choice(edges, min(edges, 200), replace=False):
            o, ln = int(e["off"][i]), int(e["len"][i])
            seq = dna.unpack_2bit(e["seq"][o:o + (ln + 3) // 4], ln)
            s_, t_ = dna.unpack(int(e["slo"][i]), int(e["shi"][i]), k), dna.unpack(int(e["elo"][i]), int(e["ehi"][i]), k)
            assert (s_ + seq).endswith(t_)
            assert g.nodeId(dna.rev_complement(s_))[0] is not None     # termKmers = T ++ T.map(revComplement)
        del e
        g.removeBubbles()
        assert g.counts()[0] == nodes and g.counts()[1] <= edges
        g.simplifyGraph()
        h1, h2 = g.componentHistograms()
        kept, comps = g.retainLargest()
        assert sum(c for _, c in h1) == comps == sum(c for _, c in h2)
        assert kept == h1[-1][0] and g.counts()[0] == kept
        report["final"] = g.counts()
        report["simplify_retain_peak_bytes"] = ctx.mem_stats()["peak"] - base1
        g.close()
        peak = max(peak_gather, report["walk_build_peak_bytes"], report["pj_build_peak_bytes"], report["simplify_retain_peak_bytes"])
        report["stage_peak_bytes"], report["times_8_GB"] = peak, 8 * peak / 1e9
        os.makedirs("gpurun_out", exist_ok=True)
        with open(f"gpurun_out/graph_stage_{tag}.json", "w") as f:
            json.dump(report, f, indent=1)
        assert 8 * peak < 288e9, report
        full.close()
    finally:
        ctx.set_mem_budget(0)
        ctx.set_option("graph_unitigs", 0)


def _share(ctx, k, n, G, e, cfg, chunk, with_plain, graph_stage=None):
    """One rank's share of an 8-GPU configuration on P = 8 logical partitions, pre-filtered."""
    L_, P = 150, 8
    stride = synth.record_stride(L_)
    nk = L_ - k + 1
    d = ctx.alloc(chunk * stride + 64)

    def chunks():
        for first in range(0, n, chunk):
            c = min(chunk, n - first)
            ctx.synth_reads(d, c, L_, "G", cfg, first, G, e)
            ctx.sync()
            yield d, c

    solid = G                                            # ~ one k-mer per genome position
    errors = int(n * L_ * e * k)                         # ~ k new k-mers per sequencing error, almost all seen once
    pm = PartitionedDNAMap(ctx, k, P, capacity_hint=int(solid * 1.2 + errors * 0.3))
    looked, admitted = pm.count_reads_dev_prefiltered(chunks(), L_, solid + errors)
    assert looked == n * nk
    assert 0 < admitted < looked
    live, bad, total, chk = pm.verify()
    assert live == pm.size() and bad == 0 and total == admitted
    assert pm.foreign_keys() == 0
    sizes = [p.size() for p in pm.parts]
    assert max(sizes) < 2.5 * (sum(sizes) / P), sizes      # minimizer owners: uneven, but every partition carries its part
    pm.deleteAll_lt(3)
    filtered = pm.verify()
    assert filtered[0] == pm.size() and filtered[1] == 0
    assert 0.5 * solid < filtered[0] < 1.6 * solid          # what survives is the genome's k-mers (plus errors seen 3 times)
    if graph_stage:
        _graph_stage(ctx, pm, k, 36_000_000_000, graph_stage)     # (closes pm)
    else:
        pm.close()
    if with_plain:
        plain = PartitionedDNAMap(ctx, k, P, capacity_hint=int((solid + errors) * 1.05))
        occ = 0
        for dd, c in chunks():
            occ += plain.count_reads_dev(dd, c, L_)
        assert occ == n * nk
        lv, bd, tot, _ = plain.verify()
        assert bd == 0 and tot == occ and plain.foreign_keys() == 0
        plain.deleteAll_lt(3)
        assert plain.verify() == filtered, "survivors with and without the pre-filter must be the same table"
        plain.close()
    ctx.free(d)


def test_c4_one_rank_share_k55_partitioned_prefiltered(ctx):
    """configs[3] / 8: 62.5 M x 150 bp over 187.5 Mbp (50x), k = 55, e = 0.5 %; 6e9 windows of 16-byte keys."""
    _share(ctx, k=55, n=62_500_000, G=187_500_000, e=0.005, cfg=4, chunk=12_500_000, with_plain=True)


def test_c5_one_rank_share_k63_partitioned_prefiltered(ctx):
    """configs[4] / 8: 250 M x 150 bp over 387.5 Mbp (~97x), k = 63, e = 0.2 %; 2.2e10 windows.  The unfiltered table
    (~5e9 16-byte keys) does not fit one GPU — which is what the pre-filter is for — so no with/without comparison here.
    Then the graph stage of configs[4] ("full build + simplify, HBM-resident graph") as one rank holds it: _graph_stage."""
    _share(ctx, k=63, n=250_000_000, G=387_500_000, e=0.002, cfg=5, chunk=25_000_000, with_plain=False, graph_stage="c5_share_k63")


@pytest.mark.parametrize("k,L_,hint", [(31, 150, 2_000_000_000), (55, 150, 960_000_000)])
def test_table_beyond_34_gb_stays_on_the_partitioned_pipeline(ctx, k, L_, hint):
    """C4's and C5's per-rank tables are 50-100 GB.  256 L1 buckets x 4096 fine buckets x 32 KiB end at 34 GB: a table that
    needs more gets 512 or 1024 L1 buckets (plan_segments) and keeps the LDS segment build — which used to hand such tables to
    the direct path.  A table created for `hint` keys (36-37 GB), two batches through the forced partitioned path (from empty,
    then on top of the content): every window counted, every key at its own slot and stored once (gk_map_verify), no direct
    launch, and the (key, count) set equal to what a small table holds after the same two batches (order-independent
    checksum).  The oracle cannot follow 2.4e8 windows; parity of the 512/1024-bucket forms at oracle sizes is what the whole
    suite checks under GK_MIN_LNB1=9|10."""
    n = 1_000_000
    nk = L_ - k + 1
    d = ctx.alloc(n * synth.record_stride(L_) + 64)
    sums = []
    for cap in (hint, n * nk):
        m = HipDNAMap(ctx, k, cap)
        if cap == hint:
            st = m.stats()
            assert st["slots"] * st["slot_bytes"] > 35e9, st
            m.set_insert_path("partitioned")
        ctx.synth_reads(d, n, L_, "U", 70 + k, 0, 0, 0.0)
        assert m.count_reads_dev(d, n, L_) == n * nk
        ctx.synth_reads(d, n, L_, "G", 71 + k, 0, 3_000_000, 0.01)
        assert m.count_reads_dev(d, n, L_) == n * nk
        st = m.stats()
        if cap == hint:
            assert st["partitioned_launches"] == 2 and st["direct_launches"] == 0 and st["retries_direct"] == 0, st
        live, bad, total, chk = m.verify_checksum()
        assert bad == 0 and total == 2 * n * nk and live == m.size()
        sums.append((live, chk))
        m.close()
    assert sums[0] == sums[1]
    ctx.free(d)
    ctx.trim()
