"""gk_dist_*: PartitionedDNAMap over RCCL behind the C-ABI (-m gpu).  The 1-GPU box can only form a communicator of ONE
rank: route -> counts exchange -> record exchange (ncclSend/Recv to self) -> owner count -> all-reduce -> all-gather all run
through RCCL, with every record going to rank 0.  world = 2 and 3 run over the library's loopback transport (ranks as
threads of this process, gk_dist_create_loopback): everything above the transport is the product code.  N > 1 placement logic
is also covered by the logical-partition tests (test_table_gpu.py, test_configs_gpu.py) and the 2-rank gloo tests
(test_exchange_cpu.py); N > 1 over RCCL is measured only by the driver's 8-GPU run."""
import random

import numpy as np
import pytest

from genome_amd import dna, synth
from genome_amd import _lib as L
from genome_amd.dist import DistDNAMap, HipDist, unique_id
from genome_amd.dnamap import Context
from genome_amd.graph import buildGraph
from oracle import oracle as O

pytestmark = pytest.mark.gpu


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


@pytest.fixture(scope="module")
def dist():
    ctx = Context(0)
    d = HipDist(ctx, 0, 1, unique_id())
    yield d
    d.close(); ctx.close()


@pytest.mark.parametrize("k,L_", [(21, 100), (31, 150), (55, 150), (64, 150)])
def test_one_rank_communicator_matches_oracle(dist, k, L_):
    ctx = dist.ctx
    n = 30000
    rec = synth.reads_mode_g(n, L_, 50000, 0.01, config_id=300 + k)
    d = ctx.alloc(rec.size + 64)
    ctx.upload(d, rec)
    ref = O.PMap(k, 1)
    occ = ref.count_reads(rec.tobytes(), n)
    pm = DistDNAMap(dist, k)
    sent, owned = pm.count_reads_dev(d, n // 2, L_)
    s2, o2 = pm.count_reads_dev(d + (n // 2) * rec.shape[1], n - n // 2, L_)          # a second batch on top
    assert sent + s2 == owned + o2 == occ
    assert pm.size() == ref.size() == pm.local.size()
    for a, b in zip(pm.local.sorted_items(), ref.export_sorted()):
        assert np.array_equal(a, b)
    ms = dist.last_ms()
    assert ms["total"] > 0 and ms["total"] >= ms["owner_count"]
    pm.deleteAll_lt(3); ref.delete_lt(3)
    full = pm.gathered()
    assert full.verify_checksum() == pm.local.verify_checksum()
    for a, b in zip(full.sorted_items(), ref.export_sorted()):
        assert np.array_equal(a, b)
    g, og = buildGraph(k, full), O.Graph(ref)
    assert g.canonical() == oracle_canonical(og)
    g.close(); full.close(); pm.close(); ctx.free(d)


@pytest.mark.parametrize("depth", [2, 3, -3])
@pytest.mark.parametrize("k,L_", [(31, 150), (47, 120)])
def test_streaming_route_begin_count_routed(dist, k, L_, depth):
    """The two-halves form of FreqFilter.add over a partitioned map.  depth 2: the route of batch i+1 is launched before batch
    i is exchanged and counted.  depth 3: with batch i+2 begun as well, count_routed(i) posts the exchange of batch i+1 on the
    communication stream before it counts batch i (three send buffers, two receive buffers).  Same table as one count over all
    the reads; a size query in the middle (an RCCL operation on the other stream) must not disturb a posted exchange."""
    ctx = dist.ctx
    if depth < 0:                                    # three deep, the exchange NOT posted ahead (the fallback order)
        depth = -depth
        ctx.set_option("dist_exchange_ahead", 0)
    n, nb = 40000, 5
    rec = synth.reads_mode_g(n, L_, 60000, 0.01, config_id=500 + k)
    d = ctx.alloc(rec.size + 64)
    ctx.upload(d, rec)
    ref = O.PMap(k, 1)
    occ = ref.count_reads(rec.tobytes(), n)
    pm = DistDNAMap(dist, k, 1 << 10)                # small hint: the table also grows while routes are in flight
    per = n // nb
    at = lambda i: d + i * per * rec.shape[1]
    with pytest.raises(L.GkError) as e:
        pm.count_routed()                            # nothing begun
    assert e.value.code == L.GK_E_STATE
    begun = 0
    tot_s = tot_o = 0
    for i in range(nb):
        while begun < min(i + depth, nb):
            pm.route_begin(at(begun), per, L_)
            begun += 1
        if i == 0 and depth == 3:
            with pytest.raises(L.GkError) as e:
                pm.route_begin(at(3), per, L_)       # a fourth one: all three send buffers are taken
            assert e.value.code == L.GK_E_STATE
        s_, o_ = pm.count_routed()
        tot_s += s_; tot_o += o_
        if i == 2:
            assert pm.size() <= ref.size()           # all-reduce while (depth 3) the next batch's records may be on the wire
    assert tot_s == tot_o == occ
    assert pm.size() == ref.size()
    for a, b in zip(pm.local.sorted_items(), ref.export_sorted()):
        assert np.array_equal(a, b)
    assert pm.local.verify()[1] == 0
    ctx.set_option("dist_exchange_ahead", -1)
    pm.close(); ctx.free(d)


@pytest.mark.parametrize("world,k,L_", [(2, 31, 150), (3, 47, 120), (8, 31, 100)])
def test_several_ranks_over_the_loopback_transport(world, k, L_):
    """gk_dist_* with world > 1 on the one-GPU box: RCCL refuses two ranks on one device, so the ranks are THREADS of this
    process, each with its own context, handle and partition, talking through the library's loopback transport (device-to-
    device copies matched pairwise in posting order; a group's end blocks until the peers have posted theirs, so an inconsistent
    order of operations across ranks would deadlock here, and a send whose size differs from its receive is an error).  Each
    rank streams its own reads three batches deep (the exchange of batch i+1 on the helper thread beside the count of batch i),
    with a size query in the middle.  Then: windows sent == windows counted by their owners == all windows, the partitions'
    sizes add up to the oracle's table, every stored key is at its owner, and the table gathered on EVERY rank is the
    oracle's, bit for bit."""
    import threading
    n, nb = 30000, 5
    rec = synth.reads_mode_g(n * world, L_, 50000 * world, 0.01, config_id=700 + k)
    stride = rec.shape[1]
    ref = O.PMap(k, 1)
    occ = ref.count_reads(rec.tobytes(), n * world)
    want = ref.export_sorted()
    id128 = bytes(random.Random(world * 1000 + k).getrandbits(8) for _ in range(128))
    out, errors = [None] * world, []

    def run(rank):
        try:
            c = Context(0)
            hd = HipDist(c, rank, world, id128, loopback=True)
            pm = DistDNAMap(hd, k, 1 << 10)
            d = c.alloc(n * stride + 64)
            c.upload(d, rec[rank * n:(rank + 1) * n])
            per = n // nb
            begun = sent = owned = 0
            for i in range(nb):
                while begun < min(i + 3, nb):
                    pm.route_begin(d + begun * per * stride, per, L_)
                    begun += 1
                s_, o_ = pm.count_routed()
                sent += s_; owned += o_
                if i == 2:
                    assert pm.size() <= len(want[0])
            tot = hd.allreduce([float(sent), float(owned)], "sum")
            size = pm.size()
            assert pm.local.verify()[1] == 0
            full = pm.gathered()
            out[rank] = (tot.tolist(), size, pm.local.size(), full.sorted_items())
            hd.barrier()
            full.close(); pm.close(); c.free(d); hd.close(); c.close()
        except BaseException as e:          # noqa: BLE001 — reported by the main thread
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=240)
    assert not any(t.is_alive() for t in threads), "a rank is stuck: the ranks issued their operations in different orders"
    assert not errors, errors
    assert sum(o[2] for o in out) == len(want[0])
    for tot, size, _local, items in out:
        assert tot == [float(occ), float(occ)]
        assert size == len(want[0])
        for a, b in zip(items, want):
            assert np.array_equal(a, b)


def _run_ranks(world, body, timeout=240):
    """`body(rank, ctx, hd)` on one thread per rank over the loopback transport; returns the ranks' results, fails on a stuck rank."""
    import threading
    id128 = bytes(random.Random(world * 7919 + 13).getrandbits(8) for _ in range(128))
    out, errors = [None] * world, []

    def run(rank):
        try:
            c = Context(0)
            hd = HipDist(c, rank, world, id128, loopback=True)
            out[rank] = body(rank, c, hd)
            hd.barrier()
            hd.close(); c.close()
        except BaseException as e:          # noqa: BLE001 — reported by the main thread
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=timeout)
    assert not any(t.is_alive() for t in threads), "a rank is stuck"
    assert not errors, errors
    return out


@pytest.mark.parametrize("world,k,L_", [(3, 31, 150), (2, 55, 150)])
def test_one_rank_fails_its_route_and_every_rank_drops_that_batch(world, k, L_):
    """A failure that only ONE rank has — its route cannot be completed, an allocation fails (injected: test_dist_fail_exchange
    on rank 1's context) — must not leave the peers waiting in a receive: the failing rank says so in the status word of the
    counts exchange, EVERY rank drops that batch and reports GK_E_COMM from the same count_routed, nobody hangs, and the batches
    after it are counted as if nothing had happened: the gathered table is the oracle's over the reads of all OTHER batches."""
    n, nb, bad = 24000, 5, 2
    per = n // nb
    rec = synth.reads_mode_g(n * world, L_, 40000 * world, 0.01, config_id=900 + k)
    stride = rec.shape[1]
    keep = np.concatenate([rec[r * n + i * per: r * n + (i + 1) * per] for r in range(world) for i in range(nb) if i != bad])
    ref = O.PMap(k, 1)
    occ = ref.count_reads(keep.tobytes(), len(keep))
    want = ref.export_sorted()

    def body(rank, c, hd):
        pm = DistDNAMap(hd, k, 1 << 10)
        d = c.alloc(n * stride + 64)
        c.upload(d, rec[rank * n:(rank + 1) * n])
        begun = sent = owned = 0
        failed = []
        for i in range(nb):
            while begun < min(i + 3, nb):
                pm.route_begin(d + begun * per * stride, per, L_)
                begun += 1
            if rank == 1 and i == bad - 1:
                c.set_option("test_dist_fail_exchange", 4)      # batch `bad` is settled inside this call (its exchange goes ahead)
            try:
                s_, o_ = pm.count_routed()
                sent += s_; owned += o_
            except L.GkError as e:
                assert e.code == L.GK_E_COMM, e
                failed.append((i, str(e)))
        tot = hd.allreduce([float(sent), float(owned)], "sum")
        full = pm.gathered()
        res = (failed, tot.tolist(), full.sorted_items(), pm.local.verify()[1])
        full.close(); pm.close(); c.free(d)
        return res

    for rank, (failed, tot, items, badslots) in enumerate(_run_ranks(world, body)):
        assert [i for i, _ in failed] == [bad], (rank, failed)
        assert "rank 1" in failed[0][1]
        assert ("injected" in failed[0][1]) == (rank == 1)          # only the failing rank knows why
        assert tot == [float(occ), float(occ)] and badslots == 0
        for a, b in zip(items, want):
            assert np.array_equal(a, b)


def test_a_send_region_that_is_too_small_is_routed_again_in_place():
    """One rank's send buffer is far too small for a batch (injected: test_dist_small_send): the route overflows its regions, the
    owner thread routes the batch again into a bigger buffer before its exchange is posted, and nothing is lost or counted twice."""
    world, k, L_, n, nb = 2, 31, 120, 20000, 4
    per = n // nb
    rec = synth.reads_mode_g(n * world, L_, 60000, 0.01, config_id=950)
    stride = rec.shape[1]
    ref = O.PMap(k, 1)
    occ = ref.count_reads(rec.tobytes(), n * world)
    want = ref.export_sorted()

    def body(rank, c, hd):
        pm = DistDNAMap(hd, k, 1 << 10)
        d = c.alloc(n * stride + 64)
        c.upload(d, rec[rank * n:(rank + 1) * n])
        begun = sent = owned = 0
        for i in range(nb):
            while begun < min(i + 3, nb):
                if rank == 0 and begun in (1, 3):
                    c.set_option("test_dist_small_send", 8)
                pm.route_begin(d + begun * per * stride, per, L_)
                begun += 1
            s_, o_ = pm.count_routed()
            sent += s_; owned += o_
        tot = hd.allreduce([float(sent), float(owned)], "sum")
        full = pm.gathered()
        res = (tot.tolist(), full.sorted_items())
        full.close(); pm.close(); c.free(d)
        return res

    for tot, items in _run_ranks(world, body):
        assert tot == [float(occ), float(occ)]
        for a, b in zip(items, want):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("world,k,L_", [(1, 31, 150), (2, 31, 150), (3, 55, 150), (8, 21, 100), (2, 64, 150)])
def test_classify_by_the_owners_gives_the_same_graph(world, k, L_):
    """SURVEY.md 8(e) "beyond counting" (Graph.scala:320-329 through PartitionedDNAMap.mapReduce :55-58): in the classified
    gather every rank classifies ITS keys — own neighbours looked up locally, the others asked of their owners in one query
    all-to-all per chunk — and the masks travel with the keys.  On every rank: the gathered table is the oracle's, buildGraph on
    it takes the masks (no neighbour lookups: classified_by_owners), and the graph is the oracle's node for node and base for
    base, the same as from the plain gather; ranks > 1 did ask their peers."""
    n = 12000
    rec = synth.reads_mode_g(n * world, L_, 30000 * world, 0.01, config_id=1100 + k)
    stride = rec.shape[1]
    ref = O.PMap(k, 1)
    ref.count_reads(rec.tobytes(), n * world)
    ref.delete_lt(2)
    want = ref.export_sorted()
    want_graph = oracle_canonical(O.Graph(ref))

    def body(rank, c, hd):
        pm = DistDNAMap(hd, k, 1 << 10)
        d = c.alloc(n * stride + 64)
        c.upload(d, rec[rank * n:(rank + 1) * n])
        pm.count_reads_dev(d, n, L_)
        pm.deleteAll_lt(2)
        local_items = pm.local.sorted_items()
        full = pm.gathered(classified=True)
        assert all(np.array_equal(a, b) for a, b in zip(pm.local.sorted_items(), local_items))      # the partition is unchanged
        items = full.sorted_items()
        g = buildGraph(k, full)
        st = g.buildStats()
        canon, chk = g.canonical(), (g.counts(), g.checksum())
        g2 = buildGraph(k, full)                          # the masks served one build: this one classifies by itself
        st2 = g2.buildStats()
        chk2 = (g2.counts(), g2.checksum())
        plain = pm.gathered()
        g3 = buildGraph(k, plain)
        res = (items, canon, chk, chk2, (g3.counts(), g3.checksum()), st["classified_by_owners"], st2["classified_by_owners"],
               g3.buildStats()["classified_by_owners"], pm.classify_queries(), pm.local.size())
        g.close(); g2.close(); g3.close(); full.close(); plain.close(); pm.close(); c.free(d)
        return res

    out = _run_ranks(world, body)
    assert sum(o[9] for o in out) == len(want[0])
    for items, canon, chk, chk2, chk3, by_owners, by_owners2, by_owners3, queries, _ in out:
        for a, b in zip(items, want):
            assert np.array_equal(a, b)
        assert canon == want_graph
        assert chk == chk2 == chk3
        assert by_owners and not by_owners2 and not by_owners3
        assert (queries > 0) == (world > 1)
    if world > 1:          # most neighbours share their k-mer's minimizer, hence its owner: far fewer than 8 questions per key
        assert sum(o[8] for o in out) < 4 * len(want[0])


def test_classified_gather_when_one_rank_cannot_stage_its_queries():
    """A failure on ONE rank inside the classified gather (injected: the staging of its queries cannot be allocated) is agreed
    on before anybody posts a receive: every rank returns an error from the same call, nobody hangs, and a plain gather right
    after it works."""
    world, k, L_, n = 3, 31, 120, 9000
    rec = synth.reads_mode_g(n * world, L_, 40000, 0.01, config_id=1177)
    stride = rec.shape[1]
    ref = O.PMap(k, 1)
    ref.count_reads(rec.tobytes(), n * world)
    want = ref.export_sorted()

    def body(rank, c, hd):
        pm = DistDNAMap(hd, k, 1 << 10)
        d = c.alloc(n * stride + 64)
        c.upload(d, rec[rank * n:(rank + 1) * n])
        pm.count_reads_dev(d, n, L_)
        if rank == 2:
            c.set_option("test_dist_fail_classify", 1)
        err = None
        try:
            pm.gathered(classified=True).close()
        except L.GkError as e:
            err = (e.code, str(e))
        full = pm.gathered()
        items = full.sorted_items()
        full.close(); pm.close(); c.free(d)
        return err, items

    for rank, (err, items) in enumerate(_run_ranks(world, body)):
        assert err is not None, rank
        assert ("injected" in err[1]) == (rank == 2), (rank, err)
        assert err[0] == (L.GK_E_CAPACITY if rank == 2 else L.GK_E_COMM)
        for a, b in zip(items, want):
            assert np.array_equal(a, b)


def test_classified_gather_steps_back_when_a_partition_holds_noncanonical_keys():
    """Verbatim inserts (gk_map_update_inc) may leave a key that is not the hash-rule orientation of its k-mer in ONE rank's
    partition: `contains` must then probe both strands everywhere (Graph.scala:270), which the owners' masks do not promise.
    Every rank takes the same decision — plain gather, buildGraph classifies by itself — and the graph is the oracle's for the
    same table."""
    world, k, L_, n = 2, 31, 120, 8000
    rec = synth.reads_mode_g(n * world, L_, 30000, 0.01, config_id=1190)
    stride = rec.shape[1]
    ref = O.PMap(k, 1)
    ref.count_reads(rec.tobytes(), n * world)
    ref.delete_lt(2)
    lo, hi, _ = ref.export_sorted()
    # the reverse complements of a few keys that rank 1 owns (x and rc(x) share their owner), inserted verbatim there
    flip = []
    for a, b in zip(lo.tolist(), hi.tolist()):
        if L.lib().gk_owner_of(k, a, b, world) == 1:
            s = dna.unpack(a, b, k)
            if dna.rev_complement(s) != s:
                flip.append(dna.rev_complement(s))
        if len(flip) == 40:
            break
    for s in flip:
        plo, phi = dna.pack(s)
        ref.update_inc(plo, phi)
    want_graph = oracle_canonical(O.Graph(ref))

    def body(rank, c, hd):
        pm = DistDNAMap(hd, k, 1 << 10)
        d = c.alloc(n * stride + 64)
        c.upload(d, rec[rank * n:(rank + 1) * n])
        pm.count_reads_dev(d, n, L_)
        pm.deleteAll_lt(2)
        if rank == 1:
            pm.local.update_inc(flip)
            assert pm.local.stats()["noncanonical_keys"] is True
        full = pm.gathered(classified=True)
        g = buildGraph(k, full)
        res = (g.buildStats()["classified_by_owners"], g.canonical(), full.stats()["noncanonical_keys"])
        g.close(); full.close(); pm.close(); c.free(d)
        return res

    for by_owners, canon, dirty in _run_ranks(world, body):
        assert not by_owners and dirty
        assert canon == want_graph
