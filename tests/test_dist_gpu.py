"""gk_dist_*: PartitionedDNAMap over RCCL behind the C-ABI (-m gpu).  The 1-GPU box can only form a communicator of ONE
rank: route -> counts exchange -> record exchange (ncclSend/Recv to self) -> owner count -> all-reduce -> all-gather all run
through RCCL, with every record going to rank 0.  N > 1 placement logic is covered by the logical-partition tests
(test_table_gpu.py, test_configs_gpu.py) and the 2-rank gloo tests (test_exchange_cpu.py); N > 1 over RCCL is measured
only by the driver's 8-GPU run."""
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


@pytest.mark.parametrize("k,L_", [(31, 150), (47, 120)])
def test_streaming_route_begin_count_routed(dist, k, L_):
    """The two-halves form of FreqFilter.add over a partitioned map: the route of batch i+1 is launched before batch i is
    exchanged and counted (two send buffers).  Same table as one count over all the reads."""
    ctx = dist.ctx
    n, nb = 40000, 4
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
    pm.route_begin(at(0), per, L_)
    tot_s = tot_o = 0
    for i in range(nb):
        if i + 1 < nb:
            pm.route_begin(at(i + 1), per, L_)
            if i == 0:
                with pytest.raises(L.GkError) as e:
                    pm.route_begin(at(2), per, L_)   # a third one: both send buffers are taken
                assert e.value.code == L.GK_E_STATE
        s_, o_ = pm.count_routed()
        tot_s += s_; tot_o += o_
    assert tot_s == tot_o == occ
    assert pm.size() == ref.size()
    for a, b in zip(pm.local.sorted_items(), ref.export_sorted()):
        assert np.array_equal(a, b)
    assert pm.local.verify()[1] == 0
    pm.close(); ctx.free(d)


def test_collectives_and_errors(dist):
    assert np.array_equal(dist.allreduce([1.5, 2.5, -3.0]), [1.5, 2.5, -3.0])
    assert np.array_equal(dist.allreduce([4.0], "max"), [4.0])
    dist.barrier()
    with pytest.raises(L.GkError) as e:
        HipDist(dist.ctx, 3, 2, unique_id())               # rank outside the world: refused before any RCCL call
    assert e.value.code == L.GK_E_INVALID
    pm = DistDNAMap(dist, 31)
    assert pm.count_reads_dev(0, 0, 150) == (0, 0)
    with pytest.raises(L.GkError) as e:
        pm.count_reads_dev(0, 5, 150)
    assert e.value.code == L.GK_E_INVALID
    with pytest.raises(L.GkError) as e:
        pm.count_reads_dev(1, 5, 300)
    assert e.value.code == L.GK_E_FORMAT
    pm.close()
