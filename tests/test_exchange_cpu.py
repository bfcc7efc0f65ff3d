"""The N>1 path on CPU: two gloo ranks run the same routing + exchange code bench.py runs over
RCCL (genome_amd.partitioned.exchange_keys), with the per-rank extract/insert legs played by the
oracle (no GPU here).  The union of the two owner partitions must equal the single-process table,
every key must sit on the rank the owner function names, and x / rc(x) must share an owner."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, k, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from genome_amd import _lib, synth
    from genome_amd.partitioned import exchange_keys
    from oracle import oracle as O
    W = 1 if k <= 32 else 2
    n, L_ = 300, 90
    rec = synth.reads_mode_g(n, L_, 2500, 0.01, config_id=77, first_read=rank * n)   # weak scaling: own reads
    # extract + canonicalise + bucket by owner (the gk_shard_reads_dev leg, restated with the oracle)
    lib = O.lib()
    buckets = [[] for _ in range(world)]
    for r in range(n):
        body = np.ascontiguousarray(rec[r, 1:])
        for p in range(L_ - k + 1):
            x = lib.gko_kmer_from_packed(body.ctypes.data_as(O.C.POINTER(O.C.c_uint8)), p, k)
            y = lib.gko_canon(x, k)
            own = _lib.lib().gk_owner_of(k, y.lo, y.hi, world)
            assert own == _lib.lib().gk_owner_of(k, x.lo, x.hi, world)      # strand symmetric
            buckets[own].append((y.lo, y.hi))
    counts = np.array([len(b) for b in buckets], np.int64)
    flat = []
    for b in buckets:
        for lo, hi in b:
            flat += [lo] if W == 1 else [lo, hi]
    send = torch.from_numpy(np.array(flat, np.uint64).view(np.int64).copy())
    recv, rcounts = exchange_keys(dist, send, counts, W)
    got = recv[:int(rcounts.sum()) * W].numpy().view(np.uint64).reshape(-1, W)
    # owner-side insert (the gk_map_update_inc_dev leg)
    pm = O.PMap(k, 1)
    for row in got:
        lo, hi = int(row[0]), int(row[1]) if W == 2 else 0
        assert _lib.lib().gk_owner_of(k, lo, hi, world) == rank
        pm.update_inc(lo, hi)
    lo, hi, cnt = pm.export_sorted()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=lo, hi=hi, cnt=cnt, sent=counts, received=rcounts)
    total = torch.tensor([pm.size()], dtype=torch.int64)
    dist.all_reduce(total)                                                   # `size` = scalar all-reduce
    np.save(os.path.join(out_dir, f"size{rank}.npy"), total.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("k", [21, 47])
def test_two_rank_exchange_equals_single_table(tmp_path, k):
    world, port = 2, 29500 + os.getpid() % 2000 + k
    mp.start_processes(_worker, args=(world, port, k, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    sys.path.insert(0, ROOT)
    from genome_amd import synth
    from oracle import oracle as O
    ref = O.PMap(k, 1)
    for rank in range(world):
        rec = synth.reads_mode_g(300, 90, 2500, 0.01, config_id=77, first_read=rank * 300)
        ref.count_reads(rec.tobytes(), 300)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    lo = np.concatenate([p["lo"] for p in parts]); hi = np.concatenate([p["hi"] for p in parts])
    cnt = np.concatenate([p["cnt"] for p in parts])
    order = np.lexsort((lo, hi))
    rlo, rhi, rcnt = ref.export_sorted()
    assert np.array_equal(lo[order], rlo) and np.array_equal(hi[order], rhi) and np.array_equal(cnt[order], rcnt)
    assert len(set(zip(lo.tolist(), hi.tolist()))) == len(lo)               # partitions are disjoint
    assert int(np.load(tmp_path / "size0.npy")[0]) == ref.size() == int(np.load(tmp_path / "size1.npy")[0])
    # what rank a sent to rank b is what rank b says it received from a
    assert parts[0]["sent"][1] == parts[1]["received"][0] and parts[1]["sent"][0] == parts[0]["received"][1]
    assert min(len(p["lo"]) for p in parts) > 0


def _worker_records(rank, world, port, out_dir):
    """exchange_records (the super-k-mer form bench.py uses over RCCL) on two gloo ranks: records are
    synthetic fixed slots tagged with (source rank, destination rank, serial)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from genome_amd.partitioned import exchange_records
    slot, region = 16, 50
    rec_counts = np.array([7 + 3 * rank, 11 - 2 * rank], np.int64)      # records for rank 0 / rank 1
    kmer_counts = rec_counts * (5 + rank)
    send = torch.zeros(world * region * slot, dtype=torch.uint8)
    for p in range(world):
        for i in range(int(rec_counts[p])):
            o = (p * region + i) * slot
            send[o], send[o + 1], send[o + 2] = rank, p, i
    recv, nrec, nkm = exchange_records(dist, send, rec_counts, kmer_counts, slot, region)
    got = recv[:nrec * slot].numpy().reshape(-1, slot)
    np.savez(os.path.join(out_dir, f"rec{rank}.npz"), got=got, nrec=nrec, nkm=nkm)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_record_exchange(tmp_path):
    world, port = 2, 31500 + os.getpid() % 2000
    mp.start_processes(_worker_records, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    sent = {0: np.array([7, 11]), 1: np.array([10, 9])}
    for rank in range(world):
        z = np.load(tmp_path / f"rec{rank}.npz")
        want_n = sent[0][rank] + sent[1][rank]
        assert int(z["nrec"]) == want_n == len(z["got"])
        assert int(z["nkm"]) == sent[0][rank] * 5 + sent[1][rank] * 6
        rows = [tuple(int(x) for x in r[:3]) for r in z["got"]]
        want = [(0, rank, i) for i in range(sent[0][rank])] + [(1, rank, i) for i in range(sent[1][rank])]
        assert rows == want            # grouped by source rank, in order; nothing from the region padding
