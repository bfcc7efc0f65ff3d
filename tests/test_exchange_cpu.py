"""What of the N > 1 path can run WITHOUT a GPU, on two gloo ranks (world_size 2):

  * the owner function the library routes by (gk_owner_of: strand-symmetric minimizer, host-callable) together with the
    all-to-all pattern of PartitionedDNAMap.update (S/ds/PartitionedDNAMap.scala:37-47, 60-63): every rank buckets the
    canonical k-mers of ITS reads by owner, one personalised exchange, owner-side inserts; the union of the two partitions
    must be the single-process table, partitions disjoint, every key on the rank the owner function names.  The exchange here
    is torch.distributed's own all_to_all over gloo, written out in this file — it is the PATTERN that is tested, not product
    code: the product's exchange is gk_dist_* (RCCL inside the library, csrc/gk_dist.hip), which needs a GPU and is tested in
    tests/test_dist_gpu.py (one rank over RCCL; 2, 3 and 8 ranks, fault injection included, over the loopback transport);
  * the bootstrap bench.py really uses for N > 1 (genome_amd.dist.share_id: rank 0's 128-byte communicator id to every rank
    over gloo) and bench.py's own launcher (--gpus N without WORLD_SIZE: N child processes, rank 0's line forwarded, the
    worst exit code returned).
The extract / insert legs are played by the oracle (no GPU here)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _all_to_all_keys(send, send_counts, W):
    """one personalised exchange of W-word keys grouped by destination: counts first, then the keys (gloo)"""
    sc = torch.as_tensor(np.asarray(send_counts, dtype=np.int64))
    rc = torch.empty_like(sc)
    dist.all_to_all_single(rc, sc)
    recv_counts = rc.numpy().astype(np.int64)
    recv = torch.empty(max(int(recv_counts.sum()) * W, 1), dtype=torch.int64)
    nsend, nrecv = int(np.sum(send_counts)), int(recv_counts.sum())
    dist.all_to_all_single(recv[:nrecv * W], send[:nsend * W], output_split_sizes=[int(c) * W for c in recv_counts],
                           input_split_sizes=[int(c) * W for c in send_counts])
    return recv, recv_counts


def _worker(rank, world, port, k, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from genome_amd import _lib, synth
    from oracle import oracle as O
    W = 1 if k <= 32 else 2
    n, L_ = 300, 90
    rec = synth.reads_mode_g(n, L_, 2500, 0.01, config_id=77, first_read=rank * n)   # weak scaling: own reads
    # extract + canonicalise + bucket by owner (what the routing kernels do on the device, restated with the oracle)
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
    recv, rcounts = _all_to_all_keys(send, counts, W)
    got = recv[:int(rcounts.sum()) * W].numpy().view(np.uint64).reshape(-1, W)
    # owner-side insert
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
def test_two_rank_owner_routing_equals_single_table(tmp_path, k):
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


def _worker_id(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from genome_amd.dist import share_id
    # (gk_dist_unique_id needs RCCL and a GPU: the id's source is a parameter, its way to the other ranks is what runs here)
    got = share_id(rank, world, make_id=lambda: bytes((7 * i + 1) % 256 for i in range(128)))
    open(os.path.join(out_dir, f"id{rank}.bin"), "wb").write(got)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bootstrap_of_the_communicator_id(tmp_path):
    """bench.py's N > 1 bootstrap (genome_amd.dist.share_id): rank 0's 128 bytes reach every rank over gloo"""
    world, port = 2, 31500 + os.getpid() % 2000
    mp.start_processes(_worker_id, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    want = bytes((7 * i + 1) % 256 for i in range(128))
    for rank in range(world):
        assert open(tmp_path / f"id{rank}.bin", "rb").read() == want


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` with WORLD_SIZE unset starts N children (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one
    port for all) before it touches torch or HIP, forwards rank 0's line and returns the worst child's exit code
    (--launch-only: the children report their environment and exit)."""
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--launch-only"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["launch_only"] and out["n_gpus"] == 2 and out["rc"] == 0
    assert [x["rank"] for x in out["ranks"]] == [0, 1] == [x["local_rank"] for x in out["ranks"]]
    assert {x["world"] for x in out["ranks"]} == {2} and len({x["master"] for x in out["ranks"]}) == 1
    assert all(x["gpus_arg"] == 2 for x in out["ranks"])          # the children get the parent's arguments
    env.update(GK_BENCH_TEST_EXIT_RANK="1", GK_BENCH_TEST_EXIT="5")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-only"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 5 and "rank 1 exited with 5" in r.stderr
