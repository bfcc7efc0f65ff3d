"""Can RCCL run two ranks on ONE GPU?  If it can, this rehearses the N = 2 exchange (gk_dist_*, three batches deep, the
exchange of the next batch on the helper thread) on the one-GPU box: every rank counts its own reads, the partitions' total
and the gathered table must equal one map over all the reads.  Launch:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 scripts/dist_two_ranks_one_gpu.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from genome_amd import synth
from genome_amd.dnamap import Context, HipDNAMap
from genome_amd.dist import DistDNAMap, HipDist, unique_id

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
ctx = Context(0)                                    # every rank on GPU 0
idt = torch.zeros(128, dtype=torch.uint8)
if rank == 0:
    idt = torch.frombuffer(bytearray(unique_id()), dtype=torch.uint8).clone()
dist.broadcast(idt, 0)
try:
    hd = HipDist(ctx, rank, world, bytes(idt.numpy().tobytes()))
except Exception as e:
    print(f"rank {rank}: gk_dist_create failed: {e}", flush=True)
    sys.exit(3)
k, L, n, nb = 31, 150, 200000, 5
stride = synth.record_stride(L)
d = ctx.alloc(n * stride + 64)
ctx.synth_reads(d, n, L, "G", 5, rank * n, 2_000_000, 0.01)
pm = DistDNAMap(hd, k, 1 << 12)
per = n // nb
begun = sent = owned = 0
for i in range(nb):
    while begun < min(i + 3, nb):
        pm.route_begin(d + begun * per * stride, per, L)
        begun += 1
    s_, o_ = pm.count_routed()
    sent += s_; owned += o_
tot = hd.allreduce([float(sent), float(owned)], "sum")
size = pm.size()
full = pm.gathered()
ok = True
if rank == 0:
    ref = HipDNAMap(ctx, k, 1 << 12)
    for r in range(world):
        ctx.synth_reads(d, n, L, "G", 5, r * n, 2_000_000, 0.01)
        ref.count_reads_dev(d, n, L)
    want = ref.verify_checksum()
    print("windows sent / owned over all ranks:", tot, "expected", world * n * (L - k + 1))
    print("size over partitions:", size, "one map:", ref.size())
    ok = tot[0] == tot[1] == world * n * (L - k + 1) and size == ref.size()
    if full is not None:
        got = full.verify_checksum()
        print("gathered table:", got, "one map:", want)
        ok = ok and got == want
    print("TWO RANKS ON ONE GPU:", "OK" if ok else "MISMATCH", flush=True)
hd.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
