"""PartitionedDNAMap[Int] over the GPUs of one node (S/ds/PartitionedDNAMap.scala:15-64), one rank per GPU, behind the
C-ABI's gk_dist_* entry points (RCCL over xGMI inside the library — no torch on the data path).

The Python side only carries the 128-byte RCCL id from rank 0 to the others (any channel will do: bench.py uses
torch.distributed's broadcast, a C++ host a file or MPI, the Scala driver its Akka channel)."""
from __future__ import annotations

import ctypes as C
import sys

import numpy as np

from . import _lib as L
from .dnamap import Context, HipDNAMap


def unique_id() -> bytes:
    """rank 0: a fresh RCCL id (128 bytes) to hand to every rank"""
    buf = C.create_string_buffer(128)
    L.check(L.lib().gk_dist_unique_id(buf))
    return buf.raw


def share_id(rank: int, world: int, make_id=unique_id, backend: str = "gloo") -> bytes:
    """Bootstrap over torch.distributed (what bench.py does; a JVM host uses its own channel): rank 0 makes the 128-byte id,
    everybody gets it.  RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT come from the launcher; the process group is created
    here if there is none.  Only the id travels this way — the data path is RCCL inside the library."""
    import torch
    import torch.distributed as td
    if not td.is_initialized():
        td.init_process_group(backend, rank=rank, world_size=world)
    idt = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        raw = make_id()
        assert len(raw) == 128
        idt = torch.frombuffer(bytearray(raw), dtype=torch.uint8).clone()
    td.broadcast(idt, 0)
    return bytes(idt.numpy().tobytes())


class HipDist:
    """One rank of the communicator (gk_dist_create is collective: every rank calls it with the same id)."""

    def __init__(self, ctx: Context, rank: int, world: int, id128: bytes, loopback: bool = False):
        """loopback: the test transport (gk_dist_create_loopback) — the ranks are threads of this process on one device."""
        assert len(id128) == 128
        self.ctx, self.rank, self.world = ctx, rank, world
        self.h = L.vp()
        create = L.test_hook("gk_dist_create_loopback") if loopback else L.lib().gk_dist_create
        L.check(create(ctx.h, rank, world, C.create_string_buffer(id128, 128), C.byref(self.h)), ctx.h)

    def close(self):
        if self.h:
            if self.ctx.h:               # (a handle that outlived its context is dropped, not followed: gk_dist_destroy uses the context)
                L.lib().gk_dist_destroy(self.h)
            self.h = None

    def __del__(self):
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def barrier(self):
        L.check(L.lib().gk_dist_barrier(self.h), self.ctx.h)

    def allreduce(self, values, op: str = "sum") -> np.ndarray:
        v = np.ascontiguousarray(values, np.float64).copy()
        L.check(L.lib().gk_dist_allreduce_f64(self.h, v.ctypes.data_as(C.POINTER(C.c_double)), len(v), 1 if op == "max" else 0), self.ctx.h)
        return v

    def last_ms(self):
        ms = (C.c_float * 4)()
        L.check(L.lib().gk_dist_last_ms(self.h, ms), self.ctx.h)
        return dict(zip(("route_wait", "exchange", "owner_count", "total"), (float(x) for x in ms)))


class DistDNAMap:
    """`PartitionedDNAMap[Int]` with one partition per rank: this object is ONE rank's view (its partition + the communicator)."""

    def __init__(self, dist: HipDist, k: int, capacity_hint: int = 0):
        self.dist, self.ctx, self.k = dist, dist.ctx, k
        self.local = HipDNAMap(dist.ctx, k, capacity_hint)

    def close(self):
        self.local.close()

    def count_reads_dev(self, d_records: int, nreads: int, read_len: int):
        """FreqFilter.add over this rank's reads -> (windows sent, windows this rank counted as owner)."""
        sent, owned = C.c_uint64(), C.c_uint64()
        L.check(L.lib().gk_dist_count_reads_dev(self.dist.h, self.local.h, d_records, nreads, read_len, C.byref(sent), C.byref(owned)), self.ctx.h)
        return sent.value, owned.value

    def route_begin(self, d_records: int, nreads: int, read_len: int):
        """First half of count_reads_dev, asynchronous: route the batch on the second stream.  Call it for batch i+1 before
        count_routed() of batch i and the routing kernel hides behind the owner pipeline; with batch i+2 begun as well (three
        is the limit), count_routed(i) also posts the exchange of batch i+1 beside the count of batch i."""
        L.check(L.lib().gk_dist_route_begin(self.dist.h, self.k, d_records, nreads, read_len), self.ctx.h)

    def count_routed(self):
        """Second half: exchange and count the batch route_begin started -> (windows sent, windows counted as owner)."""
        sent, owned = C.c_uint64(), C.c_uint64()
        L.check(L.lib().gk_dist_count_routed(self.dist.h, self.local.h, C.byref(sent), C.byref(owned)), self.ctx.h)
        return sent.value, owned.value

    def size(self) -> int:                                    # PartitionedDNAMap.scala:31
        n = C.c_uint64()
        L.check(L.lib().gk_dist_size(self.dist.h, self.local.h, C.byref(n)), self.ctx.h)
        return n.value

    def deleteAll_lt(self, rounds: int):                      # :49-51 — local on every rank
        self.local.deleteAll_lt(rounds)

    def gathered(self, classified: bool = False) -> HipDNAMap:
        """every partition's survivors in one table on this rank (what Graph.buildGraph needs).  classified: the keys' owners
        classify them first (Graph.scala:320-329 on every partition, PartitionedDNAMap.scala:55-58: local lookups + one query
        all-to-all) and the degree masks travel with the keys — buildGraph on the result then skips its neighbour lookups."""
        h = L.vp()
        f = L.lib().gk_dist_gather_classified_map if classified else L.lib().gk_dist_gather_map
        L.check(f(self.dist.h, self.local.h, C.byref(h)), self.ctx.h)
        return HipDNAMap.adopt(self.ctx, self.k, h)

    def classify_queries(self) -> int:
        """neighbour lookups this rank has asked of other ranks in classified gathers"""
        n = C.c_uint64()
        L.check(L.lib().gk_dist_classify_queries(self.dist.h, C.byref(n)), self.ctx.h)
        return n.value
