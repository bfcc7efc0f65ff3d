"""PartitionedDNAMap[Int] (S/ds/PartitionedDNAMap.scala:15-64): P partitions, each a HipDNAMap.

Logical partitions on ONE device: the form a 1-GPU box can run, and the rehearsal of what one rank of the N-GPU run
(genome_amd.dist.DistDNAMap over gk_dist_*: RCCL inside the library) does with the records it owns.
Owner = strand-symmetric minimizer hash mod P (gk_owner_of) instead of `hashCode mod P`
(PartitionedDNAMap.scala:60-63): unobservable in results, keeps x and rc(x) together.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .dnamap import Context, HipDNAMap, _keys, skm_slot_bytes


def owner_of(k: int, lo: int, hi: int, P: int) -> int:
    return L.lib().gk_owner_of(k, int(lo), int(hi), P)


def split_counts_to_offsets(counts: np.ndarray) -> np.ndarray:
    off = np.zeros(len(counts) + 1, np.int64)
    np.cumsum(counts.astype(np.int64), out=off[1:])
    return off


class PartitionedDNAMap:
    """P logical partitions on one device."""

    def __init__(self, ctx: Context, k: int, P: int, capacity_hint: int = 0):
        self.ctx, self.k, self.P = ctx, k, P
        self.W = 1 if k <= 32 else 2
        self.parts = [HipDNAMap(ctx, k, max(1, capacity_hint // P)) for _ in range(P)]
        # P logical partitions share ONE device: let every partition give its insert scratch back after each batch (a rank
        # of the N-GPU run keeps it — it has one partition)
        self.trim_scratch = P > 1

    def close(self):
        for p in self.parts:
            p.close()

    def size(self) -> int:                                    # :31
        return sum(p.size() for p in self.parts)

    def _owner_batch(self, lo, hi):
        return np.array([owner_of(self.k, a, b, self.P) for a, b in zip(lo, hi)], np.int64)

    def apply_batch(self, keys) -> np.ndarray:                # :33, routed per key
        lo, hi = _keys(self.k, keys)
        own = self._owner_batch(lo, hi)
        out = np.empty(len(lo), np.int32)
        for p in range(self.P):
            idx = np.nonzero(own == p)[0]
            if len(idx):
                out[idx] = self.parts[p].apply_batch((lo[idx], hi[idx]))
        return out

    def apply(self, key):
        v = int(self.apply_batch([key])[0])
        return None if v < 0 else v

    def contains(self, key) -> bool:                          # :53
        return self.apply(key) is not None

    def deleteAll_lt(self, rounds: int):                      # :49-51
        for p in self.parts:
            p.deleteAll_lt(rounds)

    def count_reads_dev(self, d_records: int, nreads: int, read_len: int) -> int:
        """Route by super-k-mers (gk_shard_superkmers_dev), then every owner counts the records it got
        with the ordinary read pipeline — the single-device rehearsal of the RCCL exchange."""
        nk = max(0, read_len - self.k + 1)
        if nreads * nk == 0:
            return 0
        slot = skm_slot_bytes(self.k)
        d_out, region, recs, kmers = self._route(d_records, nreads, read_len)
        try:
            for p in range(self.P):                       # owner p's records start at slot p * region
                if recs[p]:
                    self.parts[p].count_superkmers_dev(d_out + p * region * slot, int(recs[p]), int(kmers[p]))
                    if self.trim_scratch:
                        self.parts[p].trim()
        finally:
            self.ctx.free(d_out)
        return int(kmers.sum())

    def _route(self, d_records: int, nreads: int, read_len: int):
        """gk_shard_superkmers_dev into a fresh buffer -> (d_out, region slots, records per owner, k-mers per owner)."""
        slot = skm_slot_bytes(self.k)
        # a run of same-owner windows is about half a minimizer window long: (k - m + 2) / 2 windows, m = 11
        nk = max(1, read_len - self.k + 1)
        per_read = max(2.0, nk / max(1.0, (self.k - 9) / 2.0)) * 1.5 + 1
        cap = max(1024 * self.P, int(nreads * per_read) // self.P * self.P)
        while True:
            d_out = self.ctx.alloc(cap * slot)
            try:
                recs, kmers = self.ctx.shard_superkmers(self.k, d_records, nreads, read_len, self.P, d_out, cap)
                return d_out, cap // self.P, recs, kmers
            except L.GkError as e:
                self.ctx.free(d_out)
                if e.code != L.GK_E_CAPACITY:
                    raise
                cap = cap * 4

    def count_reads_dev_prefiltered(self, chunks, read_len: int, expected_distinct: int, progress=None):
        """FreqFilter.add through the exact singleton pre-filter, PARTITIONED: what one rank of the N-GPU run does with the
        records it owns.  `chunks` yields (device pointer, reads) pairs (the buffer may be reused between them): every chunk is
        routed once and its super-k-mer records are KEPT in HBM (~2 bits/base: 62.5 M reads of 150 bp are 20-30 GB), because
        the filter needs two passes over what an owner receives — pass 1 over ALL of an owner's records, then pass 2.
        The records are `.bin` records at a fixed stride, so the pre-filter's ordinary device entry points take them as they
        are.  Returns (windows looked at, windows admitted).  expected_distinct: distinct k-mers over all partitions."""
        from .prefilter import HipPrefilter
        slot = skm_slot_bytes(self.k)
        rec_len = (slot - 1) * 4                     # a record slot holds up to this many bases: stride(rec_len) == slot
        kept = []
        looked = admitted = 0
        try:
            for d_records, nreads in chunks:
                if nreads * max(0, read_len - self.k + 1) == 0:
                    continue
                kept.append(self._route(d_records, nreads, read_len))
                if progress:
                    progress("routed", len(kept))
            for p in range(self.P):
                pf = HipPrefilter(self.ctx, self.k, max(1, expected_distinct // self.P))
                try:
                    for d_out, region, recs, kmers in kept:
                        if recs[p]:
                            pf.add_reads_dev(d_out + p * region * slot, int(recs[p]), rec_len)
                    for d_out, region, recs, kmers in kept:
                        if recs[p]:
                            _, adm = pf.count_reads_dev(self.parts[p], d_out + p * region * slot, int(recs[p]), rec_len)
                            admitted += adm
                            looked += int(kmers[p])
                finally:
                    pf.close()
                    if self.trim_scratch:
                        self.parts[p].trim()
                if progress:
                    progress("partition", p)
        finally:
            for d_out, _, _, _ in kept:
                self.ctx.free(d_out)
        return looked, admitted

    def foreign_keys(self) -> int:
        """live keys that sit in a partition other than the one gk_owner_of names (must be 0)."""
        tot = 0
        for p, part in enumerate(self.parts):
            n = C.c_uint64()
            L.check(L.lib().gk_map_count_foreign(part.h, self.P, p, C.byref(n)), self.ctx.h)
            tot += n.value
        return tot

    def verify(self):
        """-> (live, bad slots, sum of counts, checksum) over all partitions (gk_map_verify: checksums of partitions add up)."""
        tot = [0, 0, 0, 0]
        for part in self.parts:
            for i, v in enumerate(part.verify_checksum()):
                tot[i] += v
        tot[3] &= (1 << 64) - 1
        return tuple(tot)

    def count_reads_dev_keys(self, d_records: int, nreads: int, read_len: int) -> int:
        """extract + canonicalise + bucket 8/16-B keys by owner on the device, then owner-side inserts."""
        nk = max(0, read_len - self.k + 1)
        total = nreads * nk
        if total == 0:
            return 0
        d_keys = self.ctx.alloc(total * 8 * self.W)
        try:
            counts = self.ctx.shard_reads(self.k, d_records, nreads, read_len, self.P, d_keys, total)
            off = split_counts_to_offsets(counts)
            for p in range(self.P):
                if counts[p]:
                    self.parts[p].update_inc_dev(d_keys + int(off[p]) * 8 * self.W, int(counts[p]))
        finally:
            self.ctx.free(d_keys)
        return int(counts.sum())

    def count_reads(self, bin_bytes, nreads: int) -> int:
        """Host `.bin` stream.  Fixed-length streams take the device sharding path; ragged streams
        are grouped by length first (the record framing is one length byte per read)."""
        buf = np.frombuffer(bin_bytes, np.uint8) if not isinstance(bin_bytes, np.ndarray) else np.ascontiguousarray(bin_bytes, np.uint8).reshape(-1)
        by_len: dict[int, list[int]] = {}
        pos = 0
        for _ in range(nreads):
            if pos >= buf.size:
                raise L.GkError(L.GK_E_FORMAT, "truncated .bin stream")
            ln = int(buf[pos])
            rb = 1 + (ln + 3) // 4
            if pos + rb > buf.size:
                raise L.GkError(L.GK_E_FORMAT, "truncated .bin stream")
            by_len.setdefault(ln, []).append(pos)
            pos += rb
        occ = 0
        for ln, starts in by_len.items():
            if ln < self.k:
                continue
            rb = 1 + (ln + 3) // 4
            idx = (np.array(starts, np.int64)[:, None] + np.arange(rb)[None, :]).reshape(-1)
            rec = np.ascontiguousarray(buf[idx])
            d_rec = self.ctx.alloc(rec.size + 64)
            try:
                self.ctx.upload(d_rec, rec)
                occ += self.count_reads_dev(d_rec, len(starts), ln)
            finally:
                self.ctx.free(d_rec)
        return occ

    def items(self):
        parts = [p.items() for p in self.parts]
        return tuple(np.concatenate([x[i] for x in parts]) for i in range(3))

    def sorted_items(self):
        lo, hi, cnt = self.items()
        order = np.lexsort((lo, hi))
        return lo[order], hi[order], cnt[order]

    def merged(self) -> HipDNAMap:
        """Gather every partition into one table (what Graph.buildGraph needs; SURVEY.md §8e "all-gather the survivors"):
        device to device, partition by partition (gk_map_add_map — the one-device form of gk_dist_gather_map)."""
        m = HipDNAMap(self.ctx, self.k, self.size(), for_graph=True)
        for p in self.parts:
            m.add_map(p)
        return m
