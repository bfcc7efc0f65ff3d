"""Exact two-pass singleton pre-filter: the GPU analogue of the reference's
unused Bloom filter (S/ds/BloomFilter.scala:17-70) in front of FreqFilter.add.  See
include/genome_amd.h for the exactness argument: for rounds >= 2 the table left by
`deleteAll(v < rounds)` is the same with and without the filter."""
from __future__ import annotations

import ctypes as C
import sys

import numpy as np

from . import _lib as L
from .dnamap import Context, HipDNAMap


class HipPrefilter:
    """2-bit saturating counters, 4 per expected distinct k-mer."""

    def __init__(self, ctx: Context, k: int, expected_distinct: int):
        self.ctx, self.k = ctx, k
        h = C.c_void_p()
        L.check(L.lib().gk_prefilter_create(ctx.h, k, int(expected_distinct), C.byref(h)), ctx.h)
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            L.lib().gk_prefilter_destroy(self.h)
            self.h = None

    def __del__(self):
        if sys.is_finalizing():      # process exit: the HIP runtime may already be gone, and frees everything anyway
            return
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _buf(bin_bytes):
        return np.frombuffer(bin_bytes, np.uint8) if not isinstance(bin_bytes, np.ndarray) else np.ascontiguousarray(bin_bytes, np.uint8).reshape(-1)

    # ---- pass 1
    def add_reads(self, bin_bytes, nreads: int):
        buf = self._buf(bin_bytes)
        L.check(L.lib().gk_prefilter_add_reads(self.h, L.ptr(buf, C.c_uint8), buf.size, nreads), self.ctx.h)

    def add_reads_dev(self, d_records: int, nreads: int, read_len: int):
        L.check(L.lib().gk_prefilter_add_reads_dev(self.h, d_records, nreads, read_len), self.ctx.h)

    # ---- pass 2: returns (windows looked at, windows inserted)
    def count_reads(self, m: HipDNAMap, bin_bytes, nreads: int):
        buf = self._buf(bin_bytes)
        occ, adm = C.c_uint64(), C.c_uint64()
        L.check(L.lib().gk_map_count_reads_prefiltered(m.h, self.h, L.ptr(buf, C.c_uint8), buf.size, nreads, C.byref(occ), C.byref(adm)), self.ctx.h)
        return occ.value, adm.value

    def count_reads_dev(self, m: HipDNAMap, d_records: int, nreads: int, read_len: int):
        occ, adm = C.c_uint64(), C.c_uint64()
        L.check(L.lib().gk_map_count_reads_prefiltered_dev(m.h, self.h, d_records, nreads, read_len, C.byref(occ), C.byref(adm)), self.ctx.h)
        return occ.value, adm.value

    def stats(self) -> dict:
        b, o, t, w = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        L.check(L.lib().gk_prefilter_stats(self.h, C.byref(b), C.byref(o), C.byref(t), C.byref(w)), self.ctx.h)
        return {"buckets": b.value, "seen_once": o.value, "seen_twice_or_more": t.value, "windows_added": w.value,
                "bytes": (b.value + 15) // 16 * 4}
