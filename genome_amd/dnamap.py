"""Host-side mirror of `trait DNAMap[T]` for T = Int (S/ds/ArrayDNAMap.scala:49-60) over the HIP
library.  Method names and meanings follow the trait; the closure-taking forms exist only for the
closures the hot path actually passes (`_ + 1`, `(k, v) => v < rounds`), everything
else is expressed on exported arrays.  All compute happens on the GPU through the C-ABI.
"""
from __future__ import annotations

import ctypes as C
import sys
import json
import os

import numpy as np

from . import _lib as L
from . import dna


class Context:
    """One device + stream (replaces ActorsHome.system, S/scripts/ActorsHome.scala:20-30)."""

    def __init__(self, device: int = 0):
        self.h = L.vp()
        L.check(L.lib().gk_ctx_create(device, C.byref(self.h)))
        self.device = device

    def close(self):
        if self.h:
            L.lib().gk_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        if sys.is_finalizing():      # process exit: the HIP runtime may already be gone, and frees everything anyway
            return
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name: str, value: int):
        """Test / A-B switches (include/genome_amd_test.h: gk_ctx_set_option; the TEST build of the library only)."""
        L.check(L.test_hook("gk_ctx_set_option")(self.h, name.encode(), int(value)), self.h)

    def sync(self):
        """Wait for everything queued on the context's stream."""
        L.check(L.lib().gk_ctx_sync(self.h), self.h)

    def trim(self):
        """Give the device buffers parked in the context's pool back to the device (gk_ctx_trim)."""
        L.check(L.lib().gk_ctx_trim(self.h), self.h)

    def mem_stats(self, reset_peak: bool = False) -> dict:
        """Device bytes this context holds in blocks of 1 MiB and more: live, their high-water mark, parked in the pool."""
        live, peak, parked = C.c_uint64(), C.c_uint64(), C.c_uint64()
        L.check(L.lib().gk_ctx_mem_stats(self.h, C.byref(live), C.byref(peak), C.byref(parked), 1 if reset_peak else 0), self.h)
        return {"live": live.value, "peak": peak.value, "parked": parked.value}

    def set_mem_budget(self, nbytes: int):
        """Plan as if the device had `nbytes` of memory (0: all of it)."""
        L.check(L.lib().gk_ctx_set_mem_budget(self.h, int(nbytes)), self.h)

    def host_alloc(self, nbytes: int) -> np.ndarray:
        """Page-locked host buffer as a uint8 array (gk_host_alloc); give it back with host_free(array)."""
        p = L.vp()
        L.check(L.lib().gk_host_alloc(self.h, nbytes, C.byref(p)), self.h)
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(nbytes, 1),))[:nbytes]
        return arr

    def host_free(self, arr: np.ndarray):
        L.check(L.lib().gk_host_free(self.h, arr.ctypes.data), self.h)

    def alloc(self, nbytes: int) -> int:
        p = L.vp()
        L.check(L.lib().gk_dev_alloc(self.h, nbytes, C.byref(p)), self.h)
        return p.value

    def free(self, dptr: int):
        L.check(L.lib().gk_dev_free(self.h, dptr), self.h)

    def upload(self, dptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        L.check(L.lib().gk_dev_upload(self.h, dptr, arr.ctypes.data, arr.nbytes), self.h)

    def download(self, dptr: int, nbytes: int) -> np.ndarray:
        out = np.empty(nbytes, np.uint8)
        L.check(L.lib().gk_dev_download(self.h, out.ctypes.data, dptr, nbytes), self.h)
        return out

    def stream_bench(self, nbytes: int = 1 << 30, reps: int = 10) -> dict:
        """What this card streams (GB/s): copy = bytes read + written, fill = written, sum = read (gk_dev_stream_bench)."""
        out = (C.c_double * 3)()
        L.check(L.lib().gk_dev_stream_bench(self.h, nbytes, reps, out), self.h)
        return {"copy_GBps": out[0], "fill_GBps": out[1], "sum_GBps": out[2], "buffer_bytes": nbytes, "reps": reps}

    def synth_reads(self, dptr: int, nreads: int, read_len: int, mode: str = "U", config_id: int = 0,
                    first_read: int = 0, genome_len: int = 0, err: float = 0.0):
        L.check(L.lib().gk_synth_reads_dev(self.h, dptr, nreads, read_len, 0 if mode == "U" else 1, config_id,
                                           first_read, genome_len, int(err * (1 << 24))), self.h)

    def shard_superkmers(self, k: int, d_records: int, nreads: int, read_len: int, P: int, d_out: int, out_cap_records: int):
        """-> (records per owner, k-mers per owner); raises GkError(GK_E_CAPACITY) if d_out is too small
        (the error message carries the number of records needed)."""
        rc_, kc = np.zeros(P, np.uint64), np.zeros(P, np.uint64)
        L.check(L.lib().gk_shard_superkmers_dev(self.h, k, d_records, nreads, read_len, P, d_out, out_cap_records,
                                                L.ptr(rc_, C.c_uint64), L.ptr(kc, C.c_uint64)), self.h)
        return rc_, kc

    def shard_reads(self, k: int, d_records: int, nreads: int, read_len: int, P: int, d_keys_out: int, keys_cap: int):
        counts = np.zeros(P, np.uint64)
        L.check(L.lib().gk_shard_reads_dev(self.h, k, d_records, nreads, read_len, P, d_keys_out, keys_cap,
                                           L.ptr(counts, C.c_uint64)), self.h)
        return counts


def skm_slot_bytes(k: int) -> int:
    return L.lib().gk_skm_slot_bytes(k)


def _keys(k: int, keys):
    """Accept base strings, (lo, hi) pairs or a pair of uint64 arrays."""
    if isinstance(keys, tuple) and len(keys) == 2 and isinstance(keys[0], np.ndarray):
        return L.as_u64(keys[0]), L.as_u64(keys[1])
    keys = list(keys)
    if keys and isinstance(keys[0], str):
        for s in keys:
            if len(s) != k:        # assert(key.length == k)  ArrayDNAMap.scala:182
                raise L.KeyLengthError(L.GK_E_KLEN, f"key length {len(s)} != k={k}")
        return dna.pack_many(keys)
    lo = np.array([a for a, _ in keys], np.uint64)
    hi = np.array([b for _, b in keys], np.uint64)
    return lo, hi


class HipDNAMap:
    """`ArrayDNAMap[Int]` resident in HBM (one partition)."""

    def __init__(self, ctx: Context, k: int, capacity_hint: int = 0, for_graph: bool = False):
        """for_graph: the map is filled once with `capacity_hint` keys and then read by buildGraph (gk_map_create_for_graph)."""
        self.ctx, self.k = ctx, k
        self.h = L.vp()
        create = L.lib().gk_map_create_for_graph if for_graph else L.lib().gk_map_create
        L.check(create(ctx.h, k, capacity_hint, C.byref(self.h)), ctx.h)
        self._apply_env()

    @classmethod
    def adopt(cls, ctx: "Context", k: int, handle) -> "HipDNAMap":
        """wrap a gk_map the library created (gk_dist_gather_map): the wrapper owns it from here on"""
        m = cls.__new__(cls)
        m.ctx, m.k, m.h = ctx, k, handle
        m._apply_env()
        return m

    def _apply_env(self):
        forced = os.environ.get("GENOME_AMD_INSERT_PATH")     # test hook: "direct" | "partitioned"
        if forced:
            self.set_insert_path(forced)

    def set_insert_path(self, path):
        """'auto' | 'direct' (global atomics) | 'partitioned' (LDS segment build): same results."""
        code = {"auto": 0, "direct": 1, "partitioned": 2}.get(path, path)
        L.check(L.lib().gk_map_set_insert_path(self.h, int(code)), self.ctx.h)

    def close(self):
        if self.h:
            if self.ctx.h:               # (a handle that outlived its context — a failed test's traceback — is dropped, not followed)
                L.lib().gk_map_destroy(self.h)
            self.h = None

    def __del__(self):
        if sys.is_finalizing():      # process exit: the HIP runtime may already be gone, and frees everything anyway
            return
        try:
            self.close()
        except Exception:
            pass

    # ---- trait DNAMap[Int] -------------------------------------------------------------
    def size(self) -> int:                                   # :50
        n = C.c_uint64()
        L.check(L.lib().gk_map_size(self.h, C.byref(n)), self.ctx.h)
        return n.value

    def apply_batch(self, keys) -> np.ndarray:               # :51 apply, batched; -1 = None
        lo, hi = _keys(self.k, keys)
        out = np.empty(len(lo), np.int32)
        L.check(L.lib().gk_map_get_batch(self.h, L.ptr(lo, C.c_uint64), L.ptr(hi, C.c_uint64), len(lo),
                                         L.ptr(out, C.c_int32), None), self.ctx.h)
        return out

    def apply(self, key):                                    # :51
        v = int(self.apply_batch([key])[0])
        return None if v < 0 else v

    def contains(self, key) -> bool:                         # :57
        return self.apply(key) is not None

    def update_inc(self, keys):                              # :54 update(key, 1, _ + 1), batched
        lo, hi = _keys(self.k, keys)
        L.check(L.lib().gk_map_update_inc(self.h, L.ptr(lo, C.c_uint64), L.ptr(hi, C.c_uint64), len(lo)), self.ctx.h)

    def add_counts(self, lo, hi, counts):                    # update(key, c, _ + c): partition merge
        lo, hi = L.as_u64(lo), L.as_u64(hi)
        counts = np.ascontiguousarray(counts, np.int32)
        L.check(L.lib().gk_map_add_counts(self.h, L.ptr(lo, C.c_uint64), L.ptr(hi, C.c_uint64),
                                          L.ptr(counts, C.c_int32), len(lo)), self.ctx.h)

    def add_map(self, other: "HipDNAMap"):                   # update(key, c, _ + c) for every entry of `other`, device to device
        L.check(L.lib().gk_map_add_map(self.h, other.h), self.ctx.h)

    def update_inc_dev(self, d_keys: int, n: int):
        L.check(L.lib().gk_map_update_inc_dev(self.h, d_keys, n), self.ctx.h)

    def deleteAll_lt(self, rounds: int):                     # :56 deleteAll((k, v) => v < rounds)
        L.check(L.lib().gk_map_filter_lt(self.h, rounds), self.ctx.h)

    def items(self):                                         # :58 mapReduce(identity) / Container.iterator
        n = self.size()
        lo = np.empty(n, np.uint64)
        hi = np.zeros(n, np.uint64)
        cnt = np.empty(n, np.int32)
        got = C.c_uint64()
        L.check(L.lib().gk_map_export(self.h, L.ptr(lo, C.c_uint64), L.ptr(hi, C.c_uint64), L.ptr(cnt, C.c_int32),
                                      n, C.byref(got)), self.ctx.h)
        return lo, hi, cnt

    def mapReduce(self, map_fn, reduce_fn):                  # :58 — closures run on the host over the export
        """DNAMap.mapReduce(map, reduce) (ArrayDNAMap.scala:234-241): `map_fn((key_str, count))` returns
        None or a value, `reduce_fn(list_of_values)` folds them.  The table comes back once
        (gk_map_export); the one closure the hot path passes (buildGraph's classify) is fused on
        the GPU instead — see genome_amd.graph.buildGraph."""
        lo, hi, cnt = self.items()
        vals = []
        for a, b, c in zip(lo.tolist(), hi.tolist(), cnt.tolist()):
            v = map_fn((dna.unpack(a, b, self.k), c))
            if v is not None:
                vals.append(v)
        return reduce_fn(vals)

    def foreach(self, f):                                    # :59
        self.mapReduce(lambda kv: (f(kv), None)[1], lambda _: None)

    def getAll(self, key):                                   # :52 — an Int map holds at most one value per key
        v = self.apply(key)
        return [] if v is None else [v]

    def sorted_items(self):
        """Canonical table serialisation: sorted by (hi, lo) unsigned."""
        lo, hi, cnt = self.items()
        order = np.lexsort((lo, hi))
        return lo[order], hi[order], cnt[order]

    # ---- FreqFilter.add over a read stream ----------------------------------------------
    def count_reads(self, bin_bytes, nreads: int) -> int:
        buf = np.frombuffer(bin_bytes, np.uint8) if not isinstance(bin_bytes, np.ndarray) else np.ascontiguousarray(bin_bytes, np.uint8).reshape(-1)
        occ = C.c_uint64()
        L.check(L.lib().gk_map_count_reads(self.h, L.ptr(buf, C.c_uint8), buf.size, nreads, C.byref(occ)), self.ctx.h)
        return occ.value

    def prefetch_reads(self, bin_bytes, nreads: int):
        """start the upload of a (pinned) `.bin` stream's head; the next count_reads of the same buffer finds it on the device"""
        buf = np.frombuffer(bin_bytes, np.uint8) if not isinstance(bin_bytes, np.ndarray) else np.ascontiguousarray(bin_bytes, np.uint8).reshape(-1)
        L.check(L.lib().gk_map_prefetch_reads(self.h, L.ptr(buf, C.c_uint8), buf.size, nreads), self.ctx.h)

    def count_reads_dev(self, d_records: int, nreads: int, read_len: int) -> int:
        occ = C.c_uint64()
        L.check(L.lib().gk_map_count_reads_dev(self.h, d_records, nreads, read_len, C.byref(occ)), self.ctx.h)
        return occ.value

    def count_superkmers_dev(self, d_records: int, nrecords: int, kmers_total: int) -> int:
        occ = C.c_uint64()
        L.check(L.lib().gk_map_count_superkmers_dev(self.h, d_records, nrecords, kmers_total, C.byref(occ)), self.ctx.h)
        return occ.value

    def clear(self):
        L.check(L.lib().gk_map_clear(self.h), self.ctx.h)

    def verify(self):
        """-> (live slots, bad slots, sum of counts): the table's invariants, checked on the device."""
        return self.verify_checksum()[:3]

    def verify_checksum(self):
        """-> (live slots, bad slots, sum of counts, order-independent checksum of the (key, count) set)."""
        a, b, c, d = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        L.check(L.lib().gk_map_verify(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)), self.ctx.h)
        return a.value, b.value, c.value, d.value

    def trim(self):
        """Release the scratch kept between calls (gk_map_trim)."""
        L.check(L.lib().gk_map_trim(self.h), self.ctx.h)

    def set_max_batch_keys(self, keys: int):
        L.check(L.lib().gk_map_set_max_batch_keys(self.h, int(keys)), self.ctx.h)

    def slots(self) -> int:
        n = C.c_uint64()
        L.check(L.lib().gk_map_slots(self.h, C.byref(n)), self.ctx.h)
        return n.value

    def stats(self) -> dict:
        buf = C.create_string_buffer(1024)
        L.check(L.lib().gk_map_stats(self.h, buf, 1024), self.ctx.h)
        return json.loads(buf.value.decode())

    def last_phase_ms(self):
        arr = (C.c_float * 5)()
        L.check(L.lib().gk_map_last_phase_ms(self.h, arr), self.ctx.h)
        return [float(x) for x in arr]

    def last_count_kernel(self):
        ms, occ = C.c_float(), C.c_uint64()
        L.check(L.lib().gk_map_last_count_kernel(self.h, C.byref(ms), C.byref(occ)), self.ctx.h)
        return ms.value, occ.value


POS_EDGE = 1 << 63


def pos_node(node_id: int) -> int:
    """NodeGraphPosition(id) (S/data/graph/GraphPosition.scala) as the map's 64-bit value"""
    return int(node_id)


def pos_edge(edge_id: int, dist: int) -> int:
    """EdgeGraphPosition(id, dist)"""
    return POS_EDGE | (int(edge_id) << 32) | int(dist)


def pos_decode(v: int):
    v = int(v)
    return ("E", (v >> 32) & 0x7fffffff, v & 0xffffffff) if v >> 63 else ("N", v & 0xffffffff, 0)


class HipValueMap:
    """`DNAMap[T]` for a 64-bit T (GraphPosition, Long) with the multimap half of the trait (ArrayDNAMap.scala:49-60):
    putNew, getAll, update(key, v), apply — over gk_vmap_*."""

    def __init__(self, ctx: Context, k: int, capacity_hint: int = 0):
        self.ctx, self.k = ctx, k
        self.h = L.vp()
        L.check(L.lib().gk_vmap_create(ctx.h, k, capacity_hint, C.byref(self.h)), ctx.h)

    def close(self):
        if self.h:
            if self.ctx.h:
                L.lib().gk_vmap_destroy(self.h)
            self.h = None

    def __del__(self):
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def size(self) -> int:                                   # :50
        n = C.c_uint64()
        L.check(L.lib().gk_vmap_size(self.h, C.byref(n)), self.ctx.h)
        return n.value

    def putNew_batch(self, keys, values):                    # :55
        lo, hi = _keys(self.k, keys)
        v = L.as_u64(values)
        assert len(v) == len(lo)
        L.check(L.lib().gk_vmap_put_new_batch(self.h, L.ptr(lo, C.c_uint64), L.ptr(hi, C.c_uint64), L.ptr(v, C.c_uint64), len(lo)), self.ctx.h)

    def putNew(self, key, value):
        self.putNew_batch([key], [value])

    def update_batch(self, keys, values):                    # :53 update(key, v)
        lo, hi = _keys(self.k, keys)
        v = L.as_u64(values)
        assert len(v) == len(lo)
        L.check(L.lib().gk_vmap_update_batch(self.h, L.ptr(lo, C.c_uint64), L.ptr(hi, C.c_uint64), L.ptr(v, C.c_uint64), len(lo)), self.ctx.h)

    def update(self, key, value):
        self.update_batch([key], [value])

    def getAll_batch(self, keys):                            # :52, batched -> list of uint64 arrays
        lo, hi = _keys(self.k, keys)
        n = len(lo)
        off = np.zeros(n + 1, np.uint64)
        tot = C.c_uint64()
        rc = L.lib().gk_vmap_get_all_batch(self.h, L.ptr(lo, C.c_uint64), L.ptr(hi, C.c_uint64), n, L.ptr(off, C.c_uint64), None, 0, C.byref(tot))
        if rc not in (L.GK_OK, L.GK_E_CAPACITY):
            L.check(rc, self.ctx.h)
        vals = np.zeros(max(tot.value, 1), np.uint64)
        if tot.value:
            L.check(L.lib().gk_vmap_get_all_batch(self.h, L.ptr(lo, C.c_uint64), L.ptr(hi, C.c_uint64), n, L.ptr(off, C.c_uint64),
                                                  L.ptr(vals, C.c_uint64), tot.value, C.byref(tot)), self.ctx.h)
        return [vals[int(off[i]):int(off[i + 1])] for i in range(n)]

    def getAll(self, key):
        return self.getAll_batch([key])[0].tolist()

    def apply_batch(self, keys):                             # :51 -> (values, found)
        lo, hi = _keys(self.k, keys)
        vals, found = np.zeros(len(lo), np.uint64), np.zeros(len(lo), np.uint8)
        L.check(L.lib().gk_vmap_get_batch(self.h, L.ptr(lo, C.c_uint64), L.ptr(hi, C.c_uint64), len(lo), L.ptr(vals, C.c_uint64), L.ptr(found, C.c_uint8)), self.ctx.h)
        return vals, found.astype(bool)

    def apply(self, key):
        v, f = self.apply_batch([key])
        return int(v[0]) if f[0] else None

    def contains(self, key) -> bool:
        return self.apply(key) is not None

    def items(self):
        n = self.size()
        lo, hi, val = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        got = C.c_uint64()
        L.check(L.lib().gk_vmap_export(self.h, L.ptr(lo, C.c_uint64), L.ptr(hi, C.c_uint64), L.ptr(val, C.c_uint64), n, C.byref(got)), self.ctx.h)
        return lo, hi, val

