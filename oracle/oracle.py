"""oracle.py — ctypes wrapper over the C oracle (oracle/libgk_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (genome_amd) never imports this module.  PARITY STATUS: parity unpinned
(see gk_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Kmer(C.Structure):
    _fields_ = [("lo", C.c_uint64), ("hi", C.c_uint64)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libgk_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("gk_oracle.c", "gk_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "libgk_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    L = C.CDLL(build())
    u64p = C.POINTER(C.c_uint64)
    i64p = C.POINTER(C.c_int64)
    i32p = C.POINTER(C.c_int32)
    u8p = C.POINTER(C.c_uint8)
    vp = C.c_void_p
    sig = {
        "gko_k_supported": (C.c_int, [C.c_int]),
        "gko_revcomp": (Kmer, [Kmer, C.c_int]),
        "gko_hash": (C.c_int32, [Kmer, C.c_int]),
        "gko_canon": (Kmer, [Kmer, C.c_int]),
        "gko_improve": (C.c_int32, [C.c_int32]),
        "gko_partition": (C.c_int, [Kmer, C.c_int, C.c_int]),
        "gko_prepend": (Kmer, [C.c_int, Kmer, C.c_int]),
        "gko_append": (Kmer, [Kmer, C.c_int, C.c_int]),
        "gko_kmer_from_packed": (Kmer, [u8p, C.c_int, C.c_int]),
        "gko_map_new": (vp, [C.c_int]),
        "gko_map_free": (None, [vp]),
        "gko_map_size": (C.c_int, [vp]),
        "gko_map_bins": (C.c_int, [vp]),
        "gko_map_rescales": (C.c_int, [vp]),
        "gko_map_update_inc": (None, [vp, Kmer]),
        "gko_map_update_set": (None, [vp, Kmer, C.c_int32]),
        "gko_map_put_new": (None, [vp, Kmer, C.c_int32]),
        "gko_map_get": (C.c_int, [vp, Kmer, i32p]),
        "gko_map_get_all": (C.c_int, [vp, Kmer, i32p, C.c_int]),
        "gko_map_delete_lt": (None, [vp, C.c_int32]),
        "gko_map_export": (C.c_size_t, [vp, u64p, u64p, i32p, C.c_size_t]),
        "gko_pmap_new": (vp, [C.c_int, C.c_int]),
        "gko_pmap_free": (None, [vp]),
        "gko_pmap_part": (vp, [vp, C.c_int]),
        "gko_pmap_size": (C.c_long, [vp]),
        "gko_pmap_contains": (C.c_int, [vp, Kmer]),
        "gko_pmap_get": (C.c_int, [vp, Kmer, i32p]),
        "gko_pmap_update_inc": (None, [vp, Kmer]),
        "gko_pmap_delete_lt": (None, [vp, C.c_int32]),
        "gko_pmap_export_sorted": (C.c_size_t, [vp, u64p, u64p, i32p, C.c_size_t]),
        "gko_count_reads": (C.c_long, [vp, u8p, C.c_size_t, C.c_uint64]),
        "gko_count_reads_mt": (C.c_long, [vp, u8p, C.c_size_t, C.c_uint64, C.c_int]),
        "gko_graph_build": (vp, [vp]),
        "gko_graph_free": (None, [vp]),
        "gko_graph_num_nodes": (C.c_long, [vp]),
        "gko_graph_num_edges": (C.c_long, [vp]),
        "gko_graph_total_edge_len": (C.c_long, [vp]),
        "gko_graph_simplify": (None, [vp]),
        "gko_graph_remove_bubbles": (None, [vp]),
        "gko_graph_remove_edge": (C.c_int, [vp, Kmer, C.c_int]),
        "gko_graph_retain_largest": (C.c_long, [vp]),
        "gko_graph_num_components": (C.c_long, [vp]),
        "gko_graph_export_nodes": (C.c_size_t, [vp, u64p, u64p, C.c_size_t]),
        "gko_graph_export_edges": (C.c_size_t, [vp, u64p, u64p, u64p, u64p, i64p, i64p, C.c_size_t,
                                                u8p, C.c_size_t, C.POINTER(C.c_size_t)]),
        "gko_graph_out_order": (C.c_int, [vp, Kmer, C.POINTER(C.c_int)]),
        "gko_graph_degree": (C.c_int, [vp, Kmer, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "gko_graph_find_node": (C.c_int64, [vp, Kmer]),
        "gko_graph_find_out_edge": (C.c_int64, [vp, C.c_int64, C.c_int]),
        "gko_graph_add_node": (C.c_int64, [vp, Kmer]),
        "gko_graph_replace_start": (None, [vp, C.c_int64, C.c_int64]),
        "gko_graph_replace_end": (None, [vp, C.c_int64, C.c_int64]),
        "gko_graph_get_graph_map": (C.c_size_t, [vp, u64p, u64p, u8p, i64p, i32p, C.c_size_t]),
        "gko_graph_node_seq": (Kmer, [vp, C.c_int64]),
        "gko_graph_edge_info": (C.c_int, [vp, C.c_int64, i64p, i64p, i64p, C.POINTER(C.c_int)]),
        "gko_support_new": (vp, []),
        "gko_support_free": (None, [vp]),
        "gko_support_bad_pairs": (C.c_long, [vp]),
        "gko_support_export": (C.c_size_t, [vp, i64p, i64p, i32p, C.c_size_t]),
        "gko_graph_walk_pairs": (C.c_long, [vp, vp, u8p, C.c_size_t, C.c_uint64, C.c_int, C.c_int]),
        "gko_graph_split_by_support": (None, [vp, vp, C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _LIB = L
    return L


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def km(lo: int, hi: int = 0) -> Kmer:
    return Kmer(lo & (2**64 - 1), hi & (2**64 - 1))


def revcomp(lo, hi, k):
    r = lib().gko_revcomp(km(lo, hi), k)
    return r.lo, r.hi


def hash_code(lo, hi, k):
    return lib().gko_hash(km(lo, hi), k)


def canon(lo, hi, k):
    r = lib().gko_canon(km(lo, hi), k)
    return r.lo, r.hi


def improve(h):
    return lib().gko_improve(h)


def partition(lo, hi, k, P):
    return lib().gko_partition(km(lo, hi), k, P)


class PMap:
    """PartitionedDNAMap[Int] oracle handle."""

    def __init__(self, k: int, P: int = 1):
        self.k, self.P = k, P
        self.h = lib().gko_pmap_new(k, P)

    def close(self):
        if self.h:
            lib().gko_pmap_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def count_reads(self, bin_bytes, nreads: int) -> int:
        buf = np.frombuffer(bin_bytes, dtype=np.uint8) if not isinstance(bin_bytes, np.ndarray) else bin_bytes
        buf = np.ascontiguousarray(buf)
        r = lib().gko_count_reads(self.h, _p(buf, C.c_uint8), buf.size, nreads)
        if r < 0:
            raise ValueError("truncated .bin stream")
        return r

    def count_reads_mt(self, bin_bytes, nreads: int, nthreads: int) -> int:
        buf = np.frombuffer(bin_bytes, dtype=np.uint8) if not isinstance(bin_bytes, np.ndarray) else bin_bytes
        buf = np.ascontiguousarray(buf)
        r = lib().gko_count_reads_mt(self.h, _p(buf, C.c_uint8), buf.size, nreads, nthreads)
        if r < 0:
            raise ValueError("truncated .bin stream")
        return r

    def update_inc(self, lo, hi=0):
        lib().gko_pmap_update_inc(self.h, km(lo, hi))

    def delete_lt(self, rounds: int):
        lib().gko_pmap_delete_lt(self.h, rounds)

    def size(self) -> int:
        return lib().gko_pmap_size(self.h)

    def get(self, lo, hi=0):
        v = C.c_int32(0)
        return v.value if lib().gko_pmap_get(self.h, km(lo, hi), C.byref(v)) else None

    def contains(self, lo, hi=0) -> bool:
        return bool(lib().gko_pmap_contains(self.h, km(lo, hi)))

    def part_stats(self, p):
        m = lib().gko_pmap_part(self.h, p)
        return lib().gko_map_size(m), lib().gko_map_bins(m), lib().gko_map_rescales(m)

    def export_sorted(self):
        n = self.size()
        lo = np.zeros(n, np.uint64)
        hi = np.zeros(n, np.uint64)
        cnt = np.zeros(n, np.int32)
        w = lib().gko_pmap_export_sorted(self.h, _p(lo, C.c_uint64), _p(hi, C.c_uint64), _p(cnt, C.c_int32), n)
        assert w == n
        return lo, hi, cnt


class Graph:
    """MapGraph oracle handle built by Graph.buildGraph."""

    def __init__(self, pmap: PMap):
        self.k = pmap.k
        self.h = lib().gko_graph_build(pmap.h)

    def close(self):
        if self.h:
            lib().gko_graph_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def num_nodes(self):
        return lib().gko_graph_num_nodes(self.h)

    def num_edges(self):
        return lib().gko_graph_num_edges(self.h)

    def total_edge_len(self):
        return lib().gko_graph_total_edge_len(self.h)

    def simplify(self):
        lib().gko_graph_simplify(self.h)

    def remove_bubbles(self):
        lib().gko_graph_remove_bubbles(self.h)

    def remove_edge(self, lo, hi, base) -> bool:
        return bool(lib().gko_graph_remove_edge(self.h, km(lo, hi), base))

    def retain_largest(self):
        return lib().gko_graph_retain_largest(self.h)

    def num_components(self):
        return lib().gko_graph_num_components(self.h)

    def nodes(self):
        n = self.num_nodes()
        lo = np.zeros(n, np.uint64)
        hi = np.zeros(n, np.uint64)
        lib().gko_graph_export_nodes(self.h, _p(lo, C.c_uint64), _p(hi, C.c_uint64), n)
        return lo, hi

    def edges(self):
        """Canonical edge list: dict of arrays + `bases` (one code per byte), sorted by (start, first base)."""
        n = self.num_edges()
        nb = self.total_edge_len()
        a = {key: np.zeros(n, np.uint64) for key in ("slo", "shi", "elo", "ehi")}
        ln = np.zeros(n, np.int64)
        off = np.zeros(n, np.int64)
        bases = np.zeros(max(nb, 1), np.uint8)
        used = C.c_size_t(0)
        lib().gko_graph_export_edges(self.h, _p(a["slo"], C.c_uint64), _p(a["shi"], C.c_uint64),
                                     _p(a["elo"], C.c_uint64), _p(a["ehi"], C.c_uint64),
                                     _p(ln, C.c_int64), _p(off, C.c_int64), n,
                                     _p(bases, C.c_uint8), nb, C.byref(used))
        a["len"] = ln
        a["off"] = off
        a["bases"] = bases[:nb]
        return a

    def out_order(self, lo, hi=0):
        arr = (C.c_int * 4)()
        n = lib().gko_graph_out_order(self.h, km(lo, hi), arr)
        return None if n < 0 else [arr[i] for i in range(n)]

    # ---- ids, point edits, Graph.getGraphMap
    def find_node(self, lo, hi=0) -> int:
        return lib().gko_graph_find_node(self.h, km(lo, hi))

    def find_out_edge(self, node: int, base: int) -> int:
        return lib().gko_graph_find_out_edge(self.h, node, base)

    def add_node(self, lo, hi=0) -> int:
        return lib().gko_graph_add_node(self.h, km(lo, hi))

    def replace_start(self, edge: int, node: int):
        lib().gko_graph_replace_start(self.h, edge, node)

    def replace_end(self, edge: int, node: int):
        lib().gko_graph_replace_end(self.h, edge, node)

    def node_seq(self, node: int):
        r = lib().gko_graph_node_seq(self.h, node)
        return r.lo, r.hi

    def edge_info(self, edge: int):
        s, e, ln, f = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int()
        alive = lib().gko_graph_edge_info(self.h, edge, C.byref(s), C.byref(e), C.byref(ln), C.byref(f))
        return {"start": s.value, "end": e.value, "len": ln.value, "first": f.value, "alive": bool(alive)}

    # ---- paired-end walking (GraphSimplifier.scala)
    def walk_pairs(self, sup: "Support", bin_bytes: bytes, npairs: int, range_lo: int = 180, range_hi: int = 250) -> int:
        buf = (C.c_uint8 * len(bin_bytes)).from_buffer_copy(bin_bytes)
        return lib().gko_graph_walk_pairs(self.h, sup.h, buf, len(bin_bytes), npairs, range_lo, range_hi)

    def split_by_support(self, sup: "Support", cutoff: int):
        rm, nn = C.c_long(0), C.c_long(0)
        lib().gko_graph_split_by_support(self.h, sup.h, cutoff, C.byref(rm), C.byref(nn))
        return rm.value, nn.value

    def graph_map_calls(self):
        """Graph.getGraphMap as its putNew sequence: arrays (lo, hi, is_edge, id, dist)."""
        n = lib().gko_graph_get_graph_map(self.h, None, None, None, None, None, 0)
        lo, hi = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        ie, ident, dist = np.zeros(n, np.uint8), np.zeros(n, np.int64), np.zeros(n, np.int32)
        lib().gko_graph_get_graph_map(self.h, _p(lo, C.c_uint64), _p(hi, C.c_uint64), _p(ie, C.c_uint8), _p(ident, C.c_int64), _p(dist, C.c_int32), n)
        return lo, hi, ie, ident, dist

    def degree(self, lo, hi=0):
        i, o = C.c_int(0), C.c_int(0)
        if lib().gko_graph_degree(self.h, km(lo, hi), C.byref(i), C.byref(o)) < 0:
            return None
        return i.value, o.value


class Support:
    """pathsMap + badPairs of GraphSimplifier.scala:209-211 (edge ids are the ORACLE graph's)."""

    def __init__(self):
        self.h = lib().gko_support_new()

    def close(self):
        if self.h:
            lib().gko_support_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def bad_pairs(self) -> int:
        return lib().gko_support_bad_pairs(self.h)

    def items(self):
        n = lib().gko_support_export(self.h, None, None, None, 0)
        e1, e2, c = np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros(n, np.int32)
        lib().gko_support_export(self.h, _p(e1, C.c_int64), _p(e2, C.c_int64), _p(c, C.c_int32), n)
        return e1, e2, c

