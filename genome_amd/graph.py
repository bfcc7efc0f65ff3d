"""Host-side mirror of `object Graph` / `trait Graph` / `MapGraph` (S/data/graph/Graph.scala) over
the HIP library: buildGraph, simplifyGraph, removeBubbles, removeEdge, components+retain, and the
canonical serialisation used for parity.
"""
from __future__ import annotations

import ctypes as C
import sys

import numpy as np

from . import _lib as L
from . import dna
from .dnamap import HipDNAMap
from .partitioned import PartitionedDNAMap


class HipGraph:
    """A MapGraph resident in HBM."""

    def __init__(self, ctx, k: int, handle):
        self.ctx, self.k, self.h = ctx, k, handle

    def close(self):
        if self.h:
            if self.ctx.h:               # (see HipDNAMap.close)
                L.lib().gk_graph_destroy(self.h)
            self.h = None

    def __del__(self):
        if sys.is_finalizing():      # process exit: the HIP runtime may already be gone, and frees everything anyway
            return
        try:
            self.close()
        except Exception:
            pass

    def counts(self):
        n, e, ln = C.c_uint64(), C.c_uint64(), C.c_uint64()
        L.check(L.lib().gk_graph_counts(self.h, C.byref(n), C.byref(e), C.byref(ln)), self.ctx.h)
        return n.value, e.value, ln.value

    def simplifyGraph(self):                       # Graph.scala:211-230
        L.check(L.lib().gk_graph_simplify(self.h), self.ctx.h)

    def removeBubbles(self):                       # Graph.scala:125-149
        L.check(L.lib().gk_graph_remove_bubbles(self.h), self.ctx.h)

    def removeEdges(self, edges) -> int:           # Graph.scala:191-195, batched: [(start k-mer str, first base char)]
        lo, hi = dna.pack_many([s for s, _ in edges])
        base = np.array([dna.BASES.index(b) for _, b in edges], np.uint8)
        removed = C.c_uint64()
        L.check(L.lib().gk_graph_remove_edges(self.h, L.ptr(lo, C.c_uint64), L.ptr(hi, C.c_uint64), L.ptr(base, C.c_uint8),
                                              len(base), C.byref(removed)), self.ctx.h)
        return removed.value

    def retainLargest(self):                       # Graph.scala:54-72,161-165; GraphBuilder.scala:52-54
        kept, comps = C.c_uint64(), C.c_uint64()
        L.check(L.lib().gk_graph_retain_largest(self.h, C.byref(kept), C.byref(comps)), self.ctx.h)
        return kept.value, comps.value

    def componentStats(self):
        """-> (nodes per component, summed out-edge length per component), one entry per component."""
        n = C.c_uint64()
        rc = L.lib().gk_graph_component_stats(self.h, None, None, 0, C.byref(n))
        if rc not in (L.GK_OK, L.GK_E_CAPACITY):
            L.check(rc, self.ctx.h)
        nodes, length = np.zeros(n.value, np.uint32), np.zeros(n.value, np.uint64)
        if n.value:
            L.check(L.lib().gk_graph_component_stats(self.h, L.ptr(nodes, C.c_uint32), L.ptr(length, C.c_uint64), n.value, C.byref(n)), self.ctx.h)
        return nodes, length

    def componentHistograms(self):
        """GraphBuilder.scala:41-47: `hist` = components by node count, `hist2` = components by summed out-edge length,
        each as a sorted list of (value, number of components)."""
        nodes, length = self.componentStats()
        h1 = sorted((int(a), int(b)) for a, b in zip(*np.unique(nodes, return_counts=True)))
        h2 = sorted((int(a), int(b)) for a, b in zip(*np.unique(length, return_counts=True)))
        return h1, h2

    def checksum(self):
        a, b = C.c_uint64(), C.c_uint64()
        L.check(L.lib().gk_graph_checksum(self.h, C.byref(a), C.byref(b)), self.ctx.h)
        return a.value, b.value

    def buildStats(self):
        ms = (C.c_float * 6)()
        bases, pj = C.c_uint64(), C.c_int()
        L.check(L.lib().gk_graph_build_stats(self.h, ms, C.byref(bases), C.byref(pj)), self.ctx.h)
        names = ("classify", "make_nodes", "unitig_measure", "reserve_pool", "unitig_emit", "index_counts")
        mb_ms, mb_slots = C.c_float(), C.c_uint64()
        L.check(L.lib().gk_graph_bucketed_table_stats(self.h, C.byref(mb_ms), C.byref(mb_slots)), self.ctx.h)
        by_owners = C.c_int()
        L.check(L.lib().gk_graph_classified_by_owners(self.h, C.byref(by_owners)), self.ctx.h)
        return {"phase_ms": {n_: float(x) for n_, x in zip(names, ms)}, "walked_bases": bases.value, "pointer_jumping": bool(pj.value),
                "bucketed_table": {"build_ms": float(mb_ms.value), "slots": mb_slots.value}, "classified_by_owners": bool(by_owners.value)}

    # ---- GraphSimplifier's tools: the position map, ids, point edits ----------------------------
    def getGraphMap(self, capacity_hint: int = 0):         # Graph.scala:90-119
        from .dnamap import HipValueMap
        n, e, ln = self.counts()
        vm = HipValueMap(self.ctx, self.k, capacity_hint or (ln + n - e))
        got = C.c_uint64()
        L.check(L.lib().gk_graph_position_map(self.h, vm.h, C.byref(got)), self.ctx.h)
        assert got.value == ln + n - e                       # the reference prints both, :117
        return vm

    def nodeId(self, kmer: str, base=None):
        """(node id or None, id of its out-edge starting with `base` or None)"""
        lo, hi = dna.pack(kmer)
        a, b = C.c_uint32(), C.c_uint32()
        L.check(L.lib().gk_graph_node_lookup(self.h, lo, hi, -1 if base is None else dna.BASES.index(base), C.byref(a), C.byref(b)), self.ctx.h)
        none = 0xffffffff
        return (None if a.value == none else a.value), (None if b.value == none else b.value)

    def nodesById(self, ids):
        ids = np.ascontiguousarray(ids, np.uint32)
        n = len(ids)
        lo, hi = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        alive, ind, outd = np.zeros(n, np.uint8), np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        L.check(L.lib().gk_graph_nodes_by_id(self.h, L.ptr(ids, C.c_uint32), n, L.ptr(lo, C.c_uint64), L.ptr(hi, C.c_uint64), L.ptr(alive, C.c_uint8),
                                             L.ptr(ind, C.c_uint32), L.ptr(outd, C.c_uint32)), self.ctx.h)
        return {"lo": lo, "hi": hi, "alive": alive.astype(bool), "in_deg": ind, "out_deg": outd}

    def edgesById(self, ids):
        ids = np.ascontiguousarray(ids, np.uint32)
        n = len(ids)
        s, e = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        ln, first, alive = np.zeros(n, np.uint64), np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        L.check(L.lib().gk_graph_edges_by_id(self.h, L.ptr(ids, C.c_uint32), n, L.ptr(s, C.c_uint32), L.ptr(e, C.c_uint32), L.ptr(ln, C.c_uint64),
                                             L.ptr(first, C.c_uint8), L.ptr(alive, C.c_uint8)), self.ctx.h)
        return {"start": s, "end": e, "len": ln, "first": first, "alive": alive.astype(bool)}

    def addNode(self, kmer: str) -> int:                   # Graph.scala:172-176
        lo, hi = dna.pack(kmer)
        out = C.c_uint32()
        L.check(L.lib().gk_graph_add_node(self.h, lo, hi, C.byref(out)), self.ctx.h)
        return out.value

    def replaceStart(self, edge_id: int, node_id: int):    # :197-202
        L.check(L.lib().gk_graph_replace_start(self.h, edge_id, node_id), self.ctx.h)

    def replaceEnd(self, edge_id: int, node_id: int):      # :204-209
        L.check(L.lib().gk_graph_replace_end(self.h, edge_id, node_id), self.ctx.h)

    # ---- paired-end walking (S/scripts/GraphSimplifier.scala:188-318)
    def idBounds(self):
        a, b = C.c_uint64(), C.c_uint64()
        L.check(L.lib().gk_graph_id_bounds(self.h, C.byref(a), C.byref(b)), self.ctx.h)
        return a.value, b.value

    def removeEdgesById(self, edge_ids) -> int:             # toRemove.foreach(id => graph.removeEdge(graph.getEdge(id)))  :316
        ids = np.ascontiguousarray(edge_ids, np.uint32)
        rm = C.c_uint64()
        L.check(L.lib().gk_graph_remove_edges_by_id(self.h, L.ptr(ids, C.c_uint32), len(ids), C.byref(rm)), self.ctx.h)
        return rm.value

    def walkPairs(self, positions, support: "Support", bin_bytes, npairs: int, range_lo: int = 180, range_hi: int = 250):
        """:213-247: the pairs' positions through `positions` (getGraphMap of this graph as it is now), annotate, the bounded
        walks; the supported (edge, edge) pairs are counted in `support`."""
        buf = np.frombuffer(bin_bytes, np.uint8) if not isinstance(bin_bytes, np.ndarray) else np.ascontiguousarray(bin_bytes, np.uint8).reshape(-1)
        L.check(L.lib().gk_graph_walk_pairs(self.h, positions.h, support.h, L.ptr(buf, C.c_uint8), buf.size, npairs, range_lo, range_hi), self.ctx.h)

    def splitBySupport(self, support: "Support", cutoff: int):
        """:272-316 -> (edges removed, nodes added); call simplifyGraph() next (:318)."""
        rm, nn = C.c_uint64(), C.c_uint64()
        L.check(L.lib().gk_graph_split_by_support(self.h, support.h, cutoff, C.byref(rm), C.byref(nn)), self.ctx.h)
        return rm.value, nn.value

    def getNodes(self):
        n = self.counts()[0]
        lo, hi = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        got = C.c_uint64()
        L.check(L.lib().gk_graph_export_nodes(self.h, L.ptr(lo, C.c_uint64), L.ptr(hi, C.c_uint64), n, C.byref(got)), self.ctx.h)
        return lo, hi

    def getEdges(self):
        _, ne, ln = self.counts()
        a = {key: np.zeros(ne, np.uint64) for key in ("slo", "shi", "elo", "ehi")}
        length, off = np.zeros(ne, np.int64), np.zeros(ne, np.int64)
        cap = (ln + 3 * ne) // 4 + 1
        seq = np.zeros(cap, np.uint8)
        got, used = C.c_uint64(), C.c_uint64()
        L.check(L.lib().gk_graph_export_edges(self.h, L.ptr(a["slo"], C.c_uint64), L.ptr(a["shi"], C.c_uint64),
                                              L.ptr(a["elo"], C.c_uint64), L.ptr(a["ehi"], C.c_uint64),
                                              L.ptr(length, C.c_int64), L.ptr(off, C.c_int64), ne, C.byref(got),
                                              L.ptr(seq, C.c_uint8), cap, C.byref(used)), self.ctx.h)
        a.update(len=length, off=off, seq=seq[:used.value])
        return a

    def out_order(self, kmer: str):
        lo, hi = dna.pack(kmer)
        arr, cnt = (C.c_int * 4)(), C.c_int()
        L.check(L.lib().gk_graph_out_order(self.h, lo, hi, arr, C.byref(cnt)), self.ctx.h)
        return None if cnt.value < 0 else [arr[i] for i in range(cnt.value)]

    def canonical(self):
        """(sorted node strings, edges (start, end, seq) sorted by (start k-mer, first base))."""
        k = self.k
        lo, hi = self.getNodes()
        order = np.lexsort((lo, hi))
        nodes = [dna.unpack(int(lo[i]), int(hi[i]), k) for i in order]
        e = self.getEdges()
        first = np.array([(int(e["seq"][o]) & 3) if ln else 0 for o, ln in zip(e["off"], e["len"])], np.int64)
        order = np.lexsort((first, e["slo"], e["shi"]))
        edges = []
        for i in order:
            o, ln = int(e["off"][i]), int(e["len"][i])
            edges.append((dna.unpack(int(e["slo"][i]), int(e["shi"][i]), k),
                          dna.unpack(int(e["elo"][i]), int(e["ehi"][i]), k),
                          dna.unpack_2bit(e["seq"][o:o + (ln + 3) // 4], ln)))
        return nodes, edges


def buildGraph(k: int, kmersFreq) -> HipGraph:
    """Graph.buildGraph(k, kmersFreq) (Graph.scala:269-382).  A PartitionedDNAMap is gathered into
    one table first (the walk crosses partitions arbitrarily)."""
    own = None
    if isinstance(kmersFreq, PartitionedDNAMap):
        own = kmersFreq = kmersFreq.merged()
    assert isinstance(kmersFreq, HipDNAMap) and kmersFreq.k == k
    h = L.vp()
    try:
        L.check(L.lib().gk_graph_build(kmersFreq.h, C.byref(h)), kmersFreq.ctx.h)
    finally:
        if own is not None:
            own.close()
    return HipGraph(kmersFreq.ctx, k, h)


class Support:
    """pathsMap + badPairs of GraphSimplifier.scala:209-211: (edge id, edge id) -> read pairs whose walk passes through both."""

    def __init__(self, ctx):
        self.ctx = ctx
        h = L.vp()
        L.check(L.lib().gk_support_create(ctx.h, C.byref(h)), ctx.h)
        self.h = h

    def close(self):
        if self.h:
            if self.ctx.h:               # (see HipDNAMap.close)
                L.lib().gk_support_destroy(self.h)
            self.h = None

    def __del__(self):
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def sizes(self):
        """-> (edge pairs, bad pairs, pair orientations walked)"""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        L.check(L.lib().gk_support_size(self.h, C.byref(a), C.byref(b), C.byref(c)), self.ctx.h)
        return a.value, b.value, c.value

    def last_ms(self):
        """wall ms of the last walkPairs into this support: keys, getAll batch, snapshot + checks, walks, merge"""
        arr = (C.c_float * 5)()
        L.check(L.lib().gk_support_last_ms(self.h, arr), self.ctx.h)
        return dict(zip(("keys", "lookup", "snapshot", "walks", "merge"), (float(x) for x in arr)))

    def items(self):
        n = self.sizes()[0]
        e1, e2, cnt = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        got = C.c_uint64()
        L.check(L.lib().gk_support_export(self.h, L.ptr(e1, C.c_uint32), L.ptr(e2, C.c_uint32), L.ptr(cnt, C.c_uint32), n, C.byref(got)), self.ctx.h)
        return e1, e2, cnt

