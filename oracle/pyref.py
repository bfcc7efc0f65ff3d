"""pyref.py — SECOND, INDEPENDENT CPU restatement (test infrastructure, NOT product code).

A deliberately naive, string-based, base-by-base Python restatement of the reference path, written
against the *generic* Scala collection semantics (sequences of bases, `sliding`, `take`, `drop`,
`:+`, `+:`, `map`, `reverse`) rather than the bit tricks the C oracle (gk_oracle.c) transliterates.
Its only purpose is to cross-check the C oracle on small inputs and to emit the golden fixtures
under tests/golden/ (tests/golden/make_golden.py).  Small cases only: pure-Python loops.

S/ = /root/reference/src/main/scala/ru/ifmo/genome/.   PARITY STATUS: parity unpinned — the
reference has no tests or fixtures and cannot run here.
"""
from __future__ import annotations

M64 = (1 << 64) - 1
BASES = "AGCT"                      # S/dna/Base.scala:13-18  A=0 G=1 C=2 T=3
COMP = {"A": "T", "T": "A", "G": "C", "C": "G"}   # Base.scala:19


def s64(v: int) -> int:
    v &= M64
    return v - (1 << 64) if v >> 63 else v


def s32(v: int) -> int:
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v >> 31 else v


def rev_comp(seq: str) -> str:
    """DNASeq.revComplement (DNASeq.scala:27-28): complement, then reverse."""
    return "".join(COMP[c] for c in seq)[::-1]


def pack(seq: str) -> tuple[int, int]:
    """DNASeq.newBuilder (DNASeq.scala:237-281): base i -> bits 2i of l1 (i<32) / 2(i-32) of l2."""
    l1 = l2 = 0
    for i, c in enumerate(seq):
        b = BASES.index(c)
        if i < 32:
            l1 |= b << (2 * i)
        else:
            l2 |= b << (2 * (i - 32))
    return l1, l2


def unpack(lo: int, hi: int, k: int) -> str:
    out = []
    for i in range(k):
        w = lo if i < 32 else hi
        out.append(BASES[(w >> (2 * (i % 32))) & 3])
    return "".join(out)


def scala291_long_hash(lv_signed: int) -> int:
    """`Long.##` in scala-library 2.9.1 (BoxesRunTime.hashFromLong) — third party, unpinned."""
    iv = s32(lv_signed)
    if iv == lv_signed:
        return iv
    u = lv_signed & M64
    return s32(u ^ (u >> 32))


def hash_code(seq: str) -> int:
    """Long1DNASeq.hashCode (DNASeq.scala:103) for len<=32; MultiHash.hashCode = multiHashCode(42)
    (BloomFilter.scala:12-15) with Long2DNASeq.multiHashCode (DNASeq.scala:204-208) for 33..64."""
    l1, l2 = pack(seq)
    if len(seq) <= 32:
        return scala291_long_hash(s64(l1))
    seed = 42
    a, b = s64(l1), s64(l2)
    t = s64((a ^ (a >> 32)) * seed)          # Python >> on negative ints is arithmetic, like Java >>
    t1 = s64((b ^ (b >> 32) ^ t) * seed)
    return s32(t1 ^ (t1 >> 32))


def canon(x: str) -> str:
    """FreqFilter.scala:31-32 — smaller signed hashCode wins, tie -> reverse complement."""
    rcx = rev_comp(x)
    return x if hash_code(x) < hash_code(rcx) else rcx


def improve(hcode: int) -> int:
    """ArrayDNAMap.improve (ArrayDNAMap.scala:267-272)."""
    hc = hcode & 0xFFFFFFFF
    h = (hc + (~(hc << 9) & 0xFFFFFFFF)) & 0xFFFFFFFF
    h ^= h >> 14
    h = (h + (h << 4)) & 0xFFFFFFFF
    return s32(h ^ (h >> 10))


def partition(key: str, P: int) -> int:
    """PartitionedDNAMap.partition (PartitionedDNAMap.scala:60-63), Java % semantics."""
    h = hash_code(key)
    r = abs(h) % P
    r = -r if h < 0 else r
    return r + P if r < 0 else r


class ArrayDNAMap:
    """ArrayDNAMap[Int] + Container (ArrayDNAMap.scala:62-243), literal incl. tombstones/rescale."""

    def __init__(self, k: int):
        self.k = k
        self.rescales = 0
        self._new_container(16)

    def _new_container(self, bins: int):
        self.bins = bins
        self.mask = bins - 1
        self.size = 0
        self.keys = [None] * bins
        self.ar = [0] * bins
        self.set = [False] * bins
        self.dele = [False] * bins

    def _start(self, key: str) -> int:
        return improve(hash_code(key)) & self.mask

    def _put_new_raw(self, key, v):            # Container.putNew :152-162
        i = self._start(key)
        while (not self.dele[i]) and self.set[i]:
            i = (i + 1) & self.mask
        self.set[i] = True
        self.dele[i] = False
        self.size += 1
        self.keys[i] = key
        self.ar[i] = v

    def rescale(self):                          # :217-230
        if (self.bins > 16 and self.size < self.bins * 0.3) or self.bins * 0.7 < self.size:
            new_bins = 16
            while new_bins * 0.7 < self.size:
                new_bins *= 2
            live = [(self.keys[i], self.ar[i]) for i in range(self.bins) if self.set[i] and not self.dele[i]]
            self._new_container(new_bins)
            for key, v in live:
                self._put_new_raw(key, v)
            self.rescales += 1

    def update_inc(self, key: str):             # Container.update(key, v0, f) :129-150, v0=1 f=_+1
        assert len(key) == self.k               # :199
        i = self._start(key)
        first = -1
        while self.set[i] and (self.dele[i] or self.keys[i] != key):
            if self.dele[i]:
                first = i
            i = (i + 1) & self.mask
        if not self.set[i]:
            if first != -1:
                i = first
                self.dele[i] = False
            self.set[i] = True
            self.keys[i] = key
            self.size += 1
            self.ar[i] = 1
        else:
            self.ar[i] = s32(self.ar[i] + 1)
        self.rescale()

    def put_new(self, key: str, v: int):
        assert len(key) == self.k
        self._put_new_raw(key, v)
        self.rescale()

    def get(self, key: str):                    # Container.apply :90-101
        assert len(key) == self.k               # :182
        i = self._start(key)
        while self.set[i]:
            if (not self.dele[i]) and self.keys[i] == key:
                return self.ar[i]
            i = (i + 1) & self.mask
        return None

    def get_all(self, key: str):                # Container.getAll :103-113
        ans = []
        i = self._start(key)
        while self.set[i]:
            if (not self.dele[i]) and self.keys[i] == key:
                ans.insert(0, self.ar[i])
            i = (i + 1) & self.mask
        return ans

    def delete_lt(self, rounds: int):           # deleteAll((k,v) => v < rounds) :164-173, :212-215
        for i in range(self.bins):
            if self.set[i] and not self.dele[i] and self.ar[i] < rounds:
                self.dele[i] = True
                self.size -= 1
        self.rescale()

    def items(self):                            # Container.iterator :175-178
        return [(self.keys[i], self.ar[i]) for i in range(self.bins) if self.set[i] and not self.dele[i]]


class PartitionedDNAMap:
    """PartitionedDNAMap[Int] (PartitionedDNAMap.scala:15-64) without the network."""

    def __init__(self, k: int, P: int = 1):
        self.k, self.P = k, P
        self.parts = [ArrayDNAMap(k) for _ in range(P)]

    def update_inc(self, key):
        self.parts[partition(key, self.P)].update_inc(key)

    def get(self, key):
        return self.parts[partition(key, self.P)].get(key)

    def contains(self, key) -> bool:
        return self.get(key) is not None

    def delete_lt(self, rounds):
        for p in self.parts:
            p.delete_lt(rounds)

    def size(self):
        return sum(p.size for p in self.parts)

    def items(self):
        out = []
        for p in self.parts:
            out.extend(p.items())
        return out

    def sorted_items(self):
        """Canonical table serialisation: (hi, lo) unsigned ascending."""
        def keyf(kv):
            lo, hi = pack(kv[0])
            return (hi, lo)
        return sorted(self.items(), key=keyf)


def reads_from_bin(bin_bytes: bytes, nreads: int):
    """PairedEndData.getPairs (PairedEndData.scala:20-36) record stream -> list of base strings."""
    pos = 0
    out = []
    for _ in range(nreads):
        ln = bin_bytes[pos]
        pos += 1
        nb = (ln + 3) // 4
        data = bin_bytes[pos:pos + nb]
        pos += nb
        out.append("".join(BASES[(data[i // 4] >> (2 * (i % 4))) & 3] for i in range(ln)))
    return out


def reads_to_bin(reads) -> bytes:
    """Convert2bin.write (Convert2bin.scala:35-38): [len:u8][toByteArray]."""
    out = bytearray()
    for r in reads:
        assert len(r) <= 255
        out.append(len(r))
        nb = (len(r) + 3) // 4
        data = bytearray(nb)
        for i, c in enumerate(r):
            data[i // 4] |= BASES.index(c) << (2 * (i % 4))
        out += data
    return bytes(out)


def extract_filtered_kmers(reads, k: int, rounds: int, P: int = 1, do_filter: bool = True):
    """FreqFilter.extractFilteredKmers (FreqFilter.scala:25-58)."""
    m = PartitionedDNAMap(k, P)
    for seq in reads:
        if len(seq) >= k:
            for p in range(len(seq) - k + 1):    # seq.sliding(k)
                m.update_inc(canon(seq[p:p + k]))
    if do_filter:
        m.delete_lt(rounds)
    return m


# ------------------------------------------------------------------ graph
class PyGraph:
    """MapGraph (Graph.scala:153-230) with ids in a deterministic order (ascending k-mer)."""

    def __init__(self, k):
        self.k = k
        self.nodes = {}      # id -> dict(seq, ins:set, outs:list[(base, edge id)])
        self.edges = {}      # id -> (start, end, seq)
        self.nid = 0
        self.eid = 0

    def add_node(self, seq):
        self.nid += 1
        self.nodes[self.nid] = {"seq": seq, "ins": set(), "outs": []}
        return self.nid

    def add_edge(self, start, end, seq):        # :178-184
        self.eid += 1
        outs = self.nodes[start]["outs"]
        for i, (b, _) in enumerate(outs):
            if b == seq[0]:
                outs[i] = (b, self.eid)
                break
        else:
            outs.append((seq[0], self.eid))
        self.nodes[end]["ins"].add(self.eid)
        self.edges[self.eid] = (start, end, seq)
        return self.eid

    def remove_edge(self, eid):                 # :191-195
        start, end, seq = self.edges[eid]
        if start in self.nodes:
            self.nodes[start]["outs"] = [(b, e) for (b, e) in self.nodes[start]["outs"] if b != seq[0]]
        if end in self.nodes:
            self.nodes[end]["ins"].discard(eid)
        del self.edges[eid]

    def simplify(self):                         # :211-230
        for nid in sorted(self.nodes):
            if nid not in self.nodes:
                continue
            n = self.nodes[nid]
            ins = list(n["ins"])
            outs = [e for (_, e) in n["outs"]]
            if not ins and not outs:
                del self.nodes[nid]
            elif len(ins) == 1 and len(outs) == 1:
                e1, e2 = ins[0], outs[0]
                if e1 == e2:
                    self.remove_edge(e1)
                else:
                    s1, _, q1 = self.edges[e1]
                    _, t2, q2 = self.edges[e2]
                    self.remove_edge(e1)
                    self.remove_edge(e2)
                    self.add_edge(s1, t2, q1 + q2)
                del self.nodes[nid]

    def remove_bubbles(self):                   # :125-149
        for nid in sorted(self.nodes):
            out = [e for (_, e) in self.nodes[nid]["outs"]]
            to_remove = []
            for i in range(len(out)):
                if out[i] in to_remove:
                    continue
                for j in range(i + 1, len(out)):
                    a, b = self.edges[out[i]], self.edges[out[j]]
                    la, lb = len(a[2]), len(b[2])
                    if a[1] == b[1] and abs(la - lb) * 5 < max(la, lb):
                        if out[j] not in to_remove:
                            to_remove.append(out[j])
            for e in to_remove:
                self.remove_edge(e)

    def canonical(self):
        def kk(seq):
            lo, hi = pack(seq)
            return (hi, lo)
        nodes = sorted((n["seq"] for n in self.nodes.values()), key=kk)
        edges = sorted(((self.nodes[s]["seq"], self.nodes[t]["seq"], q) for (s, t, q) in self.edges.values()),
                       key=lambda e: (kk(e[0]), BASES.index(e[2][0])))
        return nodes, edges


def build_graph(k: int, m: PartitionedDNAMap) -> PyGraph:
    """Graph.buildGraph (Graph.scala:269-382)."""
    def contains(x):
        return m.contains(x) or m.contains(rev_comp(x))          # :270

    def incoming(x):
        return [b for b in BASES if contains(b + x[:k - 1])]      # :272-276

    def outcoming(x):
        return [b for b in BASES if contains(x[1:] + b)]          # :278-282

    term = set()
    for read, _ in m.items():                                      # :320-329
        i, o = len(incoming(read)), len(outcoming(read))
        if (i != 1 or o != 1) and (i != 0 or o != 0):
            term.add(read)
    term |= {rev_comp(x) for x in term}                            # :330-333

    def kk(seq):
        lo, hi = pack(seq)
        return (hi, lo)

    g = PyGraph(k)
    node_map = {}
    for read in sorted(term, key=kk):                              # :343-347
        node_map[read] = g.add_node(read)
    for read in sorted(term, key=kk):                              # :349-374
        for base in outcoming(read):
            builder = base
            seq = read[1:] + base
            while seq not in node_map:
                out = outcoming(seq)
                assert len(out) == 1, (seq, out)                   # :357
                builder += out[0]
                seq = seq[1:] + out[0]
            g.add_edge(node_map[read], node_map[seq], builder)
    return g
