// gk_graph.hip — de Bruijn graph build and structural simplification as HIP kernels (gfx950).
//
// Reference path replaced (S/ = /root/reference/src/main/scala/ru/ifmo/genome/):
//   Graph.buildGraph      S/data/graph/Graph.scala:269-382   k_classify, k_collect_bits,
//                                                            k_make_nodes, k_walk
//   contains/incoming/outcoming          :270-282            gk::table_find_either (gk_device.h)
//   MapGraph.addNode/addEdge/removeEdge  :172-195            node/edge arrays below
//   MapGraph.simplifyGraph               :211-230            k_node_class, k_chain_*, k_simplify_finish
//   Graph.removeBubbles                  :125-149            k_bubbles
//   Graph.components + retain            :54-72, :161-165    k_cc_*, k_retain
//
// Layout in HBM: the k-mer table (array of 16/32-B slots) carries the degree annotation in its
// `aux` word (in-mask, out-mask, TERMINAL, SECONDARY) so one probe during a walk touches one
// sector.  The graph itself is structure-of-arrays: node k-mers, a 4-entry out-edge table per node
// (indexed by first base) plus its insertion order (the reference's immutable Map1..Map4 order),
// in-degree; edges as (start, end, length, byte offset) into one 2-bit-packed sequence pool.
// All kernels are integer, HBM/latency bound; no MFMA.
//
// Ids: node/edge ids are array indices in compaction order — arbitrary, like the reference's
// AtomicLong ids under `.par`; results are compared on canonical serialisations.
// Where the reference's result depends on node iteration order (out-edge insertion order after
// simplifyGraph), this file reproduces "ascending k-mer order", the oracle's deterministic choice.
#include <algorithm>
#include <chrono>
#include <memory>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "gk_internal.h"
#include "gk_tile.h"

using namespace gk;

#include "gk_graph.h"
#include "gk_scan.h"

template <int W> __device__ __forceinline__ Kmer<W> node_kmer(const GraphView &g, u64 n);
template <> __device__ __forceinline__ Kmer<1> node_kmer<1>(const GraphView &g, u64 n) { return Kmer<1>{g.node_lo[n]}; }
template <> __device__ __forceinline__ Kmer<2> node_kmer<2>(const GraphView &g, u64 n) { return Kmer<2>{g.node_lo[n], g.node_hi[n]}; }
__device__ __forceinline__ bool node_less(const GraphView &g, u32 a, u32 b) {   // unsigned (hi, lo) order
    u64 ah = g.node_hi[a], bh = g.node_hi[b];
    return ah != bh ? ah < bh : g.node_lo[a] < g.node_lo[b];
}

__device__ __forceinline__ u32 wave_incl_scan(u32 v) {
    const int lane = threadIdx.x & 63;
    for (int d = 1; d < 64; d <<= 1) {
        u32 t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}
// block-wide exclusive scan of small per-thread counts; *total = block sum
__device__ __forceinline__ u32 block_excl_scan(u32 v, u32 *total, u32 *lds4) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 inc = wave_incl_scan(v);
    __syncthreads();
    if (lane == 63) lds4[wave] = inc;
    __syncthreads();
    u32 base = 0, tot = 0;
    for (int w = 0; w < BLOCK / 64; ++w) {
        u32 c = lds4[w];
        if (w < wave) base += c;
        tot += c;
    }
    *total = tot;
    return base + inc - v;
}
// reserve `v` units per thread from a global cursor with ONE atomic per block; returns this
// thread's first unit
__device__ __forceinline__ u64 block_reserve(u32 v, unsigned long long *cursor, u32 *lds4, unsigned long long *s_base) {
    u32 tot;
    u32 pre = block_excl_scan(v, &tot, lds4);
    if (threadIdx.x == 0) *s_base = tot ? atomicAdd(cursor, (unsigned long long)tot) : 0ull;
    __syncthreads();
    u64 r = *s_base + pre;
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------------------------------------
// build
// ---------------------------------------------------------------------------------------------
template <class T> struct is_mb { static constexpr bool value = false; };
template <int W> struct is_mb<MbTable<W>> { static constexpr bool value = true; };
// score of one m-mer as minimizer_score has it: hash of its canonical form (gk_device.h)
__device__ __forceinline__ u32 mmer_score(u32 w, int m) {
    const u32 r = (u32)revcomp(Kmer<1>{(u64)w}, m).lo;
    return hash32(w < r ? w : r);
}

// incoming | outcoming << 4 of the stored k-mer y (Graph.scala:270-282): the eight neighbour lookups of the classify; a neighbour
// whose bit is set in `skip` is not looked up here and counts as absent (the distributed classify asks its owner instead)
template <int W, class TT>
__device__ __forceinline__ u32 neighbour_masks(const TT &t, int k, Kmer<W> y, u32 skip) {
    u32 in = 0, out = 0;
    // The 8 lookups are independent, but each is a dependent chain that starts with a cold random
    // sector; done one after the other a lane has ONE miss in flight (168 ms for 1.5e8 16-byte keys).
    // So: prepare all 8 (canonical orientation, segment, start slot), touch the 8 first sectors
    // back to back, then resolve — the later probes of a lookup mostly stay in the sector it opened.
    // The first probed slot's key word is KEPT (eight registers): with 16 waves x 64 lanes x 8 lookups in flight per
    // CU the touched lines (0.5 MB) do not survive in the 32 KiB L1 — nor, 32 CUs to an XCD, in its 4 MiB L2 — until the
    // resolve loop comes back to them, and re-reading them there fetched most sectors from memory TWICE.
    Kmer<W> q[8];
    ProbeAt<W> pa[8];
    u64 w0v[8];
    u32 ties = 0;
    // (minimizer-bucketed table: the candidates' buckets from ONE pass over this k-mer's m-mers — a candidate shares k - 1
    //  bases with it, so its minimizer is the minimum of the shared windows' scores and of its one new window)
    u32 base_succ = 0xffffffffu, base_pred = 0xffffffffu;
    const int mm_ = k < 11 ? k : 11;
    if constexpr (is_mb<TT>::value) {
        const int nwin = k - mm_ + 1;
        u32 mid = 0xffffffffu, s_first = 0xffffffffu, s_last = 0xffffffffu;
        for (int wdw = 0; wdw < nwin; wdw++) {
            const u32 sc = mmer_score((u32)window_bits(y, 2 * wdw) & (u32)low_mask(2 * mm_), mm_);
            if (wdw == 0) s_first = sc;
            if (wdw == nwin - 1) s_last = sc;
            if (wdw != 0 && wdw != nwin - 1) mid = min(mid, sc);
        }
        base_succ = nwin > 1 ? min(mid, s_last) : 0xffffffffu;          // windows 1 .. nwin-1 of y = windows 0 .. nwin-2 of a successor
        base_pred = nwin > 1 ? min(mid, s_first) : 0xffffffffu;         // windows 0 .. nwin-2 of y = windows 1 .. nwin-1 of a predecessor
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (skip >> j & 1u) { w0v[j] = KEY_EMPTY; continue; }            // (asked elsewhere: reads as "not here")
        const Kmer<W> x = (j & 1) ? append_base(y, j >> 1, k) : prepend_base(j >> 1, y, k);
        const Kmer<W> rc = revcomp(x, k);
        const i32 hx = ref_hash(x), hr = ref_hash(rc);
        if ((hx == hr || t.both) && !(x == rc)) ties |= 1u << j;   // both strands may be stored: slow path below
        q[j] = hx < hr ? x : rc;
        if constexpr (is_mb<TT>::value) {
            const u32 neww = (u32)window_bits(x, (j & 1) ? 2 * (k - mm_) : 0) & (u32)low_mask(2 * mm_);
            const u32 score = min((j & 1) ? base_succ : base_pred, mmer_score(neww, mm_));
            pa[j] = probe_at_bucket(t, q[j], (u32)(((u64)hash32(score ^ 0x5bd1e995u) * (u64)t.nb) >> 32));
        } else {
            pa[j] = probe_at(t, q[j], k);
        }
        w0v[j] = pa[j].reg[pa[j].pos].w0;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
        bool hit;
        if (ties >> j & 1u) {
            bool f;
            const Kmer<W> x = (j & 1) ? append_base(y, j >> 1, k) : prepend_base(j >> 1, y, k);
            hit = table_find_either(t, x, k, &f) >= 0;
        } else {
            // first slot from the register; only a slot that is occupied by ANOTHER key sends the probe on (to memory)
            if (w0v[j] == KEY_EMPTY) hit = false;
            else {
                bool first_is_it;
                if constexpr (W == 1) first_is_it = w0v[j] == q[j].lo;
                else { const Stored<2> sk = to_stored(q[j]); first_is_it = w0v[j] == sk.w0 && pa[j].reg[pa[j].pos].w1 == sk.w1; }
                hit = first_is_it || probe_find(pa[j], q[j], true) >= 0;
            }
        }
        if (hit) { if (j & 1) out |= 1u << (j >> 1); else in |= 1u << (j >> 1); }
    }
    return in | (out << 4);
}

// op1 of Graph.buildGraph (Graph.scala:320-329) for every live stored key: incoming/outcoming
// through `contains` on both strands (:270-282), 8 lookups per key; result kept in the slot.
template <int W, class TT>
__global__ __launch_bounds__(BLOCK) void k_classify(TT t, int k, unsigned long long *n_term /* [0] terminal k-mers, [2] their out-edges */,
                                                    unsigned long long *termbits /* bit i of word i / 64: slot i is a terminal k-mer (and not SECONDARY) */) {
    // The table this runs on is 40 % full (map_compact): walking the slots directly leaves 60 % of every wave idle through
    // the eight lookups.  A workgroup therefore takes CLASSIFY_SPT slots per thread at a time, compacts the live ones'
    // indices into LDS (ballot + one LDS atomic per wave) and classifies from that dense list: every lane has work, and a
    // wave keeps 64 x 8 independent sectors in flight instead of ~26 x 8.
    constexpr int SPT = 8;
    __shared__ u32 s_cnt, s_n, s_edges;
    __shared__ uint16_t s_idx[BLOCK * SPT];
    // The terminal slots of this round's window as a bitmap, written out once per round: what k_collect_bits turns into the
    // terminal list WITHOUT reading the table again (the separate pass over all slots was 2.5 ms of C3's buildGraph).
    __shared__ u32 s_tb[BLOCK * SPT / 32];
    if (threadIdx.x == 0) { s_cnt = 0; s_edges = 0; }
    u32 cnt = 0, edges = 0;
    const u64 ncap = t.capacity();
    const int lane = threadIdx.x & 63;
    for (u64 base = (u64)blockIdx.x * (BLOCK * SPT); base < ncap; base += (u64)gridDim.x * (BLOCK * SPT)) {
        __syncthreads();                                    // the previous round's list has been consumed (and its bitmap written)
        if (threadIdx.x == 0) s_n = 0;
        if (threadIdx.x < BLOCK * SPT / 32) s_tb[threadIdx.x] = 0;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < SPT; q++) {
            const u64 i = base + (u64)q * BLOCK + threadIdx.x;
            const bool live = i < ncap && slot_live(&t.slots[i]);
            const unsigned long long mask = __ballot(live);
            u32 wbase = 0;
            if (lane == 0 && mask) wbase = atomicAdd(&s_n, (u32)__popcll(mask));
            wbase = __shfl(wbase, 0);
            if (live) s_idx[wbase + (u32)__popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)(q * BLOCK + threadIdx.x);
        }
        __syncthreads();
        const u32 nlive = s_n;
    for (u32 li = threadIdx.x; li < nlive; li += BLOCK) {
        const u64 i = base + s_idx[li];
        Slot<W> *s = &t.slots[i];
        Kmer<W> y = slot_key(t, i);
        const u32 io = neighbour_masks<W>(t, k, y, 0u);
        const u32 in = io & 15u, out = io >> 4;
        const int ni = __popc(in), no = __popc(out);
        u32 aux = in | (out << 4);
        const bool term = (ni != 1 || no != 1) && (ni != 0 || no != 0);     // Graph.scala:323
        if (term) aux |= AUX_TERMINAL;
        // hash-rule tie (FreqFilter.scala:31-32): x and rc(x) may both be stored; the larger one
        // is marked SECONDARY so the pair yields one pair of nodes
        Kmer<W> rc = revcomp(y, k);
        bool secondary = false;
        if ((t.both || ref_hash(y) == ref_hash(rc)) && !(y == rc) && kmer_less(rc, y) && table_find(t, rc, k) >= 0) secondary = true;
        if (secondary) aux |= AUX_SECONDARY;
        s->aux = aux;
        if (term && !secondary) {
            cnt++;
            // out(y) + out(rc y) = popc(out) + popc(in); a palindrome (y == rc y, even k) is ONE node
            edges += (y == rc) ? (u32)no : (u32)(ni + no);
            const u32 w = s_idx[li];
            atomicOr(&s_tb[w >> 5], 1u << (w & 31u));
        }
    }
        __syncthreads();
        if (threadIdx.x < BLOCK * SPT / 64) {
            const u64 word = (base >> 6) + threadIdx.x;
            if (word * 64 < ncap) termbits[word] = (unsigned long long)s_tb[2 * threadIdx.x] | ((unsigned long long)s_tb[2 * threadIdx.x + 1] << 32);
        }
    }
    __syncthreads();
    if (cnt) atomicAdd(&s_cnt, cnt);
    if (edges) atomicAdd(&s_edges, edges);
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) { atomicAdd(&n_term[0], (unsigned long long)s_cnt); atomicAdd(&n_term[2], (unsigned long long)s_edges); }
}

// ---------------------------------------------------------------------------------------------
// The classify over a PartitionedDNAMap (SURVEY.md 8(e) "beyond counting"; the reference ships the classify closure to every
// partition, Graph.scala:320-329 through PartitionedDNAMap.mapReduce :55-58).  Every rank classifies ITS keys: a neighbour
// whose owner (gk::owner_of: the strand-symmetric minimizer, so x and rc(x) agree) is this rank is looked up here, the others
// are asked of their owners — one query all-to-all of canonical keys, one answer all-to-all of bytes (gk_dist.hip) — and the
// 8-bit (incoming, outcoming) mask of every key ends in its slot's annotation word, to travel with the key in the gather.
//   k_dc_scan<W, false>: slots [s0, s1): local lookups -> aux = mask of the local hits; remote neighbours counted per owner
//   k_dc_scan<W, true> : the same enumeration; every remote neighbour's canonical key goes to its owner's region of `qkeys`
//                        and (slot << 3 | j) to the same position of `qref`
//   k_dc_answer        : `contains` for every received key
//   k_dc_apply         : answers back in query order: aux |= bit
// A neighbour shares k - 1 bases with the k-mer, so its minimizer is the minimum over the shared m-mers and its one new m-mer
// (the same trick as the minimizer-bucketed table's classify above).
// ---------------------------------------------------------------------------------------------
template <int W, bool FILL>
__global__ __launch_bounds__(BLOCK) void k_dc_scan(Table<W> t, int k, int rank, int P, u64 s0, u64 s1, unsigned long long *cnt /* [P]: totals (count) / cursors (fill) */,
                                                   const unsigned long long *off /* [P] first query of each owner's region */, u64 *qkeys, u64 *qref) {
    __shared__ u32 s_cnt[64];
    __shared__ unsigned long long s_base[64];
    const int mm_ = k < 11 ? k : 11;
    const int nwin = k - mm_ + 1;
    const u64 per_round = (u64)gridDim.x * BLOCK;
    const u64 nrounds = (s1 - s0 + per_round - 1) / per_round;          // the same for every workgroup: the barriers below are uniform
    for (u64 r = 0; r < nrounds; r++) {
        const u64 i = s0 + r * per_round + (u64)blockIdx.x * BLOCK + threadIdx.x;
        __syncthreads();
        if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
        __syncthreads();
        bool live = i < s1 && slot_live(&t.slots[i]);
        // (the counting pass leaves "which neighbours are remote" in bits 8..15 of the annotation word: the filling pass skips the
        //  keys that have none — most of them — without going over their m-mers again)
        if (FILL && live && ((t.slots[i].aux >> 8) & 0xffu) == 0u) live = false;
        u32 remote = 0;                 // bit j: neighbour j lives on another rank
        u64 owners = 0;                 // 6 bits per neighbour
        Kmer<W> y{};
        if (live) {
            y = slot_key(t, i);
            u32 mid = 0xffffffffu, s_first = 0xffffffffu, s_last = 0xffffffffu;
            for (int wdw = 0; wdw < nwin; wdw++) {
                const u32 sc = mmer_score((u32)window_bits(y, 2 * wdw) & (u32)low_mask(2 * mm_), mm_);
                if (wdw == 0) s_first = sc;
                if (wdw == nwin - 1) s_last = sc;
                if (wdw != 0 && wdw != nwin - 1) mid = min(mid, sc);
            }
            const u32 base_succ = nwin > 1 ? min(mid, s_last) : 0xffffffffu, base_pred = nwin > 1 ? min(mid, s_first) : 0xffffffffu;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const Kmer<W> x = (j & 1) ? append_base(y, j >> 1, k) : prepend_base(j >> 1, y, k);
                const u32 neww = (u32)window_bits(x, (j & 1) ? 2 * (k - mm_) : 0) & (u32)low_mask(2 * mm_);
                const u32 score = min((j & 1) ? base_succ : base_pred, mmer_score(neww, mm_));
                const u32 o = (u32)(((u64)hash32(score ^ 0x5bd1e995u) * (u64)P) >> 32);          // == gk::owner_of(x, k, P)
                if ((int)o != rank) { remote |= 1u << j; owners |= (u64)o << (6 * j); }
            }
        }
        u32 posv[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            posv[j] = 0;
            if (remote >> j & 1u) posv[j] = atomicAdd(&s_cnt[(owners >> (6 * j)) & 63u], 1u);
        }
        __syncthreads();
        if (threadIdx.x < P && s_cnt[threadIdx.x]) {
            const unsigned long long b = atomicAdd(&cnt[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
            if (FILL) s_base[threadIdx.x] = b;
        }
        if constexpr (!FILL) {
            if (live) t.slots[i].aux = neighbour_masks<W>(t, k, y, remote) | (remote << 8);
        } else {
            __syncthreads();
            if (live) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    if (!(remote >> j & 1u)) continue;
                    const u32 o = (u32)(owners >> (6 * j)) & 63u;
                    const Kmer<W> x = (j & 1) ? append_base(y, j >> 1, k) : prepend_base(j >> 1, y, k);
                    const Kmer<W> rc = revcomp(x, k);
                    const Kmer<W> q = ref_hash(x) < ref_hash(rc) ? x : rc;          // the orientation a counting table stores
                    const u64 at = off[o] + s_base[o] + posv[j];
                    if constexpr (W == 1) qkeys[at] = q.lo;
                    else { qkeys[2 * at] = q.lo; qkeys[2 * at + 1] = q.hi; }
                    qref[at] = (i << 3) | (u64)j;
                }
            }
        }
    }
}
// `contains` (Graph.scala:270-272) of canonical keys another rank asks about
template <int W>
__global__ __launch_bounds__(BLOCK) void k_dc_answer(Table<W> t, int k, const u64 *__restrict__ keys, u64 n, uint8_t *ans) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        Kmer<W> x;
        if constexpr (W == 1) x = Kmer<1>{keys[i]};
        else x = Kmer<2>{keys[2 * i], keys[2 * i + 1]};
        bool f;
        ans[i] = table_find_either(t, x, k, &f) >= 0 ? 1 : 0;
    }
}
template <int W>
__global__ __launch_bounds__(BLOCK) void k_dc_apply(Table<W> t, const u64 *__restrict__ qref, const uint8_t *__restrict__ ans, u64 n) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        if (!ans[i]) continue;
        const u64 ref = qref[i];
        const int j = (int)(ref & 7u);
        atomicOr(&t.slots[ref >> 3].aux, (j & 1) ? 16u << (j >> 1) : 1u << (j >> 1));
    }
}
// what k_classify derives from the masks that came with a gathered table (terminal, the SECONDARY mark of a hash-rule tie, the count of terminal k-mers)
template <int W>
__global__ __launch_bounds__(BLOCK) void k_finish_masks(Table<W> t, int k, unsigned long long *n_term, unsigned long long *termbits) {
    __shared__ u32 s_cnt, s_edges;
    if (threadIdx.x == 0) { s_cnt = 0; s_edges = 0; }
    __syncthreads();
    u32 cnt = 0, edges = 0;
    const u64 ncap = t.capacity();
    // (wave-uniform trip count: a wave's 64 lanes hold 64 consecutive slots = one word of the terminal bitmap)
    for (u64 i0 = (u64)blockIdx.x * BLOCK + (threadIdx.x & ~63u); i0 < ncap; i0 += (u64)gridDim.x * BLOCK) {
        const u64 i = i0 + (threadIdx.x & 63u);
        bool is_term = false;
        if (i < ncap && slot_live(&t.slots[i])) {
        u32 aux = t.slots[i].aux & 0xffu;
        const int ni = __popc(aux & 15u), no = __popc(aux >> 4);
        const bool term = (ni != 1 || no != 1) && (ni != 0 || no != 0);     // Graph.scala:323
        if (term) aux |= AUX_TERMINAL;
        const Kmer<W> y = slot_key(t, i);
        const Kmer<W> rc = revcomp(y, k);
        bool secondary = false;
        if ((t.both || ref_hash(y) == ref_hash(rc)) && !(y == rc) && kmer_less(rc, y) && table_find(t, rc, k) >= 0) secondary = true;
        if (secondary) aux |= AUX_SECONDARY;
        t.slots[i].aux = aux;
        if (term && !secondary) { cnt++; edges += (y == rc) ? (u32)no : (u32)(ni + no); is_term = true; }
        }
        const unsigned long long word = __ballot(is_term);
        if ((threadIdx.x & 63u) == 0) termbits[i0 >> 6] = word;
    }
    if (cnt) atomicAdd(&s_cnt, cnt);
    if (edges) atomicAdd(&s_edges, edges);
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) { atomicAdd(&n_term[0], (unsigned long long)s_cnt); atomicAdd(&n_term[2], (unsigned long long)s_edges); }
}

// The terminal slots, compacted — from the bitmap the classify left (one word per 64 slots), not from the table: a thread takes one
// word, a workgroup reserves its output with ONE atomic per 256 words (the cursor is a single address: same-address atomics retire
// at ~88 per microsecond chip-wide).  Slots come out in ascending order inside a workgroup's share.
__global__ __launch_bounds__(BLOCK) void k_collect_bits(const unsigned long long *__restrict__ termbits, u64 nwords, u64 *tslots, unsigned long long *cursor) {
    __shared__ u32 lds4[BLOCK / 64];
    __shared__ unsigned long long s_base;
    const u64 ngroups = (nwords + BLOCK - 1) / BLOCK;
    for (u64 g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const u64 w = g * BLOCK + threadIdx.x;
        unsigned long long bits = w < nwords ? termbits[w] : 0ull;
        u64 o = block_reserve((u32)__popcll(bits), cursor, lds4, &s_base);
        while (bits) {
            const int b = __ffsll((long long)bits) - 1;
            tslots[o++] = w * 64 + (u64)b;
            bits &= bits - 1;
        }
    }
}

// nodeMap (Graph.scala:343-347): node 2j = the stored terminal k-mer, node 2j+1 = its reverse
// complement (termKmers = set ++ set.map(revComplement), :330-333); plus the (node, base) stubs of
// buildEdges (:351): one edge per outgoing base, in A,G,C,T order.
template <int W, class TT>
__global__ __launch_bounds__(BLOCK) void k_make_nodes(TT t, int k, const u64 *tslots, u64 nT, GraphView g,
                                                      unsigned long long *ecursor) {
    __shared__ u32 lds4[BLOCK / 64];
    __shared__ unsigned long long s_base;
    const u64 ngroups = (nT + BLOCK - 1) / BLOCK;
    for (u64 grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const u64 j = grp * BLOCK + threadIdx.x;
        u32 m0 = 0, m1 = 0;
        Kmer<W> y{}, rc{};
        bool pal = false;
        if (j < nT) {
            const u64 slot = tslots[j];
            const u32 aux = t.slots[slot].aux;
            y = slot_key(t, slot);
            rc = revcomp(y, k);
            pal = (y == rc);
            m0 = (aux >> 4) & 15u;                  // outcoming(y)
            m1 = pal ? 0u : rev4(aux & 15u);        // outcoming(rc y) = complemented incoming(y)
            t.slots[slot].aux = AUX_NODE | (u32)j;          // (nT < 2^31 is checked by the host)
        }
        u64 e = block_reserve((u32)(__popc(m0) + __popc(m1)), ecursor, lds4, &s_base);
        if (j < nT) {
            const u32 n0 = (u32)(2 * j), n1 = n0 + 1;
            g.node_lo[n0] = y.lo; g.node_lo[n1] = rc.lo;
            if constexpr (W == 2) { g.node_hi[n0] = y.hi; g.node_hi[n1] = rc.hi; }
            else { g.node_hi[n0] = 0; g.node_hi[n1] = 0; }
            g.node_alive[n0] = 1; g.node_alive[n1] = pal ? 0 : 1;
            for (int side = 0; side < 2; side++) {
                const u32 n = side ? n1 : n0, m = side ? m1 : m0;
                u32 ord = 0;
                for (int b = 0; b < 4; b++) {
                    u32 id = NONE;
                    if (m & (1u << b)) {
                        id = (u32)e++;
                        g.e_start[id] = n;
                        g.e_first[id] = (uint8_t)b;
                        g.e_alive[id] = 1;
                        ord = order_append(ord, b);
                    }
                    g.out_edge[(u64)n * 4 + b] = id;
                }
                g.out_order[n] = ord;
            }
        }
    }
}

// buildEdges (Graph.scala:349-365): from a node, follow the unique outgoing base until the next
// terminal k-mer.  One lane per edge; every step is one dependent table probe.
// pass 0: walk once — end node, length, in-degree — keeping the first WALK_BUF bases in registers; the workgroup then
//         reserves the pool bytes of ALL its edges with one atomic and an edge that fits the registers (nearly all of
//         them on a bushy error graph: 7 bases on average at C3) writes its sequence straight away.
// pass 1: only for edges longer than WALK_BUF: walk again and emit the bases 2 bits each at e_off (assigned in pass 0).
static constexpr u32 WALK_BUF = 128;          // bases kept in four 64-bit registers
template <int W, class TT>
__global__ __launch_bounds__(BLOCK) void k_walk(TT t, int k, GraphView g, int pass, u64 max_steps,
                                                unsigned long long *pool_cursor, unsigned long long *n_long, u32 *err) {
    __shared__ u32 lds4[BLOCK / 64];
    __shared__ unsigned long long s_base;
    const u64 ngroups = (g.n_edges + BLOCK - 1) / BLOCK;
    for (u64 grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const u64 e = grp * BLOCK + threadIdx.x;
        const bool active = e < g.n_edges && (pass == 0 || g.e_len[e] > WALK_BUF);
        u64 len = 0;
        u64 b0 = 0, b1 = 0, b2 = 0, b3 = 0;
        if (active) {
            const u32 n = g.e_start[e];
            const int first = g.e_first[e];
            Kmer<W> cur = append_base(node_kmer<W>(g, n), first, k);       // read.drop(1) :+ base  :353
            len = 1;
            u32 acc = (u32)first;                                           // builder += base        :352
            b0 = (u64)first;
            const u64 off = pass ? g.e_off[e] : 0;
            u32 end = NONE;
            for (u64 step = 0; step <= max_steps; step++) {
                bool fwd;
                i64 slot = table_find_either(t, cur, k, &fwd);
                if (slot < 0) { *err = 1; break; }
                const u32 aux = t.slots[slot].aux;
                if (aux & AUX_NODE) {                                       // nodeMap.contains(seq)  :355
                    end = aux_node(aux, fwd);
                    break;
                }
                const u32 om = fwd ? ((aux >> 4) & 15u) : rev4(aux & 15u);  // outcoming(seq)         :356
                if (__popc(om) != 1) { *err = 2; break; }                   // assert(out.size == 1)  :357
                const int nb = __ffs(om) - 1;
                if (pass) {
                    acc |= (u32)nb << ((len & 3) * 2);
                    if ((len & 3) == 3) { g.pool[off + (len >> 2)] = (uint8_t)acc; acc = 0; }
                } else if (len < WALK_BUF) {
                    const u64 bits = (u64)nb << ((len & 31) * 2);
                    if (len < 32) b0 |= bits; else if (len < 64) b1 |= bits; else if (len < 96) b2 |= bits; else b3 |= bits;
                }
                len++;
                cur = append_base(cur, nb, k);                              // seq.drop(1) :+ out(0)  :360
            }
            if (pass) {
                if (len & 3) g.pool[off + (len >> 2)] = (uint8_t)acc;
            } else {
                g.e_end[e] = end;
                g.e_len[e] = len;
                if (end != NONE) atomicAdd(&g.in_deg[end], 1u);             // end.inEdgeIds += id    :181
                else *err = 3;
            }
        }
        if (pass) continue;
        // pool bytes of this workgroup's edges: one atomic (per-block totals stay < 2^32 for edges up to 16M bases; longer ones reserve alone)
        const u64 bytes = active ? (len + 3) / 4 : 0;
        const u32 small = bytes < (1u << 22) ? (u32)bytes : 0u;
        u64 o = block_reserve(small, pool_cursor, lds4, &s_base);
        if (!active) continue;
        if (small != bytes) o = atomicAdd(pool_cursor, (unsigned long long)bytes);
        g.e_off[e] = o;
        if (len > WALK_BUF) { atomicAdd(n_long, 1ull); continue; }
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const u64 word = w == 0 ? b0 : w == 1 ? b1 : w == 2 ? b2 : b3;
#pragma unroll
            for (int j = 0; j < 8; j++)
                if ((u64)(w * 8 + j) < bytes) g.pool[o + w * 8 + j] = (uint8_t)(word >> (8 * j));
        }
    }
}

// The same walk, fed from a queue.  With one edge per lane a wave lasts as long as its LONGEST edge: on an error graph the
// mean edge is 7 bases and the longest of 64 is ten times that, so nine lanes in ten idle (17 of C3's 59 ms).  Here a lane
// that finishes its edge takes the next one: a wave claims 1024 consecutive edges from the global queue with one atomic
// (a claim per edge, or per refill, would be millions of atomics on ONE address at ~88 per microsecond) and hands them to
// its idle lanes by ballot.  Nothing else in the loop is shared: the first WALK_BUF bases of an edge go to a 32-byte staging
// slot per edge, and k_place_edges afterwards assigns the pool offsets (one atomic per 2048 edges) and copies the staged
// bases.  Edges longer than WALK_BUF are emitted by k_walk's pass 1 as before.
template <int W, class TT>
__global__ __launch_bounds__(BLOCK) void k_walk_q(TT t, int k, GraphView g, u64 max_steps,
                                                  unsigned long long *queue, ulonglong2 *stage, unsigned long long *n_long, u32 *err) {
    constexpr u64 CHUNK = 1024;
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    u64 loc_next = 0, loc_end = 0;                 // wave-uniform: claimed edges not yet handed to a lane
    bool drained = false;                          // wave-uniform: the global queue is empty
    bool have = false;
    u64 e = 0, len = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    Kmer<W> cur{};
    u32 longs = 0;
    for (;;) {
        // ---- idle lanes take edges
        for (int attempt = 0; attempt < 2; attempt++) {
            const unsigned long long need = __ballot(!have);
            if (!need) break;
            const u64 avail = loc_end - loc_next;
            const u64 rank = (u64)__popcll(need & lt_mask);
            if (!have && rank < avail) {
                e = loc_next + rank;
                const u32 n = g.e_start[e];
                const int first = g.e_first[e];
                cur = append_base(node_kmer<W>(g, n), first, k);           // read.drop(1) :+ base  :353
                len = 1;
                b0 = (u64)first; b1 = b2 = b3 = 0;                          // builder += base        :352
                have = true;
            }
            loc_next += min((u64)__popcll(need), avail);
            if (loc_next < loc_end || drained) break;
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(queue, (unsigned long long)CHUNK);
            base = __shfl(base, 0);
            if (base >= g.n_edges) { drained = true; loc_next = loc_end = 0; break; }
            loc_next = base;
            loc_end = min((u64)base + CHUNK, g.n_edges);
        }
        if (!__ballot(have)) { if (drained) break; else continue; }
        // ---- one step of every lane that holds an edge
        if (have) {
            bool fwd;
            const i64 slot = table_find_either(t, cur, k, &fwd);
            u32 end = NONE;
            bool finished = false;
            if (slot < 0) { *err = 1; finished = true; }
            else {
                const u32 aux = t.slots[slot].aux;
                if (aux & AUX_NODE) {                                       // nodeMap.contains(seq)  :355
                    end = aux_node(aux, fwd);
                    finished = true;
                } else {
                    const u32 om = fwd ? ((aux >> 4) & 15u) : rev4(aux & 15u);  // outcoming(seq)     :356
                    if (__popc(om) != 1) { *err = 2; finished = true; }     // assert(out.size == 1)  :357
                    else if (len > max_steps) { *err = 3; finished = true; }
                    else {
                        const int nb = __ffs(om) - 1;
                        if (len < WALK_BUF) {
                            const u64 bits = (u64)nb << ((len & 31) * 2);
                            if (len < 32) b0 |= bits; else if (len < 64) b1 |= bits; else if (len < 96) b2 |= bits; else b3 |= bits;
                        }
                        len++;
                        cur = append_base(cur, nb, k);                      // seq.drop(1) :+ out(0)  :360
                    }
                }
            }
            if (finished) {
                g.e_end[e] = end;
                g.e_len[e] = len;
                if (end != NONE) atomicAdd(&g.in_deg[end], 1u);             // end.inEdgeIds += id    :181
                else *err = 3;
                if (len > WALK_BUF) longs++;
                else {
                    stage[2 * e] = make_ulonglong2(b0, b1);
                    if (len > 64) stage[2 * e + 1] = make_ulonglong2(b2, b3);
                }
                have = false;
            }
        }
    }
    for (int d = 32; d; d >>= 1) longs += __shfl_down(longs, d);
    if (lane == 0 && longs) atomicAdd(n_long, (unsigned long long)longs);
}

// pool offsets of all edges (each starts on a byte boundary) + the staged bases of the short ones into the pool.
// A workgroup takes 8 consecutive edges per thread and reserves their bytes with ONE atomic.
__global__ __launch_bounds__(BLOCK) void k_place_edges(GraphView g, const ulonglong2 *stage, unsigned long long *cursor) {
    __shared__ u32 lds4[BLOCK / 64];
    __shared__ unsigned long long s_base;
    constexpr int EPT = 8;
    const u64 per_block = (u64)BLOCK * EPT;
    const u64 ngroups = (g.n_edges + per_block - 1) / per_block;
    for (u64 grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const u64 e0 = grp * per_block + (u64)threadIdx.x * EPT;
        u64 bytes[EPT];
        u32 small = 0;
#pragma unroll
        for (int j = 0; j < EPT; j++) {
            bytes[j] = e0 + j < g.n_edges ? (g.e_len[e0 + j] + 3) / 4 : 0;
            if (bytes[j] < (1u << 22)) small += (u32)bytes[j];       // (per-block totals stay < 2^32; a longer edge reserves alone)
        }
        u64 o = block_reserve(small, cursor, lds4, &s_base);
#pragma unroll
        for (int j = 0; j < EPT; j++) {
            const u64 e = e0 + j;
            if (e >= g.n_edges) break;
            const u64 nb = bytes[j];
            u64 at = o;
            if (nb < (1u << 22)) o += nb; else at = atomicAdd(cursor, (unsigned long long)nb);
            g.e_off[e] = at;
            if (nb > WALK_BUF / 4) continue;                          // a long edge: k_walk pass 1 emits it
            const ulonglong2 lo = stage[2 * e];
            ulonglong2 hi = make_ulonglong2(0, 0);
            if (nb > 16) hi = stage[2 * e + 1];
            for (u64 q = 0; q < nb; q++) {
                const u64 word = q < 8 ? lo.x : q < 16 ? lo.y : q < 24 ? hi.x : hi.y;
                g.pool[at + q] = (uint8_t)(word >> (8 * (q & 7)));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Unitigs by POINTER JUMPING (list ranking) instead of one lane walking each edge base by base:
// k_walk is one dependent probe per base, so a single 4.6 Mbp unitig (an error-free bacterial
// genome) costs 2 x 4.6M x ~1 us = 10 s on one lane.  Here every ORIENTED interior k-mer
// (id = 2 * rank of its slot among the live slots + orientation; interior = non-terminal with exactly one successor) starts with a
// pointer to its successor and distance 1, pre-terminals (successor is a terminal k-mer) absorb,
// and log2(longest unitig) rounds of  d[u] += d[next[u]]; next[u] = next[next[u]]  give every
// interior k-mer its pre-terminal and its distance to it.  From that:
//   edge (s, b): u0 = s.drop(1) :+ b;  len = d[u0] + 2;  end = successor of pt[u0]
//   the base appended after interior u (position j+1 of its edge, u = u_j) is placed by u itself:
//   the chain of rc(u) runs rc(u_{j-1}), ..., rc(u_0), rc(s), so j = d[rc u], u0 = rc(pt[rc u]),
//   s = rc(successor of pt[rc u]), b = last base of u0 — no walk anywhere.
// Members of all-(1,1) cycles never absorb and are skipped (Graph.scala:375 "perfect cycles are
// ignored").  Same results as k_walk (tests run both).
// ---------------------------------------------------------------------------------------------
// unique outgoing base of an interior oriented k-mer (from the slot's masks), -1 if not exactly one
__device__ __forceinline__ int single_out_base(u32 aux, int ori) {
    const u32 om = ori ? rev4(aux & 15u) : ((aux >> 4) & 15u);
    return __popc(om) == 1 ? __ffs(om) - 1 : -1;
}

// ---- rank of a live slot --------------------------------------------------------------------------------------------
// The pointer-jumping state is indexed by LIVE KEY, not by slot: one (live mask, live slots before) pair per 64 slots — 0.25
// bytes per slot — turns a slot index into its rank with one 16-byte read.  (Until round 3 two arrays of 2 x capacity x 16
// bytes were indexed by slot: 64 B per SLOT, 305 GB for C5's table.)
struct alignas(16) RankBlk { u64 mask; u64 base; };
static constexpr u32 RANK_CHUNK = 256;              // 64-slot blocks per chunk (16384 slots): one workgroup, one thread per block
__device__ __forceinline__ u64 rank_of(const RankBlk *rb, u64 slot) {
    const RankBlk b = rb[slot >> 6];
    return b.base + (u64)__popcll(b.mask & ((1ull << (slot & 63)) - 1ull));
}
template <int W, class TT>
__global__ __launch_bounds__(BLOCK) void k_rank_masks(TT t, RankBlk *rb, u64 nblk, u32 *chunk_tot, u64 nchunks) {
    __shared__ u32 s_tot;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 ncap = t.capacity();
    for (u64 c = blockIdx.x; c < nchunks; c += gridDim.x) {
        __syncthreads();
        if (threadIdx.x == 0) s_tot = 0;
        __syncthreads();
        u32 tot = 0;
        for (u32 j = wave; j < RANK_CHUNK; j += BLOCK / 64) {
            const u64 blk = c * RANK_CHUNK + j;
            if (blk >= nblk) break;
            const u64 i = blk * 64 + lane;
            const unsigned long long m = __ballot(i < ncap && slot_live(&t.slots[i]));
            if (lane == 0) rb[blk].mask = m;
            tot += (u32)__popcll(m);
        }
        if (lane == 0 && tot) atomicAdd(&s_tot, tot);
        __syncthreads();
        if (threadIdx.x == 0) chunk_tot[c] = s_tot;
    }
}
// exclusive scan of the chunk totals: ONE workgroup, every thread a contiguous share (3e5 chunks at C5: ~300 per thread)
__global__ __launch_bounds__(1024) void k_rank_scan(const u32 *chunk_tot, u64 *chunk_base, u64 nchunks) {
    __shared__ u64 s_sum[1024];
    const u64 per = (nchunks + 1023) / 1024, c0 = threadIdx.x * per, c1 = min(c0 + per, nchunks);
    u64 sum = 0;
    for (u64 c = c0; c < c1; c++) sum += chunk_tot[c];
    s_sum[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) { u64 run = 0; for (int i = 0; i < 1024; i++) { const u64 v = s_sum[i]; s_sum[i] = run; run += v; } chunk_base[nchunks] = run; }
    __syncthreads();
    u64 run = s_sum[threadIdx.x];
    for (u64 c = c0; c < c1; c++) { chunk_base[c] = run; run += chunk_tot[c]; }
}
__global__ __launch_bounds__(RANK_CHUNK) void k_rank_fill(RankBlk *rb, u64 nblk, const u64 *chunk_base, u64 nchunks) {
    __shared__ u32 wsum[RANK_CHUNK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (u64 c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const u64 blk = c * RANK_CHUNK + threadIdx.x;
        const u32 v = blk < nblk ? (u32)__popcll(rb[blk].mask) : 0u;
        u32 inc = wave_incl_scan(v);
        __syncthreads();
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        u32 pre = 0;
        for (int w = 0; w < wave; w++) pre += wsum[w];
        if (blk < nblk) rb[blk].base = chunk_base[c] + pre + inc - v;
    }
}

// ---- pointer-jumping state: ONE 64-bit word per oriented live k-mer (index 2 x rank + orientation), updated IN PLACE -------
//   absorbed (the successor is a terminal k-mer):  PJ_ABS | first base of this oriented k-mer << 32 | node id of that terminal
//   on its way:                                    successor index << 29 | distance covered so far (saturating)
//   not registered (terminal, secondary, (0,0)):   ~0
// A jump reads the successor's word and writes the own one; a reader that meets a word in mid-round sees either the old or
// the new state of that k-mer — both are true statements "my pointer is d steps ahead of me" — so no second buffer is needed
// (stale lines in another XCD's L2 are old states too, and kernel boundaries bring everybody up to date).
static constexpr u64 PJ_ABS = 1ull << 63, PJ_UNREG = ~0ull;
static constexpr u32 PJ_DBITS = 29;
static constexpr u64 PJ_DMAX = (1ull << PJ_DBITS) - 1ull;          // 5.4e8 bases: longer unitigs are refused (GK_E_CAPACITY)
static constexpr u64 PJ_MAX_STATES = 1ull << (63 - PJ_DBITS);      // 2^34 oriented k-mers
__device__ __forceinline__ u64 pj_pack(u64 nxt, u64 dist) { return (nxt << PJ_DBITS) | min(dist, PJ_DMAX); }
__device__ __forceinline__ u64 pj_nxt(u64 s) { return s >> PJ_DBITS; }
__device__ __forceinline__ u64 pj_dist(u64 s) { return s & PJ_DMAX; }
__device__ __forceinline__ bool pj_absorbed(u64 s) { return (s & PJ_ABS) && s != PJ_UNREG; }
__device__ __forceinline__ u32 pj_node(u64 s) { return (u32)s; }
__device__ __forceinline__ int pj_first(u64 s) { return (int)((s >> 32) & 3u); }

template <int W, class TT>
__global__ __launch_bounds__(BLOCK) void k_pj_init(TT t, int k, const RankBlk *__restrict__ rb, u64 *st) {
    const u64 ncap = t.capacity();
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < ncap; i += (u64)gridDim.x * BLOCK) {
        if (!slot_live(&t.slots[i])) continue;
        const u32 aux = t.slots[i].aux;
        if (aux & (AUX_NODE | AUX_TERMINAL | AUX_SECONDARY)) continue;
        const Kmer<W> y = slot_key(t, i);
        const u64 r = rank_of(rb, i);
        for (int ori = 0; ori < 2; ori++) {
            const int nb = single_out_base(aux, ori);
            if (nb < 0) continue;                              // (0,0) k-mers are in no unitig
            const Kmer<W> u = ori ? revcomp(y, k) : y;
            bool fwd;
            const i64 ws = table_find_either(t, append_base(u, nb, k), k, &fwd);
            if (ws < 0) continue;
            const u32 wa = t.slots[ws].aux;
            if (wa & AUX_NODE) st[2 * r + ori] = PJ_ABS | ((u64)first_base(u) << 32) | (u64)aux_node(wa, fwd);     // pre-terminal: absorbs
            else st[2 * r + ori] = pj_pack(2 * rank_of(rb, (u64)ws) + (fwd ? 0u : 1u), 1);
        }
    }
}

__global__ __launch_bounds__(BLOCK) void k_pj_round(u64 *st, u64 n, u32 *changed /* [0] a pointer moved, [1] error */) {
    bool ch = false;
    for (u64 u = (u64)blockIdx.x * BLOCK + threadIdx.x; u < n; u += (u64)gridDim.x * BLOCK) {
        const u64 su = st[u];
        if (su & PJ_ABS) continue;                          // absorbed, or not registered
        const u64 sv = st[pj_nxt(su)];                      // the one random read of the round
        if (sv == PJ_UNREG) { changed[1] = 8; continue; }   // an interior k-mer's successor is interior or terminal: cannot happen
        if (sv & PJ_ABS) continue;                          // the pointer is at the chain's pre-terminal: done
        st[u] = pj_pack(pj_nxt(sv), pj_dist(su) + pj_dist(sv));
        ch = true;
    }
    if (ch) changed[0] = 1;
}

template <int W, class TT>
__global__ __launch_bounds__(BLOCK) void k_pj_edges(TT t, int k, GraphView g, const RankBlk *__restrict__ rb, const u64 *__restrict__ st, u32 *err) {
    for (u64 e = (u64)blockIdx.x * BLOCK + threadIdx.x; e < g.n_edges; e += (u64)gridDim.x * BLOCK) {
        const u32 n = g.e_start[e];
        const Kmer<W> u0 = append_base(node_kmer<W>(g, n), g.e_first[e], k);
        bool fwd;
        const i64 s0 = table_find_either(t, u0, k, &fwd);
        if (s0 < 0) { *err = 1; continue; }
        u32 end = NONE;
        u64 len = 1;
        const u32 a0 = t.slots[s0].aux;
        if (a0 & AUX_NODE) {
            end = aux_node(a0, fwd);
        } else {
            const u64 s = st[2 * rank_of(rb, (u64)s0) + (fwd ? 0u : 1u)];
            u64 sp = s;                                     // the chain's pre-terminal
            if (!(s & PJ_ABS)) { sp = st[pj_nxt(s)]; len = pj_dist(s) + 2; if (pj_dist(s) == PJ_DMAX) { *err = 9; continue; } }
            else len = 2;
            if (!pj_absorbed(sp)) { *err = 4; continue; }   // unregistered / not absorbed: cannot happen on a chain that leaves a node
            end = pj_node(sp);
        }
        g.e_end[e] = end;
        g.e_len[e] = len;
        if (end != NONE) atomicAdd(&g.in_deg[end], 1u);
        else *err = 3;
    }
}

__device__ __forceinline__ void pool_or(uint8_t *pool, u64 off, u64 pos, int base) {
    const u64 byte = off + (pos >> 2);
    u32 *w = reinterpret_cast<u32 *>(pool) + (byte >> 2);
    atomicOr(w, (u32)base << (((byte & 3) * 8) + (pos & 3) * 2));
}

// every interior oriented k-mer u places the base that follows it (see the derivation above): a scan over the table's slots
template <int W, class TT>
__global__ __launch_bounds__(BLOCK) void k_pj_emit(TT t, int k, GraphView g, const RankBlk *__restrict__ rb, const u64 *__restrict__ st, u32 *err) {
    const u64 tid = (u64)blockIdx.x * BLOCK + threadIdx.x, stride = (u64)gridDim.x * BLOCK;
    for (u64 e = tid; e < g.n_edges; e += stride) pool_or(g.pool, g.e_off[e], 0, g.e_first[e]);       // builder += base  :352
    const u64 ncap = t.capacity();
    for (u64 i = tid; i < ncap; i += stride) {
        if (!slot_live(&t.slots[i])) continue;
        const u32 aux = t.slots[i].aux;
        if (aux & (AUX_NODE | AUX_TERMINAL | AUX_SECONDARY)) continue;
        const u64 r = rank_of(rb, i);
        for (int ori = 0; ori < 2; ori++) {
            const u64 su = st[2 * r + ori];
            if (su == PJ_UNREG) continue;
            const u64 sv = st[2 * r + (ori ^ 1)];               // v = rc(u): same slot, other orientation
            if (sv == PJ_UNREG) { *err = 7; continue; }         // the partner orientation must have been registered too
            const u64 pu = (su & PJ_ABS) ? su : st[pj_nxt(su)], pv = (sv & PJ_ABS) ? sv : st[pj_nxt(sv)];
            if (!pj_absorbed(pu) || !pj_absorbed(pv)) continue; // member of an all-(1,1) cycle
            const u64 dv = (sv & PJ_ABS) ? 0ull : pj_dist(sv);
            if (dv == PJ_DMAX) { *err = 9; continue; }
            const int nb = single_out_base(aux, ori);
            const u32 rs_node = pj_node(pv);                    // node of rc(s)
            if (nb < 0 || rs_node == NONE) { *err = 5; continue; }
            const u32 partner = rs_node ^ 1u;
            const u32 s_node = g.node_alive[partner] ? partner : rs_node;      // palindromic node: s == rc(s)
            const int b = 3 - pj_first(pv);                     // last base of u0 = complement of the first base of rc(u0)
            const u32 e = g.out_edge[(u64)s_node * 4 + b];
            if (e == NONE) { *err = 6; continue; }
            pool_or(g.pool, g.e_off[e], dv + 1, nb);
        }
    }
}

// byte offsets of the edge sequences in the pool (each edge starts on a byte boundary)
__global__ __launch_bounds__(BLOCK) void k_reserve_pool(GraphView g, u64 first_edge, unsigned long long *cursor) {
    __shared__ u32 lds4[BLOCK / 64];
    __shared__ unsigned long long s_base;
    const u64 n = g.n_edges - first_edge;
    const u64 ngroups = (n + BLOCK - 1) / BLOCK;
    for (u64 grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const u64 e = first_edge + grp * BLOCK + threadIdx.x;
        // per-block totals stay < 2^32 for edges up to 16M bases each; longer ones reserve alone
        u64 bytes = e < g.n_edges ? (g.e_len[e] + 3) / 4 : 0;
        u32 small = bytes < (1u << 22) ? (u32)bytes : 0u;
        u64 o = block_reserve(small, cursor, lds4, &s_base);
        if (e < g.n_edges) g.e_off[e] = small == bytes ? o : atomicAdd(cursor, (unsigned long long)bytes);
    }
}

// k-mer -> node id index (open addressing over node ids), for point queries on the graph
template <int W> __global__ __launch_bounds__(BLOCK) void k_build_nidx(GraphView g) {
    for (u64 n = (u64)blockIdx.x * BLOCK + threadIdx.x; n < g.n_nodes; n += (u64)gridDim.x * BLOCK) {
        if (!g.node_alive[n]) continue;
        u64 i = slot_hash(node_kmer<W>(g, n)) & g.nidx_mask;
        while (atomicCAS(&g.nidx[i], NONE, (u32)n) != NONE) i = (i + 1) & g.nidx_mask;
    }
}
template <int W> __device__ __forceinline__ u32 node_find(const GraphView &g, Kmer<W> x) {
    u64 i = slot_hash(x) & g.nidx_mask;
    for (u64 p = 0; p <= g.nidx_mask; p++) {
        u32 n = g.nidx[i];
        if (n == NONE) return NONE;
        if (node_kmer<W>(g, n) == x) return n;
        i = (i + 1) & g.nidx_mask;
    }
    return NONE;
}

// ---------------------------------------------------------------------------------------------
// simplifyGraph (Graph.scala:211-230)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_in_single(GraphView g, u32 *in_single) {
    for (u64 e = (u64)blockIdx.x * BLOCK + threadIdx.x; e < g.n_edges; e += (u64)gridDim.x * BLOCK)
        if (g.e_alive[e] && g.in_deg[g.e_end[e]] == 1) in_single[g.e_end[e]] = (u32)e;
}
// 0 keep; 1 interior (1 in, 1 out, e1 != e2 : merged away, :222-226); 2 self-loop (e1 == e2, :220-221);
// 3 isolated (:215-216)
__global__ __launch_bounds__(BLOCK) void k_node_class(GraphView g, const u32 *in_single, uint8_t *cls) {
    for (u64 n = (u64)blockIdx.x * BLOCK + threadIdx.x; n < g.n_nodes; n += (u64)gridDim.x * BLOCK) {
        uint8_t c = 0;
        if (g.node_alive[n]) {
            const u32 o = g.out_order[n];
            const int nout = order_count(o), nin = (int)g.in_deg[n];
            if (nin == 0 && nout == 0) c = 3;
            else if (nin == 1 && nout == 1) c = in_single[n] == g.out_edge[n * 4 + order_base(o, 0)] ? 2 : 1;
        }
        cls[n] = c;
    }
}
// A chain = a live edge that leaves a kept node and enters an interior node, followed through
// every interior node to the first non-interior end.  Sequential simplifyGraph produces exactly
// one merged edge per chain (e1.seq ++ e2.seq, repeatedly) whatever the node order; chains of
// interior nodes only (perfect cycles) vanish.  pass 0 counts, pass 1 writes.
// A piece of LONG_PIECE bases and more is not copied base by base by the chain's lane (a contig graph has edges of hundreds of
// kilobases: 4 Mbp of merges took 109 ms that way) but listed for k_copy_long, which copies it 16 bases per thread and ORs it
// into the zeroed output; the lane ORs the bytes it shares with such a piece instead of storing them.
static constexpr u64 LONG_PIECE = 1024;
struct LongPiece { u64 src_off, dst_off, dst_base, len; };     // source byte offset, chain's byte offset, base position inside the chain, bases
__device__ __forceinline__ void pool_or(uint8_t *pool, u64 byte, u32 val) {
    atomicOr(reinterpret_cast<u32 *>(pool + (byte & ~3ull)), val << (8u * (u32)(byte & 3ull)));
}
__global__ __launch_bounds__(BLOCK) void k_copy_long(GraphView g, const LongPiece *pieces) {
    const LongPiece p = pieces[blockIdx.x];
    for (u64 c = threadIdx.x; c * 16 < p.len; c += BLOCK) {
        const u64 first = c * 16;
        const u32 n = (u32)min((u64)16, p.len - first);
        // 16 bases = 32 bits of the source starting at base `first` (any 2-bit alignment): five bytes cover them
        const u64 sb = p.src_off + (first >> 2);
        u64 five = 0;
        for (u32 q = 0; q < 5 && (first >> 2) + q < (p.len + 3) / 4; q++) five |= (u64)g.pool[sb + q] << (8 * q);
        u32 bits = (u32)(five >> ((first & 3) * 2));
        if (n < 16) bits &= (1u << (2 * n)) - 1u;
        const u64 bitpos = p.dst_off * 8 + (p.dst_base + first) * 2;            // in the pool
        u32 *w = reinterpret_cast<u32 *>(g.pool) + (bitpos >> 5);
        const u32 sh = (u32)(bitpos & 31);
        atomicOr(w, bits << sh);
        if (sh && (bits >> (32 - sh))) atomicOr(w + 1, bits >> (32 - sh));
    }
}
__global__ __launch_bounds__(BLOCK) void k_chain(GraphView g, const uint8_t *cls, int pass, u64 old_edges, u64 old_pool,
                                                 unsigned long long *counters /* [0]=chains [1]=bytes [2]=long pieces */, u32 *merged_key,
                                                 LongPiece *long_pieces) {
    for (u64 e = (u64)blockIdx.x * BLOCK + threadIdx.x; e < old_edges; e += (u64)gridDim.x * BLOCK) {
        if (!g.e_alive[e]) continue;
        const u32 s = g.e_start[e];
        if (cls[s] != 0 || cls[g.e_end[e]] != 1) continue;
        u64 total = 0;
        u32 cur = (u32)e, maxn = NONE, nlong = 0;
        for (u64 guard = 0; guard <= old_edges; guard++) {
            total += g.e_len[cur];
            if (g.e_len[cur] >= LONG_PIECE) nlong++;
            const u32 v = g.e_end[cur];
            if (cls[v] != 1) break;
            if (maxn == NONE || node_less(g, maxn, v)) maxn = v;
            cur = g.out_edge[(u64)v * 4 + order_base(g.out_order[v], 0)];
        }
        const u64 bytes = (total + 3) / 4;
        if (pass == 0) {
            atomicAdd(&counters[0], 1ull);
            atomicAdd(&counters[1], (unsigned long long)bytes);
            if (nlong) atomicAdd(&counters[2], (unsigned long long)nlong);
            continue;
        }
        const u64 id = old_edges + atomicAdd(&counters[0], 1ull);
        const u64 off = old_pool + atomicAdd(&counters[1], (unsigned long long)bytes);
        // e1.seq ++ e2.seq ++ ...  (Graph.scala:225)
        u64 w = 0;
        u32 acc = 0;
        bool shared = false;                           // the byte being assembled also holds bases of a long piece
        cur = (u32)e;
        for (u64 guard = 0; guard <= old_edges; guard++) {
            const u64 so = g.e_off[cur], sl = g.e_len[cur];
            if (sl >= LONG_PIECE) {
                if (w & 3) { pool_or(g.pool, off + (w >> 2), acc); acc = 0; }      // the piece starts inside this byte
                long_pieces[atomicAdd(&counters[2], 1ull)] = LongPiece{so, off, w, sl};
                w += sl;
                shared = (w & 3) != 0;                                              // ... and ends inside that one
            } else {
                for (u64 i = 0; i < sl; i++) {
                    acc |= (u32)pool_get(g.pool, so, i) << ((w & 3) * 2);
                    if ((w & 3) == 3) {
                        if (shared) pool_or(g.pool, off + (w >> 2), acc); else g.pool[off + (w >> 2)] = (uint8_t)acc;
                        acc = 0; shared = false;
                    }
                    w++;
                }
            }
            g.e_alive[cur] = 0;                        // removeEdge(e1); removeEdge(e2)  :223-224
            const u32 v = g.e_end[cur];
            if (cls[v] != 1) break;
            cur = g.out_edge[(u64)v * 4 + order_base(g.out_order[v], 0)];
        }
        if (w & 3) { if (shared) pool_or(g.pool, off + (w >> 2), acc); else g.pool[off + (w >> 2)] = (uint8_t)acc; }
        g.e_start[id] = s;                             // addEdge(e1.start, e2.end, ...)   :225
        g.e_end[id] = g.e_end[cur];
        g.e_len[id] = total;
        g.e_off[id] = off;
        g.e_first[id] = g.e_first[e];
        g.e_alive[id] = 1;
        g.out_edge[(u64)s * 4 + g.e_first[e]] = (u32)id;
        merged_key[(u64)s * 4 + g.e_first[e]] = maxn;
    }
}
// Remove interior / self-loop / isolated nodes and whatever edges still hang off them (self-loops,
// perfect cycles), and replay the out-edge insertion order at kept nodes: every merge step is
// `outEdgeIds -= b` then `+= b` (removeEdge :192, addEdge :180), i.e. b moves to the end, and the
// last step of a chain happens when its largest interior node (ascending k-mer order) is visited.
__global__ __launch_bounds__(BLOCK) void k_simplify_finish(GraphView g, const uint8_t *cls, const u32 *merged_key) {
    for (u64 n = (u64)blockIdx.x * BLOCK + threadIdx.x; n < g.n_nodes; n += (u64)gridDim.x * BLOCK) {
        if (!g.node_alive[n]) continue;
        const u32 o = g.out_order[n];
        if (cls[n] != 0) {
            for (int i = 0; i < order_count(o); i++) {
                const int b = order_base(o, i);
                const u32 e = g.out_edge[n * 4 + b];
                if (e != NONE) g.e_alive[e] = 0;
                g.out_edge[n * 4 + b] = NONE;
            }
            g.out_order[n] = 0;
            g.in_deg[n] = 0;
            g.node_alive[n] = 0;                       // removeNode :216,:227
            continue;
        }
        u32 r = 0;
        int mb[4], nm = 0;
        for (int i = 0; i < order_count(o); i++) {
            const int b = order_base(o, i);
            if (merged_key[n * 4 + b] == NONE) r = order_append(r, b);
            else mb[nm++] = b;
        }
        for (int i = 1; i < nm; i++)                   // insertion sort of <= 4 entries by chain key
            for (int j = i; j > 0 && node_less(g, merged_key[n * 4 + mb[j]], merged_key[n * 4 + mb[j - 1]]); j--) {
                int tmp = mb[j]; mb[j] = mb[j - 1]; mb[j - 1] = tmp;
            }
        for (int i = 0; i < nm; i++) r = order_append(r, mb[i]);
        g.out_order[n] = r;
    }
}

// Graph.removeBubbles (Graph.scala:125-149) — per node, on the out-edges in insertion order
__global__ __launch_bounds__(BLOCK) void k_bubbles(GraphView g) {
    for (u64 n = (u64)blockIdx.x * BLOCK + threadIdx.x; n < g.n_nodes; n += (u64)gridDim.x * BLOCK) {
        if (!g.node_alive[n]) continue;
        const u32 o = g.out_order[n];
        const int cnt = order_count(o);
        if (cnt < 2) continue;
        u32 out[4];
        bool rem[4] = {false, false, false, false};
        for (int i = 0; i < cnt; i++) out[i] = g.out_edge[n * 4 + order_base(o, i)];
        for (int i = 0; i < cnt; i++) {
            if (rem[i]) continue;                                      // `if !toRemove(out(i))`  :142
            for (int j = i + 1; j < cnt; j++) {
                const u64 la = g.e_len[out[i]], lb = g.e_len[out[j]];
                const u64 d = la > lb ? la - lb : lb - la, mx = la > lb ? la : lb;
                if (g.e_end[out[i]] == g.e_end[out[j]] && d * 5 < mx) rem[j] = true;   // similar :121-123
            }
        }
        u32 r = 0;
        for (int i = 0; i < cnt; i++) {
            const int b = order_base(o, i);
            if (rem[i]) {                                              // removeEdge :191-195
                g.e_alive[out[i]] = 0;
                atomicSub(&g.in_deg[g.e_end[out[i]]], 1u);
                g.out_edge[n * 4 + b] = NONE;
            } else {
                r = order_append(r, b);
            }
        }
        g.out_order[n] = r;
    }
}

template <int W>
__global__ __launch_bounds__(BLOCK) void k_remove_edges(GraphView g, const u64 *lo, const u64 *hi, const uint8_t *base, u64 n,
                                                        unsigned long long *removed) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        Kmer<W> x;
        if constexpr (W == 1) x = Kmer<1>{lo[i]};
        else x = Kmer<2>{lo[i], hi[i]};
        const u32 v = node_find<W>(g, x);
        const int b = base[i] & 3;
        if (v == NONE || !g.node_alive[v]) continue;
        const u32 e = atomicExch(&g.out_edge[(u64)v * 4 + b], NONE);
        if (e == NONE) continue;
        g.e_alive[e] = 0;
        atomicSub(&g.in_deg[g.e_end[e]], 1u);
        u32 old = g.out_order[v], seen;
        do {
            seen = old;
            old = atomicCAS(&g.out_order[v], seen, order_remove(seen, b));
        } while (old != seen);
        atomicAdd(removed, 1ull);
    }
}

__global__ __launch_bounds__(BLOCK) void k_count_live(GraphView g, unsigned long long *out /* nodes, edges, len */) {
    __shared__ unsigned long long s[3];
    if (threadIdx.x < 3) s[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long a = 0, b = 0, c = 0;
    const u64 tid = (u64)blockIdx.x * BLOCK + threadIdx.x, stride = (u64)gridDim.x * BLOCK;
    for (u64 n = tid; n < g.n_nodes; n += stride) a += g.node_alive[n];
    for (u64 e = tid; e < g.n_edges; e += stride) if (g.e_alive[e]) { b++; c += g.e_len[e]; }
    if (a) atomicAdd(&s[0], a);
    if (b) atomicAdd(&s[1], b);
    if (c) atomicAdd(&s[2], c);
    __syncthreads();
    if (threadIdx.x < 3 && s[threadIdx.x]) atomicAdd(&out[threadIdx.x], s[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------
// components (Graph.scala:54-72) by min-label hooking + pointer jumping; retain (:161-165)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 cc_root(const u32 *parent, u32 v) {
    u32 p = parent[v];
    while (p != v) { v = p; p = parent[v]; }
    return v;
}
__global__ __launch_bounds__(BLOCK) void k_cc_init(GraphView g, u32 *parent) {
    for (u64 n = (u64)blockIdx.x * BLOCK + threadIdx.x; n < g.n_nodes; n += (u64)gridDim.x * BLOCK) parent[n] = (u32)n;
}
// One pass over the edges: a lock-free union (the larger root is hooked under the smaller by a CAS on the root's own word; a
// failed CAS hands back the word's true value and the walk continues from there) with path halving on the way up.  Pointers
// only ever go to smaller ids, so there are no cycles, a stale read is an older ancestor (a longer walk, never a wrong one),
// and the root a component ends with is its smallest node id — the label the round-2 form (min-label hooking + full
// compression, repeated until nothing moved: 5-7 rounds of two kernels and a host round trip each, 7 ms at C3) converged to.
// What bounds the pass is not its reads (12.8e9 requests/s by PMC, a quarter of the random-read ceiling) but same-address
// atomics: when two large trees meet, thousands of edges want the same root's word at once, and a CAS that fails costs what one
// that succeeds does (~88 per microsecond on one address, chip-wide).  So a device-scope LOAD looks first and the CAS is only
// sent when the word still names a root: k_cc_link 5.6 -> ~2 ms, retainLargest at C3 8.4 -> 4.7 ms
// (profiles/r03/ab_components_find_variants.txt).
// V (A/B, "cc_find"): 3 = path halving, every hop writes, and the link looks before its CAS (the default); 0 = the same without
// the look; 1 = nothing is written on the way; 2 = only the node the find started from is pointed at the root it found
template <int V> __device__ __forceinline__ u32 cc_find(u32 *parent, u32 v) {
    const u32 v0 = v;
    u32 p = parent[v];
    while (p != v) {
        const u32 gp = parent[p];
        if (gp == p) { v = p; break; }
        if (V == 0 || V == 3) parent[v] = gp;     // (v is not a root and never becomes one again: no CAS targets this word)
        v = gp; p = parent[v];
    }
    if (V == 2 && v != v0 && parent[v0] != v) parent[v0] = v;
    return v;
}
template <int V> __global__ __launch_bounds__(BLOCK) void k_cc_link(GraphView g, u32 *parent) {
    for (u64 e = (u64)blockIdx.x * BLOCK + threadIdx.x; e < g.n_edges; e += (u64)gridDim.x * BLOCK) {
        if (!g.e_alive[e]) continue;
        u32 ru = cc_find<V>(parent, g.e_start[e]), rv = cc_find<V>(parent, g.e_end[e]);
        while (ru != rv) {
            if (ru < rv) { const u32 x = ru; ru = rv; rv = x; }
            if (V == 3) {
                // look before the CAS (a device-scope load: another XCD's hook is visible): when two large trees meet, thousands of
                // edges want the same root's word, and a failed same-address CAS costs what a successful one does
                const u32 cur = __hip_atomic_load(&parent[ru], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur != ru) { ru = cc_find<V>(parent, cur); rv = cc_find<V>(parent, rv); continue; }
            }
            const u32 old = atomicCAS(&parent[ru], ru, rv);
            if (old == ru) break;
            ru = cc_find<V>(parent, old);
            rv = cc_find<V>(parent, rv);
        }
    }
}
// Component sizes and summed edge lengths.  A giant component means millions of increments of ONE counter, and same-address
// atomics retire at ~88 per microsecond chip-wide (143 ms of the 169 ms retain step at C3 in round 1; 7.7-10 ms in round 2
// with a per-wave carry: still 4.4e5 atomics on the giant root).  Now every WORKGROUP keeps a (root -> partial sum) table in
// LDS for its whole share of the nodes / edges: lanes of a wave that share a root are combined first (ballot), the wave adds
// its partial to the LDS table, and the table goes out once, at the end — one global atomic per (workgroup, root): 2048 on the
// giant root.  A root that finds the LDS table full goes straight to memory (distinct small components: no contention there).
static constexpr u32 CC_TAB = 1024;
template <class T> struct CcTable {
    u32 root[CC_TAB];
    T sum[CC_TAB];
    __device__ __forceinline__ void clear() { for (u32 i = threadIdx.x; i < CC_TAB; i += BLOCK) { root[i] = NONE; sum[i] = 0; } }
    // one lane per call and root
    __device__ __forceinline__ void add(u32 r, T v, T *global) {
        u32 h = hash32(r) & (CC_TAB - 1);
        for (u32 n = 0; n < 16; n++) {
            u32 cur = root[h];
            if (cur == NONE) cur = atomicCAS(&root[h], NONE, r);
            if (cur == NONE || cur == r) { atomicAdd(&sum[h], v); return; }
            h = (h + 1) & (CC_TAB - 1);
        }
        atomicAdd(&global[r], v);
    }
    __device__ __forceinline__ void flush(T *global) {
        for (u32 i = threadIdx.x; i < CC_TAB; i += BLOCK) if (root[i] != NONE && sum[i]) atomicAdd(&global[root[i]], sum[i]);
    }
};
// (also the one compression pass after k_cc_link: every live node's word becomes its root — a walker that passes through a
// word already compressed lands on the same root)
__global__ __launch_bounds__(BLOCK) void k_cc_sizes(GraphView g, u32 *parent, u32 *size, unsigned long long *ncomp) {
    __shared__ CcTable<u32> tab;
    __shared__ u32 s_roots;
    if (threadIdx.x == 0) s_roots = 0;
    tab.clear();
    __syncthreads();
    const int lane = threadIdx.x & 63;
    u32 nroots = 0;
    for (u64 n0 = (u64)blockIdx.x * BLOCK + (threadIdx.x & ~63u); n0 < g.n_nodes; n0 += (u64)gridDim.x * BLOCK) {   // wave-uniform trip count
        const u64 n = n0 + lane;
        const bool active = n < g.n_nodes && g.node_alive[n];
        u32 root = 0xffffffffu;
        if (active) { root = cc_root(parent, (u32)n); parent[n] = root; }
        unsigned long long todo = __ballot(active);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const u32 lr = __shfl(root, leader);
            const unsigned long long same = __ballot(active && root == lr);
            if (lane == leader) tab.add(lr, (u32)__popcll(same), size);
            todo &= ~same;
        }
        nroots += (u32)__popcll(__ballot(active && root == (u32)n));
    }
    if (lane == 0 && nroots) atomicAdd(&s_roots, nroots);
    __syncthreads();
    tab.flush(size);
    if (threadIdx.x == 0 && s_roots) atomicAdd(ncomp, (unsigned long long)s_roots);
}
// summed out-edge length per component (GraphBuilder.scala:44-46: comp.flatMap(_.outEdges.values).map(_.seq.size).sum): every
// live edge adds its length to the root of its start node
__global__ __launch_bounds__(BLOCK) void k_cc_edge_len(GraphView g, const u32 *parent, unsigned long long *len) {
    __shared__ CcTable<unsigned long long> tab;
    tab.clear();
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (u64 e0 = (u64)blockIdx.x * BLOCK + (threadIdx.x & ~63u); e0 < g.n_edges; e0 += (u64)gridDim.x * BLOCK) {
        const u64 e = e0 + lane;
        const bool active = e < g.n_edges && g.e_alive[e];
        const u32 root = active ? parent[g.e_start[e]] : 0xffffffffu;
        const unsigned long long mylen = active ? g.e_len[e] : 0ull;
        unsigned long long todo = __ballot(active);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const u32 lr = __shfl(root, leader);
            const bool mine = active && root == lr;
            const unsigned long long same = __ballot(mine);
            unsigned long long part = mine ? mylen : 0ull;
            for (int d = 32; d; d >>= 1) part += __shfl_down(part, d);
            const unsigned long long tot = __shfl(part, 0);           // (lane 0 holds the wave's sum of this root's lengths)
            if (lane == leader) tab.add(lr, tot, len);
            todo &= ~same;
        }
    }
    __syncthreads();
    tab.flush(len);
}
// one (node count, summed out-edge length) pair per component, in root order
__global__ __launch_bounds__(BLOCK) void k_cc_collect(GraphView g, const u32 *parent, const u32 *size, const unsigned long long *len,
                                                      u32 *out_nodes, unsigned long long *out_len, unsigned long long *cursor) {
    __shared__ u32 lds4[BLOCK / 64];
    __shared__ unsigned long long s_base;
    const u64 ngroups = (g.n_nodes + BLOCK - 1) / BLOCK;
    for (u64 grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const u64 n = grp * BLOCK + threadIdx.x;
        const bool root = n < g.n_nodes && g.node_alive[n] && parent[n] == (u32)n;
        const u64 o = block_reserve(root ? 1u : 0u, cursor, lds4, &s_base);
        if (root) { out_nodes[o] = size[n]; out_len[o] = len[n]; }
    }
}
// order-independent checksums of the canonical serialisation: nodes = k-mers; edges = (start k-mer,
// end k-mer, length, every base) — ids, array order and pool offsets do not enter
__global__ __launch_bounds__(BLOCK) void k_graph_checksum(GraphView g, unsigned long long *out /* nodes, edges */) {
    u64 cn = 0, ce = 0;
    const u64 tid = (u64)blockIdx.x * BLOCK + threadIdx.x, stride = (u64)gridDim.x * BLOCK;
    for (u64 n = tid; n < g.n_nodes; n += stride)
        if (g.node_alive[n]) cn += mix64(g.node_lo[n] ^ mix64(g.node_hi[n] ^ 0x13198a2e03707344ULL));
    for (u64 e = tid; e < g.n_edges; e += stride) {
        if (!g.e_alive[e]) continue;
        const u32 s = g.e_start[e], t = g.e_end[e];
        u64 h = mix64(g.node_lo[s] ^ mix64(g.node_hi[s] ^ 1)) + 3 * mix64(g.node_lo[t] ^ mix64(g.node_hi[t] ^ 2)) + 5 * mix64(g.e_len[e]);
        const u64 off = g.e_off[e], len = g.e_len[e], nbytes = (len + 3) / 4;
        for (u64 i = 0; i < nbytes; i++) {
            u32 b = g.pool[off + i];
            if (i == nbytes - 1 && (len & 3)) b &= (1u << ((len & 3) * 2)) - 1u;      // bits after the last base are not content
            h = mix64(h ^ (b + 0x9e3779b97f4a7c15ULL * (i + 1)));
        }
        ce += h;
    }
    for (int d = 32; d; d >>= 1) { cn += __shfl_down(cn, d); ce += __shfl_down(ce, d); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], cn); atomicAdd(&out[1], ce); }
}
__global__ __launch_bounds__(BLOCK) void k_cc_max(GraphView g, const u32 *size, u32 *best) {
    u32 m = 0;
    for (u64 n = (u64)blockIdx.x * BLOCK + threadIdx.x; n < g.n_nodes; n += (u64)gridDim.x * BLOCK)
        if (g.node_alive[n]) m = max(m, size[n]);
    for (int d = 32; d; d >>= 1) m = max(m, (u32)__shfl_down(m, d));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(best, m);
}
// among the components of maximal size pick the one holding the smallest k-mer: stage 0 min hi,
// stage 1 min lo among those, stage 2 record its root (one atomic per wave, see k_cc_sizes)
__global__ __launch_bounds__(BLOCK) void k_cc_pick(GraphView g, const u32 *parent, const u32 *size, const u32 *best_p, int stage,
                                                   unsigned long long *mins /* hi, lo */, u32 *winner) {
    unsigned long long m = ~0ull;
    const u32 best = *best_p;                       // (k_cc_max's result: stays on the device)
    const unsigned long long min_hi = stage >= 1 ? mins[0] : 0ull, min_lo = stage == 2 ? mins[1] : 0ull;
    for (u64 n = (u64)blockIdx.x * BLOCK + threadIdx.x; n < g.n_nodes; n += (u64)gridDim.x * BLOCK) {
        if (!g.node_alive[n] || size[parent[n]] != best) continue;
        if (stage == 0) m = min(m, (unsigned long long)g.node_hi[n]);
        else if (stage == 1) { if (g.node_hi[n] == min_hi) m = min(m, (unsigned long long)g.node_lo[n]); }
        else if (g.node_hi[n] == min_hi && g.node_lo[n] == min_lo) *winner = parent[n];
    }
    if (stage < 2) {
        for (int d = 32; d; d >>= 1) m = min(m, (unsigned long long)__shfl_down(m, d));
        if ((threadIdx.x & 63) == 0 && m != ~0ull) atomicMin(&mins[stage], m);
    }
}
__global__ __launch_bounds__(BLOCK) void k_retain(GraphView g, const u32 *parent, const u32 *winner_p) {
    const u32 winner = *winner_p;
    if (winner == NONE) return;                     // (nothing selected: the host reports it, the graph stays as it was)
    const u64 tid = (u64)blockIdx.x * BLOCK + threadIdx.x, stride = (u64)gridDim.x * BLOCK;
    for (u64 e = tid; e < g.n_edges; e += stride)
        if (g.e_alive[e] && !(parent[g.e_start[e]] == winner && parent[g.e_end[e]] == winner)) g.e_alive[e] = 0;
    for (u64 n = tid; n < g.n_nodes; n += stride)
        if (g.node_alive[n] && parent[n] != winner) {
            g.node_alive[n] = 0;
            g.out_order[n] = 0;
            g.in_deg[n] = 0;
            for (int b = 0; b < 4; b++) g.out_edge[n * 4 + b] = NONE;
        }
}

// ---------------------------------------------------------------------------------------------
// export
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_export_nodes(GraphView g, u64 *lo, u64 *hi, unsigned long long *cursor) {
    __shared__ u32 lds4[BLOCK / 64];
    __shared__ unsigned long long s_base;
    const u64 ngroups = (g.n_nodes + BLOCK - 1) / BLOCK;
    for (u64 grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const u64 n = grp * BLOCK + threadIdx.x;
        const bool live = n < g.n_nodes && g.node_alive[n];
        u64 o = block_reserve(live ? 1u : 0u, cursor, lds4, &s_base);
        if (live) { lo[o] = g.node_lo[n]; hi[o] = g.node_hi[n]; }
    }
}
__global__ __launch_bounds__(BLOCK) void k_export_edges(GraphView g, u64 *slo, u64 *shi, u64 *elo, u64 *ehi, i64 *len, i64 *off,
                                                        uint8_t *seq, unsigned long long *cursors /* [0] edges [1] bytes */) {
    __shared__ u32 lds4[BLOCK / 64];
    __shared__ unsigned long long s_base;
    const u64 ngroups = (g.n_edges + BLOCK - 1) / BLOCK;
    for (u64 grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const u64 e = grp * BLOCK + threadIdx.x;
        const bool live = e < g.n_edges && g.e_alive[e];
        u64 o = block_reserve(live ? 1u : 0u, &cursors[0], lds4, &s_base);
        if (!live) continue;
        const u64 bytes = (g.e_len[e] + 3) / 4;
        const u64 so = atomicAdd(&cursors[1], (unsigned long long)bytes);
        const u32 s = g.e_start[e], t = g.e_end[e];
        slo[o] = g.node_lo[s]; shi[o] = g.node_hi[s];
        elo[o] = g.node_lo[t]; ehi[o] = g.node_hi[t];
        len[o] = (i64)g.e_len[e];
        off[o] = (i64)so;
        const u64 src = g.e_off[e];
        for (u64 i = 0; i < bytes; i++) seq[so + i] = g.pool[src + i];
    }
}
template <int W> __global__ void k_out_order(GraphView g, u64 lo, u64 hi, int *out5) {
    Kmer<W> x;
    if constexpr (W == 1) x = Kmer<1>{lo};
    else x = Kmer<2>{lo, hi};
    const u32 v = node_find<W>(g, x);
    if (v == NONE || !g.node_alive[v]) { out5[0] = -1; return; }
    const u32 o = g.out_order[v];
    out5[0] = order_count(o);
    for (int i = 0; i < order_count(o); i++) out5[1 + i] = order_base(o, i);
}

// ---------------------------------------------------------------------------------------------
// Graph.getGraphMap (Graph.scala:90-119): every k-mer of the graph -> where it sits.  A node's k-mer -> NodeGraphPosition(id);
// for an edge, the k-mers at distance 1 .. len-1 from its start node — windows of (start.seq ++ edge.seq) — ->
// EdgeGraphPosition(id, dist); the window at distance len is the end node's own k-mer and is not added (:108-113).
// One lane per (edge, distance): the window is cut out of the start k-mer and the 2-bit pool directly, no rolling.
// ---------------------------------------------------------------------------------------------
static constexpr u64 POS_EDGE = 1ull << 63;
__device__ __forceinline__ int path_base(const GraphView &g, u32 s, u64 off, int k, u64 i) {      // base i of start.seq ++ edge.seq
    if (i < (u64)k) return (int)(((i < 32 ? g.node_lo[s] : g.node_hi[s]) >> (2 * (i & 31))) & 3);
    return pool_get(g.pool, off, i - (u64)k);
}
template <int W> __device__ __forceinline__ Kmer<W> path_window(const GraphView &g, u32 s, u64 off, int k, u64 dist) {
    Kmer<W> x{};
    for (int i = 0; i < k; i++) {
        const u64 b = (u64)path_base(g, s, off, k, dist + (u64)i);
        if (i < 32) x.lo |= b << (2 * i);
        else if constexpr (W == 2) x.hi |= b << (2 * (i - 32));
    }
    return x;
}
// pass 0: entries per live edge (len - 1) -> its first entry through a block-aggregated cursor; pass 1 (flat): fill
__global__ __launch_bounds__(BLOCK) void k_pos_reserve(GraphView g, unsigned long long *first_entry, unsigned long long *cursor) {
    __shared__ u32 lds4[BLOCK / 64];
    __shared__ unsigned long long s_base;
    const u64 ngroups = (g.n_edges + BLOCK - 1) / BLOCK;
    for (u64 grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const u64 e = grp * BLOCK + threadIdx.x;
        const u64 cnt = e < g.n_edges && g.e_alive[e] ? g.e_len[e] - 1 : 0;
        const u32 small = cnt < (1u << 22) ? (u32)cnt : 0u;
        u64 o = block_reserve(small, cursor, lds4, &s_base);
        if (e < g.n_edges) first_entry[e] = small == cnt ? o : atomicAdd(cursor, (unsigned long long)cnt);
    }
}
template <int W>
__global__ __launch_bounds__(BLOCK) void k_pos_fill(GraphView g, int k, const unsigned long long *__restrict__ first_entry, u64 node_entries,
                                                    u64 *lo, u64 *hi, u64 *val) {
    const u64 tid = (u64)blockIdx.x * BLOCK + threadIdx.x, stride = (u64)gridDim.x * BLOCK;
    // short edges: one lane walks the edge; long ones (> 64 entries): the whole grid strides over their distances below
    for (u64 e = tid; e < g.n_edges; e += stride) {
        if (!g.e_alive[e]) continue;
        const u64 cnt = g.e_len[e] - 1;
        if (cnt > 64) continue;
        const u32 s = g.e_start[e];
        const u64 off = g.e_off[e], at = node_entries + first_entry[e];
        for (u64 d = 1; d <= cnt; d++) {
            const Kmer<W> x = path_window<W>(g, s, off, k, d);
            lo[at + d - 1] = x.lo;
            if constexpr (W == 2) hi[at + d - 1] = x.hi;
            val[at + d - 1] = POS_EDGE | ((u64)e << 32) | d;
        }
    }
}
template <int W>
__global__ __launch_bounds__(BLOCK) void k_pos_fill_long(GraphView g, int k, const unsigned long long *__restrict__ first_entry, u64 node_entries,
                                                         u64 *lo, u64 *hi, u64 *val) {
    // one WORKGROUP per long edge at a time (grid-stride over edges), its lanes over the distances
    for (u64 e = blockIdx.x; e < g.n_edges; e += gridDim.x) {
        if (!g.e_alive[e]) continue;
        const u64 cnt = g.e_len[e] - 1;
        if (cnt <= 64) continue;
        const u32 s = g.e_start[e];
        const u64 off = g.e_off[e], at = node_entries + first_entry[e];
        for (u64 d = 1 + threadIdx.x; d <= cnt; d += BLOCK) {
            const Kmer<W> x = path_window<W>(g, s, off, k, d);
            lo[at + d - 1] = x.lo;
            if constexpr (W == 2) hi[at + d - 1] = x.hi;
            val[at + d - 1] = POS_EDGE | ((u64)e << 32) | d;
        }
    }
}
__global__ __launch_bounds__(BLOCK) void k_pos_nodes(GraphView g, u64 *lo, u64 *hi, u64 *val, unsigned long long *cursor) {
    __shared__ u32 lds4[BLOCK / 64];
    __shared__ unsigned long long s_base;
    const u64 ngroups = (g.n_nodes + BLOCK - 1) / BLOCK;
    for (u64 grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const u64 n = grp * BLOCK + threadIdx.x;
        const bool live = n < g.n_nodes && g.node_alive[n];
        const u64 o = block_reserve(live ? 1u : 0u, cursor, lds4, &s_base);
        if (live) { lo[o] = g.node_lo[n]; if (hi) hi[o] = g.node_hi[n]; val[o] = (u64)n; }
    }
}

// ---------------------------------------------------------------------------------------------
// point edits by id (the simplifier's node split: addNode, replaceStart, replaceEnd — Graph.scala:172-176, 197-209)
// ---------------------------------------------------------------------------------------------
__global__ void k_replace_start(GraphView g, u32 e, u32 ns, int *status) {
    if (e >= g.n_edges || !g.e_alive[e] || ns >= g.n_nodes || !g.node_alive[ns]) { *status = 1; return; }
    const u32 old = g.e_start[e];
    const int b = g.e_first[e];
    // edge.start.outEdgeIds -= edge.seq(0)            :200
    if (g.out_edge[(u64)old * 4 + b] == e) { g.out_edge[(u64)old * 4 + b] = NONE; g.out_order[old] = order_remove(g.out_order[old], b); }
    // newStart.outEdgeIds += edge.seq(0) -> edge.id   :201  (an immutable Map: an existing key keeps its place, its value is replaced)
    const u32 prev = g.out_edge[(u64)ns * 4 + b];
    g.out_edge[(u64)ns * 4 + b] = e;
    if (prev == NONE) g.out_order[ns] = order_append(g.out_order[ns], b);
    g.e_start[e] = ns;                                  // edges(edge.id) = new Edge(edge.id, newStart.id, ...)  :199
    *status = 0;
}
__global__ void k_replace_end(GraphView g, u32 e, u32 ne, int *status) {
    if (e >= g.n_edges || !g.e_alive[e] || ne >= g.n_nodes || !g.node_alive[ne]) { *status = 1; return; }
    const u32 old = g.e_end[e];
    if (g.in_deg[old]) g.in_deg[old]--;                 // edge.end.inEdgeIds -= edge.id   :207
    g.in_deg[ne]++;                                     // newEnd.inEdgeIds += edge.id     :208
    g.e_end[e] = ne;
    *status = 0;
}
__global__ void k_add_node(GraphView g, u32 n, u64 lo, u64 hi) {
    g.node_lo[n] = lo; g.node_hi[n] = hi;
    g.node_alive[n] = 1;
    g.out_order[n] = 0; g.in_deg[n] = 0;
    for (int b = 0; b < 4; b++) g.out_edge[(u64)n * 4 + b] = NONE;
}
__global__ __launch_bounds__(BLOCK) void k_nodes_by_id(GraphView g, const u32 *ids, u64 n, u64 *lo, u64 *hi, uint8_t *alive, u32 *in_deg, u32 *out_deg) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        const u32 v = ids[i];
        const bool ok = v < g.n_nodes;
        lo[i] = ok ? g.node_lo[v] : 0; hi[i] = ok ? g.node_hi[v] : 0;
        alive[i] = ok ? g.node_alive[v] : 0;
        in_deg[i] = ok ? g.in_deg[v] : 0;
        out_deg[i] = ok ? (u32)order_count(g.out_order[v]) : 0;
    }
}
__global__ __launch_bounds__(BLOCK) void k_edges_by_id(GraphView g, const u32 *ids, u64 n, u32 *start, u32 *end, u64 *len, uint8_t *first, uint8_t *alive) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        const u32 e = ids[i];
        const bool ok = e < g.n_edges;
        start[i] = ok ? g.e_start[e] : NONE; end[i] = ok ? g.e_end[e] : NONE;
        len[i] = ok ? g.e_len[e] : 0; first[i] = ok ? g.e_first[e] : 0; alive[i] = ok ? g.e_alive[e] : 0;
    }
}
template <int W> __global__ void k_node_lookup(GraphView g, u64 lo, u64 hi, int base, u32 *out2) {
    Kmer<W> x;
    if constexpr (W == 1) x = Kmer<1>{lo};
    else x = Kmer<2>{lo, hi};
    // node_find returns the first index entry with this k-mer; after a node split several nodes share a sequence: the live one
    // with the smallest id is reported
    u32 best = NONE;
    u64 i = slot_hash(x) & g.nidx_mask;
    for (u64 p = 0; p <= g.nidx_mask; p++) {
        const u32 n = g.nidx[i];
        if (n == NONE) break;
        if (g.node_alive[n] && node_kmer<W>(g, n) == x && n < best) best = n;
        i = (i + 1) & g.nidx_mask;
    }
    out2[0] = best;
    out2[1] = best != NONE && base >= 0 && base < 4 ? g.out_edge[(u64)best * 4 + base] : NONE;
}
__global__ void k_nidx_insert(GraphView g, u32 n, u64 h) {
    u64 i = h & g.nidx_mask;
    while (atomicCAS(&g.nidx[i], NONE, n) != NONE) i = (i + 1) & g.nidx_mask;
}

// =============================================================================================
// host side
// =============================================================================================
int ggrid(const gk_ctx *ctx, u64 items) {
    u64 blocks = (items + BLOCK - 1) / BLOCK;
    if (blocks < 1) blocks = 1;
    return (int)std::min<u64>(blocks, (u64)ctx->cu_count * 8);
}

template <class T> static hipError_t dev_grow(gk_ctx *ctx, T **p, u64 old_n, u64 new_n, hipStream_t st) {
    T *np_ = nullptr;
    hipError_t e = hipMalloc((void **)&np_, std::max<u64>(new_n, 1) * sizeof(T));
    if (e != hipSuccess) return e;
    if (*p && old_n) e = hipMemcpyAsync(np_, *p, old_n * sizeof(T), hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (*p) (void)hipFree(*p);
    *p = np_;
    return e;
}

// Node and edge arrays live in ONE allocation each (hipMalloc costs ~1 ms apiece at these sizes: twelve of them were 10 % of
// gk_graph_build at C3); the view's pointers are carved out of the blobs, 256-byte aligned.
static size_t al256g(size_t v) { return (v + 255) & ~(size_t)255; }
struct NodeCarve { size_t lo, hi, alive, out_edge, out_order, in_deg, total; };
struct EdgeCarve { size_t start, end, len, off, alive, first, total; };
static NodeCarve node_carve(u64 c) {
    NodeCarve k{};
    size_t o = 0;
    k.lo = o; o += al256g(c * 8);
    k.hi = o; o += al256g(c * 8);
    k.out_edge = o; o += al256g(c * 16);
    k.out_order = o; o += al256g(c * 4);
    k.in_deg = o; o += al256g(c * 4);
    k.alive = o; o += al256g(c);
    k.total = o;
    return k;
}
static EdgeCarve edge_carve(u64 c) {
    EdgeCarve k{};
    size_t o = 0;
    k.len = o; o += al256g(c * 8);
    k.off = o; o += al256g(c * 8);
    k.start = o; o += al256g(c * 4);
    k.end = o; o += al256g(c * 4);
    k.alive = o; o += al256g(c);
    k.first = o; o += al256g(c);
    k.total = o;
    return k;
}
static void node_view(GraphView &v, char *blob, const NodeCarve &k) {
    v.node_lo = (u64 *)(blob + k.lo); v.node_hi = (u64 *)(blob + k.hi); v.node_alive = (uint8_t *)(blob + k.alive);
    v.out_edge = (u32 *)(blob + k.out_edge); v.out_order = (u32 *)(blob + k.out_order); v.in_deg = (u32 *)(blob + k.in_deg);
}
static void edge_view(GraphView &v, char *blob, const EdgeCarve &k) {
    v.e_start = (u32 *)(blob + k.start); v.e_end = (u32 *)(blob + k.end); v.e_len = (u64 *)(blob + k.len); v.e_off = (u64 *)(blob + k.off);
    v.e_alive = (uint8_t *)(blob + k.alive); v.e_first = (uint8_t *)(blob + k.first);
}

static void graph_free_arrays(gk_graph *g) {
    gk_ctx *ctx = g->ctx;
    GraphView &v = g->v;
    for (void *p : {g->node_blob, g->edge_blob, (void *)v.pool, (void *)v.nidx}) if (p) (void)hipFree(p);
    g->node_blob = g->edge_blob = nullptr;
    v = GraphView{};
}

static int graph_alloc_nodes(gk_graph *g, u64 n) {
    gk_ctx *ctx = g->ctx;
    GraphView &v = g->v;
    const u64 c = std::max<u64>(n, 1);
    const NodeCarve k = node_carve(c);
    GK_HIP(ctx, hipMalloc(&g->node_blob, k.total));
    node_view(v, (char *)g->node_blob, k);
    // out_edge <- NONE (0xff..), out_order / in_deg / alive <- 0: two memsets over the two contiguous stretches
    GK_HIP(ctx, hipMemsetAsync(v.out_edge, 0xff, c * 16, ctx->stream));
    GK_HIP(ctx, hipMemsetAsync((char *)g->node_blob + k.out_order, 0, k.total - k.out_order, ctx->stream));
    v.n_nodes = n;
    g->node_cap = c;
    return GK_OK;
}
static int graph_alloc_edges(gk_graph *g, u64 n) {
    gk_ctx *ctx = g->ctx;
    GraphView &v = g->v;
    const u64 c = std::max<u64>(n, 1);
    const EdgeCarve k = edge_carve(c);
    GK_HIP(ctx, hipMalloc(&g->edge_blob, k.total));
    edge_view(v, (char *)g->edge_blob, k);
    GK_HIP(ctx, hipMemsetAsync(v.e_alive, 0, c, ctx->stream));
    v.n_edges = n;
    g->edge_cap = c;
    return GK_OK;
}
static int graph_grow_edges(gk_graph *g, u64 new_n) {
    gk_ctx *ctx = g->ctx;
    GraphView &v = g->v;
    if (new_n <= g->edge_cap) { return GK_OK; }
    const u64 old = v.n_edges;
    const EdgeCarve k = edge_carve(new_n);
    void *blob = nullptr;
    GK_HIP(ctx, hipMalloc(&blob, k.total));
    GraphView nv = v;
    edge_view(nv, (char *)blob, k);
    hipError_t e = hipMemsetAsync(nv.e_alive, 0, new_n, ctx->stream);
    if (e == hipSuccess && old) {
        e = hipMemcpyAsync(nv.e_start, v.e_start, old * 4, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(nv.e_end, v.e_end, old * 4, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(nv.e_len, v.e_len, old * 8, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(nv.e_off, v.e_off, old * 8, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(nv.e_alive, v.e_alive, old, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(nv.e_first, v.e_first, old, hipMemcpyDeviceToDevice, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipFree(blob); return hip_fail(ctx, e, "graph: growing the edge arrays"); }
    (void)hipFree(g->edge_blob);
    g->edge_blob = blob;
    v = nv;
    g->edge_cap = new_n;
    return GK_OK;
}

int graph_refresh_counts(gk_graph *g) {
    gk_ctx *ctx = g->ctx;
    g->epoch++;                  // (every edit of the graph ends here or in a point edit)
    unsigned long long *d = nullptr, h[3] = {0, 0, 0};
    GK_HIP(ctx, hipMalloc((void **)&d, 24));
    hipError_t e = hipMemsetAsync(d, 0, 24, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_count_live, dim3(ggrid(ctx, std::max(g->v.n_nodes, g->v.n_edges))), dim3(BLOCK), 0, ctx->stream, g->v, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, 24, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return hip_fail(ctx, e, "graph_refresh_counts");
    g->live_nodes = h[0]; g->live_edges = h[1]; g->live_len = h[2];
    return GK_OK;
}

int graph_build_index(gk_graph *g) {
    gk_ctx *ctx = g->ctx;
    GraphView &v = g->v;
    if (v.nidx) { GK_HIP(ctx, hipFree(v.nidx)); v.nidx = nullptr; }
    const u64 cap = pow2ceil(std::max<u64>(16, 2 * v.n_nodes + 2));
    GK_HIP(ctx, hipMalloc((void **)&v.nidx, cap * 4));
    GK_HIP(ctx, hipMemsetAsync(v.nidx, 0xff, cap * 4, ctx->stream));
    v.nidx_mask = cap - 1;
    if (v.n_nodes) {
        if (g->W == 1) hipLaunchKernelGGL(k_build_nidx<1>, dim3(ggrid(ctx, v.n_nodes)), dim3(BLOCK), 0, ctx->stream, v);
        else hipLaunchKernelGGL(k_build_nidx<2>, dim3(ggrid(ctx, v.n_nodes)), dim3(BLOCK), 0, ctx->stream, v);
        GK_HIP(ctx, hipGetLastError());
    }
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    g->index_ready = true;
    return GK_OK;
}

// TT: the table the graph phase reads — the map's own hashed table, or the minimizer-bucketed copy built from it (graph_build_entry)
template <int W, class TT> static int graph_build_impl(gk_map *m, gk_graph *g, TT t, bool masks_valid = false) {
    gk_ctx *ctx = m->ctx;
    const int k = m->k;
    const u64 tcap = t.capacity();
    unsigned long long *d_cnt = nullptr;     // [0] terminals [1] cursor [2] edges [3] ecursor [4] pool cursor
    u32 *d_err = nullptr;
    u64 *tslots = nullptr;
    unsigned long long *termbits = nullptr;  // one bit per slot of the table the build reads: a terminal k-mer lives there (k_classify / k_finish_masks -> k_collect_bits)
    int rc = GK_OK;
    hipError_t e = hipMalloc((void **)&d_cnt, 8 * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_err, 16);
    if (e == hipSuccess) e = hipMalloc((void **)&termbits, std::max<u64>(tcap / 64, 1) * 8);
    if (e == hipSuccess) e = hipMemsetAsync(d_cnt, 0, 64, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_err, 0, 16, ctx->stream);
    unsigned long long h_cnt[8] = {0};
    u32 h_err = 0;
    auto done = [&](int code) {
        if (d_cnt) (void)hipFree(d_cnt);
        if (d_err) (void)hipFree(d_err);
        if (tslots) (void)hipFree(tslots);
        if (termbits) (void)hipFree(termbits);
        return code;
    };
    if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_graph_build: alloc"));
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](int i) {
        const auto now = std::chrono::steady_clock::now();
        g->build_ms[i] += std::chrono::duration<float, std::milli>(now - t_prev).count();
        t_prev = now;
    };
    // 1. degree classification of every live key — or, on a table gathered WITH its owners' masks (gk_dist_gather_map in its
    //    classified form), only what follows from the masks: one streaming pass, no neighbour lookups
    bool from_masks = false;
    if constexpr (!is_mb<TT>::value) from_masks = masks_valid;
    if constexpr (!is_mb<TT>::value) {
        if (from_masks) hipLaunchKernelGGL((k_finish_masks<W>), dim3(ggrid(ctx, tcap)), dim3(BLOCK), 0, ctx->stream, t, k, &d_cnt[0], termbits);
    }
    if (!from_masks) hipLaunchKernelGGL((k_classify<W, TT>), dim3(ggrid(ctx, tcap)), dim3(BLOCK), 0, ctx->stream, t, k, &d_cnt[0], termbits);
    g->used_masks = from_masks ? 1 : 0;
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h_cnt, d_cnt, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_graph_build: classify"));
    lap(0);
    const u64 nT = h_cnt[0];
    if (2 * nT >= (u64)NONE) return done(fail(ctx, GK_E_CAPACITY, "more than 2^32 graph nodes"));        // (also: j < 2^31 fits AUX_NODE | j)
    // 2. terminal slots -> nodes (both strands) and edge stubs
    e = hipMalloc((void **)&tslots, std::max<u64>(nT, 1) * 8);
    if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_graph_build: alloc nodes"));
    hipLaunchKernelGGL(k_collect_bits, dim3(ggrid(ctx, tcap / 64 + 1)), dim3(BLOCK), 0, ctx->stream, termbits, tcap / 64, tslots, &d_cnt[1]);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h_cnt, d_cnt, 24, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_graph_build: collect"));
    if (h_cnt[1] != nT) return done(fail(ctx, GK_E_STATE, "terminal count mismatch"));
    (void)hipFree(termbits); termbits = nullptr;                 // (an eighth of a byte per slot: gone before the graph arrays exist)
    const u64 nE = h_cnt[2];
    if (nE >= (u64)NONE) return done(fail(ctx, GK_E_CAPACITY, "more than 2^32 graph edges"));
    if ((rc = graph_alloc_nodes(g, 2 * nT)) != GK_OK) return done(rc);
    if ((rc = graph_alloc_edges(g, nE)) != GK_OK) return done(rc);
    g->v.k = k;
    if (nT) {
        hipLaunchKernelGGL((k_make_nodes<W, TT>), dim3(ggrid(ctx, nT)), dim3(BLOCK), 0, ctx->stream, t, k, tslots, nT, g->v, &d_cnt[3]);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(h_cnt, d_cnt, 32, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_graph_build: make_nodes"));
        // every edge slot the walk will visit must have been written by k_make_nodes
        if (h_cnt[3] != nE) return done(fail(ctx, GK_E_STATE, "edge stub count mismatch: " + std::to_string(h_cnt[3]) + " vs " + std::to_string(nE)));
    }
    lap(1);
    // 3. unitigs: measure, reserve the sequence pool, emit.  One lane per edge walking base by base
    //    (k_walk) when edges are short; pointer jumping when they are long (see k_pj_* above).
    if (nE) {
        const bool use_pj = ctx->hook_unitigs ? ctx->hook_unitigs == 2 : (m->size / std::max<u64>(nE, 1) >= 16);   // (hook: gk_ctx_set_option "graph_unitigs")
        u64 *st = nullptr;                     // pointer jumping: one word per oriented live k-mer
        RankBlk *rb = nullptr;                 // ... and the slot -> rank structure
        u32 *chunk_tot = nullptr;
        u64 *chunk_base = nullptr;
        ulonglong2 *stage = nullptr;           // queue-fed walk: the first WALK_BUF bases of every edge, 32 bytes each
        auto pj_free = [&]() {
            for (void *p : {(void *)st, (void *)rb, (void *)chunk_tot, (void *)chunk_base, (void *)stage}) if (p) (void)hipFree(p);
            st = nullptr; rb = nullptr; chunk_tot = nullptr; chunk_base = nullptr; stage = nullptr;
        };
        if (use_pj) {
            // what is alive here besides the table: 16 B per live key (st) + 0.25 B per slot (rb) — DESIGN.md section 6
            const u64 nstates = 2 * m->size, nblk = (tcap + 63) / 64, nchunks = (nblk + RANK_CHUNK - 1) / RANK_CHUNK;
            if (nstates >= PJ_MAX_STATES) return done(fail(ctx, GK_E_CAPACITY, "more than 2^33 k-mers: beyond the pointer-jumping state's index"));
            e = hipMalloc((void **)&rb, nblk * sizeof(RankBlk));
            if (e == hipSuccess) e = hipMalloc((void **)&chunk_tot, nchunks * 4);
            if (e == hipSuccess) e = hipMalloc((void **)&chunk_base, (nchunks + 1) * 8);
            if (e == hipSuccess) e = hipMalloc((void **)&st, std::max<u64>(nstates, 1) * 8);
            if (e == hipSuccess) e = hipMemsetAsync(st, 0xff, std::max<u64>(nstates, 1) * 8, ctx->stream);         // PJ_UNREG
            if (e != hipSuccess) { pj_free(); return done(hip_fail(ctx, e, "gk_graph_build: pointer-jumping arrays")); }
            hipLaunchKernelGGL((k_rank_masks<W, TT>), dim3((int)std::min<u64>(nchunks, (u64)ctx->cu_count * 8)), dim3(BLOCK), 0, ctx->stream, t, rb, nblk, chunk_tot, nchunks);
            hipLaunchKernelGGL(k_rank_scan, dim3(1), dim3(1024), 0, ctx->stream, chunk_tot, chunk_base, nchunks);
            hipLaunchKernelGGL(k_rank_fill, dim3((int)std::min<u64>(nchunks, (u64)ctx->cu_count * 8)), dim3(RANK_CHUNK), 0, ctx->stream, rb, nblk, chunk_base, nchunks);
            unsigned long long ranked = 0;
            e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(&ranked, chunk_base + nchunks, 8, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) { pj_free(); return done(hip_fail(ctx, e, "gk_graph_build: slot ranks")); }
            if (ranked != m->size) { pj_free(); return done(fail(ctx, GK_E_STATE, "live slots (" + std::to_string(ranked) + ") differ from the map's size (" + std::to_string(m->size) + ")")); }
            hipLaunchKernelGGL((k_pj_init<W, TT>), dim3(ggrid(ctx, tcap)), dim3(BLOCK), 0, ctx->stream, t, k, rb, st);
            e = hipGetLastError();
            // a chain of n k-mers is resolved after ceil(log2 n) rounds; what still moves then is an all-(1,1) cycle (Graph.scala:375)
            int max_rounds = 2;
            while ((1ull << (max_rounds - 2)) < std::max<u64>(nstates, 2)) max_rounds++;
            for (int round = 0; round < max_rounds && nstates && e == hipSuccess; round++) {
                u32 flags[2] = {0, 0};
                e = hipMemsetAsync(d_err + 1, 0, 8, ctx->stream);      // (d_err has 4 words: [0] the build's error, [1] moved, [2] round error)
                if (e != hipSuccess) break;
                hipLaunchKernelGGL(k_pj_round, dim3(ggrid(ctx, nstates)), dim3(BLOCK), 0, ctx->stream, st, nstates, d_err + 1);
                e = hipMemcpyAsync(flags, d_err + 1, 8, hipMemcpyDeviceToHost, ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                if (e == hipSuccess && flags[1]) { pj_free(); return done(fail(ctx, GK_E_STATE, "pointer jumping met an unregistered successor (code " + std::to_string(flags[1]) + ")")); }
                if (!flags[0]) break;
            }
            if (e != hipSuccess) { pj_free(); return done(hip_fail(ctx, e, "gk_graph_build: pj rounds")); }
            hipLaunchKernelGGL((k_pj_edges<W, TT>), dim3(ggrid(ctx, nE)), dim3(BLOCK), 0, ctx->stream, t, k, g->v, rb, st, d_err);
        } else {
            // every oriented interior k-mer lies on exactly one edge: sum of lengths <= edges + 2 x live keys, and every edge
            // rounds up to a byte — the pool can be allocated before the walk, so the walk can write as it goes
            g->pool_cap = ((nE + 2 * m->size) / 4 + nE + 16) / 4 * 4;
            e = hipMalloc((void **)&g->v.pool, g->pool_cap);
            if (e != hipSuccess) { pj_free(); return done(hip_fail(ctx, e, "gk_graph_build: pool")); }
            if (ctx->hook_walk_queue == 0)        // ("graph_walk_queue": 0 = one edge per lane, the round-1 form; A/B)
                hipLaunchKernelGGL((k_walk<W, TT>), dim3(ggrid(ctx, nE)), dim3(BLOCK), 0, ctx->stream, t, k, g->v, 0, tcap + 1, &d_cnt[4], &d_cnt[6], d_err);
            else {
                e = hipMalloc((void **)&stage, std::max<u64>(nE, 1) * 32);
                if (e != hipSuccess) { pj_free(); return done(hip_fail(ctx, e, "gk_graph_build: edge staging")); }
                const int gq = (int)std::min<u64>((nE + 4 * BLOCK - 1) / (4 * BLOCK), (u64)ctx->cu_count * 8);
                hipLaunchKernelGGL((k_walk_q<W, TT>), dim3(std::max(gq, 1)), dim3(BLOCK), 0, ctx->stream, t, k, g->v, tcap + 1, &d_cnt[7], stage, &d_cnt[6], d_err);
                hipLaunchKernelGGL(k_place_edges, dim3(ggrid(ctx, nE / 8 + 1)), dim3(BLOCK), 0, ctx->stream, g->v, stage, &d_cnt[4]);
            }
        }
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        lap(2);
        g->used_pj = use_pj ? 1 : 0;
        if (e == hipSuccess && use_pj) {
            hipLaunchKernelGGL(k_reserve_pool, dim3(ggrid(ctx, nE)), dim3(BLOCK), 0, ctx->stream, g->v, (u64)0, &d_cnt[4]);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(h_cnt, d_cnt, 56, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&h_err, d_err, 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { pj_free(); return done(hip_fail(ctx, e, "gk_graph_build: walk")); }
        if (h_err) { pj_free(); return done(fail(ctx, GK_E_STATE, "unitig construction failed (code " + std::to_string(h_err) +
                                    "): the table changed since classification or is inconsistent")); }
        g->pool_used = h_cnt[4];
        if (use_pj) {
            g->pool_cap = (std::max<u64>(g->pool_used, 1) + 7) / 4 * 4;        // whole 32-bit words (k_pj_emit ORs words)
            e = hipMalloc((void **)&g->v.pool, g->pool_cap);
            if (e != hipSuccess) { pj_free(); return done(hip_fail(ctx, e, "gk_graph_build: pool")); }
        } else if (g->pool_used > g->pool_cap) {
            pj_free();
            return done(fail(ctx, GK_E_STATE, "edge sequences need " + std::to_string(g->pool_used) + " bytes, bound was " + std::to_string(g->pool_cap)));
        }
        lap(3);
        if (use_pj) {
            e = hipMemsetAsync(g->v.pool, 0, g->pool_cap, ctx->stream);
            if (e == hipSuccess) {
                hipLaunchKernelGGL((k_pj_emit<W, TT>), dim3(ggrid(ctx, std::max<u64>(tcap, nE))), dim3(BLOCK), 0, ctx->stream, t, k, g->v, rb, st, d_err);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipMemcpyAsync(&h_err, d_err, 4, hipMemcpyDeviceToHost, ctx->stream);
        } else if (h_cnt[6]) {       // edges longer than the walk's register buffer: second walk, emitting
            hipLaunchKernelGGL((k_walk<W, TT>), dim3(ggrid(ctx, nE)), dim3(BLOCK), 0, ctx->stream, t, k, g->v, 1, tcap + 1, &d_cnt[4], &d_cnt[6], d_err);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        pj_free();
        if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_graph_build: emit"));
        if (h_err) return done(fail(ctx, GK_E_STATE, "unitig emission failed (code " + std::to_string(h_err) + ")"));
        lap(4);
    } else {
        e = hipMalloc((void **)&g->v.pool, 1);
        if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_graph_build: pool"));
        g->pool_cap = 1;
    }
    // (the k-mer -> node index serves point queries and by-k-mer edits only — GraphBuilder's own flow, components, the export and
    //  the paired-end stage go by ids: it is built by the first call that needs it, graph_ensure_index; 1.5 ms of C3's buildGraph)
    g->index_ready = false;
    rc = graph_refresh_counts(g);
    g->walked_bases = g->live_len;
    lap(5);
    return done(rc);
}

template <int W> __global__ __launch_bounds__(BLOCK) void k_mb_clear(Slot<W> *slots, u64 n) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        if constexpr (W == 1) slots[i] = Slot<1>{KEY_EMPTY, 0u, 0u};
        else slots[i] = Slot<2>{KEY_EMPTY, KEY_EMPTY, 0u, 0u};
    }
}

// ---- the minimizer-bucketed copy (MbTable, gk_device.h) ------------------------------------------------------------------------
template <int W> __global__ __launch_bounds__(BLOCK) void k_mb_count(Table<W> src, int k, u32 nb, u32 *cnt, u32 *bucket_of) {
    const u64 ncap = src.capacity();
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < ncap; i += (u64)gridDim.x * BLOCK) {
        if (!slot_live(&src.slots[i])) continue;
        const u32 b = (u32)owner_of(slot_key(src, i), k, (int)nb);
        bucket_of[i] = b;
        atomicAdd(&cnt[b], 1u);
    }
}
// keys in a bucket -> slots of its region: the power of two that keeps the load at or under 0.5, 8 at least
__global__ __launch_bounds__(BLOCK) void k_mb_sizes(u32 *cnt, u32 nb) {
    for (u64 b = (u64)blockIdx.x * BLOCK + threadIdx.x; b < nb; b += (u64)gridDim.x * BLOCK) {
        u32 want = max(8u, 2u * cnt[b]), p = 8u;
        while (p < want) p <<= 1;
        cnt[b] = p;
    }
}
template <int W> __global__ __launch_bounds__(BLOCK) void k_mb_fill(Table<W> src, const u32 *bucket_of, MbTable<W> dst, u32 *err) {
    const u64 ncap = src.capacity();
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < ncap; i += (u64)gridDim.x * BLOCK) {
        if (!slot_live(&src.slots[i])) continue;
        const Kmer<W> key = slot_key(src, i);
        const Stored<W> kk = to_stored(key);
        const ProbeAt<W> p = probe_at_bucket(dst, key, bucket_of[i]);
        Slot<W> *reg = const_cast<Slot<W> *>(p.reg);
        u32 at = p.pos;
        bool placed = false;
        for (u32 n = 0; n <= p.mask && !placed; n++, at = (at + 1) & p.mask) {
            // (every key of the source is unique: a slot is taken by claiming its first word; with 16-byte keys a slot whose first
            //  word equals ours belongs to ANOTHER key that shares it — see seg_claim_unique)
            const u64 c0 = cas64(&reg[at].w0, KEY_EMPTY, kk.w0);
            if constexpr (W == 1) { if (c0 == KEY_EMPTY) placed = true; }
            else { if ((c0 == KEY_EMPTY || c0 == kk.w0) && cas64(&reg[at].w1, KEY_EMPTY, kk.w1) == KEY_EMPTY) placed = true; }
            if (placed) reg[at].extra = src.slots[i].extra;
        }
        if (!placed) *err = 1u;
    }
}

template <int W> static int graph_build_entry(gk_map *m, gk_graph *g, bool masks_valid) {
    gk_ctx *ctx = m->ctx;
    // (m->dirty: keys were inserted verbatim and at least one was not its k-mer's hash-rule orientation — the reference's
    //  `contains` probes both strands unconditionally, Graph.scala:270; so does every lookup below then)
    Table<W> t{reinterpret_cast<Slot<W> *>(m->slots), m->nb2, m->lnb1, m->k == 64 ? 1u : 0u, m->dirty ? 1u : 0u};
    // "graph_mbt" = 1: classify and walk on a minimizer-bucketed COPY of the table (A/B, profiles/r03); never for k = 64 (tagged slots)
    const bool use_mb = ctx->hook_graph_mbt > 0 && m->k != 64 && m->size >= 4096;
    if (!use_mb) return graph_build_impl<W, Table<W>>(m, g, t, masks_valid);
    const auto t0 = std::chrono::steady_clock::now();
    const u32 nb = (u32)std::min<u64>(std::max<u64>(m->size / (u64)std::max(ctx->hook_graph_mbt_keys, 16), 1), 1u << 30);
    u32 *d_cnt = nullptr, *d_bucket = nullptr, *d_err = nullptr;
    unsigned long long *d_off = nullptr, total = 0;
    u64 *d_sums = nullptr;
    Slot<W> *slots = nullptr;
    auto done = [&](int code) {
        for (void *p : {(void *)d_cnt, (void *)d_bucket, (void *)d_err, (void *)d_off, (void *)d_sums, (void *)slots}) if (p) (void)hipFree(p);
        return code;
    };
    hipError_t e = hipMalloc((void **)&d_cnt, (u64)nb * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_bucket, m->capacity * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_off, ((u64)nb + 1) * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_sums, ((u64)nb / SCAN_CHUNK + 2) * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_err, 4);
    if (e == hipSuccess) e = hipMemsetAsync(d_cnt, 0, (u64)nb * 4, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_err, 0, 4, ctx->stream);
    if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_graph_build: bucketed table"));
    hipLaunchKernelGGL(k_mb_count<W>, dim3(ggrid(ctx, m->capacity)), dim3(BLOCK), 0, ctx->stream, t, m->k, nb, d_cnt, d_bucket);
    hipLaunchKernelGGL(k_mb_sizes, dim3(ggrid(ctx, nb)), dim3(BLOCK), 0, ctx->stream, d_cnt, nb);
    e = scan_counts(ctx, d_cnt, nb, d_off, d_sums);
    if (e == hipSuccess) e = hipMemcpyAsync(&total, d_off + nb, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_graph_build: bucket regions"));
    e = hipMalloc((void **)&slots, total * sizeof(Slot<W>));
    if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_graph_build: bucketed table"));
    hipLaunchKernelGGL(k_mb_clear<W>, dim3(ggrid(ctx, total / 4 + 1)), dim3(BLOCK), 0, ctx->stream, slots, (u64)total);
    MbTable<W> mt{slots, reinterpret_cast<const u64 *>(d_off), nb, m->dirty ? 1u : 0u, (u64)total};
    hipLaunchKernelGGL(k_mb_fill<W>, dim3(ggrid(ctx, m->capacity)), dim3(BLOCK), 0, ctx->stream, t, d_bucket, mt, d_err);
    u32 h_err = 0;
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&h_err, d_err, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_graph_build: filling the bucketed table"));
    if (h_err) return done(fail(ctx, GK_E_STATE, "gk_graph_build: a bucket region filled up (internal sizing error)"));
    (void)hipFree(d_bucket); d_bucket = nullptr;
    (void)hipFree(d_cnt); d_cnt = nullptr;
    g->mbt_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    g->mbt_slots = total;
    return done(graph_build_impl<W, MbTable<W>>(m, g, mt));
}

int check_graph(const gk_graph *g) {
    if (!g || !g->ctx) return fail(nullptr, GK_E_INVALID, "null graph handle");
    hipError_t e = hipSetDevice(g->ctx->device);
    if (e != hipSuccess) return hip_fail(g->ctx, e, "hipSetDevice");
    return GK_OK;
}

extern "C" {

int gk_graph_build(gk_map *m, gk_graph **out) {
    if (!m || !m->ctx) return fail(nullptr, GK_E_INVALID, "null map handle");
    if (!out) return fail(m->ctx, GK_E_INVALID, "gk_graph_build: out is NULL");
    *out = nullptr;
    GK_HIP(m->ctx, hipSetDevice(m->ctx->device));
    if (int rc = map_materialize(m)) return rc;
    // The graph phase annotates every k-mer in its slot: a COUNT table of 8-byte keys (12-byte slots, no annotation word) is
    // rebuilt into the graph layout first — what deleteAll(v < rounds) does anyway on the reference's path (GraphBuilder.scala:30-36).
    if (int rc = map_to_graph_layout(m)) return rc;
    gk_graph *g = new gk_graph();
    g->ctx = m->ctx;
    g->k = m->k;
    g->W = m->W;
    // (the build re-uses the annotation word — a terminal slot's becomes its node — so owner-computed masks serve ONE build)
    const bool masks_valid = m->masks_valid;
    m->masks_valid = false;
    int rc = m->W == 1 ? graph_build_entry<1>(m, g, masks_valid) : graph_build_entry<2>(m, g, masks_valid);
    if (rc != GK_OK) {
        graph_free_arrays(g);
        delete g;
        return rc;
    }
    *out = g;
    return GK_OK;
}

void gk_graph_destroy(gk_graph *g) {
    if (!g) return;
    (void)hipSetDevice(g->ctx->device);
    (void)hipStreamSynchronize(g->ctx->stream);
    graph_free_arrays(g);
    delete g;
}

int gk_graph_counts(gk_graph *g, uint64_t *nodes, uint64_t *edges, uint64_t *total_edge_len) {
    if (int rc = check_graph(g)) return rc;
    if (nodes) *nodes = g->live_nodes;
    if (edges) *edges = g->live_edges;
    if (total_edge_len) *total_edge_len = g->live_len;
    return GK_OK;
}

int gk_graph_simplify(gk_graph *g) {
    if (int rc = check_graph(g)) return rc;
    gk_ctx *ctx = g->ctx;
    GraphView &v = g->v;
    if (v.n_nodes == 0) return GK_OK;
    u32 *in_single = nullptr, *merged_key = nullptr;
    uint8_t *cls = nullptr;
    unsigned long long *d_cnt = nullptr, h_cnt[3] = {0, 0, 0};
    LongPiece *long_pieces = nullptr;
    auto done = [&](int code) {
        if (in_single) (void)hipFree(in_single);
        if (merged_key) (void)hipFree(merged_key);
        if (cls) (void)hipFree(cls);
        if (d_cnt) (void)hipFree(d_cnt);
        if (long_pieces) (void)hipFree(long_pieces);
        return code;
    };
    hipError_t e = hipMalloc((void **)&in_single, v.n_nodes * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&merged_key, v.n_nodes * 16);
    if (e == hipSuccess) e = hipMalloc((void **)&cls, v.n_nodes);
    if (e == hipSuccess) e = hipMalloc((void **)&d_cnt, 24);
    if (e == hipSuccess) e = hipMemsetAsync(in_single, 0xff, v.n_nodes * 4, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(merged_key, 0xff, v.n_nodes * 16, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_cnt, 0, 24, ctx->stream);
    if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_graph_simplify: alloc"));
    const int gn = ggrid(ctx, v.n_nodes), ge = ggrid(ctx, std::max<u64>(v.n_edges, 1));
    hipLaunchKernelGGL(k_in_single, dim3(ge), dim3(BLOCK), 0, ctx->stream, v, in_single);
    hipLaunchKernelGGL(k_node_class, dim3(gn), dim3(BLOCK), 0, ctx->stream, v, in_single, cls);
    const u64 old_edges = v.n_edges, old_pool = g->pool_used;
    hipLaunchKernelGGL(k_chain, dim3(ge), dim3(BLOCK), 0, ctx->stream, v, cls, 0, old_edges, old_pool, d_cnt, merged_key, (LongPiece *)nullptr);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h_cnt, d_cnt, 24, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_graph_simplify: count"));
    if (h_cnt[0]) {
        if (old_edges + h_cnt[0] >= (u64)NONE) return done(fail(ctx, GK_E_CAPACITY, "more than 2^32 graph edges"));
        if (int rc = graph_grow_edges(g, old_edges + h_cnt[0])) return done(rc);
        if (old_pool + h_cnt[1] + 8 > g->pool_cap) {             // (+8: k_copy_long ORs whole 32-bit words, the last one may reach past the last byte)
            e = dev_grow(ctx, &v.pool, old_pool, old_pool + h_cnt[1] + 8, ctx->stream);
            if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_graph_simplify: pool"));
            g->pool_cap = old_pool + h_cnt[1] + 8;
        }
        e = hipMemsetAsync(d_cnt, 0, 24, ctx->stream);
        if (e == hipSuccess && h_cnt[2]) {                       // long pieces are ORed into their place: it starts as zeroes
            e = hipMalloc((void **)&long_pieces, h_cnt[2] * sizeof(LongPiece));
            if (e == hipSuccess) e = hipMemsetAsync(v.pool + old_pool, 0, h_cnt[1], ctx->stream);
        }
        if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_graph_simplify"));
        hipLaunchKernelGGL(k_chain, dim3(ge), dim3(BLOCK), 0, ctx->stream, v, cls, 1, old_edges, old_pool, d_cnt, merged_key, long_pieces);
        if (h_cnt[2]) hipLaunchKernelGGL(k_copy_long, dim3((unsigned)h_cnt[2]), dim3(BLOCK), 0, ctx->stream, v, long_pieces);
        v.n_edges = old_edges + h_cnt[0];
        g->pool_used = old_pool + h_cnt[1];
    }
    hipLaunchKernelGGL(k_simplify_finish, dim3(gn), dim3(BLOCK), 0, ctx->stream, v, cls, merged_key);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_graph_simplify: finish"));
    return done(graph_refresh_counts(g));
}

int gk_graph_remove_bubbles(gk_graph *g) {
    if (int rc = check_graph(g)) return rc;
    gk_ctx *ctx = g->ctx;
    if (g->v.n_nodes == 0) return GK_OK;
    hipLaunchKernelGGL(k_bubbles, dim3(ggrid(ctx, g->v.n_nodes)), dim3(BLOCK), 0, ctx->stream, g->v);
    GK_HIP(ctx, hipGetLastError());
    return graph_refresh_counts(g);
}

int gk_graph_remove_edges(gk_graph *g, const uint64_t *start_lo, const uint64_t *start_hi, const uint8_t *base, uint64_t n,
                          uint64_t *removed) {
    if (int rc = check_graph(g)) return rc;
    gk_ctx *ctx = g->ctx;
    if (removed) *removed = 0;
    if (n == 0) return GK_OK;
    if (int rc = graph_ensure_index(g)) return rc;             // (edges are named by their start k-mer here)
    if (!start_lo || !base || (g->W == 2 && !start_hi)) return fail(ctx, GK_E_INVALID, "null argument");
    u64 *d_lo = nullptr, *d_hi = nullptr;
    uint8_t *d_b = nullptr;
    unsigned long long *d_rm = nullptr, h_rm = 0;
    hipError_t e = hipMalloc((void **)&d_lo, n * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_hi, n * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_b, n);
    if (e == hipSuccess) e = hipMalloc((void **)&d_rm, 8);
    if (e == hipSuccess) e = hipMemcpyAsync(d_lo, start_lo, n * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = start_hi ? hipMemcpyAsync(d_hi, start_hi, n * 8, hipMemcpyHostToDevice, ctx->stream)
                                      : hipMemsetAsync(d_hi, 0, n * 8, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_b, base, n, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_rm, 0, 8, ctx->stream);
    if (e == hipSuccess) {
        if (g->W == 1) hipLaunchKernelGGL(k_remove_edges<1>, dim3(ggrid(ctx, n)), dim3(BLOCK), 0, ctx->stream, g->v, d_lo, d_hi, d_b, n, d_rm);
        else hipLaunchKernelGGL(k_remove_edges<2>, dim3(ggrid(ctx, n)), dim3(BLOCK), 0, ctx->stream, g->v, d_lo, d_hi, d_b, n, d_rm);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&h_rm, d_rm, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_lo); (void)hipFree(d_hi); (void)hipFree(d_b); (void)hipFree(d_rm);
    if (e != hipSuccess) return hip_fail(ctx, e, "gk_graph_remove_edges");
    if (removed) *removed = h_rm;
    return graph_refresh_counts(g);
}

}  // extern "C"

// ---- host side of the classify over a PartitionedDNAMap (kernels k_dc_* above; the exchanges are gk_dist.hip's) ----------------
namespace gk {
template <int W> static Table<W> graph_table_of(gk_map *m) {
    return Table<W>{reinterpret_cast<Slot<W> *>(m->slots), m->nb2, m->lnb1, m->k == 64 ? 1u : 0u, m->dirty ? 1u : 0u};
}
// slots [s0, s1) of a GRAPH-layout table: local neighbour lookups into the annotation word, remote neighbours counted per owner in d_cnt[P] (zeroed here)
int dclass_count(gk_map *m, int rank, int P, u64 s0, u64 s1, unsigned long long *d_cnt) {
    gk_ctx *ctx = m->ctx;
    GK_HIP(ctx, hipMemsetAsync(d_cnt, 0, 64 * 8, ctx->stream));
    if (s1 > m->capacity) s1 = m->capacity;
    if (s0 >= s1) return GK_OK;
    const int grid = ggrid(ctx, s1 - s0);
    if (m->W == 1) hipLaunchKernelGGL((k_dc_scan<1, false>), dim3(grid), dim3(BLOCK), 0, ctx->stream, graph_table_of<1>(m), m->k, rank, P, s0, s1, d_cnt, nullptr, nullptr, nullptr);
    else hipLaunchKernelGGL((k_dc_scan<2, false>), dim3(grid), dim3(BLOCK), 0, ctx->stream, graph_table_of<2>(m), m->k, rank, P, s0, s1, d_cnt, nullptr, nullptr, nullptr);
    GK_HIP(ctx, hipGetLastError());
    return GK_OK;
}
// the same slots again: the remote neighbours' canonical keys into their owners' regions (d_off[p] = first query for owner p; d_cur[P] zeroed here)
int dclass_fill(gk_map *m, int rank, int P, u64 s0, u64 s1, const unsigned long long *d_off, unsigned long long *d_cur, u64 *d_qkeys, u64 *d_qref) {
    gk_ctx *ctx = m->ctx;
    GK_HIP(ctx, hipMemsetAsync(d_cur, 0, 64 * 8, ctx->stream));
    if (s1 > m->capacity) s1 = m->capacity;
    if (s0 >= s1) return GK_OK;
    const int grid = ggrid(ctx, s1 - s0);
    if (m->W == 1) hipLaunchKernelGGL((k_dc_scan<1, true>), dim3(grid), dim3(BLOCK), 0, ctx->stream, graph_table_of<1>(m), m->k, rank, P, s0, s1, d_cur, d_off, d_qkeys, d_qref);
    else hipLaunchKernelGGL((k_dc_scan<2, true>), dim3(grid), dim3(BLOCK), 0, ctx->stream, graph_table_of<2>(m), m->k, rank, P, s0, s1, d_cur, d_off, d_qkeys, d_qref);
    GK_HIP(ctx, hipGetLastError());
    return GK_OK;
}
int dclass_answer(gk_map *m, const u64 *d_keys, u64 n, uint8_t *d_ans) {
    gk_ctx *ctx = m->ctx;
    if (!n) return GK_OK;
    if (m->W == 1) hipLaunchKernelGGL((k_dc_answer<1>), dim3(ggrid(ctx, n)), dim3(BLOCK), 0, ctx->stream, graph_table_of<1>(m), m->k, d_keys, n, d_ans);
    else hipLaunchKernelGGL((k_dc_answer<2>), dim3(ggrid(ctx, n)), dim3(BLOCK), 0, ctx->stream, graph_table_of<2>(m), m->k, d_keys, n, d_ans);
    GK_HIP(ctx, hipGetLastError());
    return GK_OK;
}
int dclass_apply(gk_map *m, const u64 *d_qref, const uint8_t *d_ans, u64 n) {
    gk_ctx *ctx = m->ctx;
    if (!n) return GK_OK;
    if (m->W == 1) hipLaunchKernelGGL((k_dc_apply<1>), dim3(ggrid(ctx, n)), dim3(BLOCK), 0, ctx->stream, graph_table_of<1>(m), d_qref, d_ans, n);
    else hipLaunchKernelGGL((k_dc_apply<2>), dim3(ggrid(ctx, n)), dim3(BLOCK), 0, ctx->stream, graph_table_of<2>(m), d_qref, d_ans, n);
    GK_HIP(ctx, hipGetLastError());
    return GK_OK;
}
}  // namespace gk

// Graph.components (Graph.scala:54-72): label every live node with its component's root (min-label hooking + pointer
// jumping) and count nodes per root.  *parent / *size are hipMalloc'ed here ([n_nodes] each); the caller frees them.
static int graph_components(gk_graph *g, u32 **parent_out, u32 **size_out, u64 *ncomp) {
    gk_ctx *ctx = g->ctx;
    GraphView &v = g->v;
    u32 *parent = nullptr, *size = nullptr;
    unsigned long long *d_ncomp = nullptr;
    auto bail = [&](int code) {
        for (void *p : {(void *)parent, (void *)size, (void *)d_ncomp}) if (p) (void)hipFree(p);
        return code;
    };
    hipError_t e = hipMalloc((void **)&parent, std::max<u64>(v.n_nodes, 1) * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&size, std::max<u64>(v.n_nodes, 1) * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_ncomp, 8);
    if (e != hipSuccess) return bail(hip_fail(ctx, e, "graph components: alloc"));
    const int gn = ggrid(ctx, std::max<u64>(v.n_nodes, 1)), ge = ggrid(ctx, std::max<u64>(v.n_edges, 1));
    hipLaunchKernelGGL(k_cc_init, dim3(gn), dim3(BLOCK), 0, ctx->stream, v, parent);
    if (ctx->hook_cc_find == 1) hipLaunchKernelGGL(k_cc_link<1>, dim3(ge), dim3(BLOCK), 0, ctx->stream, v, parent);
    else if (ctx->hook_cc_find == 3) hipLaunchKernelGGL(k_cc_link<3>, dim3(ge), dim3(BLOCK), 0, ctx->stream, v, parent);
    else if (ctx->hook_cc_find == 2) hipLaunchKernelGGL(k_cc_link<2>, dim3(ge), dim3(BLOCK), 0, ctx->stream, v, parent);
    else hipLaunchKernelGGL(k_cc_link<0>, dim3(ge), dim3(BLOCK), 0, ctx->stream, v, parent);
    e = hipGetLastError();
    if (e != hipSuccess) return bail(hip_fail(ctx, e, "graph components: hooking"));
    unsigned long long h = 0;
    e = hipMemsetAsync(size, 0, std::max<u64>(v.n_nodes, 1) * 4, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_ncomp, 0, 8, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_cc_sizes, dim3(gn), dim3(BLOCK), 0, ctx->stream, v, parent, size, d_ncomp);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d_ncomp, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return bail(hip_fail(ctx, e, "graph components: sizes"));
    (void)hipFree(d_ncomp);
    *parent_out = parent; *size_out = size; *ncomp = h;
    return GK_OK;
}

extern "C" {

int gk_graph_retain_largest(gk_graph *g, uint64_t *kept_nodes, uint64_t *components) {
    if (int rc = check_graph(g)) return rc;
    gk_ctx *ctx = g->ctx;
    GraphView &v = g->v;
    if (kept_nodes) *kept_nodes = 0;
    if (components) *components = 0;
    if (g->live_nodes == 0) return GK_OK;
    u32 *parent = nullptr, *size = nullptr, *d_u32 = nullptr;       // d_u32: [0] unused [1] best [2] winner
    unsigned long long *d_u64 = nullptr;                            // [0] unused [1] min hi [2] min lo
    u64 ncomp = 0;
    if (int rc = graph_components(g, &parent, &size, &ncomp)) return rc;
    auto done = [&](int code) {
        if (parent) (void)hipFree(parent);
        if (size) (void)hipFree(size);
        if (d_u32) (void)hipFree(d_u32);
        if (d_u64) (void)hipFree(d_u64);
        return code;
    };
    hipError_t e = hipMalloc((void **)&d_u32, 16);
    if (e == hipSuccess) e = hipMalloc((void **)&d_u64, 24);
    if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_graph_retain_largest: alloc"));
    const int gn = ggrid(ctx, v.n_nodes);
    unsigned long long h64[3] = {0, ~0ull, ~0ull};
    u32 h32[3] = {0, 0, NONE};
    e = hipMemcpyAsync(d_u64, h64, 24, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_u32, h32, 12, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_graph_retain_largest"));
    // max size -> smallest k-mer among the components of that size -> its root -> retain: four dependent steps, no host in between
    hipLaunchKernelGGL(k_cc_max, dim3(gn), dim3(BLOCK), 0, ctx->stream, v, size, &d_u32[1]);
    for (int stage = 0; stage < 3; stage++)
        hipLaunchKernelGGL(k_cc_pick, dim3(gn), dim3(BLOCK), 0, ctx->stream, v, parent, size, &d_u32[1], stage, &d_u64[1], &d_u32[2]);
    hipLaunchKernelGGL(k_retain, dim3(ggrid(ctx, std::max(v.n_nodes, v.n_edges))), dim3(BLOCK), 0, ctx->stream, v, parent, &d_u32[2]);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h32, d_u32, 12, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_graph_retain_largest: retain"));
    if (h32[2] == NONE) return done(fail(ctx, GK_E_STATE, "no component selected"));
    if (components) *components = ncomp;
    int rc = graph_refresh_counts(g);
    if (rc == GK_OK && kept_nodes) *kept_nodes = g->live_nodes;
    return done(rc);
}

int gk_graph_component_stats(gk_graph *g, uint32_t *nodes_per_component, uint64_t *edge_len_per_component, uint64_t cap, uint64_t *n) {
    if (int rc = check_graph(g)) return rc;
    gk_ctx *ctx = g->ctx;
    GraphView &v = g->v;
    if (n) *n = 0;
    if (g->live_nodes == 0) return GK_OK;
    u32 *parent = nullptr, *size = nullptr, *d_nodes = nullptr;
    unsigned long long *len = nullptr, *d_len = nullptr, *d_cur = nullptr;
    u64 ncomp = 0;
    if (int rc = graph_components(g, &parent, &size, &ncomp)) return rc;
    auto done = [&](int code) {
        for (void *p : {(void *)parent, (void *)size, (void *)d_nodes, (void *)len, (void *)d_len, (void *)d_cur}) if (p) (void)hipFree(p);
        return code;
    };
    if (n) *n = ncomp;
    if (ncomp > cap) return done(fail(ctx, GK_E_CAPACITY, "component buffer too small: need " + std::to_string(ncomp)));
    if (!nodes_per_component || !edge_len_per_component) return done(fail(ctx, GK_E_INVALID, "null component buffer"));
    hipError_t e = hipMalloc((void **)&len, v.n_nodes * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_nodes, ncomp * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_len, ncomp * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_cur, 8);
    if (e == hipSuccess) e = hipMemsetAsync(len, 0, v.n_nodes * 8, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_cur, 0, 8, ctx->stream);
    if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_graph_component_stats: alloc"));
    hipLaunchKernelGGL(k_cc_edge_len, dim3(ggrid(ctx, std::max<u64>(v.n_edges, 1))), dim3(BLOCK), 0, ctx->stream, v, parent, len);
    hipLaunchKernelGGL(k_cc_collect, dim3(ggrid(ctx, v.n_nodes)), dim3(BLOCK), 0, ctx->stream, v, parent, size, len, d_nodes, d_len, d_cur);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(nodes_per_component, d_nodes, ncomp * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(edge_len_per_component, d_len, ncomp * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_graph_component_stats"));
    return done(GK_OK);
}

int gk_graph_checksum(gk_graph *g, uint64_t *nodes_checksum, uint64_t *edges_checksum) {
    if (int rc = check_graph(g)) return rc;
    gk_ctx *ctx = g->ctx;
    unsigned long long *d = nullptr, h[2] = {0, 0};
    GK_HIP(ctx, hipMalloc((void **)&d, 16));
    hipError_t e = hipMemsetAsync(d, 0, 16, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_graph_checksum, dim3(ggrid(ctx, std::max<u64>(std::max(g->v.n_nodes, g->v.n_edges), 1))), dim3(BLOCK), 0, ctx->stream, g->v, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return hip_fail(ctx, e, "gk_graph_checksum");
    if (nodes_checksum) *nodes_checksum = h[0];
    if (edges_checksum) *edges_checksum = h[1];
    return GK_OK;
}

int gk_graph_build_stats(gk_graph *g, float *phase_ms6, uint64_t *walked_bases, int *pointer_jumping) {
    if (int rc = check_graph(g)) return rc;
    if (phase_ms6) for (int i = 0; i < 6; i++) phase_ms6[i] = g->build_ms[i];
    if (walked_bases) *walked_bases = g->walked_bases;
    if (pointer_jumping) *pointer_jumping = g->used_pj;
    return GK_OK;
}

int gk_graph_classified_by_owners(gk_graph *g, int *flag) {
    if (int rc = check_graph(g)) return rc;
    if (flag) *flag = g->used_masks;
    return GK_OK;
}

int gk_graph_bucketed_table_stats(gk_graph *g, float *build_ms, uint64_t *slots) {
    if (int rc = check_graph(g)) return rc;
    if (build_ms) *build_ms = g->mbt_ms;
    if (slots) *slots = g->mbt_slots;
    return GK_OK;
}

int gk_graph_export_nodes(gk_graph *g, uint64_t *lo, uint64_t *hi, uint64_t cap, uint64_t *n) {
    if (int rc = check_graph(g)) return rc;
    gk_ctx *ctx = g->ctx;
    if (n) *n = g->live_nodes;
    if (g->live_nodes > cap) return fail(ctx, GK_E_CAPACITY, "node export buffer too small: need " + std::to_string(g->live_nodes));
    if (g->live_nodes == 0) return GK_OK;
    if (!lo) return fail(ctx, GK_E_INVALID, "null export buffer");
    const u64 cnt = g->live_nodes;
    u64 *d_lo = nullptr, *d_hi = nullptr;
    unsigned long long *d_cur = nullptr;
    hipError_t e = hipMalloc((void **)&d_lo, cnt * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_hi, cnt * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_cur, 8);
    if (e == hipSuccess) e = hipMemsetAsync(d_cur, 0, 8, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_export_nodes, dim3(ggrid(ctx, g->v.n_nodes)), dim3(BLOCK), 0, ctx->stream, g->v, d_lo, d_hi, d_cur);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(lo, d_lo, cnt * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && hi) e = hipMemcpyAsync(hi, d_hi, cnt * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_lo); (void)hipFree(d_hi); (void)hipFree(d_cur);
    if (e != hipSuccess) return hip_fail(ctx, e, "gk_graph_export_nodes");
    return GK_OK;
}

int gk_graph_export_edges(gk_graph *g, uint64_t *start_lo, uint64_t *start_hi, uint64_t *end_lo, uint64_t *end_hi,
                          int64_t *len, int64_t *seq_off, uint64_t cap, uint64_t *n,
                          uint8_t *seq2bit, uint64_t seq_cap, uint64_t *seq_bytes) {
    if (int rc = check_graph(g)) return rc;
    gk_ctx *ctx = g->ctx;
    if (n) *n = g->live_edges;
    // upper bound on packed bytes: every live edge rounds up to a byte
    const u64 max_bytes = (g->live_len + 3 * g->live_edges) / 4 + 1;
    if (seq_bytes) *seq_bytes = max_bytes;
    if (g->live_edges > cap) return fail(ctx, GK_E_CAPACITY, "edge export buffer too small: need " + std::to_string(g->live_edges));
    if (g->live_edges == 0) { if (seq_bytes) *seq_bytes = 0; return GK_OK; }
    if (seq_cap < max_bytes) return fail(ctx, GK_E_CAPACITY, "sequence buffer too small: need " + std::to_string(max_bytes));
    if (!start_lo || !end_lo || !len || !seq_off || !seq2bit) return fail(ctx, GK_E_INVALID, "null export buffer");
    const u64 cnt = g->live_edges;
    u64 *d_k[4] = {nullptr, nullptr, nullptr, nullptr};
    i64 *d_len = nullptr, *d_off = nullptr;
    uint8_t *d_seq = nullptr;
    unsigned long long *d_cur = nullptr, h_cur[2] = {0, 0};
    hipError_t e = hipSuccess;
    for (int i = 0; i < 4 && e == hipSuccess; i++) e = hipMalloc((void **)&d_k[i], cnt * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_len, cnt * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_off, cnt * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_seq, max_bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&d_cur, 16);
    if (e == hipSuccess) e = hipMemsetAsync(d_cur, 0, 16, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_export_edges, dim3(ggrid(ctx, g->v.n_edges)), dim3(BLOCK), 0, ctx->stream, g->v, d_k[0], d_k[1], d_k[2],
                           d_k[3], d_len, d_off, d_seq, d_cur);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h_cur, d_cur, 16, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(start_lo, d_k[0], cnt * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && start_hi) e = hipMemcpyAsync(start_hi, d_k[1], cnt * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(end_lo, d_k[2], cnt * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && end_hi) e = hipMemcpyAsync(end_hi, d_k[3], cnt * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(len, d_len, cnt * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(seq_off, d_off, cnt * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess && h_cur[1]) e = hipMemcpy(seq2bit, d_seq, h_cur[1], hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; i++) (void)hipFree(d_k[i]);
    (void)hipFree(d_len); (void)hipFree(d_off); (void)hipFree(d_seq); (void)hipFree(d_cur);
    if (e != hipSuccess) return hip_fail(ctx, e, "gk_graph_export_edges");
    if (h_cur[0] != cnt) return fail(ctx, GK_E_STATE, "edge export count mismatch");
    if (seq_bytes) *seq_bytes = h_cur[1];
    return GK_OK;
}

int gk_graph_out_order(gk_graph *g, uint64_t lo, uint64_t hi, int *bases4, int *count) {
    if (int rc = check_graph(g)) return rc;
    gk_ctx *ctx = g->ctx;
    if (!bases4 || !count) return fail(ctx, GK_E_INVALID, "null argument");
    if (int rc = graph_ensure_index(g)) return rc;
    int *d = nullptr, h[5] = {-1, 0, 0, 0, 0};
    GK_HIP(ctx, hipMalloc((void **)&d, 20));
    if (g->W == 1) hipLaunchKernelGGL(k_out_order<1>, dim3(1), dim3(1), 0, ctx->stream, g->v, lo, hi, d);
    else hipLaunchKernelGGL(k_out_order<2>, dim3(1), dim3(1), 0, ctx->stream, g->v, lo, hi, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, 20, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return hip_fail(ctx, e, "gk_graph_out_order");
    *count = h[0];
    for (int i = 0; i < 4; i++) bases4[i] = i < h[0] ? h[1 + i] : 0;
    return GK_OK;
}

}  // extern "C"
int graph_grow_nodes(gk_graph *g, u64 new_cap) {
    gk_ctx *ctx = g->ctx;
    GraphView &v = g->v;
    if (new_cap <= g->node_cap) return GK_OK;
    const u64 old = v.n_nodes;
    const NodeCarve k = node_carve(new_cap);
    void *blob = nullptr;
    GK_HIP(ctx, hipMalloc(&blob, k.total));
    GraphView nv = v;
    node_view(nv, (char *)blob, k);
    hipError_t e = hipMemsetAsync(nv.node_alive, 0, new_cap, ctx->stream);
    if (e == hipSuccess && old) {
        e = hipMemcpyAsync(nv.node_lo, v.node_lo, old * 8, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(nv.node_hi, v.node_hi, old * 8, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(nv.node_alive, v.node_alive, old, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(nv.out_edge, v.out_edge, old * 16, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(nv.out_order, v.out_order, old * 4, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(nv.in_deg, v.in_deg, old * 4, hipMemcpyDeviceToDevice, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipFree(blob); return hip_fail(ctx, e, "graph: growing the node arrays"); }
    (void)hipFree(g->node_blob);
    g->node_blob = blob;
    v = nv;
    g->node_cap = new_cap;
    return GK_OK;
}

extern "C" {

// Graph.getGraphMap (Graph.scala:90-119): putNew of every node k-mer and of every interior k-mer of every edge into `vm`
int gk_graph_position_map(gk_graph *g, gk_vmap *vm, uint64_t *entries) {
    if (int rc = check_graph(g)) return rc;
    gk_ctx *ctx = g->ctx;
    if (entries) *entries = 0;
    if (!vm || vmap_ctx(vm) != ctx) return fail(ctx, GK_E_INVALID, "gk_graph_position_map: the value map must live on the graph's context");
    if (vmap_k(vm) != g->k) return fail(ctx, GK_E_KLEN, "gk_graph_position_map: the map's k differs from the graph's");     // key.length == k
    GraphView &v = g->v;
    unsigned long long *first = nullptr, *d_cur = nullptr, h_cur[2] = {0, 0};
    u64 *lo = nullptr, *hi = nullptr, *val = nullptr;
    auto done = [&](int code) {
        for (void *p : {(void *)first, (void *)d_cur, (void *)lo, (void *)hi, (void *)val}) if (p) (void)hipFree(p);
        return code;
    };
    hipError_t e = hipMalloc((void **)&first, std::max<u64>(v.n_edges, 1) * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_cur, 16);
    if (e == hipSuccess) e = hipMemsetAsync(d_cur, 0, 16, ctx->stream);
    if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_graph_position_map: alloc"));
    if (v.n_edges) hipLaunchKernelGGL(k_pos_reserve, dim3(ggrid(ctx, v.n_edges)), dim3(BLOCK), 0, ctx->stream, v, first, &d_cur[0]);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h_cur, d_cur, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_graph_position_map: reserve"));
    // the reference's own check, printed side by side at :117: size == sum of edge lengths + nodes - edges
    const u64 total = g->live_nodes + h_cur[0];
    if (h_cur[0] != g->live_len - g->live_edges) return done(fail(ctx, GK_E_STATE, "gk_graph_position_map: interior k-mer count does not match the graph's counters"));
    if (total == 0) return done(GK_OK);
    e = hipMalloc((void **)&lo, total * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&hi, total * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&val, total * 8);
    if (e == hipSuccess && g->W == 1) e = hipMemsetAsync(hi, 0, total * 8, ctx->stream);
    if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_graph_position_map: entries"));
    hipLaunchKernelGGL(k_pos_nodes, dim3(ggrid(ctx, std::max<u64>(v.n_nodes, 1))), dim3(BLOCK), 0, ctx->stream, v, lo, hi, val, &d_cur[1]);
    if (v.n_edges) {
        if (g->W == 1) {
            hipLaunchKernelGGL(k_pos_fill<1>, dim3(ggrid(ctx, v.n_edges)), dim3(BLOCK), 0, ctx->stream, v, g->k, first, g->live_nodes, lo, hi, val);
            hipLaunchKernelGGL(k_pos_fill_long<1>, dim3((int)std::min<u64>(v.n_edges, (u64)ctx->cu_count * 8)), dim3(BLOCK), 0, ctx->stream, v, g->k, first, g->live_nodes, lo, hi, val);
        } else {
            hipLaunchKernelGGL(k_pos_fill<2>, dim3(ggrid(ctx, v.n_edges)), dim3(BLOCK), 0, ctx->stream, v, g->k, first, g->live_nodes, lo, hi, val);
            hipLaunchKernelGGL(k_pos_fill_long<2>, dim3((int)std::min<u64>(v.n_edges, (u64)ctx->cu_count * 8)), dim3(BLOCK), 0, ctx->stream, v, g->k, first, g->live_nodes, lo, hi, val);
        }
    }
    e = hipGetLastError();
    if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_graph_position_map: fill"));
    if (int rc = vmap_put_new_dev(vm, lo, hi, val, total)) return done(rc);
    if (entries) *entries = total;
    return done(GK_OK);
}

// first live node holding this k-mer (NONE = 0xffffffff if there is none) and, if base is 0..3, its out-edge for that first base
int gk_graph_node_lookup(gk_graph *g, uint64_t lo, uint64_t hi, int base, uint32_t *node_id, uint32_t *edge_id) {
    if (int rc = check_graph(g)) return rc;
    gk_ctx *ctx = g->ctx;
    if (int rc = graph_ensure_index(g)) return rc;
    u32 *d = nullptr, h[2] = {NONE, NONE};
    GK_HIP(ctx, hipMalloc((void **)&d, 8));
    if (g->W == 1) hipLaunchKernelGGL(k_node_lookup<1>, dim3(1), dim3(1), 0, ctx->stream, g->v, lo, hi, base, d);
    else hipLaunchKernelGGL(k_node_lookup<2>, dim3(1), dim3(1), 0, ctx->stream, g->v, lo, hi, base, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return hip_fail(ctx, e, "gk_graph_node_lookup");
    if (node_id) *node_id = h[0];
    if (edge_id) *edge_id = h[1];
    return GK_OK;
}

// MapGraph.addNode(seq) (Graph.scala:172-176): a fresh node without edges; several nodes may carry the same sequence
int gk_graph_add_node(gk_graph *g, uint64_t lo, uint64_t hi, uint32_t *node_id) {
    if (int rc = check_graph(g)) return rc;
    gk_ctx *ctx = g->ctx;
    GraphView &v = g->v;
    if (g->W == 1 ? ((lo >> (2 * g->k)) != 0 || hi != 0) : (g->k < 64 && (hi >> (2 * (g->k - 32))) != 0))
        return fail(ctx, GK_E_KLEN, "gk_graph_add_node: not a " + std::to_string(g->k) + "-mer");
    if (v.n_nodes + 1 >= (u64)NONE) return fail(ctx, GK_E_CAPACITY, "more than 2^32 graph nodes");
    if (v.n_nodes + 1 > g->node_cap) { if (int rc = graph_grow_nodes(g, std::max<u64>(g->node_cap * 2, 16))) return rc; }
    const u32 n = (u32)v.n_nodes;
    g->epoch++;
    hipLaunchKernelGGL(k_add_node, dim3(1), dim3(1), 0, ctx->stream, v, n, lo, hi);
    v.n_nodes++;
    g->live_nodes++;
    if (!g->index_ready) {
        // (no index yet: the first query builds it, this node included)
    } else if (2 * v.n_nodes > v.nidx_mask) {
        if (int rc = graph_build_index(g)) return rc;          // the index outgrew its table: rebuild (power of two >= 2 n)
    } else {
        const u64 h = g->W == 1 ? slot_hash(Kmer<1>{lo}) : slot_hash(Kmer<2>{lo, hi});
        hipLaunchKernelGGL(k_nidx_insert, dim3(1), dim3(1), 0, ctx->stream, v, n, h);
    }
    GK_HIP(ctx, hipGetLastError());
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (node_id) *node_id = n;
    return GK_OK;
}

static int graph_point_edit(gk_graph *g, bool start, uint32_t edge_id, uint32_t node_id) {
    if (int rc = check_graph(g)) return rc;
    gk_ctx *ctx = g->ctx;
    g->epoch++;
    int *d = nullptr, h = 1;
    GK_HIP(ctx, hipMalloc((void **)&d, 4));
    if (start) hipLaunchKernelGGL(k_replace_start, dim3(1), dim3(1), 0, ctx->stream, g->v, edge_id, node_id, d);
    else hipLaunchKernelGGL(k_replace_end, dim3(1), dim3(1), 0, ctx->stream, g->v, edge_id, node_id, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return hip_fail(ctx, e, "graph edit");
    if (h) return fail(ctx, GK_E_INVALID, std::string(start ? "gk_graph_replace_start" : "gk_graph_replace_end") + ": no such live edge / node");
    return GK_OK;
}
int gk_graph_replace_start(gk_graph *g, uint32_t edge_id, uint32_t new_start_node) { return graph_point_edit(g, true, edge_id, new_start_node); }   // :197-202
int gk_graph_replace_end(gk_graph *g, uint32_t edge_id, uint32_t new_end_node) { return graph_point_edit(g, false, edge_id, new_end_node); }       // :204-209

int gk_graph_nodes_by_id(gk_graph *g, const uint32_t *ids, uint64_t n, uint64_t *lo, uint64_t *hi, uint8_t *alive, uint32_t *in_deg, uint32_t *out_deg) {
    if (int rc = check_graph(g)) return rc;
    gk_ctx *ctx = g->ctx;
    if (n == 0) return GK_OK;
    if (!ids || !lo || !hi || !alive || !in_deg || !out_deg) return fail(ctx, GK_E_INVALID, "gk_graph_nodes_by_id: null argument");
    u32 *d_ids = nullptr, *d_in = nullptr, *d_out = nullptr;
    u64 *d_lo = nullptr, *d_hi = nullptr;
    uint8_t *d_al = nullptr;
    hipError_t e = hipMalloc((void **)&d_ids, n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_in, n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_lo, n * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_hi, n * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_al, n);
    if (e == hipSuccess) e = hipMemcpyAsync(d_ids, ids, n * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_nodes_by_id, dim3(ggrid(ctx, n)), dim3(BLOCK), 0, ctx->stream, g->v, d_ids, n, d_lo, d_hi, d_al, d_in, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(lo, d_lo, n * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hi, d_hi, n * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(alive, d_al, n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(in_deg, d_in, n * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out_deg, d_out, n * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    for (void *p : {(void *)d_ids, (void *)d_in, (void *)d_out, (void *)d_lo, (void *)d_hi, (void *)d_al}) if (p) (void)hipFree(p);
    if (e != hipSuccess) return hip_fail(ctx, e, "gk_graph_nodes_by_id");
    return GK_OK;
}

int gk_graph_edges_by_id(gk_graph *g, const uint32_t *ids, uint64_t n, uint32_t *start_node, uint32_t *end_node, uint64_t *len, uint8_t *first_base, uint8_t *alive) {
    if (int rc = check_graph(g)) return rc;
    gk_ctx *ctx = g->ctx;
    if (n == 0) return GK_OK;
    if (!ids || !start_node || !end_node || !len || !first_base || !alive) return fail(ctx, GK_E_INVALID, "gk_graph_edges_by_id: null argument");
    u32 *d_ids = nullptr, *d_s = nullptr, *d_e = nullptr;
    u64 *d_len = nullptr;
    uint8_t *d_f = nullptr, *d_al = nullptr;
    hipError_t e = hipMalloc((void **)&d_ids, n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_s, n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_e, n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_len, n * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_f, n);
    if (e == hipSuccess) e = hipMalloc((void **)&d_al, n);
    if (e == hipSuccess) e = hipMemcpyAsync(d_ids, ids, n * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_edges_by_id, dim3(ggrid(ctx, n)), dim3(BLOCK), 0, ctx->stream, g->v, d_ids, n, d_s, d_e, d_len, d_f, d_al);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(start_node, d_s, n * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(end_node, d_e, n * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(len, d_len, n * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(first_base, d_f, n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(alive, d_al, n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    for (void *p : {(void *)d_ids, (void *)d_s, (void *)d_e, (void *)d_len, (void *)d_f, (void *)d_al}) if (p) (void)hipFree(p);
    if (e != hipSuccess) return hip_fail(ctx, e, "gk_graph_edges_by_id");
    return GK_OK;
}

}  // extern "C"
