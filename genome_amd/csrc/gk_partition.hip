// gk_partition.hip — the partitioned form of FreqFilter.add / DNAMap.update(key, 1, _+1):
// radix-partition a batch of canonical k-mers by table SEGMENT, then let one workgroup per segment
// build its 32 KiB piece of the table in LDS and stream it back.
//
// Why: the direct path (k_count_reads / k_add_keys) costs one memory-side atomic per distinct key
// and one 64-B sector per 8-B touch — 15.8 GB of HBM traffic for 2.9 GB of algorithmic bytes at C2
// (profiles/r01/pmc_count_reads_v2.json), and the chip's random-atomic rate (17.6 G/s) is the
// ceiling.  Here every byte moves in coalesced 8/16-B-per-lane streams and all read-modify-write
// happens in LDS:
//
//   P1 k_part_hist1     source -> 256-bin histogram of the L1 bucket (top bits of the slot hash)
//   -- k_part_prefix1   exclusive scan (1 workgroup): L1 region bases, chunk / range tables for P3/P4
//   P2 k_part_scatter1  source -> keys written into their L1 region (tile histogram in LDS, one
//                       global atomic per (tile, bucket) reserves a run, lanes fill it by LDS rank)
//   P3 k_part_hist2r    L1 region, cut into RANGES of <= 16 chunks: per-range histogram of the fine
//                       bucket (= segment inside the L1 bucket) -> one row of the range matrix
//   -- k_part_scan2     down every column of the matrix: row r gets the keys of its bin in EARLIER
//                       ranges, the column total is the segment's key count;  k_part_prefix2: segment bases
//   P4 k_part_scatter2  L1 region chunks -> keys in segment order.  EXACT fine level: a range keeps a
//                       running destination per bin in LDS (from the matrix) — no global atomics;
//                       OVER-PROVISIONED fine level: one returning atomic per (chunk, bin) on a fixed region
//   P5 k_seg_insert     one workgroup per segment: load the segment into LDS (or start from
//                       EMPTY when the table is known to be empty), insert the segment's keys with
//                       LDS atomics (same probe sequence as gk::table_add), store it back.
//
// Three forms.
//   EXACT (ragged read streams): all passes; the histograms size every region exactly.
//   OVER-PROVISIONED (fixed-stride records, key arrays — the key count is known up front; near-distinct
//     keys): hashed keys spread binomially, so every L1 bucket and every segment gets a fixed region of
//     mean + 8 sigma + slack keys, P1 and P3 disappear, and P2 (k_op_scatter1_reads) extracts each window
//     ONCE, parking the tile's keys in LDS while it counts and reserves.  Keys that do not fit a region
//     go to a spill list replayed through the direct path after P5.
//   HYBRID (fixed-stride records / key arrays with many REPEATS — sequencing coverage): L1 regions
//     over-provisioned (bucket sizes stay balanced: thousands of k-mers per bucket), fine level exact
//     through P3, because a segment holds a handful of high-multiplicity k-mers and its key count is
//     anything but binomial.  P2 also keeps a 1/1024 hash SAMPLE of the distinct keys it has seen
//     (a small device set that lives as long as the map's contents), from which the host learns how many NEW
//     distinct keys the batch brings and sizes the table for them before P4 — not for its windows.
//
// The source is either a `.bin` read stream (extract + canonicalise on the fly, FreqFilter.scala:28-36)
// or an array of already-routed keys (the owner side of the all-to-all).  Results are identical
// to the direct path: same slots layout, same probing; only the order of insertion differs, which
// is unobservable (parity is on sorted content).
//
// Algorithmic bytes per occurrence (k<=31): P1 0.31, P2 0.31 + 8, P3 8, P4 8 + 8, P5 8 + slot
// traffic (32 B of table per slot streamed in and out, or 16 B out only from empty).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>
#include <type_traits>
#include <vector>

#include "gk_internal.h"
#include "gk_tile.h"

using namespace gk;

GK_TIMERS_DEFINE(partition)

static constexpr int PBLOCK = 512;          // threads of the key-streaming kernels
static constexpr int KEYS_PER_THREAD = 8;
static constexpr int PTILE_READS = 256;     // reads per LDS tile in P1/P2 (runs of ~120 keys per bucket)
static constexpr int PTILE_WORDS = PTILE_READS * 65 / 4 + 64;
static constexpr int TILE2 = PBLOCK * KEYS_PER_THREAD;   // keys per chunk in P3/P4
static constexpr u32 MAXB1 = 1u << MAX_LNB1;      // L1 buckets at most (256 for every table up to 34 GB, see plan_segments)
static constexpr u32 MAX_NB2 = 4096;        // LDS bound in P3/P4 (16 B per fine bucket on top of the sorted chunk)
#ifndef GK_MAX_RANGE_CHUNKS
#define GK_MAX_RANGE_CHUNKS 96
#endif
// a range = up to 96 consecutive chunks of one L1 bucket (384 Ki keys = 32 sorts of 12288 keys, 48 of 8192: no partial sort at a
// range's end; measured at C3, P4, three batches (round 2): 16 chunks 41.7-44.9 ms, 24: 38.5-40.0, 48: 38.9-41.9, 96: 38.5;
// one batch (round 3, 23 M keys per L1 bucket): 24: 35.5-36.3, 48: 35.1-36.4, 96: 33.6-34.6.  Small batches keep enough
// ranges to fill the chip: part_plan bounds the range by the batch's chunks per CU)
static constexpr u32 MAX_RANGE_CHUNKS = GK_MAX_RANGE_CHUNKS;

// The distinct-key SAMPLE: a key belongs to it iff bits 11..20 of its slot hash are zero (1 key in 1024, independent
// of the bits that pick its segment and its start slot); the set stores the 64-bit slot hash itself (a bijection of
// the key for k <= 31, a 64-bit fingerprint of it for k >= 34).  Claims since the last gk_map_clear x 1024 estimate
// the distinct keys the map has been offered.
static constexpr u32 SAMPLE_SHIFT = 11, SAMPLE_MASK = 1023u;
struct Sampler {
    u64 *set;                       // nullptr: sampling off
    u64 mask;                       // capacity - 1 (power of two)
    unsigned long long *claims;     // Counters::sample_claims
};
__device__ __forceinline__ u32 sample_key(const Sampler &sp, u64 h) {
    if (!sp.set || ((h >> SAMPLE_SHIFT) & SAMPLE_MASK) != 0u || h == ~0ull) return 0u;
    u64 i = (h >> 21) & sp.mask;
    for (u32 n = 0; n < 4096u; n++) {                 // bounded: a full set stops learning (the host then assumes "all distinct")
        const u64 old = atomicCAS(reinterpret_cast<unsigned long long *>(&sp.set[i]), ~0ull, (unsigned long long)h);
        if (old == ~0ull) return 1u;
        if (old == h) return 0u;
        i = (i + 1) & sp.mask;
    }
    return 0u;
}

struct PartArrays {
    // (per L1 bucket: MAXB1 entries, prefixes MAXB1 + 1; entries from the table's own bucket count up hold the total)
    unsigned long long *hist1;      // [MAXB1]
    unsigned long long *l1_base;    // [MAXB1 + 1] dense prefix of the keys kept per L1 bucket (= region bases in exact mode)
    unsigned long long *cursor1;    // [MAXB1]
    unsigned long long *cbase;      // [MAXB1 + 1] chunk prefix
    unsigned long long *rbase;      // [MAXB1 + 1] range prefix
    u32 *hist2;                     // [nseg] keys per segment (exact fine level)
    unsigned long long *fine_base;  // [nseg + 1]
    u32 *cursor2;                   // [nseg] (over-provisioned fine level)
    u32 *rmat;                      // [nranges x nb2] range matrix (exact fine level)
    u32 *failed;                    // [nseg] list of segments that overflowed
    u32 *n_failed;
    // op1 = 1: L1 bucket b owns the fixed region [b * cap1, (b + 1) * cap1) of bufA (no P1); op2 = 1: segment s owns
    // the fixed region [s * cap2, (s + 1) * cap2) of bufB (no P3).  What does not fit goes to the spill list.
    int op1, op2;
    unsigned long long cap1, cap2;  // keys per L1 region / per segment region
    u32 stripe_nb1;                 // op1: number of L1 regions interleaved in bufA (see l1_slot)
    u32 range_chunks;               // TILE2-key chunks per range (exact fine level)
    u32 chunk_keys;                 // keys per chunk of the over-provisioned fine level's P4 (TILE2, or 2 x TILE2 with 1024 threads)
    u64 *spill;                     // [spill_cap * W]
    unsigned long long *nspill;
    unsigned long long spill_cap;
    u32 *overflow;                  // spill list itself overflowed: abandon the pipeline (scratch only so far)
    u32 *noncanon;                  // key-array source taken verbatim from the caller: flag keys that are not canonical (nullptr: trusted)
    int k;
    // P4 over a SLICE of every L1 region (pipelined pieces, see part_run): keys [l1_from[b], l1_to[b]) of bucket b, with the
    // chunk table in cbase counting the slice's chunks only.  nullptr: the whole region.
    const unsigned long long *l1_from, *l1_to;
};
__device__ __forceinline__ u64 l1_begin(const PartArrays &a, u32 b) { return a.op1 ? 0ull : a.l1_base[b]; }
// Over-provisioned L1: where key number `pos` of L1 region `b` lives in bufA.  From 1.5 GB up (part_prepare sets
// stripe_nb1) the regions are INTERLEAVED in blocks of L1_BLK keys (stripe s = block s of every region), not laid end to end: every tile of P2 appends to all
// regions at once and all regions fill at the same pace, so the pages being written are the current stripe or two
// (8-16 MB) instead of one page per region spread over the whole buffer — with 3.8 GB of regions laid end
// to end 73 % of P2's address translations missed the per-CU TLB and the kernel took twice the time per key.
static constexpr u32 L1_BLK = 4096;         // = one P4 chunk (TILE2), so that a chunk is still one contiguous 32/64 KiB read
__device__ __forceinline__ u64 l1_slot(const PartArrays &a, u32 b, u64 pos) {
    if (!a.stripe_nb1) return (u64)b * a.cap1 + pos;         // small buffer: end to end (no TLB pressure, and 3 % faster at C2)
    return ((pos / L1_BLK) * a.stripe_nb1 + b) * L1_BLK + (pos % L1_BLK);
}
// index in bufA of key `pos` of L1 bucket b, whichever way the L1 level was built
__device__ __forceinline__ u64 l1_key_index(const PartArrays &a, u32 b, u64 pos) { return a.op1 ? l1_slot(a, b, pos) : a.l1_base[b] + pos; }
__device__ __forceinline__ u64 l1_count(const PartArrays &a, u32 b) {
    return a.op1 ? min(a.cursor1[b], a.cap1) : a.l1_base[b + 1] - a.l1_base[b];
}
// Load a word that is the same for the whole workgroup and was written by an EARLIER kernel through
// the scalar cache (constant address space): it then counts on lgkmcnt, not vmcnt, so waiting for it
// does not also wait for every vector store the wave still has in flight.
template <class T> __device__ __forceinline__ T uniform_load(const T *p) {
    return *(const __attribute__((address_space(4))) T *)(p);
}
__device__ __forceinline__ u64 fine_begin_u(const PartArrays &a, u64 s) { return a.op2 ? s * a.cap2 : uniform_load(&a.fine_base[s]); }
__device__ __forceinline__ u64 fine_count_u(const PartArrays &a, u64 s) {
    return a.op2 ? min((unsigned long long)uniform_load(&a.cursor2[s]), a.cap2)
                 : uniform_load(&a.fine_base[s + 1]) - uniform_load(&a.fine_base[s]);
}
template <int W> __device__ __forceinline__ void spill_key(const PartArrays &a, u64 w0, u64 w1) {
    const unsigned long long si = atomicAdd(a.nspill, 1ull);
    if (si < a.spill_cap) {
        if constexpr (W == 1) a.spill[si] = w0;
        else { a.spill[2 * si] = w0; a.spill[2 * si + 1] = w1; }
    } else {
        *a.overflow = 1;
    }
}

template <int W> __device__ __forceinline__ Kmer<W> load_key(const u64 *keys, u64 i) {
    if constexpr (W == 1) return Kmer<1>{keys[i]};
    else return Kmer<2>{keys[2 * i], keys[2 * i + 1]};
}
template <int W> __device__ __forceinline__ void store_key(u64 *keys, u64 i, Kmer<W> x) {
    if constexpr (W == 1) keys[i] = x.lo;
    else { keys[2 * i] = x.lo; keys[2 * i + 1] = x.hi; }
}

// block-wide (PBLOCK threads) in-place exclusive scan of arr[0..n) in LDS; returns nothing, arr[i]
// becomes the sum of the elements before i.  wsum: PBLOCK/64 scratch words.
template <int NT> __device__ __forceinline__ void block_scan_inplace(u32 *arr, u32 n, u32 *wsum) {
    const u32 per = (n + NT - 1) / NT;
    const u32 b0 = threadIdx.x * per, b1 = min(b0 + per, n);
    u32 sum = 0;
    for (u32 b = b0; b < b1; b++) sum += arr[b];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 inc = sum;
    for (int d = 1; d < 64; d <<= 1) { u32 tt = __shfl_up(inc, d); if (lane >= d) inc += tt; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    u32 pre = 0;
    for (int w = 0; w < wave; w++) pre += wsum[w];
    u32 run = pre + inc - sum;
    for (u32 b = b0; b < b1; b++) { const u32 c = arr[b]; arr[b] = run; run += c; }
    __syncthreads();
}

// workgroup total of a per-thread count -> ONE global atomic (hist0: one LDS word the caller no longer needs)
__device__ __forceinline__ void block_add_global(u32 v, u32 *hist0, unsigned long long *dst) {
    for (int d = 32; d; d >>= 1) v += __shfl_down(v, d);
    __syncthreads();
    if (threadIdx.x == 0) *hist0 = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(hist0, v);
    __syncthreads();
    if (threadIdx.x == 0 && *hist0) atomicAdd(dst, (unsigned long long)*hist0);
}

// ---------------------------------------------------------------------------------------------
// P1 / P2 from a read stream
// ---------------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(PBLOCK) void k_part_hist1_reads(const uint8_t *__restrict__ rec, u64 nreads, const u32 *__restrict__ offsets,
                                                            u32 stride, int k, int group, int max_len, Table<W> t, unsigned long long *hist1,
                                                            Sampler sp, Counters *ctr) {
    __shared__ __attribute__((aligned(16))) u32 tile[PTILE_WORDS];
    __shared__ u32 hist[MAXB1];
    for (u32 b = threadIdx.x; b < MAXB1; b += PBLOCK) hist[b] = 0;
    u32 occ = 0, claims = 0;
    const u64 ntiles = (nreads + PTILE_READS - 1) / PTILE_READS;
    const WindowLimits lim{max_len, &ctr->format};
    for (u64 tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const u64 r0 = tl * PTILE_READS;
        const int nr = (int)min((u64)PTILE_READS, nreads - r0);
        const u64 gb = offsets ? (u64)offsets[r0] : r0 * stride, ge = offsets ? (u64)offsets[r0 + nr] : (r0 + nr) * stride;
        __syncthreads();
        const u64 a0 = stage_tile(tile, rec, gb, ge);
        __syncthreads();
        for_each_window<W>(tile, a0, r0, nr, offsets, stride, k, group, lim, [&](Kmer<W> x) {
            const u64 h = slot_hash(canonical(x, k));
            atomicAdd(&hist[seg_l1(t, h)], 1u);
            claims += sample_key(sp, h);
            occ++;
        });
    }
    __syncthreads();
    for (u32 b = threadIdx.x; b < MAXB1; b += PBLOCK) if (hist[b]) atomicAdd(&hist1[b], (unsigned long long)hist[b]);
    __syncthreads();
    block_add_global(occ, &hist[0], &ctr->occurrences);
    if (sp.set) block_add_global(claims, &hist[0], sp.claims);
}

template <int W>
__global__ __launch_bounds__(PBLOCK) void k_part_scatter1_reads(const uint8_t *__restrict__ rec, u64 nreads, const u32 *__restrict__ offsets,
                                                               u32 stride, int k, int group, int max_len, Table<W> t, const unsigned long long *l1_base,
                                                               unsigned long long *cursor1, u64 *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) u32 tile[PTILE_WORDS];
    __shared__ u32 hist[MAXB1];
    __shared__ unsigned long long base[MAXB1];
    const u64 ntiles = (nreads + PTILE_READS - 1) / PTILE_READS;
    const WindowLimits lim{max_len, nullptr};            // P1 has already reported oversized length bytes
    for (u64 tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const u64 r0 = tl * PTILE_READS;
        const int nr = (int)min((u64)PTILE_READS, nreads - r0);
        const u64 gb = offsets ? (u64)offsets[r0] : r0 * stride, ge = offsets ? (u64)offsets[r0 + nr] : (r0 + nr) * stride;
        __syncthreads();
        for (u32 b = threadIdx.x; b < MAXB1; b += PBLOCK) hist[b] = 0;
        const u64 a0 = stage_tile(tile, rec, gb, ge);
        __syncthreads();
        for_each_window<W>(tile, a0, r0, nr, offsets, stride, k, group, lim, [&](Kmer<W> x) {
            atomicAdd(&hist[seg_l1(t, slot_hash(canonical(x, k)))], 1u);
        });
        __syncthreads();
        for (u32 b = threadIdx.x; b < MAXB1; b += PBLOCK) {
            const u32 c = hist[b];
            if (c) base[b] = l1_base[b] + atomicAdd(&cursor1[b], (unsigned long long)c);
            hist[b] = 0;                    // becomes the rank counter
        }
        __syncthreads();
        for_each_window<W>(tile, a0, r0, nr, offsets, stride, k, group, lim, [&](Kmer<W> x) {
            const Kmer<W> y = canonical(x, k);
            const u32 b = seg_l1(t, slot_hash(y));
            store_key<W>(out, base[b] + atomicAdd(&hist[b], 1u), y);
        });
    }
}

// ---------------------------------------------------------------------------------------------
// P1 / P2 from a key array (keys already canonical and routed)
// ---------------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(BLOCK) void k_part_hist1_keys(const u64 *__restrict__ keys, u64 n, Table<W> t, unsigned long long *hist1) {
    __shared__ u32 hist[MAXB1];
    for (u32 b = threadIdx.x; b < MAXB1; b += BLOCK) hist[b] = 0;
    __syncthreads();
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK)
        atomicAdd(&hist[seg_l1(t, slot_hash(load_key<W>(keys, i)))], 1u);
    __syncthreads();
    for (u32 b = threadIdx.x; b < MAXB1; b += BLOCK) if (hist[b]) atomicAdd(&hist1[b], (unsigned long long)hist[b]);
}

// Chunked scatter of a key range into `nbins` bins, used for P2 from keys (bin = L1 bucket) and P4
// (bin = fine bucket inside one L1 bucket).  The chunk is SORTED BY BIN IN LDS first and then
// written out linearly, so that consecutive lanes store consecutive addresses: a store instruction
// whose 64 lanes hit 64 different lines is issued one line at a time (the scattered form of this
// kernel was bound by exactly that, whatever the run length), while a sorted write-out touches a
// handful of lines per instruction.  LDS (dynamic): sorted keys [TILE2 * W], bin of each sorted
// position [TILE2] u16, per-bin offset [nbins] u32, per-bin destination [nbins] u64.
// RANGED (level 2, exact fine level): gb[] already holds this range's running destination of every bin (from the
// range matrix), so nothing is reserved — no global atomic at all — and gb[] advances by the chunk's counts.
// A chunk's keys into registers: all loads unconditionally (index clamped; cnt >= 1): one memory round trip per chunk
// instead of one per key — a load inside `if (i < cnt)` is sunk next to its use and serialises.  P4 calls this for the
// NEXT chunk before it sorts the current one, so the round trip hides behind a chunk's worth of LDS work.
// lbase (LEVEL 2, exact L1 level only): l1_base[b1], handed in by the caller from its LDS copy — a global load of it here
// would put a full vmcnt(0) wait, i.e. a wait for the previous chunk's stores, in front of the prefetch.
template <int W, int LEVEL, int NT, int KPT>
__device__ __forceinline__ void load_chunk(Kmer<W> (&key)[KPT], const u64 *__restrict__ in, const PartArrays &a, u32 b1, u64 lbase,
                                           u64 begin, u32 cnt) {
#pragma unroll
    for (int j = 0; j < KPT; j++) {
        const u32 i = threadIdx.x + j * NT;
        const u64 src = begin + (i < cnt ? i : cnt - 1);
        key[j] = load_key<W>(in, LEVEL == 2 ? (a.op1 ? l1_slot(a, b1, src) : lbase + src) : src);
    }
}
// prefetch(): the caller's request for the NEXT chunk's keys, issued before this chunk is sorted.  On gfx9-family parts loads
// and stores share vmcnt and may complete out of order with respect to each other, so any wait for a load while stores are
// pending is a full vmcnt(0), stores included.  The prefetched keys are therefore waited for HERE, explicitly, right before
// the write-out issues its stores (they were requested a sort earlier and have long landed): the next chunk then starts
// without waiting for this chunk's stores.
struct NoPrefetch { __device__ __forceinline__ void operator()() const {} };
static constexpr bool is_prefetching(const NoPrefetch &) { return false; }
template <class F> static constexpr bool is_prefetching(const F &) { return true; }
template <int W, int LEVEL, bool RANGED, int NT, class Prefetch = NoPrefetch, int KPT = KEYS_PER_THREAD>
__device__ __forceinline__ void scatter_chunk(const Kmer<W> (&key)[KPT], u32 cnt, const Table<W> &t, u32 nbins,
                                              u64 *sorted, uint16_t *binof, u32 *off, u32 *lim, unsigned long long *gb, u32 *wsum,
                                              const PartArrays &a, u64 bin0, u64 *__restrict__ out, const Sampler &sp, u32 &claims,
                                              GK_TARGS_DECL, Prefetch prefetch = Prefetch()) {
    u32 bin[KPT], rank[KPT];
    prefetch();
    for (u32 b = threadIdx.x; b < nbins; b += NT) off[b] = 0;
    __syncthreads();
    GK_TICK(0);
#pragma unroll
    for (int j = 0; j < KPT; j++) {
        const u32 i = threadIdx.x + j * NT;
        bin[j] = 0xffffffffu;
        if (i < cnt) {
            const u64 h = slot_hash(key[j]);
            bin[j] = LEVEL == 1 ? seg_l1(t, h) : seg_fine(t, h);
            rank[j] = atomicAdd(&off[bin[j]], 1u);             // rank inside its bin
            if (LEVEL == 1) {
                claims += sample_key(sp, h);
                if (a.noncanon && !(canonical(key[j], a.k) == key[j])) *a.noncanon = 1u;
            }
        }
    }
    GK_TICK(1);
    __syncthreads();
    GK_TICK(2);
    for (u32 b = threadIdx.x; b < nbins; b += NT) {        // reserve the bin's run in the output
        const u32 c = off[b];
        u32 fit = c;
        if (c && !RANGED) {
            const unsigned long long at = LEVEL == 1 ? atomicAdd(&a.cursor1[b], (unsigned long long)c)
                                                     : (unsigned long long)atomicAdd(&a.cursor2[bin0 + b], c);
            if (LEVEL == 1 ? a.op1 : a.op2) {
                const unsigned long long cap = LEVEL == 1 ? a.cap1 : a.cap2;
                gb[b] = LEVEL == 1 ? at : (bin0 + b) * cap + at;        // level 1: position inside the region (l1_slot maps it)
                fit = at >= cap ? 0u : (u32)min((unsigned long long)c, cap - at);
            } else {
                gb[b] = a.l1_base[b] + at;                              // level 1, exact
            }
        }
        lim[b] = fit;
    }
    __syncthreads();
    GK_TICK(3);
    block_scan_inplace<NT>(off, nbins, wsum);                      // counts -> offsets in the sorted chunk
    GK_TICK(4);
#pragma unroll
    for (int j = 0; j < KPT; j++)
        if (bin[j] != 0xffffffffu) {
            const u32 pos = off[bin[j]] + rank[j];
            store_key<W>(sorted, pos, key[j]);
            binof[pos] = (uint16_t)bin[j];
        }
    if (is_prefetching(prefetch)) __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0): the next chunk's keys are in
    __syncthreads();
    GK_TICK(5);
    for (u32 i = threadIdx.x; i < cnt; i += NT) {          // linear, coalesced write-out
        const u32 b = binof[i], j = i - off[b];
        const Kmer<W> x = load_key<W>(sorted, i);
        if (j < lim[b]) store_key<W>(out, LEVEL == 1 && a.op1 ? l1_slot(a, b, gb[b] + j) : gb[b] + j, x);
        else if constexpr (W == 1) spill_key<1>(a, x.lo, 0);
        else spill_key<2>(a, x.lo, x.hi);
    }
    GK_TICK(6);
    __syncthreads();
    if (RANGED) {                                              // the range's next chunk continues where this one ended
        for (u32 b = threadIdx.x; b < nbins; b += NT) gb[b] += lim[b];
    }
    GK_TICK(7);
}

// dynamic LDS carve for scatter_chunk
template <int W, int NT = PBLOCK, int KPT = KEYS_PER_THREAD> struct ScatterLds {
    static constexpr int TILE = NT * KPT;
    u64 *sorted; uint16_t *binof; u32 *off, *lim; unsigned long long *gb; u32 *wsum;
    __device__ __forceinline__ ScatterLds(unsigned long long *base, u32 nbins) {
        gb = base;                                                       // [nbins] u64
        sorted = reinterpret_cast<u64 *>(base + nbins);                  // [TILE * W] u64
        off = reinterpret_cast<u32 *>(sorted + (size_t)TILE * W);       // [nbins] u32
        lim = off + nbins;                                               // [nbins] u32
        wsum = lim + nbins;                                              // [NT / 64]
        binof = reinterpret_cast<uint16_t *>(wsum + NT / 64);        // [TILE] u16
    }
    static size_t bytes(u32 nbins) { return (size_t)nbins * 16 + (size_t)TILE * W * 8 + (NT / 64) * 4 + (size_t)TILE * 2 + 16; }
};

template <int W>
__global__ __launch_bounds__(PBLOCK) void k_part_scatter1_keys(const u64 *__restrict__ keys, u64 n, Table<W> t, PartArrays a,
                                                               u64 *__restrict__ out, Sampler sp) {
    extern __shared__ unsigned long long lds_dyn1[];
    const u32 nb1 = max(256u, 1u << t.lnb1);
    ScatterLds<W> L(lds_dyn1, nb1);
    const u64 nchunks = (n + TILE2 - 1) / TILE2;
    u32 claims = 0;
    GK_T0();
    for (u64 c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const u64 begin = c * TILE2;
        const u32 cnt = (u32)min((u64)TILE2, n - begin);
        Kmer<W> key[KEYS_PER_THREAD];
        load_chunk<W, 1, PBLOCK, KEYS_PER_THREAD>(key, keys, a, 0u, 0ull, begin, cnt);
        scatter_chunk<W, 1, false, PBLOCK>(key, cnt, t, nb1, L.sorted, L.binof, L.off, L.lim, L.gb, L.wsum, a, 0, out, sp, claims, GK_TARGS);
    }
    if (sp.set) block_add_global(claims, &L.off[0], sp.claims);
}

// P2 of the over-provisioned L1 level for fixed-stride records: ONE window-extraction pass.  The tile's
// canonical keys and their L1 buckets are parked in LDS while the per-bucket counts are built, the
// bucket regions are reserved (one global atomic per (tile, bucket)), then the keys go out.  No P1.
#ifndef GK_OP_CAP
#define GK_OP_CAP 5632
#endif
static constexpr int OP_CAP = GK_OP_CAP;    // LDS key buffer, in 64-bit words (5632: 2 workgroups per CU)
#ifndef GK_OP_TILE_READS
#define GK_OP_TILE_READS 128
#endif
static constexpr int OP_TILE_READS = GK_OP_TILE_READS;
static constexpr int OP_TILE_WORDS = OP_TILE_READS * 65 / 4 + 64;
// SORTED: the write-out walks the tile's keys in BUCKET order (a u16 permutation built in LDS), so that consecutive lanes store
// consecutive addresses of a bucket's run (a tile holds ~22 keys per bucket): a store instruction then touches a handful of
// lines instead of 64.  The unsorted form stores in window order, every lane into another bucket.
#ifndef GK_P2_FIRE_AND_RANK
#define GK_P2_FIRE_AND_RANK 1       // P2: the per-bucket reservation is fired before, not waited for in front of, the ranking of the tile's keys (0.60 -> 0.58 ms)
#endif
#ifndef GK_OP_MIN_WAVES
#define GK_OP_MIN_WAVES 1
#endif
#ifndef GK_OP_WGS_PER_CU
#define GK_OP_WGS_PER_CU 2          // workgroups of the persistent grid per CU: what the LDS tile and the VGPR count let reside
#endif
// NB1: L1 buckets the LDS arrays are sized for — 256 (every table up to 34 GB) or MAXB1.  The 1024-bucket form carries
// 15 KB more LDS per workgroup, so only one fits a CU beside its 45 KB key buffer; it exists so that tables beyond 34 GB
// stay on this pipeline at all (the direct path costs 45-62 ps per window).
template <int W, int NT, bool SORTED, int NB1>
__global__ __launch_bounds__(NT, GK_OP_MIN_WAVES) void k_op_scatter1_reads(const uint8_t *__restrict__ rec, u64 nreads, u32 stride, int k, int group,
                                                              int rs /* reads per tile */, int max_len, int exact_len, Table<W> t, PartArrays a,
                                                              Sampler sp, Counters *ctr, u64 *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) u32 tile[OP_TILE_WORDS];
    __shared__ u64 flat[OP_CAP];
    __shared__ uint16_t fbin[OP_CAP / W];      // 0xffff = hole (bucket ids use all 256 byte values)
    // (the 1024-bucket form keeps its per-bucket arrays narrow — a tile holds < 2^16 keys, a region < 2^32 — so that two
    //  workgroups still fit a CU: 79 KB each)
    using LimT = typename std::conditional<NB1 == 256, u32, uint16_t>::type;
    using GbT = typename std::conditional<NB1 == 256, unsigned long long, u32>::type;
    __shared__ u32 hist[NB1], rank[NB1];
    __shared__ LimT lim[NB1];
    __shared__ GbT gb[NB1];
    __shared__ uint16_t perm[SORTED ? OP_CAP / W : 1];
    __shared__ u32 wsum[NT / 64];
    u32 occ = 0, claims = 0;
    const int nk_max = max_len - k + 1;         // the host sized rs so that rs * nk_max keys fit `flat`; lengths are clamped to max_len
    const u64 ntiles = (nreads + rs - 1) / rs;
    const WindowLimits wl{max_len, &ctr->format, exact_len};
    // What bounds this kernel (0.60 ms per 1.2e8 windows) is NOT any one of the obvious suspects — timing-only builds, round 2
    // (profiles/r02/p2_elimination.md): no canonical form and no hash 0.63 ms; every store into one 8 KiB run 0.57 ms; eight
    // cursors per bucket instead of one 0.56 ms; the CU's second workgroup started half a tile late 0.60 ms; three or four
    // smaller workgroups per CU 0.61-0.75 ms; bucket-ordered write-out ("p2_sorted") 0.60 ms.
    // (Tried and dropped, round 2: requesting the NEXT tile's record bytes into registers before the extraction and parking
    //  them in the LDS tile after it.  The staging share of the phase timers fell from 14.5 % to 10.5 %, the kernel went from
    //  0.585 to 0.639 ms.  Smaller tiles with more workgroups per CU lose as well: 0.75 ms at 4224 keys, 1.05 ms at 2816.)
    GK_T0();
    for (u64 tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const u64 r0 = tl * rs;
        const int nr = (int)min((u64)rs, nreads - r0);
        const u32 nflat = (u32)(nr * nk_max);
        __syncthreads();
        GK_TICK(5);
        for (u32 b = threadIdx.x; b < (u32)NB1; b += NT) { hist[b] = 0; rank[b] = 0; }
        for (u32 i = threadIdx.x; i < nflat; i += NT) fbin[i] = 0xffff;
        const u64 a0 = stage_tile(tile, rec, r0 * stride, (r0 + nr) * stride);
        __syncthreads();
        GK_TICK(0);
        for_each_window_at<W>(tile, a0, r0, nr, stride, k, group, wl, [&](int r, int p, Kmer<W> x) {
            const Kmer<W> y = canonical(x, k);
            const u64 h = slot_hash(y);
            const u32 b = seg_l1(t, h);
            const u32 i = (u32)(r * nk_max + p);
            store_key<W>(flat, i, y);
            fbin[i] = (uint16_t)b;
            atomicAdd(&hist[b], 1u);
            claims += sample_key(sp, h);
            occ++;
        });
        GK_TICK(1);
        __syncthreads();
        GK_TICK(2);
#if GK_P2_FIRE_AND_RANK
        if constexpr (!SORTED && NB1 <= NT && OP_CAP / NT <= 12) {
            // The reservation is a returning global atomic per bucket (a round trip of a microsecond or two that the whole
            // workgroup used to sit out behind a barrier).  Fire it, take every key's rank inside its bucket from the LDS
            // counters meanwhile, and only then park what came back.
            constexpr int KPT = (OP_CAP + NT - 1) / NT;
            unsigned long long at = 0;
            const u32 c = threadIdx.x < (u32)NB1 ? hist[threadIdx.x] : 0u;
            if (c) at = atomicAdd(&a.cursor1[threadIdx.x], (unsigned long long)c);
            u32 jr[KPT];
#pragma unroll
            for (int q = 0; q < KPT; q++) {
                const u32 i = threadIdx.x + q * NT;
                const u32 b = i < nflat ? (u32)fbin[i] : 0xffffu;
                jr[q] = b != 0xffffu ? atomicAdd(&rank[b], 1u) : 0u;
            }
            if (threadIdx.x < (u32)NB1) {
                gb[threadIdx.x] = (GbT)at;
                lim[threadIdx.x] = (LimT)(!c || at >= a.cap1 ? 0u : (u32)min((unsigned long long)c, a.cap1 - at));
            }
            __syncthreads();
            GK_TICK(3);
#pragma unroll
            for (int q = 0; q < KPT; q++) {
                const u32 i = threadIdx.x + q * NT;
                const u32 b = i < nflat ? (u32)fbin[i] : 0xffffu;
                if (b == 0xffffu) continue;
                const Kmer<W> x = load_key<W>(flat, i);
                if (jr[q] < (u32)lim[b]) store_key<W>(out, l1_slot(a, b, (u64)gb[b] + jr[q]), x);
                else if constexpr (W == 1) spill_key<1>(a, x.lo, 0);
                else spill_key<2>(a, x.lo, x.hi);
            }
            GK_TICK(4);
            continue;
        }
#endif
        for (u32 b = threadIdx.x; b < (u32)NB1; b += NT) {
            const u32 c = hist[b];
            u32 fit = 0;
            if (c) {
                const unsigned long long at = atomicAdd(&a.cursor1[b], (unsigned long long)c);
                gb[b] = (GbT)at;                                    // position inside the region (l1_slot maps it; only used where at < cap1)
                fit = at >= a.cap1 ? 0u : (u32)min((unsigned long long)c, a.cap1 - at);
            }
            lim[b] = (LimT)fit;
            if (SORTED) rank[b] = c;
        }
        __syncthreads();
        GK_TICK(3);
        if constexpr (SORTED) {
            block_scan_inplace<NT>(rank, (u32)NB1, wsum);           // rank[b] = first sorted position of bucket b
            for (u32 i = threadIdx.x; i < nflat; i += NT) {
                const u32 b = fbin[i];
                if (b != 0xffff) perm[atomicAdd(&rank[b], 1u)] = (uint16_t)i;      // rank[b] ends as the END of bucket b
            }
            __syncthreads();
            const u32 ntot = rank[NB1 - 1];
            // four positions per thread and round: the chain perm -> (bin, key) -> (run start, limit, destination) is three LDS
            // round trips deep, one at a time it is all latency
            constexpr int U = 4;
            for (u32 p0 = threadIdx.x; p0 < ntot; p0 += U * NT) {
                u32 idx[U], bb[U], jj[U];
                Kmer<W> xx[U];
#pragma unroll
                for (int q = 0; q < U; q++) { const u32 pos = p0 + q * NT; idx[q] = pos < ntot ? (u32)perm[pos] : 0xffffffffu; }
#pragma unroll
                for (int q = 0; q < U; q++) {
                    bb[q] = idx[q] != 0xffffffffu ? (u32)fbin[idx[q]] : 0u;
                    if (idx[q] != 0xffffffffu) xx[q] = load_key<W>(flat, idx[q]);
                }
#pragma unroll
                for (int q = 0; q < U; q++) jj[q] = (p0 + q * NT) - (bb[q] ? rank[bb[q] - 1] : 0u);
#pragma unroll
                for (int q = 0; q < U; q++) {
                    if (idx[q] == 0xffffffffu) continue;
                    if (jj[q] < (u32)lim[bb[q]]) store_key<W>(out, l1_slot(a, bb[q], (u64)gb[bb[q]] + jj[q]), xx[q]);
                    else if constexpr (W == 1) spill_key<1>(a, xx[q].lo, 0);
                    else spill_key<2>(a, xx[q].lo, xx[q].hi);
                }
            }
        } else {
            // four keys per thread and round: each key is a chain LDS read -> LDS atomic -> LDS reads -> store, and with
            // four waves per SIMD one chain at a time leaves the LDS pipe idle most of the time
            constexpr int U = 4;
            for (u32 i0 = threadIdx.x; i0 < nflat; i0 += U * NT) {
                u32 b[U], j[U];
#pragma unroll
                for (int q = 0; q < U; q++) { const u32 i = i0 + q * NT; b[q] = i < nflat ? (u32)fbin[i] : 0xffffu; }
#pragma unroll
                for (int q = 0; q < U; q++) j[q] = b[q] != 0xffffu ? atomicAdd(&rank[b[q]], 1u) : 0u;
#pragma unroll
                for (int q = 0; q < U; q++) {
                    if (b[q] == 0xffffu) continue;
                    const Kmer<W> x = load_key<W>(flat, i0 + q * NT);
                    if (j[q] < (u32)lim[b[q]]) store_key<W>(out, l1_slot(a, b[q], (u64)gb[b[q]] + j[q]), x);
                    else if constexpr (W == 1) spill_key<1>(a, x.lo, 0);
                    else spill_key<2>(a, x.lo, x.hi);
                }
            }
        }
        GK_TICK(4);
    }
    GK_TFLUSH(0);
    block_add_global(occ, &hist[0], &ctr->occurrences);
    if (sp.set) block_add_global(claims, &hist[0], sp.claims);
}

// exclusive scan of the L1 counts; chunk and range tables for P3/P4 (chunks and ranges never straddle L1 buckets).
// One workgroup of MAXB1 threads, one per possible L1 bucket (buckets the table does not have count zero, so every prefix
// entry from the table's bucket count up holds the total).
__global__ __launch_bounds__(MAXB1) void k_part_prefix1(PartArrays a, u32 nb1) {
    __shared__ unsigned long long ws[MAXB1 / 64], wc[MAXB1 / 64], wr[MAXB1 / 64];
    const u32 i = threadIdx.x;
    const unsigned long long v = i < nb1 ? (a.op1 ? min(a.cursor1[i], a.cap1) : a.hist1[i]) : 0;
    const unsigned long long cv = (v + a.chunk_keys - 1) / a.chunk_keys;                        // P4's unit of work (over-provisioned fine level)
    const unsigned long long rv = ((v + TILE2 - 1) / TILE2 + a.range_chunks - 1) / a.range_chunks;   // ranges of range_chunks x TILE2 keys (exact fine level)
    // three exclusive scans: inclusive inside each wave by shuffles, then the wave totals
    // (this kernel sits between P2 and P4 on the critical path; the serial loop it replaces took 10 us)
    unsigned long long iv = v, icv = cv, irv = rv;
    const int lane = i & 63, wave = i >> 6;
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long t0 = __shfl_up(iv, d), t1 = __shfl_up(icv, d), t2 = __shfl_up(irv, d);
        if (lane >= d) { iv += t0; icv += t1; irv += t2; }
    }
    if (lane == 63) { ws[wave] = iv; wc[wave] = icv; wr[wave] = irv; }
    __syncthreads();
    unsigned long long p0 = 0, p1 = 0, p2 = 0;
    for (int w = 0; w < wave; w++) { p0 += ws[w]; p1 += wc[w]; p2 += wr[w]; }
    a.l1_base[i] = p0 + iv - v;
    a.cbase[i] = p1 + icv - cv;
    a.rbase[i] = p2 + irv - rv;
    if (i == MAXB1 - 1) {
        a.l1_base[MAXB1] = p0 + iv;
        a.cbase[MAXB1] = p1 + icv;
        a.rbase[MAXB1] = p2 + irv;
    }
}

// The same for ONE PIECE of a pipelined batch (over-provisioned L1 level): the slice of every L1 region that the scatters
// launched so far have filled beyond the previous piece's end — [prev_to[b], min(cursor1[b], cap1)) — and its chunk table.
// Runs on the scattering stream between two pieces' scatters, so the cursors it reads are quiescent.
__global__ __launch_bounds__(MAXB1) void k_part_prefix1_piece(PartArrays a, u32 nb1, const unsigned long long *prev_to, unsigned long long *from_out,
                                                              unsigned long long *to_out, unsigned long long *cbase_out) {
    __shared__ unsigned long long c[MAXB1 / 64];
    const u32 i = threadIdx.x;
    const unsigned long long to = i < nb1 ? min(a.cursor1[i], a.cap1) : 0ull;
    const unsigned long long from = (i < nb1 && prev_to) ? min(prev_to[i], to) : 0ull;
    const unsigned long long cv = (to - from + a.chunk_keys - 1) / a.chunk_keys;
    unsigned long long icv = cv;
    const int lane = i & 63, wave = i >> 6;
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long t1 = __shfl_up(icv, d);
        if (lane >= d) icv += t1;
    }
    if (lane == 63) c[wave] = icv;
    __syncthreads();
    unsigned long long p1 = 0;
    for (int w = 0; w < wave; w++) p1 += c[w];
    from_out[i] = from;
    to_out[i] = to;
    cbase_out[i] = p1 + icv - cv;
    if (i == MAXB1 - 1) cbase_out[MAXB1] = p1 + icv;
}

// which L1 bucket does chunk / range `c` belong to (binary search over the prefix of nb1 + 1 entries)
__device__ __forceinline__ u32 chunk_bucket(const unsigned long long *cbase, u64 c, u32 nb1) {
    u32 lo = 0, hi = nb1;
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (cbase[mid] <= c) lo = mid; else hi = mid;
    }
    return lo;
}

// ---------------------------------------------------------------------------------------------
// P3 (exact fine level): per-RANGE histogram of the fine bucket -> one row of the range matrix
// ---------------------------------------------------------------------------------------------
struct RangeGeom { u32 b1; u64 begin; u32 cnt; };      // L1 bucket, first key inside it, number of keys
__device__ __forceinline__ RangeGeom range_geom(const unsigned long long *s_rbase, const unsigned long long *s_l1n, u64 r, u32 range_chunks, u32 nb1) {
    RangeGeom g;
    g.b1 = chunk_bucket(s_rbase, r, nb1);
    g.begin = (r - s_rbase[g.b1]) * (u64)range_chunks * TILE2;
    g.cnt = (u32)min((u64)range_chunks * TILE2, s_l1n[g.b1] - g.begin);
    return g;
}

template <int W>
__global__ __launch_bounds__(PBLOCK) void k_part_hist2r(const u64 *__restrict__ bufA, Table<W> t, PartArrays a, u64 max_ranges) {
    extern __shared__ u32 lds_hist[];
    __shared__ unsigned long long s_rbase[MAXB1 + 1], s_l1n[MAXB1];
    const u32 nb1 = 1u << t.lnb1;
    for (u32 b = threadIdx.x; b <= nb1; b += PBLOCK) s_rbase[b] = a.rbase[b];
    for (u32 b = threadIdx.x; b < nb1; b += PBLOCK) s_l1n[b] = l1_count(a, b);
    __syncthreads();
    const u64 total = min((u64)s_rbase[nb1], max_ranges);
    for (u64 r = blockIdx.x; r < total; r += gridDim.x) {
        const RangeGeom g = range_geom(s_rbase, s_l1n, r, a.range_chunks, nb1);
        for (u32 b = threadIdx.x; b < t.nb2; b += PBLOCK) lds_hist[b] = 0;
        __syncthreads();
        for (u32 cb = 0; cb < g.cnt; cb += TILE2) {
            const u32 cnt = min((u32)TILE2, g.cnt - cb);
            Kmer<W> key[KEYS_PER_THREAD];
#pragma unroll
            for (int j = 0; j < KEYS_PER_THREAD; j++) {
                const u32 i = threadIdx.x + j * PBLOCK;
                key[j] = load_key<W>(bufA, l1_key_index(a, g.b1, g.begin + cb + (i < cnt ? i : cnt - 1)));
            }
#pragma unroll
            for (int j = 0; j < KEYS_PER_THREAD; j++)
                if (threadIdx.x + j * PBLOCK < cnt) atomicAdd(&lds_hist[seg_fine(t, slot_hash(key[j]))], 1u);
        }
        __syncthreads();
        u32 *row = a.rmat + r * t.nb2;
        for (u32 b = threadIdx.x; b < t.nb2; b += PBLOCK) row[b] = lds_hist[b];
        __syncthreads();
    }
}

// down every column (L1 bucket b1 = blockIdx.y, fine bucket b) of the range matrix: each row gets the keys of its
// bin that EARLIER ranges hold (its write offset inside the segment's run), the total is the segment's key count
__global__ __launch_bounds__(256) void k_part_scan2(PartArrays a, u32 nb2) {
    const u32 b1 = blockIdx.y, b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nb2) return;
    const u64 r0 = a.rbase[b1], r1 = a.rbase[b1 + 1];
    u32 run = 0;
    u64 r = r0;
    for (; r + 4 <= r1; r += 4) {                     // four independent loads in flight per step
        u32 *p = a.rmat + r * nb2 + b;
        const u32 v0 = p[0], v1 = p[nb2], v2 = p[2ull * nb2], v3 = p[3ull * nb2];
        p[0] = run; run += v0;
        p[nb2] = run; run += v1;
        p[2ull * nb2] = run; run += v2;
        p[3ull * nb2] = run; run += v3;
    }
    for (; r < r1; r++) {
        u32 *p = a.rmat + r * nb2 + b;
        const u32 v = *p;
        *p = run; run += v;
    }
    a.hist2[(u64)b1 * nb2 + b] = run;
}

// one workgroup per L1 bucket: fine_base[seg] = l1_base[b1] + exclusive scan of hist2 inside b1
__global__ __launch_bounds__(256) void k_part_prefix2(PartArrays a, u32 nb1, u32 nb2) {
    __shared__ unsigned long long s_carry;
    __shared__ u32 wsum[4];
    const u32 b1 = blockIdx.x;
    if (threadIdx.x == 0) s_carry = a.l1_base[b1];
    __syncthreads();
    for (u32 base = 0; base < nb2; base += 256) {
        const u32 i = base + threadIdx.x;
        const u32 v = i < nb2 ? a.hist2[(u64)b1 * nb2 + i] : 0;
        // block exclusive scan of v
        u32 inc = v;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int d = 1; d < 64; d <<= 1) { u32 tt = __shfl_up(inc, d); if (lane >= d) inc += tt; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        u32 pre = 0, tot = 0;
        for (int w = 0; w < 4; w++) { if (w < wave) pre += wsum[w]; tot += wsum[w]; }
        if (i < nb2) a.fine_base[(u64)b1 * nb2 + i] = s_carry + pre + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
    if (b1 == nb1 - 1 && threadIdx.x == 0) a.fine_base[(u64)nb1 * nb2] = s_carry;
}

// ---------------------------------------------------------------------------------------------
// P4: L1 regions -> keys in segment order
// ---------------------------------------------------------------------------------------------
// [b_lo, b_hi): the L1 buckets this launch covers (over-provisioned fine level only: a stripe of the batch, so that P5 of one
// stripe can run beside P4 of the next; the exact fine level always covers all of them)
// KPT: keys per thread and sort (8; 12 with 1024 threads for the exact fine level over MANY fine buckets: runs half again as long)
template <int W, bool RANGED, int NT, int KPT = KEYS_PER_THREAD>
__global__ __launch_bounds__(NT) void k_part_scatter2(const u64 *__restrict__ bufA, Table<W> t, PartArrays a, u64 max_units,
                                                          u64 *__restrict__ bufB, u32 b_lo, u32 b_hi) {
    extern __shared__ unsigned long long lds_dyn[];
    constexpr u32 TILE = NT * KPT;          // keys this workgroup sorts at a time (ranges and the chunk table stay in TILE2 units)
    // chunk / range table and L1 bucket extents: LDS copies (the binary search below was eight dependent global
    // loads per chunk: 16 % of the kernel by the phase timers); in the dynamic area, sized by the table's L1 bucket count
    // (part_tables_bytes), in front of the sort buffers
    const u32 nb1 = 1u << t.lnb1;
    unsigned long long *s_ubase = lds_dyn, *s_l1n = s_ubase + nb1 + 1, *s_l1b = s_l1n + nb1, *s_l1f = s_l1b + nb1;
    ScatterLds<W, NT, KPT> L(s_l1f + nb1, t.nb2);
    for (u32 b = threadIdx.x; b <= nb1; b += NT) s_ubase[b] = RANGED ? a.rbase[b] : a.cbase[b];
    for (u32 b = threadIdx.x; b < nb1; b += NT) {
        const bool slice = !RANGED && a.l1_to != nullptr;              // (a slice of every region: pipelined pieces)
        s_l1f[b] = slice ? a.l1_from[b] : 0ull;
        s_l1n[b] = slice ? a.l1_to[b] - a.l1_from[b] : l1_count(a, b);
        s_l1b[b] = l1_begin(a, b);
    }
    __syncthreads();
    if (!RANGED) b_hi = min(b_hi, nb1);
    const u64 first = RANGED ? 0ull : (u64)s_ubase[b_lo];
    const u64 total = min((u64)s_ubase[RANGED ? nb1 : b_hi], max_units);
    const Sampler nosp{nullptr, 0, nullptr};
    u32 noclaims = 0;
    GK_T0();
    // Software pipeline: a chunk is a chain global load -> LDS sort -> global store with six barriers in between, and two or
    // three workgroups per CU do not cover the load's round trip (phase timers: 46 % of the kernel was the load and the
    // barrier behind it).  The NEXT chunk's keys are requested into a second register set before the current chunk is
    // sorted (see scatter_chunk's prefetch hook); the two sets swap roles from chunk to chunk (a copy would wait for
    // the loads at once).
    if constexpr (RANGED) {
        // (the exact fine level keeps the plain form: it has no atomics, hence nothing in a chunk that waits on vmcnt but the
        //  loads themselves; measured at C3 and at C2 with fine_exact = 1, the prefetching form is 3-7 % SLOWER there)
        for (u64 r = blockIdx.x; r < total; r += gridDim.x) {
            const RangeGeom g = range_geom(s_ubase, s_l1n, r, a.range_chunks, nb1);
            const u64 bin0 = (u64)g.b1 * t.nb2;
            const u32 *row = a.rmat + r * t.nb2;
            for (u32 b = threadIdx.x; b < t.nb2; b += NT) L.gb[b] = a.fine_base[bin0 + b] + row[b];
            for (u32 cb = 0; cb < g.cnt; cb += TILE) {          // (scatter_chunk opens with a barrier: gb[] is visible)
                Kmer<W> key[KPT];
                load_chunk<W, 2, NT, KPT>(key, bufA, a, g.b1, s_l1b[g.b1], g.begin + cb, min(TILE, g.cnt - cb));
                scatter_chunk<W, 2, true, NT, NoPrefetch, KPT>(key, min(TILE, g.cnt - cb), t, t.nb2, L.sorted, L.binof, L.off, L.lim, L.gb, L.wsum, a, bin0, bufB,
                                              nosp, noclaims, GK_TARGS);
            }
            __syncthreads();
        }
    } else {
        struct Pos { u64 c; u32 b1; u64 begin; u32 cnt; };
        auto geom = [&](u64 c) {
            Pos p{c, 0u, 0ull, 0u};
            if (c < total) {
                p.b1 = chunk_bucket(s_ubase, c, nb1);
                const u64 rel = (c - s_ubase[p.b1]) * TILE;
                p.begin = s_l1f[p.b1] + rel;
                p.cnt = (u32)min((u64)TILE, s_l1n[p.b1] - rel);
            }
            return p;
        };
        auto request = [&](Kmer<W> (&kk)[KPT], const Pos &p) {
            if (p.c < total) load_chunk<W, 2, NT, KPT>(kk, bufA, a, p.b1, s_l1b[p.b1], p.begin, p.cnt);
        };
        Pos cur = geom(first + blockIdx.x);
        auto step = [&](Kmer<W> (&kc)[KPT], Kmer<W> (&kn)[KPT]) {
            const Pos nxt = geom(cur.c + gridDim.x);
            scatter_chunk<W, 2, false, NT>(kc, cur.cnt, t, t.nb2, L.sorted, L.binof, L.off, L.lim, L.gb, L.wsum, a, (u64)cur.b1 * t.nb2, bufB,
                                           nosp, noclaims, GK_TARGS, [&]() { request(kn, nxt); });
            cur = nxt;
        };
        Kmer<W> keyA[KPT], keyB[KPT];
        request(keyA, cur);
        __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0), once: both halves of the loop then know their keys are in
        while (cur.c < total) {
            step(keyA, keyB);
            if (cur.c >= total) break;
            step(keyB, keyA);
        }
    }
    GK_TFLUSH(8);
}

// P4 of the exact fine level when a chunk of 4096 keys meets MANY fine buckets (nb2 in the thousands: 2-3 keys per bin and
// chunk).  Sorting the chunk by bin in LDS then buys no coalescing — a bin's run is shorter than a line — and costs five
// loops over all bins plus six barriers per chunk, which is what bounds scatter_chunk there (52 ms of C3's 123 ms count
// at nb2 = 1525).  The range matrix already gives every (range, bin) its exact destination, so a range keeps ONE running
// cursor per bin in LDS and every key goes straight out: one LDS atomic and one 8/16-byte store per key, one loop over the
// bins and two barriers per RANGE (64 Ki keys), nothing per chunk.  Order inside a segment's run is arbitrary (as it
// always was: P5's result does not depend on it).
template <int W>
__global__ __launch_bounds__(PBLOCK) void k_part_scatter2_direct(const u64 *__restrict__ bufA, Table<W> t, PartArrays a, u64 max_ranges,
                                                                 u64 *__restrict__ bufB) {
    extern __shared__ u32 lds_cur[];                 // [nb2] next free position of every bin, relative to the L1 bucket's first key in bufB
    __shared__ unsigned long long s_rbase[MAXB1 + 1], s_l1n[MAXB1];
    const u32 nb1 = 1u << t.lnb1;
    for (u32 b = threadIdx.x; b <= nb1; b += PBLOCK) s_rbase[b] = a.rbase[b];
    for (u32 b = threadIdx.x; b < nb1; b += PBLOCK) s_l1n[b] = l1_count(a, b);
    __syncthreads();
    const u64 total = min((u64)s_rbase[nb1], max_ranges);
    for (u64 r = blockIdx.x; r < total; r += gridDim.x) {
        const RangeGeom g = range_geom(s_rbase, s_l1n, r, a.range_chunks, nb1);
        const u64 bin0 = (u64)g.b1 * t.nb2;
        const u64 base = a.fine_base[bin0];          // (= l1_base[b1]: bufB is dense)
        const u32 *row = a.rmat + r * t.nb2;
        __syncthreads();                             // the previous range's keys are out
        for (u32 b = threadIdx.x; b < t.nb2; b += PBLOCK) lds_cur[b] = (u32)(a.fine_base[bin0 + b] - base) + row[b];
        __syncthreads();
        for (u32 cb = 0; cb < g.cnt; cb += TILE2) {
            const u32 cnt = min((u32)TILE2, g.cnt - cb);
            Kmer<W> key[KEYS_PER_THREAD];
#pragma unroll
            for (int j = 0; j < KEYS_PER_THREAD; j++) {
                const u32 i = threadIdx.x + j * PBLOCK;
                key[j] = load_key<W>(bufA, l1_key_index(a, g.b1, g.begin + cb + (i < cnt ? i : cnt - 1)));
            }
#pragma unroll
            for (int j = 0; j < KEYS_PER_THREAD; j++)
                if (threadIdx.x + j * PBLOCK < cnt) {
                    const u32 pos = atomicAdd(&lds_cur[seg_fine(t, slot_hash(key[j]))], 1u);
                    store_key<W>(bufB, base + pos, key[j]);
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// P5: one workgroup owns one segment
// ---------------------------------------------------------------------------------------------
struct LdsCas {
    __device__ __forceinline__ u64 operator()(u64 *p, u64 e, u64 v) const {
        return atomicCAS(reinterpret_cast<unsigned long long *>(p), (unsigned long long)e, (unsigned long long)v);
    }
    __device__ __forceinline__ u32 operator()(u32 *p, u32 e, u32 v) const { return atomicCAS(p, e, v); }
};
struct LdsAdd {
    __device__ __forceinline__ void operator()(u32 *p, u32 v) const { atomicAdd(p, v); }
};

// Insert into a segment held in LDS that is KNOWN to keep a free slot (built from empty with fewer
// keys than slots): the probe needs no bound and no look-before-CAS.  The general seg_add spends
// ~35 instructions per probe step on nested divergent branches, and k_seg_insert is bound by
// instruction issue (PMC: ~180 VALU + ~210 SALU per 64 keys, LDS array 40 % busy), not by HBM; this
// form is one ds_cmpst, two compares and the step.  Same slot protocol as seg_add.
__device__ __forceinline__ u32 lds_add_unbounded(Slot<1> *seg, u32 pos, Kmer<1> key) {
    constexpr u32 smask = (1u << SegBits<1>::value) - 1u;
    u32 i = pos;
    u64 old;
    for (;;) {
        old = atomicCAS(reinterpret_cast<unsigned long long *>(&seg[i].w0), (unsigned long long)KEY_EMPTY, (unsigned long long)key.lo);
        if (old == KEY_EMPTY || old == key.lo) break;
        i = (i + 1) & smask;
    }
    if (old == KEY_EMPTY) return 1u;
    atomicAdd(&seg[i].extra, 1u);
    return 0u;
}
__device__ __forceinline__ u32 lds_add_unbounded(Slot<2> *seg, u32 pos, Kmer<2> key) {
    constexpr u32 smask = (1u << SegBits<2>::value) - 1u;
    const Stored<2> k = to_stored(key);
    u32 i = pos;
    for (;;) {
        const u64 c0 = atomicCAS(reinterpret_cast<unsigned long long *>(&seg[i].w0), (unsigned long long)KEY_EMPTY, (unsigned long long)k.w0);
        if (c0 == KEY_EMPTY || c0 == k.w0) {
            const u64 c1 = atomicCAS(reinterpret_cast<unsigned long long *>(&seg[i].w1), (unsigned long long)KEY_EMPTY, (unsigned long long)k.w1);
            if (c1 == KEY_EMPTY) return 1u;
            if (c1 == k.w1) { atomicAdd(&seg[i].extra, 1u); return 0u; }
        }
        i = (i + 1) & smask;
    }
}
// The bounded form for segments that receive more keys than they have free slots — necessarily REPEATS of far fewer
// k-mers (sequencing coverage): look first (a plain LDS read of a hot slot is a broadcast; a failed ds_cmpst on it
// would serialise the lanes twice, once for the compare-and-swap and once for the add), CAS only on an empty slot.
// Key words are write-once during an insert phase, so what a plain read shows is either final or EMPTY.
// 1 = claimed a new slot, 0 = the key was there, -1 = the segment is full.
__device__ __forceinline__ int lds_add_look(Slot<1> *seg, u32 pos, Kmer<1> key) {
    constexpr u32 smask = (1u << SegBits<1>::value) - 1u;
    u32 i = pos;
    for (u32 n = 0; n <= smask; ++n) {
        u64 cur = seg[i].w0;
        if (cur == KEY_EMPTY) {
            cur = atomicCAS(reinterpret_cast<unsigned long long *>(&seg[i].w0), (unsigned long long)KEY_EMPTY, (unsigned long long)key.lo);
            if (cur == KEY_EMPTY) return 1;
        }
        if (cur == key.lo) { atomicAdd(&seg[i].extra, 1u); return 0; }
        i = (i + 1) & smask;
    }
    return -1;
}
__device__ __forceinline__ int lds_add_look(Slot<2> *seg, u32 pos, Kmer<2> key) {
    constexpr u32 smask = (1u << SegBits<2>::value) - 1u;
    const Stored<2> k = to_stored(key);
    u32 i = pos;
    for (u32 n = 0; n <= smask; ++n) {
        u64 c0 = seg[i].w0;
        if (c0 == KEY_EMPTY) {
            c0 = atomicCAS(reinterpret_cast<unsigned long long *>(&seg[i].w0), (unsigned long long)KEY_EMPTY, (unsigned long long)k.w0);
            if (c0 == KEY_EMPTY) c0 = k.w0;
        }
        if (c0 == k.w0) {
            u64 c1 = seg[i].w1;
            if (c1 == KEY_EMPTY) {
                c1 = atomicCAS(reinterpret_cast<unsigned long long *>(&seg[i].w1), (unsigned long long)KEY_EMPTY, (unsigned long long)k.w1);
                if (c1 == KEY_EMPTY) return 1;
            }
            if (c1 == k.w1) { atomicAdd(&seg[i].extra, 1u); return 0; }
        }
        i = (i + 1) & smask;
    }
    return -1;
}

// the 12-byte count slot (8-byte keys as two 31-bit halves, gk_device.h): the two-word protocol with 32-bit ds_cmpst
__device__ __forceinline__ u32 lds_add_unbounded(CSlot *seg, u32 pos, Kmer<1> key) {
    constexpr u32 smask = (1u << SegBits<1>::value) - 1u;
    const u32 k0 = c_w0(key), k1 = c_w1(key);
    u32 i = pos;
    for (;;) {
        const u32 c0 = atomicCAS(&seg[i].w0, KEY_EMPTY32, k0);
        if (c0 == KEY_EMPTY32 || c0 == k0) {
            const u32 c1 = atomicCAS(&seg[i].w1, KEY_EMPTY32, k1);
            if (c1 == KEY_EMPTY32) return 1u;
            if (c1 == k1) { atomicAdd(&seg[i].extra, 1u); return 0u; }
        }
        i = (i + 1) & smask;
    }
}
__device__ __forceinline__ int lds_add_look(CSlot *seg, u32 pos, Kmer<1> key) {
    constexpr u32 smask = (1u << SegBits<1>::value) - 1u;
    const u32 k0 = c_w0(key), k1 = c_w1(key);
    u32 i = pos;
    for (u32 n = 0; n <= smask; ++n) {
        u32 c0 = seg[i].w0;
        if (c0 == KEY_EMPTY32) {
            c0 = atomicCAS(&seg[i].w0, KEY_EMPTY32, k0);
            if (c0 == KEY_EMPTY32) c0 = k0;
        }
        if (c0 == k0) {
            u32 c1 = seg[i].w1;
            if (c1 == KEY_EMPTY32) {
                c1 = atomicCAS(&seg[i].w1, KEY_EMPTY32, k1);
                if (c1 == KEY_EMPTY32) return 1;
            }
            if (c1 == k1) { atomicAdd(&seg[i].extra, 1u); return 0; }
        }
        i = (i + 1) & smask;
    }
    return -1;
}

#ifndef GK_P5_GRID_PER_CU
#define GK_P5_GRID_PER_CU 24          // workgroups of k_seg_insert's persistent grid per CU (four are resident)
#endif
#ifndef GK_SBLOCK
#define GK_SBLOCK (GK_SEG_BITS1 <= 10 ? 256 : 512)
#endif
static constexpr int SBLOCK = GK_SBLOCK;               // threads per segment workgroup
// ST = the table's slot type in HBM.  The segment is built in LDS in the type that probes fastest there — LT: for a count table
// of 8-byte keys (ST = 12-byte CSlot) that is the 16-byte Slot<1> with its ONE 64-bit ds_cmpst per insert, not CSlot's two
// 32-bit ones (measured: the two-CAS form in LDS cost P5 0.92 ms against 0.82 — more than the quarter fewer bytes gave back;
// profiles/r03/bench_n1_run08_cslot_two_cas_LOSES.json) — and the HBM image is converted on the way in and out, four slots
// (three 16-byte vectors) per thread.
#ifndef GK_ABL_P5
#define GK_ABL_P5 0                     // timing-only ablations of k_seg_insert (variant builds; never in the product library)
#endif
#ifndef GK_P5_MIN_WAVES
#define GK_P5_MIN_WAVES 8
#endif
template <class ST> struct LdsSlotOf { typedef ST type; };
template <> struct LdsSlotOf<CSlot> { typedef Slot<1> type; };
// Both directions keep every HBM access a run of whole 16-byte vectors per wave and need few registers (the kernel lives at 64
// VGPRs, 8 waves per SIMD: the first cuts of this conversion — every thread moving "its" four slots as three vectors — spilled
// 36 bytes per lane, and scratch traffic is HBM traffic: P5 1.11-1.14 ms with 1.5 GB fetched instead of 0.98,
// profiles/r03/bench_n1_run09_*, pmc_pipeline_r3v4.json).
//   out: vector v of the packed image holds words 4v .. 4v+3 = fields of slots s0 = v + v/3 and s0 + 1: read those two LDS
//        slots, pick by v % 3, store — no staging, no barrier;
//   in:  the 24-KiB image is staged at the END of the 32-KiB area (bytes 8192 ..), then expanded slot by slot in four rounds of
//        512 slots, ascending: expanded slot s lands on bytes 16s .., which hold raw slots below 4s/3 - 681 only — consumed in an
//        earlier round or, within the round, before the barrier that separates its reads from its writes.
// field f of the 12-byte form of a 16-byte slot.  No branches: a key is below 2^62, EMPTY / TOMB are ~0 and ~0 - 1, so bit 63
// tells them apart — their low word IS the 32-bit sentinel and (w0 >> 31) truncates to ~0
__device__ __forceinline__ u32 cfield(const Slot<1> &x, u32 f) {
    if (f == 0u) return (u32)x.w0 & ((u32)((i32)(x.w0 >> 32) >> 31) | 0x7fffffffu);
    if (f == 1u) return (u32)(x.w0 >> 31);
    return x.extra;
}
__device__ __forceinline__ uint4 cslot_vec_from_lds(const Slot<1> *seg, u32 v) {
    const u32 q = v / 3u, r = v - 3u * q, s0 = v + q;
    const Slot<1> A = seg[s0], B = seg[s0 + 1u];            // (s0 + 1 <= S - 1: the image's last vector has r == 2)
    const u32 a0 = cfield(A, 0), a1 = cfield(A, 1), a2 = A.extra, b0 = cfield(B, 0), b1 = cfield(B, 1), b2 = B.extra;
    return r == 0u ? make_uint4(a0, a1, a2, b0) : r == 1u ? make_uint4(a1, a2, b0, b1) : make_uint4(a2, b0, b1, b2);
}
template <u32 S_>
__device__ __forceinline__ void cslots_expand_in_lds(uint4 *lds_raw, u32 *nfree) {
    Slot<1> *seg = reinterpret_cast<Slot<1> *>(lds_raw);
    const u32 *raw = reinterpret_cast<const u32 *>(lds_raw) + (S_ * 4u) / 4u;       // byte S_*4 = 8192 for S_ = 2048
#pragma unroll 1
    for (u32 s = threadIdx.x; s < S_; s += SBLOCK) {
        const u32 h0 = raw[3u * s], h1 = raw[3u * s + 1u], ex = raw[3u * s + 2u];
        __syncthreads();
        const bool special = h0 >= KEY_TOMB32;                  // EMPTY or TOMB: the same sentinel, widened
        seg[s] = Slot<1>{special ? (h0 == KEY_EMPTY32 ? KEY_EMPTY : KEY_TOMB) : ((u64)h0 | ((u64)h1 << 31)), ex, 0u};
        *nfree += h0 == KEY_EMPTY32 ? 1u : 0u;
    }
}
template <int W, class ST>
__global__ __launch_bounds__(SBLOCK, GK_P5_MIN_WAVES) void k_seg_insert(Table<W, ST> t, const u64 *__restrict__ keys, PartArrays a, int from_empty, Counters *ctr,
                                                          u64 seg_lo, u64 seg_hi /* this launch's segments: [seg_lo, seg_hi) */) {
    extern __shared__ uint4 lds_raw[];
    typedef typename LdsSlotOf<ST>::type LT;
    constexpr bool CONV = !std::is_same<ST, LT>::value;             // the HBM image is converted, not copied
    constexpr u32 S = 1u << SegBits<W>::value;
    constexpr u32 NVEC = S * sizeof(LT) / 16;                       // vectors of the LDS image
    constexpr u32 GVEC = S * sizeof(ST) / 16;                       // ... of the HBM image
    constexpr int KPT = (int)(S / SBLOCK);                          // keys preloaded per thread
    constexpr u32 KBLK = (u32)SBLOCK * KPT;                         // keys per register block (= S)
    LT *seg = reinterpret_cast<LT *>(lds_raw);
    const ST *gtype = nullptr;                                      // (only names the HBM slot type for the clear pattern)
    u32 *flags = reinterpret_cast<u32 *>(lds_raw + NVEC);         // [0] claims, [1] overflow, [2] free slots found while loading
    const u64 nseg = min(seg_hi, t.nseg());
    u32 wg_claims = 0;      // thread 0 only: ONE global atomic per workgroup at the end (a same-address
                            // atomic per segment caps the kernel at ~88 segments/us chip-wide)
    // Software pipeline over this workgroup's segments.  One segment is a chain of dependent round
    // trips (key range -> keys -> LDS inserts -> write-back) and only four workgroups fit a CU, so
    // the next segment's keys are requested before the current segment's inserts and the key range
    // of the one after that before that: both latencies hide behind the insert phase.
    auto range_of = [&](u64 s2, u64 &b, u32 &n) {
        const u64 sc = s2 < nseg ? s2 : nseg - 1;                     // clamped, branch-free loads
        const u64 cnt = fine_count_u(a, sc);
        b = fine_begin_u(a, sc);
        n = s2 < nseg ? (u32)min(cnt, (u64)0xffffffffu) : 0u;
    };
    // block `blk` of a segment's keys into registers (indices clamped: branch-free, one round trip)
    auto request_keys = [&](Kmer<W> (&kk)[KPT], u64 b, u32 n, u32 blk) {
        const u32 first = blk * KBLK;
        const u32 nk = n > first ? min(n - first, KBLK) : 0u;
#pragma unroll
        for (int j = 0; j < KPT; j++) {
            const u32 i = threadIdx.x + j * SBLOCK;
            kk[j] = load_key<W>(keys, nk ? b + first + (i < nk ? i : nk - 1) : 0);
        }
    };
    // an EMPTY segment image to HBM.  (The 12-byte pattern depends on the vector's index mod 3: left alone, the compiler keeps a
    // thread's three pattern vectors in 12 VGPRs for the whole kernel — and spills them; the opaque move makes it compute them here.)
    auto clear_gseg = [&](uint4 *gseg) {
#pragma unroll
        for (u32 i = threadIdx.x; i < GVEC; i += SBLOCK) {
            u32 ii = i;
            if constexpr (CONV) asm volatile("" : "+v"(ii));
            gseg[i] = empty_vec_of(gtype, ii);
        }
    };
    u64 kb, kbn; u32 cnt, cntn;
    // one segment; `cur` holds its first key block (requested one step earlier), `nxt` receives the next segment's.
    // The two register sets swap roles from step to step (a copy would have to wait for the loads).
    auto step = [&](u64 s, Kmer<W> (&cur)[KPT], Kmer<W> (&nxt)[KPT]) {
        u64 kbnn; u32 cntnn;
        range_of(s + 2ull * gridDim.x, kbnn, cntnn);                  // two ahead: its key range
        request_keys(nxt, kbn, cntn, 0u);                             // one ahead: its first keys
        uint4 *gseg = reinterpret_cast<uint4 *>(t.slots + (s << SegBits<W>::value));
        if (cnt == 0) {
            if (from_empty) clear_gseg(gseg);       // materialise the pending clear of a segment that gets no key
        } else {
            __syncthreads();
            if (threadIdx.x < 3) flags[threadIdx.x] = 0;
            if (from_empty) {
                u32 z = 0;
                if constexpr (CONV) asm volatile("" : "+v"(z));        // (keeps the pattern out of long-lived registers: see clear_gseg)
#pragma unroll
                for (u32 i = threadIdx.x; i < NVEC; i += SBLOCK) {
                    uint4 e = empty_vec_of(seg, i);
                    e.x ^= z; e.y ^= z; e.z ^= z; e.w ^= z;
                    lds_raw[i] = e;
                }
            } else {
                u32 nfree = 0;                                  // a slot is free iff its first key word is EMPTY
                if constexpr (CONV) {
#pragma unroll
                    for (u32 i = threadIdx.x; i < GVEC; i += SBLOCK) lds_raw[(NVEC - GVEC) + i] = gseg[i];
                    __syncthreads();
                    cslots_expand_in_lds<S>(lds_raw, &nfree);
                } else {
#pragma unroll
                    for (u32 i = threadIdx.x; i < NVEC; i += SBLOCK) {
                        const uint4 v = gseg[i];
                        lds_raw[i] = v;
                        nfree += empty_w0_in_vec_of(seg, i, v);
                    }
                }
                __syncthreads();                                // flags[2] = 0 is visible
                for (int d = 32; d; d >>= 1) nfree += __shfl_down(nfree, d);
                if ((threadIdx.x & 63) == 0 && nfree) atomicAdd(&flags[2], nfree);
            }
            __syncthreads();
            u32 claims = 0;
            bool overflow = false;
            // fewer keys than free slots: cannot fill up -> unbounded one-CAS probe; more keys than that are repeats
            // (or the table is too small): bounded look-first probe; k = 64 keeps the general tagged form
            const int mode = t.tagged ? 2 : (cnt < (from_empty ? S : flags[2]) ? 0 : 1);
            auto insert_block = [&](Kmer<W> (&kk)[KPT], u32 nk) {
#pragma unroll
                for (int j = 0; j < KPT; j++) {
                    if (threadIdx.x + j * SBLOCK >= nk) continue;
                    const u32 pos = home_pos(t, slot_hash(kk[j]));
                    if (mode == 0) { claims += lds_add_unbounded(seg, pos, kk[j]); continue; }
                    const int r = mode == 1 ? lds_add_look(seg, pos, kk[j]) : seg_add(seg, pos, kk[j], 1u, LdsCas(), LdsAdd(), t.tagged);
                    if (r < 0) overflow = true; else claims += (u32)r;
                }
            };
#if GK_ABL_P5 == 1                      // timing-only: no inserts (the segment goes out EMPTY)
            if (cnt == 0xffffffffu)
#endif
            insert_block(cur, min(cnt, KBLK));
            if (cnt > KBLK) {                                   // heavy segments (repeats): further blocks, each requested a block ahead
                Kmer<W> ta[KPT], tb[KPT];
                const u32 nblk = (cnt + KBLK - 1) / KBLK;
                request_keys(ta, kb, cnt, 1u);
                for (u32 blk = 1; blk < nblk; blk += 2) {
                    request_keys(tb, kb, cnt, blk + 1);         // (past the end: nk = 0, a harmless load of keys[0])
                    insert_block(ta, min(cnt - blk * KBLK, KBLK));
                    if (blk + 1 >= nblk) break;
                    request_keys(ta, kb, cnt, blk + 2);
                    insert_block(tb, min(cnt - (blk + 1) * KBLK, KBLK));
                }
            }
            for (int d = 32; d; d >>= 1) claims += __shfl_down(claims, d);
            if ((threadIdx.x & 63) == 0 && claims) atomicAdd(&flags[0], claims);
            if (overflow) flags[1] = 1;
            __syncthreads();
            if (flags[1]) {
                // segment full: leave the HBM copy as it was (or EMPTY) and hand the bucket to the host,
                // which grows the table and replays these keys through the direct path
                if (from_empty) clear_gseg(gseg);
                if (threadIdx.x == 0) a.failed[atomicAdd(a.n_failed, 1u)] = (u32)s;
            } else {
#if GK_ABL_P5 == 2                      // timing-only: no write-out
                if (cnt == 0xffffffffu)
#endif
                if constexpr (CONV) {
#if GK_ABL_P5 == 3                      // timing-only: the write-out without the conversion (the LDS image's first 24 KiB as they are)
#pragma unroll
                    for (u32 i = threadIdx.x; i < GVEC; i += SBLOCK) gseg[i] = lds_raw[i];
#elif GK_ABL_P5 == 4                    // A/B: every thread packs ITS four slots (no selects) and stores three vectors at a 48-byte stride
                    for (u32 g4 = threadIdx.x; g4 < S / 4; g4 += SBLOCK) {
                        const Slot<1> a0 = seg[4 * g4], a1 = seg[4 * g4 + 1], a2 = seg[4 * g4 + 2], a3 = seg[4 * g4 + 3];
                        gseg[3 * g4] = make_uint4(cfield(a0, 0), cfield(a0, 1), a0.extra, cfield(a1, 0));
                        gseg[3 * g4 + 1] = make_uint4(cfield(a1, 1), a1.extra, cfield(a2, 0), cfield(a2, 1));
                        gseg[3 * g4 + 2] = make_uint4(a2.extra, cfield(a3, 0), cfield(a3, 1), a3.extra);
                    }
#else
#pragma unroll 1
                    for (u32 i = threadIdx.x; i < GVEC; i += SBLOCK) gseg[i] = cslot_vec_from_lds(seg, i);
#endif
                } else {
#pragma unroll
                    for (u32 i = threadIdx.x; i < NVEC; i += SBLOCK) gseg[i] = lds_raw[i];
                }
                if (threadIdx.x == 0) wg_claims += flags[0];
            }
        }
        kb = kbn; cnt = cntn; kbn = kbnn; cntn = cntnn;
    };
    Kmer<W> keyA[KPT], keyB[KPT];
    u64 s = seg_lo + blockIdx.x;
    range_of(s, kb, cnt);
    range_of(s + gridDim.x, kbn, cntn);
    request_keys(keyA, kb, cnt, 0u);
    while (s < nseg) {
        step(s, keyA, keyB);
        s += gridDim.x;
        if (s >= nseg) break;
        step(s, keyB, keyA);
        s += gridDim.x;
    }
    if (threadIdx.x == 0 && wg_claims) atomicAdd(&ctr->size, (unsigned long long)wg_claims);
}

// ---------------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------------
namespace gk {

struct PartScratch {
    void *blob = nullptr;        // the small per-batch arrays of the L1 level
    size_t blob_bytes = 0;
    void *fblob = nullptr;       // the per-segment arrays of the fine level (sized after the table's final geometry is known)
    size_t fblob_bytes = 0;
    u32 *rmat = nullptr;         // range matrix
    u64 rmat_words = 0;
    u64 *bufA = nullptr, *bufB = nullptr, *spill = nullptr;
    u64 bufA_keys = 0, bufB_keys = 0, spill_keys = 0;     // capacities in keys
    int W = 1;
    bool lds_attr_set = false;
    unsigned long long *piece_tables = nullptr;      // inside blob: per piece [from MAXB1][to MAXB1][cbase MAXB1 + 1] (pipelined batches)
};
static constexpr int MAX_PIECES = 8;
#ifndef GK_P24_PIECES_DEFAULT
#define GK_P24_PIECES_DEFAULT 1          // pieces of a pipelined batch ("p24_pieces"; 0/1 = one piece: the default, see part_run)
#endif
static constexpr size_t PIECE_WORDS = MAXB1 + MAXB1 + (MAXB1 + 1) + 7;
// LDS the P4 kernels take for their chunk / range table and L1 extents, in front of the sort buffers
static size_t part_tables_bytes(u32 nb1) { return ((size_t)4 * nb1 + 1) * 8; }
static constexpr size_t LDS_BYTES_PER_CU = 160u << 10;

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int grow_buf(gk_ctx *ctx, u64 **buf, u64 *have, u64 want, int W) {
    if (*have >= want) return GK_OK;
    if (*buf) GK_HIP(ctx, hipFree(*buf));
    *buf = nullptr; *have = 0;
    hipError_t e = hipMalloc((void **)buf, std::max<u64>(want, 1) * 8 * W);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(ctx, GK_E_CAPACITY, "partitioned insert: cannot allocate " + std::to_string(want * 8 * W) + " bytes of key scratch: " + hipGetErrorString(e));
    }
    *have = want;
    return GK_OK;
}
static int grow_raw(gk_ctx *ctx, void **buf, size_t *have, size_t want) {
    if (*have >= want) return GK_OK;
    if (*buf) GK_HIP(ctx, hipFree(*buf));
    *buf = nullptr; *have = 0;
    GK_HIP(ctx, hipMalloc(buf, std::max<size_t>(want, 256)));
    *have = want;
    return GK_OK;
}

// L1 level: small arrays, the key buffer bufA and the spill list
static int part_prepare_l1(gk_map *m, PartScratch *ps, u64 nkeys, bool op1, PartArrays *arr) {
    gk_ctx *ctx = m->ctx;
    const u64 nb1 = 1ull << m->lnb1;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    // (nspill, overflow, n_failed are read back together at the end of a batch: contiguous, one copy)
    const size_t o_hist1 = take(MAXB1 * 8), o_l1 = take((MAXB1 + 1) * 8), o_cur1 = take(MAXB1 * 8), o_cb = take((MAXB1 + 1) * 8), o_rb = take((MAXB1 + 1) * 8),
                 o_nsp = take(16), o_ovf = o_nsp + 8, o_nf = o_nsp + 12, o_pieces = take((size_t)MAX_PIECES * PIECE_WORDS * 8);
    if (int rc = grow_raw(ctx, &ps->blob, &ps->blob_bytes, off)) return rc;
    ps->piece_tables = (unsigned long long *)((char *)ps->blob + o_pieces);
    GK_HIP(ctx, hipMemsetAsync(ps->blob, 0, off, ctx->stream));
    char *b = (char *)ps->blob;
    arr->hist1 = (unsigned long long *)(b + o_hist1); arr->l1_base = (unsigned long long *)(b + o_l1);
    arr->cursor1 = (unsigned long long *)(b + o_cur1); arr->cbase = (unsigned long long *)(b + o_cb);
    arr->rbase = (unsigned long long *)(b + o_rb);
    arr->n_failed = (u32 *)(b + o_nf); arr->nspill = (unsigned long long *)(b + o_nsp); arr->overflow = (u32 *)(b + o_ovf);
    arr->hist2 = nullptr; arr->fine_base = nullptr; arr->cursor2 = nullptr; arr->failed = nullptr; arr->rmat = nullptr;
    if (ps->W != m->W) {     // key width changed: drop the buffers
        for (u64 **bp : {&ps->bufA, &ps->bufB, &ps->spill}) { if (*bp) GK_HIP(ctx, hipFree(*bp)); *bp = nullptr; }
        ps->bufA_keys = ps->bufB_keys = ps->spill_keys = 0;
        ps->W = m->W;
    }
    arr->op1 = op1 ? 1 : 0;
    arr->op2 = 0;
    arr->noncanon = nullptr;
    arr->k = m->k;
    arr->cap1 = arr->cap2 = 0; arr->spill_cap = 0; arr->stripe_nb1 = 0;
    arr->l1_from = arr->l1_to = nullptr;
    // a range = up to MAX_RANGE_CHUNKS chunks, fewer when the batch is small (enough ranges to fill the chip)
    const u64 nchunks = nkeys / TILE2 + 1;
    arr->range_chunks = (u32)std::min<u64>(MAX_RANGE_CHUNKS, std::max<u64>(1, nchunks / ((u64)ctx->cu_count * 6)));
    arr->chunk_keys = TILE2;
    u64 wantA = nkeys, wantS = 0;
    if (op1) {
        // Bucket sizes of hashed keys concentrate: binomial for distinct keys (mean + 8 sigma never overflows); with
        // repeats (coverage c: every k-mer c times) the spread is that of mean/c heavy items — a bucket holds thousands
        // of k-mers, so 1/8 of the mean on top covers multiplicities into the thousands.  What still does not fit (a single
        // k-mer repeated millions of times) goes to the spill list, which the direct path absorbs after the segments are built.
        const double m1 = (double)nkeys / (double)nb1;
        arr->cap1 = (u64)(m1 * 1.125 + 8.0 * std::sqrt(m1) + 1024.0);
        arr->spill_cap = nkeys / 16 + 65536;
        // interleave the L1 regions (l1_slot) once the buffer is big enough for P2 to thrash the TLB (measured: fine at
        // 0.96 GB, 73 % translation misses at 3.8 GB)
        const bool striped = nb1 * arr->cap1 * 8ull * m->W >= (3ull << 29);
        arr->stripe_nb1 = striped ? (u32)nb1 : 0u;
        wantA = striped ? (arr->cap1 + L1_BLK - 1) / L1_BLK * L1_BLK * nb1 : nb1 * arr->cap1;
        wantS = arr->spill_cap;
    }
    if (int rc = grow_buf(ctx, &ps->bufA, &ps->bufA_keys, wantA, m->W)) return rc;
    if (int rc = grow_buf(ctx, &ps->spill, &ps->spill_keys, wantS, m->W)) return rc;
    arr->spill = ps->spill;
    return GK_OK;
}

// fine level: per-segment arrays for the table's FINAL geometry, bufB, and the range matrix (exact) / regions (op2)
static int part_prepare_fine(gk_map *m, PartScratch *ps, u64 nkeys, bool op2, PartArrays *arr) {
    gk_ctx *ctx = m->ctx;
    const u64 nseg = (u64)m->nb2 << m->lnb1;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_hist2 = take(nseg * 4), o_fine = take((nseg + 1) * 8), o_cur2 = take(nseg * 4), o_failed = take(nseg * 4);
    if (int rc = grow_raw(ctx, &ps->fblob, &ps->fblob_bytes, off)) return rc;
    // hist2 is fully written by k_part_scan2, fine_base by k_part_prefix2; only the cursors need zeroes
    char *b = (char *)ps->fblob;
    arr->hist2 = (u32 *)(b + o_hist2); arr->fine_base = (unsigned long long *)(b + o_fine);
    arr->cursor2 = (u32 *)(b + o_cur2); arr->failed = (u32 *)(b + o_failed);
    arr->op2 = op2 ? 1 : 0;
    u64 wantB = nkeys;
    if (op2) {
        GK_HIP(ctx, hipMemsetAsync(arr->cursor2, 0, nseg * 4, ctx->stream));
        const double m2 = (double)nkeys / (double)nseg;
        arr->cap2 = (u64)(m2 + 8.0 * std::sqrt(m2) + 64.0);
        wantB = nseg * arr->cap2;
    } else {
        const u64 max_ranges = (nkeys / TILE2 + MAXB1 + 1) / arr->range_chunks + MAXB1 + 1;
        const u64 words = max_ranges * m->nb2;
        if (ps->rmat_words < words) {
            if (ps->rmat) GK_HIP(ctx, hipFree(ps->rmat));
            ps->rmat = nullptr; ps->rmat_words = 0;
            GK_HIP(ctx, hipMalloc((void **)&ps->rmat, words * 4));
            ps->rmat_words = words;
        }
        arr->rmat = ps->rmat;
    }
    return grow_buf(ctx, &ps->bufB, &ps->bufB_keys, wantB, m->W);
}

void part_scratch_free(gk_ctx *ctx, PartScratch *ps) {
    if (!ps) return;
    for (void *p : {ps->blob, ps->fblob, (void *)ps->rmat, (void *)ps->bufA, (void *)ps->bufB, (void *)ps->spill})
        if (p) (void)hipFree(p);
    delete ps;
}

// returns GK_OK, an error (< 0), or PART_RETRY_DIRECT: the over-provisioned regions and the spill
// list overflowed (extreme skew); nothing but scratch was touched, the caller takes the direct path
template <int W>
static int part_run(gk_map *m, PartScratch *ps, const ReadSrc &src, const u64 *d_keys, u64 nkeys_in, u64 nkeys_bound, bool from_empty,
                    const PartPlan &plan) {
    const unsigned long long occ_before = m->occ_cached;     // the windows counted by earlier batches of this call (every batch ends with map_sync_counters)
    gk_ctx *ctx = m->ctx;
    Table<W> t{reinterpret_cast<Slot<W> *>(m->slots), m->nb2, m->lnb1, m->k == 64 ? 1u : 0u, 0u};
    PartArrays a;
    const uint8_t *d_rec = src.rec;
    const u32 *d_off = src.off;
    // fixed-stride records and key arrays: their key count is known exactly, so the L1 regions can be sized
    // up front and the histogram pass P1 dropped; ragged streams keep the exact pipeline
    const int max_windows = d_rec && !d_off ? std::max(0, src.max_len - m->k + 1) : 0;
    const bool op1 = !d_off && !ctx->hook_part_exact && (d_keys || (max_windows > 0 && max_windows <= OP_CAP / W));
    if (int rc = part_prepare_l1(m, ps, nkeys_bound, op1, &a)) return rc;
    const u32 nb1 = 1u << m->lnb1;
    const int cu8 = ctx->cu_count * 8;
    // P5 holds one segment in LDS in the table's own slot type: 12-byte count slots (8-byte keys in a count table), else Slot<W>
    const bool cslots = W == 1 && m->layout == LAYOUT_COUNT;
    const size_t lds = ((size_t)1 << SegBits<W>::value) * sizeof(Slot<W>) + 16;        // (the LDS image is Slot<W> for every layout)
    if (!ps->lds_attr_set) {
        const int narrow_max = (int)std::min(ScatterLds<W>::bytes(MAX_NB2) + part_tables_bytes(MAXB1), LDS_BYTES_PER_CU);
        GK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_part_scatter2<W, false, PBLOCK>), hipFuncAttributeMaxDynamicSharedMemorySize, narrow_max));
        GK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_part_scatter2<W, true, PBLOCK>), hipFuncAttributeMaxDynamicSharedMemorySize, narrow_max));
        if constexpr (W == 2) {
            GK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_part_scatter2<2, false, 1024, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES_PER_CU));
            GK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_part_scatter2<2, true, 1024, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES_PER_CU));
        }
        if constexpr (W == 1) {
            const int wide_max = (int)std::min(ScatterLds<1, 1024>::bytes(MAX_NB2) + part_tables_bytes(MAXB1), LDS_BYTES_PER_CU);
            GK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_part_scatter2<1, true, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, wide_max));
            GK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_part_scatter2<1, false, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, wide_max));
            GK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_part_scatter2<1, true, 1024, 12>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES_PER_CU));
            GK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_part_scatter2<1, false, 1024, 12>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES_PER_CU));
        }
        GK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_part_scatter1_keys<W>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)ScatterLds<W>::bytes(MAXB1)));
        GK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_seg_insert<W, Slot<W>>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)(((size_t)1 << SegBits<W>::value) * sizeof(Slot<W>) + 16)));
        if constexpr (W == 1)
            GK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_seg_insert<1, CSlot>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)(((size_t)1 << SegBits<1>::value) * sizeof(Slot<1>) + 16)));
        ps->lds_attr_set = true;
    }
    if (plan.estimate) { if (int rc = map_ensure_sample(m)) return rc; }
    if (m->d_sample) m->sample_dirty = true;
    if (plan.check_canon && d_keys) a.noncanon = &m->d_ctr->noncanon;
    // (the sample keeps learning whenever it exists, also in batches whose estimate nobody waits for: a key it has not
    //  seen counts as new later — an overestimate, the safe side)
    const Sampler sp = m->d_sample ? Sampler{m->d_sample, m->sample_mask, &m->d_ctr->sample_claims} : Sampler{nullptr, 0, nullptr};
    // Over-provisioned fine level with 8-byte keys: 8192-key chunks on 1024 threads when asked for (gk_ctx_set_option "p4_wide")
    // (measured at C2, nb2 = 370: P4 0.66 -> 0.62 ms in mode U, 0.64 -> 0.62 in mode G: half the per-bin bookkeeping per key)
    // (the 8192-key forms need the CU's whole LDS: not with more fine buckets than fit beside the tables of a 512/1024-bucket L1 level)
    auto wide_fits = [&]() { return W == 1 && ScatterLds<1, 1024>::bytes(m->nb2) + part_tables_bytes(nb1) <= LDS_BYTES_PER_CU; };
    // 12288-key sorts (12 keys per thread) where they fit the LDS: measured at C2, nb2 = 370: P4 0.58 -> 0.52-0.56 ms in mode U, no
    // change in mode G ("p4_wide" = 1 keeps 8192)
    auto op_chunk_keys = [&]() -> u32 {
        if (W == 2) return 6144u;                        // 16-byte keys: 1024 threads x 6 keys (96 KB of keys)
        const bool xl = ctx->hook_p4_wide != 1 && ScatterLds<1, 1024, 12>::bytes(m->nb2) + part_tables_bytes(nb1) <= LDS_BYTES_PER_CU;
        return (xl ? 3u : 2u) * (u32)TILE2;
    };
    auto wide2_fits = [&]() { return W == 2 && ScatterLds<2, 1024, 6>::bytes(m->nb2) + part_tables_bytes(nb1) <= LDS_BYTES_PER_CU; };
    auto op_wide_now = [&]() {
        // (16-byte keys, measured at C2 scale: k = 55 P4 0.93 -> 0.85 ms, k = 63 0.86 -> 0.73-0.80; "p4_wide" = 0 keeps 4096-key sorts)
        if (W == 2) return wide2_fits() && (ctx->hook_p4_wide >= 0 ? ctx->hook_p4_wide != 0 : m->nb2 >= 256);
        return wide_fits() && (ctx->hook_p4_wide >= 0 ? ctx->hook_p4_wide != 0 : m->nb2 >= 256);
    };
    // Over-provisioned segment regions are only safe to try on a table that is being rebuilt from empty (if they and
    // the spill list overflow, the table is simply cleared again); a table that holds data gets the exact fine level
    // unless the sample says the batch is near-distinct.
    bool fine_exact = !op1 || plan.fine_exact || !from_empty;
    const bool sync_between = plan.estimate || (op1 && !from_empty) || (src.verify_uniform && !from_empty);
    // PIPELINED PIECES (option "p24_pieces", off by default).  When nothing has to come back to the host between the levels
    // (table geometry final, both levels over-provisioned), the batch can be cut into pieces: both levels' regions are
    // append-only, so P4 can take the slice of every L1 region that piece j's scatter filled while piece j+1 is being
    // scattered (second stream).  Measured at C2 (profiles/r02/ab_p24_pieces.txt): device-resident input 2.05 ms in one piece,
    // 2.12 / 2.19 / 2.21 / 2.34 in 2 / 3 / 4 / 8 — the two kernels' workgroups do not fit a CU's LDS together, so they take
    // turns on the CUs and pay the extra launches; host-fed input gains only where the upload is slow enough to leave the
    // GPU idle (28 GB/s: 2.99 -> 2.71 ms; 56 GB/s: 2.45 -> 2.47-2.55, the scatter alone keeps pace with the link).
    const int want_pieces = ctx->hook_p24_pieces >= 0 ? ctx->hook_p24_pieces : GK_P24_PIECES_DEFAULT;
    const bool pipelined = d_rec && op1 && !fine_exact && !sync_between && want_pieces > 1 && ctx->hook_p45_stripes <= 1 && m->nb2 <= MAX_NB2;
    const u64 max_chunks = nkeys_bound / TILE2 + MAXB1 + 1;          // (every L1 bucket may end in a partial chunk)
    if (pipelined) {
        if (int rc = part_prepare_fine(m, ps, nkeys_bound, true, &a)) return rc;
        if (op_wide_now()) a.chunk_keys = op_chunk_keys();
    }
    // P4 of the over-provisioned fine level over L1 buckets [b_lo, b_hi) of `aa` (the whole batch, a stripe, or a piece's slices)
    auto launch_p4_op = [&](const PartArrays &aa, hipStream_t st, u32 b_lo, u32 b_hi, u64 share) {
        const int per_cu = ctx->hook_p4_grid > 0 ? ctx->hook_p4_grid : 4;           // ("p4_grid": workgroups per CU, A/B)
        const int gchunks = (int)std::min<u64>(max_chunks / share + 1, (u64)ctx->cu_count * per_cu);
        if (W == 2 && aa.chunk_keys == 6144u) {
            if constexpr (W == 2) {
                const size_t lds2 = ScatterLds<2, 1024, 6>::bytes(m->nb2) + part_tables_bytes(nb1);
                hipLaunchKernelGGL((k_part_scatter2<2, false, 1024, 6>), dim3(std::min(gchunks, ctx->cu_count * std::min(per_cu, 2))), dim3(1024), lds2, st, ps->bufA, t, aa,
                                   max_chunks, ps->bufB, b_lo, b_hi);
            }
        } else if (aa.chunk_keys >= 2 * TILE2) {
            if constexpr (W == 1) {
                const size_t wide_lds = ScatterLds<1, 1024>::bytes(m->nb2) + part_tables_bytes(nb1);
                const size_t xl_lds = ScatterLds<1, 1024, 12>::bytes(m->nb2) + part_tables_bytes(nb1);
                if (aa.chunk_keys == 3 * TILE2)
                    hipLaunchKernelGGL((k_part_scatter2<1, false, 1024, 12>), dim3(std::min(gchunks, ctx->cu_count * std::min(per_cu, 2))), dim3(1024), xl_lds, st,
                                       ps->bufA, t, aa, max_chunks, ps->bufB, b_lo, b_hi);
                else
                hipLaunchKernelGGL((k_part_scatter2<1, false, 1024>), dim3(std::min(gchunks, ctx->cu_count * std::min(per_cu, 2))), dim3(1024), wide_lds, st, ps->bufA, t, aa,
                                   max_chunks, ps->bufB, b_lo, b_hi);
            }
        } else
            hipLaunchKernelGGL((k_part_scatter2<W, false, PBLOCK>), dim3(gchunks), dim3(PBLOCK), ScatterLds<W>::bytes(m->nb2) + part_tables_bytes(nb1), st, ps->bufA, t, aa,
                               max_chunks, ps->bufB, b_lo, b_hi);
    };
    PartArrays a_last = a;          // pipelined: the last piece's view (its P4 runs on the main stream, after the join)
    bool have_last = false;
    // after piece j's scatter: its slice tables; every piece but the last goes to the second stream at once
    auto piece_done = [&](int j, int npieces) -> int {
        unsigned long long *tab = ps->piece_tables + (size_t)j * PIECE_WORDS;
        const unsigned long long *prev = j ? ps->piece_tables + (size_t)(j - 1) * PIECE_WORDS + MAXB1 : nullptr;
        hipLaunchKernelGGL(k_part_prefix1_piece, dim3(1), dim3(MAXB1), 0, ctx->stream, a, nb1, prev, tab, tab + MAXB1, tab + 2 * MAXB1);
        PartArrays aj = a;
        aj.l1_from = tab; aj.l1_to = tab + MAXB1; aj.cbase = tab + 2 * MAXB1;
        if (j == npieces - 1) { a_last = aj; have_last = true; return GK_OK; }
        hipEvent_t ev = ctx->cev[8 + j % 8];
        GK_HIP(ctx, hipEventRecord(ev, ctx->stream));
        GK_HIP(ctx, hipStreamWaitEvent(ctx->aux_stream, ev, 0));
        launch_p4_op(aj, ctx->aux_stream, 0u, nb1, (u64)npieces);
        GK_HIP(ctx, hipGetLastError());
        return GK_OK;
    };
    // ---- stage A: P1 + prefix + P2 (the L1 level) -----------------------------------------------------------
    GK_HIP(ctx, hipEventRecord(ctx->pev[0], ctx->stream));
    if (d_rec && op1) {
        GK_HIP(ctx, hipEventRecord(ctx->pev[1], ctx->stream));
        // reads per tile: their windows must fit the LDS key buffer and their bytes the LDS tile
        const u64 by_bytes = ((u64)OP_TILE_WORDS * 4 - 96) / src.stride;
        const int rs = (int)std::max<u64>(1, std::min<u64>(by_bytes, (u64)(OP_CAP / W) / (u64)max_windows));
        const u64 ntiles = (src.nreads + rs - 1) / rs;
        const int grid = (int)std::min<u64>(std::max<u64>(ntiles, 1), (u64)ctx->cu_count * GK_OP_WGS_PER_CU);
        const bool p2_wide = ctx->hook_p2_wide > 0;        // 1024 threads per tile: A/B option (gk_ctx_set_option "p2_wide")
        // bucket-ordered write-out ("p2_sorted"): the same 0.60 ms as the window-ordered form for 8-byte keys (the LDS passes cost
        // what the 64-lines-per-instruction stores cost), 6 % faster for 16-byte keys (0.97 -> 0.91 ms at k = 55): default there
        const bool p2_sorted = ctx->hook_p2_sorted >= 0 ? ctx->hook_p2_sorted > 0 : W == 2;
        const int vu = src.verify_uniform ? 1 : 0;
        auto launch_p2 = [&](int g, const uint8_t *recs, u64 nr) {
#define GK_P2(NT, SORTED, NB1)                                                                                                                 \
    hipLaunchKernelGGL((k_op_scatter1_reads<W, NT, SORTED, NB1>), dim3(g), dim3(NT), 0, ctx->stream, recs, nr, src.stride, m->k, src.group, rs, \
                       src.max_len, vu, t, a, sp, m->d_ctr, ps->bufA)
            if (nb1 > 256) {                          // tables beyond 34 GB (or the test hook): the 1024-bucket form
                if (p2_sorted) GK_P2(PBLOCK, true, (int)MAXB1);
                else GK_P2(PBLOCK, false, (int)MAXB1);
            } else if (p2_wide && p2_sorted) GK_P2(1024, true, 256);
            else if (p2_wide) GK_P2(1024, false, 256);
            else if (p2_sorted) GK_P2(PBLOCK, true, 256);
            else GK_P2(PBLOCK, false, 256);
#undef GK_P2
        };
        if (!src.host && !pipelined) {
            launch_p2(grid, d_rec, src.nreads);
        } else if (!src.host) {
            // equal pieces of whole tiles, each big enough to fill the chip a few times over
            // (an explicit "p24_pieces" also cuts small batches: tests)
            const u64 min_piece = (u64)rs * (ctx->hook_p24_pieces > 1 ? 64ull : (u64)ctx->cu_count * GK_OP_WGS_PER_CU * 2);
            const u64 np = std::max<u64>(1, std::min<u64>({(u64)want_pieces, (u64)MAX_PIECES, src.nreads / std::max<u64>(min_piece, 1)}));
            const u64 sub_reads = ((src.nreads + np - 1) / np + rs - 1) / rs * rs;
            const int npieces = (int)((src.nreads + sub_reads - 1) / sub_reads);
            u64 r0 = 0;
            for (int j = 0; j < npieces; j++, r0 += sub_reads) {
                const u64 nr = std::min<u64>(sub_reads, src.nreads - r0);
                const u64 nt = (nr + rs - 1) / rs;
                launch_p2((int)std::min<u64>(std::max<u64>(nt, 1), (u64)ctx->cu_count * GK_OP_WGS_PER_CU), d_rec + (size_t)r0 * src.stride, nr);
                if (int rc = piece_done(j, npieces)) return rc;
            }
        } else if (src.ready) {
            // Host-fed and PREFETCHED: the whole chunk's upload was queued on the copy stream while the previous chunk (or the
            // caller's previous call) was in its fine level; one scatter behind the event.
            GK_HIP(ctx, hipStreamWaitEvent(ctx->stream, src.ready, 0));
            launch_p2(grid, d_rec, src.nreads);
            if (pipelined) { if (int rc = piece_done(0, 1)) return rc; }
        } else {
            // Host-fed: upload in sub-chunks on the copy stream, scatter each as soon as it has landed.  The L1 regions are
            // append-only (cursor1), so P2 can run once per sub-chunk; P4 and P5 then see one batch.  With the caller's buffer
            // pinned (gk_host_alloc / gk_host_register) the upload of sub-chunk j+1 overlaps the scatter of sub-chunk j.
            // The scatter keeps pace with PCIe (0.6 ms against 0.69 ms for 39 MB at 57 GB/s) but slows down beside a running copy,
            // and every piece costs ~15 us of copy set-up.  Up to 8 equal pieces of whole tiles (measured: a small first and
            // last piece with four big ones in between — sixteenths 1,3,4,4,3,1 — is 4 % slower).
            const u64 min_piece = (u64)rs * 1024;
            u64 piece[8];
            int npieces = 0;
            const u64 sub_reads = std::max<u64>(min_piece, ((src.nreads + 7) / 8 + rs - 1) / rs * rs);
            for (u64 done = 0; done < src.nreads && npieces < 8; done += sub_reads) piece[npieces++] = std::min<u64>(sub_reads, src.nreads - done);
            u64 r0 = 0;
            for (int j = 0; j < npieces; r0 += piece[j], j++) {
                const u64 nr = piece[j];
                if (!nr) continue;
                const size_t off = (size_t)r0 * src.stride, bytes = (size_t)nr * src.stride;
                hipEvent_t ev = ctx->cev[j % 16];
                if (j == 0) {       // the copy stream must not overwrite the staging area while an earlier kernel still reads it
                    GK_HIP(ctx, hipEventRecord(ev, ctx->stream));
                    GK_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ev, 0));
                }
                GK_HIP(ctx, hipMemcpyAsync(const_cast<uint8_t *>(d_rec) + off, src.host + off, bytes, hipMemcpyHostToDevice, ctx->copy_stream));
                GK_HIP(ctx, hipEventRecord(ev, ctx->copy_stream));
                GK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ev, 0));
                const u64 nt = (nr + rs - 1) / rs;
                const int gsub = (int)std::min<u64>(std::max<u64>(nt, 1), (u64)ctx->cu_count * GK_OP_WGS_PER_CU);
                launch_p2(gsub, d_rec + off, nr);
                if (pipelined) { if (int rc = piece_done(j, npieces)) return rc; }
            }
        }
    } else if (d_rec) {
        if (int rc = stage_source(ctx, src)) return rc;
        const u64 ntiles = (src.nreads + PTILE_READS - 1) / PTILE_READS;
        const int grid = (int)std::min<u64>(std::max<u64>(ntiles, 1), (u64)ctx->cu_count * 4);
        hipLaunchKernelGGL(k_part_hist1_reads<W>, dim3(grid), dim3(PBLOCK), 0, ctx->stream, d_rec, src.nreads, d_off, src.stride, m->k, src.group,
                           src.max_len, t, a.hist1, sp, m->d_ctr);
        hipLaunchKernelGGL(k_part_prefix1, dim3(1), dim3(MAXB1), 0, ctx->stream, a, nb1);
        GK_HIP(ctx, hipEventRecord(ctx->pev[1], ctx->stream));
        // (A variant that also sorts THIS kernel's keys in LDS before writing was measured slower,
        //  0.76 vs 0.68 ms at C2: P2 is bound by the two window-extraction passes, not by its stores.)
        hipLaunchKernelGGL(k_part_scatter1_reads<W>, dim3(grid), dim3(PBLOCK), 0, ctx->stream, d_rec, src.nreads, d_off, src.stride, m->k, src.group,
                           src.max_len, t, a.l1_base, a.cursor1, ps->bufA);
    } else {
        if (!op1) {
            const int grid = (int)std::min<u64>(std::max<u64>((nkeys_in + BLOCK - 1) / BLOCK, 1), (u64)cu8);
            hipLaunchKernelGGL(k_part_hist1_keys<W>, dim3(grid), dim3(BLOCK), 0, ctx->stream, d_keys, nkeys_in, t, a.hist1);
            hipLaunchKernelGGL(k_part_prefix1, dim3(1), dim3(MAXB1), 0, ctx->stream, a, nb1);
        }
        GK_HIP(ctx, hipEventRecord(ctx->pev[1], ctx->stream));
        const int g2 = (int)std::min<u64>(std::max<u64>((nkeys_in + TILE2 - 1) / TILE2, 1), (u64)ctx->cu_count * 4);
        hipLaunchKernelGGL(k_part_scatter1_keys<W>, dim3(g2), dim3(PBLOCK), ScatterLds<W>::bytes(std::max(256u, nb1)), ctx->stream, d_keys, nkeys_in, t, a, ps->bufA, sp);
    }
    if (!pipelined) {
        if (op_wide_now()) a.chunk_keys = op_chunk_keys();      // (only the over-provisioned fine level's P4 reads it)
        if (op1) hipLaunchKernelGGL(k_part_prefix1, dim3(1), dim3(MAXB1), 0, ctx->stream, a, nb1);     // chunk / range tables from the cursors
    }
    GK_HIP(ctx, hipGetLastError());
    GK_HIP(ctx, hipEventRecord(ctx->pev[2], ctx->stream));
    // the NEXT chunk's (or the caller's next batch's) upload, armed before this batch began, starts when this L1 scatter ends
    if (int rc = map_fire_prefetch(m, ctx->pev[2])) return rc;

    // ---- between the levels ----------------------------------------------------------------------------------
    // Did the spill list overflow (extreme skew)?  Then the batch has to take the direct path.  A table
    // that holds data must not be touched before that is known (host round trip here); a table that is
    // being rebuilt from empty can simply be cleared again, so P5 goes out first and the flag is read
    // with everything else after it — one host round trip less per batch on the common path.
    // With plan.estimate the round trip is taken anyway: it brings the distinct-key sample back.
    unsigned long long nspill = 0;
    u32 ovf = 0;
    // [nspill u64][overflow u32][n_failed u32] of the blob -> the pinned area behind the counters' mirror; the caller syncs
    struct PartStatus { unsigned long long nspill; u32 overflow, n_failed; };
    PartStatus *h_st = reinterpret_cast<PartStatus *>(m->h_status + 64);
    static_assert(sizeof(Counters) <= 64 && sizeof(PartStatus) == 16, "pinned status area layout");
    auto request_status = [&]() -> int {
        GK_HIP(ctx, hipMemcpyAsync(h_st, a.nspill, sizeof(PartStatus), hipMemcpyDeviceToHost, ctx->stream));
        return GK_OK;
    };
    auto abandon = [&](bool table_touched) -> int {
        if (d_rec) {   // P2 already counted this batch's windows: back to what the counter held when the batch began.  (Until round 3
                       // the batch's BOUND was subtracted — for super-k-mer records an upper bound, not the windows they hold — which
                       // took earlier batches' windows with it: a 1.9e9-window exchange that retried its second batch reported half.)
            unsigned long long occ = occ_before;
            GK_HIP(ctx, hipMemcpy(&m->d_ctr->occurrences, &occ, 8, hipMemcpyHostToDevice));
            m->occ_cached = occ;
        }
        if (table_touched) GK_HIP(ctx, hipMemset(&m->d_ctr->size, 0, 8));     // from empty: whatever P5 claimed is void
        m->retries_direct++;
        return PART_RETRY_DIRECT;
    };
    if (src.verify_uniform && !(d_rec && op1)) return fail(ctx, GK_E_STATE, "unverified host stream reached a path that cannot verify it");
    // (an unverified host stream behind a table that is being rebuilt from empty is checked at the END with everything else:
    //  if a length byte differed, the half-built table is simply void again — one host round trip less per batch)
    if (sync_between) {
        Counters *hc = reinterpret_cast<Counters *>(m->h_status);
        if (int rc = request_status()) return rc;
        GK_HIP(ctx, hipMemcpyAsync(hc, m->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, ctx->stream));
        GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const Counters c = *hc;
        nspill = h_st->nspill; ovf = h_st->overflow;
        if (src.verify_uniform && c.format) {       // not the uniform stream it looked like: only scratch was touched
            GK_HIP(ctx, hipMemsetAsync(&m->d_ctr->format, 0, sizeof(u32), ctx->stream));
            GK_HIP(ctx, hipEventRecord(ctx->pev[5], ctx->stream));
            abandon(false);
            m->retries_direct--;
            return PART_NOT_UNIFORM;
        }
        if (ovf) {
            GK_HIP(ctx, hipEventRecord(ctx->pev[5], ctx->stream));
            return abandon(false);
        }
        if (plan.estimate) {
            // new distinct keys of this batch ~ 1024 x the sample keys it added (+ 4 sigma of that estimate + what the
            // spill list holds, which the sample has seen too but the regions have not)
            const u64 claims = c.sample_claims - std::min<u64>(c.sample_claims, m->sample_claims_seen);
            m->sample_claims_seen = c.sample_claims;
            const bool saturated = c.sample_claims * 2 > m->sample_mask;          // the set is half full: stop trusting it
            u64 est_new = saturated ? nkeys_bound : (u64)((double)claims * 1024.0 * 1.05 + 4.0 * std::sqrt((double)claims + 1.0) * 1024.0 + 4096.0);
            est_new = std::min<u64>(est_new, nkeys_bound);
            m->est_distinct_last = est_new;
            if (!ctx->hook_no_reserve) {
                // (error k-mers accumulate sub-linearly with the reads: 60 % of the proportional extrapolation)
                const u64 ahead = (u64)((double)est_new * (1.0 + 0.6 * (std::max(plan.grow_ahead, 1.0) - 1.0)));
                if (int rc = map_make_room(m, est_new, ahead, from_empty, true)) return rc;     // may replace the table (same lnb1: P2 has cut the batch by L1 bucket)
            }
            t = Table<W>{reinterpret_cast<Slot<W> *>(m->slots), m->nb2, m->lnb1, m->k == 64 ? 1u : 0u, 0u};
            // many repeats: segment sizes are far from binomial -> exact fine level
            fine_exact = !op1 || plan.fine_exact || (double)est_new < 0.5 * (double)nkeys_bound;
        }
    }
    if (!part_supported(m)) {
        // The table was just grown for this batch with its L1 fan-out kept (P2 has cut the batch by L1 bucket) and now has more
        // fine buckets per L1 bucket than P4 sorts: only scratch was touched (the table itself is valid), so the batch takes
        // the direct path; the next growth outside a batch picks a larger fan-out and the pipeline is back.
        if (!sync_between) return fail(ctx, GK_E_CAPACITY, "table too large for the partitioned insert path");     // (callers check part_supported first)
        GK_HIP(ctx, hipEventRecord(ctx->pev[5], ctx->stream));
        abandon(false);
        m->retries_direct--;
        return PART_OUTGREW;
    }
    if (!pipelined) { if (int rc = part_prepare_fine(m, ps, nkeys_bound, !fine_exact, &a)) return rc; }
    const u64 nseg = t.nseg();
    GK_HIP(ctx, hipEventRecord(ctx->gev, ctx->stream));

    // ---- stage B: P3 + scans + P4 (the fine level) ---------------------------------------------------------
    const u64 max_ranges = max_chunks / a.range_chunks + MAXB1 + 1;
    const u64 *fine_keys = ps->bufB;
    if (fine_exact) {
        const int gr = (int)std::min<u64>(max_ranges, (u64)ctx->cu_count * 4);
        hipLaunchKernelGGL(k_part_hist2r<W>, dim3(gr), dim3(PBLOCK), m->nb2 * 4, ctx->stream, ps->bufA, t, a, max_ranges);
        hipLaunchKernelGGL(k_part_scan2, dim3((m->nb2 + 255) / 256, nb1), dim3(256), 0, ctx->stream, a, m->nb2);
        hipLaunchKernelGGL(k_part_prefix2, dim3(nb1), dim3(256), 0, ctx->stream, a, nb1, m->nb2);
        GK_HIP(ctx, hipEventRecord(ctx->pev[3], ctx->stream));
        // few keys per bin and chunk: straight scatter with per-range cursors; otherwise sort each chunk in LDS (coalesced runs)
        // (measured at C3, nb2 = 1525: 67 ms against 52 ms for the sorted form — 8-byte stores to 64 different lines per
        //  instruction leave L2 as partial-line writes; kept as an A/B option, never chosen by default)
        const bool direct = ctx->hook_p4_direct > 0;
        if (direct)
            hipLaunchKernelGGL(k_part_scatter2_direct<W>, dim3((int)std::min<u64>(max_ranges, (u64)ctx->cu_count * 8)), dim3(PBLOCK), m->nb2 * 4,
                               ctx->stream, ps->bufA, t, a, max_ranges, ps->bufB);
        else if (wide_fits() && (ctx->hook_p4_wide >= 0 ? ctx->hook_p4_wide != 0 : m->nb2 >= 512)) {
            // many fine buckets: sort 8192 keys at a time (1024 threads) — twice the keys per bin and visit, so fewer
            // partial-line writes, and half the per-bin bookkeeping per key
            if constexpr (W == 1) {
                const size_t wide_lds = ScatterLds<1, 1024>::bytes(m->nb2) + part_tables_bytes(nb1);
                const int gw = (int)std::min<u64>(max_ranges, (u64)ctx->cu_count * 2);
                // from ~1000 fine buckets up a 12288-key sort where it still fits the LDS (runs half again as long: "p4_wide" = 2 forces it)
                const size_t xl_lds = ScatterLds<1, 1024, 12>::bytes(m->nb2) + part_tables_bytes(nb1);
                const bool xl = xl_lds <= LDS_BYTES_PER_CU && (ctx->hook_p4_wide >= 0 ? ctx->hook_p4_wide == 2 : m->nb2 >= 1024);
                if (xl) hipLaunchKernelGGL((k_part_scatter2<1, true, 1024, 12>), dim3(gw), dim3(1024), xl_lds, ctx->stream, ps->bufA, t, a, max_ranges, ps->bufB, 0u, nb1);
                else hipLaunchKernelGGL((k_part_scatter2<1, true, 1024>), dim3(gw), dim3(1024), wide_lds, ctx->stream, ps->bufA, t, a, max_ranges, ps->bufB, 0u, nb1);
            }
        } else if (wide2_fits() && ctx->hook_p4_wide > 0) {
            if constexpr (W == 2) {      // 16-byte keys: 6144-key sorts on 1024 threads — opt-in ("p4_wide" = 1): 1.00 / 0.94 -> 0.91 / 0.98 ms at k = 55, no clear gain
                const size_t lds2 = ScatterLds<2, 1024, 6>::bytes(m->nb2) + part_tables_bytes(nb1);
                const int gw = (int)std::min<u64>(max_ranges, (u64)ctx->cu_count * 2);
                hipLaunchKernelGGL((k_part_scatter2<2, true, 1024, 6>), dim3(gw), dim3(1024), lds2, ctx->stream, ps->bufA, t, a, max_ranges, ps->bufB, 0u, nb1);
            }
        } else
            hipLaunchKernelGGL((k_part_scatter2<W, true, PBLOCK>), dim3(gr), dim3(PBLOCK), ScatterLds<W>::bytes(m->nb2) + part_tables_bytes(nb1), ctx->stream, ps->bufA, t, a,
                               max_ranges, ps->bufB, 0u, nb1);
    }
    auto launch_p5 = [&](hipStream_t st, u64 seg_lo, u64 seg_hi) {
        const int gseg = (int)std::min<u64>(seg_hi - seg_lo, (u64)ctx->cu_count * GK_P5_GRID_PER_CU);
        if constexpr (W == 1) {
            if (cslots) {
                hipLaunchKernelGGL((k_seg_insert<1, CSlot>), dim3(gseg), dim3(SBLOCK), lds, st, Table<1, CSlot>{reinterpret_cast<CSlot *>(m->slots), t.nb2, t.lnb1, 0u, 0u}, fine_keys, a,
                                   from_empty ? 1 : 0, m->d_ctr, seg_lo, seg_hi);
                return;
            }
        }
        hipLaunchKernelGGL((k_seg_insert<W, Slot<W>>), dim3(gseg), dim3(SBLOCK), lds, st, t, fine_keys, a, from_empty ? 1 : 0, m->d_ctr, seg_lo, seg_hi);
    };
    // Stripes (over-provisioned fine level): P4 is bound by its LDS sort and leaves half the memory system idle, P5 streams
    // the table and leaves the ALUs idle.  With the L1 buckets cut into stripes, P5 of stripe i runs on the second stream
    // beside P4 of stripe i+1 (option "p45_stripes": 1 = one after the other).
    int stripes = fine_exact ? 1 : std::max(1, std::min<int>({ctx->hook_p45_stripes > 0 ? ctx->hook_p45_stripes : 1, (int)nb1, 16}));
    while (nb1 % (u64)stripes) stripes--;
    if (pipelined && !have_last) return fail(ctx, GK_E_STATE, "partitioned insert: a pipelined batch ended without its last piece");
    if (pipelined) {
        // the earlier pieces' P4 (second stream) join here; the last piece's runs on this stream
        GK_HIP(ctx, hipEventRecord(ctx->gev2, ctx->aux_stream));
        GK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->gev2, 0));
        GK_HIP(ctx, hipEventRecord(ctx->pev[3], ctx->stream));
        launch_p4_op(a_last, ctx->stream, 0u, (u32)nb1, 1);
    } else if (!fine_exact && stripes == 1) {
        GK_HIP(ctx, hipEventRecord(ctx->pev[3], ctx->stream));
        launch_p4_op(a, ctx->stream, 0u, (u32)nb1, 1);
    }
    if (stripes == 1) {
        GK_HIP(ctx, hipGetLastError());
        GK_HIP(ctx, hipEventRecord(ctx->pev[4], ctx->stream));
        // ---- P5 --------------------------------------------------------------------------------------------
        launch_p5(ctx->stream, 0, nseg);
        GK_HIP(ctx, hipGetLastError());
        GK_HIP(ctx, hipEventRecord(ctx->pev[5], ctx->stream));
    } else {
        // (phase events: [3,4] = the P4 launches with the overlapped P5s beside them, [4,5] = what is left of P5 after the last P4)
        GK_HIP(ctx, hipEventRecord(ctx->pev[3], ctx->stream));
        const u32 per = (u32)(nb1 / (u64)stripes);
        for (int s = 0; s < stripes; s++) {
            launch_p4_op(a, ctx->stream, s * per, (s + 1) * per, (u64)stripes);
            GK_HIP(ctx, hipGetLastError());
            hipEvent_t ev = ctx->cev[s % 16];
            GK_HIP(ctx, hipEventRecord(ev, ctx->stream));
            GK_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ev, 0));
            ctx->copy_other_pending = true;
            launch_p5(ctx->copy_stream, (u64)s * per * m->nb2, (u64)(s + 1) * per * m->nb2);
            GK_HIP(ctx, hipGetLastError());
        }
        GK_HIP(ctx, hipEventRecord(ctx->pev[4], ctx->stream));
        GK_HIP(ctx, hipEventRecord(ctx->gev2, ctx->copy_stream));
        GK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->gev2, 0));
        GK_HIP(ctx, hipEventRecord(ctx->pev[5], ctx->stream));
    }
    // the spill list may also have grown in P4 (over-provisioned fine level): read it (again) behind P5
    // failures (a segment filled up): grow, then replay those buckets through the direct path
    if (int rc = request_status()) return rc;
    if (src.verify_uniform && !sync_between) {                 // the deferred check of an optimistic uniform stream
        Counters *hc = reinterpret_cast<Counters *>(m->h_status);
        GK_HIP(ctx, hipMemcpyAsync(hc, m->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, ctx->stream));
        GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (hc->format) {
            GK_HIP(ctx, hipMemsetAsync(&m->d_ctr->format, 0, sizeof(u32), ctx->stream));
            abandon(true);
            m->retries_direct--;
            return PART_NOT_UNIFORM;
        }
    }
    if (int rc = map_sync_counters(m)) return rc;              // (one stream sync for both copies)
    nspill = h_st->nspill; ovf = h_st->overflow;
    const u32 n_failed = h_st->n_failed;
    if (ovf) {
        if (from_empty) return abandon(true);
        // P4's regions overflowed into a full spill list while the table held data: the keys that did not fit are lost
        // to this pipeline but P5 has already merged the rest.  Cannot happen with the exact fine level; the
        // over-provisioned one is only chosen for a table that holds data when the batch is known to be near-distinct.
        return fail(ctx, GK_E_STATE, "partitioned insert: spill list overflowed behind a non-empty table (internal sizing error)");
    }
    m->failed_segments += n_failed;
    m->spilled_keys += nspill;
    if (!fine_exact && nspill > nkeys_bound / 64) m->repeats = true;       // regions sized for distinct keys are the wrong tool for this data
    if (n_failed) {
        std::vector<u32> failed(n_failed);
        GK_HIP(ctx, hipMemcpy(failed.data(), a.failed, n_failed * 4ull, hipMemcpyDeviceToHost));
        std::vector<unsigned long long> fb;
        std::vector<u32> cur;
        if (a.op2) { cur.resize(nseg); GK_HIP(ctx, hipMemcpy(cur.data(), a.cursor2, nseg * 4, hipMemcpyDeviceToHost)); }
        else { fb.resize(nseg + 1); GK_HIP(ctx, hipMemcpy(fb.data(), a.fine_base, (nseg + 1) * 8, hipMemcpyDeviceToHost)); }
        auto seg_begin = [&](u32 s) { return a.op2 ? (u64)s * a.cap2 : (u64)fb[s]; };
        auto seg_count = [&](u32 s) { return a.op2 ? std::min<u64>(cur[s], a.cap2) : (u64)(fb[s + 1] - fb[s]); };
        u64 total = 0;
        for (u32 s : failed) total += seg_count(s);
        if (int rc = map_reserve(m, std::max<u64>(std::min<u64>(total, m->capacity), m->capacity / 2))) return rc;
        for (u32 s : failed) {
            if (int rc = map_add_keys_direct(m, fine_keys + seg_begin(s) * m->W, seg_count(s))) return rc;
        }
    }
    if (nspill) {
        if (int rc = map_add_keys_direct(m, ps->spill, nspill)) return rc;
    }
    return GK_OK;
}

int part_count(gk_map *m, PartScratch **pps, const ReadSrc &src, const u64 *d_keys, u64 nkeys_in, u64 nkeys_bound, bool from_empty,
               const PartPlan &plan) {
    if (!*pps) *pps = new PartScratch();
    if (m->W == 1) return part_run<1>(m, *pps, src, d_keys, nkeys_in, nkeys_bound, from_empty, plan);
    return part_run<2>(m, *pps, src, d_keys, nkeys_in, nkeys_bound, from_empty, plan);
}

// the fine level's 4096-key sort must fit the CU's LDS beside its per-bin arrays and the L1 tables (16-byte keys with 1024 L1
// buckets: ~3500 fine buckets, a 117 GB table)
bool part_supported(const gk_map *m) {
    if (m->nb2 > (m->ctx->hook_max_nb2 > 0 ? (u32)m->ctx->hook_max_nb2 : MAX_NB2)) return false;
    const size_t need = (m->W == 1 ? ScatterLds<1>::bytes(m->nb2) : ScatterLds<2>::bytes(m->nb2)) + part_tables_bytes(1u << m->lnb1);
    return need <= LDS_BYTES_PER_CU;
}
uint64_t part_max_slots(int W) { return ((u64)MAX_NB2 << MAX_LNB1) << seg_bits_for(W); }

}  // namespace gk
