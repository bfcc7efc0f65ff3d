// gk_table.hip — the DNAMap[Int] hot path as HIP kernels for gfx950, plus its C-ABI entry points.
//
// Reference path replaced here (S/ = /root/reference/src/main/scala/ru/ifmo/genome/):
//   FreqFilter.add            S/data/FreqFilter.scala:28-36      -> k_count_reads
//   PairedEndData.getPairs    S/data/PairedEndData.scala:20-36   -> record framing read in-kernel
//   Container.update(k,v0,f)  S/ds/ArrayDNAMap.scala:129-150     -> gk::table_add (gk_device.h)
//   Container.apply           S/ds/ArrayDNAMap.scala:90-101      -> k_get
//   Container.deleteAll       S/ds/ArrayDNAMap.scala:164-173     -> k_filter_lt
//   ArrayDNAMap.rescale       S/ds/ArrayDNAMap.scala:217-230     -> k_rehash (pre-sized / doubled)
//   Container.iterator        S/ds/ArrayDNAMap.scala:175-178     -> k_export
//
// Roofline: every kernel here is HBM-bound integer work (random 64-B sector touches for probes,
// streaming for scans); no MFMA.  Algorithmic bytes per unit are stated in DESIGN.md.
#include <algorithm>
#include <sys/syscall.h>
#include <unistd.h>
#include <cstdio>
#include <cstring>
#include <vector>

#include "gk_internal.h"
#include "gk_tile.h"

using namespace gk;

// ============================================================================================
// error plumbing
// ============================================================================================
namespace gk {
static thread_local std::string tls_err;

void set_error(const gk_ctx *ctx, const std::string &msg) {
    tls_err = msg;
    if (ctx) const_cast<gk_ctx *>(ctx)->err = msg;
}
int fail(const gk_ctx *ctx, int code, const std::string &msg) {
    set_error(ctx, msg);
    return code;
}
int hip_fail(const gk_ctx *ctx, hipError_t e, const char *what) {
    (void)hipGetLastError();
    return fail(ctx, e == hipErrorNoDevice || e == hipErrorInvalidDevice ? GK_E_NODEVICE : GK_E_HIP,
                std::string(what) + ": " + hipGetErrorString(e));
}
}  // namespace gk

// ============================================================================================
// kernels
// ============================================================================================
// every slot EMPTY, written as 16-byte vectors (n slots of any of the three slot types are a whole number of vectors per segment)
template <class S> __global__ __launch_bounds__(BLOCK) void k_clear(S *slots, u64 n) {
    uint4 *v = reinterpret_cast<uint4 *>(slots);
    const u64 nvec = n * sizeof(S) / 16;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < nvec; i += (u64)gridDim.x * BLOCK) v[i] = empty_vec_of(slots, (u32)(i % 3072u));
}

// FreqFilter.add (FreqFilter.scala:28-36) for a stream of `.bin` records: one workgroup stages a
// tile of 64 reads in LDS, each wave takes reads round-robin, each lane one window of the read:
// extract -> reverse complement -> hash rule -> insert-or-increment in the HBM table.
//   offsets == nullptr : fixed stride records (record r at r*stride)
//   offsets != nullptr : offsets[r] = byte offset of record r, offsets[nreads] = end
template <int W, class S>
__global__ __launch_bounds__(BLOCK) void k_count_reads(const uint8_t *__restrict__ rec, u64 nreads,
                                                       const u32 *__restrict__ offsets, u32 stride, int k, int group, int max_len,
                                                       Table<W, S> t, Counters *ctr) {
    __shared__ __attribute__((aligned(16))) u32 tile[TILE_WORDS];
    __shared__ u32 s_claimed, s_occ;
    if (threadIdx.x == 0) { s_claimed = 0; s_occ = 0; }
    u32 claimed = 0, occ = 0, err = 0;
    const u64 ntiles = (nreads + TILE_READS - 1) / TILE_READS;
    for (u64 tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const u64 r0 = tl * TILE_READS;
        const int nr = (int)min((u64)TILE_READS, nreads - r0);
        const u64 gb = offsets ? (u64)offsets[r0] : r0 * stride;
        const u64 ge = offsets ? (u64)offsets[r0 + nr] : (r0 + nr) * stride;
        __syncthreads();                       // previous tile fully consumed
        const u64 a0 = stage_tile(tile, rec, gb, ge);
        __syncthreads();
        for_each_window<W>(tile, a0, r0, nr, offsets, stride, k, group, WindowLimits{max_len, &ctr->format}, [&](Kmer<W> x) {
            Kmer<W> y = canonical(x, k);                  // FreqFilter.scala:31-32
            claimed += table_add(t, y, 1u, &err);         // kmersFreq.update(y, 1, _ + 1)  :33
            occ++;
        });
    }
    atomicAdd(&s_claimed, claimed);
    atomicAdd(&s_occ, occ);
    if (err) ctr->error = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_claimed) atomicAdd(&ctr->size, (unsigned long long)s_claimed);
        if (s_occ) atomicAdd(&ctr->occurrences, (unsigned long long)s_occ);
    }
}

// DNAMap.update(key, c, _ + c) for keys already canonical and routed (owner-side insert of the
// PartitionedDNAMap exchange; also partition merge).  keys: W words per key, interleaved.
template <int W, class S>
__global__ __launch_bounds__(BLOCK) void k_add_keys(const u64 *__restrict__ keys, const i32 *__restrict__ counts,
                                                    u64 n, Table<W, S> t, Counters *ctr, int check_canon_k /* 0: trusted keys */) {
    __shared__ u32 s_claimed;
    if (threadIdx.x == 0) s_claimed = 0;
    __syncthreads();
    u32 claimed = 0, err = 0;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        Kmer<W> key;
        if constexpr (W == 1) key = Kmer<1>{keys[i]};
        else key = Kmer<2>{keys[2 * i], keys[2 * i + 1]};
        claimed += table_add(t, key, counts ? (u32)counts[i] : 1u, &err);
        if (check_canon_k && !(canonical(key, check_canon_k) == key)) ctr->noncanon = 1u;
    }
    if (claimed) atomicAdd(&s_claimed, claimed);
    if (err) ctr->error = 1;
    __syncthreads();
    if (threadIdx.x == 0 && s_claimed) atomicAdd(&ctr->size, (unsigned long long)s_claimed);
}

// putNew-like insert of keys that are KNOWN to be absent and distinct (the gather of a PartitionedDNAMap: every key has exactly
// one owner), graph layout: one CAS claims the slot, count and — when the owners classified their keys — the degree mask go in
// with plain stores.  No second lookup for the mask, no count atomic.
template <int W>
__global__ __launch_bounds__(BLOCK) void k_add_unique(const u64 *__restrict__ keys, const i32 *__restrict__ counts, const uint8_t *__restrict__ masks, u64 n,
                                                      Table<W> t, Counters *ctr) {
    u32 err = 0;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        Kmer<W> key;
        if constexpr (W == 1) key = Kmer<1>{keys[i]};
        else key = Kmer<2>{keys[2 * i], keys[2 * i + 1]};
        const u64 h = slot_hash(key);
        Slot<W> *seg = t.slots + ((u64)seg_of(t, h) << SegBits<W>::value);
        const i64 at = seg_claim_unique(seg, home_pos(t, h), key, t.tagged);
        if (at < 0) { err = 1; continue; }
        seg[at].extra = (u32)counts[i] - 1u;
        seg[at].aux = masks ? (u32)masks[i] : 0u;
    }
    if (err) ctr->error = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&ctr->size, (unsigned long long)n);
}

// ArrayDNAMap.rescale (ArrayDNAMap.scala:217-230): move every live (key, count) into a new table.
template <int W, class SO, class SN>
__global__ __launch_bounds__(BLOCK) void k_rehash(const SO *__restrict__ old, u64 ncap, Table<W, SN> t, Counters *ctr) {   // old and new share t.tagged
    u32 err = 0;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < ncap; i += (u64)gridDim.x * BLOCK) {
        if (!slot_live(&old[i])) continue;
        Kmer<W> key = slot_key(old, i, t.tagged);
        // every key of the old table is unique: one CAS claims its new slot, the count goes in with a plain store
        const u64 h = slot_hash(key);
        SN *seg = t.slots + ((u64)seg_of(t, h) << SegBits<W>::value);
        const i64 at = seg_claim_unique(seg, home_pos(t, h), key, t.tagged);
        if (at < 0) { err = 1; continue; }
        seg[at].extra = old[i].extra;
    }
    if (err) ctr->error = 1;
}

// Container.deleteAll((k, v) => v < rounds) (ArrayDNAMap.scala:164-173): full scan, tombstone.
template <class S>
__global__ __launch_bounds__(BLOCK) void k_filter_lt(S *slots, u64 ncap, i32 rounds, unsigned long long *removed) {
    __shared__ u32 s_rm;
    if (threadIdx.x == 0) s_rm = 0;
    __syncthreads();
    u32 rm = 0;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < ncap; i += (u64)gridDim.x * BLOCK) {
        if (slot_live(&slots[i]) && (i32)slot_count(&slots[i]) < rounds) {
            slot_tomb(&slots[i]);
            rm++;
        }
    }
    if (rm) atomicAdd(&s_rm, rm);
    __syncthreads();
    if (threadIdx.x == 0 && s_rm) atomicAdd(removed, (unsigned long long)s_rm);
}

// deleteAll((k, v) => v < rounds) (ArrayDNAMap.scala:164-173) and the rescale that follows it (:214), as ONE streaming pass:
// the new table keeps the L1 bucket of every key and its fine bucket is a monotone rescaling of the same 32 hash bits
// (seg_fine), so the keys of new segment (b1, fn) all live in a short run of OLD segments of the same L1 bucket.  One
// workgroup per new segment reads that run (coalesced), keeps the survivors that are its own, builds the segment in LDS and
// writes it out whole: no tombstone writes, no clear of the new table, no random access to HBM.
static constexpr int CBLOCK = 512;
// live slots with count >= rounds in every `every`-th segment (sizes the new table; exact sizes come from k_compact_seg)
template <int W, class S>
__global__ __launch_bounds__(BLOCK) void k_count_ge_sample(const S *slots, u64 nseg, u32 every, i32 rounds, unsigned long long *out) {
    __shared__ u32 s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    constexpr u32 NS = 1u << SegBits<W>::value;
    u32 n = 0;
    for (u64 s = (u64)blockIdx.x * every; s < nseg; s += (u64)gridDim.x * every)
        for (u32 i = threadIdx.x; i < NS; i += BLOCK) {
            const S *p = &slots[(s << SegBits<W>::value) + i];
            if (slot_live(p) && (i32)slot_count(p) >= rounds) n++;
        }
    if (n) atomicAdd(&s_n, n);
    __syncthreads();
    if (threadIdx.x == 0 && s_n) atomicAdd(out, (unsigned long long)s_n);
}
// claim a slot of a segment held in LDS for a key known to be absent (every key of the source table is unique): the slot's index,
// or -1 if the segment is full.  Same slot protocols as gk_device.h, with the LDS forms of the atomics.
__device__ __forceinline__ i64 lds_claim_unique(Slot<1> *seg, u32 pos, Kmer<1> key) {
    constexpr u32 smask = (1u << SegBits<1>::value) - 1u;
    u32 p = pos;
    for (u32 n = 0; n <= smask; n++, p = (p + 1) & smask)
        if (atomicCAS(reinterpret_cast<unsigned long long *>(&seg[p].w0), (unsigned long long)KEY_EMPTY, (unsigned long long)key.lo) == KEY_EMPTY) return (i64)p;
    return -1;
}
__device__ __forceinline__ i64 lds_claim_unique(Slot<2> *seg, u32 pos, Kmer<2> key) {
    constexpr u32 smask = (1u << SegBits<2>::value) - 1u;
    const Stored<2> k = to_stored(key);
    u32 p = pos;
    for (u32 n = 0; n <= smask; n++, p = (p + 1) & smask) {
        // (keys are unique: a slot whose w0 we claim — or that holds an equal w0 — is ours only if its w1 is free)
        const unsigned long long c0 = atomicCAS(reinterpret_cast<unsigned long long *>(&seg[p].w0), (unsigned long long)KEY_EMPTY, (unsigned long long)k.w0);
        if ((c0 == KEY_EMPTY || c0 == k.w0) &&
            atomicCAS(reinterpret_cast<unsigned long long *>(&seg[p].w1), (unsigned long long)KEY_EMPTY, (unsigned long long)k.w1) == KEY_EMPTY) return (i64)p;
    }
    return -1;
}
__device__ __forceinline__ i64 lds_claim_unique(CSlot *seg, u32 pos, Kmer<1> key) {
    constexpr u32 smask = (1u << SegBits<1>::value) - 1u;
    const u32 k0 = c_w0(key), k1 = c_w1(key);
    u32 p = pos;
    for (u32 n = 0; n <= smask; n++, p = (p + 1) & smask) {
        const u32 c0 = atomicCAS(&seg[p].w0, KEY_EMPTY32, k0);
        if ((c0 == KEY_EMPTY32 || c0 == k0) && atomicCAS(&seg[p].w1, KEY_EMPTY32, k1) == KEY_EMPTY32) return (i64)p;
    }
    return -1;
}
// SO / SN: slot types of the old and the new table (a count table of 12-byte slots becomes a graph table of 16-byte slots here)
template <int W, class SO, class SN>
__global__ __launch_bounds__(CBLOCK) void k_compact_seg(Table<W, SO> old, Table<W, SN> nw, i32 rounds, Counters *ctr, unsigned long long *kept_total) {
    extern __shared__ uint4 cseg_raw[];
    constexpr u32 S = 1u << SegBits<W>::value;
    constexpr u32 NVEC = S * sizeof(SN) / 16;
    SN *seg = reinterpret_cast<SN *>(cseg_raw);
    __shared__ u32 s_kept, s_err;
    u32 wg_kept = 0;
    const u64 nseg_new = nw.nseg();
    for (u64 sn = blockIdx.x; sn < nseg_new; sn += gridDim.x) {
        const u32 b1 = (u32)(sn / nw.nb2), fn = (u32)(sn % nw.nb2);
        __syncthreads();
        if (threadIdx.x == 0) { s_kept = 0; s_err = 0; }
        for (u32 i = threadIdx.x; i < NVEC; i += CBLOCK) cseg_raw[i] = empty_vec_of(seg, i);
        // the 32 hash bits x with seg_fine(nw, .) == fn: [xlo, xhi); the old fine buckets they fall into: f0 .. f1
        const u64 xlo = (((u64)fn << 32) + nw.nb2 - 1) / nw.nb2, xhi = ((((u64)fn + 1) << 32) + nw.nb2 - 1) / nw.nb2;
        const u32 f0 = (u32)((xlo * old.nb2) >> 32), f1 = (u32)(((xhi - 1) * old.nb2) >> 32);
        __syncthreads();
        u32 kept = 0;
        bool err = false;
        for (u32 f = f0; f <= f1; f++) {
            const SO *src = old.slots + (((u64)b1 * old.nb2 + f) << SegBits<W>::value);
            if constexpr (std::is_same<SO, CSlot>::value) {
                // 12-byte source slots: four of them are three whole 16-byte vectors — read those, not 4-byte fields
                const uint4 *sv = reinterpret_cast<const uint4 *>(src);
                for (u32 g4 = threadIdx.x; g4 < S / 4; g4 += CBLOCK) {
                    const uint4 a = sv[3 * g4], b = sv[3 * g4 + 1], c = sv[3 * g4 + 2];
                    const u32 w[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const u32 h0 = w[3 * q], h1 = w[3 * q + 1], ex = w[3 * q + 2];
                        if (h0 >= KEY_TOMB32 || (i32)(ex + 1u) < rounds) continue;
                        const Kmer<W> key = Kmer<1>{(u64)h0 | ((u64)h1 << 31)};
                        const u64 h = slot_hash(key);
                        if (seg_fine(nw, h) != fn) continue;
                        const i64 at = lds_claim_unique(seg, home_pos(nw, h), key);
                        if (at >= 0) { seg[at].extra = ex; kept++; }
                        else err = true;
                    }
                }
            } else {
                for (u32 i = threadIdx.x; i < S; i += CBLOCK) {
                    if (!slot_live(&src[i]) || (i32)slot_count(&src[i]) < rounds) continue;
                    const Kmer<W> key = slot_key(src, i, 0u);
                    const u64 h = slot_hash(key);
                    if (seg_fine(nw, h) != fn) continue;
                    const i64 at = lds_claim_unique(seg, home_pos(nw, h), key);
                    if (at >= 0) { seg[at].extra = src[i].extra; kept++; }
                    else err = true;
                }
            }
        }
        if (kept) atomicAdd(&s_kept, kept);
        if (err) s_err = 1;
        __syncthreads();
        uint4 *dst = reinterpret_cast<uint4 *>(nw.slots + (sn << SegBits<W>::value));
        for (u32 i = threadIdx.x; i < NVEC; i += CBLOCK) dst[i] = cseg_raw[i];
        if (threadIdx.x == 0) { wg_kept += s_kept; if (s_err) ctr->error = 1; }
    }
    if (threadIdx.x == 0 && wg_kept) atomicAdd(kept_total, (unsigned long long)wg_kept);
}

// Container.apply (ArrayDNAMap.scala:90-101) for a batch of keys.
template <int W, class S>
__global__ __launch_bounds__(BLOCK) void k_get(const u64 *__restrict__ lo, const u64 *__restrict__ hi, u64 n,
                                               Table<W, S> t, i32 *counts, uint8_t *found) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        Kmer<W> key;
        if constexpr (W == 1) key = Kmer<1>{lo[i]};
        else key = Kmer<2>{lo[i], hi[i]};
        i64 s = table_find(t, key);
        if (counts) counts[i] = s >= 0 ? (i32)slot_count(&t.slots[s]) : -1;
        if (found) found[i] = s >= 0;
    }
}

// Self-check of the table's invariants (what `size` and the slot protocol promise): every live key is found again at
// its own slot (a key stored twice, or in a segment its hash does not name, is not), counts add up.
template <int W, class S>
__global__ __launch_bounds__(BLOCK) void k_verify(Table<W, S> t, unsigned long long *out /* live, duplicates or misplaced, sum of counts, checksum */) {
    unsigned long long live = 0, bad = 0, sum = 0, chk = 0;
    const u64 ncap = t.capacity();
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < ncap; i += (u64)gridDim.x * BLOCK) {
        if (!slot_live(&t.slots[i])) continue;
        live++;
        sum += slot_count(&t.slots[i]);
        const Kmer<W> key = slot_key(t.slots, i, t.tagged);
        if (table_find(t, key) != (i64)i) bad++;
        // order-independent content checksum: sum over entries of a mix of (key, count) — slot order, table size and
        // insertion order do not enter
        u64 kh;
        if constexpr (W == 1) kh = mix64(key.lo ^ 0x243f6a8885a308d3ULL);
        else kh = mix64(key.lo ^ mix64(key.hi ^ 0x13198a2e03707344ULL));
        chk += mix64(kh + (u64)slot_count(&t.slots[i]) * 0x9e3779b97f4a7c15ULL);
    }
    for (int d = 32; d; d >>= 1) {
        live += __shfl_down(live, d); bad += __shfl_down(bad, d); sum += __shfl_down(sum, d); chk += __shfl_down(chk, d);
    }
    if ((threadIdx.x & 63) == 0) {
        if (live) atomicAdd(&out[0], live);
        if (bad) atomicAdd(&out[1], bad);
        if (sum) atomicAdd(&out[2], sum);
        atomicAdd(&out[3], chk);
    }
}

// block-wide exclusive scan of a 0/1 flag using ballots (wave64) + one LDS word per wave
__device__ __forceinline__ u32 block_scan_flag(bool flag, u32 *total, u32 *lds4) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long b = __ballot(flag);
    u32 wprefix = (u32)__popcll(b & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) lds4[wave] = (u32)__popcll(b);
    __syncthreads();
    u32 base = 0, tot = 0;
    for (int w = 0; w < BLOCK / 64; ++w) {
        u32 c = lds4[w];
        if (w < wave) base += c;
        tot += c;
    }
    *total = tot;
    return base + wprefix;
}

// Container.iterator (ArrayDNAMap.scala:175-178): compact every live (key, count) to dense arrays.
// One atomic per 256-slot group reserves the output range; order within a group is slot order.
template <int W, class S>
__global__ __launch_bounds__(BLOCK) void k_export(const S *__restrict__ slots, u64 ncap, u32 tagged, u64 *lo, u64 *hi, i32 *cnt,
                                                  unsigned long long *cursor) {
    __shared__ u32 lds4[BLOCK / 64];
    __shared__ unsigned long long s_base;
    const u64 ngroups = (ncap + BLOCK - 1) / BLOCK;
    for (u64 g = blockIdx.x; g < ngroups; g += gridDim.x) {
        u64 i = g * BLOCK + threadIdx.x;
        bool live = i < ncap && slot_live(&slots[i]);
        u32 tot;
        u32 pos = block_scan_flag(live, &tot, lds4);
        if (threadIdx.x == 0 && tot) s_base = atomicAdd(cursor, (unsigned long long)tot);
        __syncthreads();
        if (live) {
            Kmer<W> key = slot_key(slots, i, tagged);
            u64 o = s_base + pos;
            lo[o] = key.lo;
            if constexpr (W == 2) { if (hi) hi[o] = key.hi; }
            else { if (hi) hi[o] = 0; }
            cnt[o] = (i32)slot_count(&slots[i]);
        }
        __syncthreads();
    }
}

// live (key, count) of a RANGE of slots, packed: keys interleaved W words each (what k_add_keys takes), counts apart.
// `first` = index of slots[0] in its table (a multiple of the segment size: a tagged slot's last base is its index mod 4).
template <int W, class S>
__global__ __launch_bounds__(BLOCK) void k_export_packed(const S *__restrict__ slots, u64 n, u64 first, u32 tagged, u64 *keys, i32 *cnt,
                                                         unsigned long long *cursor, uint8_t *masks = nullptr /* the annotation word's low byte (graph layout only) */) {
    __shared__ unsigned long long s_base;
    __shared__ u32 wsum[BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 ngroups = (n + BLOCK - 1) / BLOCK;
    for (u64 g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const u64 i = g * BLOCK + threadIdx.x;
        const bool live = i < n && slot_live(&slots[i]);
        const unsigned long long b = __ballot(live);
        const u32 wprefix = (u32)__popcll(b & ((1ull << lane) - 1ull));
        __syncthreads();
        if (lane == 0) wsum[wave] = (u32)__popcll(b);
        __syncthreads();
        u32 base = 0, tot = 0;
        for (int w = 0; w < BLOCK / 64; ++w) { if (w < wave) base += wsum[w]; tot += wsum[w]; }
        if (threadIdx.x == 0 && tot) s_base = atomicAdd(cursor, (unsigned long long)tot);
        __syncthreads();
        if (live) {
            const Kmer<W> key = slot_key_tag(slots, i, tagged ? (u32)((first + i) & 3u) : 0u);
            const u64 o = s_base + base + wprefix;
            if constexpr (W == 1) keys[o] = key.lo;
            else { keys[2 * o] = key.lo; keys[2 * o + 1] = key.hi; }
            cnt[o] = (i32)slot_count(&slots[i]);
            if constexpr (!std::is_same<S, CSlot>::value) { if (masks) masks[o] = (uint8_t)slots[i].aux; }
        }
    }
}

// ============================================================================================
// host side
// ============================================================================================
static inline int grid_for(const gk_ctx *ctx, u64 work_items, int per_block) {
    u64 blocks = (work_items + per_block - 1) / per_block;
    u64 cap = (u64)ctx->cu_count * 8;      // 8 resident 256-thread workgroups per CU
    if (blocks < 1) blocks = 1;
    return (int)std::min(blocks, cap);
}

template <int W, class S = Slot<W>> static Table<W, S> table_of(const gk_map *m) {
    return Table<W, S>{reinterpret_cast<S *>(m->slots), m->nb2, m->lnb1, m->k == 64 ? 1u : 0u, 0u};
}

static int streaming_rebuild(gk_map *m, uint32_t nnb2, uint32_t nlnb1, uint64_t ncap, int32_t rounds, int new_layout, bool *done);     // (defined next to its kernel's host code)
static void launch_clear(gk_ctx *ctx, int W, int layout, void *slots, uint64_t cap) {
    const int grid = grid_for(ctx, cap, BLOCK * 4);
    if (W == 2) hipLaunchKernelGGL(k_clear<Slot<2>>, dim3(grid), dim3(BLOCK), 0, ctx->stream, (Slot<2> *)slots, cap);
    else if (layout == LAYOUT_GRAPH) hipLaunchKernelGGL(k_clear<Slot<1>>, dim3(grid), dim3(BLOCK), 0, ctx->stream, (Slot<1> *)slots, cap);
    else hipLaunchKernelGGL(k_clear<CSlot>, dim3(grid), dim3(BLOCK), 0, ctx->stream, (CSlot *)slots, cap);
}
static int alloc_table(gk_ctx *ctx, int W, int layout, uint64_t cap, void **out) {
    GK_HIP(ctx, hipMalloc(out, cap * slot_bytes(W, layout)));
    launch_clear(ctx, W, layout, *out, cap);
    GK_HIP(ctx, hipGetLastError());
    return GK_OK;
}
// move every live (key, count) of `old` (layout lo) into the table `nslots` (geometry nnb2 / nlnb1, layout ln) with k_rehash
static void launch_rehash(gk_map *m, int ln, void *nslots, uint32_t nnb2, uint32_t nlnb1) {
    gk_ctx *ctx = m->ctx;
    const int grid = grid_for(ctx, m->capacity, BLOCK);
    const u32 tagged = m->k == 64 ? 1u : 0u;
    if (m->W == 2)
        hipLaunchKernelGGL((k_rehash<2, Slot<2>, Slot<2>>), dim3(grid), dim3(BLOCK), 0, ctx->stream, (const Slot<2> *)m->slots, m->capacity, Table<2>{(Slot<2> *)nslots, nnb2, nlnb1, tagged, 0u}, m->d_ctr);
    else if (m->layout == LAYOUT_GRAPH && ln == LAYOUT_GRAPH)
        hipLaunchKernelGGL((k_rehash<1, Slot<1>, Slot<1>>), dim3(grid), dim3(BLOCK), 0, ctx->stream, (const Slot<1> *)m->slots, m->capacity, Table<1>{(Slot<1> *)nslots, nnb2, nlnb1, 0u, 0u}, m->d_ctr);
    else if (m->layout == LAYOUT_GRAPH)
        hipLaunchKernelGGL((k_rehash<1, Slot<1>, CSlot>), dim3(grid), dim3(BLOCK), 0, ctx->stream, (const Slot<1> *)m->slots, m->capacity, Table<1, CSlot>{(CSlot *)nslots, nnb2, nlnb1, 0u, 0u}, m->d_ctr);
    else if (ln == LAYOUT_GRAPH)
        hipLaunchKernelGGL((k_rehash<1, CSlot, Slot<1>>), dim3(grid), dim3(BLOCK), 0, ctx->stream, (const CSlot *)m->slots, m->capacity, Table<1>{(Slot<1> *)nslots, nnb2, nlnb1, 0u, 0u}, m->d_ctr);
    else
        hipLaunchKernelGGL((k_rehash<1, CSlot, CSlot>), dim3(grid), dim3(BLOCK), 0, ctx->stream, (const CSlot *)m->slots, m->capacity, Table<1, CSlot>{(CSlot *)nslots, nnb2, nlnb1, 0u, 0u}, m->d_ctr);
}

namespace gk {

static constexpr size_t POOL_MIN_BLOCK = 1u << 20;
hipError_t pool_malloc(gk_ctx *ctx, void **p, size_t bytes) {
    *p = nullptr;
    std::lock_guard<std::recursive_mutex> lock(ctx->pool_mu);
    if (bytes < POOL_MIN_BLOCK) return (hipMalloc)(p, bytes ? bytes : 1);
    // best fit, but never a block more than half again as big as asked for (a 12 GB table must not sit in a 50 GB block)
    auto it = ctx->pool_free_blocks.lower_bound(bytes);
    if (it != ctx->pool_free_blocks.end() && it->first <= bytes + bytes / 2) {
        *p = it->second;
        ctx->pool_held -= it->first;
        ctx->pool_live += it->first;
        ctx->pool_peak = std::max(ctx->pool_peak, ctx->pool_live);
        ctx->pool_free_blocks.erase(it);
        ctx->pool_hits++;
        return hipSuccess;
    }
    hipError_t e = (hipMalloc)(p, bytes);
    if (e != hipSuccess && !ctx->pool_free_blocks.empty()) {         // out of memory with blocks parked: give them back, once
        (void)hipGetLastError();
        pool_release(ctx);
        e = (hipMalloc)(p, bytes);
    }
    if (e == hipSuccess) {
        ctx->pool_sizes[*p] = bytes;
        ctx->pool_misses++;
        ctx->pool_live += bytes;
        ctx->pool_peak = std::max(ctx->pool_peak, ctx->pool_live);
    }
    return e;
}
hipError_t pool_free(gk_ctx *ctx, void *p) {
    if (!p) return hipSuccess;
    std::lock_guard<std::recursive_mutex> lock(ctx->pool_mu);
    auto it = ctx->pool_sizes.find(p);
    if (it == ctx->pool_sizes.end()) return (hipFree)(p);
    const size_t bytes = it->second;
    ctx->pool_live -= std::min(ctx->pool_live, bytes);
    // (hipFree waits for the device; a parked block may be handed out again at once, so wait for this context's work here:
    //  all three of its streams — the pipelined pieces of a batch run on the auxiliary one)
    // The copy stream only when it carries work on pooled blocks (a route of gk_dist, a striped P5): the uploads of a host-fed
    // count go to STAGING areas, which stage_reserve / gk_map_destroy wait for themselves — a table replaced between a batch's two
    // levels must not wait here for the next chunk's 0.7 GB to arrive over PCIe (12 ms of C3's host-fed count, measured).
    // (A gk_dist handle's communication stream is not waited for either, by protocol: a send or receive buffer is only replaced
    //  — by dist_grow, or through the handle's garbage list — after the exchange that used it has been consumed by its owner
    //  count, whose own stream IS waited for here; waiting for the wire in every free would stall the count beside it.)
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess && ctx->copy_stream && ctx->copy_other_pending) { e = hipStreamSynchronize(ctx->copy_stream); ctx->copy_other_pending = false; }
    if (e == hipSuccess && ctx->aux_stream) e = hipStreamSynchronize(ctx->aux_stream);
    if (e != hipSuccess || ctx->pool_held + bytes > ctx->pool_limit) {
        ctx->pool_sizes.erase(it);
        const hipError_t e2 = (hipFree)(p);
        return e != hipSuccess ? e : e2;
    }
    ctx->pool_free_blocks.emplace(bytes, p);
    ctx->pool_held += bytes;
    return hipSuccess;
}
// what sizing decisions may count on: free device memory plus what the pool has parked, or what a caller-set budget leaves
size_t mem_available(gk_ctx *ctx) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
    std::lock_guard<std::recursive_mutex> lock(ctx->pool_mu);
    size_t avail = free_b + ctx->pool_held;
    if (ctx->mem_budget) avail = std::min(avail, ctx->mem_budget > ctx->pool_live ? ctx->mem_budget - ctx->pool_live : (size_t)0);
    return avail;
}
// Load factor of a table the graph phase will read.  What follows deleteAll / the gather is read-only: 8 lookups per key of
// which ~6 MISS, and a miss in a linearly probed table walks (1 + 1/(1-load)^2)/2 slots; measured at C3 (1.28e8 keys),
// classify / unitig walk / buildGraph in ms: load 0.25 35 / 17 / 59, 0.40 44 / 21 / 70, 0.60 75 / 29 / 109, 0.75 171 / 44 / 219.
// So: as sparse as 0.25 while the table stays under a third of what is available, denser in steps when it would not — the
// table PLUS what gk_graph_build puts beside it (16.3 bytes per key in the pointer-jumping form) must fit 80 % of it.  C5's
// replica (3.1e9 63-mers in 288 GB) lands at 0.55, C4's (1.5e9 55-mers) at 0.4: DESIGN.md section 6 has the byte table.
double graph_table_load(gk_ctx *ctx, int k, uint64_t keys) {
    if (ctx->hook_graph_load_pct > 0) return ctx->hook_graph_load_pct / 100.0;       // ("graph_load_pct": A/B)
    const double avail = (double)mem_available(ctx), sb = (double)slot_bytes(words_for_k(k)), beside = 17.0 * (double)keys;
    const double steps64[] = {0.2, 0.3, 0.45}, steps[] = {0.25, 0.4, 0.55, 0.7};
    const double *st = k == 64 ? steps64 : steps;
    const int n = k == 64 ? 3 : 4;
    for (int i = 0; i < n; i++) {
        const double tb = (double)keys * sb / st[i];
        if (i == 0 ? (tb <= avail / 3.0 && tb + beside <= 0.8 * avail) : (tb + beside <= 0.8 * avail)) return st[i];
    }
    return st[n - 1];
}
void pool_release(gk_ctx *ctx) {
    std::lock_guard<std::recursive_mutex> lock(ctx->pool_mu);
    for (auto &b : ctx->pool_free_blocks) { ctx->pool_sizes.erase(b.second); (void)(hipFree)(b.second); }
    ctx->pool_free_blocks.clear();
    ctx->pool_held = 0;
}

int map_sync_counters(gk_map *m) {
    Counters *hc = reinterpret_cast<Counters *>(m->h_status);
    GK_HIP(m->ctx, hipMemcpyAsync(hc, m->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, m->ctx->stream));
    GK_HIP(m->ctx, hipStreamSynchronize(m->ctx->stream));
    const Counters c = *hc;
    m->size = c.size;
    m->occ_cached = c.occurrences;
    m->masks_valid = false;              // (every path that changes the contents ends here)
    if (c.noncanon) m->dirty = true;
    if (c.error || c.format) {
        GK_HIP(m->ctx, hipMemsetAsync(&m->d_ctr->error, 0, 2 * sizeof(u32), m->ctx->stream));
        if (c.format)
            return fail(m->ctx, GK_E_FORMAT, "a device record's length byte exceeds the declared read length (clamped; the map's contents are "
                                             "unspecified: clear it)");
        return fail(m->ctx, GK_E_CAPACITY, "a table segment filled up (internal sizing error)");
    }
    return GK_OK;
}

int stage_source(gk_ctx *ctx, const ReadSrc &src) {
    if (!src.host) return GK_OK;
    if (src.ready) { GK_HIP(ctx, hipStreamWaitEvent(ctx->stream, src.ready, 0)); return GK_OK; }      // prefetched: already on its way
    GK_HIP(ctx, hipMemcpyAsync(const_cast<uint8_t *>(src.rec), src.host, src.host_bytes, hipMemcpyHostToDevice, ctx->stream));
    return GK_OK;
}

int ctx_check_format(gk_ctx *ctx) {
    u32 f = 0;
    GK_HIP(ctx, hipMemcpyAsync(&f, ctx->d_flags, 4, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!f) return GK_OK;
    GK_HIP(ctx, hipMemsetAsync(ctx->d_flags, 0, 4, ctx->stream));
    return fail(ctx, GK_E_FORMAT, "a device record's length byte exceeds the declared read length");
}

// Load limits: grow before a batch could exceed max_load; size for target_load.  A tagged table
// (k = 64) is four interleaved sub-tables of a quarter segment each, so it runs emptier.
static inline double max_load(const gk_map *m) { return m->k == 64 ? 0.6 : 0.8; }
static inline double target_load(const gk_map *m) {
    if (m->k == 64) return 0.45;
    if (m->ctx && m->ctx->hook_target_load_pct > 0) return m->ctx->hook_target_load_pct / 100.0;        // A/B ("target_load_pct")
    return 0.65;
}

// Replace the table by one of (at least) want_slots slots.  rehash = false: the old contents are void (a deferred
// clear is pending and the caller is about to rebuild every segment from EMPTY): nothing is moved and nothing is cleared.
// The old table is only released once the new one is known to be good: a rehash that fails (a segment of the NEW table
// filled up — a sizing error) leaves the map exactly as it was.
static int map_grow_to(gk_map *m, uint64_t want_slots, bool rehash, bool keep_lnb1 = false) {
    gk_ctx *ctx = m->ctx;
    uint32_t nnb2, nlnb1;
    uint64_t ncap;
    plan_segments(m->W, want_slots, &nnb2, &nlnb1, &ncap, (uint32_t)ctx->hook_min_lnb1, keep_lnb1 ? (int)m->lnb1 : -1);
    if (rehash && m->k != 64 && nlnb1 == m->lnb1 && ctx->hook_filter_classic <= 0) {
        // same L1 fan-out: the keys move between neighbouring segments only — one streaming pass instead of a CAS per key
        bool done = false;
        if (int rc = ::streaming_rebuild(m, nnb2, nlnb1, ncap, INT32_MIN, m->layout, &done)) return rc;
        if (done) { m->grows++; return GK_OK; }
    }
    void *nslots = nullptr;
    if (!rehash) {
        hipError_t e = hipMalloc(&nslots, ncap * map_slot_bytes(m));
        if (e != hipSuccess) { (void)hipGetLastError(); return fail(ctx, GK_E_CAPACITY, "cannot grow table to " + std::to_string(ncap) + " slots: " + hipGetErrorString(e)); }
    } else {
        int rc = alloc_table(ctx, m->W, m->layout, ncap, &nslots);
        if (rc) { if (nslots) (void)hipFree(nslots); return fail(ctx, GK_E_CAPACITY, "cannot grow table to " + std::to_string(ncap) + " slots: " + ctx->err); }
        launch_rehash(m, m->layout, nslots, nnb2, nlnb1);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { (void)hipFree(nslots); return hip_fail(ctx, e, "table rehash"); }
        if (int rc2 = map_sync_counters(m)) { (void)hipFree(nslots); return rc2; }      // rehash error flag: the old table stays
    }
    GK_HIP(ctx, hipFree(m->slots));
    m->slots = nslots;
    m->capacity = ncap;
    m->nb2 = nnb2;
    m->lnb1 = nlnb1;
    m->tombstones = 0;
    m->grows++;
    return GK_OK;
}

// ArrayDNAMap.rescale analogue: make room for `extra_keys` more distinct keys.
int map_reserve(gk_map *m, uint64_t extra_keys) {
    uint64_t need = m->size + m->tombstones + extra_keys;
    if ((double)need <= max_load(m) * (double)m->capacity) return GK_OK;
    return map_grow_to(m, std::max<uint64_t>((uint64_t)((double)(m->size + extra_keys) / target_load(m)) + 1, m->capacity + m->capacity / 2), true);
}

// the partitioned pipeline's mid-batch form: room for `new_distinct` more keys; from_empty = the table's contents are void
int map_make_room(gk_map *m, uint64_t new_distinct, uint64_t size_for, bool from_empty, bool keep_lnb1) {
    const uint64_t have = from_empty ? 0 : m->size + m->tombstones;
    // (a table whose contents are void is replaced for free — nothing to rehash — so it is sized for what the whole call
    //  is expected to bring, `size_for`, right away: the second count over a map that had been compacted in between kept
    //  the small table for two batches and then paid a 20 ms rehash of 4e8 keys in the third)
    const uint64_t fit_now = from_empty ? std::max(new_distinct, size_for) : new_distinct;
    if ((double)(have + fit_now) <= max_load(m) * (double)m->capacity) return GK_OK;
    const uint64_t want = std::max<uint64_t>((uint64_t)((double)(have + std::max(new_distinct, size_for)) / target_load(m)) + 1,
                                             from_empty ? 0 : m->capacity + m->capacity / 2);
    return map_grow_to(m, want, !from_empty, keep_lnb1);
}

static constexpr uint64_t SAMPLE_SLOTS = 1ull << 23;      // 64 MB: 4 M sample keys at load 0.5 = 4.3e9 distinct keys
int map_ensure_sample(gk_map *m) {
    if (m->d_sample) return GK_OK;
    gk_ctx *ctx = m->ctx;
    GK_HIP(ctx, hipMalloc((void **)&m->d_sample, SAMPLE_SLOTS * 8));
    GK_HIP(ctx, hipMemsetAsync(m->d_sample, 0xff, SAMPLE_SLOTS * 8, ctx->stream));
    m->sample_mask = SAMPLE_SLOTS - 1;
    return GK_OK;
}

int map_materialize(gk_map *m) {
    if (!m->pending_clear) return GK_OK;
    gk_ctx *ctx = m->ctx;
    launch_clear(ctx, m->W, m->layout, m->slots, m->capacity);
    GK_HIP(ctx, hipGetLastError());
    m->pending_clear = false;
    return GK_OK;
}

}  // namespace gk

static int check_map_lazy(const gk_map *m) {       // does not touch the slots
    if (!m || !m->ctx) return fail(nullptr, GK_E_INVALID, "null map handle");
    hipError_t e = hipSetDevice(m->ctx->device);
    if (e != hipSuccess) return hip_fail(m->ctx, e, "hipSetDevice");
    return GK_OK;
}
static int check_map(gk_map *m) {                  // slots must be valid afterwards
    if (int rc = check_map_lazy(m)) return rc;
    return map_materialize(m);
}

extern "C" __attribute__((weak)) void gk_testhooks_env(gk_ctx *ctx);      // gk_testhooks.hip; absent from the product library

extern "C" {

int gk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int gk_ctx_create(int device, gk_ctx **out) {
    if (!out) return fail(nullptr, GK_E_INVALID, "gk_ctx_create: out is NULL");
    *out = nullptr;
    int n = gk_device_count();
    if (n <= 0) return fail(nullptr, GK_E_NODEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail(nullptr, GK_E_INVALID, "device index out of range");
    gk_ctx *ctx = new gk_ctx();
    ctx->device = device;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev1);
    for (int i = 0; i < 6 && e == hipSuccess; i++) e = hipEventCreate(&ctx->pev[i]);
    if (e == hipSuccess) e = hipEventCreate(&ctx->gev);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->gev2, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess) {      // lowest priority: what runs there (P4 of the pieces that have landed) fills gaps and must not delay the main stream's scatters
        int least = 0, greatest = 0;
        e = hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&ctx->aux_stream, hipStreamNonBlocking, least);
    }
    for (int i = 0; i < 16 && e == hipSuccess; i++) e = hipEventCreateWithFlags(&ctx->cev[i], hipEventDisableTiming);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_flags, 64);
    if (e == hipSuccess) e = hipMemset(ctx->d_flags, 0, 64);
    if (e != hipSuccess) {
        int rc = hip_fail(nullptr, e, "gk_ctx_create");
        delete ctx;
        return rc;
    }
    ctx->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ctx->pool_limit = (size_t)prop.totalGlobalMem / 3;
    // Test / A-B switches exist only in the TEST build of the library (libgenome_amd_test.so = this library + gk_testhooks.o):
    // there the environment is read HERE, once, and gk_ctx_set_option is the programmatic form.  The product library has
    // neither: nothing in a host JVM's environment can change its behaviour.
    if (gk_testhooks_env) gk_testhooks_env(ctx);
    *out = ctx;
    return GK_OK;
}

void gk_ctx_destroy(gk_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    gk::pool_release(ctx);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->skm_counts) (void)hipFree(ctx->skm_counts);
    if (ctx->d_flags) (void)hipFree(ctx->d_flags);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    for (int i = 0; i < 6; i++) if (ctx->pev[i]) (void)hipEventDestroy(ctx->pev[i]);
    if (ctx->gev) (void)hipEventDestroy(ctx->gev);
    if (ctx->gev2) (void)hipEventDestroy(ctx->gev2);
    for (int i = 0; i < 16; i++) if (ctx->cev[i]) (void)hipEventDestroy(ctx->cev[i]);
    if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
    if (ctx->aux_stream) { (void)hipStreamSynchronize(ctx->aux_stream); (void)hipStreamDestroy(ctx->aux_stream); }
    delete ctx;
}

int gk_ctx_trim(gk_ctx *ctx) {
    if (!ctx) return fail(nullptr, GK_E_INVALID, "gk_ctx_trim: null context");
    GK_HIP(ctx, hipSetDevice(ctx->device));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
    if (ctx->aux_stream) GK_HIP(ctx, hipStreamSynchronize(ctx->aux_stream));
    gk::pool_release(ctx);
    return GK_OK;
}

int gk_ctx_mem_stats(gk_ctx *ctx, uint64_t *live, uint64_t *peak, uint64_t *parked, int reset_peak) {
    if (!ctx) return fail(nullptr, GK_E_INVALID, "gk_ctx_mem_stats: null context");
    std::lock_guard<std::recursive_mutex> lock(ctx->pool_mu);
    if (live) *live = ctx->pool_live;
    if (peak) *peak = ctx->pool_peak;
    if (parked) *parked = ctx->pool_held;
    if (reset_peak) ctx->pool_peak = ctx->pool_live;
    return GK_OK;
}

int gk_ctx_set_mem_budget(gk_ctx *ctx, uint64_t bytes) {
    if (!ctx) return fail(nullptr, GK_E_INVALID, "gk_ctx_set_mem_budget: null context");
    std::lock_guard<std::recursive_mutex> lock(ctx->pool_mu);
    ctx->mem_budget = (size_t)bytes;
    return GK_OK;
}

const char *gk_last_error(const gk_ctx *ctx) { return ctx ? ctx->err.c_str() : tls_err.c_str(); }
int gk_ctx_device(const gk_ctx *ctx) { return ctx ? ctx->device : -1; }

int gk_ctx_sync(gk_ctx *ctx) {
    if (!ctx) return fail(nullptr, GK_E_INVALID, "null ctx");
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GK_OK;
}
// The pages go to the NUMA node next to the GPU: on a two-socket host a pinned buffer that lands on the far socket uploads
// at half the rate (26 instead of 56 GB/s measured; where the default allocation lands varies from process to process).
// set_mempolicy(MPOL_PREFERRED) for the duration of the allocation; if the policy call is refused (seccomp) or the runtime
// does not know the node, the default placement is taken.
int gk_host_alloc(gk_ctx *ctx, size_t nbytes, void **host_ptr) {
    if (!ctx || !host_ptr) return fail(ctx, GK_E_INVALID, "gk_host_alloc: null argument");
    GK_HIP(ctx, hipSetDevice(ctx->device));
    int node = -1;
    if (hipDeviceGetAttribute(&node, hipDeviceAttributeHostNumaId, ctx->device) != hipSuccess) { (void)hipGetLastError(); node = -1; }
    bool bound = false;
    if (node >= 0 && node < 1024) {
        unsigned long mask[16] = {0};
        mask[node / 64] |= 1ul << (node % 64);
        bound = syscall(SYS_set_mempolicy, 1 /* MPOL_PREFERRED */, mask, 1025ul) == 0;
    }
    const hipError_t e = hipHostMalloc(host_ptr, nbytes ? nbytes : 1, bound ? hipHostMallocNumaUser : hipHostMallocDefault);
    if (bound) (void)syscall(SYS_set_mempolicy, 0 /* MPOL_DEFAULT */, nullptr, 0ul);
    GK_HIP(ctx, e);
    return GK_OK;
}
int gk_host_free(gk_ctx *ctx, void *host_ptr) {
    if (!ctx) return fail(ctx, GK_E_INVALID, "null ctx");
    if (host_ptr) GK_HIP(ctx, hipHostFree(host_ptr));
    return GK_OK;
}
int gk_host_register(gk_ctx *ctx, void *host_ptr, size_t nbytes) {
    if (!ctx || !host_ptr) return fail(ctx, GK_E_INVALID, "gk_host_register: null argument");
    GK_HIP(ctx, hipSetDevice(ctx->device));
    GK_HIP(ctx, hipHostRegister(host_ptr, nbytes, hipHostRegisterDefault));
    return GK_OK;
}
int gk_host_unregister(gk_ctx *ctx, void *host_ptr) {
    if (!ctx || !host_ptr) return fail(ctx, GK_E_INVALID, "gk_host_unregister: null argument");
    GK_HIP(ctx, hipHostUnregister(host_ptr));
    return GK_OK;
}
int gk_dev_alloc(gk_ctx *ctx, size_t nbytes, void **dev_ptr) {
    if (!ctx || !dev_ptr) return fail(ctx, GK_E_INVALID, "gk_dev_alloc: null argument");
    GK_HIP(ctx, hipSetDevice(ctx->device));
    GK_HIP(ctx, hipMalloc(dev_ptr, nbytes ? nbytes : 1));
    return GK_OK;
}
int gk_dev_free(gk_ctx *ctx, void *dev_ptr) {
    if (!ctx) return fail(ctx, GK_E_INVALID, "null ctx");
    if (dev_ptr) GK_HIP(ctx, hipFree(dev_ptr));
    return GK_OK;
}
int gk_dev_upload(gk_ctx *ctx, void *dev_dst, const void *host_src, size_t nbytes) {
    if (!ctx) return fail(ctx, GK_E_INVALID, "null ctx");
    GK_HIP(ctx, hipMemcpyAsync(dev_dst, host_src, nbytes, hipMemcpyHostToDevice, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GK_OK;
}
int gk_dev_download(gk_ctx *ctx, void *host_dst, const void *dev_src, size_t nbytes) {
    if (!ctx) return fail(ctx, GK_E_INVALID, "null ctx");
    GK_HIP(ctx, hipMemcpyAsync(host_dst, dev_src, nbytes, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GK_OK;
}

// ---- what this card streams: the yardstick SURVEY.md §8(d) asks for next to the nominal peak -----------------------------
// copy (one 16-byte load + one 16-byte store per lane and step), fill (stores only), sum (loads only); grid-stride,
// 256 threads, 8 workgroups per CU — the shape of the library's own streaming kernels.
__global__ __launch_bounds__(256) void k_stream_copy(const uint4 *__restrict__ in, uint4 *__restrict__ out, u64 n) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) out[i] = in[i];
}
__global__ __launch_bounds__(256) void k_stream_fill(uint4 *__restrict__ out, u64 n, u32 v) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) out[i] = make_uint4(v, v, v, v);
}
__global__ __launch_bounds__(256) void k_stream_sum(const uint4 *__restrict__ in, u64 n, u32 *sink) {
    u32 acc = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        const uint4 v = in[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9e3779b9u) *sink = acc;       // (keeps the loads alive; practically never taken)
}

int gk_dev_stream_bench(gk_ctx *ctx, size_t nbytes, int reps, double *gbps3) {
    if (!ctx || !gbps3) return fail(ctx, GK_E_INVALID, "gk_dev_stream_bench: null argument");
    if (nbytes < (1u << 20) || reps < 1 || reps > 1000) return fail(ctx, GK_E_INVALID, "gk_dev_stream_bench: nbytes >= 1 MiB, 1 <= reps <= 1000");
    GK_HIP(ctx, hipSetDevice(ctx->device));
    const u64 n = nbytes / 16;
    uint4 *a = nullptr, *b = nullptr;
    GK_HIP(ctx, hipMalloc((void **)&a, n * 16));
    if (hipError_t e = hipMalloc((void **)&b, n * 16); e != hipSuccess) { (void)hipFree(a); return hip_fail(ctx, e, "gk_dev_stream_bench"); }
    const int grid = (int)std::min<u64>((n + 255) / 256, (u64)ctx->cu_count * 8);
    auto timed = [&](int which, double bytes_per_rep, double *out) -> int {
        for (int r = -1; r < reps; r++) {          // (one untimed launch first)
            if (r == 0) GK_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
            if (which == 0) hipLaunchKernelGGL(k_stream_copy, dim3(grid), dim3(256), 0, ctx->stream, a, b, n);
            else if (which == 1) hipLaunchKernelGGL(k_stream_fill, dim3(grid), dim3(256), 0, ctx->stream, b, n, 0xffffffffu);
            else hipLaunchKernelGGL(k_stream_sum, dim3(grid), dim3(256), 0, ctx->stream, a, n, ctx->d_flags + 8);
        }
        GK_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        GK_HIP(ctx, hipEventSynchronize(ctx->ev1));
        float ms = 0;
        GK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        *out = ms > 0 ? bytes_per_rep * reps / (ms * 1e-3) / 1e9 : 0.0;
        return GK_OK;
    };
    int rc = GK_OK;
    hipLaunchKernelGGL(k_stream_fill, dim3(grid), dim3(256), 0, ctx->stream, a, n, 0x12345678u);
    if (!rc) rc = timed(0, 32.0 * (double)n, &gbps3[0]);
    if (!rc) rc = timed(1, 16.0 * (double)n, &gbps3[1]);
    if (!rc) rc = timed(2, 16.0 * (double)n, &gbps3[2]);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(a); (void)hipFree(b);
    return rc;
}

int gk_map_create(gk_ctx *ctx, int k, uint64_t capacity_hint, gk_map **out) {
    if (!ctx || !out) return fail(ctx, GK_E_INVALID, "gk_map_create: null argument");
    *out = nullptr;
    if (!k_supported(k))
        return fail(ctx, GK_E_UNSUPPORTED_K,
                    "k=" + std::to_string(k) + " unsupported (2..31 and 34..64; k=32,33 are broken in the reference)");
    GK_HIP(ctx, hipSetDevice(ctx->device));
    gk_map *m = new gk_map();
    m->ctx = ctx;
    m->k = k;
    m->W = words_for_k(k);
    uint64_t want = capacity_hint ? capacity_hint : 1024;
    plan_segments(m->W, (uint64_t)((double)want / target_load(m)) + 1, &m->nb2, &m->lnb1, &m->capacity, (uint32_t)ctx->hook_min_lnb1);
    // The table is NOT cleared here: a new map starts like one after gk_map_clear — contents void, the clear deferred — so
    // that the first partitioned insert builds every segment from EMPTY without reading it (a 98 GB table: 21 ms of clearing
    // saved, and its first batch takes the pipeline instead of paying the table in and out); anything else materialises it.
    int rc = GK_OK;
    {
        m->layout = LAYOUT_COUNT;        // (8-byte keys: 12-byte count slots; the graph phase's 16-byte layout is what deleteAll / the gather build)
        hipError_t ea = hipMalloc(&m->slots, m->capacity * map_slot_bytes(m));
        if (ea != hipSuccess) rc = hip_fail(ctx, ea, "gk_map_create: table");
        m->pending_clear = true;
    }
    if (rc == GK_OK) {
        hipError_t e = hipMalloc((void **)&m->d_ctr, sizeof(Counters));
        if (e == hipSuccess) e = hipHostMalloc((void **)&m->h_status, 256, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMemsetAsync(m->d_ctr, 0, sizeof(Counters), ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = hip_fail(ctx, e, "gk_map_create");
    }
    if (rc != GK_OK) {
        if (m->slots) (void)hipFree(m->slots);
        if (m->d_ctr) (void)hipFree(m->d_ctr);
        if (m->h_status) (void)hipHostFree(m->h_status);
        delete m;
        return rc == GK_E_HIP ? fail(ctx, GK_E_CAPACITY, "cannot allocate table: " + ctx->err) : rc;
    }
    *out = m;
    return GK_OK;
}

int gk_map_create_for_graph(gk_ctx *ctx, int k, uint64_t keys, gk_map **out) {
    if (!ctx || !out) return fail(ctx, GK_E_INVALID, "gk_map_create_for_graph: null argument");
    *out = nullptr;
    if (!k_supported(k)) return fail(ctx, GK_E_UNSUPPORTED_K, "k=" + std::to_string(k) + " unsupported (2..31 and 34..64)");
    GK_HIP(ctx, hipSetDevice(ctx->device));
    return map_create_for_graph(ctx, k, keys, out);
}

void gk_map_destroy(gk_map *m) {
    if (!m) return;
    gk_ctx *ctx = m->ctx;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    if (m->slots) (void)hipFree(m->slots);
    if (m->d_ctr) (void)hipFree(m->d_ctr);
    if (m->h_status) (void)hipHostFree(m->h_status);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);       // (a prefetch may be on its way)
    for (int i = 0; i < 2; i++) { if (m->stage[i].d) (void)hipFree(m->stage[i].d); if (m->stage[i].ev) (void)hipEventDestroy(m->stage[i].ev); }
    if (m->d_offsets) (void)hipFree(m->d_offsets);
    if (m->d_sample) (void)hipFree(m->d_sample);
    if (m->d_scratch) (void)hipFree(m->d_scratch);
    part_scratch_free(ctx, m->part);
    delete m;
}

int gk_map_k(const gk_map *m) { return m ? m->k : 0; }

int gk_map_set_insert_path(gk_map *m, int path) {
    if (int rc = check_map_lazy(m)) return rc;
    if (path < 0 || path > 2) return fail(m->ctx, GK_E_INVALID, "insert path must be 0 (auto), 1 (direct) or 2 (partitioned)");
    if (path == 2 && !part_supported(m)) return fail(m->ctx, GK_E_INVALID, "table too large for the partitioned path");
    m->insert_path = path;
    return GK_OK;
}

int gk_map_clear(gk_map *m) {
    if (int rc = check_map_lazy(m)) return rc;
    gk_ctx *ctx = m->ctx;
    // Deferred: the next partitioned insert rebuilds every segment from EMPTY without reading it;
    // anything else materialises the clear first (map_materialize).
    GK_HIP(ctx, hipMemsetAsync(m->d_ctr, 0, sizeof(Counters), ctx->stream));     // stream-ordered: no host round trip needed
    if (m->d_sample && m->sample_dirty) GK_HIP(ctx, hipMemsetAsync(m->d_sample, 0xff, (m->sample_mask + 1) * 8, ctx->stream));
    m->sample_claims_seen = 0;
    m->sample_dirty = false;
    m->est_distinct_last = 0;
    m->pending_clear = true;
    m->layout = LAYOUT_COUNT;    // (the contents are void: the same allocation is read as 12-byte count slots from here on — same slot count, fewer bytes)
    m->size = 0;
    m->tombstones = 0;
    m->total_occurrences = 0;
    m->dirty = false;
    m->masks_valid = false;
    return GK_OK;
}

int gk_map_size(gk_map *m, uint64_t *n) {
    if (int rc = check_map_lazy(m)) return rc;
    if (!n) return fail(m->ctx, GK_E_INVALID, "gk_map_size: n is NULL");
    *n = m->size;
    return GK_OK;
}

int gk_map_slots(gk_map *m, uint64_t *slots) {
    if (int rc = check_map_lazy(m)) return rc;
    if (!slots) return fail(m->ctx, GK_E_INVALID, "gk_map_slots: slots is NULL");
    *slots = m->capacity;
    return GK_OK;
}

static int add_keys_dev(gk_map *m, const u64 *d_keys, const i32 *d_counts, u64 n, bool verbatim = false);

// launch k_count_reads over device-resident records; accumulates event time into last_count_ms
static int launch_count(gk_map *m, const ReadSrc &src) {
    gk_ctx *ctx = m->ctx;
    u64 ntiles = (src.nreads + TILE_READS - 1) / TILE_READS;
    int grid = (int)std::min<u64>(std::max<u64>(ntiles, 1), (u64)ctx->cu_count * 8);
    if (int rc = stage_source(ctx, src)) return rc;
    GK_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    GK_BY_SLOT(m, hipLaunchKernelGGL((k_count_reads<W, S>), dim3(grid), dim3(BLOCK), 0, ctx->stream, src.rec, src.nreads, src.off, src.stride, m->k, src.group,
                                     src.max_len, table_of<W, S>(m), m->d_ctr));
    GK_HIP(ctx, hipGetLastError());
    GK_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    if (int rc = map_sync_counters(m)) return rc;
    float ms = 0.f;
    GK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    m->last_count_ms += ms;
    m->phase_ms[0] += ms;
    m->direct_launches++;
    return GK_OK;
}

// bytes of `.bin` records one staging area holds at most: what one partitioned batch of 2^31 windows takes (150 bp reads, k = 31:
// 17.9 M records of 39 bytes) — a host-fed count then cuts its stream where the device-resident one cuts its batches
static constexpr size_t GK_MAX_STAGE_DEFAULT = 704u << 20;
static inline size_t max_stage(const gk_ctx *ctx) { return ctx->hook_max_stage > 0 ? (size_t)ctx->hook_max_stage : GK_MAX_STAGE_DEFAULT; }
static int reset_occ_counter(gk_map *m) {
    GK_HIP(m->ctx, hipMemsetAsync(&m->d_ctr->occurrences, 0, sizeof(unsigned long long), m->ctx->stream));
    m->occ_cached = 0;
    return GK_OK;
}
// Every path that counts windows (launch_count, launch_partitioned, add_keys_dev) ends with map_sync_counters, which
// brings `occurrences` back with the rest of the counters: no further device read here.
static int read_occ_counter(gk_map *m, uint64_t *occ) {
    *occ = m->occ_cached;
    return GK_OK;
}

// ---- which path inserts a batch: a TIME model with per-path coefficients measured on MI355X ----------------------
// (picoseconds of device time per k-mer occurrence and per key word, chip-wide; sources under profiles/.  The
//  coefficients are measurements, the model is only their sum — re-measure with scripts/measure_paths.py.)
struct PathCost {
    double direct_ps = 45.0;        // k_count_reads, one global CAS/add per window: 5.45 ms / 1.2e8 windows (C2-U); 40 ps at C3's ~1000x repeats
    double p2_ps = 5.0;             // k_op_scatter1_reads: 0.60 ms / 1.2e8 (x W for 16-byte keys: 0.97 ms / 9.6e7 at k = 55)
    double p3_ps = 2.3;             // k_part_hist2r + scans (exact fine level only): one more read of the keys
    double p4_ps = 3.9;             // k_part_scatter2 at few fine buckets (0.55 ms / 1.2e8 at nb2 = 370 with the next chunk's keys prefetched) ...
    double p4_ps_per_nb2 = 0.0018;  // ... + what each fine bucket per L1 bucket adds (0.57 / 0.65 / 0.80 / 1.2 ms per 1.2e8 at nb2 = 93 / 370 / 1479 / 2958)
    double p5_ps = 1.7;             // k_seg_insert: its key stream and LDS inserts (its table traffic is priced below)
    double stream_tbps = 4.6;       // k_seg_insert writes / reads the table at 4.6 TB/s; k_clear reaches 4.95
    double fixed_us = 60.0;         // launches, the small scans, one or two host round trips
};
// The coefficients above were measured on ONE MI355X (round 1-2 profiles).  What differs from card to card (another HBM stack
// speed, a power cap, another part of the family) is taken from TWO numbers measured on the context's own device the first time
// a path has to be chosen: its streaming copy rate (the box of the profiles: 4.85 TB/s) and its random 64-bit CAS rate (17.6 G/s).
// The streaming phases' prices scale with the first, the direct path's with the second; ~3 ms, once per context.
__global__ __launch_bounds__(256) void k_calib_cas(u64 *arr, u64 mask, u64 n) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) (void)cas64(&arr[mix64(i) & mask], ~0ull, i);
}
static const PathCost &path_cost(gk_ctx *ctx) {
    static const PathCost REF;
    if (ctx->cost_state == 1) return *reinterpret_cast<const PathCost *>(ctx->cost_blob);
    if (ctx->cost_state == 2) return REF;
    static_assert(sizeof(PathCost) <= sizeof(ctx->cost_blob), "cost blob");
    ctx->cost_state = 2;                                           // (whatever happens below, measure once)
    constexpr u64 N16 = (64u << 20) / 16, NCAS = 1u << 23;           // 64 MiB copied, 8 M CAS into a 64 MiB table
    uint4 *a = nullptr, *b = nullptr;
    if (hipMalloc((void **)&a, N16 * 16) != hipSuccess || hipMalloc((void **)&b, N16 * 16) != hipSuccess) { (void)hipGetLastError(); if (a) (void)hipFree(a); return REF; }
    const int grid = ctx->cu_count * 8;
    float ms_copy = 0, ms_cas = 0;
    hipError_t e = hipMemsetAsync(a, 0xff, N16 * 16, ctx->stream);
    for (int r = 0; r < 3 && e == hipSuccess; r++) {               // the third launch is the timed one
        if (r == 2) e = hipEventRecord(ctx->ev0, ctx->stream);
        hipLaunchKernelGGL(k_stream_copy, dim3(grid), dim3(256), 0, ctx->stream, a, b, N16);
    }
    if (e == hipSuccess) e = hipEventRecord(ctx->ev1, ctx->stream);
    if (e == hipSuccess) e = hipEventSynchronize(ctx->ev1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms_copy, ctx->ev0, ctx->ev1);
    if (e == hipSuccess) e = hipEventRecord(ctx->ev0, ctx->stream);
    if (e == hipSuccess) hipLaunchKernelGGL(k_calib_cas, dim3(grid), dim3(256), 0, ctx->stream, (u64 *)a, (u64)(N16 * 2 - 1), (u64)NCAS);
    if (e == hipSuccess) e = hipEventRecord(ctx->ev1, ctx->stream);
    if (e == hipSuccess) e = hipEventSynchronize(ctx->ev1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms_cas, ctx->ev0, ctx->ev1);
    (void)hipFree(a); (void)hipFree(b);
    if (e != hipSuccess || ms_copy <= 0 || ms_cas <= 0) { (void)hipGetLastError(); return REF; }
    const double copy_tbps = 2.0 * N16 * 16 / (ms_copy * 1e-3) / 1e12, cas_gps = NCAS / (ms_cas * 1e-3) / 1e9;
    // (a 64 MiB working set partly lives in the 256 MB Infinity Cache: the RATIOS to the reference box are what is used, and they are
    //  clamped — a calibration that is off by more than 2x says more about the moment it ran than about the card)
    const double fs = std::min(2.0, std::max(0.5, copy_tbps / ctx->ref_copy_tbps)), fa = std::min(2.0, std::max(0.5, cas_gps / ctx->ref_cas_gps));
    PathCost c = REF;
    c.stream_tbps *= fs;
    c.p2_ps /= fs; c.p3_ps /= fs; c.p4_ps /= fs; c.p4_ps_per_nb2 /= fs; c.p5_ps /= fs;
    c.direct_ps /= fa;
    memcpy(ctx->cost_blob, &c, sizeof(c));
    ctx->measured_copy_tbps = copy_tbps; ctx->measured_cas_gps = cas_gps;
    ctx->cost_state = 1;
    return *reinterpret_cast<const PathCost *>(ctx->cost_blob);
}
static bool use_partitioned(const gk_map *m, u64 occ) {
    if (m->insert_path == 1 || !part_supported(m)) return false;
    if (m->insert_path == 2) return true;
    if (m->skewed) return false;          // this map's data has already defeated even the L1 regions once (one k-mer, millions of times)
    const PathCost &COST = path_cost(m->ctx);
    const double tb = (double)m->capacity * (double)map_slot_bytes(m);
    const double pass_us = tb / (COST.stream_tbps * 1e6);
    const double us_direct = (double)occ * COST.direct_ps * 1e-6 + (m->pending_clear ? pass_us : 0.0);
    const double per_key = (COST.p2_ps + COST.p4_ps + COST.p4_ps_per_nb2 * m->nb2 + COST.p5_ps + (m->repeats || !m->pending_clear ? COST.p3_ps : 0.0)) * m->W;
    const double us_part = (double)occ * per_key * 1e-6 + pass_us * (m->pending_clear ? 1.0 : 2.0) + COST.fixed_us;
    return us_part < us_direct;
}

// partitioned launch over device records or device keys, timed with the same events as launch_count
static int launch_partitioned(gk_map *m, const ReadSrc &src, const u64 *d_keys, u64 nkeys_in, u64 bound, const PartPlan &plan) {
    gk_ctx *ctx = m->ctx;
    GK_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    const bool from_empty = m->pending_clear;
    m->pending_clear = false;
    const int prc = part_count(m, &m->part, src, d_keys, nkeys_in, bound, from_empty, plan);
    if (prc < 0) {
        if (from_empty) { m->pending_clear = true; m->size = 0; }       // a half-built table is void: the map is empty again
        return prc;
    }
    if (prc == PART_NOT_UNIFORM) { m->pending_clear = from_empty; return prc; }     // nothing but scratch (or a table that was being rebuilt from empty) was touched
    if (prc == PART_RETRY_DIRECT || prc == PART_OUTGREW) {        // extreme skew / outgrown fan-out: nothing but scratch (or a table that was being rebuilt from empty) was touched
        m->pending_clear = from_empty;
        if (prc == PART_RETRY_DIRECT) m->skewed = true;
        if (from_empty) m->size = 0;
        if (int rc = map_materialize(m)) return rc;
        if (int rc = map_reserve(m, bound)) return rc;
        if (src.rec) return launch_count(m, src);
        return add_keys_dev(m, d_keys, nullptr, nkeys_in, plan.check_canon);
    }
    GK_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    GK_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    GK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    m->last_count_ms += ms;
    for (int i = 0; i < 5; i++) {
        float pm = 0.f;
        GK_HIP(ctx, hipEventElapsedTime(&pm, ctx->pev[i], ctx->pev[i + 1]));
        m->phase_ms[i] += pm;
    }
    {
        float gm = 0.f;
        GK_HIP(ctx, hipEventElapsedTime(&gm, ctx->pev[2], ctx->gev));
        m->gap_ms += gm;
    }
    m->part_launches++;
    return GK_OK;
}

static double map_room(const gk_map *m) {
    return max_load(m) * (double)m->capacity - (m->pending_clear ? 0.0 : (double)(m->size + m->tombstones));
}
// how many reads (of nk windows each) may go into one launch without risking the load limit
static u64 reads_per_launch(gk_map *m, u64 nk) {
    if (nk == 0) return ~0ull;
    double room = map_room(m);
    u64 floor_occ = std::max<u64>(1ull << 24, m->capacity / 4);
    u64 occ = room > (double)floor_occ ? (u64)room : floor_occ;
    return std::max<u64>(1, occ / nk);
}
// how many windows one partitioned batch may hold: bounded by the scratch it needs (two key buffers, the spill list)
// next to what is free in HBM — half of it at most — and by 64 GiB of keys per buffer (gk_map_set_max_batch_keys changes
// that).  Big batches pay: every batch after the first streams the whole table in and out again
static u64 part_batch_keys(gk_map *m) {
    // (default: key buffers of up to 64 GiB, or scratch as large as the table itself, whichever is more — and never more than
    //  half of what is free, below.  Every further batch of a call streams the whole table in and out again: C3's 6e9 windows
    //  in three batches of 2^31 (16 GiB buffers, the default until the end of round 3) 103.3 ms, in one batch 98.5
    //  (profiles/r03/c3_count_vs_batch_size.txt); k = 55 into a 74 GB table 101 -> 70 ms (big_table_74GB_k55.json).  The card
    //  has 288 GB to be used; allocating is not what costs (alloc_cost_by_size.txt), and a caller that shares the card says so
    //  with gk_ctx_set_mem_budget or gk_map_set_max_batch_keys.)
    const double per_key = 8.0 * m->W * (1.125 + 1.0 + 1.0 / 16) + 1.0;      // bufA + bufB + spill (+ range matrix, bounded above)
    u64 cap = m->max_batch_keys ? m->max_batch_keys : (1ull << 33) / (u64)m->W;         // 64 GiB of keys per buffer
    if (!m->max_batch_keys) cap = std::max<u64>(cap, (u64)((double)m->capacity * (double)map_slot_bytes(m) / per_key));
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        // the scratch a previous batch left allocated is reused, so it counts as available: approximate by a share of the whole card
        // (a caller-set memory budget replaces the card's size)
        const double whole = m->ctx->mem_budget ? (double)m->ctx->mem_budget : (double)total_b;
        const double avail = std::max((double)mem_available(m->ctx), 0.35 * whole);
        cap = std::min<u64>(cap, (u64)(0.5 * avail / per_key));
    }
    return std::max<u64>(cap, 1ull << 20);
}

// One batch of windows (device records) or keys: pick the path, make room, insert.  `bound` = the batch's windows.
// A batch that FITS the table's room even if every window were a new key goes in as before.  One that does not is, on
// sequencing data, mostly repeats: the partitioned pipeline then runs with the distinct-key sample (PartPlan::estimate)
// and sizes the table between its two levels for what the batch really brings.
static int insert_batch(gk_map *m, const ReadSrc &src, const u64 *d_keys, u64 nkeys_in, u64 bound, bool verbatim = false, double grow_ahead = 1.0) {
    gk_ctx *ctx = m->ctx;
    if (bound && use_partitioned(m, bound)) {
        PartPlan plan;
        plan.grow_ahead = grow_ahead;
        plan.fine_exact = ctx->hook_fine_exact >= 0 ? ctx->hook_fine_exact != 0 : m->repeats;
        plan.check_canon = verbatim;
        const bool fits = (double)bound <= map_room(m);
        if (!fits) {
            plan.estimate = true;
            // the L1 bucket of a key must survive a mid-batch growth: only tables of >= 256 segments (lnb1 = 8) keep theirs
            if (m->lnb1 < 8 && !ctx->hook_no_reserve) {
                const u64 want = (u64)(256.0 * (double)(1ull << seg_bits_for(m->W)) * target_load(m));
                const u64 have = m->pending_clear ? 0 : m->size + m->tombstones;
                const u64 more = std::max<u64>(want, have) - have + 1;
                if (int rc = map_make_room(m, more, more, m->pending_clear)) return rc;
            }
            if (m->lnb1 < 8) plan.estimate = false;
            if (!plan.estimate && !m->pending_clear && !ctx->hook_no_reserve) { if (int rc = map_reserve(m, bound)) return rc; }
        }
        return launch_partitioned(m, src, d_keys, nkeys_in, bound, plan);
    }
    if (int rc = map_materialize(m)) return rc;
    if (int rc = map_reserve(m, bound)) return rc;
    if (src.rec) return launch_count(m, src);
    GK_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    if (int rc = add_keys_dev(m, d_keys, nullptr, nkeys_in, verbatim)) return rc;
    GK_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    GK_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    GK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    m->last_count_ms += ms;
    m->direct_launches++;
    return GK_OK;
}

static void reset_call_stats(gk_map *m) {
    m->gap_ms = 0.f;
    m->last_count_ms = 0.f;
    m->last_count_occ = 0;
    for (float &x : m->phase_ms) x = 0.f;
}

// fixed-stride device records, nk_max windows per record at most: cut into batches and insert
static int count_fixed_records(gk_map *m, ReadSrc src, u64 nk, u64 windows_total /* exact if known, else 0 */) {
    const uint8_t *rec = src.rec;
    const u64 nreads = src.nreads;
    u64 done = 0;
    while (done < nreads) {
        const u64 left = nreads - done;
        // candidate: everything that is left as ONE partitioned batch (bounded by its scratch); the direct path
        // instead takes what the table has room for
        u64 chunk = std::min<u64>(left, std::max<u64>(1, part_batch_keys(m) / std::max<u64>(nk, 1)));
        u64 bound = windows_total && chunk == nreads ? windows_total : chunk * nk;
        if (!(nk && use_partitioned(m, bound))) {
            chunk = std::min(left, reads_per_launch(m, nk));
            bound = windows_total && chunk == nreads ? windows_total : chunk * nk;
        }
        src.rec = rec + done * src.stride;
        src.nreads = chunk;
        if (int rc = insert_batch(m, src, nullptr, 0, bound, false, (double)left / (double)chunk)) return rc;
        done += chunk;
    }
    return GK_OK;
}

int gk_map_count_reads_dev(gk_map *m, const void *dev_records, uint64_t nreads, int read_len, uint64_t *occurrences) {
    if (int rc = check_map_lazy(m)) return rc;
    gk_ctx *ctx = m->ctx;
    if (occurrences) *occurrences = 0;
    if (!dev_records && nreads) return fail(ctx, GK_E_INVALID, "gk_map_count_reads_dev: null records");
    if (read_len < 0 || read_len > 255) return fail(ctx, GK_E_FORMAT, "read_len must be 0..255 (one length byte per record)");
    reset_call_stats(m);
    if (nreads == 0) return GK_OK;
    ReadSrc src;
    src.rec = (const uint8_t *)dev_records;
    src.nreads = nreads;
    src.stride = 1 + (read_len + 3) / 4;
    src.max_len = read_len;
    const u64 nk = read_len >= m->k ? (u64)(read_len - m->k + 1) : 0;
    if (int rc = reset_occ_counter(m)) return rc;
    if (int rc = count_fixed_records(m, src, nk, 0)) return rc;
    uint64_t occ = 0;
    if (int rc = read_occ_counter(m, &occ)) return rc;
    m->last_count_occ = occ;
    m->total_occurrences += occ;
    if (occurrences) *occurrences = occ;
    return GK_OK;
}

// Owner side of the super-k-mer exchange (gk_skm.hip): the records are short reads in fixed slots;
// kmers_total (exchanged next to the record counts) is the exact number of windows they hold.
int gk_map_count_superkmers_dev(gk_map *m, const void *dev_records, uint64_t nrecords, uint64_t kmers_total, uint64_t *occurrences) {
    if (int rc = check_map_lazy(m)) return rc;
    gk_ctx *ctx = m->ctx;
    if (occurrences) *occurrences = 0;
    if (!dev_records && nrecords) return fail(ctx, GK_E_INVALID, "gk_map_count_superkmers_dev: null records");
    reset_call_stats(m);
    if (nrecords == 0) return GK_OK;
    ReadSrc src;
    src.rec = (const uint8_t *)dev_records;
    src.nreads = nrecords;
    src.stride = m->k <= 31 ? 16u : 32u;
    src.max_len = (int)(src.stride - 1) * 4;
    const u64 max_run = (u64)(src.max_len - m->k + 1);
    if (kmers_total > nrecords * max_run) return fail(ctx, GK_E_INVALID, "kmers_total exceeds what the records can hold");
    if (int rc = reset_occ_counter(m)) return rc;
    // a record holds <= max_run windows but typically a minimizer's reach, ~(k-m+2)/2: pick the lane
    // group that keeps the wave full
    src.group = lanes_per_read(std::min<int>((int)max_run, std::max(8, (m->k - 9) * 3 / 4)));
    // The announced total sizes the pipeline's regions; records that hold MORE than announced (a peer that lies) can only
    // overflow capped regions into the spill list and are reported below.  The exact-L1 test form has no caps: give it the hard bound.
    const u64 bound = ctx->hook_part_exact ? nrecords * max_run : kmers_total;
    if (part_batch_keys(m) >= bound) {
        if (int rc = insert_batch(m, src, nullptr, 0, bound)) return rc;
    } else {
        if (int rc = count_fixed_records(m, src, max_run, 0)) return rc;       // (huge exchange: batches bounded by what their slots can hold)
    }
    uint64_t occ = 0;
    if (int rc = read_occ_counter(m, &occ)) return rc;
    m->last_count_occ = occ;
    m->total_occurrences += occ;
    if (occurrences) *occurrences = occ;
    if (occ != kmers_total) return fail(ctx, GK_E_FORMAT, "records held " + std::to_string(occ) + " k-mers, caller announced " + std::to_string(kmers_total));
    return GK_OK;
}

// ---- host-fed streams: TWO staging areas in HBM ---------------------------------------------------------------------------
// While the pipeline works on the chunk in one of them (its P3 / P4 / P5 never touch the records again), the copy stream
// fills the other with the NEXT chunk — of the same call (the loop below looks one chunk ahead) or of the caller's next call
// (gk_map_prefetch_reads).  The chunk that was prefetched is then scattered in one launch behind the upload's event; a chunk
// that was not (the first of a call) is uploaded in pieces with P2 running on each piece as it lands (gk_partition.hip).
// Every insert ends synchronised, so an area is free again as soon as its chunk's call has returned: no further events.
static int stage_reserve(gk_map *m, int slot, size_t bytes) {
    gk_ctx *ctx = m->ctx;
    gk_map::StageSlot &st = m->stage[slot];
    if (st.cap >= bytes) return GK_OK;
    if (st.d) {
        GK_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));        // (an upload into this area may be in flight: pool_free does not wait for those)
        GK_HIP(ctx, hipFree(st.d));
    }
    st.d = nullptr; st.cap = 0;
    GK_HIP(ctx, hipMalloc(&st.d, bytes));
    st.cap = bytes;
    return GK_OK;
}
// ARM the upload of host bytes [p, p + bytes) into a staging area that is not `in_use` (-1: none is): the area is chosen and
// sized, the copy itself is issued by map_fire_prefetch — from the insert pipeline right after its L1 scatter has been
// launched, gated on that scatter's end, so that the copy runs beside the fine level (P3 / P4 / P5) and NOT beside the
// scatter: P2 is latency-bound and a running host copy doubles its time (measured: 0.58 -> 1.18 ms at C2 with the next step's
// 39 MB arriving meanwhile; profiles/r03).  Of two free areas the one that holds no unconsumed prefetch is taken.
static int stage_prefetch_arm(gk_map *m, const uint8_t *p, size_t bytes, int in_use) {
    gk_ctx *ctx = m->ctx;
    int slot;
    if (in_use >= 0) slot = 1 - in_use;
    else if (m->stage[0].valid != m->stage[1].valid) slot = m->stage[0].valid ? 1 : 0;
    else slot = 1 - m->stage_last_pf;
    gk_map::StageSlot &st = m->stage[slot];
    st.valid = false; st.armed = false;
    if (int rc = stage_reserve(m, slot, bytes + 64)) return rc;
    if (!st.ev) GK_HIP(ctx, hipEventCreateWithFlags(&st.ev, hipEventDisableTiming));
    st.host = p; st.bytes = bytes; st.armed = true;
    m->stage_last_pf = slot;
    return GK_OK;
}
}  // extern "C"
namespace gk {
// issue every armed upload on the copy stream, behind `after` (an event of the main stream; nullptr: at once)
int map_fire_prefetch(gk_map *m, hipEvent_t after) {
    gk_ctx *ctx = m->ctx;
    for (int sl = 0; sl < 2; sl++) {
        gk_map::StageSlot &st = m->stage[sl];
        if (!st.armed) continue;
        st.armed = false;
        if (after) GK_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, after, 0));
        GK_HIP(ctx, hipMemcpyAsync(st.d, st.host, st.bytes, hipMemcpyHostToDevice, ctx->copy_stream));
        GK_HIP(ctx, hipEventRecord(st.ev, ctx->copy_stream));
        st.valid = true;
    }
    return GK_OK;
}
}
extern "C" {

int gk_map_prefetch_reads(gk_map *m, const uint8_t *bin, size_t nbytes, uint64_t nreads) {
    if (int rc = check_map_lazy(m)) return rc;
    gk_ctx *ctx = m->ctx;
    if (!bin || !nbytes || !nreads) return fail(ctx, GK_E_INVALID, "gk_map_prefetch_reads: empty stream");
    // (only the head of a long stream: what the next call's first chunk will hold at most)
    {
        // the head of the stream: whole records, as many as the next call's first chunk can hold (equal-length records assumed — a
        // ragged stream's first chunk may end elsewhere, and the prefetch is then simply not used)
        const size_t rb = 1 + (size_t)(bin[0] + 3) / 4;
        const size_t want = std::min<size_t>(nbytes, std::max<size_t>(rb, max_stage(ctx) / rb * rb));
        return stage_prefetch_arm(m, bin, want, -1);
    }
}

int gk_map_count_reads(gk_map *m, const uint8_t *bin, size_t nbytes, uint64_t nreads, uint64_t *occurrences) {
    if (int rc = check_map_lazy(m)) return rc;
    gk_ctx *ctx = m->ctx;
    if (occurrences) *occurrences = 0;
    if (!bin && nreads) return fail(ctx, GK_E_INVALID, "gk_map_count_reads: null stream");
    reset_call_stats(m);
    if (nreads == 0) return GK_OK;
    if (int rc = reset_occ_counter(m)) return rc;
    // Walk the record framing once on the host (one length byte per record, PairedEndData.scala:24-31),
    // cutting the stream into launches bounded in bytes (staging area) and in windows (load limit, key scratch).
    struct Chunk {
        size_t begin = 0, bytes = 0;
        u64 r_begin = 0, reads = 0, occ = 0;
        int first_len = -1;
        bool uniform = true, unverified = false, valid = false;
        std::vector<u32> offs;
    };
    // the chunk that starts at (pos, r); GK_E_FORMAT through *err
    auto cut = [&](size_t pos, u64 r, bool walk, int *err) {
        Chunk c;
        c.begin = pos; c.r_begin = r;
        if (r >= nreads) return c;
        const size_t chunk_begin = pos;
        u64 occ = 0;
        // windows per chunk: what the table has room for on the direct path; one partitioned batch's key scratch otherwise
        const u64 occ_limit = std::max<u64>(reads_per_launch(m, 1), use_partitioned(m, part_batch_keys(m)) ? part_batch_keys(m) : 0);
        bool fast_prefix = false;
        // Fast prefix: a run of equal-length records (one sequencing run) is recognised by comparing one byte per
        // record, with no offset table built; it becomes a chunk of its own when it is long enough to be worth it.
        if (pos < nbytes) {
            const int len0 = bin[pos];
            const size_t rb0 = 1 + (size_t)(len0 + 3) / 4;
            const u64 nk0 = len0 >= m->k ? (u64)(len0 - m->k + 1) : 0;
            u64 cap = std::min<u64>(nreads - r, (nbytes - pos) / rb0);
            // (A short FIRST chunk — its upload cannot hide behind anything — was tried and lost: 128 MiB first, C3 from host memory
            //  129 ms against 107-116 with equal chunks: one more batch, i.e. one more pass over the table, costs more than the
            //  2 ms of exposed upload it saves; profiles/r03/c3_host_fed_variants.txt)
            const size_t stage_cap = max_stage(ctx);
            cap = std::min<u64>(cap, std::max<u64>(1, stage_cap / rb0));
            if (nk0) cap = std::min<u64>(cap, std::max<u64>(1, occ_limit / nk0));
            const uint8_t *p0 = bin + pos;
            u64 run = 0;
            // What is left is EXACTLY (reads left) records of this length: one sequencing run, almost surely.  Then the
            // host does not walk a million length bytes (~1 ms per 39 MB: a third of the whole insert) — the L1 scatter
            // checks them on the device as it goes, and gives the chunk back if one differs (PART_NOT_UNIFORM).
            const bool looks_uniform = !walk && !ctx->hook_host_ragged && !ctx->hook_part_exact && nk0 && (nbytes - pos) == (size_t)(nreads - r) * rb0 &&
                                       cap >= 4096 && use_partitioned(m, cap * nk0) && nk0 * (u64)m->W <= (u64)5632;
            if (looks_uniform) { run = cap; c.unverified = true; }
            while (run < cap && p0[run * rb0] == (uint8_t)len0) run++;
            if (run >= 4096 || (run == nreads - r && run > 0)) {
                fast_prefix = true;
                c.first_len = len0;
                pos += run * rb0;
                occ = run * nk0;
                r += run;
            } else c.unverified = false;
        }
        while (!fast_prefix && r < nreads) {
            if (pos >= nbytes) { *err = 1; return c; }
            int len = bin[pos];
            if (c.first_len < 0) c.first_len = len; else if (len != c.first_len) c.uniform = false;
            size_t rb = 1 + (size_t)(len + 3) / 4;
            if (pos + rb > nbytes) { *err = 2; return c; }
            u64 nk = len >= m->k ? (u64)(len - m->k + 1) : 0;
            if (r > c.r_begin && (pos + rb - chunk_begin > max_stage(ctx) || occ + nk > occ_limit)) break;
            c.offs.push_back((u32)(pos - chunk_begin));
            pos += rb;
            occ += nk;
            r++;
        }
        if (!c.offs.empty()) c.offs.push_back((u32)(pos - chunk_begin));
        c.bytes = pos - chunk_begin; c.reads = r - c.r_begin; c.occ = occ;
        c.valid = c.reads > 0;
        return c;
    };
    auto format_error = [&](int err, const Chunk &c) {
        return fail(ctx, GK_E_FORMAT, err == 1 ? "truncated .bin stream: record " + std::to_string(c.r_begin + c.reads) + " starts past the end"
                                                : "truncated .bin stream inside record " + std::to_string(c.r_begin + c.reads));
    };
    int err = 0;
    Chunk cur = cut(0, 0, false, &err);
    if (err) return format_error(err, cur);
    while (cur.valid) {
        // this chunk's staging area: the one its bytes were prefetched into (by the previous round of this loop, or by
        // gk_map_prefetch_reads before the call), else the one that is free
        bool preloaded = false;
        for (int sl = 0; sl < 2 && !preloaded; sl++) {
            gk_map::StageSlot &st = m->stage[sl];
            if (st.valid && st.host == bin + cur.begin && st.bytes >= cur.bytes) { preloaded = true; m->stage_cur = sl; st.valid = false; }
        }
        if (!preloaded) {
            // Armed for this very stream but never fired (no insert ran since): the piece-wise upload below is the faster way for
            // THIS chunk, so the request is dropped — the OLDER one if two are armed: a streaming caller that re-uses one buffer
            // has armed the next batch too (same address), and that one must stay armed to fire behind this batch's scatter.
            const int older = 1 - m->stage_last_pf;
            const bool a0 = m->stage[older].armed && m->stage[older].host == bin + cur.begin;
            const bool a1 = m->stage[1 - older].armed && m->stage[1 - older].host == bin + cur.begin;
            if (a0) m->stage[older].armed = false;
            else if (a1) m->stage[1 - older].armed = false;
        }
        if (!preloaded) {
            const bool busy0 = m->stage[0].valid || m->stage[0].armed, busy1 = m->stage[1].valid || m->stage[1].armed;
            m->stage_cur = busy0 && !busy1 ? 1 : 0;       // (spare a pending prefetch if one of the areas holds none)
            m->stage[m->stage_cur].valid = m->stage[m->stage_cur].armed = false;
            if (int rc = stage_reserve(m, m->stage_cur, cur.bytes + 64)) return rc;
        }
        // one chunk ahead: its upload runs on the copy stream beside this chunk's P3 / P4 / P5
        Chunk nxt = cut(cur.begin + cur.bytes, cur.r_begin + cur.reads, false, &err);
        if (err) return format_error(err, nxt);
        if (nxt.valid && ctx->hook_host_prefetch != 0) { if (int rc = stage_prefetch_arm(m, bin + nxt.begin, nxt.bytes, m->stage_cur)) return rc; }
        if (m->offsets_bytes < cur.offs.size() * sizeof(u32)) {
            if (m->d_offsets) GK_HIP(ctx, hipFree(m->d_offsets));
            m->d_offsets = nullptr;
            m->offsets_bytes = 0;
            GK_HIP(ctx, hipMalloc(&m->d_offsets, cur.offs.size() * sizeof(u32)));
            m->offsets_bytes = cur.offs.size() * sizeof(u32);
        }
        // the records stay in host memory for now: the consumer uploads them (ReadSrc::host) — unless they are on their way already
        // a chunk of equal-length records (the usual case: one sequencing run) is a fixed-stride array: no offset
        // table to upload, and the partitioned path can take its one-extraction form
        ReadSrc src;
        src.rec = (const uint8_t *)m->stage[m->stage_cur].d;
        src.nreads = cur.reads;
        src.host = bin + cur.begin;
        src.host_bytes = cur.bytes;
        src.ready = preloaded ? m->stage[m->stage_cur].ev : nullptr;
        src.verify_uniform = cur.unverified;
        if (cur.uniform && cur.first_len >= 0 && !ctx->hook_host_ragged) {
            src.stride = 1 + (u32)(cur.first_len + 3) / 4;
            src.max_len = cur.first_len;
        } else {
            GK_HIP(ctx, hipMemcpyAsync(m->d_offsets, cur.offs.data(), cur.offs.size() * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
            src.off = (const u32 *)m->d_offsets;
            src.max_len = 255;             // the host has walked this framing: every length byte is what the offsets say
        }
        // (grow_ahead: what the call still holds beyond this chunk, so that a table that has to grow leaves room for it)
        const int brc = insert_batch(m, src, nullptr, 0, cur.occ, false, (double)(nreads - cur.r_begin) / (double)std::max<u64>(cur.reads, 1));
        if (brc == GK_OK) { if (int rc = map_fire_prefetch(m, nullptr)) return rc; }      // (still armed: the path that ran had no L1 scatter to gate it on)
        if (brc == PART_NOT_UNIFORM) {       // take the chunk again, this time walking its framing
            GK_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));     // (a prefetch of the chunk after it may be in flight: it is simply dropped)
            for (int sl = 0; sl < 2; sl++) m->stage[sl].valid = m->stage[sl].armed = false;
            cur = cut(cur.begin, cur.r_begin, true, &err);
            if (err) return format_error(err, cur);
            continue;
        }
        if (brc) { (void)hipStreamSynchronize(ctx->copy_stream); for (int sl = 0; sl < 2; sl++) m->stage[sl].valid = m->stage[sl].armed = false; return brc; }
        cur = nxt;
    }
    uint64_t occ = 0;
    if (int rc = read_occ_counter(m, &occ)) return rc;
    m->last_count_occ = occ;
    m->total_occurrences += occ;
    if (occurrences) *occurrences = occ;
    return GK_OK;
}

static int add_keys_dev(gk_map *m, const u64 *d_keys, const i32 *d_counts, u64 n, bool verbatim) {
    gk_ctx *ctx = m->ctx;
    u64 done = 0;
    while (done < n) {
        u64 chunk = std::min(n - done, reads_per_launch(m, 1));
        if (int rc = map_reserve(m, chunk)) return rc;
        int grid = grid_for(ctx, chunk, BLOCK);
        GK_BY_SLOT(m, hipLaunchKernelGGL((k_add_keys<W, S>), dim3(grid), dim3(BLOCK), 0, ctx->stream, d_keys + W * done,
                                         d_counts ? d_counts + done : nullptr, chunk, table_of<W, S>(m), m->d_ctr, verbatim ? m->k : 0));
        GK_HIP(ctx, hipGetLastError());
        if (int rc = map_sync_counters(m)) return rc;
        done += chunk;
    }
    return GK_OK;
}

}  // extern "C"
namespace gk {
int map_add_keys_direct(gk_map *m, const uint64_t *d_keys, uint64_t n) { return n ? add_keys_dev(m, d_keys, nullptr, n) : GK_OK; }
int map_add_counted_keys_dev(gk_map *m, const uint64_t *d_keys, const int32_t *d_counts, uint64_t n) {
    if (int rc = map_materialize(m)) return rc;
    return n ? add_keys_dev(m, d_keys, d_counts, n) : GK_OK;
}
// the same for keys known to be absent and distinct, into a GRAPH-layout table that was sized for them (gk_map_create_for_graph):
// counts and optional degree masks stored beside the claim (k_add_unique)
int map_add_unique_keys_dev(gk_map *m, const uint64_t *d_keys, const int32_t *d_counts, const uint8_t *d_masks, uint64_t n) {
    gk_ctx *ctx = m->ctx;
    if (int rc = map_materialize(m)) return rc;
    if (!n) return GK_OK;
    if (m->W == 1 && m->layout != LAYOUT_GRAPH) return fail(ctx, GK_E_STATE, "map_add_unique_keys_dev: count-layout table");
    if (int rc = map_reserve(m, n)) return rc;
    const int grid = grid_for(ctx, n, BLOCK);
    if (m->W == 1) hipLaunchKernelGGL((k_add_unique<1>), dim3(grid), dim3(BLOCK), 0, ctx->stream, d_keys, d_counts, d_masks, n, table_of<1, Slot<1>>(m), m->d_ctr);
    else hipLaunchKernelGGL((k_add_unique<2>), dim3(grid), dim3(BLOCK), 0, ctx->stream, d_keys, d_counts, d_masks, n, table_of<2, Slot<2>>(m), m->d_ctr);
    GK_HIP(ctx, hipGetLastError());
    return map_sync_counters(m);
}
// the live (key, count) of slots [s0, s1) of m, packed into d_keys (W words per key) / d_cnt; d_cursor: 8 bytes of device scratch
int map_export_range_dev(gk_map *m, uint64_t s0, uint64_t s1, uint64_t *d_keys, int32_t *d_cnt, unsigned long long *d_cursor, uint64_t *n_out, uint8_t *d_masks) {
    gk_ctx *ctx = m->ctx;
    *n_out = 0;
    if (d_masks && m->W == 1 && m->layout != LAYOUT_GRAPH) return fail(ctx, GK_E_STATE, "masks of a count-layout table");
    if (s1 > m->capacity) s1 = m->capacity;
    if (s0 >= s1) return GK_OK;
    if (int rc = map_materialize(m)) return rc;
    GK_HIP(ctx, hipMemsetAsync(d_cursor, 0, 8, ctx->stream));
    const int grid = grid_for(ctx, s1 - s0, BLOCK);
    GK_BY_SLOT(m, hipLaunchKernelGGL((k_export_packed<W, S>), dim3(grid), dim3(BLOCK), 0, ctx->stream, (const S *)m->slots + s0, s1 - s0, s0, m->k == 64 ? 1u : 0u, d_keys,
                                     d_cnt, d_cursor, d_masks));
    GK_HIP(ctx, hipGetLastError());
    unsigned long long n = 0;
    GK_HIP(ctx, hipMemcpyAsync(&n, d_cursor, 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = n;
    return GK_OK;
}
// a new, empty map whose table is sized for `keys` keys the way the graph phase wants it (graph_table_load)
int map_create_for_graph(gk_ctx *ctx, int k, uint64_t keys, gk_map **out) {
    const double load = graph_table_load(ctx, k, keys);
    // gk_map_create sizes for its own target load: hand it the key count that gives the wanted number of slots
    gk_map probe; probe.k = k; probe.ctx = ctx;
    const uint64_t hint = (uint64_t)((double)std::max<uint64_t>(keys, 1) / load * target_load(&probe)) + 1;
    if (int rc = gk_map_create(ctx, k, hint, out)) return rc;
    gk_map *m = *out;
    if (m->W == 1) {            // the graph layout (16-byte slots with the annotation word): the table was allocated for 12-byte slots
        GK_HIP(ctx, hipFree(m->slots));
        m->slots = nullptr;
        m->layout = LAYOUT_GRAPH;
        hipError_t e = hipMalloc(&m->slots, m->capacity * map_slot_bytes(m));
        if (e != hipSuccess) { const int rc = hip_fail(ctx, e, "gk_map_create_for_graph: table"); gk_map_destroy(m); *out = nullptr; return rc == GK_E_HIP ? fail(ctx, GK_E_CAPACITY, "cannot allocate table: " + ctx->err) : rc; }
    }
    return GK_OK;
}
}
extern "C" {

}  // extern "C"
namespace gk {
int map_insert_keys_dev(gk_map *m, const uint64_t *keys, uint64_t n, bool verbatim) {
    reset_call_stats(m);
    m->last_count_occ = n;
    const ReadSrc none;
    u64 done = 0;
    while (done < n) {
        u64 chunk = std::min<u64>(n - done, part_batch_keys(m));
        if (!use_partitioned(m, chunk)) chunk = std::min(n - done, reads_per_launch(m, 1));
        if (int rc = insert_batch(m, none, keys + done * m->W, chunk, chunk, verbatim)) return rc;
        done += chunk;
    }
    return GK_OK;
}
}
extern "C" {

int gk_map_update_inc_dev(gk_map *m, const void *dev_keys, uint64_t n) {
    if (int rc = check_map_lazy(m)) return rc;
    if (!dev_keys && n) return fail(m->ctx, GK_E_INVALID, "gk_map_update_inc_dev: null keys");
    if (n == 0) return GK_OK;
    return map_insert_keys_dev(m, (const u64 *)dev_keys, n, true);       // verbatim: non-canonical keys are noticed (gk_graph_build)
}

// validate that every key fits in 2k bits — the C-ABI form of `assert(key.length == k)`
// (ArrayDNAMap.scala:182,199): a key with bits above 2k cannot be a k-mer of this map.
static int check_key_bits(const gk_map *m, const uint64_t *lo, const uint64_t *hi, uint64_t n) {
    const int k = m->k;
    for (uint64_t i = 0; i < n; i++) {
        bool bad;
        if (m->W == 1) bad = (lo[i] >> (2 * k)) != 0 || (hi && hi[i] != 0);
        else bad = k < 64 && (hi[i] >> (2 * (k - 32))) != 0;
        if (bad) return fail(m->ctx, GK_E_KLEN, "key " + std::to_string(i) + " is not a " + std::to_string(k) + "-mer (bits set above 2k)");
    }
    return GK_OK;
}

}  // extern "C"
namespace gk {
// Pooled scratch of the point-query / scan entry points: grown on demand, kept with the map — `apply()` from the
// Scala adapter is one gk_map_get_batch per call, and a hipMalloc/hipFree pair per call is two implicit device syncs.
void *map_scratch(gk_map *m, size_t bytes) {
    if (m->scratch_bytes >= bytes) return m->d_scratch;
    gk_ctx *ctx = m->ctx;
    if (m->d_scratch) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(m->d_scratch); }
    m->d_scratch = nullptr; m->scratch_bytes = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 4, 1 << 16);
    hipError_t e = hipMalloc(&m->d_scratch, want);
    if (e != hipSuccess) { (void)hipGetLastError(); (void)fail(ctx, GK_E_CAPACITY, std::string("scratch allocation failed: ") + hipGetErrorString(e)); return nullptr; }
    m->scratch_bytes = want;
    return m->d_scratch;
}
}
extern "C" {
static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

int gk_map_add_counts(gk_map *m, const uint64_t *lo, const uint64_t *hi, const int32_t *counts, uint64_t n) {
    if (int rc = check_map(m)) return rc;
    gk_ctx *ctx = m->ctx;
    if (n == 0) return GK_OK;
    if (!lo || (m->W == 2 && !hi)) return fail(ctx, GK_E_INVALID, "null key array");
    if (int rc = check_key_bits(m, lo, hi, n)) return rc;
    const size_t kb = al256(n * 8 * m->W);
    char *base = (char *)map_scratch(m, kb + al256(n * 4));
    if (!base) return GK_E_CAPACITY;
    u64 *d_keys = (u64 *)base;
    i32 *d_counts = counts ? (i32 *)(base + kb) : nullptr;
    if (m->W == 1) {
        GK_HIP(ctx, hipMemcpyAsync(d_keys, lo, n * 8, hipMemcpyHostToDevice, ctx->stream));
    } else {
        std::vector<u64> inter((size_t)n * 2);
        for (uint64_t i = 0; i < n; i++) { inter[2 * i] = lo[i]; inter[2 * i + 1] = hi[i]; }
        GK_HIP(ctx, hipMemcpyAsync(d_keys, inter.data(), inter.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        GK_HIP(ctx, hipStreamSynchronize(ctx->stream));          // `inter` dies with this scope
    }
    if (counts) GK_HIP(ctx, hipMemcpyAsync(d_counts, counts, n * sizeof(i32), hipMemcpyHostToDevice, ctx->stream));
    return add_keys_dev(m, d_keys, d_counts, n, true);      // host keys are taken verbatim
}

int gk_map_update_inc(gk_map *m, const uint64_t *lo, const uint64_t *hi, uint64_t n) {
    return gk_map_add_counts(m, lo, hi, nullptr, n);
}

// update(key, c, _ + c) for every (key, c) of `src`, device to device, in chunks of the source's slots: how the partitions of a
// PartitionedDNAMap that share a device are merged (the counterpart of gk_dist_gather_map, which does the same per chunk with
// an exchange in between).  Staging: 2^25 slots' worth of packed entries at most (0.4 / 0.67 GB).
int gk_map_add_map(gk_map *dst, gk_map *src) {
    if (int rc = check_map(dst)) return rc;
    gk_ctx *ctx = dst->ctx;
    if (!src || src->ctx != ctx || src == dst) return fail(ctx, GK_E_INVALID, "gk_map_add_map: the source must be another map of the same context");
    if (src->k != dst->k) return fail(ctx, GK_E_KLEN, "gk_map_add_map: the maps' k differ");
    if (int rc = map_materialize(src)) return rc;
    if (src->size == 0) return GK_OK;
    if (int rc = map_reserve(dst, src->size)) return rc;
    constexpr u64 CHS = 1ull << 25;
    const u64 cap = std::min<u64>(CHS, src->capacity);
    u64 *d_keys = nullptr;
    i32 *d_cnt = nullptr;
    unsigned long long *d_cur = nullptr;
    auto done = [&](int code) { for (void *p : {(void *)d_keys, (void *)d_cnt, (void *)d_cur}) if (p) (void)hipFree(p); return code; };
    hipError_t e = hipMalloc((void **)&d_keys, cap * 8 * dst->W);
    if (e == hipSuccess) e = hipMalloc((void **)&d_cnt, cap * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_cur, 8);
    if (e != hipSuccess) return done(hip_fail(ctx, e, "gk_map_add_map: staging"));
    for (u64 s0 = 0; s0 < src->capacity; s0 += CHS) {
        uint64_t n = 0;
        if (int rc = map_export_range_dev(src, s0, s0 + CHS, d_keys, d_cnt, d_cur, &n)) return done(rc);
        if (n) { if (int rc = add_keys_dev(dst, d_keys, d_cnt, n, src->dirty)) return done(rc); }
    }
    if (src->dirty) dst->dirty = true;
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return done(GK_OK);
}

// replace the table by one sized for its live keys (after deleteAll: the reference rescales too, ArrayDNAMap.scala:214)
static int map_compact(gk_map *m) {
    gk_ctx *ctx = m->ctx;
    uint32_t nnb2, nlnb1;
    uint64_t ncap;
    // the read-only graph phase follows: a sparse table (graph_table_load has the measurements and the memory rule)
    const double graph_load = graph_table_load(ctx, m->k, m->size);
    plan_segments(m->W, (uint64_t)((double)m->size / graph_load) + 1, &nnb2, &nlnb1, &ncap, (uint32_t)ctx->hook_min_lnb1);
    if (ncap > m->capacity) { nnb2 = m->nb2; nlnb1 = m->lnb1; ncap = m->capacity; }
    void *nslots = nullptr;
    if (alloc_table(ctx, m->W, LAYOUT_GRAPH, ncap, &nslots) != GK_OK) { if (nslots) (void)hipFree(nslots); return GK_OK; }   // keep tombstones if memory is short
    launch_rehash(m, LAYOUT_GRAPH, nslots, nnb2, nlnb1);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipFree(nslots); return hip_fail(ctx, e, "table compaction"); }
    if (int rc = map_sync_counters(m)) { (void)hipFree(nslots); return rc; }       // a failed rehash leaves the old table in place
    GK_HIP(ctx, hipFree(m->slots));
    m->slots = nslots;
    m->capacity = ncap;
    m->nb2 = nnb2;
    m->lnb1 = nlnb1;
    m->tombstones = 0;
    m->layout = LAYOUT_GRAPH;
    return GK_OK;
}

// Move the live keys with count >= rounds into a new table of geometry (nnb2, nlnb1 == m->lnb1) with k_compact_seg and install
// it; the old table stays (and *done stays false) if the new one cannot be allocated.  Not for k = 64.
}  // extern "C"
template <int W, class SO, class SN>
static hipError_t launch_compact(gk_map *m, void *nslots, uint32_t nnb2, uint32_t nlnb1, int32_t rounds, int gc) {
    gk_ctx *ctx = m->ctx;
    const size_t lds = sizeof(SN) << SegBits<W>::value;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_compact_seg<W, SO, SN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) hipLaunchKernelGGL((k_compact_seg<W, SO, SN>), dim3(gc), dim3(CBLOCK), lds, ctx->stream, table_of<W, SO>(m), Table<W, SN>{(SN *)nslots, nnb2, nlnb1, 0u}, rounds,
                                            m->d_ctr, &m->d_ctr->rebuild_kept);
    return e;
}
extern "C" {
static int streaming_rebuild(gk_map *m, uint32_t nnb2, uint32_t nlnb1, uint64_t ncap, int32_t rounds, int new_layout, bool *done) {
    gk_ctx *ctx = m->ctx;
    // (the counter lives in the map's Counters block, not in the pooled scratch: a caller further up — gk_map_add_counts through
    //  map_reserve — may be holding its keys there)
    unsigned long long h_kept = 0;
    GK_HIP(ctx, hipMemsetAsync(&m->d_ctr->rebuild_kept, 0, 8, ctx->stream));
    void *nslots = nullptr;
    if (hipMalloc(&nslots, ncap * slot_bytes(m->W, new_layout)) != hipSuccess) { (void)hipGetLastError(); return GK_OK; }    // (every slot is written below: no clear)
    const int gc = (int)std::min<u64>((u64)nnb2 << nlnb1, (u64)ctx->cu_count * 16);
    hipError_t e = hipSuccess;
    if (m->W == 2) e = launch_compact<2, Slot<2>, Slot<2>>(m, nslots, nnb2, nlnb1, rounds, gc);
    else if (m->layout == LAYOUT_GRAPH && new_layout == LAYOUT_GRAPH) e = launch_compact<1, Slot<1>, Slot<1>>(m, nslots, nnb2, nlnb1, rounds, gc);
    else if (m->layout == LAYOUT_GRAPH) e = launch_compact<1, Slot<1>, CSlot>(m, nslots, nnb2, nlnb1, rounds, gc);
    else if (new_layout == LAYOUT_GRAPH) e = launch_compact<1, CSlot, Slot<1>>(m, nslots, nnb2, nlnb1, rounds, gc);
    else e = launch_compact<1, CSlot, CSlot>(m, nslots, nnb2, nlnb1, rounds, gc);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&h_kept, &m->d_ctr->rebuild_kept, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipFree(nslots); return hip_fail(ctx, e, "streaming table rebuild"); }
    const u64 kept = h_kept;
    if (int rc = map_sync_counters(m)) {                   // a segment of the new table filled up (a sizing error): the old table stays
        (void)hipFree(nslots);
        return rc;
    }
    unsigned long long sz = kept;
    GK_HIP(ctx, hipMemcpyAsync(&m->d_ctr->size, &sz, sizeof(sz), hipMemcpyHostToDevice, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));         // (`sz` is a stack variable)
    GK_HIP(ctx, hipFree(m->slots));
    m->slots = nslots;
    m->capacity = ncap;
    m->nb2 = nnb2;
    m->lnb1 = nlnb1;
    m->tombstones = 0;
    m->size = kept;
    m->layout = new_layout;
    *done = true;
    return GK_OK;
}

// filter + compaction as one streaming pass (k_compact_seg).  GK_OK with *done = true when it replaced the table; *done = false
// when the geometry does not allow it (k = 64's tagged slots, another L1 fan-out, no memory for the new table) — the caller
// then takes the tombstone + rehash path.
static int filter_compact_streaming(gk_map *m, int32_t rounds, bool *done) {
    gk_ctx *ctx = m->ctx;
    *done = false;
    if (m->k == 64 || ctx->hook_filter_classic > 0) return GK_OK;
    const u64 nseg = (u64)m->nb2 << m->lnb1;
    unsigned long long *d2 = &m->d_ctr->rebuild_sample;
    unsigned long long h2[1] = {0};
    GK_HIP(ctx, hipMemsetAsync(d2, 0, 8, ctx->stream));
    // 1. survivors, estimated from every 16th segment (the hash spreads keys evenly: +-1 % at any size that matters)
    const u32 every = nseg >= 4096 ? 16u : 1u;
    const int gs = (int)std::min<u64>((nseg + every - 1) / every, (u64)ctx->cu_count * 8);
    GK_BY_SLOT(m, hipLaunchKernelGGL((k_count_ge_sample<W, S>), dim3(gs), dim3(BLOCK), 0, ctx->stream, (const S *)m->slots, nseg, every, rounds, d2));
    GK_HIP(ctx, hipGetLastError());
    GK_HIP(ctx, hipMemcpyAsync(h2, d2, 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const u64 est = (u64)((double)h2[0] * every * 1.03) + 1024;
    // 2. the new table's geometry (same rule as map_compact)
    const double graph_load = graph_table_load(ctx, m->k, est);
    uint32_t nnb2, nlnb1;
    uint64_t ncap;
    plan_segments(m->W, (uint64_t)((double)est / graph_load) + 1, &nnb2, &nlnb1, &ncap, (uint32_t)ctx->hook_min_lnb1);
    if (ncap > m->capacity) { nnb2 = m->nb2; nlnb1 = m->lnb1; ncap = m->capacity; }
    if (nlnb1 != m->lnb1) return GK_OK;                    // the L1 bucket of a key would change: not a segment-local move
    return streaming_rebuild(m, nnb2, nlnb1, ncap, rounds, LAYOUT_GRAPH, done);
}

int gk_map_filter_lt(gk_map *m, int32_t rounds) {
    if (int rc = check_map(m)) return rc;
    gk_ctx *ctx = m->ctx;
    m->masks_valid = false;
    {
        bool done = false;
        if (int rc = filter_compact_streaming(m, rounds, &done)) return rc;
        if (done) return GK_OK;
    }
    unsigned long long *d_removed = (unsigned long long *)map_scratch(m, 256);
    if (!d_removed) return GK_E_CAPACITY;
    GK_HIP(ctx, hipMemsetAsync(d_removed, 0, sizeof(unsigned long long), ctx->stream));
    int grid = grid_for(ctx, m->capacity, BLOCK * 4);
    GK_BY_SLOT(m, hipLaunchKernelGGL(k_filter_lt<S>, dim3(grid), dim3(BLOCK), 0, ctx->stream, (S *)m->slots, m->capacity, rounds, d_removed));
    GK_HIP(ctx, hipGetLastError());
    unsigned long long removed = 0;
    GK_HIP(ctx, hipMemcpyAsync(&removed, d_removed, sizeof(removed), hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    m->size -= removed;
    m->tombstones += removed;
    unsigned long long sz = m->size;
    GK_HIP(ctx, hipMemcpyAsync(&m->d_ctr->size, &sz, sizeof(sz), hipMemcpyHostToDevice, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // Rebuild into a table sized for the survivors whenever tombstones exist, so that the read-only graph
    // phase probes a clean, cache-friendlier table.
    return m->tombstones ? map_compact(m) : GK_OK;
}

}  // extern "C"
namespace gk {
int map_to_graph_layout(gk_map *m) {
    if (m->W != 1 || m->layout == LAYOUT_GRAPH) return GK_OK;
    gk_ctx *ctx = m->ctx;
    // same geometry when the table is sparse enough for the graph phase already, else sized by graph_table_load; tombstones go either way
    uint32_t nnb2 = m->nb2, nlnb1 = m->lnb1;
    uint64_t ncap = m->capacity;
    const double load = graph_table_load(ctx, m->k, m->size);
    if ((double)m->size > load * (double)m->capacity) {
        plan_segments(m->W, (uint64_t)((double)m->size / load) + 1, &nnb2, &nlnb1, &ncap, (uint32_t)ctx->hook_min_lnb1);
        if (ncap < m->capacity) { nnb2 = m->nb2; nlnb1 = m->lnb1; ncap = m->capacity; }
    }
    if (nlnb1 == m->lnb1 && ctx->hook_filter_classic <= 0) {
        bool done = false;
        if (int rc = ::streaming_rebuild(m, nnb2, nlnb1, ncap, INT32_MIN, LAYOUT_GRAPH, &done)) return rc;
        if (done) return GK_OK;
    }
    // (another L1 fan-out, or no memory for the streaming form's second table... the rehash form needs one too)
    void *nslots = nullptr;
    if (alloc_table(ctx, m->W, LAYOUT_GRAPH, ncap, &nslots) != GK_OK) { if (nslots) (void)hipFree(nslots); return fail(ctx, GK_E_CAPACITY, "cannot rebuild the table in the graph layout: " + ctx->err); }
    launch_rehash(m, LAYOUT_GRAPH, nslots, nnb2, nlnb1);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipFree(nslots); return hip_fail(ctx, e, "table rebuild (graph layout)"); }
    if (int rc = map_sync_counters(m)) { (void)hipFree(nslots); return rc; }
    GK_HIP(ctx, hipFree(m->slots));
    m->slots = nslots; m->capacity = ncap; m->nb2 = nnb2; m->lnb1 = nlnb1; m->tombstones = 0; m->layout = LAYOUT_GRAPH;
    return GK_OK;
}
}  // namespace gk
extern "C" {

int gk_map_get_batch(gk_map *m, const uint64_t *lo, const uint64_t *hi, uint64_t n, int32_t *counts_out, uint8_t *found_out) {
    if (int rc = check_map(m)) return rc;
    gk_ctx *ctx = m->ctx;
    if (n == 0) return GK_OK;
    if (!lo || (m->W == 2 && !hi)) return fail(ctx, GK_E_INVALID, "null key array");
    if (int rc = check_key_bits(m, lo, hi, n)) return rc;
    const size_t b8 = al256(n * 8), b4 = al256(n * 4), b1 = al256(n);
    char *base = (char *)map_scratch(m, 2 * b8 + b4 + b1);
    if (!base) return GK_E_CAPACITY;
    u64 *d_lo = (u64 *)base, *d_hi = m->W == 2 ? (u64 *)(base + b8) : nullptr;
    i32 *d_cnt = counts_out ? (i32 *)(base + 2 * b8) : nullptr;
    uint8_t *d_found = found_out ? (uint8_t *)(base + 2 * b8 + b4) : nullptr;
    GK_HIP(ctx, hipMemcpyAsync(d_lo, lo, n * 8, hipMemcpyHostToDevice, ctx->stream));
    if (d_hi) GK_HIP(ctx, hipMemcpyAsync(d_hi, hi, n * 8, hipMemcpyHostToDevice, ctx->stream));
    int grid = grid_for(ctx, n, BLOCK);
    GK_BY_SLOT(m, hipLaunchKernelGGL((k_get<W, S>), dim3(grid), dim3(BLOCK), 0, ctx->stream, d_lo, d_hi, n, table_of<W, S>(m), d_cnt, d_found));
    GK_HIP(ctx, hipGetLastError());
    if (counts_out) GK_HIP(ctx, hipMemcpyAsync(counts_out, d_cnt, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (found_out) GK_HIP(ctx, hipMemcpyAsync(found_out, d_found, n, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GK_OK;
}

int gk_map_export(gk_map *m, uint64_t *lo, uint64_t *hi, int32_t *counts, uint64_t cap, uint64_t *n) {
    if (int rc = check_map(m)) return rc;
    gk_ctx *ctx = m->ctx;
    if (n) *n = m->size;
    if (m->size > cap) return fail(ctx, GK_E_CAPACITY, "export buffer too small: need " + std::to_string(m->size));
    if (m->size == 0) return GK_OK;
    if (!lo || !counts || (m->W == 2 && !hi)) return fail(ctx, GK_E_INVALID, "null export buffer");
    const u64 cnt = m->size;
    const size_t b8 = al256(cnt * 8), b4 = al256(cnt * 4);
    char *base = (char *)map_scratch(m, 2 * b8 + b4 + 256);
    if (!base) return GK_E_CAPACITY;
    u64 *d_lo = (u64 *)base, *d_hi = (u64 *)(base + b8);
    i32 *d_cnt = (i32 *)(base + 2 * b8);
    unsigned long long *d_cursor = (unsigned long long *)(base + 2 * b8 + b4);
    GK_HIP(ctx, hipMemsetAsync(d_cursor, 0, 8, ctx->stream));
    int grid = grid_for(ctx, m->capacity, BLOCK);
    GK_BY_SLOT(m, hipLaunchKernelGGL((k_export<W, S>), dim3(grid), dim3(BLOCK), 0, ctx->stream, (const S *)m->slots, m->capacity, m->k == 64 ? 1u : 0u, d_lo, d_hi, d_cnt, d_cursor));
    GK_HIP(ctx, hipGetLastError());
    unsigned long long written = 0;
    GK_HIP(ctx, hipMemcpyAsync(&written, d_cursor, 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipMemcpyAsync(lo, d_lo, cnt * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (hi) GK_HIP(ctx, hipMemcpyAsync(hi, d_hi, cnt * 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipMemcpyAsync(counts, d_cnt, cnt * 4, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (written != cnt) return fail(ctx, GK_E_STATE, "export wrote " + std::to_string(written) + " entries, size says " + std::to_string(cnt));
    return GK_OK;
}

int gk_map_verify(gk_map *m, uint64_t *live, uint64_t *bad_slots, uint64_t *sum_counts, uint64_t *checksum) {
    if (int rc = check_map(m)) return rc;
    gk_ctx *ctx = m->ctx;
    unsigned long long *d = (unsigned long long *)map_scratch(m, 256), h[4] = {0, 0, 0, 0};
    if (!d) return GK_E_CAPACITY;
    GK_HIP(ctx, hipMemsetAsync(d, 0, 32, ctx->stream));
    int grid = grid_for(ctx, m->capacity, BLOCK);
    GK_BY_SLOT(m, hipLaunchKernelGGL((k_verify<W, S>), dim3(grid), dim3(BLOCK), 0, ctx->stream, table_of<W, S>(m), d));
    GK_HIP(ctx, hipGetLastError());
    GK_HIP(ctx, hipMemcpyAsync(h, d, 32, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (live) *live = h[0];
    if (bad_slots) *bad_slots = h[1];
    if (sum_counts) *sum_counts = h[2];
    if (checksum) *checksum = h[3];
    return GK_OK;
}

int gk_map_trim(gk_map *m) {
    if (int rc = check_map_lazy(m)) return rc;
    gk_ctx *ctx = m->ctx;
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
    for (int i = 0; i < 2; i++) { m->stage[i].valid = m->stage[i].armed = false; if (m->stage[i].d) { (void)hipFree(m->stage[i].d); m->stage[i].d = nullptr; m->stage[i].cap = 0; } }
    if (m->d_offsets) { (void)hipFree(m->d_offsets); m->d_offsets = nullptr; m->offsets_bytes = 0; }
    if (m->d_scratch) { (void)hipFree(m->d_scratch); m->d_scratch = nullptr; m->scratch_bytes = 0; }
    part_scratch_free(ctx, m->part);
    m->part = nullptr;
    pool_release(ctx);                  // "release" means back to the device, not parked in the context's pool
    return GK_OK;
}

int gk_map_set_max_batch_keys(gk_map *m, uint64_t keys) {
    if (int rc = check_map_lazy(m)) return rc;
    m->max_batch_keys = keys;
    return GK_OK;
}

int gk_map_stats(gk_map *m, char *json, size_t cap) {
    if (int rc = check_map_lazy(m)) return rc;
    if (!json || cap == 0) return fail(m->ctx, GK_E_INVALID, "null stats buffer");
    int w = snprintf(json, cap,
                     "{\"k\":%d,\"key_words\":%d,\"slot_bytes\":%zu,\"slots\":%llu,\"size\":%llu,\"tombstones\":%llu,"
                     "\"load\":%.6f,\"occurrences\":%llu,\"grows\":%llu,\"last_count_kernel_ms\":%.6f,"
                     "\"last_count_occurrences\":%llu,\"partitioned_launches\":%llu,\"direct_launches\":%llu,"
                     "\"spilled_keys\":%llu,\"failed_segments\":%llu,\"retries_direct\":%llu,\"est_new_distinct_last_batch\":%llu,"
                     "\"noncanonical_keys\":%s,\"repeat_heavy\":%s,\"last_count_host_gap_ms\":%.4f,\"device\":%d,\"cu_count\":%d,"
                     "\"calib_copy_tbps\":%.4f,\"calib_cas_gps\":%.4f}",
                     m->k, m->W, map_slot_bytes(m), (unsigned long long)m->capacity, (unsigned long long)m->size,
                     (unsigned long long)m->tombstones, m->capacity ? (double)m->size / (double)m->capacity : 0.0,
                     (unsigned long long)m->total_occurrences, (unsigned long long)m->grows, m->last_count_ms,
                     (unsigned long long)m->last_count_occ, (unsigned long long)m->part_launches,
                     (unsigned long long)m->direct_launches, (unsigned long long)m->spilled_keys,
                     (unsigned long long)m->failed_segments, (unsigned long long)m->retries_direct, (unsigned long long)m->est_distinct_last,
                     m->dirty ? "true" : "false", m->repeats ? "true" : "false", m->gap_ms, m->ctx->device, m->ctx->cu_count,
                     m->ctx->measured_copy_tbps, m->ctx->measured_cas_gps);
    if (w < 0 || (size_t)w >= cap) return fail(m->ctx, GK_E_CAPACITY, "stats buffer too small");
    return GK_OK;
}

int gk_map_last_phase_ms(gk_map *m, float *ms5) {
    if (int rc = check_map_lazy(m)) return rc;
    if (!ms5) return fail(m->ctx, GK_E_INVALID, "null buffer");
    for (int i = 0; i < 5; i++) ms5[i] = m->phase_ms[i];
    return GK_OK;
}

int gk_map_last_count_kernel(gk_map *m, float *ms, uint64_t *occurrences) {
    if (int rc = check_map_lazy(m)) return rc;
    if (ms) *ms = m->last_count_ms;
    if (occurrences) *occurrences = m->last_count_occ;
    return GK_OK;
}

}  // extern "C"
