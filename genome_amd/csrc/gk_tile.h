// gk_tile.h — LDS staging of `.bin` read records and k-mer window extraction, shared by the
// kernels that stream reads (k_count_reads, k_shard_*).
#pragma once

#include "gk_device.h"

namespace gk {

static constexpr int BLOCK = 256;          // 4 waves of 64
static constexpr int TILE_READS = 64;      // reads staged per LDS tile
static constexpr int TILE_WORDS = 1088;    // 64 reads x 65 B + alignment slack + extraction slack

// Pull the k-mer starting at bit `bitpos` out of the LDS tile (2 bits per base, LSB first — the
// reference's own packing, ArrayDNASeq.apply DNASeq.scala:46-51, so a window IS a bit range).
__device__ __forceinline__ Kmer<1> tile_kmer(const u32 *tile, u32 bitpos, int k, Kmer<1> *) {
    u32 wi = bitpos >> 5, sh = bitpos & 31;
    u64 v0 = (u64)tile[wi] | ((u64)tile[wi + 1] << 32);
    u64 v1 = tile[wi + 2];
    u64 x = (v0 >> sh) | (sh ? (v1 << (64 - sh)) : 0ull);
    return Kmer<1>{x & ((1ull << (2 * k)) - 1)};
}
__device__ __forceinline__ Kmer<2> tile_kmer(const u32 *tile, u32 bitpos, int k, Kmer<2> *) {
    u32 wi = bitpos >> 5, sh = bitpos & 31;
    u64 v0 = (u64)tile[wi] | ((u64)tile[wi + 1] << 32);
    u64 v1 = (u64)tile[wi + 2] | ((u64)tile[wi + 3] << 32);
    u64 v2 = tile[wi + 4];
    u64 lo = (v0 >> sh) | (sh ? (v1 << (64 - sh)) : 0ull);
    u64 hi = (v1 >> sh) | (sh ? (v2 << (64 - sh)) : 0ull);
    return Kmer<2>{lo, hi & low_mask(2 * (k - 32))};
}

// Stage the records of reads [r0, r0+nr) into LDS with 16-byte coalesced loads.  The byte range
// is widened to 16-byte alignment on both sides; an aligned 16-byte block that holds at least one
// valid byte never crosses a page, so the widening cannot fault.  Returns the global byte offset
// that LDS byte 0 corresponds to.
__device__ __forceinline__ u64 stage_tile(u32 *tile, const uint8_t *rec, u64 gb, u64 ge) {
    u64 base = (u64)(uintptr_t)rec;
    u64 a0 = ((base + gb) & ~15ull) - base;             // may be "negative" (wraps) by < 16: fine, same block
    u64 a1 = ((base + ge + 15) & ~15ull) - base;
    u32 nvec = (u32)((a1 - a0) >> 4);
    const uint4 *src = reinterpret_cast<const uint4 *>(rec + (i64)a0);
    uint4 *dst = reinterpret_cast<uint4 *>(tile);
    for (u32 i = threadIdx.x; i < nvec; i += blockDim.x) dst[i] = src[i];
    return a0;
}


// Walk every window of a staged tile: f(kmer) is called once per window by the lane that owns it.
// A wave is cut into 64/G groups of G lanes (G = 64, 32 or 16); each group takes one read and its
// lanes take window positions, so short records (super-k-mers hold ~10 windows) still fill the wave.
// offsets == nullptr: fixed stride.
// max_len: the longest read the caller's buffers were sized for (the declared read_len of a `_dev`
// call, what the record slot can hold for super-k-mers, 255 for host streams whose framing the host
// has walked).  A device record is untrusted input: a length byte above max_len is CLAMPED — LDS
// arrays and key regions downstream are sized from max_len — and reported through *bad
// (-> GK_E_FORMAT), never followed.
struct WindowLimits {
    int max_len;     // bases
    u32 *bad;        // device flag, may be nullptr
    int exact = 0;   // 1: every record must be EXACTLY max_len long (a host stream taken for uniform without walking its framing)
};
__device__ __forceinline__ int record_len(const uint8_t *tb, u32 ro, const WindowLimits &lim) {
    int len = (int)tb[ro];                            // [len:u8]
    if (lim.exact ? len != lim.max_len : len > lim.max_len) {
        len = len > lim.max_len ? lim.max_len : len;
        if (lim.bad) *lim.bad = 1u;
    }
    return len;
}
template <int W, class F>
__device__ __forceinline__ void for_each_window(const u32 *tile, u64 a0, u64 r0, int nr, const u32 *offsets, u32 stride, int k,
                                                int G, WindowLimits lim, F f) {
    const uint8_t *tb = reinterpret_cast<const uint8_t *>(tile);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int per_wave = 64 / G, sub = lane / G, gl = lane & (G - 1);
    for (int r = wave * per_wave + sub; r < nr; r += nwaves * per_wave) {
        const u32 ro = (u32)((offsets ? (u64)offsets[r0 + r] : (r0 + r) * stride) - a0);
        const int nk = record_len(tb, ro, lim) - k + 1;     // reads shorter than k are skipped
        const u32 bit0 = (ro + 1) * 8;
        for (int p = gl; p < nk; p += G) f(tile_kmer(tile, bit0 + 2 * p, k, (Kmer<W> *)nullptr));
    }
}
// same walk, but the callback also gets the tile-local read index and the window position
template <int W, class F>
__device__ __forceinline__ void for_each_window_at(const u32 *tile, u64 a0, u64 r0, int nr, u32 stride, int k, int G,
                                                   WindowLimits lim, F f) {
    const uint8_t *tb = reinterpret_cast<const uint8_t *>(tile);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int per_wave = 64 / G, sub = lane / G, gl = lane & (G - 1);
    for (int r = wave * per_wave + sub; r < nr; r += nwaves * per_wave) {
        const u32 ro = (u32)((r0 + r) * stride - a0);
        const int nk = record_len(tb, ro, lim) - k + 1;
        const u32 bit0 = (ro + 1) * 8;
        for (int p = gl; p < nk; p += G) f(r, p, tile_kmer(tile, bit0 + 2 * p, k, (Kmer<W> *)nullptr));
    }
}

}  // namespace gk
