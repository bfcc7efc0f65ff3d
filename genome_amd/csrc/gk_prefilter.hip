// gk_prefilter.hip — EXACT two-pass singleton pre-filter in front of the k-mer table
// (the reference's analogue is the unused S/ds/BloomFilter.scala:17-70).
//
// At sequencing error rates most distinct k-mers are errors seen once; FreqFilter drops them anyway
// (deleteAll(v < rounds), S/data/FreqFilter.scala:55), but only after each has cost a 16/32-byte
// table slot.  The filter is an array of 2-bit saturating counters (0, 1, "2 or more"), one counter
// per hashed bucket, 0.25 byte each:
//   pass 1  gk_prefilter_add_reads*      every window of every read bumps its bucket's counter
//   pass 2  gk_map_count_reads_prefiltered*   a window is inserted into the table only if its
//                                        bucket says "2 or more"
// Exactness.  Pass 1 sees ALL occurrences before pass 2 starts, so a k-mer with true count >= 2 has a
// saturated bucket and every one of its occurrences is admitted: its table count is its true count.
// A k-mer with true count 1 is admitted only when it shares a bucket with another k-mer (a false
// positive); it then sits in the table with count 1.  Hence for rounds >= 2
//     filter_lt(rounds) after prefiltered counting  ==  filter_lt(rounds) after plain counting,
// bit for bit, whatever the filter size; only the memory high-water mark differs.  (For rounds <= 1
// the filter must not be used; the host mirror refuses.)
#include <algorithm>
#include <string>
#include <vector>

#include "gk_internal.h"
#include "gk_tile.h"

using namespace gk;

struct gk_prefilter {
    gk_ctx *ctx = nullptr;
    int k = 0, W = 1;
    u64 nbuckets = 0, nwords = 0;
    u32 *words = nullptr;                 // 16 two-bit counters per word
    u64 *keybuf = nullptr;                // admitted canonical keys of one pass-2 chunk
    u64 keybuf_keys = 0;
    unsigned long long *d_cursor = nullptr;   // [0] admitted keys, [1] windows seen
    void *d_stage = nullptr;              // host `.bin` streams: staged records + record offsets
    size_t stage_bytes = 0;
    u32 *d_offsets = nullptr;
    size_t offsets_bytes = 0;
    u64 windows_added = 0;
};

// bucket of a canonical k-mer: a second mix of the slot hash, so that the filter's collisions are
// independent of the table's
template <int W> __device__ __forceinline__ u64 pf_bucket(Kmer<W> y, u64 nbuckets) {
    const u64 h = mix64(slot_hash(y) ^ 0x9e3779b97f4a7c15ULL);
    return __umul64hi(h, nbuckets);
}

static constexpr int PF_PARK_BYTES = 64 << 10;   // pass 2 parks a tile's admitted keys in LDS: reads per tile = what fits, at most 64

// pass 1: counter 0 -> 1 -> 3 (bit 0 = seen, bit 1 = seen again).  The plain load is only a hint
// (bits are only ever set): it saves the atomics of the common "already saturated" case.
template <int W>
__global__ __launch_bounds__(BLOCK) void k_pf_add(const uint8_t *__restrict__ rec, u64 nreads, const u32 *__restrict__ offsets, u32 stride,
                                                  int k, WindowLimits lim, u32 *words, u64 nbuckets, unsigned long long *seen) {
    __shared__ __attribute__((aligned(16))) u32 tile[TILE_WORDS];
    __shared__ u32 s_occ;
    if (threadIdx.x == 0) s_occ = 0;
    u32 occ = 0;
    const u64 ntiles = (nreads + TILE_READS - 1) / TILE_READS;
    for (u64 tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const u64 r0 = tl * TILE_READS;
        const int nr = (int)min((u64)TILE_READS, nreads - r0);
        const u64 gb = offsets ? (u64)offsets[r0] : r0 * stride;
        const u64 ge = offsets ? (u64)offsets[r0 + nr] : (r0 + nr) * stride;
        __syncthreads();
        const u64 a0 = stage_tile(tile, rec, gb, ge);
        __syncthreads();
        for_each_window<W>(tile, a0, r0, nr, offsets, stride, k, 64, lim, [&](Kmer<W> x) {
            const u64 b = pf_bucket(canonical(x, k), nbuckets);
            u32 *w = words + (b >> 4);
            const u32 sh = (u32)(b & 15) * 2;
            const u32 c = (*w >> sh) & 3u;
            if (c == 0) {
                const u32 old = atomicOr(w, 1u << sh);
                if ((old >> sh) & 1u) atomicOr(w, 2u << sh);
            } else if (c == 1) {
                atomicOr(w, 2u << sh);
            }
            occ++;
        });
    }
    atomicAdd(&s_occ, occ);
    __syncthreads();
    if (threadIdx.x == 0 && s_occ) atomicAdd(seen, (unsigned long long)s_occ);
}

// pass 2: the canonical keys whose bucket is saturated, packed into `out` (W words per key).  A tile's
// admitted keys are parked in LDS, one global atomic per tile reserves their run, a coalesced copy
// writes them.
template <int W>
__global__ __launch_bounds__(BLOCK) void k_pf_select(const uint8_t *__restrict__ rec, u64 nreads, const u32 *__restrict__ offsets, u32 stride,
                                                     int k, WindowLimits lim, int rs /* reads per tile */, const u32 *__restrict__ words, u64 nbuckets, u64 *__restrict__ out, u64 out_cap,
                                                     unsigned long long *cursor /* [0] admitted, [1] windows */, u32 *overflow) {
    __shared__ __attribute__((aligned(16))) u32 tile[TILE_WORDS];
    extern __shared__ u64 park[];                 // [rs * max windows per read * W]
    __shared__ u32 s_n, s_occ;
    __shared__ unsigned long long s_base;
    if (threadIdx.x == 0) s_occ = 0;
    u32 occ = 0;
    const u64 ntiles = (nreads + rs - 1) / rs;
    for (u64 tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const u64 r0 = tl * rs;
        const int nr = (int)min((u64)rs, nreads - r0);
        const u64 gb = offsets ? (u64)offsets[r0] : r0 * stride;
        const u64 ge = offsets ? (u64)offsets[r0 + nr] : (r0 + nr) * stride;
        __syncthreads();
        if (threadIdx.x == 0) s_n = 0;
        const u64 a0 = stage_tile(tile, rec, gb, ge);
        __syncthreads();
        for_each_window<W>(tile, a0, r0, nr, offsets, stride, k, 64, lim, [&](Kmer<W> x) {
            const Kmer<W> y = canonical(x, k);
            const u64 b = pf_bucket(y, nbuckets);
            occ++;
            if (((words[b >> 4] >> ((u32)(b & 15) * 2)) & 3u) == 3u) {
                const u32 at = atomicAdd(&s_n, 1u);
                if constexpr (W == 1) park[at] = y.lo;
                else { park[2 * at] = y.lo; park[2 * at + 1] = y.hi; }
            }
        });
        __syncthreads();
        const u32 n = s_n;
        if (threadIdx.x == 0 && n) s_base = atomicAdd(&cursor[0], (unsigned long long)n);
        __syncthreads();
        if (n) {
            const u64 base = s_base;
            if (base + n > out_cap) { if (threadIdx.x == 0) *overflow = 1; }
            else for (u32 i = threadIdx.x; i < n * W; i += BLOCK) out[base * W + i] = park[i];
        }
    }
    atomicAdd(&s_occ, occ);
    __syncthreads();
    if (threadIdx.x == 0 && s_occ) atomicAdd(&cursor[1], (unsigned long long)s_occ);
}

__global__ __launch_bounds__(BLOCK) void k_pf_stats(const u32 *__restrict__ words, u64 nwords, unsigned long long *out /* [once, twice+] */) {
    u32 once = 0, twice = 0;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < nwords; i += (u64)gridDim.x * BLOCK) {
        const u32 w = words[i];
        const u32 lo = w & 0x55555555u, hi = (w >> 1) & 0x55555555u;
        twice += __popc(lo & hi);
        once += __popc(lo & ~hi);
    }
    for (int d = 32; d; d >>= 1) { once += __shfl_down(once, d); twice += __shfl_down(twice, d); }
    if ((threadIdx.x & 63) == 0) {
        if (once) atomicAdd(&out[0], (unsigned long long)once);
        if (twice) atomicAdd(&out[1], (unsigned long long)twice);
    }
}

namespace {

int pf_check(gk_prefilter *pf) { return pf && pf->ctx ? GK_OK : GK_E_INVALID; }

int pf_launch_add(gk_prefilter *pf, const uint8_t *d_rec, u64 nreads, const u32 *d_off, u32 stride, int max_len = 255) {
    gk_ctx *ctx = pf->ctx;
    const u64 ntiles = (nreads + TILE_READS - 1) / TILE_READS;
    const int grid = (int)std::min<u64>(std::max<u64>(ntiles, 1), (u64)ctx->cu_count * 8);
    if (pf->W == 1)
        hipLaunchKernelGGL(k_pf_add<1>, dim3(grid), dim3(BLOCK), 0, ctx->stream, d_rec, nreads, d_off, stride, pf->k, WindowLimits{max_len, ctx->d_flags}, pf->words, pf->nbuckets, pf->d_cursor + 1);
    else
        hipLaunchKernelGGL(k_pf_add<2>, dim3(grid), dim3(BLOCK), 0, ctx->stream, d_rec, nreads, d_off, stride, pf->k, WindowLimits{max_len, ctx->d_flags}, pf->words, pf->nbuckets, pf->d_cursor + 1);
    GK_HIP(ctx, hipGetLastError());
    return GK_OK;
}

// pass 2 for one chunk of records whose windows fit the key buffer; feeds the admitted keys to the table
int pf_select_and_insert(gk_prefilter *pf, gk_map *m, const uint8_t *d_rec, u64 nreads, const u32 *d_off, u32 stride, u64 max_windows,
                         u64 max_windows_per_read, u64 *admitted_total, int max_len = 255) {
    gk_ctx *ctx = pf->ctx;
    if (pf->keybuf_keys < max_windows) {
        if (pf->keybuf) GK_HIP(ctx, hipFree(pf->keybuf));
        pf->keybuf = nullptr; pf->keybuf_keys = 0;
        GK_HIP(ctx, hipMalloc((void **)&pf->keybuf, std::max<u64>(max_windows, 1) * 8 * pf->W));
        pf->keybuf_keys = max_windows;
    }
    GK_HIP(ctx, hipMemsetAsync(pf->d_cursor, 0, 8, ctx->stream));                     // admitted
    u32 *d_ovf = reinterpret_cast<u32 *>(pf->d_cursor + 2);
    GK_HIP(ctx, hipMemsetAsync(d_ovf, 0, 4, ctx->stream));
    const u64 per_read = std::max<u64>(1, max_windows_per_read);
    const int rs = (int)std::max<u64>(1, std::min<u64>(TILE_READS, (u64)PF_PARK_BYTES / (8ull * pf->W) / per_read));
    const u64 ntiles = (nreads + rs - 1) / rs;
    const int grid = (int)std::min<u64>(std::max<u64>(ntiles, 1), (u64)ctx->cu_count * 8);
    const size_t lds = (size_t)rs * per_read * 8 * pf->W;
    if (pf->W == 1)
        hipLaunchKernelGGL(k_pf_select<1>, dim3(grid), dim3(BLOCK), lds, ctx->stream, d_rec, nreads, d_off, stride, pf->k, WindowLimits{max_len, ctx->d_flags}, rs, pf->words, pf->nbuckets,
                           pf->keybuf, pf->keybuf_keys, pf->d_cursor, d_ovf);
    else
        hipLaunchKernelGGL(k_pf_select<2>, dim3(grid), dim3(BLOCK), lds, ctx->stream, d_rec, nreads, d_off, stride, pf->k, WindowLimits{max_len, ctx->d_flags}, rs, pf->words, pf->nbuckets,
                           pf->keybuf, pf->keybuf_keys, pf->d_cursor, d_ovf);
    GK_HIP(ctx, hipGetLastError());
    unsigned long long h[3] = {0, 0, 0};
    GK_HIP(ctx, hipMemcpyAsync(h, pf->d_cursor, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if ((u32)h[2]) return fail(ctx, GK_E_CAPACITY, "prefilter: admitted keys exceed the key buffer (internal sizing error)");
    if (int rc = ctx_check_format(ctx)) return rc;
    *admitted_total += h[0];
    if (h[0]) { if (int rc = map_insert_keys_dev(m, pf->keybuf, h[0], false)) return rc; }     // canonical by construction
    return GK_OK;
}

int pf_ensure_stage(gk_prefilter *pf, size_t bytes, size_t noffs) {
    gk_ctx *ctx = pf->ctx;
    if (pf->stage_bytes < bytes + 64) {
        if (pf->d_stage) GK_HIP(ctx, hipFree(pf->d_stage));
        pf->d_stage = nullptr; pf->stage_bytes = 0;
        GK_HIP(ctx, hipMalloc(&pf->d_stage, bytes + 64));
        pf->stage_bytes = bytes + 64;
    }
    if (pf->offsets_bytes < noffs * sizeof(u32)) {
        if (pf->d_offsets) GK_HIP(ctx, hipFree(pf->d_offsets));
        pf->d_offsets = nullptr; pf->offsets_bytes = 0;
        GK_HIP(ctx, hipMalloc((void **)&pf->d_offsets, noffs * sizeof(u32)));
        pf->offsets_bytes = noffs * sizeof(u32);
    }
    return GK_OK;
}

// Walk a host `.bin` stream (PairedEndData.scala:24-31) in chunks bounded in bytes and in windows;
// f(d_records, nreads_in_chunk, d_offsets or nullptr, fixed stride or 0, windows_in_chunk) runs once per staged chunk.
template <class F>
int pf_for_each_host_chunk(gk_prefilter *pf, const uint8_t *bin, size_t nbytes, u64 nreads, u64 max_windows, F f) {
    gk_ctx *ctx = pf->ctx;
    const size_t MAX_STAGE = 256u << 20;
    size_t pos = 0;
    u64 r = 0;
    std::vector<u32> offs;
    while (r < nreads) {
        const size_t chunk_begin = pos;
        const u64 r_begin = r;
        u64 occ = 0;
        offs.clear();
        // a run of equal-length records: one byte compare per record, no offset table (as in gk_map_count_reads)
        bool fast_prefix = false;
        u32 fixed_stride = 0;
        if (pos < nbytes) {
            const int len0 = bin[pos];
            const size_t rb0 = 1 + (size_t)(len0 + 3) / 4;
            const u64 nk0 = len0 >= pf->k ? (u64)(len0 - pf->k + 1) : 0;
            u64 cap = std::min<u64>(nreads - r, (nbytes - pos) / rb0);
            cap = std::min<u64>(cap, std::max<u64>(1, MAX_STAGE / rb0));
            if (nk0 && max_windows != ~0ull) cap = std::min<u64>(cap, std::max<u64>(1, max_windows / nk0));
            const uint8_t *p0 = bin + pos;
            u64 run = 0;
            while (run < cap && p0[run * rb0] == (uint8_t)len0) run++;
            if (run >= 4096 || (run == nreads - r && run > 0)) {
                fast_prefix = true;
                fixed_stride = (u32)rb0;
                pos += run * rb0;
                occ = run * nk0;
                r += run;
            }
        }
        while (!fast_prefix && r < nreads) {
            if (pos >= nbytes) return fail(ctx, GK_E_FORMAT, "truncated .bin stream: record " + std::to_string(r) + " starts past the end");
            const int len = bin[pos];
            const size_t rb = 1 + (size_t)(len + 3) / 4;
            if (pos + rb > nbytes) return fail(ctx, GK_E_FORMAT, "truncated .bin stream inside record " + std::to_string(r));
            const u64 nk = len >= pf->k ? (u64)(len - pf->k + 1) : 0;
            if (r > r_begin && (pos + rb - chunk_begin > MAX_STAGE || occ + nk > max_windows)) break;
            offs.push_back((u32)(pos - chunk_begin));
            pos += rb; occ += nk; r++;
        }
        if (!fast_prefix) offs.push_back((u32)(pos - chunk_begin));
        const size_t cbytes = pos - chunk_begin;
        if (int rc = pf_ensure_stage(pf, cbytes, std::max<size_t>(offs.size(), 1))) return rc;
        GK_HIP(ctx, hipMemcpyAsync(pf->d_stage, bin + chunk_begin, cbytes, hipMemcpyHostToDevice, ctx->stream));
        if (!fast_prefix) GK_HIP(ctx, hipMemcpyAsync(pf->d_offsets, offs.data(), offs.size() * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
        if (int rc = f((const uint8_t *)pf->d_stage, r - r_begin, fast_prefix ? nullptr : (const u32 *)pf->d_offsets, fixed_stride, occ)) return rc;
        GK_HIP(ctx, hipStreamSynchronize(ctx->stream));      // `offs` and the staging buffer are reused
    }
    return GK_OK;
}

constexpr u64 PF_CHUNK_WINDOWS = 1ull << 28;     // pass-2 key buffer: at most 2 GiB (k <= 31) / 4 GiB of admitted keys

}  // namespace

extern "C" {

int gk_prefilter_create(gk_ctx *ctx, int k, uint64_t expected_distinct, gk_prefilter **out) {
    if (!ctx || !out) return fail(ctx, GK_E_INVALID, "gk_prefilter_create: null argument");
    *out = nullptr;
    if (!k_supported(k)) return fail(ctx, GK_E_UNSUPPORTED_K, "k=" + std::to_string(k) + " unsupported");
    GK_HIP(ctx, hipSetDevice(ctx->device));
    gk_prefilter *pf = new gk_prefilter();
    pf->ctx = ctx; pf->k = k; pf->W = k <= 32 ? 1 : 2;
    // 4 buckets per expected distinct k-mer (1 byte per k-mer): a singleton shares its bucket with
    // another k-mer with probability ~ 1 - exp(-1/4) = 22 %
    pf->nbuckets = std::max<u64>(expected_distinct, 1ull << 16) * 4;
    pf->nwords = (pf->nbuckets + 15) / 16;
    hipError_t e = hipMalloc((void **)&pf->words, pf->nwords * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&pf->d_cursor, 4 * sizeof(unsigned long long));
    if (e != hipSuccess) {
        if (pf->words) (void)hipFree(pf->words);
        delete pf;
        (void)hipGetLastError();
        return fail(ctx, GK_E_CAPACITY, "gk_prefilter_create: " + std::string(hipGetErrorString(e)) + " for " + std::to_string(pf->nwords * 4) + " bytes");
    }
    GK_HIP(ctx, hipMemsetAsync(pf->words, 0, pf->nwords * 4, ctx->stream));
    GK_HIP(ctx, hipMemsetAsync(pf->d_cursor, 0, 4 * sizeof(unsigned long long), ctx->stream));
    // one read of 255 bases at k = 2 has 254 windows: the parking area never needs more than max(PF_PARK_BYTES, one read)
    GK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pf_select<1>), hipFuncAttributeMaxDynamicSharedMemorySize, PF_PARK_BYTES));
    GK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pf_select<2>), hipFuncAttributeMaxDynamicSharedMemorySize, PF_PARK_BYTES));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = pf;
    return GK_OK;
}

void gk_prefilter_destroy(gk_prefilter *pf) {
    if (!pf) return;
    gk_ctx *ctx = pf->ctx;
    if (pf->ctx) (void)hipSetDevice(pf->ctx->device);
    if (pf->words) (void)hipFree(pf->words);
    if (pf->keybuf) (void)hipFree(pf->keybuf);
    if (pf->d_cursor) (void)hipFree(pf->d_cursor);
    if (pf->d_stage) (void)hipFree(pf->d_stage);
    if (pf->d_offsets) (void)hipFree(pf->d_offsets);
    delete pf;
}

int gk_prefilter_add_reads_dev(gk_prefilter *pf, const void *dev_records, uint64_t nreads, int read_len) {
    if (pf_check(pf)) return GK_E_INVALID;
    gk_ctx *ctx = pf->ctx;
    if (!dev_records && nreads) return fail(ctx, GK_E_INVALID, "gk_prefilter_add_reads_dev: null records");
    if (read_len < 0 || read_len > 255) return fail(ctx, GK_E_FORMAT, "read_len must be 0..255 (one length byte per record)");
    GK_HIP(ctx, hipSetDevice(ctx->device));
    if (nreads == 0 || read_len < pf->k) return GK_OK;
    if (int rc = pf_launch_add(pf, (const uint8_t *)dev_records, nreads, nullptr, 1 + (read_len + 3) / 4, read_len)) return rc;
    if (int rc = ctx_check_format(ctx)) return rc;
    pf->windows_added += nreads * (u64)(read_len - pf->k + 1);
    return GK_OK;
}

int gk_prefilter_add_reads(gk_prefilter *pf, const uint8_t *bin, size_t nbytes, uint64_t nreads) {
    if (pf_check(pf)) return GK_E_INVALID;
    gk_ctx *ctx = pf->ctx;
    if (!bin && nreads) return fail(ctx, GK_E_INVALID, "gk_prefilter_add_reads: null stream");
    GK_HIP(ctx, hipSetDevice(ctx->device));
    return pf_for_each_host_chunk(pf, bin, nbytes, nreads, ~0ull, [&](const uint8_t *d_rec, u64 n, const u32 *d_off, u32 stride, u64 occ) {
        pf->windows_added += occ;
        return pf_launch_add(pf, d_rec, n, d_off, stride);
    });
}

int gk_map_count_reads_prefiltered_dev(gk_map *m, gk_prefilter *pf, const void *dev_records, uint64_t nreads, int read_len,
                                       uint64_t *occurrences, uint64_t *admitted) {
    if (occurrences) *occurrences = 0;
    if (admitted) *admitted = 0;
    if (pf_check(pf) || !m) return GK_E_INVALID;
    gk_ctx *ctx = pf->ctx;
    if (gk_map_k(m) != pf->k) return fail(ctx, GK_E_KLEN, "prefilter and map were created for different k");
    if (!dev_records && nreads) return fail(ctx, GK_E_INVALID, "gk_map_count_reads_prefiltered_dev: null records");
    if (read_len < 0 || read_len > 255) return fail(ctx, GK_E_FORMAT, "read_len must be 0..255 (one length byte per record)");
    GK_HIP(ctx, hipSetDevice(ctx->device));
    if (nreads == 0 || read_len < pf->k) return GK_OK;
    const u32 stride = 1 + (read_len + 3) / 4;
    const u64 nk = (u64)(read_len - pf->k + 1);
    const u64 chunk_reads = std::max<u64>(1, PF_CHUNK_WINDOWS / nk);
    GK_HIP(ctx, hipMemsetAsync(pf->d_cursor + 1, 0, 8, ctx->stream));
    u64 adm = 0;
    for (u64 done = 0; done < nreads; done += chunk_reads) {
        const u64 n = std::min(chunk_reads, nreads - done);
        if (int rc = pf_select_and_insert(pf, m, (const uint8_t *)dev_records + done * stride, n, nullptr, stride, n * nk, nk, &adm, read_len)) return rc;
    }
    if (occurrences) *occurrences = nreads * nk;
    if (admitted) *admitted = adm;
    return GK_OK;
}

int gk_map_count_reads_prefiltered(gk_map *m, gk_prefilter *pf, const uint8_t *bin, size_t nbytes, uint64_t nreads,
                                   uint64_t *occurrences, uint64_t *admitted) {
    if (occurrences) *occurrences = 0;
    if (admitted) *admitted = 0;
    if (pf_check(pf) || !m) return GK_E_INVALID;
    gk_ctx *ctx = pf->ctx;
    if (gk_map_k(m) != pf->k) return fail(ctx, GK_E_KLEN, "prefilter and map were created for different k");
    if (!bin && nreads) return fail(ctx, GK_E_INVALID, "gk_map_count_reads_prefiltered: null stream");
    GK_HIP(ctx, hipSetDevice(ctx->device));
    u64 adm = 0, occ_total = 0;
    int rc = pf_for_each_host_chunk(pf, bin, nbytes, nreads, PF_CHUNK_WINDOWS, [&](const uint8_t *d_rec, u64 n, const u32 *d_off, u32 stride, u64 occ) {
        occ_total += occ;
        const u64 per_read = stride ? occ / std::max<u64>(n, 1) : (u64)std::max(1, 255 - pf->k + 1);
        return occ ? pf_select_and_insert(pf, m, d_rec, n, d_off, stride, occ, per_read, &adm) : GK_OK;
    });
    if (rc) return rc;
    if (occurrences) *occurrences = occ_total;
    if (admitted) *admitted = adm;
    return GK_OK;
}

int gk_prefilter_stats(gk_prefilter *pf, uint64_t *buckets, uint64_t *seen_once, uint64_t *seen_twice_or_more, uint64_t *windows_added) {
    if (pf_check(pf)) return GK_E_INVALID;
    gk_ctx *ctx = pf->ctx;
    GK_HIP(ctx, hipSetDevice(ctx->device));
    GK_HIP(ctx, hipMemsetAsync(pf->d_cursor + 2, 0, 16, ctx->stream));
    const int grid = (int)std::min<u64>(std::max<u64>((pf->nwords + BLOCK - 1) / BLOCK, 1), (u64)ctx->cu_count * 8);
    hipLaunchKernelGGL(k_pf_stats, dim3(grid), dim3(BLOCK), 0, ctx->stream, pf->words, pf->nwords, pf->d_cursor + 2);
    GK_HIP(ctx, hipGetLastError());
    unsigned long long h[2] = {0, 0};
    GK_HIP(ctx, hipMemcpyAsync(h, pf->d_cursor + 2, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (buckets) *buckets = pf->nbuckets;
    if (seen_once) *seen_once = h[0];
    if (seen_twice_or_more) *seen_twice_or_more = h[1];
    if (windows_added) *windows_added = pf->windows_added;
    return GK_OK;
}

}  // extern "C"
