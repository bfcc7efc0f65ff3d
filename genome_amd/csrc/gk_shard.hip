// gk_shard.hip — owner routing for the PartitionedDNAMap path and the device-side synthetic read
// generator.
//
//   PartitionedDNAMap.partition  S/ds/PartitionedDNAMap.scala:60-63  -> k_shard_count / k_shard_scatter
//     (the reference sends one Akka message per k-mer to `hashCode mod P`; here every k-mer of a
//      read chunk is bucketed by a strand-symmetric minimizer owner in two streaming passes, and
//      the buckets are exchanged by one RCCL all-to-all — see bench.py / genome_amd/partitioned.py)
//   synthetic inputs             SURVEY.md §8d                        -> k_synth_reads
#include <algorithm>
#include <vector>

#include "gk_internal.h"
#include "gk_tile.h"

using namespace gk;

static constexpr int MAX_PARTS = 64;

// pass 1: how many canonical k-mers go to each owner
template <int W>
__global__ __launch_bounds__(BLOCK) void k_shard_count(const uint8_t *__restrict__ rec, u64 nreads, u32 stride, int k, int P,
                                                       WindowLimits lim, unsigned long long *counts) {
    __shared__ __attribute__((aligned(16))) u32 tile[TILE_WORDS];
    __shared__ u32 hist[MAX_PARTS];
    if (threadIdx.x < MAX_PARTS) hist[threadIdx.x] = 0;
    const u64 ntiles = (nreads + TILE_READS - 1) / TILE_READS;
    for (u64 tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const u64 r0 = tl * TILE_READS;
        const int nr = (int)min((u64)TILE_READS, nreads - r0);
        __syncthreads();
        const u64 a0 = stage_tile(tile, rec, r0 * stride, (r0 + nr) * stride);
        __syncthreads();
        for_each_window<W>(tile, a0, r0, nr, nullptr, stride, k, 64, lim, [&](Kmer<W> x) {
            atomicAdd(&hist[owner_of(x, k, P)], 1u);
        });
    }
    __syncthreads();
    if (threadIdx.x < (u32)P && hist[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
}

// pass 2: write each canonical k-mer into its owner's region.  Per tile: histogram in LDS, one
// global atomic per (tile, owner) reserves a contiguous run, lanes fill the run by LDS rank.
template <int W>
__global__ __launch_bounds__(BLOCK) void k_shard_scatter(const uint8_t *__restrict__ rec, u64 nreads, u32 stride, int k, int P,
                                                         WindowLimits lim, unsigned long long *cursors, u64 *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) u32 tile[TILE_WORDS];
    __shared__ u32 hist[MAX_PARTS], rank[MAX_PARTS];
    __shared__ unsigned long long base[MAX_PARTS];
    const u64 ntiles = (nreads + TILE_READS - 1) / TILE_READS;
    for (u64 tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const u64 r0 = tl * TILE_READS;
        const int nr = (int)min((u64)TILE_READS, nreads - r0);
        __syncthreads();
        if (threadIdx.x < MAX_PARTS) { hist[threadIdx.x] = 0; rank[threadIdx.x] = 0; }
        const u64 a0 = stage_tile(tile, rec, r0 * stride, (r0 + nr) * stride);
        __syncthreads();
        for_each_window<W>(tile, a0, r0, nr, nullptr, stride, k, 64, lim, [&](Kmer<W> x) { atomicAdd(&hist[owner_of(x, k, P)], 1u); });
        __syncthreads();
        if (threadIdx.x < (u32)P && hist[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursors[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
        __syncthreads();
        for_each_window<W>(tile, a0, r0, nr, nullptr, stride, k, 64, lim, [&](Kmer<W> x) {
            Kmer<W> y = canonical(x, k);                 // FreqFilter.scala:31-32, done by the sender
            int p = owner_of(x, k, P);                   // same owner for x and rc(x)
            u64 o = base[p] + atomicAdd(&rank[p], 1u);
            if constexpr (W == 1) out[o] = y.lo;
            else { out[2 * o] = y.lo; out[2 * o + 1] = y.hi; }
        });
    }
}

// how many live keys of a partition's table belong to ANOTHER owner under gk_owner_of (must be 0: PartitionedDNAMap.partition,
// PartitionedDNAMap.scala:60-63, names exactly one partition per key)
template <int W, class S>
__global__ __launch_bounds__(BLOCK) void k_count_foreign(Table<W, S> t, int k, int P, int p, unsigned long long *out) {
    unsigned long long bad = 0;
    const u64 ncap = t.capacity();
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < ncap; i += (u64)gridDim.x * BLOCK)
        if (slot_live(&t.slots[i]) && owner_of(slot_key(t.slots, i, t.tagged), k, P) != p) bad++;
    for (int d = 32; d; d >>= 1) bad += __shfl_down(bad, d);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(out, bad);
}

// SplitMix64 in counter form — identical to genome_amd/synth.py
__device__ __forceinline__ u64 splitmix_at(u64 seed, u64 idx) {
    u64 z = seed + (idx + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// one thread per output byte of the record stream [len:u8][ceil(L/4) bytes]
__global__ __launch_bounds__(BLOCK) void k_synth_reads(uint8_t *rec, u64 nreads, int L, int mode, u64 config_id, u64 first_read,
                                                       u64 G, u32 err_thresh24) {
    const u32 stride = 1 + (L + 3) / 4;
    const u64 total = nreads * stride;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (u64)gridDim.x * BLOCK) {
        const u64 rl = i / stride;
        const u32 j = (u32)(i - rl * stride);
        if (j == 0) { rec[i] = (uint8_t)L; continue; }
        const u64 r = rl + first_read;
        u32 byte = 0;
        if (mode == 0) {
            const u64 seed = 0xC0FFEEull ^ config_id;
            for (int q = 0; q < 4; q++) {
                int b = (j - 1) * 4 + q;
                if (b < L) byte |= (u32)(splitmix_at(seed, r * (u64)L + b) >> 62) << (2 * q);
            }
        } else {
            const u64 seed_g = 0xD1CEull ^ config_id, seed_r = 0xBEEFull ^ config_id;
            const u64 bi = r * (u64)(L + 2);
            const u64 start = splitmix_at(seed_r, bi) % (G - L + 1);
            const bool rc = (splitmix_at(seed_r, bi + 1) >> 63) != 0;
            for (int q = 0; q < 4; q++) {
                int b = (j - 1) * 4 + q;
                if (b >= L) break;
                u64 gpos = rc ? start + (u64)(L - 1 - b) : start + (u64)b;
                u32 base = (u32)(splitmix_at(seed_g, gpos) >> 62);
                if (rc) base = 3 - base;
                u64 draw = splitmix_at(seed_r, bi + 2 + b);
                if ((draw >> 40) < (u64)err_thresh24) base = (base + 1 + (u32)((draw & 0xFFFFFFFFull) % 3)) & 3;
                byte |= base << (2 * q);
            }
        }
        rec[i] = (uint8_t)byte;
    }
}

extern "C" {

int gk_owner_of(int k, uint64_t lo, uint64_t hi, int P) {
    if (!k_supported(k) || P <= 0) return -1;
    if (words_for_k(k) == 1) return owner_of(Kmer<1>{lo}, k, P);
    return owner_of(Kmer<2>{lo, hi}, k, P);
}

int gk_map_count_foreign(gk_map *m, int P, int p, uint64_t *foreign) {
    if (!m || !m->ctx) return fail(nullptr, GK_E_INVALID, "null map handle");
    gk_ctx *ctx = m->ctx;
    if (!foreign || P < 1 || p < 0 || p >= P) return fail(ctx, GK_E_INVALID, "gk_map_count_foreign: bad argument");
    GK_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc = map_materialize(m)) return rc;
    unsigned long long *d = (unsigned long long *)map_scratch(m, 256), h = 0;
    if (!d) return GK_E_CAPACITY;
    GK_HIP(ctx, hipMemsetAsync(d, 0, 8, ctx->stream));
    const int grid = (int)std::min<u64>(std::max<u64>((m->capacity + BLOCK - 1) / BLOCK, 1), (u64)ctx->cu_count * 8);
    GK_BY_SLOT(m, hipLaunchKernelGGL((k_count_foreign<W, S>), dim3(grid), dim3(BLOCK), 0, ctx->stream,
                                     Table<W, S>{(S *)m->slots, m->nb2, m->lnb1, m->k == 64 ? 1u : 0u}, m->k, P, p, d));
    GK_HIP(ctx, hipGetLastError());
    GK_HIP(ctx, hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *foreign = h;
    return GK_OK;
}

int gk_shard_reads_dev(gk_ctx *ctx, int k, const void *dev_records, uint64_t nreads, int read_len, int P,
                       void *dev_keys_out, uint64_t keys_cap, uint64_t *counts_host) {
    if (!ctx) return fail(nullptr, GK_E_INVALID, "null ctx");
    if (!k_supported(k)) return fail(ctx, GK_E_UNSUPPORTED_K, "k=" + std::to_string(k) + " unsupported");
    if (P < 1 || P > MAX_PARTS) return fail(ctx, GK_E_INVALID, "P must be 1.." + std::to_string(MAX_PARTS));
    if (!counts_host || (!dev_records && nreads)) return fail(ctx, GK_E_INVALID, "null argument");
    if (read_len < 0 || read_len > 255) return fail(ctx, GK_E_FORMAT, "read_len must be 0..255");
    GK_HIP(ctx, hipSetDevice(ctx->device));
    for (int p = 0; p < P; p++) counts_host[p] = 0;
    const u64 nk = read_len >= k ? (u64)(read_len - k + 1) : 0;
    if (nreads == 0 || nk == 0) return GK_OK;
    if (nreads * nk > keys_cap) return fail(ctx, GK_E_CAPACITY, "keys_cap too small: need " + std::to_string(nreads * nk));
    if (!dev_keys_out) return fail(ctx, GK_E_INVALID, "null key buffer");
    const int W = words_for_k(k);
    const u32 stride = 1 + (read_len + 3) / 4;
    unsigned long long *d_cnt = nullptr;
    GK_HIP(ctx, hipMalloc((void **)&d_cnt, 2 * MAX_PARTS * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d_cnt, 0, 2 * MAX_PARTS * sizeof(unsigned long long), ctx->stream);
    const u64 ntiles = (nreads + TILE_READS - 1) / TILE_READS;
    const int grid = (int)std::min<u64>(ntiles, (u64)ctx->cu_count * 8);
    const uint8_t *rec = (const uint8_t *)dev_records;
    if (e == hipSuccess) {
        if (W == 1) hipLaunchKernelGGL(k_shard_count<1>, dim3(grid), dim3(BLOCK), 0, ctx->stream, rec, nreads, stride, k, P, WindowLimits{read_len, ctx->d_flags}, d_cnt);
        else hipLaunchKernelGGL(k_shard_count<2>, dim3(grid), dim3(BLOCK), 0, ctx->stream, rec, nreads, stride, k, P, WindowLimits{read_len, ctx->d_flags}, d_cnt);
        e = hipGetLastError();
    }
    unsigned long long h_cnt[MAX_PARTS], h_cur[MAX_PARTS];
    if (e == hipSuccess) e = hipMemcpyAsync(h_cnt, d_cnt, P * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) {
        unsigned long long acc = 0;
        for (int p = 0; p < P; p++) { h_cur[p] = acc; acc += h_cnt[p]; counts_host[p] = h_cnt[p]; }
        e = hipMemcpyAsync(d_cnt + MAX_PARTS, h_cur, P * sizeof(unsigned long long), hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess) {
        if (W == 1) hipLaunchKernelGGL(k_shard_scatter<1>, dim3(grid), dim3(BLOCK), 0, ctx->stream, rec, nreads, stride, k, P, WindowLimits{read_len, nullptr}, d_cnt + MAX_PARTS, (u64 *)dev_keys_out);
        else hipLaunchKernelGGL(k_shard_scatter<2>, dim3(grid), dim3(BLOCK), 0, ctx->stream, rec, nreads, stride, k, P, WindowLimits{read_len, nullptr}, d_cnt + MAX_PARTS, (u64 *)dev_keys_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_cnt);
    if (e != hipSuccess) return hip_fail(ctx, e, "gk_shard_reads_dev");
    return ctx_check_format(ctx);
}

int gk_synth_reads_dev(gk_ctx *ctx, void *dev_records, uint64_t nreads, int read_len, int mode, uint64_t config_id,
                       uint64_t first_read, uint64_t genome_len, uint32_t err_thresh24) {
    if (!ctx) return fail(nullptr, GK_E_INVALID, "null ctx");
    if (!dev_records && nreads) return fail(ctx, GK_E_INVALID, "null record buffer");
    if (read_len < 1 || read_len > 255) return fail(ctx, GK_E_FORMAT, "read_len must be 1..255");
    if (mode != 0 && mode != 1) return fail(ctx, GK_E_INVALID, "mode must be 0 (U) or 1 (G)");
    if (mode == 1 && genome_len < (uint64_t)read_len) return fail(ctx, GK_E_INVALID, "genome shorter than a read");
    if (nreads == 0) return GK_OK;
    GK_HIP(ctx, hipSetDevice(ctx->device));
    const u64 total = nreads * (1 + (read_len + 3) / 4);
    const int grid = (int)std::min<u64>((total + BLOCK - 1) / BLOCK, (u64)ctx->cu_count * 8);
    hipLaunchKernelGGL(k_synth_reads, dim3(grid), dim3(BLOCK), 0, ctx->stream, (uint8_t *)dev_records, nreads, read_len, mode,
                       config_id, first_read, genome_len, err_thresh24);
    GK_HIP(ctx, hipGetLastError());
    GK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GK_OK;
}

}  // extern "C"
