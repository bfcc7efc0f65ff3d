// gk_scan.h — exclusive scan of u32 counts into u64 offsets on the device (three small kernels); shared by the paired-end stage
// (CSR of the getAll batch, in-edge lists) and the graph build (bucket regions of the minimizer-bucketed table).
#pragma once

#include "gk_internal.h"

using namespace gk;

// ---------------------------------------------------------------------------------------------
// small device utilities: exclusive scan of u32 counts into u64 offsets (n + 1 entries)
// ---------------------------------------------------------------------------------------------
static constexpr u32 SCAN_CHUNK = 4096;
static __global__ __launch_bounds__(256) void k_scan_sums(const u32 *__restrict__ in, u64 n, u64 *__restrict__ sums) {
    __shared__ u64 s_w[4];
    const u64 c = blockIdx.x;
    u64 v = 0;
    for (u32 j = threadIdx.x; j < SCAN_CHUNK; j += 256) { const u64 i = c * SCAN_CHUNK + j; if (i < n) v += in[i]; }
    for (int d = 32; d; d >>= 1) v += __shfl_down(v, d);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) sums[c] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
static __global__ __launch_bounds__(1024) void k_scan_sums_excl(u64 *sums, u64 nchunks) {      // one workgroup; sums[nchunks] = total
    __shared__ u64 s_sum[1024];
    const u64 per = (nchunks + 1023) / 1024, c0 = threadIdx.x * per, c1 = min(c0 + per, nchunks);
    u64 sum = 0;
    for (u64 c = c0; c < c1; c++) sum += sums[c];
    s_sum[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) { u64 run = 0; for (int i = 0; i < 1024; i++) { const u64 v = s_sum[i]; s_sum[i] = run; run += v; } sums[nchunks] = run; }
    __syncthreads();
    u64 run = s_sum[threadIdx.x];
    for (u64 c = c0; c < c1; c++) { const u64 v = sums[c]; sums[c] = run; run += v; }
}
static __global__ __launch_bounds__(256) void k_scan_fill(const u32 *__restrict__ in, u64 n, const u64 *__restrict__ sums, u64 nchunks, unsigned long long *__restrict__ out) {
    __shared__ u64 s_w[4];
    const u64 c = blockIdx.x;
    constexpr u32 PER = SCAN_CHUNK / 256;                    // consecutive elements per thread
    const u64 i0 = c * SCAN_CHUNK + (u64)threadIdx.x * PER;
    u32 loc[PER];
    u64 tsum = 0;
#pragma unroll
    for (u32 j = 0; j < PER; j++) { loc[j] = i0 + j < n ? in[i0 + j] : 0u; tsum += loc[j]; }
    u64 inc = tsum;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int d = 1; d < 64; d <<= 1) { const u64 t = __shfl_up(inc, d); if (lane >= d) inc += t; }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    u64 run = sums[c] + inc - tsum;
    for (int w = 0; w < wave; w++) run += s_w[w];
#pragma unroll
    for (u32 j = 0; j < PER; j++) { if (i0 + j < n) out[i0 + j] = run; run += loc[j]; }
    if (c == nchunks - 1 && threadIdx.x == 0) out[n] = sums[nchunks];
}
// d_out[0..n] = exclusive prefix of d_in[0..n); d_sums: scratch of (n / 4096 + 2) u64.  Stream-ordered.
static hipError_t scan_counts(gk_ctx *ctx, const u32 *d_in, u64 n, unsigned long long *d_out, u64 *d_sums) {
    if (n == 0) return hipMemsetAsync(d_out, 0, 8, ctx->stream);
    const u64 nchunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)nchunks), dim3(256), 0, ctx->stream, d_in, n, d_sums);
    hipLaunchKernelGGL(k_scan_sums_excl, dim3(1), dim3(1024), 0, ctx->stream, d_sums, nchunks);
    hipLaunchKernelGGL(k_scan_fill, dim3((unsigned)nchunks), dim3(256), 0, ctx->stream, d_in, n, d_sums, nchunks, d_out);
    return hipGetLastError();
}

