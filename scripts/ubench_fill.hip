// ubench_fill.hip — how fast can this card WRITE?  Variants of a 3 GiB fill (the size of C2's table): plain / nontemporal
// stores, grid-stride lanes vs one 32 KiB block per workgroup step (k_seg_insert's write-back shape), grid sizes.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench_fill.hip -o scripts/ubench_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store(uint4 v, uint4 *p) { v4u x = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(x, reinterpret_cast<v4u *>(p)); }
__device__ __forceinline__ uint4 nt_load(const uint4 *p) { v4u x = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p)); return make_uint4(x.x, x.y, x.z, x.w); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void fill_stride(uint4 *out, u64 n) {
    const uint4 v = make_uint4(~0u, ~0u, 0u, 0u);
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        if (NT) nt_store(v, &out[i]); else out[i] = v;
    }
}
// one workgroup writes whole 32 KiB blocks (2048 x 16 B), block index grid-strided
template <bool NT, int THREADS>
__global__ __launch_bounds__(THREADS) void fill_blocks(uint4 *out, u64 nblocks) {
    const uint4 v = make_uint4(~0u, ~0u, 0u, 0u);
    for (u64 b = blockIdx.x; b < nblocks; b += gridDim.x) {
        uint4 *p = out + b * 2048;
#pragma unroll
        for (int i = threadIdx.x; i < 2048; i += THREADS) {
            if (NT) nt_store(v, &p[i]); else p[i] = v;
        }
    }
}
// the same out of LDS (the write-back reads the segment it built)
template <bool NT>
__global__ __launch_bounds__(512) void fill_blocks_lds(uint4 *out, u64 nblocks) {
    __shared__ uint4 seg[2048];
    for (int i = threadIdx.x; i < 2048; i += 512) seg[i] = make_uint4(~0u, ~0u, threadIdx.x, 0u);
    __syncthreads();
    for (u64 b = blockIdx.x; b < nblocks; b += gridDim.x) {
        uint4 *p = out + b * 2048;
#pragma unroll
        for (int i = threadIdx.x; i < 2048; i += 512) {
            if (NT) nt_store(seg[i], &p[i]); else p[i] = seg[i];
        }
    }
}
__global__ __launch_bounds__(256) void copy_stride(const uint4 *__restrict__ in, uint4 *__restrict__ out, u64 n, int nt) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        const uint4 v = nt ? nt_load(&in[i]) : in[i];
        if (nt) nt_store(v, &out[i]); else out[i] = v;
    }
}

int main() {
    const u64 bytes = 3ull << 30, n = bytes / 16, nblocks = bytes / 32768;
    uint4 *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](const char *name, double bytes_moved, auto launch) {
        launch();
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; r++) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %8.1f GB/s  (%.3f ms)\n", name, bytes_moved * 5 / (ms * 1e-3) / 1e9, ms / 5);
    };
    char nm[128];
    for (int per : {2, 4, 8, 16, 32}) {
        snprintf(nm, sizeof nm, "fill stride plain   grid %2d/CU", per);
        timed(nm, (double)bytes, [&] { hipLaunchKernelGGL(fill_stride<false>, dim3(cus * per), dim3(256), 0, 0, b, n); });
        snprintf(nm, sizeof nm, "fill stride nontemp grid %2d/CU", per);
        timed(nm, (double)bytes, [&] { hipLaunchKernelGGL(fill_stride<true>, dim3(cus * per), dim3(256), 0, 0, b, n); });
    }
    for (int per : {2, 4, 8, 24}) {
        snprintf(nm, sizeof nm, "fill 32K blocks plain   512 thr grid %2d/CU", per);
        timed(nm, (double)bytes, [&] { hipLaunchKernelGGL((fill_blocks<false, 512>), dim3(cus * per), dim3(512), 0, 0, b, nblocks); });
        snprintf(nm, sizeof nm, "fill 32K blocks nontemp 512 thr grid %2d/CU", per);
        timed(nm, (double)bytes, [&] { hipLaunchKernelGGL((fill_blocks<true, 512>), dim3(cus * per), dim3(512), 0, 0, b, nblocks); });
    }
    for (int per : {4, 24}) {
        snprintf(nm, sizeof nm, "fill 32K blocks from LDS plain   grid %2d/CU", per);
        timed(nm, (double)bytes, [&] { hipLaunchKernelGGL(fill_blocks_lds<false>, dim3(cus * per), dim3(512), 0, 0, b, nblocks); });
        snprintf(nm, sizeof nm, "fill 32K blocks from LDS nontemp grid %2d/CU", per);
        timed(nm, (double)bytes, [&] { hipLaunchKernelGGL(fill_blocks_lds<true>, dim3(cus * per), dim3(512), 0, 0, b, nblocks); });
    }
    timed("hipMemsetAsync", (double)bytes, [&] { CK(hipMemsetAsync(b, 0xff, bytes, 0)); });
    for (int nt : {0, 1}) {
        snprintf(nm, sizeof nm, "copy stride %s grid 8/CU (r+w bytes)", nt ? "nontemp" : "plain  ");
        timed(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_stride, dim3(cus * 8), dim3(256), 0, 0, a, b, n, nt); });
    }
    timed("hipMemcpyAsync D2D (r+w bytes)", 2.0 * bytes, [&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); });
    return 0;
}
