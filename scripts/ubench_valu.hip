// ubench_valu.hip — issue cost of the integer VALU operations the k-mer kernels are made of (MI355X).
// One wave per SIMD-slot worth of work, N dependent-free operations of one kind per lane in a loop, 4 independent
// chains per lane so that latency is covered; reports cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64; typedef unsigned int u32;
#define REP 4096
template <int OP> __global__ void k(u64 *out, u32 seed) {
    u32 a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    u64 b0 = a0 | ((u64)a1 << 32), b1 = b0 * 3, b2 = b0 * 5, b3 = b0 * 7;
    const u32 sh = (seed & 15) + 1;
    u64 t0 = __builtin_readcyclecounter();
    for (int i = 0; i < REP; i++) {
        if (OP == 0) { a0 += a1; a1 += a2; a2 += a3; a3 += a0; }                                   // v_add_u32
        if (OP == 1) { a0 *= a1 | 1u; a1 *= a2 | 1u; a2 *= a3 | 1u; a3 *= a0 | 1u; }                 // v_or + v_mul_lo_u32
        if (OP == 2) { b0 = (b0 >> sh) ^ b1; b1 = (b1 >> sh) ^ b2; b2 = (b2 >> sh) ^ b3; b3 = (b3 >> sh) ^ b0; }   // v_lshrrev_b64 + 2 xor
        if (OP == 3) { b0 *= b1 | 1ull; b1 *= b2 | 1ull; b2 *= b3 | 1ull; b3 *= b0 | 1ull; }     // 64-bit mul
        if (OP == 4) { a0 = __builtin_amdgcn_alignbit(a1, a0, sh); a1 = __builtin_amdgcn_alignbit(a2, a1, sh); a2 = __builtin_amdgcn_alignbit(a3, a2, sh); a3 = __builtin_amdgcn_alignbit(a0, a3, sh); }
        if (OP == 5) { b0 = (u64)(u32)b0 * (u32)(b0 >> 32) + b1; b1 = (u64)(u32)b1 * (u32)(b1 >> 32) + b2; b2 = (u64)(u32)b2 * (u32)(b2 >> 32) + b3; b3 = (u64)(u32)b3 * (u32)(b3 >> 32) + b0; }  // v_mad_u64_u32
        if (OP == 6) { a0 = __brev(a0) ^ a1; a1 = __brev(a1) ^ a2; a2 = __brev(a2) ^ a3; a3 = __brev(a3) ^ a0; }   // v_bfrev + xor
        if (OP == 7) { a0 = __umulhi(a0, 0x846ca68bU) ^ a1; a1 = __umulhi(a1, 0x846ca68bU) ^ a2; a2 = __umulhi(a2, 0x846ca68bU) ^ a3; a3 = __umulhi(a3, 0x846ca68bU) ^ a0; }  // v_mul_hi_u32 + xor
    }
    u64 t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if ((a0 ^ a1 ^ a2 ^ a3 ^ (u32)b0 ^ (u32)b1 ^ (u32)b2 ^ (u32)b3) == 0x12345) out[1] = 1;
}
template <int OP> void run(const char *name, int ops_per_iter, u64 *d) {
    // 4 waves per workgroup x 1 workgroup per CU-ish: each SIMD runs one wave -> cycles per instruction = issue cost
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 0, 0, d, 1u);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 0, 0, d, 1u);
    hipDeviceSynchronize();
    u64 h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-34s %6.2f cycles per wave-instruction-group (%d ops per group)\n", name, (double)h[0] / (REP * 4.0), ops_per_iter);
}
int main() {
    u64 *d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    run<0>("v_add_u32", 1, d);
    run<1>("v_or_b32 + v_mul_lo_u32", 2, d);
    run<2>("v_lshrrev_b64 + v_xor x2", 3, d);
    run<3>("v_or + 64-bit multiply (mul_lo x2 + mad)", 4, d);
    run<4>("v_alignbit_b32", 1, d);
    run<5>("v_mad_u64_u32", 1, d);
    run<6>("v_bfrev_b32 + v_xor", 2, d);
    run<7>("v_mul_hi_u32 + v_xor", 2, d);
    return 0;
}
