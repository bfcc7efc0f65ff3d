// ubench_runs.hip — how much do partial-sector writes cost on MI355X?
// Writes `nkeys` 8-byte keys as runs of L consecutive keys; run i goes to region (hash(i) % R) at
// offset (i / R) * stride keys.  Consecutive lanes write consecutive keys of a run (the write-out
// pattern of gk_partition.hip's P4).  stride == L: packed, runs start at arbitrary 8-B offsets;
// stride rounded up to 8 keys: every run starts on a 64-B sector.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef unsigned long long u64; typedef unsigned int u32;
__device__ __forceinline__ u64 mix(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; return x ^ (x >> 33); }
__global__ void k_runs(u64 *out, u64 nkeys, u32 L, u32 stride, u32 R, u64 cap) {
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < nkeys; t += (u64)gridDim.x * blockDim.x) {
        const u64 run = t / L; const u32 pos = (u32)(t - run * L);
        const u64 region = mix(run) % R;
        const u64 off = (run / R) * stride;       // approximately: runs per region are sequential in time
        out[region * cap + off + pos] = t;
    }
}
int main(int argc, char **argv) {
    const u64 nkeys = 120000000ull; const u32 R = 94720;
    struct Cfg { u32 L, stride; const char *name; } cfgs[] = {
        {11, 11, "L=11 packed (88 B runs, unaligned)"}, {11, 16, "L=11 stride 16 (aligned starts, holes)"},
        {8, 8, "L=8 aligned full sectors"}, {16, 16, "L=16 aligned 128 B"}, {22, 22, "L=22 packed"}, {22, 24, "L=22 stride 24"},
        {32, 32, "L=32 aligned 256 B"}, {64, 64, "L=64 aligned 512 B"}, {5, 5, "L=5 packed"}, {1000000, 1000000, "linear"}};
    u64 *out; const u64 cap = 4096; hipMalloc(&out, (u64)R * cap * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto &c : cfgs) {
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_runs, dim3(256 * 8), dim3(512), 0, 0, out, nkeys, c.L, c.stride, R, cap);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-44s %.3f ms  %.2f TB/s payload\n", c.name, best, nkeys * 8.0 / best / 1e9);
    }
    return 0;
}
