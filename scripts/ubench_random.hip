// Micro-benchmark: random-access ceilings of MI355X HBM / Infinity Cache for the access shapes the
// k-mer table uses (8-B load, 64-bit CAS, 32-bit no-return add, 16-B load) at several footprints.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_random scripts/ubench_random.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint64_t u64; typedef uint32_t u32;
__device__ __forceinline__ u64 mix64(u64 x){x^=x>>33;x*=0xff51afd7ed558ccdULL;x^=x>>33;x*=0xc4ceb9fe1a85ec53ULL;x^=x>>33;return x;}
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

template<int MODE> __global__ __launch_bounds__(256) void k(u64* tab, u64 mask16 /* number of 16-B slots - 1 */, u64 n, u64* sink){
  u64 acc=0;
  for(u64 i=(u64)blockIdx.x*256+threadIdx.x;i<n;i+=(u64)gridDim.x*256){
    u64 s = mix64(i*0x9E3779B97F4A7C15ULL+12345) & mask16;
    u64* p = tab + 2*s;
    if(MODE==0){ acc += *p; }                                   // 8-B load
    else if(MODE==1){ u64 exp=~0ull; __hip_atomic_compare_exchange_strong(p,&exp,i,__ATOMIC_RELAXED,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT); acc+=exp; } // CAS
    else if(MODE==2){ (void)__hip_atomic_fetch_add((u32*)(p+1),1u,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT);} // add noret
    else if(MODE==3){ u64 v=*p; u64 exp=~0ull; if(v==~0ull){__hip_atomic_compare_exchange_strong(p,&exp,i,__ATOMIC_RELAXED,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT);} (void)__hip_atomic_fetch_add((u32*)(p+1),1u,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT); acc+=exp; } // load+CAS+add (current insert)
    else if(MODE==4){ u64 exp=~0ull; __hip_atomic_compare_exchange_strong(p,&exp,i,__ATOMIC_RELAXED,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT); if(exp!=~0ull)(void)__hip_atomic_fetch_add((u32*)(p+1),1u,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT); acc+=exp; } // CAS-first, add only on dup
    else if(MODE==5){ uint4 v=*(uint4*)p; acc+=v.x+v.w; }       // 16-B load
    else if(MODE==6){ *(uint4*)p = make_uint4((u32)i,1,2,3); }   // 16-B random store
    else if(MODE==7){ u64 v=__hip_atomic_fetch_add(p,1ull<<40,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT); acc+=v; } // 64-bit returning add
  }
  if(acc==0x1234567) *sink=acc;
}
int main(){
  const u64 n = 120000000ull;
  u64* sink; CHECK(hipMalloc(&sink,8));
  const char* names[]={"load8","cas64","add32_noret","load+cas+add","cas_first","load16","store16","add64_ret"};
  size_t sizes[]={64ull<<20, 1ull<<30, 4ull<<30, 32ull<<30};
  for(size_t sz: sizes){
    u64* tab; CHECK(hipMalloc(&tab,sz));
    u64 slots=sz/16;
    hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
    for(int mode=0;mode<8;mode++){
      float best=1e9;
      for(int rep=0;rep<3;rep++){
        CHECK(hipMemset(tab,0xff,sz)); CHECK(hipDeviceSynchronize());
        hipEventRecord(a);
        switch(mode){
          case 0: k<0><<<2048,256>>>(tab,slots-1,n,sink);break; case 1: k<1><<<2048,256>>>(tab,slots-1,n,sink);break;
          case 2: k<2><<<2048,256>>>(tab,slots-1,n,sink);break; case 3: k<3><<<2048,256>>>(tab,slots-1,n,sink);break;
          case 4: k<4><<<2048,256>>>(tab,slots-1,n,sink);break; case 5: k<5><<<2048,256>>>(tab,slots-1,n,sink);break;
          case 6: k<6><<<2048,256>>>(tab,slots-1,n,sink);break; case 7: k<7><<<2048,256>>>(tab,slots-1,n,sink);break;}
        hipEventRecord(b); CHECK(hipEventSynchronize(b));
        float ms; hipEventElapsedTime(&ms,a,b); if(ms<best)best=ms;
      }
      printf("footprint %6zu MiB  %-14s %8.3f ms  %7.2f Gop/s\n", sz>>20, names[mode], best, n/best*1e-6);
      fflush(stdout);
    }
    CHECK(hipFree(tab));
  }
  return 0;
}
