// Micro-benchmark: LDS random-access op rates per CU (ds_read_b64, ds_write_b64, ds_add_u32,
// ds_cmpst_rtn_b32, ds_cmpst_rtn_b64, ds_add_rtn_u32) — which LDS primitive can carry a hash build.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64; typedef uint32_t u32;
__device__ __forceinline__ u32 h32(u32 x){x^=x>>16;x*=0x7feb352dU;x^=x>>15;x*=0x846ca68bU;x^=x>>16;return x;}
template<int MODE> __global__ __launch_bounds__(256) void k(u64* sink, int iters){
  __shared__ u64 tab[2048];   // 16 KiB
  for(int i=threadIdx.x;i<2048;i+=256) tab[i]=~0ull;
  __syncthreads();
  u64 acc=0; u32 x=threadIdx.x*2654435761u+blockIdx.x;
  for(int it=0;it<iters;it++){
    x=h32(x+it);
    u32 s = x & 2047;
    if(MODE==0) acc+=tab[s];
    else if(MODE==1) tab[s]=x;
    else if(MODE==2) atomicAdd((u32*)&tab[s],1u);
    else if(MODE==3) acc+=atomicCAS((u32*)&tab[s],0xffffffffu,x);
    else if(MODE==4) acc+=atomicCAS((unsigned long long*)&tab[s],~0ull,(unsigned long long)x);
    else if(MODE==5) acc+=atomicAdd((u32*)&tab[s],1u);
    else if(MODE==6) { u64 v=tab[s]; if(v==~0ull) v=atomicCAS((unsigned long long*)&tab[s],~0ull,(unsigned long long)x); acc+=v; }
    else if(MODE==7) acc+=atomicExch((unsigned long long*)&tab[s],(unsigned long long)x);
    else if(MODE==8) acc+=atomicMax((unsigned long long*)&tab[s],(unsigned long long)x);
  }
  if(acc==0x12345) *sink=acc;
}
int main(){
  u64* sink; hipMalloc(&sink,8);
  const char* names[]={"ds_read_b64","ds_write_b64","ds_add_u32(noret)","cas32_rtn","cas64_rtn","add32_rtn","read+cas64","xchg64_rtn","max64_rtn"};
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters=2000, blocks=256*8;
  for(int mode=0;mode<9;mode++){
    float best=1e9;
    for(int rep=0;rep<3;rep++){
      hipEventRecord(a);
      switch(mode){case 0:k<0><<<blocks,256>>>(sink,iters);break;case 1:k<1><<<blocks,256>>>(sink,iters);break;case 2:k<2><<<blocks,256>>>(sink,iters);break;
        case 3:k<3><<<blocks,256>>>(sink,iters);break;case 4:k<4><<<blocks,256>>>(sink,iters);break;case 5:k<5><<<blocks,256>>>(sink,iters);break;
        case 6:k<6><<<blocks,256>>>(sink,iters);break;case 7:k<7><<<blocks,256>>>(sink,iters);break;case 8:k<8><<<blocks,256>>>(sink,iters);break;}
      hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); if(ms<best)best=ms;
    }
    double ops=(double)blocks*256*iters;
    printf("%-18s %8.3f ms  %8.2f Gop/s chip  %6.2f lane-ops/clk/CU (2.4GHz)\n",names[mode],best,ops/best*1e-6, ops/best*1e-6/256/2.4);
  }
  return 0;
}
