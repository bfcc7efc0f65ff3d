// One-off search (gcc -O3 -pthread) for k-mers x != rc(x) whose reference hashCodes are equal — the
// tie case of FreqFilter.scala:31-32.  Its output is tests/golden/hash_ties.json.  For odd k <= 31 no
// tie exists: the middle base maps to itself under reverse-complement and is alone in its XOR-fold
// class of (int)(v ^ v>>>32), which forces bit = ~bit (4.8e10 random trials at k = 21, 27, 31 found none).
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <pthread.h>
static inline uint64_t rev_groups(uint64_t v){ v=__builtin_bswap64(v); v=((v>>4)&0x0f0f0f0f0f0f0f0fULL)|((v&0x0f0f0f0f0f0f0f0fULL)<<4); v=((v>>2)&0x3333333333333333ULL)|((v&0x3333333333333333ULL)<<2); return v; }
static inline int32_t h1(uint64_t v){ return (int32_t)(uint32_t)(v ^ (v>>32)); }
static inline int32_t h2(int64_t l1,int64_t l2){ int64_t t=(int64_t)((uint64_t)(l1^(l1>>32))*42ULL); int64_t t1=(int64_t)((uint64_t)(l2^(l2>>32)^t)*42ULL); return (int32_t)(uint32_t)(uint64_t)(t1^(t1>>32)); }
static uint64_t sm(uint64_t *s){ uint64_t z=(*s+=0x9E3779B97F4A7C15ULL); z=(z^(z>>30))*0xBF58476D1CE4E5B9ULL; z=(z^(z>>27))*0x94D049BB133111EBULL; return z^(z>>31);} 
typedef struct { int k; uint64_t seed; } arg_t;
static void* run(void*a_){ arg_t*a=a_; uint64_t s=a->seed; int k=a->k; int found=0;
  for(uint64_t it=0; it<9000000000ULL && found<2; it++){
    if(k<=32){ uint64_t mask=((1ULL<<(2*k))-1); uint64_t x=sm(&s)&mask; uint64_t r=rev_groups(~x)>>(64-2*k); if(x!=r && h1(x)==h1(r)){ printf("k=%d lo=0x%llx hi=0x0 rclo=0x%llx rchi=0x0 h=%d\n",k,(unsigned long long)x,(unsigned long long)r,h1(x)); fflush(stdout); found++; } }
    else { uint64_t lo=sm(&s), hi=sm(&s)&((k==64)?~0ULL:((1ULL<<(2*(k-32)))-1)); uint64_t nlo=rev_groups(~hi), nhi=rev_groups(~lo); int sft=128-2*k; uint64_t rlo,rhi; if(sft==0){rlo=nlo;rhi=nhi;} else {rlo=(nlo>>sft)|(nhi<<(64-sft)); rhi=nhi>>sft;}
      if(h2((int64_t)lo,(int64_t)hi)==h2((int64_t)rlo,(int64_t)rhi) && !(lo==rlo&&hi==rhi)){ printf("k=%d lo=0x%llx hi=0x%llx rclo=0x%llx rchi=0x%llx h=%d\n",k,(unsigned long long)lo,(unsigned long long)hi,(unsigned long long)rlo,(unsigned long long)rhi,h2(lo,hi)); fflush(stdout); found++; } }
  } return 0; }
int main(){ pthread_t t[8]; arg_t a[8]; int ks[8]={20,30,35,55,63,64,47,22};
  for(int i=0;i<8;i++){a[i].k=ks[i];a[i].seed=987654321ULL*(i+3);pthread_create(&t[i],0,run,&a[i]);}
  for(int i=0;i<8;i++)pthread_join(t[i],0); return 0; }
