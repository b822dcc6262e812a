// tools/atomics_bench.hip — microbenchmark that prices the primitives the count kernel is built from on one MI355X:
// device-scope atomics (no-return add, CAS) by address distribution and table size, with/without a preceding probe load.
// Not part of the product; numbers go into DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

__device__ __forceinline__ uint64_t mix(uint64_t x){ x ^= x>>33; x*=0xff51afd7ed558ccdULL; x^=x>>33; x*=0xc4ceb9fe1a85ec53ULL; x^=x>>33; return x; }

struct Slot { uint64_t key; uint32_t count; uint32_t rep; };

// mode 0: atomicAdd u64 no-return on slot[idx].count ; 1: probe load of key then add ; 2: atomicAdd u32 ; 3: plain load only (gather) ; 4: CAS returning
__global__ void k_rand(Slot* t, uint32_t nslots, const uint32_t* idxs, uint64_t n, int mode, uint64_t* sink){
  uint64_t acc=0;
  for(uint64_t i=blockIdx.x*(uint64_t)blockDim.x+threadIdx.x;i<n;i+=(uint64_t)gridDim.x*blockDim.x){
    uint32_t idx = idxs? idxs[i] : (uint32_t)(((mix(i+1)>>32)*(uint64_t)nslots)>>32);
    if(mode==0) atomicAdd((unsigned long long*)&t[idx].count, 1ull);
    else if(mode==1){ uint64_t k=t[idx].key; acc+=k; atomicAdd((unsigned long long*)&t[idx].count, 1ull); }
    else if(mode==2) atomicAdd(&t[idx].count,1u);
    else if(mode==3){ acc+=t[idx].key; }
    else if(mode==4){ acc+=atomicCAS((unsigned long long*)&t[idx].key, ~0ull, (unsigned long long)i); }
    else if(mode==5){ acc+=atomicAdd(&t[idx].count,1u); }   // returning add
    else if(mode==6){ acc+= __hip_atomic_load(&t[idx].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); atomicAdd((unsigned long long*)&t[idx].count, 1ull);} // sc1 load + add
  }
  if(acc==0x1234567) *sink=acc;
}

int main(int argc,char**argv){
  uint64_t n = 100000000ull;
  Slot* t; uint64_t maxslots = 160000000ull; CK(hipMalloc(&t, maxslots*sizeof(Slot)));
  uint64_t* sink; CK(hipMalloc(&sink,8));
  uint32_t* idxs; CK(hipMalloc(&idxs, n*4));
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto run=[&](const char* name, uint32_t nslots, const uint32_t* ix, int mode, int grid){
    CK(hipMemset(t,0xFF,(size_t)nslots*sizeof(Slot)));
    CK(hipDeviceSynchronize());
    float best=1e9;
    for(int r=0;r<3;r++){ CK(hipEventRecord(a)); hipLaunchKernelGGL(k_rand,dim3(grid),dim3(256),0,0,t,nslots,ix,n,mode,sink); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms,a,b)); if(ms<best)best=ms; }
    printf("%-44s slots=%10u (%7.1f MB) mode=%d grid=%5d : %8.3f ms  %7.2f Gop/s\n",name,nslots,nslots*16.0/1e6,mode,grid,best,n/best/1e6); fflush(stdout);
  };
  // uniform random addresses by table size
  for(uint32_t ns : {2000000u, 16000000u, 150000000u}){
    for(int mode: {0,1,2,3,4,5,6}) run("uniform", ns, nullptr, mode, 4096);
  }
  for(int grid: {1024, 2048, 8192, 16384}) run("uniform grid sweep", 150000000u, nullptr, 0, grid);
  // zipf addresses over V=1e6 keys scattered in a 2M-slot table (order-1 like), without any aggregation
  {
    std::vector<uint32_t> h(n); const uint32_t V=1000000; std::vector<double> cdf(V); double s=0; for(uint32_t r=0;r<V;r++){ s+=1.0/(r+1); cdf[r]=s; } for(auto&x:cdf)x/=s;
    uint64_t st=88172645463325252ull; auto rnd=[&](){ st^=st<<13; st^=st>>7; st^=st<<17; return (st>>11)*(1.0/9007199254740992.0); };
    for(uint64_t i=0;i<n;i++){ double u=rnd(); uint32_t lo=0,hi=V-1; while(lo<hi){uint32_t m=(lo+hi)/2; if(cdf[m]<u)lo=m+1; else hi=m;} uint64_t hh=(uint64_t)lo*0x9E3779B97F4A7C15ull; h[i]=(uint32_t)(((hh>>32)*2000000ull)>>32); }
    CK(hipMemcpy(idxs,h.data(),n*4,hipMemcpyHostToDevice));
    for(int mode: {0,1,2,3,6}) run("zipf V=1e6 raw (no aggregation)", 2000000u, idxs, mode, 4096);
    // drop the top-K keys (as if perfectly pre-aggregated): replace their occurrences by uniform random
    for(uint32_t K : {100u, 10000u}){
      std::vector<uint32_t> g(h); std::vector<uint32_t> top(K); for(uint32_t r=0;r<K;r++){ uint64_t hh=(uint64_t)r*0x9E3779B97F4A7C15ull; top[r]=(uint32_t)(((hh>>32)*2000000ull)>>32);} 
      std::vector<char> ishot(2000000,0); for(auto x:top) ishot[x]=1; uint64_t repl=0; for(uint64_t i=0;i<n;i++) if(ishot[g[i]]){ g[i]=(uint32_t)(rnd()*2000000); repl++; }
      CK(hipMemcpy(idxs,g.data(),n*4,hipMemcpyHostToDevice)); char nm[64]; snprintf(nm,64,"zipf minus top-%u (%.0f%% replaced)",K,100.0*repl/n);
      for(int mode: {0,1}) run(nm, 2000000u, idxs, mode, 4096);
    }
  }
  // same address
  { std::vector<uint32_t> h(n, 12345u); CK(hipMemcpy(idxs,h.data(),n*4,hipMemcpyHostToDevice)); n=4000000; run("same address (4M ops)", 2000000u, idxs, 0, 4096); run("same address returning (4M ops)", 2000000u, idxs, 5, 4096); }
  return 0;
}
