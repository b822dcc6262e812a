// tools/atomics_scope_bench.hip — does a narrower atomic scope or an XCD-private table move global atomics from the memory side
// into the XCD's L2 on MI355X?  Random no-return u32 adds: one shared table (agent / workgroup scope) against eight per-XCD
// copies selected by the hardware XCC id. The sum over all copies is checked, so a scope that loses updates shows up.
// Not part of the product; numbers go into DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
__device__ __forceinline__ uint64_t mix(uint64_t x){ x ^= x>>33; x*=0xff51afd7ed558ccdULL; x^=x>>33; x*=0xc4ceb9fe1a85ec53ULL; x^=x>>33; return x; }
__device__ __forceinline__ uint32_t xcc_id(){ return __builtin_amdgcn_s_getreg((3u<<11)|(0u<<6)|20u) & 15u; }  // HW_REG_XCC_ID[3:0]

template<int SCOPE, bool PRIVATE, bool RET>
__global__ void k(uint32_t* t, uint32_t n_entries, uint64_t n, uint32_t* xcc_seen, uint64_t* sink){
  const uint32_t x = xcc_id();
  if(threadIdx.x==0) atomicOr(&xcc_seen[blockIdx.x & 1023], 1u<<x);
  uint32_t* tab = PRIVATE ? t + (size_t)x*n_entries : t;
  uint64_t acc=0;
  for(uint64_t i=blockIdx.x*(uint64_t)blockDim.x+threadIdx.x;i<n;i+=(uint64_t)gridDim.x*blockDim.x){
    const uint32_t idx=(uint32_t)(((mix(i+1)>>32)*(uint64_t)n_entries)>>32);
    if(RET) acc+=__hip_atomic_fetch_add(&tab[idx],1u,__ATOMIC_RELAXED,SCOPE);
    else (void)__hip_atomic_fetch_add(&tab[idx],1u,__ATOMIC_RELAXED,SCOPE);
  }
  if(acc==0x1234567) *sink=acc;
}
__global__ void sum_k(const uint32_t* t, size_t n, unsigned long long* out){
  unsigned long long s=0; for(size_t i=blockIdx.x*(size_t)blockDim.x+threadIdx.x;i<n;i+=(size_t)gridDim.x*blockDim.x) s+=t[i];
  atomicAdd(out,s);
}
int main(){
  const uint64_t n=200000000ull; const size_t maxe=8ull*64*1024*1024;  // up to 8 x 256 MB
  uint32_t* t; CK(hipMalloc(&t,maxe*4)); uint32_t* seen; CK(hipMalloc(&seen,4096)); uint64_t* sink; CK(hipMalloc(&sink,8)); unsigned long long* out; CK(hipMalloc(&out,8));
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto run=[&](const char* name, auto kern, uint32_t ne, bool priv, int grid){
    const size_t total=(size_t)ne*(priv?8:1);
    float best=1e9; unsigned long long s=0;
    for(int r=0;r<3;r++){
      CK(hipMemset(t,0,total*4)); CK(hipMemset(seen,0,4096)); CK(hipMemset(out,0,8)); CK(hipDeviceSynchronize());
      CK(hipEventRecord(a)); hipLaunchKernelGGL(kern,dim3(grid),dim3(256),0,0,t,ne,n,seen,sink); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms,a,b)); if(ms<best)best=ms;
      hipLaunchKernelGGL(sum_k,dim3(1024),dim3(256),0,0,t,total,out); CK(hipMemcpy(&s,out,8,hipMemcpyDeviceToHost));
    }
    std::vector<uint32_t> h(1024); CK(hipMemcpy(h.data(),seen,4096,hipMemcpyDeviceToHost)); int multi=0; uint32_t all=0; for(int i=0;i<1024;i++){ all|=h[i]; if(h[i]&(h[i]-1)) multi++; }
    printf("%-34s entries=%9u (%6.1f MB%s) grid=%5d : %8.3f ms %7.2f Gop/s  sum %s  xcc mask %#x, block-slots seen on >1 xcc: %d\n",
           name,ne,ne*4.0/1e6,priv?" x8":"",grid,best,n/best/1e6,s==n?"ok":"LOST UPDATES",all,multi); fflush(stdout);
  };
  for(uint32_t ne: {1u<<18, 1u<<20, 1u<<23, 1u<<26}){
    run("shared, agent scope",        k<__HIP_MEMORY_SCOPE_AGENT,false,false>,     ne,false,4096);
    run("shared, workgroup scope",    k<__HIP_MEMORY_SCOPE_WORKGROUP,false,false>, ne,false,4096);
    run("xcd-private, agent scope",   k<__HIP_MEMORY_SCOPE_AGENT,true,false>,      ne,true,4096);
    run("xcd-private, workgroup scope",k<__HIP_MEMORY_SCOPE_WORKGROUP,true,false>, ne,true,4096);
    run("xcd-private, wavefront scope",k<__HIP_MEMORY_SCOPE_WAVEFRONT,true,false>, ne,true,4096);
    run("xcd-private, agent, returning",k<__HIP_MEMORY_SCOPE_AGENT,true,true>,     ne,true,4096);
    run("xcd-private, wg, returning", k<__HIP_MEMORY_SCOPE_WORKGROUP,true,true>,   ne,true,4096);
  }
  return 0;
}
