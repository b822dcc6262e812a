// tools/lds_hist_bench.hip — what does an LDS histogram / an LDS-ranked partition cost per element on MI355X, by strategy?
// 100 M Zipf(1.0, V = 1e6)-distributed u32 "classes" (rank = V^u), 512 persistent blocks of 256 lanes, 8 loads in flight per lane.
// Not part of the product; numbers go into DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
constexpr int kBlock=256, kPer=8, kHead=8192;
__device__ __forceinline__ uint64_t mix(uint64_t x){ x ^= x>>33; x*=0xff51afd7ed558ccdULL; x^=x>>33; x*=0xc4ceb9fe1a85ec53ULL; x^=x>>33; return x; }
__global__ void gen(uint32_t* cls, uint32_t n, float lnV){
  for(uint32_t i=blockIdx.x*blockDim.x+threadIdx.x;i<n;i+=gridDim.x*blockDim.x){
    float u=(mix(i+1)>>40)*(1.0f/16777216.0f); uint32_t r=(uint32_t)__expf(u*lnV); cls[i]= (i%21==20)?0u:(r+5u); }
}
// MODE 0 read only | 1 head hist no-return | 2 head hist returning | 3 head + 256 tail bins no-return | 4 256-bin hist of (c>>12), all elements, returning (partition ranks)
// 5 head hist with wave-level folding of equal classes (ballot of the wave leader's class, repeated while lanes remain, max 4 rounds then plain atomics)
template<int MODE>
__global__ __launch_bounds__(kBlock) void k(const uint32_t* __restrict__ cls, uint32_t n, uint32_t* __restrict__ out){
  __shared__ uint32_t histL[kHead], binL[256];
  for(int t=threadIdx.x;t<kHead;t+=kBlock) histL[t]=0; binL[threadIdx.x]=0; __syncthreads();
  const uint32_t per=(n+gridDim.x-1)/gridDim.x, begin=blockIdx.x*per, end=min(n,begin+per);
  uint32_t acc=0;
  for(uint32_t i0=begin;i0<end;i0+=kBlock*kPer){
    uint32_t c[kPer];
#pragma unroll
    for(int q=0;q<kPer;++q){ const uint32_t i=i0+q*kBlock+threadIdx.x; c[q]=(i<end)?cls[i]:0u; }
#pragma unroll
    for(int q=0;q<kPer;++q){
      const uint32_t v=c[q];
      if(MODE==0){ acc+=v; }
      else if(MODE==1){ if(v && v<kHead) atomicAdd(&histL[v],1u); }
      else if(MODE==2){ if(v && v<kHead) acc+=atomicAdd(&histL[v],1u); }
      else if(MODE==3){ if(v){ if(v<kHead) atomicAdd(&histL[v],1u); else atomicAdd(&binL[v>>12],1u);} }
      else if(MODE==4){ if(v) acc+=atomicAdd(&binL[(v>>12)&255u],1u); }
      else if(MODE==5){
        bool live = v && v<kHead;
        for(int round=0; round<4; ++round){
          const uint64_t m=__ballot(live); if(!m) break;
          const int leader=__builtin_ctzll(m); const uint32_t lv=__shfl(v,leader,64);
          const uint64_t same=__ballot(live && v==lv);
          if(live && v==lv){ if((int)(threadIdx.x&63)==leader) atomicAdd(&histL[v],(uint32_t)__popcll(same)); live=false; }
        }
        if(live) atomicAdd(&histL[v],1u);
      }
    }
  }
  __syncthreads();
  if(MODE!=0){ for(int t=threadIdx.x;t<kHead;t+=kBlock) acc+=histL[t]; acc+=binL[threadIdx.x]; }
  if(acc==0x12345678u) out[0]=acc;
}
int main(){
  const uint32_t n=105000000u; uint32_t* cls; CK(hipMalloc(&cls,(size_t)n*4)); uint32_t* out; CK(hipMalloc(&out,64));
  hipLaunchKernelGGL(gen,dim3(4096),dim3(256),0,0,cls,n,logf(1e6f)); CK(hipDeviceSynchronize());
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto run=[&](const char* name, auto kern, int grid){ float best=1e9; for(int r=0;r<4;r++){ CK(hipEventRecord(a)); hipLaunchKernelGGL(kern,dim3(grid),dim3(kBlock),0,0,cls,n,out); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms,a,b)); if(ms<best)best=ms; }
    printf("%-64s grid=%5d : %7.3f ms  %7.1f G elem/s  %6.2f TB/s\n",name,grid,best,n/best/1e6,n*4.0/best/1e9); fflush(stdout); };
  for(int grid: {512,1024,2048}){
    run("0 read only",k<0>,grid);
    run("1 head histogram (8192 classes), no-return LDS atomics",k<1>,grid);
    run("2 head histogram, returning LDS atomics",k<2>,grid);
    run("3 head histogram + 256 tail bins, no-return",k<3>,grid);
    run("4 256-bin ranks for every element, returning",k<4>,grid);
    run("5 head histogram, wave-folded (4 rounds)",k<5>,grid);
  }
  return 0;
}
