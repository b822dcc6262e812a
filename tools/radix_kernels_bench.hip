// tools/radix_kernels_bench.hip — times the product's radix-path kernels (binned.hpp) one by one at an order-2-like load:
// ids = Zipf class per position (delimiters invalid), key = (ids[i], ids[i+1]). Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include "kernels.hpp"
#include "binned.hpp"
using namespace colibri;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
__global__ void gen(uint32_t* ids, uint32_t n, float lnV){
  for(uint32_t i=blockIdx.x*blockDim.x+threadIdx.x;i<n;i+=gridDim.x*blockDim.x){
    float u=(mix64(i+1)>>40)*(1.0f/16777216.0f); uint32_t r=(uint32_t)__expf(u*lnV); ids[i]= (i%21==20)?kInvalid:(r+5u); }
}
int main(){
  const uint32_t n=105000000u; uint32_t *ids,*rep_of,*ids_at,*ids2; CK(hipMalloc(&ids,(size_t)n*4+64)); CK(hipMalloc(&rep_of,(size_t)n*4+64)); CK(hipMalloc(&ids_at,(size_t)n*4+64)); CK(hipMalloc(&ids2,(size_t)n*4+64));
  const size_t nrec=(size_t)n+n/8+kBins*4096; Rec *r0,*r1; CK(hipMalloc(&r0,nrec*sizeof(Rec))); CK(hipMalloc(&r1,nrec*sizeof(Rec)));
  DevState* st; CK(hipMalloc(&st,sizeof(DevState))); BinState* bs; CK(hipMalloc(&bs,sizeof(BinState)));
  uint32_t* nlist; CK(hipMalloc(&nlist,64));
  hipLaunchKernelGGL(gen,dim3(4096),dim3(256),0,0,ids,n,logf(1e6f)); CK(hipDeviceSynchronize());
  const uint32_t region=(uint32_t)(nrec/kASlots), tiles=(n+kScatTile-1)/kScatTile+1+kASlots;
  uint32_t* sp=(uint32_t*)r0;
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  KeyNgram fn{ids,2};
  auto emit=[&](){ hipLaunchKernelGGL((bin_emit_kernel<KeyNgram,false>),dim3(256*4),dim3(kBlock),0,0,fn,r0,region,rep_of,st,bs,n,(const uint32_t*)nullptr,(const uint32_t*)nullptr,ids_at); };
  auto offs=[&](){ hipLaunchKernelGGL(bin_offsets_kernel,dim3(1),dim3(kBlock),0,0,bs,region); };
  auto hist=[&](){ hipLaunchKernelGGL(bin_hist2_kernel,dim3(tiles+kBins),dim3(kBlock),0,0,r0,st,bs); };
  auto scan=[&](){ hipLaunchKernelGGL(bin_scan2_kernel,dim3(kBins),dim3(kBlock),0,0,bs); };
  auto scat=[&](){ hipLaunchKernelGGL(bin_scatter_kernel,dim3(tiles+kBins),dim3(kBlock),0,0,r0,r1,st,bs); };
  auto cnt=[&](){ hipLaunchKernelGGL(bin_count_kernel,dim3(256*12),dim3(kBlock),0,0,r1,st,bs,2u,sp,sp+n,(unsigned long long*)nullptr,ids_at); };
  auto reso=[&](){ hipLaunchKernelGGL((bin_resolve_kernel<false>),dim3(4096),dim3(kBlock),0,0,rep_of,ids_at,ids2,st,n,(const uint32_t*)nullptr,(const uint32_t*)nullptr,(uint32_t*)r0,nlist,(const uint32_t*)nullptr,0u); };
  auto timeit=[&](const char* name, auto pre, auto fn){ float best=1e9; for(int r=0;r<3;r++){ CK(hipMemset(st,0,sizeof(DevState))); CK(hipMemset(bs,0,sizeof(BinState))); CK(hipMemset(nlist,0,64)); pre(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(a)); fn(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms,a,b)); if(ms<best)best=ms; }
    printf("%-28s %8.3f ms\n",name,best); fflush(stdout); };
  timeit("bin_emit<KeyNgram,false>",[&](){},emit);
  timeit("bin_hist2",[&](){emit();offs();},hist);
  timeit("bin_scatter",[&](){emit();offs();hist();scan();},scat);
  timeit("bin_count",[&](){emit();offs();hist();scan();scat();},cnt);
  timeit("bin_resolve<false>",[&](){emit();offs();hist();scan();scat();cnt();},reso);
  BinState h; CK(hipMemcpy(&h,bs,sizeof(BinState),hipMemcpyDeviceToHost)); DevState hs; CK(hipMemcpy(&hs,st,sizeof(DevState),hipMemcpyDeviceToHost));
  printf("nrec %u bshift %u admitted %u radix_overflow %u\n",h.nrec,h.bshift,hs.admitted,hs.radix_overflow);
  return 0;
}
