// tools/uni_kernels_bench.hip — times the product's order-1 kernels (kernels.hpp) one by one on a synthetic Zipf class array.
// Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include "kernels.hpp"
using namespace colibri;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
__global__ void gen(uint32_t* cls, uint32_t n, float lnV){
  for(uint32_t i=blockIdx.x*blockDim.x+threadIdx.x;i<n;i+=gridDim.x*blockDim.x){
    float u=(mix64(i+1)>>40)*(1.0f/16777216.0f); uint32_t r=(uint32_t)__expf(u*lnV); cls[i]= (i%21==20)?0u:(r+5u); }
}
int main(){
  const uint32_t n=105000000u, nclasses=1000008u, shift=12; uint32_t* cls; CK(hipMalloc(&cls,(size_t)n*4));
  uint32_t *cnt1,*rep1,*ids; CK(hipMalloc(&cnt1,nclasses*4+64)); CK(hipMalloc(&rep1,nclasses*4+64)); CK(hipMalloc(&ids,(size_t)n*4));
  DevState* st; CK(hipMalloc(&st,sizeof(DevState))); CK(hipMemset(st,0,sizeof(DevState)));
  UniState* us; CK(hipMalloc(&us,sizeof(UniState))); uint16_t* tail; CK(hipMalloc(&tail,(size_t)n*2)); uint32_t* rows; CK(hipMalloc(&rows,(size_t)512*kUniHead*4));
  hipLaunchKernelGGL(gen,dim3(4096),dim3(256),0,0,cls,n,logf(1e6f)); CK(hipDeviceSynchronize());
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto timeit=[&](const char* name, auto fn){ float best=1e9; for(int r=0;r<3;r++){ CK(hipMemset(cnt1,0,nclasses*4)); CK(hipMemset(us,0,sizeof(UniState))); CK(hipMemset(st,0,sizeof(DevState))); fn(false); CK(hipDeviceSynchronize());
      CK(hipEventRecord(a)); fn(true); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms,a,b)); if(ms<best)best=ms; }
    printf("%-40s %8.3f ms\n",name,best); fflush(stdout); };
  // each lambda: fn(false) runs the prerequisites, fn(true) the kernel under test
  timeit("uni_count_kernel (atomics)",[&](bool t){ if(t) hipLaunchKernelGGL(uni_count_kernel,dim3(512),dim3(kBlock),0,0,cls,n,cnt1,rep1,st); });
  timeit("uni_head_kernel",[&](bool t){ if(t) hipLaunchKernelGGL(uni_head_kernel,dim3(512),dim3(kBlock),0,0,cls,n,shift,rows,us,st); });
  timeit("uni_head_reduce_kernel",[&](bool t){ if(!t) hipLaunchKernelGGL(uni_head_kernel,dim3(512),dim3(kBlock),0,0,cls,n,shift,rows,us,st); else hipLaunchKernelGGL(uni_head_reduce_kernel,dim3(kUniHead/kBlock,16),dim3(kBlock),0,0,rows,512u,cnt1,nclasses,st); });
  timeit("uni_partition_kernel",[&](bool t){ if(!t){ hipLaunchKernelGGL(uni_head_kernel,dim3(512),dim3(kBlock),0,0,cls,n,shift,rows,us,st); hipLaunchKernelGGL(uni_offsets_kernel,dim3(1),dim3(kBlock),0,0,us);} else hipLaunchKernelGGL(uni_partition_kernel,dim3(2048),dim3(kBlock),0,0,cls,n,shift,us,tail,st); });
  timeit("uni_tail_count_kernel",[&](bool t){ if(!t){ hipLaunchKernelGGL(uni_head_kernel,dim3(512),dim3(kBlock),0,0,cls,n,shift,rows,us,st); hipLaunchKernelGGL(uni_offsets_kernel,dim3(1),dim3(kBlock),0,0,us); hipLaunchKernelGGL(uni_partition_kernel,dim3(2048),dim3(kBlock),0,0,cls,n,shift,us,tail,st);} else hipLaunchKernelGGL(uni_tail_count_kernel,dim3(kUniBins*kUniSlices),dim3(kBlock),sizeof(uint32_t)<<shift,0,tail,us,shift,cnt1,nclasses,st); });
  timeit("uni_ids_kernel",[&](bool t){ if(t) hipLaunchKernelGGL(uni_ids_kernel,dim3(4096),dim3(kBlock),0,0,cls,cnt1,2u,ids,st,n); });
  uint32_t* surv; CK(hipMalloc(&surv,(nclasses/32+64)*4)); uint32_t *rr,*rc; CK(hipMalloc(&rr,nclasses*4+64)); CK(hipMalloc(&rc,nclasses*4+64));
  auto full=[&](){ hipLaunchKernelGGL(uni_head_kernel,dim3(512),dim3(kBlock),0,0,cls,n,shift,rows,us,st); hipLaunchKernelGGL(uni_head_reduce_kernel,dim3(kUniHead/kBlock,16),dim3(kBlock),0,0,rows,512u,cnt1,nclasses,st); hipLaunchKernelGGL(uni_offsets_kernel,dim3(1),dim3(kBlock),0,0,us); hipLaunchKernelGGL(uni_partition_kernel,dim3(2048),dim3(kBlock),0,0,cls,n,shift,us,tail,st); hipLaunchKernelGGL(uni_tail_count_kernel,dim3(kUniBins*kUniSlices),dim3(kBlock),sizeof(uint32_t)<<shift,0,tail,us,shift,cnt1,nclasses,st); };
  timeit("uni_finish_kernel (+bitmap)",[&](bool t){ if(!t) full(); else hipLaunchKernelGGL(uni_finish_kernel,dim3(245),dim3(kBlock),0,0,cnt1,(const uint32_t*)nullptr,nclasses,2u,st,rr,rc,nclasses,(uint16_t*)surv); });
  timeit("uni_ids_bitmap_kernel",[&](bool t){ if(!t){ full(); hipLaunchKernelGGL(uni_finish_kernel,dim3(245),dim3(kBlock),0,0,cnt1,(const uint32_t*)nullptr,nclasses,2u,st,rr,rc,nclasses,(uint16_t*)surv);} else hipLaunchKernelGGL(uni_ids_bitmap_kernel,dim3(4096),dim3(kBlock),0,0,cls,surv,(nclasses+31)/32,ids,st,n); });
  timeit("whole order 1",[&](bool t){ if(t){ full(); hipLaunchKernelGGL(uni_finish_kernel,dim3(245),dim3(kBlock),0,0,cnt1,(const uint32_t*)nullptr,nclasses,2u,st,rr,rc,nclasses,(uint16_t*)surv); hipLaunchKernelGGL(uni_ids_bitmap_kernel,dim3(4096),dim3(kBlock),0,0,cls,surv,(nclasses+31)/32,ids,st,n);} });
  uint32_t h[4]; CK(hipMemcpy(h,cnt1+6,16,hipMemcpyDeviceToHost)); printf("cnt1[6..9] = %u %u %u %u\n",h[0],h[1],h[2],h[3]);
  return 0;
}
