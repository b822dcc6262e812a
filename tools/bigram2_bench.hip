// tools/bigram2_bench.hip — times and VALIDATES the second-generation order-2 pipeline (csrc/bigram2.hpp) on a synthetic Zipf class stream:
// device result (surviving bigrams with counts, order-3 active list, counters) against a host-side sort/count of the same stream.
// Not part of the product.   usage: bigram2_bench [npos] [nsub] [grid_per_cu] [validate 0/1]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels.hpp"
#include "binned.hpp"
#include "bigram2.hpp"
using namespace colibri;
#define CK(x) do{hipError_t err_=(x); if(err_!=hipSuccess){printf("%s: %s (line %d)\n",#x,hipGetErrorString(err_),__LINE__); exit(1);} }while(0)
__global__ void gen(uint32_t* cls, uint32_t n, float lnV, uint32_t seed){
  for(uint32_t i=blockIdx.x*blockDim.x+threadIdx.x;i<n;i+=gridDim.x*blockDim.x){
    const uint64_t h=mix64(((uint64_t)seed<<32)|(i+1)); float u=(h>>40)*(1.0f/16777216.0f); uint32_t r=(uint32_t)__expf(u*lnV);
    cls[i]= ((mix64(h)%21)==20 || i+1==n)?0u:(r+5u); }
}
__global__ void count1(const uint32_t* cls, uint32_t n, uint32_t* cnt){ for(uint32_t i=blockIdx.x*blockDim.x+threadIdx.x;i<n;i+=gridDim.x*blockDim.x) if(cls[i]) atomicAdd(&cnt[cls[i]],1u); }
__global__ void surv1(const uint32_t* cnt, uint32_t ncls, uint32_t thr, uint32_t* surv){ uint32_t w=blockIdx.x*blockDim.x+threadIdx.x; if(w*32<ncls){ uint32_t b=0; for(int k=0;k<32;k++){ uint32_t c=w*32+k; if(c<ncls && cnt[c]>=thr) b|=1u<<k;} surv[w]=b; } }

int main(int argc,char**argv){
  const uint32_t n = argc>1? (uint32_t)atoll(argv[1]) : 105000000u;
  const uint32_t nsub = argc>2? atoi(argv[2]) : 8;
  const uint32_t gpc = argc>3? atoi(argv[3]) : 2;
  const int validate = argc>4? atoi(argv[4]) : (n<=20000000u);
  const uint32_t V=1000000, ncls=V+8, thr=2;
  uint32_t *cls,*cnt,*surv; CK(hipMalloc(&cls,((size_t)n+128)*4)); CK(hipMemset(cls,0,((size_t)n+128)*4)); CK(hipMalloc(&cnt,(size_t)ncls*4)); CK(hipMalloc(&surv,(size_t)(ncls/32+4)*4));
  CK(hipMemset(cnt,0,(size_t)ncls*4)); CK(hipMemset(surv,0,(size_t)(ncls/32+4)*4));
  hipLaunchKernelGGL(gen,dim3(4096),dim3(256),0,0,cls,n,logf((float)V),12345u);
  hipLaunchKernelGGL(count1,dim3(4096),dim3(256),0,0,cls,n,cnt);
  hipLaunchKernelGGL(surv1,dim3((ncls/32+256)/256),dim3(256),0,0,cnt,ncls,thr,surv); CK(hipDeviceSynchronize());
  const uint32_t nslots=kBins*nsub;
  const uint32_t region=(uint32_t)(((size_t)n*2)/nslots + 4096);
  unsigned long long *recsA,*recsB; CK(hipMalloc(&recsA,(size_t)nslots*region*8)); CK(hipMalloc(&recsB,(size_t)nslots*region*8));
  uint32_t* boff; CK(hipMalloc(&boff,(size_t)nslots*(kBi2BBins+1)*4));
  uint32_t *sp_rep,*sp_cnt,*res_rep,*res_cnt,*list3,*nlist3; CK(hipMalloc(&sp_rep,(size_t)n*4+64)); CK(hipMalloc(&sp_cnt,(size_t)n*4+64)); CK(hipMalloc(&res_rep,(size_t)n*4+64)); CK(hipMalloc(&res_cnt,(size_t)n*4+64));
  CK(hipMalloc(&list3,(size_t)n*4+64)); CK(hipMalloc(&nlist3,64));
  uint32_t pshift=12; while(((uint64_t)n>>pshift) > (uint64_t)kBi2Buckets-1) ++pshift;
  const uint32_t nbuckets=(uint32_t)(((uint64_t)n+(1u<<pshift)-1)>>pshift);
  Bi2Lists pl; pl.pshift=pshift; pl.pcap=(1u<<pshift)/4+4096; uint32_t* plist; CK(hipMalloc(&plist,(size_t)kBi2Shards*kBi2Buckets*pl.pcap*4));
  const uint32_t egrid=256*gpc/ nsub * nsub; uint32_t* head_rows; CK(hipMalloc(&head_rows,(size_t)egrid*2*kBi2HeadN*4));
  DevState* st; CK(hipMalloc(&st,sizeof(DevState))); Bi2State* bs; CK(hipMalloc(&bs,sizeof(Bi2State)));
  const uint32_t W = 256*(argc>5?atoi(argv[5]):16);
  const uint32_t wcap=(uint32_t)(((uint64_t)n*6/10/W)*2+4096);
  uint32_t *wlist,*wcnt; CK(hipMalloc(&wlist,(size_t)W*wcap*4)); CK(hipMalloc(&wcnt,(size_t)W*4)); CK(hipMemset(wcnt,0,(size_t)W*4));
  const size_t bm_bytes=((size_t)(1u<<pshift)/32)*4;
  CK(hipFuncSetAttribute((const void*)bi2_bitmap_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bm_bytes));
  uint32_t clsbits=1; while((1u<<clsbits)<ncls) ++clsbits; uint32_t* headsurv; CK(hipMalloc(&headsurv,kBi2HeadN/8));
  uint32_t* bitmap; CK(hipMalloc(&bitmap,((size_t)n/32+8)*4)); CK(hipMemset(bitmap,0,((size_t)n/32+8)*4));
  printf("npos %u nsub %u region %u nslots %u pshift %u nbuckets %u egrid %u waves %u wcap %u sizeof(Bi2State) %zu\n",n,nsub,region,nslots,pshift,nbuckets,egrid,W,wcap,sizeof(Bi2State));
  if(nsub!=8){printf("count kernel is instantiated for nsub = 8\n"); return 1;}
  hipEvent_t ev[16]; for(auto&e:ev) CK(hipEventCreate(&e));
  float best[16]; for(auto&b:best) b=1e9f; float best_total=1e9f;
  const char* names[]={"memsets","emit","head_reduce+offsets","levelB","binoff","count","scan+finish+compact","pospart","bitmap","list3"};
  for(int rep=0;rep<4;rep++){
    CK(hipMemset(st,0,sizeof(DevState))); CK(hipDeviceSynchronize());
    int e=0;
    CK(hipEventRecord(ev[e++]));
    CK(hipMemsetAsync(bs,0,sizeof(Bi2State))); CK(hipMemsetAsync(nlist3,0,64));
    CK(hipEventRecord(ev[e++]));
    hipLaunchKernelGGL(bi2_emit_kernel,dim3(egrid),dim3(kBi2Threads),0,0,cls,surv,ncls/32+4,n,clsbits,0u,0u,27u,recsA,region,nsub,bs,st,head_rows);
    CK(hipEventRecord(ev[e++]));
    hipLaunchKernelGGL(bi2_head_reduce_kernel,dim3(kBi2HeadN/kBlock,kBi2HeadSplit),dim3(kBlock),0,0,head_rows,egrid,bs,st);
    hipLaunchKernelGGL(bi2_offsets_kernel,dim3(1),dim3(kBlock),0,0,bs,region,nsub,st);
    CK(hipEventRecord(ev[e++]));
    hipLaunchKernelGGL(bi2_levelB_kernel,dim3(nslots),dim3(kBi2Threads),0,0,recsA,recsB,region,bs,boff,st);
    CK(hipEventRecord(ev[e++]));
    hipLaunchKernelGGL(bi2_binoff_kernel,dim3(kBins),dim3(kBi2BBins),0,0,bs,boff,nsub,st);
    CK(hipEventRecord(ev[e++]));
    hipLaunchKernelGGL((bi2_count_kernel<8>),dim3(W),dim3(kWave),0,0,recsB,region,boff,bs,st,thr,sp_rep,sp_cnt,wlist,wcnt,wcap,true);
    CK(hipEventRecord(ev[e++]));
    hipLaunchKernelGGL(bi2_kept_scan_kernel,dim3(kBins),dim3(kBi2BBins),0,0,bs,st);
    hipLaunchKernelGGL(bi2_finish_kernel,dim3(1),dim3(kBlock),0,0,st,bs,thr,n,headsurv);
    hipLaunchKernelGGL(bi2_compact_kernel,dim3(1025),dim3(kBlock),0,0,sp_rep,sp_cnt,st,bs,res_rep,res_cnt,n);
    CK(hipEventRecord(ev[e++]));
    hipLaunchKernelGGL(bi2_pospart_kernel,dim3(512),dim3(kBi2Threads),0,0,wlist,wcnt,W,wcap,bs,st,plist,pl);
    CK(hipEventRecord(ev[e++]));
    hipLaunchKernelGGL(bi2_bitmap_kernel,dim3(nbuckets),dim3(kBi2BmThreads),bm_bytes,0,n,bs,plist,pl,st,bitmap);
    CK(hipEventRecord(ev[e++]));
    hipLaunchKernelGGL(bi2_list3_kernel,dim3(2048),dim3(kBlock),0,0,cls,surv,n,headsurv,bitmap,st,list3,nlist3);
    CK(hipEventRecord(ev[e++]));
    CK(hipDeviceSynchronize()); CK(hipGetLastError());
    float tot; CK(hipEventElapsedTime(&tot,ev[0],ev[e-1])); if(tot<best_total)best_total=tot;
    for(int k=0;k+1<e;k++){ float ms; CK(hipEventElapsedTime(&ms,ev[k],ev[k+1])); if(ms<best[k])best[k]=ms; }
  }
#ifdef BI2_PROF
  { unsigned long long acc[16]; CK(hipMemcpyFromSymbol(acc,HIP_SYMBOL(bi2_prof),sizeof acc)); const char* ph[]={"loop top","bounds","issue loads","size+init","pass1 regs","pass1 stream","scan","mark","pass2 regs","pass2 stream","rep write","-"};
    unsigned long long tot=0; for(int k=0;k<12;k++) tot+=acc[k]; for(int k=0;k<11;k++) printf("  phase %-18s %6.2f %%   %.2f us per wave (avg of 4 reps)\n",ph[k],100.0*acc[k]/tot,acc[k]/100.0/W/4); }
#endif
  for(int k=0;k<10;k++) printf("%-22s %8.3f ms\n",names[k],best[k]);
  printf("%-22s %8.3f ms\n","TOTAL order 2",best_total);
  static Bi2State h; DevState hs; CK(hipMemcpy(&h,bs,sizeof(Bi2State),hipMemcpyDeviceToHost)); CK(hipMemcpy(&hs,st,sizeof(DevState),hipMemcpyDeviceToHost));
  uint32_t hn3; CK(hipMemcpy(&hn3,nlist3,4,hipMemcpyDeviceToHost));
  uint32_t maxslot=0; for(uint32_t s=0;s<nslots;s++) maxslot=std::max(maxslot,h.curA[s]);
  printf("nrec %u bshift %u overflow %u admitted %u found %u kept %u (bins %u head %u) valid %u list3 %u maxslot %u (region %u)\n",h.nrec,h.bshift,h.overflow,hs.admitted,hs.found,hs.kept,h.kept_bins,h.kept_head,hs.valid,hn3,maxslot,region);
  if(!validate) return 0;
  // ---- host reference ----
  std::vector<uint32_t> c(n); CK(hipMemcpy(c.data(),cls,(size_t)n*4,hipMemcpyDeviceToHost));
  std::vector<uint32_t> hsurv(ncls/32+4); CK(hipMemcpy(hsurv.data(),surv,hsurv.size()*4,hipMemcpyDeviceToHost));
  auto sv=[&](uint32_t x){ return (hsurv[x>>5]>>(x&31))&1u; };
  std::vector<uint64_t> keys; keys.reserve(n);
  for(uint32_t i=0;i+1<n;i++) if(c[i]&&c[i+1]&&sv(c[i])&&sv(c[i+1])) keys.push_back(((uint64_t)c[i]<<21)|c[i+1]);
  const uint64_t adm=keys.size(); std::sort(keys.begin(),keys.end());
  std::vector<std::pair<uint64_t,uint32_t>> want; uint64_t found=0;
  for(size_t i=0;i<keys.size();){ size_t j=i; while(j<keys.size()&&keys[j]==keys[i])++j; ++found; if(j-i>=thr) want.push_back({keys[i],(uint32_t)(j-i)}); i=j; }
  std::vector<uint32_t> rr(hs.kept),rc(hs.kept); CK(hipMemcpy(rr.data(),res_rep,(size_t)hs.kept*4,hipMemcpyDeviceToHost)); CK(hipMemcpy(rc.data(),res_cnt,(size_t)hs.kept*4,hipMemcpyDeviceToHost));
  std::vector<std::pair<uint64_t,uint32_t>> got; got.reserve(hs.kept); int bad=0;
  for(uint32_t k=0;k<hs.kept;k++){ uint32_t p=rr[k]; if(p+1>=n||!c[p]||!c[p+1]){ if(bad++<5)printf("bad rep %u at %u\n",p,k); continue;} got.push_back({((uint64_t)c[p]<<21)|c[p+1],rc[k]}); }
  std::sort(got.begin(),got.end());
  // set of keys that survived -> expected list3 and valid
  std::vector<uint64_t> wk; for(auto&w:want) wk.push_back(w.first);
  auto alive=[&](uint32_t i){ if(i+1>=n||!c[i]||!c[i+1]||!sv(c[i])||!sv(c[i+1])) return false; return std::binary_search(wk.begin(),wk.end(),((uint64_t)c[i]<<21)|c[i+1]); };
  std::vector<uint8_t> al(n+1,0); uint64_t valid=0; for(uint32_t i=0;i<n;i++){ al[i]=alive(i); valid+=al[i]; }
  std::vector<uint32_t> want3; for(uint32_t i=0;i+1<n;i++) if(al[i]&&al[i+1]) want3.push_back(i);
  std::vector<uint32_t> got3(hn3); CK(hipMemcpy(got3.data(),list3,(size_t)hn3*4,hipMemcpyDeviceToHost)); std::sort(got3.begin(),got3.end());
  printf("host: admitted %llu found %llu kept %zu valid %llu list3 %zu\n",(unsigned long long)adm,(unsigned long long)found,want.size(),(unsigned long long)valid,want3.size());
  const bool ok = !bad && h.overflow==0 && adm==hs.admitted && found==hs.found && want.size()==hs.kept && got==want && valid==hs.valid && got3==want3;
  printf(ok?"VALIDATION OK\n":"VALIDATION FAILED\n");
  // bijection spot check of mix42 on the host
  return ok?0:1;
}
