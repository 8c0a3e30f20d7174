// standalone timing harness for the K1 kernels: dlopen a libscae variant, run on random data
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "scae_hip.h"
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
static float* dev(const std::vector<float>& h){ float* d; CK(hipMalloc(&d,h.size()*4)); CK(hipMemcpy(d,h.data(),h.size()*4,hipMemcpyHostToDevice)); return d; }
static float* devz(size_t n){ float* d; CK(hipMalloc(&d,n*4)); CK(hipMemset(d,0,n*4)); return d; }
static float frand(){ return rand()/(float)RAND_MAX; }
int main(int argc,char**argv){
  const char* libp = argc>1?argv[1]:"./libscae_hip.so";
  int B=argc>2?atoi(argv[2]):128, M=argc>3?atoi(argv[3]):24, C=argc>4?atoi(argv[4]):1, H=argc>5?atoi(argv[5]):40, W=H, th=11,tw=11, K=M+1;
  void* h=dlopen(libp,RTLD_NOW); if(!h){printf("dlopen %s\n",dlerror());return 1;}
  auto render=(int(*)(const scae_decoder_desc*,float*,float*,void*))dlsym(h,"scae_template_render_fwd_f32");
  auto lpf=(int(*)(const scae_decoder_desc*,const float*,float*,float*,float*,void*))dlsym(h,"scae_render_gmm_logprob_fwd_f32");
  auto bwd=(int(*)(const scae_decoder_desc*,const float*,const float*,const float*,const float*,const float*,const float*,float*,float*,float*,float*,float*,float*,void*))dlsym(h,"scae_render_gmm_bwd_f32");
  srand(1);
  std::vector<float> t((size_t)B*M*C*th*tw), a((size_t)M*th*tw), pose((size_t)B*M*6), pres((size_t)B*M), x((size_t)B*C*H*W), g((size_t)B*C*H*W,1.f);
  for(auto&v:t)v=frand(); for(auto&v:a)v=frand()-0.5f; for(auto&v:pres)v=frand(); for(auto&v:x)v=frand();
  for(size_t i=0;i<pose.size();++i){ float n=(frand()-0.5f)*0.6f; int j=i%6; pose[i]=n*0.5f+((j==0||j==4)?0.55f:0.f);}    
  std::vector<float> one(1,0.f);
  scae_decoder_desc d{dev(t),dev(a),dev(pose),dev(pres),nullptr,dev(one),dev(one),nullptr,nullptr,B,M,C,th,tw,H,W};
  float*dx=dev(x),*dg=dev(g);
  float*tt=devz((size_t)B*K*C*H*W),*ml=devz((size_t)B*K*H*W),*lp=devz((size_t)B*C*H*W),*lpo=devz((size_t)B*C*H*W),*lpr=devz((size_t)B*H*W);
  float*gt=devz((size_t)B*M*C*th*tw),*ga=devz((size_t)B*M*th*tw),*gp=devz((size_t)B*M*6),*gpr=devz((size_t)B*M),*gs=devz((size_t)B*K*4);
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit=[&](const char*name, auto fn){ for(int i=0;i<5;++i) fn(); CK(hipDeviceSynchronize()); float best=1e9; for(int r=0;r<3;++r){ CK(hipEventRecord(e0,0)); for(int i=0;i<100;++i) fn(); CK(hipEventRecord(e1,0)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); if(ms<best)best=ms;} printf("%-14s %8.2f us\n",name,best*10.f); };
  timeit("render_fwd",[&]{ int rc=render(&d,tt,ml,0); if(rc){printf("rc %d\n",rc);exit(1);} });
  timeit("logprob_fwd",[&]{ int rc=lpf(&d,dx,lp,lpo,lpr,0); if(rc){printf("rc %d\n",rc);exit(1);} });
  timeit("bwd_fused",[&]{ int rc=bwd(&d,dx,lpo,lpr,dg,nullptr,nullptr,gt,ga,gp,gpr,nullptr,gs,0); if(rc){printf("rc %d\n",rc);exit(1);} });
  std::vector<float> hgp(B*M*6); CK(hipMemcpy(hgp.data(),gp,hgp.size()*4,hipMemcpyDeviceToHost)); double s=0; for(float v:hgp)s+=fabs(v); printf("checksum gpose %.6f\n",s);
  return 0;
}
