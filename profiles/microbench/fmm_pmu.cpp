// fmm_pmu.cpp -- cycles / instructions / branch misses per accepted node of the eikonal sources' fast-marching solve (plain and optimised,
// kiwi_amd/csrc/kiwi_host_fmm.hpp) on BASELINE config 4's 1200 x 360 grid, through perf_event_open (works on the GPU box, not in the dev container).
//   clang++ -O3 -std=c++17 -fno-fast-math -ffp-contract=off -fPIC -o fmm_pmu fmm_pmu.cpp && ./fmm_pmu
#include "../../kiwi_amd/csrc/kiwi_host_fmm.hpp"
#include <linux/perf_event.h>
#include <sys/syscall.h>
#include <sys/ioctl.h>
#include <unistd.h>
#include <cstdio>
#include <cerrno>
#include <chrono>
using namespace kiwi::eik;
static int openc(unsigned type, unsigned long long config){ perf_event_attr pe; memset(&pe,0,sizeof pe); pe.type=type; pe.size=sizeof pe; pe.config=config; pe.disabled=1; pe.exclude_kernel=1; pe.exclude_hv=1; return syscall(SYS_perf_event_open,&pe,0,-1,-1,0); }
static void make_grid(int nx, int ny, float relv, std::vector<float> &sp){ sp.assign((size_t)nx*ny,0.f); float mn=1e30f; for(int y=0;y<ny;y++)for(int x=0;x<nx;x++){ const float px=(x+0.5f-nx*0.5f)*25.f, depth=6500.f+(y+0.5f)*25.f, pz=depth-11000.f; if(std::sqrt(px*px+pz*pz)>15000.f) continue; float v=(depth<=12000.f?3500.f:3700.f)*relv; sp[(size_t)y*nx+x]=v; mn=std::min(mn,v);} for(auto&v:sp) if(v==0.f) v=mn*0.5f; }
int main(){ std::vector<float> sp,t; make_grid(1200,360,0.9f,sp); float mn=*std::min_element(sp.begin(),sp.end()); float origin[2]={-15000.f,-4500.f}, delta[2]={25.f,25.f};
 int fd[4]={openc(PERF_TYPE_HARDWARE,PERF_COUNT_HW_CPU_CYCLES),openc(PERF_TYPE_HARDWARE,PERF_COUNT_HW_INSTRUCTIONS),openc(PERF_TYPE_HARDWARE,PERF_COUNT_HW_BRANCH_MISSES),openc(PERF_TYPE_HARDWARE,PERF_COUNT_HW_BRANCH_INSTRUCTIONS)};
 printf("perf fds %d %d %d %d (errno %d)\n",fd[0],fd[1],fd[2],fd[3],errno);
 for(int which=0;which<2;which++) for(int r=0;r<3;r++){ float start[2]={2000.f+100.f*r,0.f};
   for(int i=0;i<4;i++) if(fd[i]>=0){ ioctl(fd[i],PERF_EVENT_IOC_RESET,0); ioctl(fd[i],PERF_EVENT_IOC_ENABLE,0);} 
   auto t0=std::chrono::steady_clock::now();
   if(which==0) fast_marching_plain(sp.data(),1200,360,origin,delta,start,t,mn); else fast_marching(sp.data(),1200,360,origin,delta,start,t,mn);
   double ms=std::chrono::duration<double,std::milli>(std::chrono::steady_clock::now()-t0).count();
   long long v[4]={0,0,0,0}; for(int i=0;i<4;i++) if(fd[i]>=0){ ioctl(fd[i],PERF_EVENT_IOC_DISABLE,0); read(fd[i],&v[i],8);} 
   printf("%s: %.2f ms | per node: cycles %.1f instr %.1f br-miss %.2f branches %.1f\n",which?"fast ":"plain",ms,v[0]/430144.0,v[1]/430144.0,v[2]/430144.0,v[3]/430144.0); }
}
