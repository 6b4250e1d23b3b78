// where the cycles of panel_factor_mfma's 8-column step go (wave 0's clock; tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
__device__ long long* g_pf_out = nullptr;
__shared__ long long s_pf[10];
#define DLG_PF_DECL if(threadIdx.x == 0) { for(int _i = 0; _i < 8; _i++) s_pf[_i] = 0; s_pf[9] = clock64(); }
#define DLG_PF_STAMP(i) do { if(threadIdx.x == 0) { const long long _n = clock64(); s_pf[i] += _n - s_pf[9]; s_pf[9] = _n; } } while(0)
#define DLG_PF_PIN(x) asm volatile("" :: "v"(x))
#define DLG_PF_DONE if(threadIdx.x == 0 && g_pf_out) { for(int _i = 0; _i < 8; _i++) g_pf_out[_i] = s_pf[_i]; }
#include "../../libdogleg_amd/csrc/panel_factor.h"

template <int NT>
__global__ void __launch_bounds__(NT) k_panel(double* G, int nrows, int w, int* info, long long* stamps)
{
  extern __shared__ __attribute__((aligned(16))) double P[];
  const int tid = threadIdx.x;
  const int ldp = (nrows + 1) & ~1;
  for(int e = tid; e < nrows*w; e += NT) { int j = e / nrows; P[e + j*(ldp - nrows)] = G[e]; }
  if(tid == 0) g_pf_out = stamps + 2;
  __syncthreads();
  long long t1 = clock64();
  panel_factor_mfma<NT>(P, ldp, nrows, w, tid, info, 0);
  __syncthreads();
  long long t2 = clock64();
  for(int e = tid; e < nrows*w; e += NT) { int j = e / nrows; G[e] = P[e + j*(ldp - nrows)]; }
  if(tid == 0) stamps[0] = t2 - t1;
}
template <int NT>
void run(int nrows, int w)
{
  const size_t n = (size_t)nrows*w;
  std::vector<double> h(n);
  for(int j = 0; j < w; j++)
    for(int i = 0; i < nrows; i++)
      h[i + (size_t)j*nrows] = (i == j) ? (double)(w + 1) : ((i < w && i < j) ? 0.0 : 0.3*sin(0.37*i + 1.3*j));
  double* d; int* info; long long* st;
  hipMalloc(&d, n*8); hipMalloc(&info, 4); hipMalloc(&st, 128);
  const int lds = (int)(((nrows + 1) & ~1)*w*8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_panel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  long long c[10] = {0};
  for(int rep = 0; rep < 3; rep++)
  {
    int big = 0x7fffffff; hipMemcpy(info, &big, 4, hipMemcpyHostToDevice);
    hipMemcpy(d, h.data(), n*8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_panel<NT>), dim3(1), dim3(NT), lds, 0, d, nrows, w, info, st);
    hipMemcpy(c, st, 80, hipMemcpyDeviceToHost);
  }
  const int nblk = (w + 7)/8;
  printf("NT=%d %d x %d: %lld cycles (%lld per 8 columns): tiles(wave0) %lld  loadD %lld  factor %lld  writeback %lld  barrier %lld  solve %lld  barrier %lld\n",
         NT, nrows, w, c[0], c[0]/nblk, c[2+2]/nblk, c[2+3]/nblk, c[2+6]/nblk, c[2+0]/nblk, c[2+1]/nblk, c[2+4]/nblk, c[2+5]/nblk);
  hipFree(d); hipFree(info); hipFree(st);
}
int main()
{
  run<512>(187, 60); run<512>(205, 66); run<512>(187, 96); run<512>(91, 90); run<512>(2000, 64); run<256>(187, 60);
  return 0;
}
