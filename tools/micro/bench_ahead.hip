// micro-benchmark + check of panel_factor_ahead against panel_factor_mfma (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
__device__ long long* g_pf_out = nullptr;
__shared__ long long s_pf[8];
#define DLG_PF_DECL if(threadIdx.x == 0) { for(int _i = 0; _i < 6; _i++) s_pf[_i] = 0; s_pf[7] = clock64(); }
#define DLG_PF_STAMP(i) do { if(threadIdx.x == 0) { const long long _n = clock64(); s_pf[i] += _n - s_pf[7]; s_pf[7] = _n; } } while(0)
#define DLG_PF_PIN(x) asm volatile("" :: "v"(x))
#define DLG_PF_DONE if(threadIdx.x == 0 && g_pf_out) { for(int _i = 0; _i < 6; _i++) g_pf_out[_i] = s_pf[_i]; }
#include "../../libdogleg_amd/csrc/panel_factor.h"

template <int NT, int MODE>
__global__ void __launch_bounds__(NT) k_panel(double* G, int nrows, int w, int* info, long long* stamps)
{
  extern __shared__ __attribute__((aligned(16))) double P[];
  __shared__ double s_dv[PF_AHEAD_MAXW]; __shared__ int s_dn;
  double* g = G + (size_t)blockIdx.x*nrows*w;
  const int tid = threadIdx.x;
  const int ldp = (nrows + 1) & ~1;
  for(int e = tid; e < nrows*w; e += NT) { int j = e / nrows; P[e + j*(ldp - nrows)] = g[e]; }
  __syncthreads();
  long long t1 = clock64();
  if(MODE == 0) panel_factor_mfma<NT>(P, ldp, nrows, w, tid, info, 0);
  else          { if(blockIdx.x == 0 && tid == 0) g_pf_out = stamps + 2; panel_factor_ahead<NT>(P, ldp, nrows, w, tid, info, 0, s_dv, &s_dn); }
  __syncthreads();
  long long t2 = clock64();
  for(int e = tid; e < nrows*w; e += NT) { int j = e / nrows; g[e] = P[e + j*(ldp - nrows)]; }
  if(tid == 0 && blockIdx.x == 0) stamps[0] = t2 - t1;
}

template <int NT>
void run(int nrows, int w, int G)
{
  const size_t n = (size_t)nrows*w;
  std::vector<double> h(n*G), r0(n*G), r1(n*G);
  for(int b = 0; b < G; b++)
    for(int j = 0; j < w; j++)
      for(int i = 0; i < nrows; i++)
        h[b*n + i + (size_t)j*nrows] = (i == j) ? (double)(w + 1) : ((i < w && i < j) ? 0.0 : 0.3*sin(0.37*i + 1.3*j + b));
  double* d; int* info; long long* st;
  hipMalloc(&d, n*G*8); hipMalloc(&info, 4); hipMalloc(&st, 128);
  const int lds = (int)(((nrows + 1) & ~1)*w*8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_panel<NT, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_panel<NT, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  long long c0 = 0, c1 = 0, ph[6] = {0};
  for(int rep = 0; rep < 3; rep++)
  {
    int big = 0x7fffffff; hipMemcpy(info, &big, 4, hipMemcpyHostToDevice);
    hipMemcpy(d, h.data(), n*G*8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_panel<NT, 0>), dim3(G), dim3(NT), lds, 0, d, nrows, w, info, st);
    hipMemcpy(r0.data(), d, n*G*8, hipMemcpyDeviceToHost); hipMemcpy(&c0, st, 8, hipMemcpyDeviceToHost);
    hipMemcpy(d, h.data(), n*G*8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_panel<NT, 1>), dim3(G), dim3(NT), lds, 0, d, nrows, w, info, st);
    hipMemcpy(r1.data(), d, n*G*8, hipMemcpyDeviceToHost); hipMemcpy(&c1, st, 8, hipMemcpyDeviceToHost); hipMemcpy(ph, st + 2, 48, hipMemcpyDeviceToHost);
  }
  double worst = 0, big = 0;
  for(int b = 0; b < G; b++)
    for(int j = 0; j < w; j++)
      for(int i = j; i < nrows; i++)
      {
        const size_t e = b*n + i + (size_t)j*nrows;
        worst = fmax(worst, fabs(r0[e] - r1[e])); big = fmax(big, fabs(r0[e]));
      }
  int inf; hipMemcpy(&inf, info, 4, hipMemcpyDeviceToHost);
  printf("   wave 0 of ahead: mfma %lld  loadD %lld  factor %lld  writeback %lld  solve %lld\n", ph[0], ph[1], ph[2], ph[3], ph[4]);
  printf("NT=%d nrows=%d w=%d G=%d: mfma %lld cycles, ahead %lld cycles, max|diff| = %.3e (max|L| %.3g) info %d\n", NT, nrows, w, G, c0, c1, worst, big, inf);
  hipFree(d); hipFree(info); hipFree(st);
}

int main(int argc, char** argv)
{
  if(argc > 1) { run<512>(187, 60, 1); run<512>(187, 96, 1); return 0; }
  run<512>(187, 60, 1); run<512>(187, 60, 64); run<512>(193, 60, 128); run<512>(187, 96, 1); run<512>(91, 90, 1);
  run<256>(187, 60, 1); run<512>(300, 37, 4); run<512>(60, 60, 1); run<512>(130, 128, 1); run<512>(700, 21, 2);
  run<256>(250, 48, 8); run<512>(17, 3, 1); run<512>(9, 9, 1);
  return 0;
}
