// micro-benchmark of panel_factor (tools only): times the in-LDS panel Cholesky for a
// given (nrows, w) with G workgroups, and reports per-phase cycle shares via s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "../../libdogleg_amd/csrc/panel_factor.h"

template <int NT, bool ALIGNED16>
__device__ __forceinline__ void panel_factor_prof(double* P, int ldp, int nrows, int w, int tid,
                                             int* __restrict__ info, int col0, long long* ph)
{
  long long c1 = 0, c2 = 0, c3 = 0, tA, tB;
  for(int kb = 0; kb < w; kb += 8)
  {
    const int nb = (w - kb < 8) ? w - kb : 8;
    tA = clock64();
    for(int r = kb + tid; r < nrows; r += NT)
    {
      double x[8];
#pragma unroll
      for(int c = 0; c < 8; c++) x[c] = (c < nb) ? P[r + (kb + c)*ldp] : 0.0;
#pragma unroll 4
      for(int k = 0; k < kb; k++)
      {
        const double a = P[r + k*ldp];
        const double* bp = P + kb + k*ldp;
        if(ALIGNED16)
        {
          const double2 b0 = *reinterpret_cast<const double2*>(bp);
          const double2 b1 = *reinterpret_cast<const double2*>(bp + 2);
          const double2 b2 = *reinterpret_cast<const double2*>(bp + 4);
          const double2 b3 = *reinterpret_cast<const double2*>(bp + 6);
          x[0] -= a*b0.x; x[1] -= a*b0.y; x[2] -= a*b1.x; x[3] -= a*b1.y;
          x[4] -= a*b2.x; x[5] -= a*b2.y; x[6] -= a*b3.x; x[7] -= a*b3.y;
        }
        else
        {
#pragma unroll
          for(int c = 0; c < 8; c++) if(c < nb) x[c] -= a*bp[c];
        }
      }
#pragma unroll
      for(int c = 0; c < 8; c++) if(c < nb) P[r + (kb + c)*ldp] = x[c];
    }
    __syncthreads();
    tB = clock64(); c1 += tB - tA; tA = tB;
    double D[8][8];
    const bool need_d = (kb + tid < nrows);      // threads without a row >= kb never use the block
#pragma unroll
    for(int c = 0; c < 8; c++)
#pragma unroll
      for(int q = 0; q <= c; q++)
        D[c][q] = (c < nb && need_d) ? P[(kb + c) + (kb + q)*ldp] : ((c == q) ? 1.0 : 0.0);
    double Dinv[8] = {1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0};
    bool bad = false; int badcol = 0;
    if(need_d)
#pragma unroll
    for(int c = 0; c < 8; c++)
    {
      double d = D[c][c];
#pragma unroll
      for(int q = 0; q < c; q++) d -= D[c][q]*D[c][q];
      if(!(d > 0.0)) { if(!bad) { bad = true; badcol = c; } d = 1.0; }
      const double inv = dlg_rsqrt(d);
      const double piv = d*inv;
      D[c][c] = piv;
      Dinv[c] = inv;
#pragma unroll
      for(int i = c + 1; i < 8; i++)
      {
        double v = D[i][c];
#pragma unroll
        for(int q = 0; q < c; q++) v -= D[i][q]*D[c][q];
        D[i][c] = v*inv;
      }
    }
    if(bad && tid == 0) atomicMin(info, col0 + kb + badcol);
    __syncthreads();
    tB = clock64(); c2 += tB - tA; tA = tB;
    for(int r = kb + tid; r < nrows; r += NT)
    {
      if(r < kb + nb)
      {
        const int c = r - kb;
#pragma unroll
        for(int cc = 0; cc < 8; cc++)
#pragma unroll
          for(int q = 0; q <= cc; q++) if(cc == c) P[r + (kb + q)*ldp] = D[cc][q];
      }
      else
      {
        double x[8];
#pragma unroll
        for(int c = 0; c < 8; c++) x[c] = (c < nb) ? P[r + (kb + c)*ldp] : 0.0;
#pragma unroll
        for(int c = 0; c < 8; c++)
        {
          double v = x[c];
#pragma unroll
          for(int q = 0; q < c; q++) v -= x[q]*D[c][q];
          x[c] = v*Dinv[c];
        }
#pragma unroll
        for(int c = 0; c < 8; c++) if(c < nb) P[r + (kb + c)*ldp] = x[c];
      }
    }
    __syncthreads();
    tB = clock64(); c3 += tB - tA;
  }
  if(tid == 0) { ph[0] = c1; ph[1] = c2; ph[2] = c3; }
}

template <int NT, int MODE>
__global__ void __launch_bounds__(NT) k_panel(double* G, int nrows, int w, int* info, long long* stamps)
{
  extern __shared__ __attribute__((aligned(16))) double P[];
  double* g = G + (size_t)blockIdx.x*nrows*w;
  const int tid = threadIdx.x;
  const int ldp = (nrows + 1) & ~1;
  long long t0 = clock64();
  for(int base = 0; base < nrows*w; base += 8*NT)
  {
    double v[8];
#pragma unroll
    for(int u = 0; u < 8; u++) { int e = base + u*NT + tid; v[u] = e < nrows*w ? g[e] : 0.0; }
#pragma unroll
    for(int u = 0; u < 8; u++) { int e = base + u*NT + tid; if(e < nrows*w) { int j = e / nrows; P[e + j*(ldp - nrows)] = v[u]; } }
  }
  __syncthreads();
  long long t1 = clock64();
  if(MODE == 0) panel_factor<NT, true>(P, ldp, nrows, w, tid, info, 0);
  if(MODE == 1) panel_factor_prof<NT, true>(P, ldp, nrows, w, tid, info, 0, stamps + 4);
  __syncthreads();
  long long t2 = clock64();
  for(int e = tid; e < nrows*w; e += NT) { int j = e / nrows; g[e] = P[e + j*(ldp - nrows)]; }
  long long t3 = clock64();
  if(tid == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = t2 - t1; stamps[2] = t3 - t2; }
}

template <int NT>
void run(int nrows, int w, int G, int iters)
{
  const size_t n = (size_t)nrows*w;
  std::vector<double> h(n*G);
  for(int b = 0; b < G; b++)
    for(int j = 0; j < w; j++)
      for(int i = 0; i < nrows; i++)
        h[b*n + i + (size_t)j*nrows] = (i == j) ? (double)(w + 1) : ((i < w && i < j) ? 0.0 : 0.3*sin(0.37*i + 1.3*j));
  double* d; int* info; long long* st;
  hipMalloc(&d, n*G*8); hipMalloc(&info, 4); hipMalloc(&st, 64);
  const int lds = (int)(((nrows + 1) & ~1)*w*8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_panel<NT, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for(int rep = 0; rep < 3; rep++)
  {
    hipMemcpy(d, h.data(), n*G*8, hipMemcpyHostToDevice);
    hipEventRecord(e0);
    for(int it = 0; it < iters; it++) { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_panel<NT, 0>), dim3(G), dim3(NT), lds, 0, d, nrows, w, info, st); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if(ms < best) best = ms;
  }
  long long hs[3]; hipMemcpy(hs, st, 24, hipMemcpyDeviceToHost);
  printf("NT=%d nrows=%d w=%d G=%d: %.1f us/launch   cycles load %lld factor %lld store %lld\n", NT, nrows, w, G,
         best*1e3/iters, hs[0], hs[1], hs[2]);
  hipMemcpy(d, h.data(), n*G*8, hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_panel<NT, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_panel<NT, 1>), dim3(G), dim3(NT), lds, 0, d, nrows, w, info, st);
  hipDeviceSynchronize();
  long long ph[7]; hipMemcpy(ph, st, 56, hipMemcpyDeviceToHost);
  printf("      phases (cycles, thread 0): update-sweep %lld  8x8-factor %lld  row-solve %lld\n", ph[4], ph[5], ph[6]);
  hipFree(d); hipFree(info); hipFree(st);
}

int main(int argc, char** argv)
{
  const int G = argc > 1 ? atoi(argv[1]) : 64;
  run<512>(170, 96, G, 20); run<256>(170, 96, G, 20); run<128>(170, 96, G, 20);
  run<256>(128, 64, G, 20); run<256>(250, 48, G, 20); run<128>(100, 24, G, 20);
  return 0;
}
