// micro-benchmark of panel_factor (tools only): times the in-LDS panel Cholesky for a
// given (nrows, w) with G workgroups, and reports per-phase cycle shares via s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
// phase stamps: thread 0 accumulates clock deltas in LDS, publishes them at the end
__device__ long long* g_pf_out = nullptr;
__shared__ long long s_pf[16];
#define DLG_PF_DECL if(threadIdx.x == 0) { for(int _i = 0; _i < 15; _i++) s_pf[_i] = 0; s_pf[15] = clock64(); }
#define DLG_PF_STAMP(i) do { if(threadIdx.x == 0) { const long long _n = clock64(); s_pf[i] += _n - s_pf[15]; s_pf[15] = _n; } } while(0)
#define DLG_PF_PIN(x) asm volatile("" :: "v"(x))
#define DLG_PF_DONE if(threadIdx.x == 0 && g_pf_out) { for(int _i = 0; _i < 12; _i++) g_pf_out[_i] = s_pf[_i]; }
__device__ long long g_evt[64*8];
#define DLG_PF_EVT(J, e) do { if((threadIdx.x & 63) == 0 && blockIdx.x == 0) g_evt[(J)*8 + (e)] = clock64(); } while(0)
#ifdef BP_WLOG
// every wave's own log of the block-16 sweep: (block << 16 | tile << 8 | event, clock), workgroup 0 only
__device__ long long g_wlog[8*512]; __device__ int g_wlog_n[8];
#define DLG_PF_WLOG(J, t, e) do { if((threadIdx.x & 63) == 0 && blockIdx.x == 0) { const int _w = threadIdx.x >> 6; const int _n = g_wlog_n[_w]; if(_n < 255) { g_wlog[_w*512 + 2*_n] = ((long long)(J) << 16) | ((long long)(t) << 8) | (e); g_wlog[_w*512 + 2*_n + 1] = clock64(); g_wlog_n[_w] = _n + 1; } } } while(0)
#endif
#ifndef BP_LDP_EXTRA
#define BP_LDP_EXTRA 0      // -DBP_LDP_EXTRA=16: the leading dimension of the LDS panel padded (bank layout experiments)
#endif
#include "../../libdogleg_amd/csrc/panel_factor.h"

__device__ int g_mcol[256];
template <int NT, int MODE>
__global__ void __launch_bounds__(NT) k_panel(double* G, int nrows, int w, int* info, long long* stamps)
{
  extern __shared__ __attribute__((aligned(16))) double P[];
  double* g = G + (size_t)blockIdx.x*nrows*w;
  const int tid = threadIdx.x;
  const int ldp = ((nrows + 1) & ~1) + BP_LDP_EXTRA;
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
  long long t1 = clock64(); long long w1 = wall_clock64();
  if(MODE == 0) panel_factor<NT, true>(P, ldp, nrows, w, tid, info, 0);
  if(MODE == 1) { if(blockIdx.x == 0 && tid == 0) g_pf_out = stamps + 4; panel_factor<NT, true>(P, ldp, nrows, w, tid, info, 0); }
  if(MODE == 4) { __shared__ int s_mcol[260]; __shared__ double s_rdiag[256]; panel_factor_blockdiag<NT>(P, ldp, nrows, w, tid, g_mcol, w/3, info, 0, s_mcol, s_rdiag); }
  if(MODE == 5) { if(blockIdx.x == 0 && tid == 0) g_pf_out = stamps + 4; panel_factor_b16<(NT >= 256 ? NT : 256)>(P, ldp, nrows, w, tid, info, 0); }
  if(MODE == 2) { if(NT >= 256) panel_factor_mfma<(NT >= 256 ? NT : 256)>(P, ldp, nrows, w, tid, info, 0); else if(NT >= 256) panel_factor_mfma<(NT >= 256 ? NT : 256)>(P, ldp, nrows, w, tid, info, 0); else panel_factor<NT, true, true>(P, ldp, nrows, w, tid, info, 0); }
  if(MODE == 3) { if(blockIdx.x == 0 && tid == 0) g_pf_out = stamps + 4; if(NT >= 256) panel_factor_mfma<(NT >= 256 ? NT : 256)>(P, ldp, nrows, w, tid, info, 0); else panel_factor<NT, true, true>(P, ldp, nrows, w, tid, info, 0); }
  __syncthreads();
  long long t2 = clock64(); long long w2 = wall_clock64();
  for(int e = tid; e < nrows*w; e += NT) { int j = e / nrows; g[e] = P[e + j*(ldp - nrows)]; }
  long long t3 = clock64();
  if(tid == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = t2 - t1; stamps[2] = t3 - t2; stamps[3] = w2 - w1; }
}

template <int NT>
void run(int nrows, int w, int G, int iters)
{
  const size_t n = (size_t)nrows*w;
  std::vector<double> h(n*G);
  for(int b = 0; b < G; b++)
    for(int j = 0; j < w; j++)
      for(int i = 0; i < nrows; i++)
        h[b*n + i + (size_t)j*nrows] = (i == j) ? (double)(w + 1) : ((i < w && i < j) ? (getenv("DLG_PF_UPPER_NAN") ? NAN : 0.0) : 0.3*sin(0.37*i + 1.3*j));
  double* d; int* info; long long* st;
  hipMalloc(&d, n*G*8); hipMalloc(&info, 4); hipMalloc(&st, 256);
  const int lds = (int)((((nrows + 1) & ~1) + BP_LDP_EXTRA)*w*8);
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
  long long ph[11]; hipMemcpy(ph, st, 80, hipMemcpyDeviceToHost);
  printf("      scalar: sweep %lld +wait %lld | factor %lld +wait %lld | solve %lld +wait %lld\n", ph[4], ph[5], ph[6], ph[7], ph[8], ph[9]);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_panel<NT, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  best = 1e9;
  for(int rep = 0; rep < 3; rep++)
  {
    hipMemcpy(d, h.data(), n*G*8, hipMemcpyHostToDevice);
    hipEventRecord(e0);
    for(int it = 0; it < iters; it++) { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_panel<NT, 2>), dim3(G), dim3(NT), lds, 0, d, nrows, w, info, st); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if(ms < best) best = ms;
  }
  hipMemcpy(d, h.data(), n*G*8, hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_panel<NT, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_panel<NT, 3>), dim3(G), dim3(NT), lds, 0, d, nrows, w, info, st);
  hipDeviceSynchronize();
  hipMemcpy(ph, st, 80, hipMemcpyDeviceToHost);
  hipMemcpy(ph, st, 88, hipMemcpyDeviceToHost);
  printf("      MFMA %.1f us: wave0 tile %lld load %lld factor %lld write-back %lld | barrier %lld | solve %lld | barrier %lld\n", best*1e3/iters, ph[6], ph[7], ph[10], ph[4], ph[5], ph[8], ph[9]);
  {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_panel<NT, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    float bb = 1e9;
    for(int rep = 0; rep < 3; rep++)
    {
      hipMemcpy(d, h.data(), n*G*8, hipMemcpyHostToDevice);
      hipEventRecord(e0);
      for(int it = 0; it < iters; it++) { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_panel<NT, 5>), dim3(G), dim3(NT), lds, 0, d, nrows, w, info, st); }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if(ms < bb) bb = ms;
    }
    std::vector<double> r0(n), r5(n);
    hipMemcpy(d, h.data(), n*8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_panel<NT, 0>), dim3(1), dim3(NT), lds, 0, d, nrows, w, info, st);
    hipMemcpy(r0.data(), d, n*8, hipMemcpyDeviceToHost);
    hipMemcpy(d, h.data(), n*8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_panel<NT, 5>), dim3(1), dim3(NT), lds, 0, d, nrows, w, info, st);
    hipMemcpy(r5.data(), d, n*8, hipMemcpyDeviceToHost);
    long long ph2[16]; hipMemcpy(ph2, st, 128, hipMemcpyDeviceToHost);
    printf("      vfin phases: pass %lld | barrier %lld | partials+pivot sums %lld | factor %lld | solve+store %lld | barrier %lld\n", ph2[4+6], ph2[4+7], ph2[4+8], ph2[4+9], ph2[4+10], ph2[4+11]);
    hipMemcpy(ph, st, 88, hipMemcpyDeviceToHost);
    double worst = 0, big = 0, up = 0; int wi = 0, wj = 0;
    for(int j = 0; j < w; j++) for(int i = 0; i < nrows; i++)
    {
      const double dd = fabs(r0[i + (size_t)j*nrows] - r5[i + (size_t)j*nrows]);
      if(i >= j) { if(dd > worst || dd != dd) { worst = dd; wi = i; wj = j; } big = fmax(big, fabs(r0[i + (size_t)j*nrows])); }
      else up = fmax(up, fabs(h[i + (size_t)j*nrows] - r5[i + (size_t)j*nrows]));
    }
#ifdef BP_WLOG
    {
      // one more launch of workgroup 0 alone with the logs cleared, then every wave's events in time order
      int zero[8] = {0,0,0,0,0,0,0,0}; hipMemcpyToSymbol(HIP_SYMBOL(g_wlog_n), zero, sizeof(zero));
      hipMemcpy(d, h.data(), n*8, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_panel<NT, 5>), dim3(1), dim3(NT), lds, 0, d, nrows, w, info, st);
      hipDeviceSynchronize();
      static long long lg[8*512]; int ln[8];
      hipMemcpyFromSymbol(lg, HIP_SYMBOL(g_wlog), sizeof(lg)); hipMemcpyFromSymbol(ln, HIP_SYMBOL(g_wlog_n), sizeof(ln));
      long long t0 = 0; for(int wv = 0; wv < 8; wv++) for(int k = 0; k < ln[wv]; k++) if(t0 == 0 || lg[wv*512 + 2*k + 1] < t0) t0 = lg[wv*512 + 2*k + 1];
      const char* en[] = {"tile start", "updated", "inverse seen", "stored", "handed to wave 0", "inverse out", "next tile done"};
      for(int wv = 0; wv < 8; wv++)
        for(int k = 0; k < ln[wv]; k++)
        {
          const long long c = lg[wv*512 + 2*k]; const int J = (int)(c >> 16), t = (int)((c >> 8) & 255), e = (int)(c & 255);
          printf("        wlog wave %d block %d tile %2d %-16s %7lld\n", wv, J, t, en[e], lg[wv*512 + 2*k + 1] - t0);
        }
    }
#endif
    if(getenv("DLG_PF_EVENTS"))
    {
      long long ev[64*8]; hipMemcpyFromSymbol(ev, HIP_SYMBOL(g_evt), sizeof(ev));
      const int nblk = (w + 15)/16;
      for(int J = 0; J + 1 < nblk; J++)
        printf("        block %d: steps %lld .. %lld | next tile's wave: at the tile %+lld, diagonal rows seen %+lld, handed over %+lld | wave 0: hand-over seen %+lld, tile ready %+lld (all relative to the inverse being out)\n", J,
               ev[J*8] - ev[0], ev[J*8+1] - ev[0], ev[J*8+4] - ev[J*8+1], ev[J*8+5] - ev[J*8+1], ev[J*8+6] - ev[J*8+1], ev[J*8+2] - ev[J*8+1], ev[J*8+3] - ev[J*8+1]);
    }
    printf("      B16 tail: to the end of the sweep %lld clocks\n", ph[9]);
    { long long hs5[4]; hipMemcpy(hs5, st, 32, hipMemcpyDeviceToHost); printf("      B16 factor phase: %lld shader clocks in %lld x 10 ns of wall clock = %.2f GHz\n", hs5[1], hs5[3], hs5[3] > 0 ? (double)hs5[1]/(10.0*hs5[3]) : 0.0); }
    printf("      B16 %.1f us: wave 0: wait %lld steps %lld next tile %lld (of which waiting %lld) | scalar vs b16: max |diff| %.3g at (%d,%d) (max |L| %.3g), strict upper triangle touched by %.3g\n", bb*1e3/iters, ph[4], ph[5], ph[6], ph[7], worst, wi, wj, big, up);
  }
  {
    // the two sweeps on the same panel: largest difference over the lower trapezoid
    std::vector<double> r0(n), r2(n);
    hipMemcpy(d, h.data(), n*8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_panel<NT, 0>), dim3(1), dim3(NT), lds, 0, d, nrows, w, info, st);
    hipMemcpy(r0.data(), d, n*8, hipMemcpyDeviceToHost);
    hipMemcpy(d, h.data(), n*8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_panel<NT, 2>), dim3(1), dim3(NT), lds, 0, d, nrows, w, info, st);
    hipMemcpy(r2.data(), d, n*8, hipMemcpyDeviceToHost);
    double worst = 0, big = 0;
    for(int j = 0; j < w; j++) for(int i = j; i < nrows; i++) { worst = fmax(worst, fabs(r0[i + (size_t)j*nrows] - r2[i + (size_t)j*nrows])); big = fmax(big, fabs(r0[i + (size_t)j*nrows])); }
    printf("      scalar vs MFMA sweep: max |diff| %.3g (max |L| %.3g)\n", worst, big);
  }
  hipFree(d); hipFree(info); hipFree(st);
}

template <int NT>
void run_bd(int nrows, int w, int G, int iters)
{
  const size_t n = (size_t)nrows*w;
  std::vector<double> h(n*G);
  for(int b = 0; b < G; b++)
    for(int j = 0; j < w; j++)
      for(int i = 0; i < nrows; i++)
        h[b*n + i + (size_t)j*nrows] = (i == j) ? (double)(w + 1) : ((i < w) ? ((i/3 == j/3 && i > j) ? 0.2 : 0.0) : 0.3*sin(0.37*i + 1.3*j));
  std::vector<int> mc(256); for(int m = 0; m < 256; m++) mc[m] = 3*m;
  hipMemcpyToSymbol(HIP_SYMBOL(g_mcol), mc.data(), 256*4);
  double* d; int* info; long long* st;
  hipMalloc(&d, n*G*8); hipMalloc(&info, 4); hipMalloc(&st, 256);
  const int lds = (int)(((nrows + 1) & ~1)*w*8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_panel<NT, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for(int rep = 0; rep < 3; rep++)
  {
    hipMemcpy(d, h.data(), n*G*8, hipMemcpyHostToDevice);
    hipEventRecord(e0);
    for(int it = 0; it < iters; it++) { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_panel<NT, 4>), dim3(G), dim3(NT), lds, 0, d, nrows, w, info, st); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if(ms < best) best = ms;
  }
  long long hs[3]; hipMemcpy(hs, st, 24, hipMemcpyDeviceToHost);
  printf("blockdiag NT=%d nrows=%d w=%d G=%d: %.1f us/launch   cycles load %lld factor %lld store %lld\n", NT, nrows, w, G,
         best*1e3/iters, hs[0], hs[1], hs[2]);
  hipFree(d); hipFree(info); hipFree(st);
}

int main(int argc, char** argv)
{
  const int G = argc > 1 ? atoi(argv[1]) : 64;
  if(argc > 4) { run<512>(atoi(argv[2]), atoi(argv[3]), G, 20); return 0; }      // G nrows w 512
  if(argc > 3) { run<256>(atoi(argv[2]), atoi(argv[3]), G, 20); return 0; }
  run_bd<256>(124, 51, G, 20); run_bd<128>(124, 51, G, 20); run_bd<256>(124, 51, 2489, 20); run_bd<128>(124, 51, 2489, 20);
  run<512>(170, 96, G, 20); run<256>(170, 96, G, 20); run<128>(170, 96, G, 20);
  run<256>(128, 64, G, 20); run<256>(250, 48, G, 20); run<128>(100, 24, G, 20);
  return 0;
}
