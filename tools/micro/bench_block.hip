// the 8x8 diagonal-block step of the panel sweep alone: LDS load, register Cholesky, write back (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "../../libdogleg_amd/csrc/panel_factor.h"

template <int MODE>
__global__ void __launch_bounds__(512) k(double* G, long long* cyc, int reps)
{
  __shared__ __attribute__((aligned(16))) double P[64*8];
  __shared__ __attribute__((aligned(16))) double s_dinv[8];
  const int tid = threadIdx.x, ldp = 64;
  for(int e = tid; e < 64*8; e += blockDim.x) P[e] = G[e];
  __syncthreads();
  if(tid >= 64) return;
  const long long t0 = clock64();
  double D[8][8], Dinv[8];
  double keep = 0.0;
  for(int r = 0; r < reps; r++)
  {
    if(MODE != 2 || r == 0) pf_load_block(D, P, ldp, 0, 8);
    if(MODE == 2)
    {
      // registers only: restore a positive-definite block from the last factor (cheap, dependent on it)
#pragma unroll
      for(int c = 0; c < 8; c++) { D[c][c] = D[c][c]*D[c][c] + 8.0;
#pragma unroll
        for(int q = 0; q < c; q++) D[c][q] = D[c][q]*0.5; }
    }
    pf_factor_block(D, Dinv);
    if(MODE == 0)
    {
      if(tid == 0)
      {
#pragma unroll
        for(int q = 0; q < 8; q++)
#pragma unroll
          for(int c2 = (q & ~1); c2 < 8; c2 += 2)
            *reinterpret_cast<double2*>(&P[c2 + q*ldp]) = make_double2(D[c2][q]*D[c2][q] + 8.0*(c2 == q), D[c2 + 1][q]*D[c2 + 1][q] + 8.0*(c2 + 1 == q));
#pragma unroll
        for(int q = 0; q < 8; q += 2) *reinterpret_cast<double2*>(&s_dinv[q]) = make_double2(Dinv[q], Dinv[q + 1]);
      }
    }
    else if(MODE == 1)
    {
      const int c = tid >> 3, q = tid & 7;
      double v = 0.0, dv = 0.0;
#pragma unroll
      for(int cc = 0; cc < 8; cc++)
      {
#pragma unroll
        for(int qq = 0; qq <= cc; qq++) v = (cc == c && qq == q) ? D[cc][qq] : v;
        dv = (cc == q) ? Dinv[cc] : dv;
      }
      if(q <= c) P[c + q*ldp] = v*v + 8.0*(c == q);
      if(c == 0) s_dinv[q] = dv;
    }
    keep += Dinv[7];
  }
  const long long t1 = clock64();
  if(tid == 0) { cyc[0] = t1 - t0; G[0] = keep + D[7][3]; }
}
template <int MODE> void run(const char* name, int nt)
{
  std::vector<double> h(512);
  for(int j = 0; j < 8; j++) for(int i = 0; i < 64; i++) h[i + 64*j] = (i == j) ? 9.0 : 0.3*sin(0.37*i + 1.3*j);
  double* d; long long* c; hipMalloc(&d, 4096); hipMalloc(&c, 8);
  long long t = 0; const int reps = 200;
  for(int r = 0; r < 2; r++) { hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice); hipLaunchKernelGGL(HIP_KERNEL_NAME(k<MODE>), dim3(1), dim3(nt), 0, 0, d, c, reps); hipMemcpy(&t, c, 8, hipMemcpyDeviceToHost); }
  printf("%-52s %.0f ticks per block\n", name, (double)t/reps);
  hipFree(d); hipFree(c);
}
int main()
{
  run<0>("load + factor + lane-0 pair stores", 64);
  run<1>("load + factor + select write-back", 64);
  run<2>("factor only (registers)", 64);
  return 0;
}
