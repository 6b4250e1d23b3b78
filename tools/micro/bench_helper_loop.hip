// tools only: the helper wave's pass of panel_factor_b16 (per 8 columns: four LDS reads, the next round's in flight, two
// or four dependent-in-pairs MFMAs) alone on the CU, with a partner on its SIMD, with all helper waves running
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(512) k(long long* out, double* sink, unsigned mask, int with_e, int ldp, int kb, int prio, int useneg)
{
  extern __shared__ double P[];
  for(int i = threadIdx.x; i < ldp*64; i += 512) P[i] = 0.001*i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, mm = lane & 15, kq = lane >> 4;
  v4d acc = {0, 0, 0, 0}, E = {0, 0, 0, 0};
  long long t0 = 0, t1 = 0;
  if(mask >> wv & 1)
  {
    const double* ap = P + 64 + mm + kq*ldp;
    const double* bp = P + 16*wv + mm + kq*ldp;
    if(prio && wv == 1) __builtin_amdgcn_s_setprio(3);
    t0 = clock64();
    if(useneg)
    for(int rep = 0; rep < 16; rep++)
    {
      double a0 = ap[0], b0 = bp[0], a1 = ap[4*ldp], b1 = bp[4*ldp];
      for(int k0 = 0; k0 < kb; k0 += 8)
      {
        const int kn = (k0 + 8 < kb) ? k0 + 8 : k0;
        const double na0 = ap[kn*ldp], nb0 = bp[kn*ldp], na1 = ap[(kn + 4)*ldp], nb1 = bp[(kn + 4)*ldp];
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 1);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 1);
        if(with_e)
        {
          E = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, b0, E, 0, 0, 1);
          E = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, b1, E, 0, 0, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        a0 = na0; b0 = nb0; a1 = na1; b1 = nb1;
      }
    }
    else
    for(int rep = 0; rep < 16; rep++)
    {
      double a0 = ap[0], b0 = bp[0], a1 = ap[4*ldp], b1 = bp[4*ldp];
      for(int k0 = 0; k0 < kb; k0 += 8)
      {
        const int kn = (k0 + 8 < kb) ? k0 + 8 : k0;
        const double na0 = ap[kn*ldp], nb0 = bp[kn*ldp], na1 = ap[(kn + 4)*ldp], nb1 = bp[(kn + 4)*ldp];
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1, b1, acc, 0, 0, 0);
        if(with_e)
        {
          E = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, -b0, E, 0, 0, 0);
          E = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, -b1, E, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        a0 = na0; b0 = nb0; a1 = na1; b1 = nb1;
      }
    }
    t1 = clock64();
  }
  if(lane == 0) out[wv] = t1 - t0;
  sink[threadIdx.x] = acc[0] + 3.0*E[1];
}
int main()
{
  long long* o; double* s; hipMalloc(&o, 64); hipMalloc(&s, 8*512);
  const int ldp = 200, kb = 48;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, ldp*64*8);
  struct { unsigned mask; int e; const char* name; int prio, neg; } cases[] = {
    {0x22, 1, "waves 1 + 5, wave 1 at priority 3", 1, 0}, {0x02, 1, "wave 1 alone, neg modifier", 0, 1}, {0x22, 1, "waves 1 + 5, neg modifier", 0, 1}, {0x22, 1, "waves 1 + 5, neg + priority", 1, 1},
    {0x02, 0, "wave 1 alone, 2 MFMAs a round"}, {0x02, 1, "wave 1 alone, 4 MFMAs a round"},
    {0x22, 1, "waves 1 + 5 (one SIMD), 4 MFMAs"}, {0x06, 1, "waves 1 + 2 (two SIMDs), 4 MFMAs"},
    {0xEE, 1, "waves 1-3, 5-7, 4 MFMAs"}, {0xEE, 0, "waves 1-3, 5-7, 2 MFMAs"}, {0xFF, 1, "all eight, 4 MFMAs"} };
  for(auto& c : cases)
  {
    for(int r = 0; r < 2; r++) { hipLaunchKernelGGL(k, dim3(1), dim3(512), ldp*64*8, 0, o, s, c.mask, c.e, ldp, kb, c.prio, c.neg); hipDeviceSynchronize(); }
    long long h[8]; hipMemcpy(h, o, 64, hipMemcpyDeviceToHost);
    const int nm = 16*(kb/8)*(c.e ? 4 : 2);
    double sk[128]; hipMemcpy(sk, s, 8*128, hipMemcpyDeviceToHost);
    printf("%-36s wave 1: %.0f clocks a round of 8 columns, %.0f per MFMA   (check %.10g)\n", c.name, h[1]/(16.0*kb/8), (double)h[1]/nm, sk[64 + 5]);
  }
  return 0;
}
