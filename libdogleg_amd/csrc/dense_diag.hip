// dense_diag.hip -- diagonal-block kernel of the dense blocked Cholesky (K5-dense, see
// kernels_dense.hip).  A file of its own because the matrix-core panel factorisation wants the
// VGPR form of the fp64 MFMA (build.py), which the SYRK kernels with their many accumulators do not.
#include "dlg_internal.h"
#include "panel_factor.h"

namespace {
constexpr int TPB = 256;
constexpr int NB = 64;

__global__ void __launch_bounds__(TPB) k_potrf_diag_inv(double* __restrict__ A, int lda, int kb,
                                                        int nb, int* __restrict__ info,
                                                        double* __restrict__ Linv)
{
  // rows 0..63: the diagonal block; rows 64..127: the identity.  Factoring the
  // 128 x 64 panel leaves L in the top block and L^-T in the bottom block (the
  // row solve X L^T = I), i.e. the inverse comes out of the same sweep.
  __shared__ __attribute__((aligned(16))) double P[2*NB*NB];
  __shared__ int sbad;
  const int t = threadIdx.x;
  constexpr int LD = 2*NB;
  for(int e = t; e < NB*NB; e += TPB)
  {
    const int i = e % NB, j = e / NB;
    double v = (i == j) ? 1.0 : 0.0;                          // identity padding of a short last block
    if(i < nb && j < nb) v = (i >= j) ? A[(size_t)(kb + j)*lda + kb + i] : 0.0;
    P[i + j*LD] = v;
    P[NB + i + j*LD] = (i == j) ? 1.0 : 0.0;
  }
  if(t == 0) sbad = 0x7fffffff;
  __syncthreads();
  panel_factor_mfma<TPB>(P, LD, 2*NB, NB, t, &sbad, 0);
  if(t == 0) { const int bad = sbad; if(bad < nb && *info == 0) *info = kb + bad + 1; }
  for(int e = t; e < NB*NB; e += TPB)
  {
    const int i = e % NB, j = e / NB;
    if(i < nb && j < nb && i >= j) A[(size_t)(kb + j)*lda + kb + i] = P[i + j*LD];
    Linv[e] = (i >= j) ? P[NB + j + i*LD] : 0.0;              // Linv(i,j) = (L^-T)(j,i)
  }
}

// The diagonal block and the rows below it in ONE launch: workgroup 0 is k_potrf_diag_inv (it hands the
// inverse of the block over through write-through stores + a flag carrying the epoch of the launch),
// workgroups 1.. are k_trsm_gemm for 64 rows each: they stage their rows of A while the diagonal block
// is being factored and read the inverse around L1 once the flag is up.  One kernel boundary and the
// load of the rows less per 64 columns (32 of them for N = 2000).  One workgroup per CU (the launch
// asks for more than half of the LDS): the form the hand-off is measured in (MI355X_MICROARCH.md).
__global__ void __launch_bounds__(TPB) k_potrf_diag_trsm(double* __restrict__ A, int lda, int kb, int nb, int n,
                                                         int* __restrict__ info, double* Linv,
                                                         int* flag, int epoch)
{
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ int sbad;
  const int t = threadIdx.x;
  typedef __attribute__((address_space(1))) double* gd_t;
  typedef const __attribute__((address_space(1))) double* gcd_t;
  if(blockIdx.x == 0)
  {
    double* P = sm;                       // [2*NB][NB], leading dimension 2*NB
    constexpr int LD = 2*NB;
    for(int e = t; e < NB*NB; e += TPB)
    {
      const int i = e % NB, j = e / NB;
      double v = (i == j) ? 1.0 : 0.0;
      if(i < nb && j < nb) v = (i >= j) ? A[(size_t)(kb + j)*lda + kb + i] : 0.0;
      P[i + j*LD] = v;
      P[NB + i + j*LD] = (i == j) ? 1.0 : 0.0;
    }
    if(t == 0) sbad = 0x7fffffff;
    __syncthreads();
    panel_factor_mfma<TPB>(P, LD, 2*NB, NB, t, &sbad, 0);
    if(t == 0) { const int bad = sbad; if(bad < nb && *info == 0) *info = kb + bad + 1; }
    for(int e = t; e < NB*NB; e += TPB)
    {
      const int i = e % NB, j = e / NB;
      __hip_atomic_store((gd_t)(Linv + e), (i >= j) ? P[NB + j + i*LD] : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if(t == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for(int e = t; e < NB*NB; e += TPB)
    {
      const int i = e % NB, j = e / NB;
      if(i < nb && j < nb && i >= j) A[(size_t)(kb + j)*lda + kb + i] = P[i + j*LD];
    }
    return;
  }
  double (*As)[NB + 1] = reinterpret_cast<double (*)[NB + 1]>(sm);                    // As[r][k]
  double (*Ls)[NB + 1] = reinterpret_cast<double (*)[NB + 1]>(sm + NB*(NB + 1));      // Ls[c][k] = Linv[c][k]
  const int r0 = kb + nb + ((int)blockIdx.x - 1)*NB;
  for(int e = t; e < NB*NB; e += TPB)
  {
    const int i = e % NB, k = e / NB;
    const int r = r0 + i;
    As[i][k] = (r < n && k < nb) ? A[(size_t)(kb + k)*lda + r] : 0.0;
  }
  if(t == 0)
  {
    int spins = 0;
    while(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch)
    { __builtin_amdgcn_s_sleep(1); if(++spins > (1 << 21)) break; }
  }
  __syncthreads();
  for(int e = t; e < NB*NB; e += TPB)
  {
    const int i = e % NB, k = e / NB;
    Ls[i][k] = __hip_atomic_load((gcd_t)(Linv + e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const int rl = t & 63, cg = t >> 6;
  const int r = r0 + rl;
  double out[16];
#pragma unroll
  for(int cc = 0; cc < 16; cc++)
  {
    const int c = cg*16 + cc;
    double sacc = 0.0;
    for(int k = 0; k <= c; k++) sacc += As[rl][k]*Ls[c][k];
    out[cc] = sacc;
  }
  if(r < n)
  {
#pragma unroll
    for(int cc = 0; cc < 16; cc++)
    {
      const int c = cg*16 + cc;
      if(c < nb) A[(size_t)(kb + c)*lda + r] = out[cc];
    }
  }
}

} // namespace

void dense_launch_potrf_diag_trsm(hipStream_t st, double* A, int lda, int kb, int nb, int n, int* info_dev, double* Linv,
                                  int* flag, int epoch)
{
  static bool attr = false;
  constexpr int LDSB = 88*1024;           // > half of the CU's LDS: one workgroup per CU
  if(!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_potrf_diag_trsm), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB); attr = true; }
  const int ntr = (n - kb - nb + NB - 1)/NB;
  hipLaunchKernelGGL(k_potrf_diag_trsm, dim3(1 + ntr), dim3(TPB), LDSB, st, A, lda, kb, nb, n, info_dev, Linv, flag, epoch);
}

void dense_launch_potrf_diag(hipStream_t st, double* A, int lda, int kb, int nb, int* info_dev, double* Linv)
{
  hipLaunchKernelGGL(k_potrf_diag_inv, dim3(1), dim3(TPB), 0, st, A, lda, kb, nb, info_dev, Linv);
}
