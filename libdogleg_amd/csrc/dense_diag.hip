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

} // namespace

void dense_launch_potrf_diag(hipStream_t st, double* A, int lda, int kb, int nb, int* info_dev, double* Linv)
{
  hipLaunchKernelGGL(k_potrf_diag_inv, dim3(1), dim3(TPB), 0, st, A, lda, kb, nb, info_dev, Linv);
}
