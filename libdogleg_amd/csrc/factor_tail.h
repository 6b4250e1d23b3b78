// factor_tail.h -- the update matrix U = B B' of a supernode panel held in LDS, on the matrix cores
// (the level kernels of sparse_factor.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace {
typedef double dlg_v4d __attribute__((ext_vector_type(4)));
typedef double dlg_v2d __attribute__((ext_vector_type(2)));
// ------------------------------------------------------------------ K5 ------
// Update matrices (U = B B' of a supernode, B = the mb rows below its diagonal block, plus in
// the multifrontal region what its children left over) are stored as packed lower triangles,
// column-major: element (i, j), i >= j, at  j*mb - j*(j-1)/2 + (i - j).
__device__ __forceinline__ int tri_col(int j, int mb) { return j*mb - j*(j - 1)/2 - j; }   // (i, j) at tri_col + i

// NCH consecutive lower 16x16 tiles (column-major tile order, first tile `first`) of
// U = B B' (B = the mb rows below the diagonal block of the LDS panel Pb, w columns), one
// accumulator chain per tile, every chain with its own operands (the tiles may span two tile
// columns).  Rows / columns past the end are clamped (their results are never written);
// KD k-steps with their own operand registers, so the loads of the next ones are in flight
// during the products of one.  mode 2: the multifrontal region keeps W = (children) - U;
template <int NCH, int KD = 4>
__device__ __forceinline__ void factor_tail_tiles(const double* Pb, int ldp, int w, int mb, int T, int first,
                                                  double* Ud, int mode, bool w_hbm, bool mf_acc, int lane,
                                                  int64_t acc_shift, bool st_wt, int usp, double* Pgap, int ush)
{
  const int jn = lane & 15, kq = lane >> 4;
  int ti[NCH], tjq[NCH], oa[NCH], ob[NCH];
  dlg_v4d c4[NCH];
  {
    int rem = first, tj = 0;
    while(rem >= T - tj) { rem -= T - tj; tj++; }
    int tcur = tj + rem;
#pragma unroll
    for(int q = 0; q < NCH; q++)
    {
      ti[q] = tcur; tjq[q] = tj;
      tcur++; if(tcur >= T) { tj++; tcur = tj; }
      oa[q] = w + min(16*ti[q] + jn, mb - 1) + kq*ldp;
      ob[q] = w + min(16*tjq[q] + jn, mb - 1) + kq*ldp;
      c4[q] = (dlg_v4d){0.0, 0.0, 0.0, 0.0};
    }
  }
  // an update matrix kept in HBM: the children's sums of these tiles, on their way during the
  // products (read around L1: they were formed by atomics in L2)
  double w0[NCH][4];
  if(w_hbm)
  {
#pragma unroll
    for(int q = 0; q < NCH; q++)
    {
      const int j = 16*tjq[q] + jn, jt0 = tri_col(j, mb);
#pragma unroll
      for(int r = 0; r < 4; r++)
      {
        const int i = 16*ti[q] + kq + 4*r;
        w0[q][r] = (i < mb && j <= i) ? __hip_atomic_load(Ud + acc_shift + jt0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
      }
    }
  }
  const int w4 = w & ~3;
  const int st = 4*ldp;
  // Whole k-steps, four of them in flight: a wave has one to six accumulator chains and a tile's k-loop is 15 - 17 steps --
  // with the operands of only the next step on their way (rounds 1 - 4) every step waited out most of an LDS round trip
  // (~200 clocks at this occupancy against ~80 for the product): 3 - 4 us of a level of the one-launch region for what is
  // 1.3 us of products.  Same products in the same order: the same bits.
  // (KD = 2 for the lean leaf instantiation of k_factor_level: 128 registers, four workgroups a CU)
  const int nk = w4 >> 2;
  double a[KD][NCH], b[KD][NCH];
#pragma unroll
  for(int u = 0; u < KD; u++)
  {
    const int ku = min(u, max(nk - 1, 0))*st;
#pragma unroll
    for(int q = 0; q < NCH; q++) { a[u][q] = (nk > 0) ? Pb[oa[q] + ku] : 0.0; b[u][q] = (nk > 0) ? Pb[ob[q] + ku] : 0.0; }
  }
  int ks = 0;
  for(; ks + KD <= nk; ks += KD)
  {
#pragma unroll
    for(int u = 0; u < KD; u++)
    {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for(int q = 0; q < NCH; q++) c4[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][q], b[u][q], c4[q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // the step KD ahead takes this one's registers (clamped to the last whole one: a harmless re-read at the end)
      const int kn = min(ks + KD + u, nk - 1)*st;
#pragma unroll
      for(int q = 0; q < NCH; q++) { a[u][q] = Pb[oa[q] + kn]; b[u][q] = Pb[ob[q] + kn]; }
    }
  }
#pragma unroll
  for(int u = 0; u < KD; u++)
    if(ks + u < nk)
    {
#pragma unroll
      for(int q = 0; q < NCH; q++) c4[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][q], b[u][q], c4[q], 0, 0, 0);
    }
  const int kk = w4, ko = nk*st;
  if(kk < w)
  {
    // the last, partial k-step: columns past the end contribute zeros
    const bool kok = kk + kq < w;
    const int kz = ko - (kok ? 0 : (kk + kq - (w - 1))*ldp);
#pragma unroll
    for(int q = 0; q < NCH; q++)
    {
      const double az = kok ? Pb[oa[q] + kz] : 0.0, bz = kok ? Pb[ob[q] + kz] : 0.0;
      c4[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(az, bz, c4[q], 0, 0, 0);
    }
  }
#pragma unroll
  for(int q = 0; q < NCH; q++)
  {
    const int j = 16*tjq[q] + jn, jtri = tri_col(j, mb);
    // columns from usp on live in the strict upper triangle of the panel's top block (sym_w_split)
    // (ush: a slice of the update matrix behind the panel starts at packed index -ush)
    double* Uc = (j >= usp) ? Pgap + (mb - j)*ldp - j : Ud + (jtri + ush);
#pragma unroll
    for(int r = 0; r < 4; r++)
    {
      const int i = 16*ti[q] + kq + 4*r;
      if(i < mb && j <= i)
      {
        if(st_wt)
        {
          // persistent top region, update matrix not staged in LDS: the parent reads it in this launch --
          // write-through stores into a slot no plain store or atomic ever touches (the children's
          // sums were formed in the shadow slot at acc_shift)
          typedef __attribute__((address_space(1))) double* gwptr_t;
          __hip_atomic_store((gwptr_t)(Ud + jtri + i), (w_hbm ? w0[q][r] : 0.0) - c4[q][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        else if(mode == 2) Uc[i] = (w_hbm ? w0[q][r] : (mf_acc ? Uc[i] : 0.0)) - c4[q][r];     // the region keeps W = -U
        else Uc[i] = c4[q][r];
      }
    }
  }
}

} // namespace
