// panel_factor.h -- in-workgroup Cholesky of a dense column-major panel
// (nrows x w, nrows >= w): the w x w top block is factored, the rows below are
// solved against it.  Shared by the sparse supernode kernel and the dense
// blocked potrf (diagonal blocks).
//
// Thread-per-row, left-looking over column blocks of 8:
//   (1) every thread brings the 8 block-column entries of its row(s) up to date
//       against all previous columns: per previous column one own read and the
//       8 entries of the block rows as 4 broadcast 16-byte reads -> 8 FMAs;
//   (2) barrier; every thread factors the 8x8 diagonal block redundantly in
//       registers (no broadcast step);
//   (3) barrier; forward substitution of the thread's row against the 8x8 factor.
// 3 barriers per 8 columns, ~0.6 LDS reads per FMA.
// ALIGNED16: P is 16-byte aligned and ldp is even (LDS panels) -> double2 reads.
// On a non-positive pivot the column index (col0 + j) is min-reduced into *info
// and the pivot is replaced by 1 so that the sweep finishes without NaN storms.
#pragma once
#include <hip/hip_runtime.h>

// 1/sqrt(d) for d > 0: hardware v_rsq_f64 seed (about 1e-8 relative) + two Newton
// steps (relative error ~1e-16).  The pivot sqrt(d) = d * rsqrt(d) and its reciprocal
// come out of one short dependency chain instead of an IEEE sqrt followed by an IEEE
// divide.
__device__ __forceinline__ double dlg_rsqrt(double d)
{
  double y = __builtin_amdgcn_rsq(d);
  y = y*(1.5 - 0.5*d*y*y);
  y = y*(1.5 - 0.5*d*y*y);
  return y;
}

template <int NT, bool ALIGNED16>
__device__ __forceinline__ void panel_factor(double* P, int ldp, int nrows, int w, int tid,
                                             int* __restrict__ info, int col0)
{
  for(int kb = 0; kb < w; kb += 8)
  {
    const int nb = (w - kb < 8) ? w - kb : 8;
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
    double D[8][8];
#pragma unroll
    for(int c = 0; c < 8; c++)
#pragma unroll
      for(int q = 0; q <= c; q++)
        D[c][q] = (c < nb) ? P[(kb + c) + (kb + q)*ldp] : ((c == q) ? 1.0 : 0.0);
    bool bad = false; int badcol = 0;
    double Dinv[8];
#pragma unroll
    for(int c = 0; c < 8; c++)
    {
      double d = D[c][c];
#pragma unroll
      for(int q = 0; q < c; q++) d -= D[c][q]*D[c][q];
      if(!(d > 0.0)) { if(!bad) { bad = true; badcol = c; } d = 1.0; }
      const double inv = dlg_rsqrt(d);
      D[c][c] = d*inv;
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
  }
}

// Panel whose w x w top block is block diagonal (a supernode made of independent
// sibling leaves, e.g. the points seen by one set of cameras): member m owns columns
// [mcol[m], mcol[m+1]) (at most 8).  The members do not couple, so there is no sweep:
//   A: thread m factors the diagonal block of member m in registers (<= 8x8);
//   B: after one barrier every below row solves against each member's block.
template <int NT>
__device__ __forceinline__ void panel_factor_blockdiag(double* P, int ldp, int nrows, int w, int tid,
                                                       const int* __restrict__ mcol, int nmem,
                                                       int* __restrict__ info, int col0)
{
  for(int m = tid; m < nmem; m += NT)
  {
    const int c0 = mcol[m], nb = ((m + 1 < nmem) ? mcol[m+1] : w) - c0;
    double D[8][8];
#pragma unroll
    for(int c = 0; c < 8; c++)
#pragma unroll
      for(int q = 0; q <= c; q++)
        D[c][q] = (c < nb) ? P[(c0 + c) + (c0 + q)*ldp] : ((c == q) ? 1.0 : 0.0);
    bool bad = false; int badcol = 0;
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
#pragma unroll
      for(int i = c + 1; i < 8; i++)
      {
        double v = D[i][c];
#pragma unroll
        for(int q = 0; q < c; q++) v -= D[i][q]*D[c][q];
        D[i][c] = v*inv;
      }
    }
    if(bad) atomicMin(info, col0 + c0 + badcol);
#pragma unroll
    for(int c = 0; c < 8; c++)
#pragma unroll
      for(int q = 0; q <= c; q++) if(c < nb) P[(c0 + c) + (c0 + q)*ldp] = D[c][q];
  }
  __syncthreads();
  for(int r = w + tid; r < nrows; r += NT)
  {
    for(int m = 0; m < nmem; m++)
    {
      const int c0 = mcol[m], nb = ((m + 1 < nmem) ? mcol[m+1] : w) - c0;
      double x[8];
#pragma unroll
      for(int c = 0; c < 8; c++) x[c] = (c < nb) ? P[r + (c0 + c)*ldp] : 0.0;
#pragma unroll
      for(int c = 0; c < 8; c++)
      {
        if(c < nb)
        {
          double v = x[c];
#pragma unroll
          for(int q = 0; q < c; q++) v -= x[q]*P[(c0 + c) + (c0 + q)*ldp];
          x[c] = v/P[(c0 + c) + (c0 + c)*ldp];
        }
      }
#pragma unroll
      for(int c = 0; c < 8; c++) if(c < nb) P[r + (c0 + c)*ldp] = x[c];
    }
  }
  __syncthreads();
}
