// sparse_solve.hip -- K6: triangular solves with the supernodal factor on gfx950,
// replaces cholmod_solve(CHOLMOD_A) (dogleg.c:853).
#include "sparse_internal.h"

namespace {

// ------------------------------------------------------------------ K6 ------
// forward: per supernode  y_t = L_tt^-1 (P b - gathered updates);  u_t = L_below y_t.
// The diagonal block is staged in LDS (odd leading dimension); the column sweep
// keeps y_i in a register and needs one barrier per column.
__global__ void __launch_bounds__(TPB) k_solve_fwd_level(const int* __restrict__ lvl_sn,
                                                         const int* __restrict__ sn_c0,
                                                         const int* __restrict__ sn_rowptr,
                                                         const int64_t* __restrict__ sn_lx,
                                                         const int* __restrict__ sn_scr,
                                                         const int* __restrict__ rl_ptr,
                                                         const int* __restrict__ rl_pos,
                                                         const int* __restrict__ perm,
                                                         const double* __restrict__ Lx,
                                                         const double* __restrict__ rhs,
                                                         double* __restrict__ scr,
                                                         double* __restrict__ ywork)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ double y[256];
  __shared__ double red[4];
  const int s = lvl_sn[blockIdx.x];
  const int c0 = sn_c0[s], w = sn_c0[s+1] - c0;
  const int nrows = sn_rowptr[s+1] - sn_rowptr[s];
  const double* L = Lx + sn_lx[s];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int ldp = w | 1;
  // stage the diagonal block
  batched_copy<TPB, 8>(w*w, tid, [&](int e) { const int j = e / w; return L[(e - j*w) + (size_t)j*nrows]; },
                       [&](int e, double v) { const int j = e / w; lds[(e - j*w) + j*ldp] = v; });
  // gather: long lists (a dense last block is fed by every supernode) use the whole workgroup
  for(int j = 0; j < w; j++)
  {
    const int k = c0 + j;
    const int e0 = rl_ptr[k], e1 = rl_ptr[k+1];
    if(e1 - e0 >= 2048)
    {
      double sum = 0.0;
      int e = e0 + tid;
      for(; e + 7*TPB < e1; e += 8*TPB)         // 8 independent gathers in flight per thread
      {
        int pz[8]; double vz[8];
#pragma unroll
        for(int u = 0; u < 8; u++) pz[u] = rl_pos[e + u*TPB];
#pragma unroll
        for(int u = 0; u < 8; u++) vz[u] = scr[pz[u]];
#pragma unroll
        for(int u = 0; u < 8; u++) sum += vz[u];
      }
      for(; e < e1; e += TPB) sum += scr[rl_pos[e]];
      sum = wave_sum(sum);
      __syncthreads();
      if(lane == 0) red[wv] = sum;
      __syncthreads();
      if(tid == 0) y[j] = rhs[perm[k]] - ((red[0] + red[1]) + (red[2] + red[3]));
    }
    else if((j & 3) == wv)
    {
      double sum = 0.0;
      for(int e = e0 + lane; e < e1; e += 64) sum += scr[rl_pos[e]];
      sum = wave_sum(sum);
      if(lane == 0) y[j] = rhs[perm[k]] - sum;
    }
  }
  __syncthreads();
  double yi = (tid < w) ? y[tid] : 0.0;
  for(int j = 0; j < w; j++)
  {
    if(tid == j) y[j] = yi / lds[j + j*ldp];
    __syncthreads();
    if(tid > j && tid < w) yi -= lds[tid + j*ldp]*y[j];
  }
  __syncthreads();
  for(int j = tid; j < w; j += TPB) ywork[c0 + j] = y[j];
  const int r = nrows - w - 1;            // the augmented row is not part of the solve
  double* u = scr + sn_scr[s];
  for(int i = tid; i < r; i += TPB)
  {
    double sum = 0.0;
#pragma unroll 8
    for(int j = 0; j < w; j++) sum += L[w + i + (size_t)j*nrows]*y[j];
    u[i] = sum;
  }
}
// backward: x_t = L_tt^-T (y_t - L_below^T x[below rows]); out[perm] = x.
// x at the below rows is gathered into LDS once; 8 waves share the columns of the L_below^T
// mat-vec.  The triangular solve runs over blocks of 8 columns from the bottom, thread = row:
// the 8 owners of a block publish their right-hand sides, after ONE barrier every thread
// solves the 8x8 block itself (the diagonal blocks sit in LDS with reciprocal pivots) and
// applies the 8 new unknowns to its own row with values of L it fetched a block ahead.
// w/8 barriers instead of w, no staging of the w x w block.
template <int BWD_NT>
__global__ void __launch_bounds__(BWD_NT) k_solve_bwd_level(const int* __restrict__ lvl_sn,
                                                            const int* __restrict__ sn_c0,
                                                            const int* __restrict__ sn_rowptr,
                                                            const int* __restrict__ sn_rows,
                                                            const int64_t* __restrict__ sn_lx,
                                                            const int* __restrict__ perm,
                                                            const double* __restrict__ Lx,
                                                            double* __restrict__ ywork,
                                                            double* __restrict__ out, int use_aug,
                                                            const int* __restrict__ sn_bd_ptr,
                                                            const int* __restrict__ sn_bd_col)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int s = lvl_sn[blockIdx.x];
  const int c0 = sn_c0[s], w = sn_c0[s+1] - c0;
  const int nrows = sn_rowptr[s+1] - sn_rowptr[s];
  const int* rows = sn_rows + sn_rowptr[s];
  const double* L = Lx + sn_lx[s];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = nrows - w - 1;
  const int nblk = (w + 7) >> 3;
  double* xb = lds;                       // [r]   x at the below rows
  double* xs = lds + ((r + 1) & ~1);      // [256] right-hand side, then the solution
  double* T = xs + 256;                   // [nblk][8][8] diagonal blocks (lower), reciprocal pivots
  double* rhs = T + nblk*64;              // [2][8]
  // block-diagonal top (merged sibling leaves): the members do not couple, every member is a
  // little triangular system of its own -- no sweep over the columns at all
  const int nmem = sn_bd_ptr[s+1] - sn_bd_ptr[s];
  for(int i = tid; i < r; i += BWD_NT) xb[i] = ywork[rows[w + i]];
  for(int e = tid; e < (nmem > 0 ? 0 : nblk*64); e += BWD_NT)
  {
    const int j0 = (e >> 6)*8, a = (e >> 3) & 7, b = e & 7;
    const bool valid = a >= b && j0 + a < w;
    double v = valid ? L[(j0 + a) + (size_t)(j0 + b)*nrows] : 0.0;
    if(a == b) v = valid ? 1.0/v : 1.0;
    T[e] = v;
  }
  __syncthreads();
  // a wave takes 4 columns at a time: their loads are all in flight together
  for(int jg = 4*wv; jg < w; jg += 4*(BWD_NT/64))
  {
    const double* Lj = L + (size_t)jg*nrows + w;
    const int nc = min(4, w - jg);
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 2
    for(int i = lane; i < r; i += 64)
    {
      const double x = xb[i];
#pragma unroll
      for(int c = 0; c < 4; c++) acc[c] += ((c < nc) ? Lj[i + (size_t)c*nrows] : 0.0)*x;
    }
#pragma unroll
    for(int c = 0; c < 4; c++)
    {
      const double sum = wave_sum(acc[c]);
      if(lane == 0 && c < nc)
        xs[jg + c] = (use_aug ? L[(nrows - 1) + (size_t)(jg + c)*nrows] : ywork[c0 + jg + c]) - sum;
    }
  }
  __syncthreads();
  if(nmem > 0)
  {
    const int* mcol = sn_bd_col + sn_bd_ptr[s];
    for(int m = tid; m < nmem; m += BWD_NT)
    {
      const int m0 = mcol[m], nb = ((m + 1 < nmem) ? mcol[m+1] : w) - m0;
      double Lm[8][8], xk[8];
#pragma unroll
      for(int a = 0; a < 8; a++)
#pragma unroll
        for(int b = 0; b <= a; b++) Lm[a][b] = (a < nb) ? L[(m0 + a) + (size_t)(m0 + b)*nrows] : (a == b ? 1.0 : 0.0);
#pragma unroll
      for(int a = 7; a >= 0; a--)
      {
        double v = (a < nb) ? xs[m0 + a] : 0.0;
#pragma unroll
        for(int b = a + 1; b < 8; b++) v -= Lm[b][a]*xk[b];
        xk[a] = v/Lm[a][a];
      }
#pragma unroll
      for(int a = 0; a < 8; a++) if(a < nb) xs[m0 + a] = xk[a];
    }
    __syncthreads();
    for(int j = tid; j < w; j += BWD_NT) { ywork[c0 + j] = xs[j]; out[perm[c0 + j]] = xs[j]; }
    return;
  }
  double xi = (tid < w) ? xs[tid] : 0.0;
  const double* Lcol = L + (size_t)min(tid, w - 1)*nrows;      // column tid of L = row tid of L^T
  double lv[8];
  {
    const int j0 = 8*(nblk - 1);
#pragma unroll
    for(int a = 0; a < 8; a++) lv[a] = (tid < j0 && j0 + a < w) ? Lcol[j0 + a] : 0.0;
  }
  for(int blk = nblk - 1; blk >= 0; blk--)
  {
    const int j0 = 8*blk;
    double* rh = rhs + 8*(blk & 1);
    if(tid >= j0 && tid < j0 + 8) rh[tid - j0] = xi;
    double ln[8];
#pragma unroll
    for(int a = 0; a < 8; a++) ln[a] = (blk > 0 && tid < j0 - 8) ? Lcol[j0 - 8 + a] : 0.0;
    __syncthreads();
    const double* Tb = T + blk*64;
    double xk[8];
#pragma unroll
    for(int a = 7; a >= 0; a--)
    {
      double v = rh[a];
#pragma unroll
      for(int b = a + 1; b < 8; b++) v -= Tb[b*8 + a]*xk[b];
      xk[a] = v*Tb[a*8 + a];
    }
    if(tid < j0)
    {
#pragma unroll
      for(int a = 0; a < 8; a++) xi -= lv[a]*xk[a];
    }
    else if(tid < j0 + 8)
    {
      double v = 0.0;
#pragma unroll
      for(int a = 0; a < 8; a++) v = (tid - j0 == a) ? xk[a] : v;
      xs[tid] = v;
    }
#pragma unroll
    for(int a = 0; a < 8; a++) lv[a] = ln[a];
  }
  __syncthreads();
  for(int j = tid; j < w; j += BWD_NT) { ywork[c0 + j] = xs[j]; out[perm[c0 + j]] = xs[j]; }
}

} // namespace

// per-level launch parameters of the solve kernels
int sparse_solve_setup(dlg_backend* b)
{
  SparseSym* Y = b->sym;
  const SymHost& H = Y->H;
  Y->slv_lds.assign(H.nlevels, 0); Y->bwd_lds.assign(H.nlevels, 0); Y->bwd_nt.assign(H.nlevels, 512);
  for(int l = 0; l < H.nlevels; l++)
  {
    long maxw = 0, mb = 0;
    for(int i = H.lvl_ptr[l]; i < H.lvl_ptr[l+1]; i++)
    {
      const int s = H.lvl_sn[i];
      const long wv = H.sn_c0[s+1] - H.sn_c0[s], nr = H.sn_rowptr[s+1] - H.sn_rowptr[s];
      if(wv > maxw) maxw = wv;
      const long need = (nr - wv + 2) + 256 + ((wv + 7)/8)*64 + 16;   // xb, xs, diagonal blocks, rhs
      if(need > mb) mb = need;
    }
    Y->slv_lds[l] = (int)(maxw*(maxw | 1)*8);
    if(Y->slv_lds[l] > LDS_BUDGET) { dlg_set_error("supernode of width %ld is too wide for the solve kernels", maxw); return DLG_ERR_ARG; }
    if(mb*8 > LDS_BUDGET) { dlg_set_error("supernode too large for the backward-solve kernel (%ld doubles)", mb); return DLG_ERR_ARG; }
    Y->bwd_lds[l] = (int)(mb*8);
    // thread = row of the diagonal block: 256 threads when every supernode of a populous level is
    // narrow (more workgroups per CU), else 512
    Y->bwd_nt[l] = (maxw <= 128 && H.lvl_ptr[l+1] - H.lvl_ptr[l] >= 512) ? 256 : 512;
  }
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_fwd_level),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET));
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_bwd_level<256>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET));
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_bwd_level<512>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET));
  return DLG_OK;
}

// K6: out = (L L')^-1 rhs in the original variable order
int sparse_solve(dlg_backend* b, const double* rhs, double* out)
{
  SparseSym* Y = b->sym;
  if(!Y) { dlg_set_error("dlg_sparse_set_pattern must be called first"); return DLG_ERR_STATE; }
  const SymHost& H = Y->H;
  hipStream_t st = b->stream;
  const int use_aug = (Y->aug_rhs != nullptr && Y->aug_rhs == rhs) ? 1 : 0;
  for(int l = 0; l < H.nlevels && !use_aug; l++)
  {
    const int n = H.lvl_ptr[l+1] - H.lvl_ptr[l];
    if(n > 0)
      hipLaunchKernelGGL(k_solve_fwd_level, dim3(n), dim3(TPB), Y->slv_lds[l], st, Y->lvl_sn + H.lvl_ptr[l],
                         Y->sn_c0, Y->sn_rowptr, Y->sn_lx, Y->sn_scr, Y->rl_ptr, Y->rl_pos, Y->perm,
                         Y->Lx, rhs, Y->scr, Y->ywork);
  }
  for(int l = H.nlevels - 1; l >= 0; l--)
  {
    const int n = H.lvl_ptr[l+1] - H.lvl_ptr[l];
    // thread = row of the diagonal block: 256 threads when every supernode of a populous level is
    // narrow (more workgroups per CU), else 512
    if(n > 0 && Y->bwd_nt[l] == 256)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_solve_bwd_level<256>), dim3(n), dim3(256), Y->bwd_lds[l], st,
                         Y->lvl_sn + H.lvl_ptr[l], Y->sn_c0, Y->sn_rowptr, Y->sn_rows, Y->sn_lx, Y->perm, Y->Lx,
                         Y->ywork, out, use_aug, Y->sn_bd_ptr, Y->sn_bd_col);
    else if(n > 0)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_solve_bwd_level<512>), dim3(n), dim3(512), Y->bwd_lds[l], st,
                         Y->lvl_sn + H.lvl_ptr[l], Y->sn_c0, Y->sn_rowptr, Y->sn_rows, Y->sn_lx, Y->perm, Y->Lx,
                         Y->ywork, out, use_aug, Y->sn_bd_ptr, Y->sn_bd_col);
  }
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}

