// sparse_multi.hip -- blocked multi-right-hand-side solves with the resident supernodal factor
// (SURVEY 8f-3): (JtJ + lambda I) U = R for MR = 16 right-hand sides per pass over the factor,
// replacing cholmod_solve / cholmod_spsolve on a dense block of right-hand sides (reference:
// pseudoinverse_J_sparse dogleg.c:1863-1921, cholmod_spsolve 2864-2868).
//
// Layout: the MR right-hand sides are interleaved, element (variable k, rhs c) at [k*MR + c], so
// a supernode's rows are contiguous 128-byte records.  Per supernode and sweep one workgroup:
//   forward   y_t = L_tt^-1 (b_t - gathered updates);  U = L_below y_t  (r x 16) on the matrix
//             cores: A = a 16-row tile of L_below read straight from HBM (each entry of the panel
//             exactly once for all 16 right-hand sides), B = y_t from LDS (v_mfma_f64_16x16x4_f64);
//   backward  x_t = L_tt^-T (y_t - L_below^T x_below): the mat-mat product again on the matrix
//             cores, L_below staged through LDS in chunks of 32 rows (coalesced reads).
// The triangular part keeps L_tt as a packed lower triangle in LDS (w <= 128) and runs one
// barrier per column with thread = (right-hand side, row group).
#include "sparse_internal.h"

namespace {
constexpr int MR = 16;                 // right-hand sides per pass
constexpr int MS_WMAX = 128;           // widest supernode these kernels take
typedef double ms_v4d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int tri(int i, int j) { return i*(i + 1)/2 + j; }     // packed lower triangle, row-major

// stage the w x w top block of panel L (column-major, ld = nrows) as a packed lower triangle + reciprocal pivots
__device__ __forceinline__ void ms_stage_top(const double* __restrict__ L, int nrows, int w, double* Lt, double* dinv, int tid)
{
  for(int e = tid; e < w*w; e += TPB)
  {
    const int j = e / w, i = e - j*w;
    if(i >= j) { const double v = L[i + (size_t)j*nrows]; Lt[tri(i, j)] = v; if(i == j) dinv[j] = 1.0/v; }
  }
}

__global__ void __launch_bounds__(TPB) k_msolve_fwd_level(const int* __restrict__ lvl_sn,
                                                          const int* __restrict__ sn_c0,
                                                          const int* __restrict__ sn_rowptr,
                                                          const int64_t* __restrict__ sn_lx,
                                                          const int* __restrict__ sn_scr,
                                                          const int* __restrict__ rl_ptr,
                                                          const int* __restrict__ rl_pos,
                                                          const int* __restrict__ perm,
                                                          const double* __restrict__ Lx,
                                                          const double* __restrict__ B,
                                                          double* __restrict__ scr,
                                                          double* __restrict__ Y)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int s = lvl_sn[blockIdx.x];
  const int c0 = sn_c0[s], w = sn_c0[s+1] - c0;
  const int nrows = sn_rowptr[s+1] - sn_rowptr[s];
  const int r = nrows - w - 1;                    // the augmented row is not part of these solves
  const double* L = Lx + sn_lx[s];
  const int tid = threadIdx.x, c = tid & (MR - 1), g = tid >> 4;      // 16 row groups
  double* Lt = lds;                               // packed lower triangle
  double* dinv = Lt + ((w*(w + 1)/2 + 1) & ~1);
  double* Ys = dinv + ((w + 1) & ~1);             // [w][MR] running right-hand sides
  double* Yd = Ys + w*MR;                         // [w][MR] the solution y_t
  ms_stage_top(L, nrows, w, Lt, dinv, tid);
  for(int j = g; j < w; j += TPB/MR)
  {
    const int k = c0 + j;
    double sum = 0.0;
    for(int e = rl_ptr[k]; e < rl_ptr[k+1]; e++) sum += scr[(size_t)rl_pos[e]*MR + c];
    Ys[j*MR + c] = B[(size_t)perm[k]*MR + c] - sum;
  }
  __syncthreads();
  for(int j = 0; j < w; j++)
  {
    const double yj = Ys[j*MR + c]*dinv[j];
    if(g == (j & 15)) Yd[j*MR + c] = yj;
    for(int i = j + 1 + g; i < w; i += TPB/MR) Ys[i*MR + c] -= Lt[tri(i, j)]*yj;
    __syncthreads();
  }
  for(int j = g; j < w; j += TPB/MR) Y[(size_t)(c0 + j)*MR + c] = Yd[j*MR + c];
  // U = L_below y_t on the matrix cores: a wave takes row tiles of 16
  const int lane = tid & 63, wv = tid >> 6, mm = lane & 15, kq = lane >> 4;
  double* U = scr + (size_t)sn_scr[s]*MR;
  for(int t = wv; 16*t < r; t += TPB/64)
  {
    const int row = 16*t + mm;
    const double* Lr = L + w + min(row, r - 1);
    ms_v4d acc = {0.0, 0.0, 0.0, 0.0};
    for(int k4 = 0; k4 < w; k4 += 8)
    {
      // two k-steps per round, their loads issued together
      const int ka = k4 + kq, kb = k4 + 4 + kq;
      const double a0 = (ka < w && row < r) ? Lr[(size_t)ka*nrows] : 0.0;
      const double a1 = (kb < w && row < r) ? Lr[(size_t)kb*nrows] : 0.0;
      const double b0 = (ka < w) ? Yd[ka*MR + mm] : 0.0;
      const double b1 = (kb < w) ? Yd[kb*MR + mm] : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
    }
#pragma unroll
    for(int q = 0; q < 4; q++) { const int i = 16*t + kq + 4*q; if(i < r) U[(size_t)i*MR + mm] = acc[q]; }
  }
}

constexpr int MS_CH = 32;              // below rows staged per round of the backward product
__global__ void __launch_bounds__(TPB) k_msolve_bwd_level(const int* __restrict__ lvl_sn,
                                                          const int* __restrict__ sn_c0,
                                                          const int* __restrict__ sn_rowptr,
                                                          const int* __restrict__ sn_rows,
                                                          const int64_t* __restrict__ sn_lx,
                                                          const int* __restrict__ perm,
                                                          const double* __restrict__ Lx,
                                                          double* __restrict__ Y,       // in: y (permuted), out: x (permuted)
                                                          double* __restrict__ out)     // x in the original variable order
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int s = lvl_sn[blockIdx.x];
  const int c0 = sn_c0[s], w = sn_c0[s+1] - c0;
  const int nrows = sn_rowptr[s+1] - sn_rowptr[s];
  const int* rows = sn_rows + sn_rowptr[s];
  const int r = nrows - w - 1;
  const double* L = Lx + sn_lx[s];
  const int tid = threadIdx.x, c = tid & (MR - 1), g = tid >> 4;
  const int lane = tid & 63, wv = tid >> 6, mm = lane & 15, kq = lane >> 4;
  double* Lt = lds;
  double* dinv = Lt + ((w*(w + 1)/2 + 1) & ~1);
  double* Vs = dinv + ((w + 1) & ~1);             // [w][MR] y_t - L_below^T x_below
  double* Xd = Vs + w*MR;                         // [w][MR] the solution x_t
  double* Ls = Xd + w*MR;                         // [w][MS_CH + 1] a chunk of L_below, column j at Ls + j*(MS_CH + 1)
  double* Xs = Ls + w*(MS_CH + 1);                // [MS_CH][MR] x at the chunk's rows
  ms_stage_top(L, nrows, w, Lt, dinv, tid);
  // V = L_below^T X_below: wave wv owns the column tiles wv, wv + 4 (w <= 128: at most two)
  ms_v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
  const int t0 = wv, t1 = wv + TPB/64;
  for(int i0 = 0; i0 < r; i0 += MS_CH)
  {
    const int nr = min(MS_CH, r - i0);
    __syncthreads();
    for(int e = tid; e < MS_CH*w; e += TPB)
    {
      const int j = e / MS_CH, i = e - j*MS_CH;                 // consecutive threads: consecutive rows of one column
      Ls[j*(MS_CH + 1) + i] = (i < nr) ? L[w + i0 + i + (size_t)j*nrows] : 0.0;
    }
    for(int e = tid; e < MS_CH*MR; e += TPB)
    {
      const int i = e / MR, cc = e - i*MR;
      Xs[e] = (i < nr) ? Y[(size_t)rows[w + i0 + i]*MR + cc] : 0.0;
    }
    __syncthreads();
    for(int k4 = 0; k4 < MS_CH; k4 += 4)
    {
      const double b = Xs[(k4 + kq)*MR + mm];
      if(16*t0 < w) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ls[min(16*t0 + mm, w - 1)*(MS_CH + 1) + k4 + kq], b, acc0, 0, 0, 0);
      if(16*t1 < w) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ls[min(16*t1 + mm, w - 1)*(MS_CH + 1) + k4 + kq], b, acc1, 0, 0, 0);
    }
  }
  __syncthreads();
  // D[i][n]: this lane holds rows i = kq + 4q of its tiles, column n = mm
#pragma unroll
  for(int q = 0; q < 4; q++)
  {
    const int j0 = 16*t0 + kq + 4*q, j1 = 16*t1 + kq + 4*q;
    if(j0 < w) Vs[j0*MR + mm] = Y[(size_t)(c0 + j0)*MR + mm] - acc0[q];
    if(j1 < w) Vs[j1*MR + mm] = Y[(size_t)(c0 + j1)*MR + mm] - acc1[q];
  }
  __syncthreads();
  for(int j = w - 1; j >= 0; j--)
  {
    const double xj = Vs[j*MR + c]*dinv[j];
    if(g == (j & 15)) Xd[j*MR + c] = xj;
    for(int i = g; i < j; i += TPB/MR) Vs[i*MR + c] -= Lt[tri(j, i)]*xj;        // (L^T)[i][j] = L[j][i]
    __syncthreads();
  }
  for(int j = g; j < w; j += TPB/MR)
  {
    const double v = Xd[j*MR + c];
    Y[(size_t)(c0 + j)*MR + c] = v;
    out[(size_t)perm[c0 + j]*MR + c] = v;
  }
}

// column-major (ld = N) block of ncols <= MR columns <-> interleaved [N][MR] (unused columns zero)
__global__ void __launch_bounds__(TPB) k_cols_to_interleaved(const double* __restrict__ cols, int N, int ncols, double* __restrict__ il)
{
  const size_t e = (size_t)blockIdx.x*TPB + threadIdx.x;
  if(e >= (size_t)N*MR) return;
  const int k = (int)(e / MR), c = (int)(e % MR);
  il[e] = (c < ncols) ? cols[(size_t)c*N + k] : 0.0;
}
__global__ void __launch_bounds__(TPB) k_interleaved_to_cols(const double* __restrict__ il, int N, int ncols, double* __restrict__ cols)
{
  const size_t e = (size_t)blockIdx.x*TPB + threadIdx.x;
  if(e >= (size_t)N*ncols) return;
  const int c = (int)(e / N), k = (int)(e % N);
  cols[e] = il[(size_t)k*MR + c];
}
// interleaved block of Jt[:, row0 : row0 + ncols] from the rank-local CSC pattern / values (zero elsewhere)
__global__ void __launch_bounds__(TPB) k_jt_chunk_sparse(const int* __restrict__ Jp, const int* __restrict__ Ji,
                                                         const double* __restrict__ Jv, int row0, int ncols,
                                                         double* __restrict__ il)
{
  const int c = blockIdx.x;
  if(c >= ncols) return;
  for(int q = Jp[row0 + c] + threadIdx.x; q < Jp[row0 + c + 1]; q += TPB) il[(size_t)Ji[q]*MR + c] = Jv[q];
}

size_t ms_lds_fwd(int w) { return sizeof(double)*(size_t)(((w*(w + 1)/2 + 1) & ~1) + ((w + 1) & ~1) + 2*w*MR); }
size_t ms_lds_bwd(int w) { return ms_lds_fwd(w) + sizeof(double)*(size_t)(w*(MS_CH + 1) + MS_CH*MR); }
} // namespace

int sparse_multi_width_ok(const dlg_backend* b)
{
  const SymHost& H = b->sym->H;
  for(int s = 0; s < H.nsn; s++) if(H.sn_c0[s+1] - H.sn_c0[s] > MS_WMAX) return 0;
  return H.part_nranks <= 1;
}
int sparse_multi_rhs() { return MR; }

// d_il: [N][MR] interleaved right-hand sides in the ORIGINAL variable order; solved in place
int sparse_solve_multi(dlg_backend* b, double* d_il)
{
  SparseSym* Y = b->sym;
  if(!Y) { dlg_set_error("dlg_sparse_set_pattern must be called first"); return DLG_ERR_STATE; }
  const SymHost& H = Y->H;
  hipStream_t st = b->stream;
  if(!Y->ms_scr)
  {
    DLG_HIP(hipMalloc(&Y->ms_scr, sizeof(double)*((size_t)H.scr_size + 1)*MR)); Y->allocs.push_back(Y->ms_scr);
    DLG_HIP(hipMalloc(&Y->ms_y, sizeof(double)*(size_t)H.N*MR)); Y->allocs.push_back(Y->ms_y);
    int wmax = 1;
    for(int s = 0; s < H.nsn; s++) wmax = std::max(wmax, H.sn_c0[s+1] - H.sn_c0[s]);
    Y->ms_lds_f = (int)ms_lds_fwd(wmax); Y->ms_lds_b = (int)ms_lds_bwd(wmax);
    DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_msolve_fwd_level), hipFuncAttributeMaxDynamicSharedMemorySize, Y->ms_lds_f));
    DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_msolve_bwd_level), hipFuncAttributeMaxDynamicSharedMemorySize, Y->ms_lds_b));
  }
  for(int l = 0; l < H.nlevels; l++)
  {
    const int n = H.xl_ptr[l+1] - H.xl_ptr[l];
    if(n > 0)
      hipLaunchKernelGGL(k_msolve_fwd_level, dim3(n), dim3(TPB), Y->ms_lds_f, st, Y->xl_sn + H.xl_ptr[l], Y->sn_c0,
                         Y->sn_rowptr, Y->sn_lx, Y->sn_scr, Y->rl_ptr, Y->rl_pos, Y->perm, Y->Lx, d_il, Y->ms_scr, Y->ms_y);
  }
  for(int l = H.nlevels - 1; l >= 0; l--)
  {
    const int n = H.xl_ptr[l+1] - H.xl_ptr[l];
    if(n > 0)
      hipLaunchKernelGGL(k_msolve_bwd_level, dim3(n), dim3(TPB), Y->ms_lds_b, st, Y->xl_sn + H.xl_ptr[l], Y->sn_c0,
                         Y->sn_rowptr, Y->sn_rows, Y->sn_lx, Y->perm, Y->Lx, Y->ms_y, d_il);
  }
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}
int multi_cols_to_interleaved(dlg_backend* b, const double* d_cols, int ncols, double* d_il)
{
  hipLaunchKernelGGL(k_cols_to_interleaved, dim3(dlg_cdiv((long)b->N*MR, TPB)), dim3(TPB), 0, b->stream, d_cols, b->N, ncols, d_il);
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}
int multi_interleaved_to_cols(dlg_backend* b, const double* d_il, int ncols, double* d_cols)
{
  hipLaunchKernelGGL(k_interleaved_to_cols, dim3(dlg_cdiv((long)b->N*ncols, TPB)), dim3(TPB), 0, b->stream, d_il, b->N, ncols, d_cols);
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}
int sparse_jt_chunk_interleaved(dlg_backend* b, int s, int row0, int ncols, double* d_il)
{
  SparseSym* Y = b->sym;
  DLG_HIP(hipMemsetAsync(d_il, 0, sizeof(double)*(size_t)b->N*MR, b->stream));
  hipLaunchKernelGGL(k_jt_chunk_sparse, dim3(ncols), dim3(TPB), 0, b->stream, Y->Jp, Y->Ji, b->slot[s].Jin(), row0, ncols, d_il);
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}
