// kernels_dense.hip -- DOGLEG_DENSE / DOGLEG_DENSE_PRODUCTS hot path on gfx950.
//
//   K1-dense  Jt_x = J^T x            replaces mul_matrix_t_densevector (dogleg.c:284-292)
//   K3/K8     |J v|^2                 replaces norm2_mul_matrix_vector  (dogleg.c:293-306)
//   K4-dense  JtJ = J^T J (+lambda I) replaces the rank-1 loop          (dogleg.c:709-723)
//             -> fp64 MFMA SYRK (v_mfma_f64_16x16x4_f64), split over the
//                measurement rows, lower triangle only
//   K5-dense  Cholesky                replaces dpptrf_/dpotrf_          (dogleg.c:782-803)
//             -> blocked right-looking: LDS diagonal block, row-parallel TRSM,
//                MFMA trailing update (same SYRK kernel)
//   K6-dense  two triangular solves   replaces dpptrs_/dpotrs_          (dogleg.c:875-891)
//
// Storage: the factor lives in G, an N x N column-major array whose LOWER
// triangle is used.  The reference's "row-major packed upper" triangle is the
// same matrix as a column-major lower triangle (dogleg.c:788-790); the packed
// layout exists only at the API edge (dlg_factor_download_dense).
//
// J is row-major [M][N]: element (r,c) at r*N + c, i.e. J^T in column-major
// with leading dimension N.  So "A[i + k*lda]" of the SYRK kernel reads J
// directly with i = state index, k = measurement row: for a fixed row the 128
// state entries of a tile are contiguous -> coalesced 512-B wave loads.
#include "dlg_internal.h"
#include "panel_factor.h"

typedef double double4_t __attribute__((ext_vector_type(4)));

namespace {

constexpr int TPB = 256;

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for(int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// ------------------------------------------------------------------ K1 -----
// partial[chunk][c] = sum_{r in chunk} J[r][c] x[r]
__global__ void __launch_bounds__(TPB) k_gemvT_part(const double* __restrict__ J,
                                                    const double* __restrict__ x, int M, int N,
                                                    int rows_per_chunk, double* __restrict__ part)
{
  const int c = blockIdx.x*TPB + threadIdx.x;
  const int r0 = blockIdx.y*rows_per_chunk;
  int r1 = r0 + rows_per_chunk; if(r1 > M) r1 = M;
  if(c >= N) return;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  int r = r0;
  for(; r + 3 < r1; r += 4)
  {
    a0 += J[(size_t)(r  )*N + c]*x[r  ];
    a1 += J[(size_t)(r+1)*N + c]*x[r+1];
    a2 += J[(size_t)(r+2)*N + c]*x[r+2];
    a3 += J[(size_t)(r+3)*N + c]*x[r+3];
  }
  for(; r < r1; r++) a0 += J[(size_t)r*N + c]*x[r];
  part[(size_t)blockIdx.y*N + c] = (a0 + a1) + (a2 + a3);
}
__global__ void __launch_bounds__(TPB) k_colsum(const double* __restrict__ part, int nchunks, int N,
                                                double* __restrict__ out)
{
  const int c = blockIdx.x*TPB + threadIdx.x;
  if(c >= N) return;
  double s = 0;
  for(int k = 0; k < nchunks; k++) s += part[(size_t)k*N + c];
  out[c] = s;
}

// first stage of the column sums for long lists of partials: slice y adds its share of the chunks
// (in chunk order), k_colsum then adds the slices (in slice order): deterministic, and S times more
// workgroups than one column-sum pass over all chunks
__global__ void __launch_bounds__(TPB) k_colsum_slices(const double* __restrict__ part, int nchunks, int N,
                                                       double* __restrict__ part2)
{
  const int c = blockIdx.x*TPB + threadIdx.x;
  if(c >= N) return;
  const int S = gridDim.y, y = blockIdx.y;
  const int k0 = (int)((long)y*nchunks/S), k1 = (int)((long)(y + 1)*nchunks/S);
  double s0 = 0, s1 = 0;
  int k = k0;
  for(; k + 1 < k1; k += 2) { s0 += part[(size_t)k*N + c]; s1 += part[(size_t)(k + 1)*N + c]; }
  if(k < k1) s0 += part[(size_t)k*N + c];
  part2[(size_t)y*N + c] = s0 + s1;
}

// --------------------------------------------------------------- K3 / K8 ---
// one wave per measurement row: dot(J[r,:], v)^2 accumulated per wave
__global__ void __launch_bounds__(TPB) k_norm2_Jv_part(const double* __restrict__ J,
                                                       const double* __restrict__ v, int M, int N,
                                                       double* __restrict__ part,
                                                       const double* __restrict__ psrc, double* __restrict__ pdst, int pn,
                                                       const double* __restrict__ skipf)
{
  __shared__ double sh[4];
  // (behind the decision point, dlg_take_step: p_new goes to its page-locked destination with this pass, a slice a workgroup)
  if(psrc)
  {
    const int per = (pn + (int)gridDim.x - 1)/(int)gridDim.x, i0 = (int)blockIdx.x*per, i1 = min(i0 + per, pn);
    for(int i = i0 + (int)threadIdx.x; i < i1; i += TPB) pdst[i] = psrc[i];
  }
  // K8 of a step whose |J step|^2 the host takes from the solved system (the step kernel's word, k_part_take_step)
  if(skipf && *skipf != 0.0) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wave = blockIdx.x*4 + w, nwaves = gridDim.x*4;
  double acc = 0;
  // (rows of an even length from a 16-byte aligned base start on 16-byte boundaries)
  const bool pairs = (N & 1) == 0 && (((size_t)J | (size_t)v) & 15) == 0;
  for(int r = wave; r < M; r += nwaves)
  {
    const double* Jr = J + (size_t)r*N;
    double d = 0;
    if(pairs)
    {
      // 16-byte loads: a lane takes columns 2*lane, 2*lane + 1, then 128 further on (four loads in flight)
#pragma unroll 4
      for(int c = 2*lane; c < N; c += 128)
      {
        const double2 jv = *reinterpret_cast<const double2*>(Jr + c), vv = *reinterpret_cast<const double2*>(v + c);
        d += jv.x*vv.x; d += jv.y*vv.y;
      }
    }
    else
      for(int c = lane; c < N; c += 64) d += Jr[c]*v[c];
    d = wave_sum(d);
    acc += d*d;          // only lane 0 holds the full sum; others add garbage we ignore
  }
  if(lane == 0) sh[w] = acc;
  __syncthreads();
  if(threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// ------------------------------------------------------- products helpers ---
// v' A v with A row-major packed upper (dogleg.c:309-332) or full (335-347)
__global__ void __launch_bounds__(TPB) k_quadform_part(const double* __restrict__ A,
                                                       const double* __restrict__ v, int N,
                                                       int packed_upper, double* __restrict__ part)
{
  __shared__ double sh[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wave = blockIdx.x*4 + w, nwaves = gridDim.x*4;
  double acc = 0;
  for(int i = wave; i < N; i += nwaves)
  {
    double d = 0;
    if(packed_upper)
    {
      const double* row = A + ((size_t)i*N - (size_t)i*(i-1)/2) - i;   // row[j] valid for j>=i
      for(int j = i + lane; j < N; j += 64) d += (j == i ? 1.0 : 2.0)*row[j]*v[j];
    }
    else
    {
      const double* row = A + (size_t)i*N;
      for(int j = lane; j < N; j += 64) d += row[j]*v[j];
    }
    d = wave_sum(d);
    acc += d*v[i];
  }
  if(lane == 0) sh[w] = acc;
  __syncthreads();
  if(threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
// G (col-major lower, ld N) <- JtJ given as row-major packed upper, or full
__global__ void __launch_bounds__(TPB) k_unpack_to_G(const double* __restrict__ A, int N,
                                                     int packed_upper, double lambda,
                                                     double* __restrict__ G)
{
  const size_t t = (size_t)blockIdx.x*TPB + threadIdx.x;
  if(t >= (size_t)N*N) return;
  const int i = (int)(t % N), j = (int)(t / N);       // G[i + j*N]
  if(i < j) return;
  double v;
  if(packed_upper) v = A[((size_t)j*N - (size_t)j*(j-1)/2) + (i - j)];
  else             v = A[(size_t)j*N + i];
  if(i == j) v += lambda;
  G[t] = v;
}
__global__ void __launch_bounds__(TPB) k_add_diag(double* __restrict__ G, int N, double lambda)
{
  const int i = blockIdx.x*TPB + threadIdx.x;
  if(i < N) G[(size_t)i*N + i] += lambda;
}
// packed (as dpptrf('L') leaves it) <- G
__global__ void __launch_bounds__(TPB) k_pack_from_G(const double* __restrict__ G, int N,
                                                     double* __restrict__ P)
{
  const size_t t = (size_t)blockIdx.x*TPB + threadIdx.x;
  if(t >= (size_t)N*N) return;
  const int i = (int)(t % N), j = (int)(t / N);
  if(i < j) return;
  P[((size_t)j*N - (size_t)j*(j-1)/2) + (i - j)] = G[t];
}

// ------------------------------------------------------------- MFMA SYRK ----
// C[i,j] (i>=j, column-major, ldc) (+)= alpha * sum_{k<K} A[i + k*lda] A[j + k*lda]
//
// Workgroup = 4 waves in a 2x2 arrangement; block tile BT x BT with BT = 2*WT;
// each wave owns WT x WT = (WT/16)^2 MFMA tiles of 16x16 (v_mfma_f64_16x16x4_f64).
// K is consumed in chunks of KC=16 rows staged through LDS (double-buffered,
// one barrier per chunk).  LDS rows are padded to BT+16 doubles so that the four
// k-rows a wave reads in one ds_read_b64 fall in different bank halves.
// The MFMA is issued "transposed" (operand A <- column index j, operand B <- row
// index i) so that the 16 lanes of an accumulator register hold 16 consecutive
// i of one column j: 128-B contiguous stores into column-major C.
constexpr int KC = 16;

template <int WT>
__global__ void __launch_bounds__(TPB, 2)
k_syrk_lower(double* __restrict__ C, int ldc, const double* __restrict__ A, int lda, int n, int K,
             double alpha, int nsplit, int kper, double* __restrict__ slabs)
{
  constexpr int BT = 2*WT;
  constexpr int LDS_LD = BT + 16;
  constexpr int NT = WT/16;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  // layout: [buf][operand][KC][LDS_LD]
  auto sA = [&](int buf) -> double* { return smem + (size_t)(2*buf    )*KC*LDS_LD; };
  auto sB = [&](int buf) -> double* { return smem + (size_t)(2*buf + 1)*KC*LDS_LD; };

  // (split-major: the workgroups resident at a time are the tiles of ONE range of rows, marching through it together --
  // the rows a chunk needs, 16 x N doubles, are shared by all of them in L2; tile-major, every resident workgroup read
  // rows of its own and 11.9 GB crossed the fabric per launch for 0.8 GB of J, profiles/r04_pmc.md)
  const int ntl   = gridDim.x / nsplit;
  const int split = blockIdx.x / ntl;
  const int pos   = blockIdx.x - split*ntl;
  // ... and XCD-aware: consecutive workgroups go to consecutive XCDs, each with an L2 of its own -- the positions
  // pos = r (mod 8) of a split, which share an XCD, get a CONTIGUOUS stretch of the tiles listed by 4 x 4 super-tiles,
  // i.e. about one super-tile: 8 of the 16 column blocks of J instead of all of them through that L2
  int ti, tj;
  {
    const int T = (int)((sqrt(8.0*(double)ntl + 1.0) - 1.0)*0.5 + 0.5);      // ntl = T (T + 1) / 2
    const int r = pos & 7, q = pos >> 3;
    int idx = q;
    for(int x = 0; x < r; x++) idx += (ntl - x + 7) >> 3;                    // positions of the groups in front
    constexpr int S = 4;
    const int TS = (T + S - 1)/S;
    ti = tj = 0;
    bool found = false;
    for(int a = 0; a < TS && !found; a++)
      for(int bq = 0; bq <= a && !found; bq++)
      {
        const int ra = min(S, T - a*S), rb = min(S, T - bq*S);
        const int cnt = (a == bq) ? ra*(ra + 1)/2 : ra*rb;
        if(idx >= cnt) { idx -= cnt; continue; }
        int li, lj;
        if(a != bq) { li = idx / rb; lj = idx - li*rb; }
        else { li = 0; while(idx > li) { idx -= li + 1; li++; } lj = idx; }
        ti = a*S + li; tj = bq*S + lj; found = true;
      }
  }
  const int tile = ti*(ti + 1)/2 + tj;
  const int i0 = ti*BT, j0 = tj*BT;
  const bool diag = (ti == tj);

  const int k_begin = split*kper;
  int k_end = k_begin + kper; if(k_end > K) k_end = K;
  const int nchunks = (k_end > k_begin) ? (k_end - k_begin + KC - 1)/KC : 0;

  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wi = (w & 1)*WT, wj = (w >> 1)*WT;     // wave offsets inside the block tile

  // global -> register staging: BT*KC doubles per operand, TPB threads
  constexpr int PER_T = BT*KC/TPB;                 // 8 (BT=128) or 4 (BT=64)
  constexpr int ROWS_PER_IT = TPB/BT;              // 2 or 4
  double ra[PER_T], rb[PER_T];
  const int li = t % BT, lk = t / BT;

  auto gload = [&](int chunk) {
    const int kb = k_begin + chunk*KC;
#pragma unroll
    for(int it = 0; it < PER_T; it++)
    {
      const int k = kb + it*ROWS_PER_IT + lk;
      const bool kin = k < k_end;
#ifdef DLG_SYRK_NO_GLOAD                      // (tools/variant_lib.sh: the kernel without its loads of J)
      ra[it] = (kin && i0 + li < n) ? (double)((k + li) & 7) : 0.0;
      if(!diag) rb[it] = (kin && j0 + li < n) ? (double)((k - li) & 7) : 0.0;
#else
      ra[it] = (kin && i0 + li < n) ? A[(size_t)k*lda + i0 + li] : 0.0;
      if(!diag) rb[it] = (kin && j0 + li < n) ? A[(size_t)k*lda + j0 + li] : 0.0;
#endif
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for(int it = 0; it < PER_T; it++)
    {
      const int k = it*ROWS_PER_IT + lk;
      sA(buf)[k*LDS_LD + li] = ra[it];
      if(!diag) sB(buf)[k*LDS_LD + li] = rb[it];
    }
  };

  double4_t acc[NT][NT];
#pragma unroll
  for(int a = 0; a < NT; a++)
#pragma unroll
    for(int c = 0; c < NT; c++) acc[a][c] = (double4_t){0.0, 0.0, 0.0, 0.0};

  if(nchunks > 0) { gload(0); sstore(0); }
  __syncthreads();

  const int fr = lane & 15, fk = lane >> 4;
  for(int c = 0; c < nchunks; c++)
  {
    const int buf = c & 1;
    if(c + 1 < nchunks) gload(c + 1);
    const double* pa = sA(buf);
    const double* pb = diag ? sA(buf) : sB(buf);
#pragma unroll
    for(int kk = 0; kk < KC; kk += 4)
    {
      double fi[NT], fj[NT];
#pragma unroll
      for(int m = 0; m < NT; m++)
      {
        fi[m] = pa[(kk + fk)*LDS_LD + wi + m*16 + fr];
        fj[m] = pb[(kk + fk)*LDS_LD + wj + m*16 + fr];
      }
#pragma unroll
      for(int a = 0; a < NT; a++)          // a: j sub-tile (MFMA rows)
#pragma unroll
        for(int bq = 0; bq < NT; bq++)     // bq: i sub-tile (MFMA cols)
          acc[a][bq] = __builtin_amdgcn_mfma_f64_16x16x4f64(fj[a], fi[bq], acc[a][bq], 0, 0, 0);
    }
    if(c + 1 < nchunks) sstore(buf ^ 1);
    __syncthreads();
  }

  // epilogue.  acc[a][bq][q]: j = j0+wj+a*16 + (lane>>4) + 4q ; i = i0+wi+bq*16 + (lane&15)
  if(nsplit == 1)
  {
#pragma unroll
    for(int a = 0; a < NT; a++)
#pragma unroll
      for(int bq = 0; bq < NT; bq++)
#pragma unroll
        for(int q = 0; q < 4; q++)
        {
          const int j = j0 + wj + a*16 + fk + 4*q;
          const int i = i0 + wi + bq*16 + fr;
          if(i < n && j < n && i >= j) C[(size_t)j*ldc + i] += alpha*acc[a][bq][q];
        }
  }
  else
  {
    double* slab = slabs + ((size_t)split*ntl + tile)*(size_t)(BT*BT);
#pragma unroll
    for(int a = 0; a < NT; a++)
#pragma unroll
      for(int bq = 0; bq < NT; bq++)
#pragma unroll
        for(int q = 0; q < 4; q++)
        {
          const int jl = wj + a*16 + fk + 4*q;
          const int il = wi + bq*16 + fr;
          slab[(size_t)jl*BT + il] = acc[a][bq][q];
        }
  }
}

// C[i,j] = beta*C[i,j] + alpha * sum_s slab[s][tile][jl][il] (+ lambda on the diagonal)
// (SYRK_RSUB workgroups per tile, a thread sums a PAIR of rows over the splits in split order -- 16-byte
// loads, several splits in flight: as one workgroup per tile with 8-byte loads and one dependent load per
// add this pass read its 214 MB (config #2: 12 splits) at 0.7 TB/s)
constexpr int SYRK_RSUB = 8;
template <int BT>
__global__ void __launch_bounds__(TPB) k_syrk_reduce(double* __restrict__ C, int ldc, int n,
                                                     const double* __restrict__ slabs, int nsplit,
                                                     int ntiles, double alpha, double beta,
                                                     double lambda)
{
  static_assert((BT*BT) % (2*SYRK_RSUB*TPB) == 0, "a tile is a whole number of rounds");
  const int tile = blockIdx.x / SYRK_RSUB, sub = blockIdx.x % SYRK_RSUB;
  int ti = (int)((sqrt(8.0*(double)tile + 1.0) - 1.0)*0.5);
  while((long)ti*(ti+1)/2 > tile) ti--;
  while((long)(ti+1)*(ti+2)/2 <= tile) ti++;
  const int tj = tile - (int)((long)ti*(ti+1)/2);
  const size_t tsz = (size_t)BT*BT, sstride = (size_t)ntiles*tsz;
  const double* base = slabs + (size_t)tile*tsz;
  for(int e = 2*(sub*TPB + threadIdx.x); e < BT*BT; e += 2*SYRK_RSUB*TPB)
  {
    const int il = e % BT, jl = e / BT;        // il is even: rows i, i + 1 of column j
    const int i = ti*BT + il, j = tj*BT + jl;
    if(i >= n || j >= n || i + 1 < j) continue;
    double s0 = 0, s1 = 0;
#pragma unroll 4
    for(int sp = 0; sp < nsplit; sp++)
    {
      const double2 t = *reinterpret_cast<const double2*>(base + (size_t)sp*sstride + e);
      s0 += t.x; s1 += t.y;
    }
    if(i >= j)
    {
      double r = alpha*s0;
      if(beta != 0.0) r += beta*C[(size_t)j*ldc + i];
      if(i == j) r += lambda;
      C[(size_t)j*ldc + i] = r;
    }
    if(i + 1 < n)
    {
      double r = alpha*s1;
      if(beta != 0.0) r += beta*C[(size_t)j*ldc + i + 1];
      if(i + 1 == j) r += lambda;
      C[(size_t)j*ldc + i + 1] = r;
    }
  }
}

// --------------------------------------------------------------- potrf ------
// Blocked right-looking Cholesky, NB = 64:
//   k_potrf_diag_inv : one workgroup factors the 64x64 diagonal block in LDS
//                      (panel_factor.h) and also forms its inverse Linv (64 columns in
//                      parallel, forward substitution in LDS);
//   k_trsm_gemm      : the rows below become X = A * Linv^T (a small GEMM, fully
//                      parallel) instead of a 64-step substitution per row;
//   k_syrk_lower     : fp64-MFMA trailing update.
// The Linv blocks are kept (nblk x 64 x 64): the triangular solves of K6 use them as
// 64x64 mat-vecs, so no kernel on the solve path has a serial 64-step loop either.
constexpr int NB = 64;

// X = A_panel * Linv^T for the rows r >= kb+nb; one workgroup per 64 rows.
__global__ void __launch_bounds__(TPB) k_trsm_gemm(double* __restrict__ A, int lda, int kb, int nb,
                                                   int n, const double* __restrict__ Linv)
{
  __shared__ double As[NB][NB + 1];      // As[r][k]
  __shared__ double Ls[NB][NB + 1];      // Ls[c][k] = Linv[c][k]
  const int t = threadIdx.x;
  const int r0 = kb + nb + blockIdx.x*NB;
  for(int e = t; e < NB*NB; e += TPB)
  {
    const int i = e % NB, k = e / NB;     // i fastest: coalesced along rows of A and Linv
    const int r = r0 + i;
    As[i][k] = (r < n && k < nb) ? A[(size_t)(kb + k)*lda + r] : 0.0;
    Ls[i][k] = Linv[e];
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

// ------------------------------------------------------------- trsv ---------
// forward, diagonal block: y_blk = Linv * b_blk
__global__ void __launch_bounds__(64) k_trsv_diag_fwd(const double* __restrict__ Linv, int kb, int nb,
                                                      double* __restrict__ y)
{
  __shared__ double v[NB];
  const int t = threadIdx.x;
  v[t] = (t < nb) ? y[kb + t] : 0.0;
  __syncthreads();
  double sacc = 0.0;
#pragma unroll 8
  for(int k = 0; k < NB; k++) sacc += Linv[t + k*NB]*v[k];   // Linv is zero above the diagonal
  if(t < nb) y[kb + t] = sacc;
}
// forward, fused: y[i] -= sum_k L[i, kb+k] y[kb+k] for the rows below block column kb and -- by workgroup 0, whose
// rows are the NEXT diagonal block's -- the next block's y = Linv_next * y right behind it: one
// launch per block column instead of two.  Workgroup 0 covers rows kb+nb .. kb+nb+TPB-1 >= the 64
// rows of the next block.
__global__ void __launch_bounds__(TPB) k_trsv_step_fwd(const double* __restrict__ A, int lda,
                                                       int kb, int nb, int n,
                                                       const double* __restrict__ Linv_next, int nb_next,
                                                       double* __restrict__ y)
{
  __shared__ double v[NB];
  __shared__ double w[NB];
  if(threadIdx.x < nb) v[threadIdx.x] = y[kb + threadIdx.x];
  __syncthreads();
  const int i = kb + nb + blockIdx.x*TPB + threadIdx.x;
  double yi = 0.0;
  if(i < n)
  {
    double s = 0;
#pragma unroll 8
    for(int k = 0; k < nb; k++) s += A[(size_t)(kb + k)*lda + i]*v[k];
    yi = y[i] - s;
    if(blockIdx.x > 0 || threadIdx.x >= NB) y[i] = yi;         // the next block's rows are written below, solved
  }
  if(blockIdx.x == 0)
  {
    if(threadIdx.x < NB) w[threadIdx.x] = (threadIdx.x < nb_next) ? yi : 0.0;
    __syncthreads();
    if(threadIdx.x < NB)
    {
      const int t = threadIdx.x;
      double sacc = 0.0;
#pragma unroll 8
      for(int k = 0; k < NB; k++) sacc += Linv_next[t + k*NB]*w[k];   // Linv is zero above the diagonal
      if(t < nb_next) y[kb + nb + t] = sacc;
    }
  }
}
// backward, fused: x[i] -= sum_k L[kb+k, i] x[kb+k] for the rows i < kb of block column kb, and -- by
// workgroup 0, which takes the 64 rows of the PREVIOUS diagonal block (thread = (row, quarter of k)) --
// that block's x = Linv_prev^T * (updated rows) right behind it.  The other workgroups: one wave per row
// (column i of L is contiguous: a coalesced read, a wave reduction).
__global__ void __launch_bounds__(TPB) k_trsv_step_bwd(const double* __restrict__ A, int lda, int kb, int nb,
                                                       const double* __restrict__ Linv_prev,
                                                       double* __restrict__ y)
{
  __shared__ double v[NB];
  __shared__ double part[4][NB];
  __shared__ double w[NB];
  if(threadIdx.x < NB) v[threadIdx.x] = (threadIdx.x < nb) ? y[kb + threadIdx.x] : 0.0;
  __syncthreads();
  if(blockIdx.x == 0)
  {
    const int r = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int i = kb - NB + r;                     // kb >= NB: the previous block is a full one
    const double* Ai = A + (size_t)i*lda + kb + 16*g;
    double s = 0.0;
#pragma unroll
    for(int k = 0; k < 16; k++) s += (16*g + k < nb) ? Ai[k]*v[16*g + k] : 0.0;
    part[g][r] = s;
    __syncthreads();
    if(threadIdx.x < NB) w[r] = y[i] - ((part[0][r] + part[1][r]) + (part[2][r] + part[3][r]));
    __syncthreads();
    if(threadIdx.x < NB)
    {
      const double* Lc = Linv_prev + (size_t)r*NB;     // (Linv^T)[r][k] = Linv[k][r] = Linv[k + r*NB]
      double sacc = 0.0;
#pragma unroll 8
      for(int k = r; k < NB; k++) sacc += Lc[k]*w[k];
      y[i] = sacc;
    }
    return;
  }
  const int lane = threadIdx.x & 63;
  const int i = (blockIdx.x - 1)*4 + (threadIdx.x >> 6);
  if(i >= kb - NB) return;
  double s = (lane < nb) ? A[(size_t)i*lda + kb + lane]*v[lane] : 0.0;
  s = wave_sum(s);
  if(lane == 0) y[i] -= s;
}

// backward, diagonal block: x_blk = Linv^T * y_blk
__global__ void __launch_bounds__(64) k_trsv_diag_bwd(const double* __restrict__ Linv, int kb, int nb,
                                                      double* __restrict__ y)
{
  __shared__ double v[NB];
  __shared__ double Ls[NB][NB + 1];
  const int t = threadIdx.x;
  v[t] = (t < nb) ? y[kb + t] : 0.0;
#pragma unroll 8
  for(int k = 0; k < NB; k++) Ls[t][k] = Linv[t + k*NB];     // Ls[i][k] = Linv[i][k]
  __syncthreads();
  double sacc = 0.0;
  for(int k = 0; k < NB; k++) sacc += Ls[k][t]*v[k];          // (Linv^T)[t][k] = Linv[k][t]
  if(t < nb) y[kb + t] = sacc;
}
// ---- blocked multi-right-hand-side triangular solves (dpptrs / dpotrs with nrhs > 1:
// reference pseudoinverse_J_dense, dogleg.c:1831-1855).  MR = 16 right-hand sides interleaved
// [n][MR]; per 64-column block of the factor: the diagonal block is a product with its stored
// inverse, the rest of the block column is applied to all 16 right-hand sides on the matrix cores
// (v_mfma_f64_16x16x4_f64: A = a 16-row tile of L, B = the 64 x 16 block of solved unknowns in
// LDS) -- the factor is read once per 16 right-hand sides.
constexpr int DMR = 16;
// X_blk = Linv_blk * Y_blk (forward) or Linv_blk^T * Y_blk (backward)
__global__ void __launch_bounds__(TPB) k_trsm_diag_m(const double* __restrict__ Linv, int kb, int nb,
                                                     double* __restrict__ y, int transpose)
{
  __shared__ double v[NB*DMR];
  __shared__ double Ls[NB][NB + 1];
  const int tid = threadIdx.x, c = tid & (DMR - 1), tg = tid >> 4;
  for(int e = tid; e < NB*DMR; e += TPB) { const int k = e / DMR; v[e] = (k < nb) ? y[(size_t)(kb + k)*DMR + (e - k*DMR)] : 0.0; }
  for(int e = tid; e < NB*NB; e += TPB) { const int k = e / NB, t = e - k*NB; Ls[t][k] = Linv[e]; }    // Ls[i][k] = Linv[i][k]
  __syncthreads();
  for(int t = tg; t < nb; t += TPB/DMR)
  {
    double sacc = 0.0;
    if(!transpose) { for(int k = 0; k <= t; k++) sacc += Ls[t][k]*v[k*DMR + c]; }
    else           { for(int k = t; k < NB; k++) sacc += Ls[k][t]*v[k*DMR + c]; }
    y[(size_t)(kb + t)*DMR + c] = sacc;
  }
}
// forward: Y[i][:] -= L[i, kb:kb+nb] X[kb:kb+nb][:] for the rows i >= kb + nb; a wave per 16 rows
__global__ void __launch_bounds__(TPB) k_trsm_update_fwd_m(const double* __restrict__ A, int lda, int kb, int nb,
                                                           int n, double* __restrict__ y)
{
  __shared__ double v[NB*DMR];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, mm = lane & 15, kq = lane >> 4;
  for(int e = tid; e < NB*DMR; e += TPB) { const int k = e / DMR; v[e] = (k < nb) ? y[(size_t)(kb + k)*DMR + (e - k*DMR)] : 0.0; }
  __syncthreads();
  const int i0 = kb + nb + 16*(blockIdx.x*(TPB/64) + wv);
  if(i0 >= n) return;
  const int row = min(i0 + mm, n - 1);
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  for(int k4 = 0; k4 < NB; k4 += 4)
  {
    const int k = k4 + kq;
    const double a = (k < nb) ? A[(size_t)(kb + k)*lda + row] : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, v[k*DMR + mm], acc, 0, 0, 0);
  }
#pragma unroll
  for(int q = 0; q < 4; q++) { const int i = i0 + kq + 4*q; if(i < n) y[(size_t)i*DMR + mm] -= acc[q]; }
}
// backward: Y[i][:] -= L[kb:kb+nb, i]^T X[kb:kb+nb][:] for the rows i < kb
__global__ void __launch_bounds__(TPB) k_trsm_update_bwd_m(const double* __restrict__ A, int lda, int kb, int nb,
                                                           double* __restrict__ y)
{
  __shared__ double v[NB*DMR];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, mm = lane & 15, kq = lane >> 4;
  for(int e = tid; e < NB*DMR; e += TPB) { const int k = e / DMR; v[e] = (k < nb) ? y[(size_t)(kb + k)*DMR + (e - k*DMR)] : 0.0; }
  __syncthreads();
  const int i0 = 16*(blockIdx.x*(TPB/64) + wv);
  if(i0 >= kb) return;
  const int col = min(i0 + mm, kb - 1);
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  for(int k4 = 0; k4 < NB; k4 += 4)
  {
    const int k = k4 + kq;
    const double a = (k < nb) ? A[(size_t)col*lda + kb + k] : 0.0;          // (L^T)[i][k] = L[kb + k][i]
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, v[k*DMR + mm], acc, 0, 0, 0);
  }
#pragma unroll
  for(int q = 0; q < 4; q++) { const int i = i0 + kq + 4*q; if(i < kb) y[(size_t)i*DMR + mm] -= acc[q]; }
}
// interleaved block of Jt[:, row0 : row0 + ncols] of the dense J (row-major [M][N]): element (k, c) = J[row0 + c][k]
__global__ void __launch_bounds__(TPB) k_jt_chunk_dense(const double* __restrict__ J, int N, int row0, int ncols,
                                                        double* __restrict__ il)
{
  const size_t e = (size_t)blockIdx.x*TPB + threadIdx.x;
  if(e >= (size_t)N*DMR) return;
  const int k = (int)(e / DMR), c = (int)(e % DMR);
  il[e] = (c < ncols) ? J[(size_t)(row0 + c)*N + k] : 0.0;
}

// ------------------------------------------------------------ probes --------
__global__ void __launch_bounds__(TPB, 8) k_probe_mfma(double* out, int iters)
{
  double4_t acc[4];
  for(int a = 0; a < 4; a++) acc[a] = (double4_t){0.0, 0.0, 0.0, 0.0};
  const double x = 1.0 + threadIdx.x*1e-9, y = 1.0 - threadIdx.x*1e-9;
  // (a workgroup in the middle of the grid: the shader clock's count against the constant 100 MHz counter over its loop
  // = the clock the matrix cores actually ran at while the whole chip did nothing but fp64 MFMAs)
  const bool stamp = blockIdx.x == gridDim.x/2 && threadIdx.x == 0;
  long long c0 = 0, w0 = 0;
  if(stamp) { c0 = clock64(); w0 = wall_clock64(); }
  for(int it = 0; it < iters; it++)
  {
#pragma unroll
    for(int a = 0; a < 4; a++) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[a], 0, 0, 0);
  }
  double s = 0;
  for(int a = 0; a < 4; a++) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
  if(stamp) { asm volatile("" :: "v"(s)); out[1] = (double)(clock64() - c0); out[2] = (double)(wall_clock64() - w0); }
  if(s == 12345.678) out[0] = s;
}
__global__ void __launch_bounds__(TPB) k_probe_copy(const double2* __restrict__ in,
                                                    double2* __restrict__ out, size_t n)
{
  for(size_t i = (size_t)blockIdx.x*TPB + threadIdx.x; i < n; i += (size_t)gridDim.x*TPB) out[i] = in[i];
}

// number of K-splits for a tile count / K: about 1536 workgroups, and -- two workgroups of the SYRK kernel are resident on a
// CU -- a number of them that fills its LAST round of 2 * #CUs too.  Config #2: 136 tiles x 12 splits = 1632 workgroups are
// 3.19 rounds of 512, i.e. four rounds with the last one a fifth full; measured (profiles/r04_experiments.md) 4.46 ms with
// 12 splits, 4.08 with 11 (2.92 rounds), 4.09 with 15 (3.98), 4.67 with 13.  Among the split counts from two thirds of
// the nominal one to 1.5 times it: the smallest whose rounds are filled to within 3 % of the best.
static int syrk_nsplit(int ntiles, int K, int resident = 512)
{
  int nom = dlg_cdiv(1536, ntiles);
  const int maxs = K/(4*KC);
  if(nom > maxs) nom = maxs;
  if(nom < 1) nom = 1;
  const int lo = std::max(1, 2*nom/3), hi = std::max(lo, std::min(std::max(maxs, 1), nom + nom/2 + 1));
  auto eff = [&](int ns) { const double r = (double)ntiles*ns/resident; return r/std::ceil(r); };
  double beff = 0.0;
  for(int ns = lo; ns <= hi; ns++) beff = std::max(beff, eff(ns));
  for(int ns = lo; ns <= hi; ns++) if(eff(ns) >= beff - 0.03) return ns;
  return nom;
}

// overwrite=false: C += alpha*A'A-style update, accumulated in place (one split).
// overwrite=true : C  = alpha*update + lambda*I through the split-K slabs in `ws`.
template <int WT>
int launch_syrk(hipStream_t st, double* C, int ldc, const double* A, int lda, int n, int K,
                double alpha, double lambda, bool overwrite, double* ws, size_t ws_bytes,
                dlg_backend* prof = nullptr)
{
  constexpr int BT = 2*WT;
  const int T = dlg_cdiv(n, BT);
  const int ntiles = T*(T+1)/2;
  const size_t lds = (size_t)4*KC*(BT + 16)*sizeof(double);
  static bool attr_set = false;            // per instantiation
  if(!attr_set)
  {
    DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_syrk_lower<WT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  if(!overwrite)
  {
    const int kper = dlg_cdiv(K, KC)*KC;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_syrk_lower<WT>), dim3(ntiles), dim3(TPB), lds, st, C, ldc, A,
                       lda, n, K, alpha, 1, kper, (double*)nullptr);
    DLG_LAUNCH_CHECK();
    return DLG_OK;
  }
  int ks = syrk_nsplit(ntiles, K);
  if(ks < 2) ks = 2;                       // the slab path needs nsplit > 1 (an empty split is fine)
  const size_t per = (size_t)ntiles*BT*BT*sizeof(double);
  while(ks > 2 && per*ks > ws_bytes) ks--;
  if(!ws || per*ks > ws_bytes) { dlg_set_error("syrk workspace too small"); return DLG_ERR_ARG; }
  int kper = dlg_cdiv(dlg_cdiv(K, ks), KC)*KC;
  if(kper < KC) kper = KC;
  {
    hipEvent_t pe = (prof && (prof->prof_mask >> DLG_PROF_K4_KERNEL & 1u)) ? dlg_prof_begin(prof) : nullptr;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_syrk_lower<WT>), dim3(ntiles*ks), dim3(TPB), lds, st, C, ldc, A,
                       lda, n, K, alpha, ks, kper, ws);
    if(pe) dlg_prof_end(prof, DLG_PROF_K4_KERNEL, pe);
  }
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_syrk_reduce<BT>), dim3(ntiles*SYRK_RSUB), dim3(TPB), 0, st, C, ldc, n, ws,
                     ks, ntiles, alpha, 0.0, lambda);
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}

// blocked right-looking Cholesky, one launch chain per 64 columns (the step form; the default is the one-launch
// k_potrf_tiles, dense_launch_potrf_tiles): diagonal block + triangular solve (one launch where `fuse`), then the
// trailing update on the matrix cores.  (A look-ahead variant over two streams -- the next panel's columns first, the
// rest of the trailing update beside the next diagonal block -- was measured slower, 2.23 ms against 2.15 on config #2:
// three stream dependencies per step cost more than the overlapped SYRK saves; removed in round 4.)
int potrf_lower(hipStream_t st, double* A, int lda, int n, int* info_dev, double* Linv,
                int* flag = nullptr, int* epoch = nullptr, bool fuse = false, const DlgHandoff* ho = nullptr)
{
  for(int kb = 0, blk = 0; kb < n; kb += NB, blk++)
  {
    const int nb = (n - kb < NB) ? n - kb : NB;
    double* Li = Linv + (size_t)blk*NB*NB;
    const int rem = n - kb - nb;
    const bool fused = flag && epoch && ho && rem > 0 && fuse;
    if(fused) dense_launch_potrf_diag_trsm(st, A, lda, kb, nb, n, info_dev, Li, flag, ++*epoch, *ho);
    else dense_launch_potrf_diag(st, A, lda, kb, nb, info_dev, Li);
    if(rem > 0)
    {
      if(!fused) hipLaunchKernelGGL(k_trsm_gemm, dim3(dlg_cdiv(rem, NB)), dim3(TPB), 0, st, A, lda, kb, nb, n, Li);
      // trailing: C = A[kb+nb:, kb+nb:], panel P[i,k] = A[(kb+k)*lda + kb+nb+i]
      double* Cc = A + (size_t)(kb + nb)*lda + (kb + nb);
      const double* P = A + (size_t)kb*lda + (kb + nb);
      int rc;
      if(rem >= 1024) rc = launch_syrk<64>(st, Cc, lda, P, lda, rem, nb, -1.0, 0.0, false, nullptr, 0);
      else            rc = launch_syrk<32>(st, Cc, lda, P, lda, rem, nb, -1.0, 0.0, false, nullptr, 0);
      if(rc != DLG_OK) return rc;
    }
  }
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}

} // namespace

// =========================================================== host entry =====
int dense_create(dlg_backend* b)
{
  const size_t N = (size_t)b->N;
  DLG_HIP(hipMalloc(&b->G, N*N*sizeof(double)));
  DLG_HIP(hipMemsetAsync(b->G, 0, N*N*sizeof(double), b->stream));
  if(b->type == DLG_DENSE)
  {
    const int T = dlg_cdiv(b->N, 128);
    const size_t ntiles = (size_t)T*(T+1)/2;
    size_t ns = (size_t)syrk_nsplit((int)ntiles, b->M);
    if(ns < 2) ns = 2;
    b->slabs_bytes = ns*ntiles*128*128*sizeof(double);
    DLG_HIP(hipMalloc(&b->slabs, b->slabs_bytes));
  }
  // the pivot word rides in the last slot of the scalar block (as the sparse one does): whatever fetches the step's
  // scalars brings it along -- no copy of its own on the critical stream between the factorisation and the solve
  b->d_info = reinterpret_cast<int*>(b->d_scal + (dlg_backend::NSCAL - 1));
  b->h_info = reinterpret_cast<int*>(b->h_scal + (dlg_backend::NSCAL - 1));
  DLG_HIP(hipMalloc(&b->Linv, sizeof(double)*(size_t)dlg_cdiv(b->N, NB)*NB*NB));
  DLG_HIP(hipMemset(b->Linv, 0, sizeof(double)*(size_t)dlg_cdiv(b->N, NB)*NB*NB));      // (k_potrf_tiles stores the lower triangles only)
  return DLG_OK;
}
void dense_destroy(dlg_backend* b)
{
  if(b->G) (void)hipFree(b->G);
  if(b->Linv) (void)hipFree(b->Linv);
  b->Linv = nullptr;
  if(b->slabs) (void)hipFree(b->slabs);
  if(b->potrf_flag) { (void)hipFree(b->potrf_flag); b->potrf_flag = nullptr; }
  if(b->trsv_flag) { (void)hipFree(b->trsv_flag); b->trsv_flag = nullptr; }
  if(b->trsv_y) { (void)hipFree(b->trsv_y); b->trsv_y = nullptr; }
  if(b->trsv_x) { (void)hipFree(b->trsv_x); b->trsv_x = nullptr; }
  b->G = b->slabs = nullptr; b->d_info = nullptr; b->h_info = nullptr;
}

// K1: Jt_x into slot.Jt_x; d_scal[0] = norm2_x, d_scal[1] = (unused), d_scal[2..3] = norm2/absmax of Jt_x
int dense_eval(dlg_backend* b, int s)
{
  DlgSlot& S = b->slot[s];
  const int M = dlg_mloc(b), N = b->N;
  if(M == 0) { DLG_HIP(hipMemsetAsync(S.Jt_x, 0, sizeof(double)*(size_t)N, b->stream)); return DLG_OK; }
  int rpc = 128;
  while(rpc > 16 && (long)dlg_cdiv(M, rpc)*dlg_cdiv(N, TPB) < 2048) rpc >>= 1;
  const int nchunks = dlg_cdiv(M, rpc);
  const int nsl = (nchunks >= 64) ? 32 : 0;      // long lists: column sums in two stages
  DLG_CHECK(dlg_ensure_partials(b, (size_t)(nchunks + nsl)*N + 8192));
  double* part = b->d_part + 8192;         // first 8192 doubles are used by the vec reductions
  hipLaunchKernelGGL(k_gemvT_part, dim3(dlg_cdiv(N, TPB), nchunks), dim3(TPB), 0, b->stream, S.Jin(),
                     S.xin(), M, N, rpc, part);
  if(nsl > 0)
  {
    double* part2 = part + (size_t)nchunks*N;
    hipLaunchKernelGGL(k_colsum_slices, dim3(dlg_cdiv(N, TPB), nsl), dim3(TPB), 0, b->stream, part, nchunks, N, part2);
    hipLaunchKernelGGL(k_colsum, dim3(dlg_cdiv(N, TPB)), dim3(TPB), 0, b->stream, part2, nsl, N, S.Jt_x);
  }
  else
    hipLaunchKernelGGL(k_colsum, dim3(dlg_cdiv(N, TPB)), dim3(TPB), 0, b->stream, part, nchunks, N,
                       S.Jt_x);
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}

int dense_norm2_Jv(dlg_backend* b, int s, const double* v, double* out_dev)
{
  DlgSlot& S = b->slot[s];
  const int M = dlg_mloc(b);
  int g = dlg_cdiv(M, 4); if(g > 2048) g = 2048; if(g < 1) g = 1;
  // K8 behind the decision point (dlg_take_step, dlg_backend_set_defer_tail): the partial sums go to page-locked memory,
  // dlg_step_tail adds them; p_new rides along
  if(b->tail_mode)
    if(double* hp = dlg_tail_partials(b, g))
    {
      hipLaunchKernelGGL(k_norm2_Jv_part, dim3(g), dim3(TPB), 0, b->stream, S.Jin(), v, M, b->N, hp,
                         b->fold_p_src, b->fold_p_dst, (int)b->N, b->k8_skip);
      DLG_LAUNCH_CHECK();
      b->p_copied = b->fold_p_src != nullptr;
      return DLG_OK;
    }
  DLG_CHECK(dlg_ensure_partials(b, 8192));
  double* part = b->d_part + 5120;          // behind the regions of the vector reductions (kernels_vec.hip)
  hipLaunchKernelGGL(k_norm2_Jv_part, dim3(g), dim3(TPB), 0, b->stream, S.Jin(), v, M, b->N, part,
                     (const double*)nullptr, (double*)nullptr, 0, b->k8_skip);
  DLG_LAUNCH_CHECK();
  return k_reduce_sum(b, part, g, out_dev);
}
int dense_norm2_chunks(const dlg_backend* b) { int g = dlg_cdiv(dlg_mloc(b), 4); if(g > 2048) g = 2048; if(g < 1) g = 1; return g; }

int products_quadform(dlg_backend* b, int s, const double* v, double* out_dev)
{
  DlgSlot& S = b->slot[s];
  const bool packed = b->flags & DLG_FLAG_JTJ_PACKED, upper = b->flags & DLG_FLAG_JTJ_UPPER;
  if(packed && !upper)
  { dlg_set_error("only JtJ unpacked || (packed,upper) is supported (reference dogleg.c:597-601)"); return DLG_ERR_ARG; }
  int g = dlg_cdiv(b->N, 4); if(g > 2048) g = 2048; if(g < 1) g = 1;
  DLG_CHECK(dlg_ensure_partials(b, 8192));
  double* part = b->d_part + 5120;          // behind the regions of the vector reductions (kernels_vec.hip)
  hipLaunchKernelGGL(k_quadform_part, dim3(g), dim3(TPB), 0, b->stream, S.Jin(), v, b->N,
                     packed ? 1 : 0, part);
  DLG_LAUNCH_CHECK();
  return k_reduce_sum(b, part, g, out_dev);
}

// K5-dense on b->G: the one-launch form, or the step-by-step one
static int run_potrf(dlg_backend* b)
{
  const int T = dlg_cdiv(b->N, NB);
  if(!b->potrf_flag)
  {
    DLG_HIP(hipMalloc(&b->potrf_flag, sizeof(int)*((size_t)T*T + 2)));      // (+ the step form's flag, + the word for the second stream)
    DLG_HIP(hipMemsetAsync(b->potrf_flag, 0, sizeof(int)*((size_t)T*T + 2), b->stream));
  }
  // one launch for the whole factorisation (dense_diag.hip: k_potrf_tiles); DOGLEG_AMD_POTRF_STEPS: the
  // step-by-step form (its fused diagonal + rows launch uses the last flag)
  const DlgHandoff ho = dlg_handoff(b, 1 << 22);
  if(!b->knobs.potrf_steps && T >= 2)
  {
    DlgRegionTurn turn(b);
    int* gate = b->potrf_flag + (size_t)T*T + 1;
    dense_launch_potrf_tiles(b->stream, b->G, b->N, b->N, b->d_info, b->Linv, b->potrf_flag, ++b->potrf_epoch, ho, gate);
    DLG_LAUNCH_CHECK();
    dlg_fork_gate(b, gate, b->potrf_epoch);
    return DLG_OK;
  }
  return potrf_lower(b->stream, b->G, b->N, b->N, b->d_info, b->Linv, b->potrf_flag + (size_t)T*T, &b->potrf_epoch,
                     true, &ho);
}

static int finish_potrf(dlg_backend* b, int* ok)
{
  if(b->defer_factor_sync) { *ok = 1; return DLG_OK; }        // the caller reads dense_factor_ok() behind its fetch of the scalars
  DLG_HIP(hipMemcpyAsync(b->h_info, b->d_info, sizeof(int), hipMemcpyDeviceToHost, b->stream));
  DLG_HIP(hipStreamSynchronize(b->stream));
  *ok = (*b->h_info == 0);
  return DLG_OK;
}

// (the one-launch factorisation clears the pivot word itself -- its first workgroup, in front of every pivot: no fill
// kernel in front of the SYRK)
static bool potrf_clears_info(const dlg_backend* b) { return !b->knobs.potrf_steps && dlg_cdiv(b->N, NB) >= 2; }
int dense_factorize(dlg_backend* b, int s, double lambda, int* ok)
{
  DlgSlot& S = b->slot[s];
  if(!potrf_clears_info(b)) DLG_HIP(hipMemsetAsync(b->d_info, 0, sizeof(int), b->stream));
  // K4: G(lower) = J^T J + lambda I   (K = M measurement rows, A = J with lda = N)
  const bool sharded = b->sharded();
  {
    DlgProfScope pt(b, DLG_PROF_K4_TOTAL);
    DLG_CHECK(launch_syrk<64>(b->stream, b->G, b->N, S.Jin(), b->N, b->N, dlg_mloc(b), 1.0,
                              sharded ? 0.0 : lambda, true, b->slabs, b->slabs_bytes, b));
  }
  if(sharded)
  {
    DLG_CHECK(dlg_allreduce_dev(b, b->G, (size_t)b->N*b->N));
    if(lambda != 0.0)
      hipLaunchKernelGGL(k_add_diag, dim3(dlg_cdiv(b->N, TPB)), dim3(TPB), 0, b->stream, b->G, b->N, lambda);
  }
  // K5: independent work may run beside it from here on (the one-launch form raises a word for the second stream
  // itself: no event between the SYRK's reduce and the factorisation; the chain of small kernels: an event)
  if(!potrf_clears_info(b)) dlg_fork_point(b);
  {
    DlgProfScope pf(b, DLG_PROF_K5_FACTOR);
    DLG_CHECK(run_potrf(b));
  }
  return finish_potrf(b, ok);
}

int products_factorize(dlg_backend* b, int s, double lambda, int* ok)
{
  DlgSlot& S = b->slot[s];
  const bool packed = b->flags & DLG_FLAG_JTJ_PACKED, upper = b->flags & DLG_FLAG_JTJ_UPPER;
  if(packed && !upper)
  { dlg_set_error("packed-lower JtJ is not supported (reference dogleg.c:597-601)"); return DLG_ERR_ARG; }
  if(!potrf_clears_info(b)) DLG_HIP(hipMemsetAsync(b->d_info, 0, sizeof(int), b->stream));
  const size_t nn = (size_t)b->N*b->N;
  hipLaunchKernelGGL(k_unpack_to_G, dim3(dlg_cdiv((long)nn, TPB)), dim3(TPB), 0, b->stream, S.Jin(),
                     b->N, packed ? 1 : 0, lambda, b->G);
  DLG_LAUNCH_CHECK();
  DLG_CHECK(run_potrf(b));
  return finish_potrf(b, ok);
}

// out = (L L^T)^-1 rhs
int dense_solve(dlg_backend* b, const double* rhs, double* out)
{
  const int n = b->N;
  {
    // one launch for both sweeps where every block row gets a CU of its own (dense_diag.hip: k_trsv_tiles);
    // DOGLEG_AMD_TRSV_STEPS: a launch per block column and sweep
    const int T = dlg_cdiv(n, NB);
    // (every block row needs a CU of its own at the same time -- the sweep back waits for higher-numbered
    // workgroups --: half the chip at most, so that the pass over J on the second stream does not matter)
    if(T >= 2 && T <= b->ncu/2 && !b->knobs.trsv_steps)
    {
      if(!b->trsv_y)
      {
        // two sets of hand-off buffers (even / odd launches), every element a sentinel until its owner stores it
        const size_t ne = 3*(size_t)T*NB;         // (+ a set that stays empty: the forced time-out of the tests)
        DLG_HIP(hipMalloc(&b->trsv_y, sizeof(double)*ne));
        DLG_HIP(hipMalloc(&b->trsv_x, sizeof(double)*ne));
        dense_trsv_arm(b->stream, b->trsv_y, b->trsv_x, ne);
      }
      DlgRegionTurn turn(b);
      dense_launch_trsv_tiles(b->stream, b->G, n, n, b->Linv, rhs, b->trsv_y, out, b->trsv_x, ++b->trsv_epoch, dlg_handoff(b, 1 << 22));
      DLG_LAUNCH_CHECK();
      return DLG_OK;
    }
  }
  if(out != rhs) DLG_HIP(hipMemcpyAsync(out, rhs, sizeof(double)*(size_t)n, hipMemcpyDeviceToDevice, b->stream));
  // forward: the first diagonal block, then one fused launch per block column (update + next diagonal block)
  hipLaunchKernelGGL(k_trsv_diag_fwd, dim3(1), dim3(64), 0, b->stream, b->Linv, 0, (n < NB) ? n : NB, out);
  for(int kb = 0, blk = 0; kb < n; kb += NB, blk++)
  {
    const int nb = (n - kb < NB) ? n - kb : NB;
    const int rem = n - kb - nb;
    if(rem > 0)
      hipLaunchKernelGGL(k_trsv_step_fwd, dim3(dlg_cdiv(rem, TPB)), dim3(TPB), 0, b->stream, b->G, n,
                         kb, nb, n, b->Linv + (size_t)(blk + 1)*NB*NB, (rem < NB) ? rem : NB, out);
  }
  const int nblk = dlg_cdiv(n, NB);
  // backward: the last diagonal block, then one fused launch per block column (update + previous diagonal block)
  {
    const int kb = (nblk - 1)*NB;
    hipLaunchKernelGGL(k_trsv_diag_bwd, dim3(1), dim3(64), 0, b->stream, b->Linv + (size_t)(nblk - 1)*NB*NB, kb, n - kb, out);
  }
  for(int blk = nblk - 1; blk >= 1; blk--)
  {
    const int kb = blk*NB;
    const int nb = (n - kb < NB) ? n - kb : NB;
    hipLaunchKernelGGL(k_trsv_step_bwd, dim3(1 + dlg_cdiv(kb - NB, 4)), dim3(TPB), 0, b->stream, b->G, n, kb, nb,
                       b->Linv + (size_t)(blk - 1)*NB*NB, out);
  }
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}

// (L L^T) X = B for 16 interleaved right-hand sides [n][16], in place
int dense_solve_multi(dlg_backend* b, double* d_il)
{
  const int n = b->N;
  hipStream_t st = b->stream;
  for(int kb = 0, blk = 0; kb < n; kb += NB, blk++)
  {
    const int nb = (n - kb < NB) ? n - kb : NB;
    hipLaunchKernelGGL(k_trsm_diag_m, dim3(1), dim3(TPB), 0, st, b->Linv + (size_t)blk*NB*NB, kb, nb, d_il, 0);
    const int rem = n - kb - nb;
    if(rem > 0)
      hipLaunchKernelGGL(k_trsm_update_fwd_m, dim3(dlg_cdiv(rem, 16*(TPB/64))), dim3(TPB), 0, st, b->G, n, kb, nb, n, d_il);
  }
  const int nblk = dlg_cdiv(n, NB);
  for(int blk = nblk - 1; blk >= 0; blk--)
  {
    const int kb = blk*NB;
    const int nb = (n - kb < NB) ? n - kb : NB;
    hipLaunchKernelGGL(k_trsm_diag_m, dim3(1), dim3(TPB), 0, st, b->Linv + (size_t)blk*NB*NB, kb, nb, d_il, 1);
    if(kb > 0)
      hipLaunchKernelGGL(k_trsm_update_bwd_m, dim3(dlg_cdiv(kb, 16*(TPB/64))), dim3(TPB), 0, st, b->G, n, kb, nb, d_il);
  }
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}
int dense_jt_chunk_interleaved(dlg_backend* b, int s, int row0, int ncols, double* d_il)
{
  hipLaunchKernelGGL(k_jt_chunk_dense, dim3(dlg_cdiv((long)b->N*DMR, TPB)), dim3(TPB), 0, b->stream, b->slot[s].Jin(), b->N,
                     row0, ncols, d_il);
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}

// ------------------------------------------------- stand-alone C-ABI entries
extern "C" int dlg_kernel_syrk_lower(void* hip_stream, double* C_dev, int ldc, const double* A_dev,
                                     int lda, int n, int K, double alpha, double* ws,
                                     size_t ws_bytes)
{
  hipStream_t st = (hipStream_t)hip_stream;
  const bool overwrite = (ws != nullptr);
  if(n >= 1024) return launch_syrk<64>(st, C_dev, ldc, A_dev, lda, n, K, alpha, 0.0, overwrite, ws, ws_bytes);
  return launch_syrk<32>(st, C_dev, ldc, A_dev, lda, n, K, alpha, 0.0, overwrite, ws, ws_bytes);
}
extern "C" int dlg_kernel_potrf_lower(void* hip_stream, double* A_dev, int lda, int n, int* info_dev)
{
  double* Linv = nullptr;
  DLG_HIP(hipMalloc(&Linv, sizeof(double)*(size_t)dlg_cdiv(n, NB)*NB*NB));
  const int rc = potrf_lower((hipStream_t)hip_stream, A_dev, lda, n, info_dev, Linv);
  (void)hipStreamSynchronize((hipStream_t)hip_stream);
  (void)hipFree(Linv);
  return rc;
}
extern "C" int dlg_factor_download_dense(dlg_backend_t* b, double* host, size_t n)
{
  if(!b || b->type == DLG_SPARSE || !b->G) { dlg_set_error("no dense factor"); return DLG_ERR_STATE; }
  const size_t N = (size_t)b->N;
  const bool full = (b->type == DLG_DENSE_PRODUCTS) && !(b->flags & DLG_FLAG_JTJ_PACKED);
  const size_t need = full ? N*N : N*(N+1)/2;
  if(n < need) { dlg_set_error("factor buffer too small"); return DLG_ERR_ARG; }
  if(full)
  {
    DLG_HIP(hipMemcpyAsync(host, b->G, need*sizeof(double), hipMemcpyDeviceToHost, b->stream));
    DLG_HIP(hipStreamSynchronize(b->stream));
    return DLG_OK;
  }
  double* tmp = nullptr;
  DLG_HIP(hipMalloc(&tmp, need*sizeof(double)));
  hipLaunchKernelGGL(k_pack_from_G, dim3(dlg_cdiv((long)(N*N), TPB)), dim3(TPB), 0, b->stream, b->G,
                     b->N, tmp);
  hipError_t e = hipMemcpyAsync(host, tmp, need*sizeof(double), hipMemcpyDeviceToHost, b->stream);
  if(e == hipSuccess) e = hipStreamSynchronize(b->stream);
  (void)hipFree(tmp);
  if(e != hipSuccess) { dlg_set_error("factor download: %s", hipGetErrorString(e)); return DLG_ERR_HIP; }
  return DLG_OK;
}

// tflops: sustained fp64 MFMA rate of the whole chip, ONE resident round of `waves_per_simd` workgroups of 4 waves on
// every CU, four independent accumulators a wave; clock3 (may be NULL): {shader clock in MHz during the loop, clocks per
// MFMA and wave, clocks per MFMA and SIMD}.  (Rounds 1-3 read 48 TFLOP/s here: the probe kernel was compiled without a
// bound on its registers -- its accumulators travelled between AGPRs and VGPRs in every iteration -- and its grid
// of 2048 workgroups ran a second, partly filled round behind the first.  profiles/r04_probe.txt.)
extern "C" int dlg_probe_mfma_f64_waves(int waves_per_simd, double* tflops, double* clock3)
{
  if(waves_per_simd < 1 || waves_per_simd > 8 || !tflops) return DLG_ERR_ARG;
  double* d = nullptr;
  DLG_HIP(hipMalloc(&d, 32));
  int ncu = 256;
  { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev); if(ncu <= 0) ncu = 256; }
  const int iters = 20000, blocks = ncu*waves_per_simd;
  hipEvent_t e0, e1;
  DLG_HIP(hipEventCreate(&e0)); DLG_HIP(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_probe_mfma, dim3(blocks), dim3(TPB), 0, 0, d, 100);
  DLG_HIP(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k_probe_mfma, dim3(blocks), dim3(TPB), 0, 0, d, iters);
  DLG_HIP(hipEventRecord(e1, 0));
  DLG_HIP(hipEventSynchronize(e1));
  float ms = 0; DLG_HIP(hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)blocks*4 /*waves*/ * iters * 4 /*mfma*/ * 2048.0;
  *tflops = flops/(ms*1e-3)/1e12;
  if(clock3)
  {
    double h[4] = {0, 0, 0, 0};
    DLG_HIP(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
    const double cyc = h[1], ticks = h[2];                        // ticks: 100 MHz
    clock3[0] = ticks > 0 ? cyc/(ticks/100.0) : 0.0;             // MHz
    clock3[1] = cyc/((double)iters*4.0);                          // one wave issues iters * 4 MFMAs
    clock3[2] = clock3[1]/(double)waves_per_simd;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(d);
  return DLG_OK;
}
extern "C" int dlg_probe_mfma_f64_clock(double* tflops, double* clock3) { return dlg_probe_mfma_f64_waves(2, tflops, clock3); }
extern "C" int dlg_probe_mfma_f64(double* tflops) { return dlg_probe_mfma_f64_clock(tflops, nullptr); }
extern "C" int dlg_probe_hbm_copy(double* gbs)
{
  const size_t bytes = (size_t)1 << 30;
  double2 *a = nullptr, *c = nullptr;
  DLG_HIP(hipMalloc(&a, bytes)); DLG_HIP(hipMalloc(&c, bytes));
  DLG_HIP(hipMemset(a, 1, bytes));
  hipEvent_t e0, e1;
  DLG_HIP(hipEventCreate(&e0)); DLG_HIP(hipEventCreate(&e1));
  const size_t n = bytes/sizeof(double2);
  hipLaunchKernelGGL(k_probe_copy, dim3(2048), dim3(TPB), 0, 0, a, c, n);
  DLG_HIP(hipEventRecord(e0, 0));
  for(int i = 0; i < 5; i++) hipLaunchKernelGGL(k_probe_copy, dim3(2048), dim3(TPB), 0, 0, a, c, n);
  DLG_HIP(hipEventRecord(e1, 0));
  DLG_HIP(hipEventSynchronize(e1));
  float ms = 0; DLG_HIP(hipEventElapsedTime(&ms, e0, e1));
  *gbs = 5.0*2.0*(double)bytes/(ms*1e-3)/1e9;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(a); (void)hipFree(c);
  return DLG_OK;
}

bool dense_factor_ok(const dlg_backend* b) { return *b->h_info == 0; }
