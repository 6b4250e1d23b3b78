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
  panel_factor_b16<TPB>(P, LD, 2*NB, NB, t, &sbad, 0);
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
                                                         int* flag, int epoch, DlgHandoff ho)
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
    panel_factor_b16<TPB>(P, LD, 2*NB, NB, t, &sbad, 0);
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
    while(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch + ho.skew)
    { __builtin_amdgcn_s_sleep(1); if(++spins > ho.spins) { atomicOr(ho.status, DLG_HANDOFF_TRSM); break; } }      // report, never hang
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

// ---- the whole factorisation in ONE launch: a workgroup per 64 x 64 tile of the lower triangle ------
// Left-looking, owner computes: the owner of tile (i, j) keeps it in MFMA accumulators and subtracts
// L(i,k) L(j,k)' for k = 0 .. j-1 as those blocks of L appear (one flag per block, carrying the epoch
// of the launch); then the owner of a diagonal tile factors it (panel_factor_mfma on [A; I]: L and
// its inverse) and publishes the inverse, the owner of an off-diagonal tile multiplies with that
// inverse and publishes its block of L.  Workgroups are numbered column by column, the diagonal tile
// first: a workgroup only waits for lower-numbered ones, so in-order dispatch cannot deadlock.  The
// step-by-step form pays three launches per 64 columns (96 for N = 2000) and every one of them waits
// for the one before; here the critical path is factor -> one multiply -> one update per 64 columns.
// Hand-offs as in the sparse one-launch regions: write-through stores, drained, then the flag;
// consumers poll and read around L1; one workgroup per CU.
typedef double dd_v4d __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(TPB) k_potrf_tiles(double* A, int lda, int n, int T, int* __restrict__ info,
                                                     double* Linv, int* flags, int epoch, DlgHandoff ho, int self_x)
{
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ int sbad;
  typedef __attribute__((address_space(1))) double* gd_t;
  typedef const __attribute__((address_space(1))) double* gcd_t;
  constexpr int LDT = NB + 1;
  double (*Li)[LDT] = reinterpret_cast<double (*)[LDT]>(sm);                // [row][k]
  double (*Lj)[LDT] = reinterpret_cast<double (*)[LDT]>(sm + NB*LDT);
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int jn = lane & 15, kq = lane >> 4;
  // tile of this workgroup: column by column, the diagonal tile of a column first.  self_x: the diagonal tile of the NEXT
  // column comes right behind this column's diagonal tile, in front of its left neighbour -- the neighbour overwrites
  // tile (j, j-1) with L(j, j-1) in place and may only do so once the diagonal owner (j, j) has read the original, and a
  // workgroup only ever waits for lower-numbered ones: (0,0) | (1,1) (1,0) (2,0) ... | (2,2) (2,1) (3,1) ... | ...
  int ti, tj;
  if(self_x)
  {
    if(blockIdx.x == 0) { ti = tj = 0; }
    else
    {
      int rem = (int)blockIdx.x - 1; tj = 0;
      while(rem >= T - tj) { rem -= T - tj; tj++; }
      if(rem == 0) { ti = tj = tj + 1; } else ti = tj + rem;
    }
  }
  else
  {
    int rem = blockIdx.x; tj = 0;
    while(rem >= T - tj) { rem -= T - tj; tj++; }
    ti = tj + rem;
  }
  const int row0 = NB*ti, col0 = NB*tj;
  // the tile in accumulators: wave wv holds rows 16 wv .. 16 wv + 15, four 16-column pieces;
  // lane (jn, kq): column jn of a piece, rows kq + 4 r
  dd_v4d acc[4];
#pragma unroll
  for(int ct = 0; ct < 4; ct++)
#pragma unroll
    for(int r = 0; r < 4; r++)
    {
      const int row = row0 + 16*wv + kq + 4*r, col = col0 + 16*ct + jn;
      double v = (row == col) ? 1.0 : 0.0;                      // identity padding past the end
      if(row < n && col < n) v = (row >= col) ? A[(size_t)col*lda + row] : 0.0;
      acc[ct][r] = v;
    }
  auto wait_flag = [&](int fi, int fj) {
    int spins = 0;
    while(__hip_atomic_load(flags + fi*T + fj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch + ho.skew)
    { __builtin_amdgcn_s_sleep(1); if(++spins > ho.spins) { atomicOr(ho.status, DLG_HANDOFF_POTRF); break; } }     // its own status word: not a pivot
  };
  // a published 64 x 64 block of L (rows r0.., columns c0..) into LDS, read around L1; past the end: zeros
  auto stage = [&](double (*D)[LDT], int r0, int c0) {
    for(int e = t; e < NB*NB; e += TPB)
    {
      const int i = e % NB, k = e / NB;
      const int row = r0 + i, col = c0 + k;
      D[i][k] = (row < n && col < n) ? __hip_atomic_load((gcd_t)(A + (size_t)col*lda + row), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    }
  };
  // self_x: the owner of a diagonal tile (j, j) also keeps the tile to its left, (j, j - 1), up to date and forms
  // L(j, j-1) = tile * inv(L(j-1, j-1))' ITSELF once that inverse is published -- what lies between the factorisation of
  // one diagonal tile and the next is then inverse -> one product -> one update in ONE workgroup, not the neighbour's
  // product, its stores, their drain, its flag, this workgroup's poll and a 32 KB load (the neighbour still publishes the
  // block for the tiles below)
  const bool selfx = self_x && ti == tj && tj > 0;
  dd_v4d acc2[4];
  if(selfx)
  {
#pragma unroll
    for(int ct = 0; ct < 4; ct++)
#pragma unroll
      for(int r = 0; r < 4; r++)
      {
        const int row = row0 + 16*wv + kq + 4*r, col = col0 - NB + 16*ct + jn;
        acc2[ct][r] = (row < n) ? A[(size_t)col*lda + row] : 0.0;
      }
    // (the original tile is in registers: its owner may overwrite it -- the word in the unused upper triangle of the flags)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if(t == 0) __hip_atomic_store(flags + (tj - 1)*T + tj, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  for(int k = 0; k < (selfx ? tj - 1 : tj); k++)
  {
    if(t == 0) wait_flag(ti, k);
    if(t == 64 && ti != tj) wait_flag(tj, k);
    if(t == 64 && selfx) wait_flag(tj - 1, k);
    __syncthreads();
    stage(Li, row0, NB*k);
    if(ti != tj) stage(Lj, col0, NB*k);
    if(selfx) stage(Lj, col0 - NB, NB*k);
    __syncthreads();
    double (*Lb)[LDT] = (ti != tj) ? Lj : Li;
#pragma unroll 4
    for(int kk = 0; kk < NB; kk += 4)
    {
      const double a = Li[16*wv + jn][kk + kq];
#pragma unroll
      for(int ct = 0; ct < 4; ct++)
        acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, -Lb[16*ct + jn][kk + kq], acc[ct], 0, 0, 0);
      if(selfx)
      {
#pragma unroll
        for(int ct = 0; ct < 4; ct++)
          acc2[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, -Lj[16*ct + jn][kk + kq], acc2[ct], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  if(selfx)
  {
    // L(j, j-1) from the tile kept here and the inverse of the diagonal block to the left, then this tile's last update
    if(t == 0) wait_flag(tj - 1, tj - 1);
    __syncthreads();
    {
      const double* Lv = Linv + (size_t)(tj - 1)*NB*NB;
      for(int e = t; e < NB*NB; e += TPB)
      {
        const int i = e % NB, k = e / NB;
        Lj[i][k] = __hip_atomic_load((gcd_t)(Lv + e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for(int ct = 0; ct < 4; ct++)
#pragma unroll
        for(int r = 0; r < 4; r++) Li[16*wv + kq + 4*r][16*ct + jn] = acc2[ct][r];
    }
    __syncthreads();
    dd_v4d x[4];
#pragma unroll
    for(int ct = 0; ct < 4; ct++) x[ct] = (dd_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for(int kk = 0; kk < NB; kk += 4)
    {
      const double a = Li[16*wv + jn][kk + kq];
#pragma unroll
      for(int ct = 0; ct < 4; ct++)
        x[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Lj[16*ct + jn][kk + kq], x[ct], 0, 0, 0);
    }
    __syncthreads();
#pragma unroll
    for(int ct = 0; ct < 4; ct++)
#pragma unroll
      for(int r = 0; r < 4; r++) Li[16*wv + kq + 4*r][16*ct + jn] = x[ct][r];
    __syncthreads();
#pragma unroll 4
    for(int kk = 0; kk < NB; kk += 4)
    {
      const double a = Li[16*wv + jn][kk + kq];
#pragma unroll
      for(int ct = 0; ct < 4; ct++)
        acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, -Li[16*ct + jn][kk + kq], acc[ct], 0, 0, 0);
    }
    __syncthreads();
  }
  if(ti == tj)
  {
    // the diagonal tile: [A; I] -> [L; L^-T] in one panel sweep (as k_potrf_diag_inv)
    double* P = sm;
    constexpr int LD = 2*NB;
#pragma unroll
    for(int ct = 0; ct < 4; ct++)
#pragma unroll
      for(int r = 0; r < 4; r++)
      {
        const int i = 16*wv + kq + 4*r, j = 16*ct + jn;
        P[i + j*LD] = (i >= j) ? acc[ct][r] : 0.0;
      }
    for(int e = t; e < NB*NB; e += TPB) { const int i = e % NB, j = e / NB; P[NB + i + j*LD] = (i == j) ? 1.0 : 0.0; }
    if(t == 0) sbad = 0x7fffffff;
    __syncthreads();
    panel_factor_b16<TPB>(P, LD, 2*NB, NB, t, &sbad, 0);
    const int nb = min(NB, n - col0);
    if(t == 0) { const int bad = sbad; if(bad < nb) atomicCAS(info, 0, col0 + bad + 1); }
    double* Lv = Linv + (size_t)tj*NB*NB;
    for(int e = t; e < NB*NB; e += TPB)
    {
      const int i = e % NB, j = e / NB;
      __hip_atomic_store((gd_t)(Lv + e), (i >= j) ? P[NB + j + i*LD] : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if(t == 0) __hip_atomic_store(flags + tj*T + tj, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for(int e = t; e < NB*NB; e += TPB)
    {
      const int i = e % NB, j = e / NB;
      if(i >= j && row0 + i < n && col0 + j < n) A[(size_t)(col0 + j)*lda + row0 + i] = P[i + j*LD];
    }
    return;
  }
  // an off-diagonal tile: X = tile * inv(L(j,j))'
  if(t == 0) wait_flag(tj, tj);
  __syncthreads();
  {
    const double* Lv = Linv + (size_t)tj*NB*NB;
    for(int e = t; e < NB*NB; e += TPB)
    {
      const int i = e % NB, k = e / NB;                                 // Linv(i, k), zero above the diagonal
      Lj[i][k] = __hip_atomic_load((gcd_t)(Lv + e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for(int ct = 0; ct < 4; ct++)
#pragma unroll
      for(int r = 0; r < 4; r++) Li[16*wv + kq + 4*r][16*ct + jn] = acc[ct][r];
  }
  __syncthreads();
  dd_v4d x[4];
#pragma unroll
  for(int ct = 0; ct < 4; ct++) x[ct] = (dd_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for(int kk = 0; kk < NB; kk += 4)
  {
    const double a = Li[16*wv + jn][kk + kq];
#pragma unroll
    for(int ct = 0; ct < 4; ct++)
      x[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Lj[16*ct + jn][kk + kq], x[ct], 0, 0, 0);
  }
  // (the diagonal owner of this block row keeps the original of this tile up to date itself, self_x: it must have read it)
  if(self_x && ti == tj + 1)
  {
    if(t == 0) wait_flag(tj, ti);
    __syncthreads();
  }
#pragma unroll
  for(int ct = 0; ct < 4; ct++)
#pragma unroll
    for(int r = 0; r < 4; r++)
    {
      const int row = row0 + 16*wv + kq + 4*r, col = col0 + 16*ct + jn;
      if(row < n && col < n) __hip_atomic_store((gd_t)(A + (size_t)col*lda + row), x[ct][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if(t == 0) __hip_atomic_store(flags + ti*T + tj, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- both triangular solves in ONE launch: a workgroup per 64 rows -------------------------------
// forward  y_i = Linv_i (b_i - sum_{k<i} L(i,k) y_k), backward  x_i = Linv_i' (y_i - sum_{k>i} L(k,i)' x_k).
// Workgroup i waits for y_k (k < i) going down and for x_k (k > i) coming back up.  The vectors ARE the signal: a
// block of y (x) is handed over in a buffer that holds a sentinel (a NaN no computation produces) until its owner
// stores the values -- write-through, 8 bytes a lane --, and the 64 lanes that need a block each poll their own
// element until it is not the sentinel: one trip through L2 per block instead of flag, barrier, load of the
// block (and no drain + barrier + flag store on the owner's side): 3.5 -> 2 us a block, 32 + 32 of them in a row
// on config #2.  Two sets of buffers, used by launches of even / odd epoch; a launch re-arms the set of the NEXT
// launch (the launch before this one, which used it, is over).  The tile of L a product needs does not depend on
// the vector it waits for: it is on its way (forward: in registers, backward: staged in LDS) before the poll.
// All T workgroups must be resident (they wait for higher-numbered ones on the way back): T <= #CUs / 2.
constexpr unsigned long long TRSV_EMPTY = 0x7FF8DEADBEEF0001ull;
__global__ void __launch_bounds__(TPB) k_trsv_tiles(const double* __restrict__ A, int lda, int n, int T,
                                                    const double* __restrict__ Linv, const double* __restrict__ rhs,
                                                    double* Yh, double* X, double* Xh, int epoch, DlgHandoff ho)
{
  extern __shared__ __attribute__((aligned(16))) double sm[];
  typedef __attribute__((address_space(1))) unsigned long long* gu_t;
  constexpr int LDT = NB + 1;
  double (*Lt)[LDT] = reinterpret_cast<double (*)[LDT]>(sm);                 // a tile of L: Lt[row][col]
  double* Li = sm + NB*LDT;                                                  // [NB*LDT] this block's inverse, Li[r + c*LDT] = Linv(r, c) (padded: its transpose is read too)
  double* v = Li + NB*LDT;                                                    // [NB]  the vector waited for
  double* part = v + NB;                                                     // [4][NB] partial sums
  const int t = threadIdx.x, r = t & 63, g = t >> 6;
  const int i = blockIdx.x, row0 = NB*i;
  const int npad = T*NB, par = epoch & 1;
  unsigned long long* ycur = reinterpret_cast<unsigned long long*>(Yh) + (size_t)par*npad;
  unsigned long long* xcur = reinterpret_cast<unsigned long long*>(Xh) + (size_t)par*npad;
  // (ho.skew != 0, the tests' forced time-out: the consumers look at a third set that nobody ever fills)
  unsigned long long* ytake = ho.skew ? reinterpret_cast<unsigned long long*>(Yh) + (size_t)2*npad : ycur;
  unsigned long long* xtake = ho.skew ? reinterpret_cast<unsigned long long*>(Xh) + (size_t)2*npad : xcur;
  // the next launch's set, this block's part
  if(t < NB)
  {
    reinterpret_cast<unsigned long long*>(Yh)[(size_t)(1 - (epoch & 1))*npad + row0 + t] = TRSV_EMPTY;
    reinterpret_cast<unsigned long long*>(Xh)[(size_t)(1 - (epoch & 1))*npad + row0 + t] = TRSV_EMPTY;
  }
  // (lanes 0..NB-1: element t of block k, once it is there)
  auto take = [&](unsigned long long* buf, int k) -> double {
    unsigned long long u; int spins = 0;
    while((u = __hip_atomic_load((gu_t)(buf + NB*k + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == TRSV_EMPTY)
    { __builtin_amdgcn_s_sleep(1); if(++spins > ho.spins) { atomicOr(ho.status, DLG_HANDOFF_TRSV); u = 0; break; } }
    return __longlong_as_double((long long)u);
  };
  for(int e = t; e < NB*NB; e += TPB) Li[(e & (NB - 1)) + (e/NB)*LDT] = Linv[(size_t)i*NB*NB + e];
  const double rhs_t = (t < NB && row0 + t < n) ? rhs[row0 + t] : 0.0;      // (fetched now: behind the last block of y it would sit on everybody's path)
  // ---- forward: thread (row r, column group g) keeps 16 values of the tile in registers
  double acc = 0.0;
  double cur[16];
  auto load_row_tile = [&](int k, double (&d)[16]) {
#pragma unroll
    for(int c = 0; c < 16; c++)
    {
      const int row = row0 + r, col = NB*k + 16*g + c;
      d[c] = (row < n && k < i) ? A[(size_t)col*lda + row] : 0.0;
    }
  };
  load_row_tile(0, cur);
  for(int k = 0; k < i; k++)
  {
    double nxt[16];
    load_row_tile(k + 1, nxt);                       // (k + 1 == i: zeros, no loads)
    __syncthreads();                                 // (v of the step before is done with)
    if(t < NB) v[t] = take(ytake, k);
    __syncthreads();
#pragma unroll
    for(int c = 0; c < 16; c++) acc += cur[c]*v[16*g + c];
#pragma unroll
    for(int c = 0; c < 16; c++) cur[c] = nxt[c];
  }
  __syncthreads();
  part[g*NB + r] = acc;
  __syncthreads();
  if(t < NB)
  {
    const double s = (part[t] + part[NB + t]) + (part[2*NB + t] + part[3*NB + t]);
    v[t] = (row0 + t < n) ? rhs_t - s : 0.0;
  }
  __syncthreads();
  // y_i = Linv * v: every thread a quarter of a row's 64 terms (a chain of 16, not 64, on everybody's path), then the
  // four quarters in a fixed order
  {
    double q4 = 0.0;
#pragma unroll
    for(int k = 0; k < 16; k++) q4 += Li[r + (16*g + k)*LDT]*v[16*g + k];          // Linv is zero above the diagonal
    part[g*NB + r] = q4;
  }
  __syncthreads();
  double yi = 0.0;
  if(t < NB)
  {
    yi = (part[t] + part[NB + t]) + (part[2*NB + t] + part[3*NB + t]);
    // (rows past the end hand over zeros: a block is NB values, and none of them may stay the sentinel)
    __hip_atomic_store((gu_t)(ycur + row0 + t), (unsigned long long)__double_as_longlong(row0 + t < n ? yi : 0.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- backward: the tile L(k, i) goes through LDS (thread = column afterwards)
  double bacc = 0.0;
  for(int k = T - 1; k > i; k--)
  {
    __syncthreads();
    for(int e = t; e < NB*NB; e += TPB)
    {
      const int rr = e % NB, cc = e / NB;
      const int row = NB*k + rr, col = row0 + cc;
      Lt[rr][cc] = (row < n && col < n) ? A[(size_t)col*lda + row] : 0.0;
    }
    if(t < NB) v[t] = take(xtake, k);
    __syncthreads();
#pragma unroll
    for(int q = 0; q < 16; q++) bacc += Lt[16*g + q][r]*v[16*g + q];      // thread (column r, row group g)
  }
  __syncthreads();
  part[g*NB + r] = bacc;
  __syncthreads();
  if(t < NB)
  {
    const double s = (part[t] + part[NB + t]) + (part[2*NB + t] + part[3*NB + t]);
    v[t] = yi - s;                                   // (rows past the end: yi = 0, s = 0)
  }
  __syncthreads();
  {
    double q4 = 0.0;
#pragma unroll
    for(int k = 0; k < 16; k++) q4 += Li[(16*g + k) + r*LDT]*v[16*g + k];         // Linv' : column r of Linv
    part[g*NB + r] = q4;
  }
  __syncthreads();
  if(t < NB)
  {
    const double xi = (part[t] + part[NB + t]) + (part[2*NB + t] + part[3*NB + t]);
    if(row0 + t < n) X[row0 + t] = xi;
    __hip_atomic_store((gu_t)(xcur + row0 + t), (unsigned long long)__double_as_longlong(row0 + t < n ? xi : 0.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

} // namespace

// the dynamic-LDS limit of a kernel, set once per device (a failure is left for the launch to report)
constexpr int DLG_MAX_DEV = 64;
static void dlg_func_lds_once(bool (&done)[DLG_MAX_DEV], const void* fn, int bytes)
{
  int dev = 0;
  if(hipGetDevice(&dev) != hipSuccess) return;
  dev &= DLG_MAX_DEV - 1;
  if(done[dev]) return;
  if(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
  { dlg_set_error("hipFuncSetAttribute(%d bytes of dynamic LDS) failed on device %d", bytes, dev); return; }
  done[dev] = true;
}

void dense_launch_trsv_tiles(hipStream_t st, const double* A, int lda, int n, const double* Linv, const double* rhs,
                             double* Yh, double* X, double* Xh, int epoch, const DlgHandoff& ho)
{
  static bool attr[DLG_MAX_DEV] = {};       // a function attribute is a property of (function, device)
  constexpr int LDSB = 88*1024;           // > half of the CU's LDS: one workgroup per CU
  dlg_func_lds_once(attr, reinterpret_cast<const void*>(&k_trsv_tiles), LDSB);
  const int T = (n + NB - 1)/NB;
  hipLaunchKernelGGL(k_trsv_tiles, dim3(T), dim3(TPB), LDSB, st, A, lda, n, T, Linv, rhs, Yh, X, Xh, epoch, ho);
}
// (both sets of hand-off buffers empty: before the first launch)
namespace { __global__ void k_trsv_arm(unsigned long long* a, size_t n) { for(size_t i = blockIdx.x*(size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x*blockDim.x) a[i] = TRSV_EMPTY; } }
void dense_trsv_arm(hipStream_t st, double* Yh, double* Xh, size_t n_each)
{
  hipLaunchKernelGGL(k_trsv_arm, dim3(64), dim3(256), 0, st, reinterpret_cast<unsigned long long*>(Yh), n_each);
  hipLaunchKernelGGL(k_trsv_arm, dim3(64), dim3(256), 0, st, reinterpret_cast<unsigned long long*>(Xh), n_each);
}

void dense_launch_potrf_tiles(hipStream_t st, double* A, int lda, int n, int* info_dev, double* Linv, int* flags, int epoch, const DlgHandoff& ho)
{
  static bool attr[DLG_MAX_DEV] = {};       // a function attribute is a property of (function, device)
  constexpr int LDSB = 88*1024;           // > half of the CU's LDS: one workgroup per CU
  dlg_func_lds_once(attr, reinterpret_cast<const void*>(&k_potrf_tiles), LDSB);
  const int T = (n + NB - 1)/NB;
  const int self_x = getenv("DOGLEG_AMD_NO_POTRF_SELF") ? 0 : 1;      // (the form of rounds 2 - 3: every block of L comes from its owner)
  hipLaunchKernelGGL(k_potrf_tiles, dim3(T*(T + 1)/2), dim3(TPB), LDSB, st, A, lda, n, T, info_dev, Linv, flags, epoch, ho, self_x);
}

void dense_launch_potrf_diag_trsm(hipStream_t st, double* A, int lda, int kb, int nb, int n, int* info_dev, double* Linv,
                                  int* flag, int epoch, const DlgHandoff& ho)
{
  static bool attr[DLG_MAX_DEV] = {};       // a function attribute is a property of (function, device)
  constexpr int LDSB = 88*1024;           // > half of the CU's LDS: one workgroup per CU
  dlg_func_lds_once(attr, reinterpret_cast<const void*>(&k_potrf_diag_trsm), LDSB);
  const int ntr = (n - kb - nb + NB - 1)/NB;
  hipLaunchKernelGGL(k_potrf_diag_trsm, dim3(1 + ntr), dim3(TPB), LDSB, st, A, lda, kb, nb, n, info_dev, Linv, flag, epoch, ho);
}

void dense_launch_potrf_diag(hipStream_t st, double* A, int lda, int kb, int nb, int* info_dev, double* Linv)
{
  hipLaunchKernelGGL(k_potrf_diag_inv, dim3(1), dim3(TPB), 0, st, A, lda, kb, nb, info_dev, Linv);
}
