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
#ifdef DLG_POTRF_PROFILE
// tools/potrf_prof.py: the diagonal owners' time line (tile (j, j) -> slot j)
__device__ long long g_potrf_dbg[64*8];
#define POTRF_STAMP(k) do { if(threadIdx.x == 0 && ti == tj && tj < 64) g_potrf_dbg[tj*8 + (k)] = wall_clock64(); } while(0)
#else
#define POTRF_STAMP(k)
#endif
__global__ void __launch_bounds__(TPB) k_potrf_tiles(double* A, int lda, int n, int T, int* __restrict__ info,
                                                     double* Linv, int* flags, int epoch, DlgHandoff ho, int self_x, int* gate)
{
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ int sbad;
  // the first workgroup -- in front of every pivot of the launch -- clears the pivot word (write-through: a plain store
  // could reach memory behind another XCD's atomic) and tells the second stream that the factorisation is on the chip
  if(blockIdx.x == 0 && threadIdx.x == 0)
  {
    __hip_atomic_store(info, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if(gate) __hip_atomic_store(gate, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
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
  POTRF_STAMP(0);
  // (the panel [A; I] of a diagonal tile's sweep has LDS of its own behind the two staging tiles: its identity half is
  // written here, long before the tile is complete, and the tile goes there straight from the accumulators)
  double* P = sm + 2*NB*LDT;
  constexpr int LD = 2*NB;
  if(ti == tj)
    for(int e = t; e < NB*NB; e += TPB) { const int i = e % NB, j = e / NB; P[NB + i + j*LD] = (i == j) ? 1.0 : 0.0; }
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
  // (thread t: row t % 64 of the columns t / 64 + 4 u -- all sixteen loads on their way before the first is used)
  auto stage = [&](double (*D)[LDT], int r0, int c0) {
    const int i = t & (NB - 1), row = r0 + i;
    double v[NB*NB/TPB];
#pragma unroll
    for(int u = 0; u < NB*NB/TPB; u++)
    {
      const int col = c0 + (t >> 6) + (TPB/NB)*u;
      v[u] = (row < n && col < n) ? __hip_atomic_load((gcd_t)(A + (size_t)col*lda + row), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    }
#pragma unroll
    for(int u = 0; u < NB*NB/TPB; u++) D[i][(t >> 6) + (TPB/NB)*u] = v[u];
  };
  // the published inverse of a diagonal block (lower triangular: only that half travels)
  auto stage_inv = [&](double (*D)[LDT], const double* Lv) {
    const int i = t & (NB - 1);
    double v[NB*NB/TPB];
#pragma unroll
    for(int u = 0; u < NB*NB/TPB; u++)
    {
      const int k = (t >> 6) + (TPB/NB)*u;
      v[u] = (i >= k) ? __hip_atomic_load((gcd_t)(Lv + k*NB + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    }
#pragma unroll
    for(int u = 0; u < NB*NB/TPB; u++) D[i][(t >> 6) + (TPB/NB)*u] = v[u];
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
    POTRF_STAMP(1);
    if(t == 0) wait_flag(tj - 1, tj - 1);
    __syncthreads();
    POTRF_STAMP(2);
    {
      stage_inv(Lj, Linv + (size_t)(tj - 1)*NB*NB);
#pragma unroll
      for(int ct = 0; ct < 4; ct++)
#pragma unroll
        for(int r = 0; r < 4; r++) Li[16*wv + kq + 4*r][16*ct + jn] = acc2[ct][r];
    }
    __syncthreads();
    POTRF_STAMP(3);
    dd_v4d x[4];
#pragma unroll
    for(int ct = 0; ct < 4; ct++) x[ct] = (dd_v4d){0.0, 0.0, 0.0, 0.0};
    // (the inverse is lower triangular: columns 16 ct .. of the product take k < 16 (ct + 1) only -- 40 products, not 64)
#pragma unroll
    for(int kk = 0; kk < NB; kk += 4)
    {
      const double a = Li[16*wv + jn][kk + kq];
#pragma unroll
      for(int ct = 0; ct < 4; ct++)
        if(kk < 16*(ct + 1)) x[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Lj[16*ct + jn][kk + kq], x[ct], 0, 0, 0);
    }
    __syncthreads();
#pragma unroll
    for(int ct = 0; ct < 4; ct++)
#pragma unroll
      for(int r = 0; r < 4; r++) Li[16*wv + kq + 4*r][16*ct + jn] = x[ct][r];
    __syncthreads();
    POTRF_STAMP(4);
#pragma unroll 4
    for(int kk = 0; kk < NB; kk += 4)
    {
      const double a = Li[16*wv + jn][kk + kq];
#pragma unroll
      for(int ct = 0; ct < 4; ct++)
        acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, -Li[16*ct + jn][kk + kq], acc[ct], 0, 0, 0);
    }
  }
  if(ti == tj)
  {
    // the diagonal tile: [A; I] -> [L; L^-T] in one panel sweep (as k_potrf_diag_inv)
#pragma unroll
    for(int ct = 0; ct < 4; ct++)
#pragma unroll
      for(int r = 0; r < 4; r++)
      {
        const int i = 16*wv + kq + 4*r, j = 16*ct + jn;
        P[i + j*LD] = (i >= j) ? acc[ct][r] : 0.0;
      }
    if(t == 0) sbad = 0x7fffffff;
    __syncthreads();
    POTRF_STAMP(5);
    panel_factor_b16<TPB>(P, LD, 2*NB, NB, t, &sbad, 0);
    POTRF_STAMP(6);
    const int nb = min(NB, n - col0);
    if(t == 0) { const int bad = sbad; if(bad < nb) atomicCAS(info, 0, col0 + bad + 1); }
    double* Lv = Linv + (size_t)tj*NB*NB;
    // Linv(i, j) = P[NB + j + i*LD]: a row of the inverse is a column of the panel.  Read along the panel's columns
    // (lanes over j: no bank conflict; lanes over i, LD = 128 apart, all hit ONE bank) into a staging tile, stored from
    // there with lanes over i (coalesced).  Above the diagonal the buffer holds zeros from its allocation on.
    {
      const int l = t & (NB - 1);
#pragma unroll
      for(int u = 0; u < NB*NB/TPB; u++) { const int i = (t >> 6) + (TPB/NB)*u; Li[l][i] = P[NB + l + i*LD]; }      // Li[j][i] = Linv(i, j)
      __syncthreads();
#pragma unroll
      for(int u = 0; u < NB*NB/TPB; u++)
      {
        const int j = (t >> 6) + (TPB/NB)*u;
        if(l >= j) __hip_atomic_store((gd_t)(Lv + j*NB + l), Li[j][l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if(t == 0) __hip_atomic_store(flags + tj*T + tj, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    POTRF_STAMP(7);
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
    stage_inv(Lj, Linv + (size_t)tj*NB*NB);
#pragma unroll
    for(int ct = 0; ct < 4; ct++)
#pragma unroll
      for(int r = 0; r < 4; r++) Li[16*wv + kq + 4*r][16*ct + jn] = acc[ct][r];
  }
  __syncthreads();
  dd_v4d x[4];
#pragma unroll
  for(int ct = 0; ct < 4; ct++) x[ct] = (dd_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for(int kk = 0; kk < NB; kk += 4)
  {
    const double a = Li[16*wv + jn][kk + kq];
#pragma unroll
    for(int ct = 0; ct < 4; ct++)
      if(kk < 16*(ct + 1)) x[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Lj[16*ct + jn][kk + kq], x[ct], 0, 0, 0);
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
// block (and no drain + barrier + flag store on the owner's side).  Two sets of buffers, used by launches of even /
// odd epoch; a launch re-arms the set of the NEXT launch (the launch before this one, which used it, is over).
// All T workgroups must be resident (they wait for higher-numbered ones on the way back): T <= #CUs / 2.
constexpr unsigned long long TRSV_EMPTY = 0x7FF8DEADBEEF0001ull;

#ifdef DLG_TRSV_PROFILE
__device__ long long g_trsv_dbg[64*8];
#define TRSV_STAMP(k) do { if(threadIdx.x == 0 && blockIdx.x < 64) g_trsv_dbg[blockIdx.x*8 + (k)] = wall_clock64(); } while(0)
#else
#define TRSV_STAMP(k)
#endif
// ---- the same two sweeps with the last TWO blocks a workgroup waits for folded into products formed ahead (round 5) ----
// y_i = Linv_i (b_i - sum_{k<i} L(i,k) y_k).  Measured per workgroup (tools/trsv_prof.py): the way down ran at 1.4 us a
// hop although a block of y crosses in 0.36 us -- behind the arrival of y_{i-2} sat the tile product, its four-way sum,
// the product with Linv_i and its sum (six barriers), longer than a hop --, and the way back at 4.5 us a hop: every
// workgroup staged its T - 1 - i tiles through LDS one after the other, a 32 KB load each that nothing covered.  Now
//   * with  M1 = Linv_i L(i, i-1),  M2 = Linv_i L(i, i-2)  (64 x 64 x 64 products on the matrix cores while the workgroup
//     waits; a thread keeps ITS 16 entries of each in registers) and  z = Linv_i (b_i - sum_{k<i-2} L(i,k) y_k),
//     y_i = z - M2 y_{i-2} - M1 y_{i-1}:  behind the arrival of y_{i-1} there is one product of 16 terms a thread and one
//     four-way sum; what depends on y_{i-3} (the sums, z) has two hops to finish in;
//   * the way back the same with  Linv_i' L(i+1, i)',  Linv_i' L(i+2, i)',  and the tiles L(k, i), k > i + 2, come
//     straight into registers (a thread: 16 consecutive rows of its column), four tiles ahead of their use.
__global__ void __launch_bounds__(TPB) k_trsv_tiles(const double* __restrict__ A, int lda, int n, int T,
                                                    const double* __restrict__ Linv, const double* __restrict__ rhs,
                                                    double* Yh, double* X, double* Xh, int epoch, DlgHandoff ho)
{
  extern __shared__ __attribute__((aligned(16))) double sm[];
  typedef __attribute__((address_space(1))) unsigned long long* gu_t;
  constexpr int LDT = NB + 1;
  double (*Lt)[LDT] = reinterpret_cast<double (*)[LDT]>(sm);                 // a tile of L: Lt[row][col]
  double* Li = sm + NB*LDT;                                                  // Li[r + c*LDT] = Linv(r, c)
  double* Ms = Li + NB*LDT;                                                  // scratch: a product on its way into registers, Ms[r + c*LDT]
  double* v = Ms + NB*LDT;                                                   // [NB]  the vector waited for
  double* part = v + NB;                                                     // [4][NB] partial sums
  double* part2 = part + 4*NB;                                               // [4][NB] the second set (a product read from the first)
  double* vb = part2 + 4*NB;                                                 // [NB] b_i, then y_i
  const int t = threadIdx.x, r = t & 63, g = t >> 6;
  const int lane = t & 63, wv = t >> 6, jn = lane & 15, kq = lane >> 4;
  const int i = blockIdx.x, row0 = NB*i;
  const int npad = T*NB, par = epoch & 1;
  unsigned long long* ycur = reinterpret_cast<unsigned long long*>(Yh) + (size_t)par*npad;
  unsigned long long* xcur = reinterpret_cast<unsigned long long*>(Xh) + (size_t)par*npad;
  unsigned long long* ytake = ho.skew ? reinterpret_cast<unsigned long long*>(Yh) + (size_t)2*npad : ycur;
  unsigned long long* xtake = ho.skew ? reinterpret_cast<unsigned long long*>(Xh) + (size_t)2*npad : xcur;
  if(t < NB)
  {
    reinterpret_cast<unsigned long long*>(Yh)[(size_t)(1 - (epoch & 1))*npad + row0 + t] = TRSV_EMPTY;
    reinterpret_cast<unsigned long long*>(Xh)[(size_t)(1 - (epoch & 1))*npad + row0 + t] = TRSV_EMPTY;
  }
  auto take = [&](unsigned long long* buf, int k) -> double {
    unsigned long long u; int spins = 0;
    while((u = __hip_atomic_load((gu_t)(buf + NB*k + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == TRSV_EMPTY)
    { __builtin_amdgcn_s_sleep(1); if(++spins > ho.spins) { atomicOr(ho.status, DLG_HANDOFF_TRSV); u = 0; break; } }
    return __longlong_as_double((long long)u);
  };
  // a tile of L into Lt (rows of block kr, columns of block kc; zeros past the end)
  auto stage_tile = [&](int kr, int kc) {
    for(int e = t; e < NB*NB; e += TPB)
    {
      const int rr = e % NB, cc = e / NB;
      const int row = NB*kr + rr, col = NB*kc + cc;
      Lt[rr][cc] = (row < n && col < n) ? A[(size_t)col*lda + row] : 0.0;
    }
  };
  // Ms(row, col) = sum_k Aop(row, k) Bop(col, k) on the matrix cores, wave w the rows 16 w .. 16 w + 15; then this
  // thread's 16 entries of it: m[c] = Ms(r, 16 g + c)
  auto mm64 = [&](auto Aop, auto Bop, double (&m)[16]) {
    dd_v4d x[4];
#pragma unroll
    for(int ct = 0; ct < 4; ct++) x[ct] = (dd_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for(int kk = 0; kk < NB; kk += 4)
    {
      const double a = Aop(16*wv + jn, kk + kq);
#pragma unroll
      for(int ct = 0; ct < 4; ct++) x[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bop(16*ct + jn, kk + kq), x[ct], 0, 0, 0);
    }
#pragma unroll
    for(int ct = 0; ct < 4; ct++)
#pragma unroll
      for(int q = 0; q < 4; q++) Ms[(16*wv + kq + 4*q) + (16*ct + jn)*LDT] = x[ct][q];
    __syncthreads();
#pragma unroll
    for(int c = 0; c < 16; c++) m[c] = Ms[r + (16*g + c)*LDT];
    __syncthreads();
  };
  // (forward products: Linv_i times the tile; backward: Linv_i' times the tile's transpose)
  auto fold_fwd = [&](int k, double (&m)[16]) {
    stage_tile(i, k);
    __syncthreads();
    mm64([&](int row, int kk) { return Li[row + kk*LDT]; }, [&](int col, int kk) { return Lt[kk][col]; }, m);
  };
  auto fold_bwd = [&](int k, double (&m)[16]) {
    stage_tile(k, i);
    __syncthreads();
    mm64([&](int row, int kk) { return Li[kk + row*LDT]; }, [&](int col, int kk) { return Lt[col][kk]; }, m);
  };
  auto dot16 = [&](const double (&m)[16]) { double q4 = 0.0;
#pragma unroll
    for(int c = 0; c < 16; c++) q4 += m[c]*v[16*g + c];
    return q4; };
  auto sum4 = [&]() { return (part[t] + part[NB + t]) + (part[2*NB + t] + part[3*NB + t]); };      // (t < NB)
  TRSV_STAMP(0);
  for(int e = t; e < NB*NB; e += TPB) Li[(e & (NB - 1)) + (e/NB)*LDT] = Linv[(size_t)i*NB*NB + e];
  if(t < NB) vb[t] = (row0 + t < n) ? rhs[row0 + t] : 0.0;
  __syncthreads();
  double M1[16], M2[16], B1[16], B2[16];
#pragma unroll
  for(int c = 0; c < 16; c++) { M1[c] = 0.0; M2[c] = 0.0; B1[c] = 0.0; B2[c] = 0.0; }
  // (the first block rows are needed before two products would be done: they take their tiles the plain way)
  const bool ffold = i >= 8;
  if(ffold) fold_fwd(i - 1, M1);
  if(ffold) fold_fwd(i - 2, M2);
  // the products of the way back: behind the departure of y_i where the turn-round leaves the time (2 (T - 1 - i) hops
  // until x_{i+1} is there; two products take ~12 us), now for the last block rows -- they have the whole way down to wait
  const bool b_early = T - 1 - i < 8;
  if(b_early && i + 1 < T) fold_bwd(i + 1, B1);
  if(b_early && i + 2 < T) fold_bwd(i + 2, B2);
  TRSV_STAMP(1);
  // ---- forward: thread (row r, column group g) keeps 16 values of a tile in registers; blocks 0 .. i - 3
  const int nf = ffold ? i - 2 : i;                  // tiles of the loop
  double acc = 0.0;
  double cur[16];
  auto load_row_tile = [&](int k, double (&d)[16]) {
#pragma unroll
    for(int c = 0; c < 16; c++)
    {
      const int row = row0 + r, col = NB*k + 16*g + c;
      d[c] = (row < n && k < nf) ? A[(size_t)col*lda + row] : 0.0;
    }
  };
  auto fstep = [&](int k, const double (&d)[16]) {
    __syncthreads();                                 // (v of the step before is done with)
    if(t < NB) v[t] = take(ytake, k);
    __syncthreads();
#pragma unroll
    for(int c = 0; c < 16; c++) acc += d[c]*v[16*g + c];
  };
  {
    // (three tiles on their way: a block row near the end starts its loop late -- its products of the way back come first --
    // and has to catch up with blocks of y that are all there)
    double f1[16], f2[16];
    load_row_tile(0, cur); load_row_tile(1, f1); load_row_tile(2, f2);
    for(int k = 0; k < nf; k += 3)
    {
      fstep(k, cur); load_row_tile(k + 3, cur);
      if(k + 1 < nf) { fstep(k + 1, f1); load_row_tile(k + 4, f1); }
      if(k + 2 < nf) { fstep(k + 2, f2); load_row_tile(k + 5, f2); }
    }
  }
  __syncthreads();
  part[g*NB + r] = acc;
  __syncthreads();
  // z = Linv (b - s): every thread a quarter of a row's 64 terms, the four partial sums of s taken in as it goes (no
  // pass of its own for s: one barrier less on the path), then the four quarters in a fixed order
  {
    double q4 = 0.0;
#pragma unroll
    for(int k = 0; k < 16; k++)
    {
      const int c = 16*g + k;
      const double sv = (part[c] + part[NB + c]) + (part[2*NB + c] + part[3*NB + c]);
      q4 += Li[r + c*LDT]*(vb[c] - sv);                              // Linv is zero above the diagonal; vb: b_i (zeros past the end)
    }
    part2[g*NB + r] = q4;
  }
  __syncthreads();
  double yi = (t < NB) ? (part2[t] + part2[NB + t]) + (part2[2*NB + t] + part2[3*NB + t]) : 0.0;
  double pq = 0.0;                                   // this thread's quarter of  M2 y_{i-2} + M1 y_{i-1}
  if(ffold)
  {
    __syncthreads();
    if(t < NB) v[t] = take(ytake, i - 2);
    __syncthreads();
    pq = dot16(M2);
  }
  if(ffold)
  {
    // the one block on everybody's path: y_{i-1}
    __syncthreads();
    TRSV_STAMP(2);
    if(t < NB) v[t] = take(ytake, i - 1);
    __syncthreads();
    TRSV_STAMP(3);
    pq += dot16(M1);
    part[g*NB + r] = pq;
    __syncthreads();
    if(t < NB) yi -= sum4();
  }
  if(t < NB)
  {
    __hip_atomic_store((gu_t)(ycur + row0 + t), (unsigned long long)__double_as_longlong(row0 + t < n ? yi : 0.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    vb[t] = (row0 + t < n) ? yi : 0.0;               // (for the way back: x_i = Linv_i' (y_i - ...))
  }
  TRSV_STAMP(4);
  if(!b_early)
  {
    __syncthreads();
    if(i + 1 < T) fold_bwd(i + 1, B1);
    if(i + 2 < T) fold_bwd(i + 2, B2);
  }
  // ---- backward: the tiles L(k, i), k = T - 1 .. i + 3, three of them on their way in registers -- fetched with
  // lane = row (whole lines; a thread = 16 consecutive rows of its column asked for 64 lines an instruction and could not
  // keep up with the blocks of x: 2.2 us a hop) -- and turned through LDS when their block of x is there: thread (column r,
  // row group g) then
  auto load_tile_rows = [&](int k, double (&d)[16]) {
    const bool on = k > i + 2 && k < T;
#pragma unroll
    for(int j = 0; j < 16; j++)
    {
      const int rr = t & 63, cc = (t >> 6) + 4*j;
      const int row = NB*k + rr, col = row0 + cc;
      d[j] = (on && row < n && col < n) ? A[(size_t)col*lda + row] : 0.0;
    }
  };
  double bacc = 0.0;
  auto bstep = [&](int k, const double (&d)[16]) {
    __syncthreads();                                 // (v and Lt of the step before are done with)
#pragma unroll
    for(int j = 0; j < 16; j++) Lt[t & 63][(t >> 6) + 4*j] = d[j];
    if(t < NB) v[t] = take(xtake, k);
    __syncthreads();
#pragma unroll
    for(int q = 0; q < 16; q++) bacc += Lt[16*g + q][r]*v[16*g + q];
  };
  {
    double t0[16], t1[16], t2[16];
    int k = T - 1;
    load_tile_rows(k, t0); load_tile_rows(k - 1, t1); load_tile_rows(k - 2, t2);
    for(; k > i + 2; k -= 3)
    {
      bstep(k, t0); load_tile_rows(k - 3, t0);
      if(k - 1 > i + 2) { bstep(k - 1, t1); load_tile_rows(k - 4, t1); }
      if(k - 2 > i + 2) { bstep(k - 2, t2); load_tile_rows(k - 5, t2); }
    }
  }
  __syncthreads();
  part[g*NB + r] = bacc;
  __syncthreads();
  {
    double q4 = 0.0;
#pragma unroll
    for(int k = 0; k < 16; k++)
    {
      const int c = 16*g + k;
      const double sv = (part[c] + part[NB + c]) + (part[2*NB + c] + part[3*NB + c]);
      q4 += Li[c + r*LDT]*(vb[c] - sv);                              // Linv' : column r of Linv; vb: y_i
    }
    part2[g*NB + r] = q4;
  }
  __syncthreads();
  double xi = (t < NB) ? (part2[t] + part2[NB + t]) + (part2[2*NB + t] + part2[3*NB + t]) : 0.0;
  pq = 0.0;
  if(i + 2 < T)
  {
    __syncthreads();
    if(t < NB) v[t] = take(xtake, i + 2);
    __syncthreads();
    pq = dot16(B2);
  }
  if(i + 1 < T)
  {
    __syncthreads();
    TRSV_STAMP(5);
    if(t < NB) v[t] = take(xtake, i + 1);
    __syncthreads();
    TRSV_STAMP(6);
    pq += dot16(B1);
    part[g*NB + r] = pq;
    __syncthreads();
    if(t < NB) xi -= sum4();
  }
  if(t < NB)
  {
    if(row0 + t < n) X[row0 + t] = xi;
    __hip_atomic_store((gu_t)(xcur + row0 + t), (unsigned long long)__double_as_longlong(row0 + t < n ? xi : 0.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  TRSV_STAMP(7);
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
  const int T = (n + NB - 1)/NB;
  constexpr int LDSB = (3*NB*(NB + 1) + 10*NB + 16)*(int)sizeof(double);           // three 64 x 65 tiles + the vectors: 102 KB, one workgroup per CU
  dlg_func_lds_once(attr, reinterpret_cast<const void*>(&k_trsv_tiles), LDSB);
  hipLaunchKernelGGL(k_trsv_tiles, dim3(T), dim3(TPB), LDSB, st, A, lda, n, T, Linv, rhs, Yh, X, Xh, epoch, ho);
}
#ifdef DLG_POTRF_PROFILE
extern "C" void dlg_potrf_profile_dump(int T)
{
  long long h[64*8];
  if(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_potrf_dbg), sizeof(h)) != hipSuccess) return;
  const long long t0 = h[0];
  for(int i = 0; i < T && i < 64; i++)
    fprintf(stderr, "  potrf diag %2d: start %6lld | updates k < j-1 done, waits for inv(j-1) at %6lld, has it %6lld, staged %6lld, L(j,j-1) formed %6lld | sweep from %6lld to %6lld, inverse published %6lld\n",
            i, h[i*8] - t0, h[i*8+1] - t0, h[i*8+2] - t0, h[i*8+3] - t0, h[i*8+4] - t0, h[i*8+5] - t0, h[i*8+6] - t0, h[i*8+7] - t0);
}
#endif
#ifdef DLG_TRSV_PROFILE
extern "C" void dlg_trsv_profile_dump(int T)
{
  long long h[64*8];
  if(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_trsv_dbg), sizeof(h)) != hipSuccess) return;
  const long long t0 = h[0];
  for(int i = 0; i < T && i < 64; i++)
    fprintf(stderr, "  trsv wg %2d: start %6lld set-up done %6lld | waits for y(i-1) at %6lld, has it at %6lld, y(i) out %6lld | waits for x(i+1) at %6lld, has it %6lld, x(i) out %6lld\n",
            i, h[i*8] - t0, h[i*8+1] - t0, h[i*8+2] - t0, h[i*8+3] - t0, h[i*8+4] - t0, h[i*8+5] - t0, h[i*8+6] - t0, h[i*8+7] - t0);
}
#endif
// (both sets of hand-off buffers empty: before the first launch)
namespace { __global__ void k_trsv_arm(unsigned long long* a, size_t n) { for(size_t i = blockIdx.x*(size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x*blockDim.x) a[i] = TRSV_EMPTY; } }
void dense_trsv_arm(hipStream_t st, double* Yh, double* Xh, size_t n_each)
{
  hipLaunchKernelGGL(k_trsv_arm, dim3(64), dim3(256), 0, st, reinterpret_cast<unsigned long long*>(Yh), n_each);
  hipLaunchKernelGGL(k_trsv_arm, dim3(64), dim3(256), 0, st, reinterpret_cast<unsigned long long*>(Xh), n_each);
}

void dense_launch_potrf_tiles(hipStream_t st, double* A, int lda, int n, int* info_dev, double* Linv, int* flags, int epoch, const DlgHandoff& ho, int* gate)
{
  static bool attr[DLG_MAX_DEV] = {};       // a function attribute is a property of (function, device)
  constexpr int LDSB = (2*NB*(NB + 1) + 2*NB*NB)*8;      // two staging tiles + the panel [A; I] of a diagonal tile (130 KB: one workgroup per CU)
  dlg_func_lds_once(attr, reinterpret_cast<const void*>(&k_potrf_tiles), LDSB);
  const int T = (n + NB - 1)/NB;
  const int self_x = 1;      // (0, the form of rounds 2 - 3: every block of L comes from its owner)
  hipLaunchKernelGGL(k_potrf_tiles, dim3(T*(T + 1)/2), dim3(TPB), LDSB, st, A, lda, n, T, info_dev, Linv, flags, epoch, ho, self_x, gate);
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
