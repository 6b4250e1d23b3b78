// sparse_solve.hip -- K6: triangular solves with the supernodal factor on gfx950,
// replaces cholmod_solve(CHOLMOD_A) (dogleg.c:853).
#include "sparse_internal.h"

namespace {
// -DDLG_FL_PROFILE: phase clocks of workgroup 0 of every backward-solve launch (tools only)
#ifdef DLG_FL_PROFILE
__device__ long long g_bw_prof[64*8];
__device__ long long g_bw_chain[256*8];      // the persistent launch: every workgroup, 100 MHz clock
#define BW_STAMP(k) do { if(threadIdx.x == 0 && blockIdx.x == 0) g_bw_prof[(prof_lvl & 63)*8 + (k)] = clock64(); \
                         if(threadIdx.x == 0 && pr_flag && blockIdx.x < 256) g_bw_chain[blockIdx.x*8 + (k)] = wall_clock64(); } while(0)
#else
#define BW_STAMP(k)
#endif

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
// A dependent global load costs microseconds at the top of the tree, so the kernel is built in
// three rounds of loads: (1) the supernode's flat record (SolveItem); (2) everything that only
// needs the record -- the indices of the below rows, the right-hand side, the operands of the
// L_below^T mat-vec (up to two passes of 16 values per thread, in registers), the diagonal
// blocks and, when it fits, the whole top block into LDS; (3) x at the below rows.  Then the
// waves share the columns of the mat-vec, and the triangular solve runs over blocks of 8
// columns from the bottom, thread = row: the 8 owners of a block publish their right-hand
// sides, after ONE barrier every thread solves the 8x8 block itself (diagonal blocks in LDS
// with reciprocal pivots) and applies the 8 new unknowns to its own row with the values of L
// from the LDS copy (or, for wide supernodes, fetched a block ahead from HBM).
template <int BWD_NT, bool BD_ONLY>
__global__ void __launch_bounds__(BWD_NT, BD_ONLY ? 4 : 1) k_solve_bwd_level(const SolveItem* __restrict__ items,
                                                            const int* __restrict__ sn_rows,
                                                            const int* __restrict__ perm,
                                                            const double* __restrict__ Lx,
                                                            double* __restrict__ ywork,
                                                            double* __restrict__ out, int use_aug,
                                                            const int* __restrict__ sn_bd_col, int top_lds, int xb_cap,
                                                            int* pr_flag, int pr_epoch, const int* __restrict__ info, DlgHandoff ho,
                                                            double* xh, int xh_n, double* __restrict__ mm)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ int s_skip;
  // (mm: smallest / largest diagonal entry of this supernode's part of L, a pair per WAVE that holds diagonal entries
  // ([workgroup][8][2]; no LDS, no barrier: the waves store their own) -- the step kernel turns them into the pivot ratio
  // that decides whether the expected improvement may come from the solved system, backend.hip)
  double dmin = 1e300, dmax = 0.0;
  if(threadIdx.x == 0) s_skip = *info != 0x7fffffff;      // the factor is that of a failed factorisation: its solution is never used
  constexpr int NW = BWD_NT/64;
  // persistent top region (pr_flag != null): ONE launch for the last levels of the tree, workgroups
  // from the root down; x of the ancestors was written in this very launch -- write-through stores,
  // loads around L1 (global_ ... sc1), a flag per supernode carrying the epoch of the launch
  typedef const __attribute__((address_space(1))) double* gcd_t;
  typedef __attribute__((address_space(1))) double* gd_t;
  auto ldx = [&](const double* p) -> double { return pr_flag ? __hip_atomic_load((gcd_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p; };
  // One-launch region, x of the ancestors staged in LDS: x IS the signal.  A copy of the region's part of x (xh, two
  // sets for launches of even / odd epoch) holds a sentinel -- a NaN no solve produces -- until a supernode stores its
  // solution there; a workgroup polls the entries of its below rows themselves, every lane its own, instead of flag,
  // barrier, gather (one trip through L2 a level instead of two, and the parent's drain + barrier + flag store are
  // off the children's path).  A launch re-arms the set of the next launch.  (The forced time-out of the tests and
  // supernodes whose below rows do not fit LDS keep the flags, which are raised as before.)
  constexpr unsigned long long X_EMPTY = 0x7FF8DEADBEEF0002ull;
  typedef __attribute__((address_space(1))) unsigned long long* gu_t;
  const bool xh_on = pr_flag != nullptr && xh != nullptr && ho.skew == 0;
  unsigned long long* xcur = reinterpret_cast<unsigned long long*>(xh) + (size_t)(pr_epoch & 1)*xh_n;
  auto put_xh = [&](int col, double v) {
    if(xh_on) __hip_atomic_store((gu_t)(xcur + col), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto take_xh = [&](int col) -> double {
    unsigned long long u; int spins = 0;
    while((u = __hip_atomic_load((gu_t)(xcur + col), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == X_EMPTY)
    { __builtin_amdgcn_s_sleep(1); if(++spins > ho.spins) { atomicOr(ho.status, DLG_HANDOFF_SOLVE); u = 0; break; } }
    return __longlong_as_double((long long)u); };

  const int prof_lvl = top_lds >> 8; (void)prof_lvl;
  top_lds &= 1;
  BW_STAMP(0);
  const SolveItem it = items[blockIdx.x];
  const int c0 = it.c0, w = it.w, nrows = it.nrows;
  if(xh_on && (int)threadIdx.x < w) reinterpret_cast<unsigned long long*>(xh)[(size_t)(1 - (pr_epoch & 1))*xh_n + c0 + threadIdx.x] = X_EMPTY;
  const int* rows = sn_rows + it.rowoff;
  const double* L = Lx + it.lx;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = nrows - w - 1;
  const int nblk = (w + 7) >> 3;
  // x at the below rows is staged in LDS -- unless the supernode has more below rows than the level's
  // LDS allows (xb_cap; e.g. the first supernode of a wide dense tail): then the mat-vec gathers x
  // from HBM as it streams the columns
  const bool xb_lds = r <= xb_cap;
  double* xb = lds;                       // [r]   x at the below rows
  double* xs = lds + (xb_lds ? ((r + 1) & ~1) : 0);      // [256] -(L_below^T x), then the solution
  double* T = xs + 256;                   // [nblk][8][8] diagonal blocks (lower), reciprocal pivots
  double* rhs = T + nblk*64;              // [2][8]
  double* xp = rhs + 16;                  // [parts][256] partial sums of the mat-vec
  double* Lt = xp + 8*256;                // [w][ldt] top block (top_lds)
  const int ldt = w | 1;
  // Pre-multiplied sweep (premul): the sweep over the 8-column blocks needs, per block and thread,
  // upd = sum_a L(j0+a, tid) * x_blk(a) with x_blk = inv(block)' rh.  Written as sum_b rh(b) * m(b),
  // m(b) = sum_{a<=b} inv(block)(b,a) * L(j0+a, tid), the m's depend on the factor alone: they are
  // formed for ALL blocks before x of the ancestors is there (in the one-launch region: while the
  // workgroup waits for its parent), and the serial sweep shrinks to 8 multiply-adds per block and
  // thread plus one 8-term dot product in the 8 threads that own the block.
  const int MW = (w + 15) & ~15;
  double* Mx = Lt + (top_lds ? w*ldt : 0); // [nblk][8][MW]
  // block-diagonal top (merged sibling leaves): the members do not couple, every member is a
  // little triangular system of its own -- no sweep over the columns at all
  const int nmem = BD_ONLY ? max(it.nbd, 1) : it.nbd;      // BD_ONLY: every supernode of the launch has a block-diagonal top
  // ---- round 2
  const int myrow = (tid < r) ? rows[w + tid] : 0;
  double myrhs = 0.0; int myperm = 0;
  if(tid < w) { myrhs = use_aug ? L[(nrows - 1) + (size_t)tid*nrows] : ywork[c0 + tid]; myperm = perm[c0 + tid]; }
  // L_below^T x: thread = (column j, part of the below rows); MV_SLOTS consecutive rows per pass,
  // the first pass in flight from here on
  // 256 threads = populous levels of narrow supernodes: latency is hidden by resident workgroups,
  // so registers are kept low there (fewer values in flight, the member blocks fetched late)
  constexpr bool LEAN = BWD_NT == 256;
  constexpr int MV_SLOTS = LEAN ? 20 : 24;
  // (a thread reads its own stretch of a column: nothing is gained from a power-of-two column count,
  // and 66 columns are 7 parts of the rows, not 4; at most 8 parts -- xp)
  const int mv_cols = min(BWD_NT, max(w, 64)), mv_parts = BWD_NT/mv_cols;
  const int mv_j = tid % mv_cols, mv_p = tid/mv_cols;
  const int mv_len = (r + mv_parts - 1)/mv_parts;           // rows per part
  const int mv_i0 = mv_p*mv_len, mv_i1 = min(r, mv_i0 + mv_len);
  const bool mv_thread = mv_len <= MV_SLOTS && xb_lds;       // else: long columns, the waves stream them (below)
  const bool mv_on = mv_thread && mv_j < w && mv_p < mv_parts;
  const double* mv_L = L + (size_t)min(mv_j, w - 1)*nrows + w;
  double mv[MV_SLOTS];
  // (a thread's stretch of its column in 16-byte loads: 8-byte aligned pairs; K6 0.098 -> 0.090 ms on config #4)
  {
    typedef double bw_v2d __attribute__((ext_vector_type(2)));
    typedef bw_v2d bw_v2d_u __attribute__((aligned(8)));
#pragma unroll
    for(int q = 0; q < MV_SLOTS; q += 2)
    {
      if(mv_on && mv_i0 + q + 1 < mv_i1) { const bw_v2d t = *reinterpret_cast<const bw_v2d_u*>(mv_L + mv_i0 + q); mv[q] = t.x; mv[q + 1] = t.y; }
      else { mv[q] = (mv_on && mv_i0 + q < mv_i1) ? mv_L[mv_i0 + q] : 0.0; mv[q + 1] = 0.0; }
    }
  }
  if(!BD_ONLY)
  for(int e = tid; e < (nmem > 0 ? 0 : nblk*64); e += BWD_NT)
  {
    const int j0 = (e >> 6)*8, a = (e >> 3) & 7, b = e & 7;
    const bool valid = a >= b && j0 + a < w;
    double v = valid ? L[(j0 + a) + (size_t)(j0 + b)*nrows] : 0.0;
    if(a == b && valid) { dmin = fmin(dmin, v); dmax = fmax(dmax, v); }
    if(a == b) v = valid ? 1.0/v : 1.0;
    T[e] = v;
  }
  if(!BD_ONLY && top_lds && nmem == 0)
  {
    // thread = (row, column group), lower triangle only
    const int cp_rows = min(BWD_NT, (w + 63) & ~63), cp_ng = BWD_NT/cp_rows, cp_g = tid/cp_rows;
    for(int i = tid - cp_g*cp_rows; i < w && cp_g < cp_ng; i += cp_rows)
      for(int j0 = cp_g; j0 <= i; j0 += 16*cp_ng)
      {
        double v[16];
#pragma unroll
        for(int u = 0; u < 16; u++) v[u] = (j0 + u*cp_ng <= i) ? L[i + (size_t)(j0 + u*cp_ng)*nrows] : 0.0;
#pragma unroll
        for(int u = 0; u < 16; u++) if(j0 + u*cp_ng <= i) Lt[i + (j0 + u*cp_ng)*ldt] = v[u];
      }
  }
  // members of a block-diagonal top: thread = member, its little block in registers
  double Lm[8][8];
  int m0 = 0, nbm = 0;
  if(nmem > 0 && tid < nmem)
  {
    // (rsv of a block-diagonal supernode: the common width of its members, 0 if they differ -- no list
    // lookup, i.e. no dependent load in front of the members' loads)
    if(it.rsv > 0) { m0 = tid*it.rsv; nbm = it.rsv; }
    else
    {
      const int* mcol = sn_bd_col + it.bd0;
      m0 = mcol[tid]; nbm = ((tid + 1 < nmem) ? mcol[tid + 1] : w) - m0;
    }
  }
  if(nmem > 0)
  {
#pragma unroll
    for(int a = 0; a < 8; a++)
#pragma unroll
      for(int b = 0; b <= a; b++) Lm[a][b] = (a < nbm) ? L[(m0 + a) + (size_t)(m0 + b)*nrows] : (a == b ? 1.0 : 0.0);
    if(mm)
    {
#pragma unroll
      for(int a = 0; a < 8; a++) if(a < nbm) { dmin = fmin(dmin, Lm[a][a]); dmax = fmax(dmax, Lm[a][a]); }
    }
  }
  if(mm && (nmem > 0 ? 64*wv < nmem : wv < nblk))
  {
#pragma unroll
    for(int o = 32; o > 0; o >>= 1) { dmin = fmin(dmin, __shfl_down(dmin, o, 64)); dmax = fmax(dmax, __shfl_down(dmax, o, 64)); }
    if(lane == 0) { mm[(8*blockIdx.x + wv)*2] = dmin; mm[(8*blockIdx.x + wv)*2 + 1] = dmax; }
  }
  BW_STAMP(1);
  // (rsv: the supernode's level has LDS room for it, sparse_solve_setup; only where there is a wait to
  // hide the preparation behind: a workgroup of the one-launch region that has a parent -- the root and
  // the workgroups of per-level launches would pay the preparation on their critical path, 7 us against
  // the 3 us the shorter sweep saves)
  const bool premul = it.rsv != 0 && !BD_ONLY && nmem == 0 && pr_flag != nullptr && it.pflag >= 0;
  if(premul)
  {
    // diagonal blocks -> their inverses (in place), then the m's of every block below this thread's own
    __syncthreads();
    double ti[8];
    const bool ton = tid < 8*nblk;
    if(ton)
    {
      const double* Tb = T + (tid >> 3)*64;
      const int c = tid & 7;
      double Lb[8][8];
#pragma unroll
      for(int a = 0; a < 8; a++)
#pragma unroll
        for(int b2 = 0; b2 <= a; b2++) Lb[a][b2] = Tb[a*8 + b2];
#pragma unroll
      for(int i = 0; i < 8; i++)
      {
        double v = 0.0;
#pragma unroll
        for(int k = 0; k < i; k++) v -= (k >= c) ? Lb[i][k]*ti[k] : 0.0;
        ti[i] = (i == c) ? Lb[i][i] : ((i > c) ? v*Lb[i][i] : 0.0);
      }
    }
    __syncthreads();
    if(ton)
    {
      double* Tb = T + (tid >> 3)*64;
      const int c = tid & 7;
#pragma unroll
      for(int i = 0; i < 8; i++) if(i >= c) Tb[i*8 + c] = ti[i];
    }
    __syncthreads();
    if(tid < w)
    {
      const double* Lc = L + (size_t)tid*nrows;            // column tid of L = row tid of L'
      const int wl = w - 1, b0 = (tid >> 3) + 1;           // first block strictly below this thread's row
      double cur[8], nxt[8];
#pragma unroll
      for(int a = 0; a < 8; a++) { cur[a] = Lc[min(8*min(b0, nblk - 1) + a, wl)]; nxt[a] = Lc[min(8*min(b0 + 1, nblk - 1) + a, wl)]; }
      for(int blk = b0; blk < nblk; blk++)
      {
        double far[8];
#pragma unroll
        for(int a = 0; a < 8; a++) far[a] = Lc[min(8*min(blk + 2, nblk - 1) + a, wl)];
        const double* Tb = T + blk*64;
#pragma unroll
        for(int b2 = 0; b2 < 8; b2++)
        {
          double m = Tb[b2*8]*cur[0];
#pragma unroll
          for(int a = 1; a <= b2; a++) m += Tb[b2*8 + a]*cur[a];
          Mx[(blk*8 + b2)*MW + tid] = m;
        }
#pragma unroll
        for(int a = 0; a < 8; a++) { cur[a] = nxt[a]; nxt[a] = far[a]; }
      }
    }
  }
  // ---- round 3
  if(tid < 256) xs[tid] = 0.0;            // columns without below rows (a root) get no mat-vec pass
  const bool by_value = xh_on && xb_lds;          // x of the below rows polled entry by entry (see xh above)
  if(pr_flag && !by_value)
  {
    // everything that does not depend on the ancestors is on its way; now wait for the parent (it
    // waited for its own: all ancestors are done)
    if(tid == 0 && it.pflag >= 0)
    {
      int spins = 0;
      while(__hip_atomic_load(pr_flag + it.pflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != pr_epoch + ho.skew)
      {
        __builtin_amdgcn_s_sleep(1);
        if(++spins > ho.spins) { atomicOr(ho.status, DLG_HANDOFF_SOLVE); break; }    // never hang (the parent is always dispatched first), but say so: x of the ancestors is not there
      }
    }
    __syncthreads();
  }
  if(by_value)
  {
    if(tid < r) xb[tid] = take_xh(myrow);
    for(int i = tid + BWD_NT; i < r; i += BWD_NT) xb[i] = take_xh(rows[w + i]);
  }
  else if(xb_lds)
  {
    if(tid < r) xb[tid] = ldx(ywork + myrow);
    for(int i = tid + BWD_NT; i < r; i += BWD_NT) xb[i] = ldx(ywork + rows[w + i]);
  }
  __syncthreads();
  if(s_skip)
  {
    if(tid < w) put_xh(c0 + tid, 0.0);              // (the children must not wait for values that never come)
    if(pr_flag && tid == 0) __hip_atomic_store(pr_flag + blockIdx.x, pr_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  BW_STAMP(2);
  // The 8x8 diagonal blocks are inverted once, all of them in parallel (thread = (block, column),
  // forward substitution in registers), so that the sequential sweep over the blocks below is eight
  // independent dot products per block instead of a 36-step substitution chain.  The column is
  // written back behind the next barrier (every thread has read the original block by then).
  double tinv[8];
  const bool tinv_on = !BD_ONLY && nmem == 0 && tid < 8*nblk && !premul;
  if(tinv_on)
  {
    const double* Tb = T + (tid >> 3)*64;
    const int c = tid & 7;
    double Lb[8][8];
#pragma unroll
    for(int a = 0; a < 8; a++)
#pragma unroll
      for(int b = 0; b <= a; b++) Lb[a][b] = Tb[a*8 + b];        // diagonal entries are reciprocals already
#pragma unroll
    for(int i = 0; i < 8; i++)
    {
      double v = 0.0;
#pragma unroll
      for(int k = 0; k < i; k++) v -= (k >= c) ? Lb[i][k]*tinv[k] : 0.0;
      tinv[i] = (i == c) ? Lb[i][i] : ((i > c) ? v*Lb[i][i] : 0.0);
    }
  }
  if(mv_thread)
  {
    // partial sums of the mat-vec into xp[part][column]; thread j < w adds them up below
    double acc = 0.0;
#pragma unroll
    for(int q = 0; q < MV_SLOTS; q++) acc += mv[q]*((mv_i0 + q < mv_i1) ? xb[mv_i0 + q] : 0.0);
    if(mv_on) xp[mv_p*256 + mv_j] = acc;
  }
  else
  {
    // many below rows: a wave takes 4 columns at a time, lanes over the rows, all loads of a
    // round in flight together; the sums land in part 0
    for(int e = 256 + tid; e < mv_parts*256; e += BWD_NT) xp[e] = 0.0;     // (part 0 is written in full)
    for(int jg = 4*wv; jg < w; jg += 4*NW)
    {
      const double* Lj = L + (size_t)jg*nrows + w;
      const int nc = min(4, w - jg);
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 2
      for(int i = lane; i < r; i += 64)
      {
        const double x = xb_lds ? xb[i] : ldx(ywork + rows[w + i]);
#pragma unroll
        for(int c = 0; c < 4; c++) acc[c] += ((c < nc) ? Lj[i + (size_t)c*nrows] : 0.0)*x;
      }
#pragma unroll
      for(int c = 0; c < 4; c++)
      {
        const double sum = wave_sum(acc[c]);
        if(lane == 0 && c < nc) xp[jg + c] = sum;
      }
    }
  }
  __syncthreads();
  BW_STAMP(3);
  if(tinv_on)
  {
    double* Tb = T + (tid >> 3)*64;
    const int c = tid & 7;
#pragma unroll
    for(int i = 0; i < 8; i++) if(i >= c) Tb[i*8 + c] = tinv[i];       // Tb = (diagonal block)^-1, lower triangle
  }
  if(nmem > 0)
  {
    // (nmem <= BWD_NT: a member has at least one column and w <= 256)
    if(tid < w)
    {
      double sum = 0.0;
      for(int p = 0; p < mv_parts; p++) sum += xp[p*256 + tid];
      xs[tid] = myrhs - sum;
    }
    __syncthreads();
    if(LEAN)
    {
      // rolled loops, the solved unknowns stay in xs: few registers, the latency is hidden by the
      // other workgroups of the CU
      if(tid < nmem)
        for(int a = nbm - 1; a >= 0; a--)
        {
          const double* Lc = L + (size_t)(m0 + a)*nrows + m0;
          double v = xs[m0 + a];
          for(int b = a + 1; b < nbm; b++) v -= Lc[b]*xs[m0 + b];
          xs[m0 + a] = v/Lc[a];
        }
    }
    else if(tid < nmem)
    {
      double xk[8];
#pragma unroll
      for(int a = 7; a >= 0; a--)
      {
        double v = (a < nbm) ? xs[m0 + a] : 0.0;
#pragma unroll
        for(int b = a + 1; b < 8; b++) v -= Lm[b][a]*xk[b];
        xk[a] = v/Lm[a][a];
      }
#pragma unroll
      for(int a = 0; a < 8; a++) if(a < nbm) xs[m0 + a] = xk[a];
    }
    __syncthreads();
    if(tid < w)
    {
      put_xh(c0 + tid, xs[tid]);
      if(pr_flag) __hip_atomic_store((gd_t)(ywork + c0 + tid), xs[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else ywork[c0 + tid] = xs[tid];
      out[myperm] = xs[tid];
    }
    if(pr_flag)
    {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if(tid == 0) __hip_atomic_store(pr_flag + blockIdx.x, pr_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  if(BD_ONLY) return;
  // only the waves that hold a row of the diagonal block take part in the substitution: the others
  // leave, and the barriers below (one per 8 columns) are among one or two waves instead of all
  if(tid >= ((8*nblk + 63) & ~63)) return;
  double xi = 0.0;
  if(tid < w)
  {
    double sum = 0.0;
    for(int p = 0; p < mv_parts; p++) sum += xp[p*256 + tid];
    xi = myrhs - sum;
  }
  // The sweep over the blocks is bound by instruction issue (one or two waves, ~4.5 cycles per
  // instruction): the body is kept branch-free and short.  Operands come from the LDS copy of the
  // top block when the level has room for it, else from HBM two blocks ahead with clamped,
  // unconditional loads; every thread computes the 8 unknowns of a block itself (dot products with
  // the inverted diagonal block) and ONE thread parks them in xs -- the owners pick them up at the end.
  if(premul)
  {
    const int own = tid & 7;
    for(int blk = nblk - 1; blk >= 0; blk--)
    {
      const int j0 = 8*blk;
      double* rh = rhs + 8*(blk & 1);
      if(tid >= j0 && tid < j0 + 8) rh[tid - j0] = xi;
      // the thread's eight operands do not depend on rh: their loads are in flight while rh is published
      // (a thread above the block: its pre-multiplied column; an owner -- and, harmlessly, a thread whose
      // unknown is already out -- column `own` of the inverted diagonal block, zeros above the diagonal)
      const double* Mb = (tid < j0) ? Mx + (size_t)blk*8*MW + tid : T + blk*64 + own;
      const int mst = (tid < j0) ? MW : 8;
      double mb[8];
#pragma unroll
      for(int b2 = 0; b2 < 8; b2++) mb[b2] = Mb[b2*mst];
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      double r8[8];
#pragma unroll
      for(int b2 = 0; b2 < 8; b2 += 2)
      {
        const double2 rr = *reinterpret_cast<const double2*>(rh + b2);
        r8[b2] = rr.x; r8[b2 + 1] = rr.y;
      }
      double v = mb[0]*r8[0];
#pragma unroll
      for(int b2 = 1; b2 < 8; b2++) v += mb[b2]*r8[b2];
      if(tid < j0) xi -= v;
      else if(tid < j0 + 8) xs[min(tid, 255)] = v;   // an owner: x(j0 + own) = sum_b inv(b, own) rh(b)
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if(tid < w) xi = xs[tid];
    BW_STAMP(4);
    if(tid < w)
    {
      put_xh(c0 + tid, xi);
      if(pr_flag) __hip_atomic_store((gd_t)(ywork + c0 + tid), xi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else ywork[c0 + tid] = xi;
      out[myperm] = xi;
    }
    if(pr_flag)
    {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if(tid == 0) __hip_atomic_store(pr_flag + blockIdx.x, pr_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    BW_STAMP(5);
    return;
  }
  const double* Lcol = L + (size_t)min(tid, w - 1)*nrows;      // column tid of L = row tid of L^T
  const double* Ltc = Lt + min(tid, w - 1)*ldt;
  const int wlast = w - 1;
  double lv[8], ln[8];
#pragma unroll
  for(int a = 0; a < 8; a++) { lv[a] = 0.0; ln[a] = 0.0; }
  if(!top_lds)
  {
    const int j0 = 8*(nblk - 1), j1 = max(j0 - 8, 0);
#pragma unroll
    for(int a = 0; a < 8; a++) { lv[a] = Lcol[min(j0 + a, wlast)]; ln[a] = Lcol[min(j1 + a, wlast)]; }
  }
  for(int blk = nblk - 1; blk >= 0; blk--)
  {
    const int j0 = 8*blk;
    double* rh = rhs + 8*(blk & 1);
    if(tid >= j0 && tid < j0 + 8) rh[tid - j0] = xi;
    double lf[8];                          // two blocks ahead (HBM path)
    if(top_lds)
    {
#pragma unroll
      for(int a = 0; a < 8; a++) lv[a] = Ltc[min(j0 + a, wlast)];
    }
    else
    {
      const int j2 = max(j0 - 16, 0);
#pragma unroll
      for(int a = 0; a < 8; a++) lf[a] = Lcol[min(j2 + a, wlast)];
    }
    // the barrier only has to publish rh (LDS): __syncthreads() would also wait for the global loads
    // in flight for the blocks ahead (s_waitcnt vmcnt(0)), i.e. pay a memory latency per block
    if(blk == nblk - 1) BW_STAMP(6);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if(blk == nblk - 1) BW_STAMP(7);
    const double* Tb = T + blk*64;
    double xk[8];
    // x_blk = (block^-1)' rhs: eight independent dot products (rows past the end: rh = 0 -> 0)
#pragma unroll
    for(int a = 0; a < 8; a++)
    {
      double v = Tb[a*8 + a]*rh[a];
#pragma unroll
      for(int b = a + 1; b < 8; b++) v += Tb[b*8 + a]*rh[b];
      xk[a] = v;
    }
    if(tid == j0)
    {
#pragma unroll
      for(int a = 0; a < 8; a++) xs[min(j0 + a, 255)] = xk[a];
    }
    double upd = 0.0;
#pragma unroll
    for(int a = 0; a < 8; a++) upd += lv[a]*xk[a];
    xi -= (tid < j0) ? upd : 0.0;
    if(!top_lds)
    {
#pragma unroll
      for(int a = 0; a < 8; a++) { lv[a] = ln[a]; ln[a] = lf[a]; }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if(tid < w) xi = xs[tid];
  BW_STAMP(4);
  if(tid < w)
  {
    put_xh(c0 + tid, xi);
    if(pr_flag) __hip_atomic_store((gd_t)(ywork + c0 + tid), xi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else ywork[c0 + tid] = xi;
    out[myperm] = xi;
  }
  if(pr_flag)
  {
    // every storing wave drains its stores, then (behind a barrier of the waves still here) the flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if(tid == 0) __hip_atomic_store(pr_flag + blockIdx.x, pr_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  BW_STAMP(5);
}

// read a range and drop it: pulls the panels of the leaves into the Infinity Cache while the top of
// the tree is being factored (sparse_touch_factor)
__global__ void __launch_bounds__(TPB) k_touch(const double2* __restrict__ p, size_t n2, double* __restrict__ sink)
{
  double acc = 0.0;
  for(size_t i = blockIdx.x*(size_t)TPB + threadIdx.x; i < n2; i += (size_t)gridDim.x*TPB)
  { const double2 v = p[i]; acc += v.x + v.y; }
  if(acc == 1.2345678e300) sink[0] = acc;          // (never: keeps the loads)
}
// (extra: a device scalar that rides behind the vector, at v[n], through the sum over the ranks)
__global__ void __launch_bounds__(TPB) k_mask_vec(double* __restrict__ v, const double* __restrict__ mask, int n,
                                                  const double* __restrict__ extra)
{
  const int i = blockIdx.x*TPB + threadIdx.x;
  if(i < n) v[i] *= mask[i];
  if(extra && i == 0) v[n] = *extra;
}

} // namespace

// per-level launch parameters of the solve kernels
static long env_int_solve(const char* n, long d) { const char* v = getenv(n); return v ? atol(v) : d; }
#ifdef DLG_FL_PROFILE
extern "C" void dlg_bw_profile_dump(int nlevels)
{
  long long h[64*8];
  hipDeviceSynchronize();
  {
    std::vector<long long> c(256*8);
    hipMemcpyFromSymbol(c.data(), HIP_SYMBOL(g_bw_chain), sizeof(long long)*c.size());
    long long t0 = 0;
    for(int g = 0; g < 256; g++) if(c[g*8] && (t0 == 0 || c[g*8] < t0)) t0 = c[g*8];
    for(int g = 0; g < 256 && g < (getenv("DLG_FL_DUMP_ALL") ? 256 : 24); g++)
    {
      const long long* q = &c[g*8];
      if(q[5] == 0) continue;
      fprintf(stderr, "   bwd wg %3d: start %6lld loads issued %6lld parent there + x gathered %6lld mat-vec %6lld sweep %6lld published %6lld (10 ns)\n",
              g, q[0] - t0, q[1] - t0, q[2] - t0, q[3] - t0, q[4] - t0, q[5] - t0);
    }
  }
  hipMemcpyFromSymbol(h, HIP_SYMBOL(g_bw_prof), sizeof(h));
  for(int l = 0; l < nlevels && l < 64; l++)
    fprintf(stderr, "bwd level %2d: issue %6lld  gather+barrier %6lld  matvec %6lld  solve %6lld  store %6lld cycles | to first barrier %6lld, in it %6lld, rest of the loop %6lld\n", l,
            h[l*8+1] - h[l*8], h[l*8+2] - h[l*8+1], h[l*8+3] - h[l*8+2], h[l*8+4] - h[l*8+3], h[l*8+5] - h[l*8+4],
            h[l*8+6] - h[l*8+3], h[l*8+7] - h[l*8+6], h[l*8+4] - h[l*8+7]);
}
#endif
// Second stream, behind the Cauchy step's pass over J: the leaf panels (final since the leaf level was
// factored) are read once so that the backward solve of the leaves, half a millisecond later, finds
// them in the 256 MiB Infinity Cache instead of HBM.  A hint: nothing waits for it.
int sparse_touch_factor(dlg_backend* b, hipStream_t st)
{
  SparseSym* Y = b->sym;
  if(!Y || !Y->touch_n) return DLG_OK;
  const int nwg = b->knobs.touch_wg;   // (1024: 9 us more in the factorisation; 16: still running when the next step needs the stream)
  hipLaunchKernelGGL(k_touch, dim3(nwg), dim3(TPB), 0, st, reinterpret_cast<const double2*>(Y->Lx + Y->touch_off),
                     (size_t)Y->touch_n/2, Y->ywork);
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}
int sparse_solve_setup(dlg_backend* b)
{
  SparseSym* Y = b->sym;
  const SymHost& H = Y->H;
  // below rows of a supernode that the backward solve stages in LDS (96 KB of x); beyond that it gathers x from HBM
  const long xb_cap = env_int_solve("DOGLEG_AMD_BWD_XB_CAP", 12288);
  Y->bwd_xb_cap = (int)xb_cap;
  Y->slv_lds.assign(H.nlevels, 0); Y->bwd_lds.assign(H.nlevels, 0); Y->bwd_nt.assign(H.nlevels, 512); Y->bwd_top.assign(H.nlevels, 0); Y->bwd_bd.assign(H.nlevels, 0); Y->bwd_pmx.assign(H.nlevels, 0);
  for(int l = 0; l < H.nlevels; l++)
  {
    long maxw = 0, mb = 0, mbt = 0, wmax_all = 0;
    for(int i = H.lvl_ptr[l]; i < H.lvl_ptr[l+1]; i++)
    {
      const int s = H.lvl_sn[i];
      const long wv = H.sn_c0[s+1] - H.sn_c0[s], nr = H.sn_rowptr[s+1] - H.sn_rowptr[s];
      if(wv > maxw) maxw = wv;
      wmax_all = std::max(wmax_all, wv);
      // xb (x at the below rows, unless there are more of them than xb_cap: those supernodes gather x
      // from HBM), xs, diagonal blocks, rhs, mat-vec partial sums
      const long rb = nr - wv - 1;
      const long need = (rb <= xb_cap ? rb + 3 : 0) + 256 + ((wv + 7)/8)*64 + 16 + 8*256;
      if(need > mb) mb = need;
      // + the top block, for the supernodes that read it (not the block-diagonal ones)
      if(H.sn_bd_ptr[s+1] == H.sn_bd_ptr[s]) mbt = std::max(mbt, need + wv*(wv | 1));
    }
    mbt = std::max(mbt, mb);
    if(wmax_all > 256) { dlg_set_error("supernode of width %ld is too wide for the backward-solve kernel", wmax_all); return DLG_ERR_ARG; }
    Y->slv_lds[l] = (int)(maxw*(maxw | 1)*8);
    if(Y->slv_lds[l] > LDS_BUDGET) { dlg_set_error("supernode of width %ld is too wide for the solve kernels", maxw); return DLG_ERR_ARG; }
    if(mb*8 > LDS_BUDGET) { dlg_set_error("supernode too large for the backward-solve kernel (%ld doubles)", mb); return DLG_ERR_ARG; }
    // the top block rides in LDS when every supernode of the level has room for it
    Y->bwd_top[l] = 0;     // (the top block staged in LDS was measured slower than its prefetch from HBM, rounds 2 - 5: k_solve_bwd_level keeps the path, nothing selects it)
    Y->bwd_lds[l] = (int)((Y->bwd_top[l] ? mbt : mb)*8);
    Y->bwd_pmx[l] = 0;
    {
      // room for the pre-multiplied operands of the block sweep (k_solve_bwd_level: premul), [nblk][8][w16] per supernode
      long pm = 0;
      for(int i = H.lvl_ptr[l]; i < H.lvl_ptr[l+1]; i++)
      {
        const int s = H.lvl_sn[i];
        const long wv = H.sn_c0[s+1] - H.sn_c0[s];
        if(H.sn_bd_ptr[s+1] == H.sn_bd_ptr[s]) pm = std::max(pm, ((wv + 7)/8)*8*((wv + 15) & ~15L));
      }
      if(pm > 0 && !getenv("DOGLEG_AMD_NO_PREMUL") && (long)Y->bwd_lds[l] + pm*8 <= LDS_BUDGET) Y->bwd_pmx[l] = (int)(pm*8);
    }
    // thread = row of the diagonal block: 256 threads when every supernode of a populous level is
    // narrow (more workgroups per CU), else 512
    Y->bwd_nt[l] = (maxw <= 128 && H.lvl_ptr[l+1] - H.lvl_ptr[l] >= 512) ? 256 : 512;
    // every supernode of the level has a block-diagonal top: the leaner kernel variant
    bool allbd = true;
    for(int i = H.lvl_ptr[l]; i < H.lvl_ptr[l+1]; i++) { const int s = H.lvl_sn[i]; if(H.sn_bd_ptr[s+1] == H.sn_bd_ptr[s]) allbd = false; }
    Y->bwd_bd[l] = allbd ? 1 : 0;
    if(Y->bwd_nt[l] != 512 || Y->bwd_bd[l]) Y->bwd_pmx[l] = 0;
    Y->bwd_lds[l] += Y->bwd_pmx[l];
  }
  {
    // (the supernodes this rank works on: all of them, or its subtrees + the replicated top)
    std::vector<SolveItem> items(H.xl_sn.size());
    for(int k = 0; k < (int)H.xl_sn.size(); k++)
    {
      const int s = H.xl_sn[k];
      SolveItem& it = items[k];
      it.c0 = H.sn_c0[s]; it.w = H.sn_c0[s+1] - H.sn_c0[s]; it.nrows = H.sn_rowptr[s+1] - H.sn_rowptr[s];
      it.rowoff = H.sn_rowptr[s]; it.lx = H.sn_lx[s]; it.bd0 = H.sn_bd_ptr[s]; it.nbd = H.sn_bd_ptr[s+1] - H.sn_bd_ptr[s];
      it.pflag = -1; it.rsv = Y->bwd_pmx[H.sn_level[s]] ? 1 : 0;
      if(it.nbd > 0)
      {
        // block-diagonal top: rsv = the common width of the members (0: they differ)
        const int* mc = &H.sn_bd_col[it.bd0];
        const int wd0 = (it.nbd > 1 ? mc[1] : it.w) - mc[0];
        bool same = mc[0] == 0;
        for(int m = 0; m < it.nbd && same; m++) if(((m + 1 < it.nbd) ? mc[m+1] : it.w) - mc[m] != wd0 || mc[m] != m*wd0) same = false;
        it.rsv = same ? wd0 : 0;
      }
    }
    DLG_CHECK(upload(Y->slv_item, items)); Y->allocs.push_back(Y->slv_item);
    // smallest / largest diagonal entry of L per supernode and wave, left behind by every backward solve (k_solve_bwd_level:
    // mm); a wave without diagonal entries never writes its pair: (huge, 0) changes no minimum and no maximum
    std::vector<double> none(16*std::max<size_t>(items.size(), 1));
    for(size_t i = 0; i < none.size(); i += 2) { none[i] = 1e300; none[i + 1] = 0.0; }
    DLG_CHECK(upload(Y->diag_mm, none)); Y->allocs.push_back(Y->diag_mm);
    Y->n_diag_mm = 8*(int)items.size();
  }
  // Persistent top region of the backward solve: the last levels of the tree hold a few supernodes each
  // and every level waits for the one above.  They go out as ONE launch, workgroups ordered from the
  // root down (a workgroup only waits for a lower-numbered one): a workgroup pulls its part of L
  // while it waits for its parent's flag, and raises its own as soon as its x is out.  One workgroup
  // per CU (the hand-off through write-through stores and loads around L1 is measured for that):
  // the launch asks for more than half of the LDS.
  {
    // the contiguous stretch of Lx that holds the panels of the lowest level (even offset: 16-byte loads)
    int64_t lo = -1, hi = 0;
    for(int i = H.lvl_ptr[0]; i < H.lvl_ptr[1]; i++)
    {
      const int s = H.lvl_sn[i];
      const int64_t a = H.sn_lx[s], e = a + (int64_t)(H.sn_rowptr[s+1] - H.sn_rowptr[s])*(H.sn_c0[s+1] - H.sn_c0[s]);
      if(lo < 0 || a < lo) lo = a;
      if(e > hi) hi = e;
    }
    Y->touch_off = 0; Y->touch_n = 0;
    if(lo >= 0 && H.nlevels >= 3 && hi - lo >= (1 << 20) && (hi - lo)*8 <= (int64_t)200 << 20)
    { Y->touch_off = lo & ~(int64_t)1; Y->touch_n = (hi - Y->touch_off) & ~(int64_t)1; }
  }
  Y->bw_level0 = H.nlevels; Y->bw_n = 0;
  if(!getenv("DOGLEG_AMD_NO_PERSIST") && H.nlevels >= 2)
  {
    int ncu = 256;
    { int dev = 0; if(hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev); }
    const int cap = (int)env_int_solve("DOGLEG_AMD_PERSIST_MAX", 2*ncu);
    int total = 0, l0 = H.nlevels, ldsb = 0;
    for(int l = H.nlevels - 1; l >= 1; l--)
    {
      const int n = H.xl_ptr[l+1] - H.xl_ptr[l];
      if(n == 0 || total + n > cap || Y->bwd_nt[l] != 512 || Y->bwd_bd[l] || Y->bwd_top[l]) break;
      total += n; l0 = l; ldsb = std::max(ldsb, Y->bwd_lds[l]);
    }

    if(H.nlevels - l0 >= 2)
    {
      std::vector<SolveItem> items;
      std::vector<int> pos(H.nsn, -1);
      for(int l = H.nlevels - 1; l >= l0; l--)
        for(int k = H.xl_ptr[l]; k < H.xl_ptr[l+1]; k++)
        {
          const int s = H.xl_sn[k];
          SolveItem it;
          it.c0 = H.sn_c0[s]; it.w = H.sn_c0[s+1] - H.sn_c0[s]; it.nrows = H.sn_rowptr[s+1] - H.sn_rowptr[s];
          it.rowoff = H.sn_rowptr[s]; it.lx = H.sn_lx[s]; it.bd0 = H.sn_bd_ptr[s]; it.nbd = H.sn_bd_ptr[s+1] - H.sn_bd_ptr[s];
          it.rsv = Y->bwd_pmx[l] ? 1 : 0; it.pflag = -1;
          if(it.nbd > 0) it.rsv = 0;          // (a block-diagonal top in the region: rsv would mean the members' width; 0 = look it up)
          // the parent: the supernode of the first below row (the last row is the augmented one)
          if(it.nrows - it.w - 1 > 0)
          {
            const int prow = H.sn_rows[H.sn_rowptr[s] + it.w];
            const int ps = H.col_sn[prow];
            it.pflag = pos[ps];
            if(it.pflag < 0) { dlg_set_error("internal error: the parent of supernode %d is not part of the persistent backward launch", s); return DLG_ERR_ARG; }
          }
          pos[s] = (int)items.size();
          items.push_back(it);
        }
      Y->bw_level0 = l0; Y->bw_n = (int)items.size(); Y->bw_lds = std::max(ldsb, 84*1024);
      DLG_CHECK(upload(Y->slv_item_pr, items)); Y->allocs.push_back(Y->slv_item_pr);
      // x of the region as its own signal (k_solve_bwd_level, xh): two sets, every entry a sentinel until it is stored
      if(true)
      {
        std::vector<unsigned long long> empty(2*(size_t)H.N, 0x7FF8DEADBEEF0002ull);
        DLG_HIP(hipMalloc(&Y->bwd_xh, sizeof(double)*empty.size())); Y->allocs.push_back(Y->bwd_xh);
        DLG_HIP(hipMemcpy(Y->bwd_xh, empty.data(), sizeof(double)*empty.size(), hipMemcpyHostToDevice));
      }
      DLG_HIP(hipMalloc(&Y->bwd_flag, sizeof(int)*items.size())); Y->allocs.push_back(Y->bwd_flag);
      DLG_HIP(hipMemsetAsync(Y->bwd_flag, 0, sizeof(int)*items.size(), b->stream));
      Y->bwd_epoch = 0;
    }
  }
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_fwd_level),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET));
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_bwd_level<256, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET));
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_bwd_level<256, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET));
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_bwd_level<512, false>),
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
  if(H.part_nranks > 1 && !use_aug)
  { dlg_set_error("with a subtree partition only the right-hand side that rode along with the factorisation (Jt x) can be solved for"); return DLG_ERR_STATE; }
  for(int l = 0; l < H.nlevels && !use_aug; l++)
  {
    const int n = H.xl_ptr[l+1] - H.xl_ptr[l];
    if(n > 0)
      hipLaunchKernelGGL(k_solve_fwd_level, dim3(n), dim3(TPB), Y->slv_lds[l], st, Y->xl_sn + H.xl_ptr[l],
                         Y->sn_c0, Y->sn_rowptr, Y->sn_lx, Y->sn_scr, Y->rl_ptr, Y->rl_pos, Y->perm,
                         Y->Lx, rhs, Y->scr, Y->ywork);
  }
  // subtree partition: the entries of the other ranks' variables are zero here, and the replicated
  // ones count on rank 0 only: the solution is the sum over the ranks
  if(H.part_nranks > 1) DLG_HIP(hipMemsetAsync(out, 0, sizeof(double)*(size_t)H.N, st));
  int ltop = H.nlevels - 1;
  if(Y->bw_level0 < H.nlevels)
  {
    // the persistent top region: its levels in one launch, workgroups from the root down (sparse_solve_setup)
    DlgRegionTurn turn(b);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_solve_bwd_level<512, false>), dim3(Y->bw_n), dim3(512), Y->bw_lds, st,
                       Y->slv_item_pr, Y->sn_rows, Y->perm, Y->Lx, Y->ywork, out, use_aug, Y->sn_bd_col,
                       256*Y->bw_level0, Y->bwd_xb_cap, Y->bwd_flag, ++Y->bwd_epoch, Y->d_info, dlg_handoff(b, 1 << 21), Y->bwd_xh, H.N,
                       (Y->diag_mm && !b->knobs.ei_jpass) ? Y->diag_mm + 16*(size_t)H.xl_ptr[Y->bw_level0] : (double*)nullptr);
    ltop = Y->bw_level0 - 1;
  }
  for(int l = ltop; l >= 0; l--)
  {
    const int n = H.xl_ptr[l+1] - H.xl_ptr[l];
    // thread = row of the diagonal block: 256 threads when every supernode of a populous level is
    // narrow (more workgroups per CU), else 512
    if(n > 0 && Y->bwd_nt[l] == 256 && Y->bwd_bd[l])
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_solve_bwd_level<256, true>), dim3(n), dim3(256), Y->bwd_lds[l], st,
                         Y->slv_item + H.xl_ptr[l], Y->sn_rows, Y->perm, Y->Lx, Y->ywork, out, use_aug, Y->sn_bd_col,
                         Y->bwd_top[l] + 256*l, Y->bwd_xb_cap, (int*)nullptr, 0, Y->d_info, dlg_handoff(b, 1 << 21), (double*)nullptr, 0,
                         (Y->diag_mm && !b->knobs.ei_jpass) ? Y->diag_mm + 16*(size_t)H.xl_ptr[l] : (double*)nullptr);
    else if(n > 0 && Y->bwd_nt[l] == 256)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_solve_bwd_level<256, false>), dim3(n), dim3(256), Y->bwd_lds[l], st,
                         Y->slv_item + H.xl_ptr[l], Y->sn_rows, Y->perm, Y->Lx, Y->ywork, out, use_aug, Y->sn_bd_col,
                         Y->bwd_top[l] + 256*l, Y->bwd_xb_cap, (int*)nullptr, 0, Y->d_info, dlg_handoff(b, 1 << 21), (double*)nullptr, 0,
                         (Y->diag_mm && !b->knobs.ei_jpass) ? Y->diag_mm + 16*(size_t)H.xl_ptr[l] : (double*)nullptr);
    else if(n > 0)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_solve_bwd_level<512, false>), dim3(n), dim3(512), Y->bwd_lds[l], st,
                         Y->slv_item + H.xl_ptr[l], Y->sn_rows, Y->perm, Y->Lx, Y->ywork, out, use_aug, Y->sn_bd_col,
                         Y->bwd_top[l] + 256*l, Y->bwd_xb_cap, (int*)nullptr, 0, Y->d_info, dlg_handoff(b, 1 << 21), (double*)nullptr, 0,
                         (Y->diag_mm && !b->knobs.ei_jpass) ? Y->diag_mm + 16*(size_t)H.xl_ptr[l] : (double*)nullptr);
  }
  DLG_LAUNCH_CHECK();
  if(H.part_nranks > 1)
  {
    // a scalar the caller wants summed over the ranks (the Cauchy step's |J g|^2 of the rank's rows, formed on the
    // second stream beside the factorisation) rides behind the solution: one collective less per step.
    // (only into the Gauss-Newton vectors of the slots: they have room behind their N entries)
    const bool room = out == b->slot[0].gn || out == b->slot[1].gn;
    const double* extra = room ? b->fold_scalar : nullptr;
    hipLaunchKernelGGL(k_mask_vec, dim3(dlg_cdiv(H.N, TPB)), dim3(TPB), 0, st, out, Y->colmask, H.N, extra);
    DLG_LAUNCH_CHECK();
    DLG_CHECK(dlg_allreduce_dev(b, out, (size_t)H.N + (extra ? 1 : 0)));
    if(extra) { b->fold_result = out + H.N; b->fold_scalar = nullptr; }
  }
  return DLG_OK;
}

