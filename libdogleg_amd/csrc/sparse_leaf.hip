// sparse_leaf.hip -- leaf fronts: K4 (JtJ), K1 (Jt*x) and the leaf level of K5 in ONE pass over J.
//
// Every measurement row belongs to the supernode of its first-eliminated variable; its variables are a
// clique of JtJ, so the whole outer product of the row lands in that supernode's front.  Where every
// row is owned by a merged leaf of the elimination tree (the points of a bundle adjustment: SymHost::lf_ok)
// a workgroup per leaf
//   1. stages the leaf's rows in LDS (values + x of the row at position 15), coalesced, J read ONCE and
//      without its index array (the schedule says where everything goes);
//   2. forms the front there: the members' diagonal blocks and the rows below them (the leaf's panel), and
//      the rows' direct contributions to the blocks of the ancestors (cameras x cameras, ...), which start
//      the leaf's update matrix -- "strip tasks", one wave each (sparse_symbolic.h, LfLeaf);
//   3. eliminates the members (the arithmetic of the leaf level of k_factor_level: bd_compact_*), forms
//      U = B B' on the matrix cores and leaves  U - (direct contributions)  as the leaf's update matrix;
//   4. writes the FINAL panel, the update matrix, Jt*x of the members and the leaf's share of Jt*x of the
//      ancestors' variables (summed by k_jtx_fin2_*).
// Replaces, for such patterns, k_assemble_mfma (265 MB of panels written, read back and rewritten on
// config #4) and k_factor_level<256, true>; the update gather and everything above the leaves are unchanged:
// the panels of the ancestors start from zero and receive every contribution through the update matrices.
// Reference: cholmod_factorize on A A' (dogleg.c:659-664), mul_spmatrix_densevector (dogleg.c:249-261).
#include "sparse_internal.h"
#include "panel_factor.h"
#include "factor_tail.h"

namespace {
constexpr int LF_NT = 1024;
constexpr int LF_RS = 16;          // doubles per staged row: the row's entries (<= 15), x of the row at 15
constexpr int LF_Q = 8;            // rows per thread group and staging pass (64 rows per pass and workgroup)

#ifdef DLG_LF_PROFILE
__device__ long long g_lf_prof[4096*8];
#define LF_STAMP(k) do { if(threadIdx.x == 0 && leaf < 4096) g_lf_prof[leaf*8 + (k)] = wall_clock64(); } while(0)
__device__ long long g_lf_wave[16*64];      // workgroup 300: per wave, cycle stamps around its tasks' parts
#define LF_WSTAMP(i) do { if(leaf == 300 && (threadIdx.x & 63) == 0 && wslot < 60) { g_lf_wave[(threadIdx.x >> 6)*64 + wslot] = ((long long)(i) << 56) | (clock64() & 0xffffffffffffffLL); wslot++; } } while(0)
#else
#define LF_STAMP(k)
#define LF_WSTAMP(i)
#endif

// member block of NB columns from its accumulated entries in Dg (row c, column q at Dg[(c0 + c)*4 + q]):
// factored in registers, left in Dg for the rows' solves and written to the panel's top block
template <int NB>
__device__ __forceinline__ int lf_factor_member(double* Dg, int c0, double* rdiag, double lambda)
{
  double D[NB][NB];
#pragma unroll
  for(int c = 0; c < NB; c++)
#pragma unroll
    for(int q = 0; q <= c; q++) D[c][q] = Dg[(c0 + c)*4 + q];
#pragma unroll
  for(int c = 0; c < NB; c++) D[c][c] += lambda;
  int badcol = -1;
#pragma unroll
  for(int c = 0; c < NB; c++)
  {
    double d = D[c][c];
    if(!(d > 0.0)) { if(badcol < 0) badcol = c; d = 1.0; }
    const double inv = dlg_rsqrt(d);
    rdiag[c0 + c] = inv;
#pragma unroll
    for(int i = c + 1; i < NB; i++) D[i][c] *= inv;
#pragma unroll
    for(int j = c + 1; j < NB; j++)
#pragma unroll
      for(int i = j; i < NB; i++) D[i][j] -= D[i][c]*D[j][c];
    D[c][c] = d*inv;
  }
#pragma unroll
  for(int c = 0; c < NB; c++)
#pragma unroll
    for(int q = 0; q <= c; q++) Dg[(c0 + c)*4 + q] = D[c][q];       // (to the panel's top block with the rows below: k_leaf_front, part 5)
  return badcol;
}

__device__ __forceinline__ int lf_sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

// where a finished strip entry goes.  kind 0: the leaf's panel (rows below: P[f + column*ldp]; the member's own
// rows: its diagonal block in Dg); kind 1: the packed update matrix (row >= column) or, position 15, the Jt*x record
__device__ __forceinline__ void lf_put(double acc, int kind, int pdk, int k, int j, int wj, int kj, int col0,
                                       double* P, int ldp, double* Dg, double* Us, int mb, double* jtp_rec)
{
  if(j >= wj || pdk == 0xFF || pdk == 0xFE) return;
  if(kind == 0)
  {
    if(pdk == 0xFD) { const int c = k - kj; if(c >= j) Dg[(col0 + c)*4 + j] = acc; }
    else P[pdk + (col0 + j)*ldp] = acc;
  }
  else
  {
    const int c = col0 + j;
    if(pdk == 0xFC) { if(jtp_rec) jtp_rec[c] = acc; }
    else if(pdk >= c) Us[tri_col(c, mb) + pdk] = acc;
  }
}

// all of a workgroup's LDS traffic done, then the barrier -- WITHOUT waiting for its global loads (PF: the next leaf's
// rows are on their way across the phases of this one; __syncthreads would wait for them at every barrier)
__device__ __forceinline__ void lf_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// gfx950's 16-byte copy from memory straight into LDS (buffer_load_dwordx4 ... lds): every lane fetches 16 bytes at
// its own offset into the buffer (out of range: zeros), the wave's 64 x 16 bytes land lane after lane behind the LDS
// address in M0.  No destination registers, nothing for the compiler to wait for or to spill: the issuing code
// decides when to wait (s_waitcnt vmcnt).  Checked on its own in tools/micro/test_lds_dma.hip.
typedef int lf_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ lf_v4i lf_rsrc(const void* base, unsigned bytes)
{
  const uint64_t a = (uint64_t)base;
  return (lf_v4i){ (int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xffff), (int)bytes, 0x00020000 };
}
__device__ __forceinline__ void lf_dma16(lf_v4i rsrc, unsigned lds_addr, unsigned voff)
{
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory", "m0");
}
__device__ __forceinline__ unsigned lf_lds_addr(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p; }

// PF = false: a workgroup per leaf; record, staging table and schedule are fetched at once (the schedules lie at one
// stride), the rows behind the table: two dependent round trips to memory in front of the first barrier, 5 us of the
// 19 a leaf takes (and every CU fetches at the same time: the chip's memory idles while they all compute).
// PF = true: one workgroup per CU, leaf after leaf; while leaf i is eliminated the rows and the schedule of leaf
// i + (number of workgroups) are COPIED INTO LDS by the copy engine of the texture path (lf_dma16) -- the rows' part of
// LDS is free from the end of the strip tasks on, the schedule's from the end of the riders' sums.  One LDS layout for all
// leaves (sizes from the symbolic phase).  The table of the next leaf (where its rows are) is fetched during the strip
// tasks into four registers a lane.  x of the rows (position 15 of a slot) cannot come with the copy (16-byte granules
// land lane after lane): one register a thread, stored at the top of the next round.
template <bool PF>
__global__ void __launch_bounds__(LF_NT, 1) k_leaf_front(const LfLeaf* __restrict__ leaves, int nleaf,
                                                         const uint8_t* __restrict__ blob_g, int b_stride, int b_smax, int b_tb,
                                                         int pf_b, int pf_pud, unsigned jv_bytes, unsigned blob_bytes,
                                                         const int* __restrict__ perm,
                                                         const double* __restrict__ Jv,
                                                         const double* __restrict__ x,
                                                         const double* __restrict__ rhs, double lambda,
                                                         double* __restrict__ Lx, double* __restrict__ uscr,
                                                         double* __restrict__ Jt_x, double* __restrict__ jtp,
                                                         int* __restrict__ lf_word)
{
  extern __shared__ __attribute__((aligned(16))) double LDS0[];
  __shared__ int s_mcol[68];
  __shared__ double s_rdiag[64];
  __shared__ int sbad;
  __shared__ double s_jo[64];                        // Jt*x of the leaf's own columns (the right-hand side row before it is solved)
  __shared__ double s_jt[240];                      // the leaf's shares of Jt*x of the ancestors' variables (out with the panel)
  constexpr int NT = LF_NT, NW = NT/64;
  const int tid0 = threadIdx.x;
  double* const P = PF ? LDS0 + (pf_b >> 3) : LDS0;
  // ---- PF: what is fetched ahead for the next leaf
  int n_w[4] = {0, 0, 0, 0}, n_xr = 0, n_pvar = 0, n_nsl = 0, n_lb = 0;      // (n_nsl, n_lb: its rows and the bytes of its schedule)
  double n_x = 0.0;
  const int o_s = (10*b_smax + 3) & ~3;                       // the by-slot tables inside a blob
  auto pf_table = [&](int li, int tid, int lane, int wv) {
    const uint8_t* blob = blob_g + (size_t)li*b_stride;
    const int32_t* svs = reinterpret_cast<const int32_t*>(blob + o_s);
#pragma unroll
    for(int q = 0; q < 4; q++) n_w[q] = svs[min(8*(wv + NW*q) + (lane >> 3), b_smax - 1)];
    n_xr = svs[b_smax + min(tid, b_smax - 1)];
    const int wn = leaves[li].w, c0n = leaves[li].col0;
    n_nsl = leaves[li].nslots; n_lb = leaves[li].lds_bytes;
    if(tid >= NT - 64 && tid - (NT - 64) < wn) n_pvar = perm[c0n + tid - (NT - 64)];
  };
  auto pf_issue = [&](int li, int tid, int lane, int wv) {
    const int nsl = n_nsl, lb = n_lb;                 // (fetched with the table: a scalar load here would be waited for)
    const lf_v4i rj = lf_rsrc(Jv, jv_bytes), rb = lf_rsrc(blob_g, blob_bytes);
    double* Rn = P + pf_pud;
    // (x first: its address waits for the table, and the compiler counts only the loads it knows -- behind the copies
    // that wait would be for them)
    n_x = (x && tid < nsl) ? x[n_xr] : 0.0;
#pragma unroll
    for(int q = 0; q < 4; q++)
    {
      const int s0 = 8*(wv + NW*q);
      if(s0 < nsl)
      {
        const int sl = s0 + (lane >> 3);
        const unsigned voff = (sl < nsl) ? ((unsigned)(n_w[q] & 0xfffffff) << 3) + 16u*(lane & 7) : 0xfffffff0u;
        lf_dma16(rj, lf_sgpr((int)lf_lds_addr(Rn + s0*LF_RS)), voff);
      }
    }
    if(1024*wv < lb)
      lf_dma16(rb, lf_sgpr((int)lf_lds_addr(reinterpret_cast<uint8_t*>(LDS0) + 1024*wv)), (unsigned)li*(unsigned)b_stride + (unsigned)b_tb + 16u*tid);
  };
  // ---- PF = false: the staged rows of a leaf, in the order they have in J: thread = (row of the pass, position); every
  // load unconditional (a branch around a load costs its whole latency again), all of a thread's loads in flight at once
  // (per row one word: lanes 0..14 the first value of the row | its entries << 28, lane 15 the row itself (for x))
  int g_w[LF_Q], g_sd[LF_Q];
  double g_v[LF_Q];
  auto load_table_at = [&](const uint8_t* blob, int smax, int nsl, int tid, int k16) {
    const int32_t* svg = reinterpret_cast<const int32_t*>(blob);
    const uint16_t* sdg = reinterpret_cast<const uint16_t*>(svg + 2*smax);
    const int32_t* wg = (k16 == 15) ? svg + smax : svg;
#pragma unroll
    for(int q = 0; q < LF_Q; q++)
    {
      const int g = min((tid >> 4) + (NT/16)*q, nsl - 1);
      g_w[q] = wg[g];
      g_sd[q] = sdg[g];
    }
  };
  auto load_values = [&](int k16) {
#pragma unroll
    for(int q = 0; q < LF_Q; q++)
    {
      const int len = (int)((unsigned)g_w[q] >> 28), off = g_w[q] & 0xfffffff;
      const bool isx = k16 == 15;
      const double* src = isx ? (x ? x + g_w[q] : Jv) : Jv + ((size_t)off + max(min(k16, len - 1), 0));
      g_v[q] = *src;
    }
  };
  int leaf = blockIdx.x;
  if(PF)
  {
    // the first leaf of this workgroup: nothing to hide its fetch behind
    pf_table(leaf, tid0, tid0 & 63, tid0 >> 6);
    pf_issue(leaf, tid0, tid0 & 63, tid0 >> 6);
  }
  do        // (PF = false: once)
  {
  // (PF: the thread's index through an empty asm, so that nothing derived from it counts as invariant of the loop -- hoisted
  // out, those few dozen values took the registers the kernel has at 1024 threads, and ONE spilled value reloaded
  // between two copies waits for the first copy to land)
  int tid = tid0;
  if(PF) asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wv = tid >> 6, k16 = tid & 15;
  LF_STAMP(0);
  uint4 bl0 = {0u, 0u, 0u, 0u};
  if(!PF && b_stride > 0)
  {
    // table and schedule from the leaf's index alone (beside the load of its record)
    const uint8_t* blob = blob_g + (size_t)leaf*b_stride;
    load_table_at(blob, b_smax, b_smax, tid, k16);
    bl0 = reinterpret_cast<const uint4*>(blob + b_tb)[tid];      // (the upload is padded by a pass of the workgroup)
  }
  const LfLeaf lf = leaves[leaf];
  const bool more = PF && leaf + (int)gridDim.x < nleaf;
  if(!PF)
  {
    if(b_stride > 0) { if(tid >= NT - 64 && tid - (NT - 64) < lf.w) n_pvar = perm[lf.col0 + tid - (NT - 64)]; }
    else
    {
      load_table_at(blob_g + lf.blob, lf.nslots, lf.nslots, tid, k16);
      if(tid >= NT - 64 && tid - (NT - 64) < lf.w) n_pvar = perm[lf.col0 + tid - (NT - 64)];
    }
    load_values(k16);
  }
  const int w = lf.w, nrows = lf.nrows, mb = nrows - w, ldp = (mb + 1) & ~1, ntri = mb*(mb + 1)/2;
  const int nslots = lf.nslots;
  double* Us = P + ldp*w;                         // the packed update matrix (direct contributions, then U - them)
  double* Dg = Us + ((ntri + 2) & ~1);            // member blocks
  double* R  = PF ? P + pf_pud : Dg + ((4*w + 1) & ~1);             // staged rows (+ a row of zeros: slot nslots)
  double* Sc = R + LF_RS*((PF ? b_smax : nslots) + 1);              // partial strips of split tasks / the riders' shares
  uint8_t* B = PF ? reinterpret_cast<uint8_t*>(LDS0) : reinterpret_cast<uint8_t*>(Sc + 128*lf.nscr);   // the schedule
  const uint8_t* bg = blob_g + lf.blob;
  double* G = Lx + lf.lx;
  if(tid == 0) sbad = 0x7fffffff;
  // ---- 1. zeros where the front is formed; the rows and the schedule of this leaf in LDS
  if(PF)
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's copies have landed ...
    lf_barrier();                                               // ... everybody's
    if(tid < nslots) R[tid*LF_RS + 15] = n_x;                   // x of the rows (whatever followed a row in J lay there)
    if(tid < LF_RS) R[nslots*LF_RS + tid] = 0.0;
    const int nz = (int)((Dg + ((4*w + 1) & ~1)) - P) >> 1;
    for(int c = tid; c < nz; c += NT) reinterpret_cast<dlg_v2d*>(P)[c] = (dlg_v2d){0.0, 0.0};
  }
  else
  {
    // (the schedule: fetched at the top, stored at the end of this phase)
    const uint4 bl = (b_stride > 0) ? bl0 : reinterpret_cast<const uint4*>(bg + lf.o_lds)[min(tid, (lf.lds_bytes >> 4) - 1)];
    const int nz = (int)(R - P) >> 1;
    for(int c = tid; c < nz; c += NT) reinterpret_cast<dlg_v2d*>(P)[c] = (dlg_v2d){0.0, 0.0};
    if(tid < LF_RS) R[nslots*LF_RS + tid] = 0.0;
    for(int c = tid + NT; c < (lf.lds_bytes >> 4); c += NT)         // (a schedule of more than 16 KB: the rest)
      reinterpret_cast<uint4*>(B)[c] = reinterpret_cast<const uint4*>(bg + lf.o_lds)[c];
#pragma unroll
    for(int q = 0; q < LF_Q; q++)
    {
      const int g = (tid >> 4) + (NT/16)*q;
      // (positions past the row's entries hold whatever followed the row in J: no strip keeps a product with them;
      // position 15 without x: zero)
      if(g < nslots) R[g_sd[q]*LF_RS + k16] = (k16 == 15 && !x) ? 0.0 : g_v[q];
    }
    // (a leaf with more rows than one pass of the workgroup holds: the rest now, at full latency)
    if(nslots > (NT/16)*LF_Q)
    {
      const int smx = b_stride > 0 ? b_smax : nslots;
      const int32_t* svg = reinterpret_cast<const int32_t*>(bg);
      const int32_t* srg = svg + smx;
      const uint16_t* sdg = reinterpret_cast<const uint16_t*>(srg + smx);
      for(int g = (NT/16)*LF_Q + (tid >> 4); g < nslots; g += NT/16)
      {
        const int wd = (k16 == 15) ? srg[g] : svg[g], sd = sdg[g];
        const int len = (int)((unsigned)wd >> 28), off = wd & 0xfffffff;
        const bool isx = k16 == 15;
        const double t = isx ? (x ? x[wd] : 0.0) : Jv[(size_t)off + max(min(k16, len - 1), 0)];
        R[sd*LF_RS + k16] = (isx || k16 < len) ? t : 0.0;
      }
    }
    if(tid < (lf.lds_bytes >> 4)) reinterpret_cast<uint4*>(B)[tid] = bl;
  }
  const int pvar = n_pvar;
  lf_barrier();
  LF_STAMP(1);
  // (PF: where the next leaf's rows are -- on its way during the strip tasks)
  if(more) pf_table(leaf + gridDim.x, tid, lane, wv);
  // ---- 2. strip tasks, one wave each
  const LfTask* tk = reinterpret_cast<const LfTask*>(B + lf.o_task);
  const uint32_t* rbh = reinterpret_cast<const uint32_t*>(B + lf.o_rbh);
  const uint8_t* fib = B + lf.o_fi;
  const uint16_t* l16 = reinterpret_cast<const uint16_t*>(B);
  // (no global store between the fetch of the next leaf's table and its use: the compiler's wait for those loads would
  // be a wait for the stores behind them -- the Jt*x shares go to LDS first, out with the panel)
  double* jtp_rec = (x && jtp) ? s_jt : nullptr;
#ifdef DLG_LF_PROFILE
  int wslot = 0;
#endif
  LF_WSTAMP(1);
  constexpr bool env_nomv = false;
  for(int t = wv; t < lf.ntask; t += NW)
  {
    LF_WSTAMP(2);
    const uint4 tr = *reinterpret_cast<const uint4*>(tk + t);
    const uint4 ta = *(reinterpret_cast<const uint4*>(tk + t) + 1);
    const int plist = lf_sgpr(tr.x & 0xFFFF), nprow = lf_sgpr(tr.x >> 16);
    const int tlist = lf_sgpr(tr.y & 0xFFFF), ntrb = lf_sgpr(tr.y >> 16);
    const int col0 = lf_sgpr(tr.z & 0xFFFF), kj = lf_sgpr((tr.z >> 16) & 0xFF), wj = lf_sgpr(tr.z >> 24);
    const int kind = lf_sgpr(tr.w & 0xFF), scr = lf_sgpr((tr.w >> 8) & 0xFF), flags = lf_sgpr((tr.w >> 16) & 0xFF), hh = lf_sgpr(tr.w >> 24);
    const int rkj = lf_sgpr(ta.z & 0xFF), rwj = lf_sgpr((ta.z >> 8) & 0xFF);
    const int a_slot0 = lf_sgpr(ta.x & 0xFFFF), a_nout = lf_sgpr(ta.x >> 16), a_stride = lf_sgpr(ta.y & 0xFFFF), a_nin = lf_sgpr(ta.y >> 16);
    const uint8_t* pd = tk[t].pd;
    // Member strips whose rows are consecutive slots: the persistent positions on the VECTOR pipe.  Only the member's
    // few columns are wanted (3 of the 16 a product on the matrix cores would form, and an fp64 MFMA holds its SIMD for
    // its whole duration): lane = (row parity, position k, column pair) -- even and odd rows in the two halves of the
    // wave, columns jj and jj + 2 in one lane -- one read of the row's entry k and two of the member's columns per row
    // pair, the halves added at the end (even rows + odd rows: a fixed order).
    const bool vec_member = kind == 0 && (flags & 1) && a_nout == 1 && wj <= 4 && !env_nomv;
    if(vec_member)
    {
      LF_WSTAMP(3);
      const int par = lane >> 5, k = (lane >> 1) & 15, jj = lane & 1;
      const double* pa = R + (a_slot0 + par)*LF_RS + k;
      const double* pb = R + (a_slot0 + par)*LF_RS + kj + jj;
      double acc0 = 0.0, acc1 = 0.0;
      const int npair = (nprow + 1 - par) >> 1;                 // rows par, par + 2, ...
      int i = 0;
      constexpr int UM = 4;
      for(; i + UM <= npair; i += UM)
      {
        double a[UM], b0[UM], b1[UM];
#pragma unroll
        for(int u = 0; u < UM; u++) { a[u] = pa[(2*(i + u))*LF_RS]; b0[u] = pb[(2*(i + u))*LF_RS]; b1[u] = pb[(2*(i + u))*LF_RS + 2]; }
#pragma unroll
        for(int u = 0; u < UM; u++) { acc0 = fma(a[u], b0[u], acc0); acc1 = fma(a[u], b1[u], acc1); }
      }
      for(; i < npair; i++)
      {
        const double a = pa[(2*i)*LF_RS];
        acc0 = fma(a, pb[(2*i)*LF_RS], acc0); acc1 = fma(a, pb[(2*i)*LF_RS + 2], acc1);
      }
      LF_WSTAMP(4);
      acc0 += __shfl_down(acc0, 32, 64); acc1 += __shfl_down(acc1, 32, 64);
      if(par == 0)
      {
        const int pdk = pd[k];
        lf_put(acc0, 0, pdk, k, jj, wj, kj, col0, P, ldp, Dg, Us, mb, jtp_rec);
        lf_put(acc1, 0, pdk, k, jj + 2, wj, kj, col0, P, ldp, Dg, Us, mb, jtp_rec);
      }
    }
    else
    // persistent positions on the matrix cores: four rows per product, A = the rows' windows (lane = (position, row of
    // the four)), B = their entries in the column block; lane (n, q) ends up with positions q, q + 4, q + 8, q + 12 of
    // column n.  (The vector form -- a lane per (position, column), two LDS reads per row and lane -- is bound by
    // the LDS return path at a third of the lanes doing anything: the matrix cores buy operand bandwidth.)
    {
      LF_WSTAMP(3);
      const int mm = lane & 15, kq = lane >> 4;
      const double* ra = R + mm;
      // (B: the column block's entries; behind them the rider's, if the task carries one)
      const double* rb = R + ((mm < wj) ? kj + mm : min(rkj + (mm - wj), 15));
      const bool bval = mm < wj + rwj;
      // where this lane's four results go, worked out under the products: an index from P, the spare slot behind the
      // update matrix for the ones that go nowhere (no branch between the last product and the stores)
      int didx[4];
      {
        const int dump = (int)(Us - P) + ntri;
#pragma unroll
        for(int r4 = 0; r4 < 4; r4++)
        {
          const int k = kq + 4*r4, pdk = pd[k];
          const bool live = pdk != 0xFF && pdk != 0xFE;
          int idx = dump;
          if(mm < wj)
          {
            if(kind == 0)
            {
              const int c = k - kj;
              const int own = (int)(Dg - P) + (col0 + c)*4 + mm, bel = pdk + (col0 + mm)*ldp;
              idx = (pdk == 0xFD) ? ((c >= mm) ? own : dump) : (live ? bel : dump);
            }
            else if(kind == 1)
            {
              const int c = col0 + mm;
              idx = (live && pdk != 0xFC && pdk >= c) ? (int)(Us - P) + tri_col(c, mb) + pdk : dump;
            }
            else idx = (mm < 8) ? (int)(Sc - P) + scr*128 + k*8 + mm : dump;
          }
          else if(mm < wj + rwj) idx = (int)(Sc - P) + scr*128 + k*8 + (mm - wj);      // the rider's share (summed later)
          didx[r4] = idx;
        }
      }
      dlg_v4d acc = {0.0, 0.0, 0.0, 0.0};
      const int zrow = nslots*LF_RS;
      if(flags & 1)
      {
        // row i of the task = a_slot0 + (i / a_nin)*a_stride + i % a_nin; this lane's rows: kq, kq + 4, ...
        int o = 0, r = kq;
        while(r >= a_nin) { r -= a_nin; o++; }
        const int ngr = (nprow + 3) >> 2;
        auto slot_now = [&](int i) { return (i < nprow) ? (a_slot0 + o*a_stride + r)*LF_RS : zrow; };
        auto step = [&]() {
          r += 4;
          if((a_nin & 3) == 0) { if(r >= a_nin) { r -= a_nin; o++; } }
          else { while(r >= a_nin) { r -= a_nin; o++; } } };
        int g = 0;
        constexpr int UG = 4;                     // (groups of rows whose operands are in flight together; PF: registers)
        for(; g + UG <= ngr; g += UG)
        {
          double a[UG], bq[UG];
#pragma unroll
          for(int u = 0; u < UG; u++)
          {
            const int sl = slot_now(4*(g + u) + kq);
            a[u] = ra[sl]; bq[u] = rb[sl];
            step();
          }
#pragma unroll
          for(int u = 0; u < UG; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], bval ? bq[u] : 0.0, acc, 0, 0, 0);
        }
        for(; g < ngr; g++)
        {
          const int sl = slot_now(4*g + kq);
          const double a = ra[sl], bq = rb[sl];
          step();
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bval ? bq : 0.0, acc, 0, 0, 0);
        }
      }
      else
      {
        // (a list is padded to four with the zero row)
        for(int i = 0; i < nprow; i += 4)
        {
          const int sl = (int)l16[plist + i + kq]*LF_RS;
          const double a = ra[sl], bq = rb[sl];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bval ? bq : 0.0, acc, 0, 0, 0);
        }
      }
      LF_WSTAMP(4);
#pragma unroll
      for(int r4 = 0; r4 < 4; r4++) P[didx[r4]] = acc[r4];
      // (position 15 of a strip of the ancestors: the leaf's share of Jt*x)
      if(kind == 1 && kq == 3 && mm < wj && jtp_rec && pd[15] == 0xFC) jtp_rec[col0 + mm] = acc[3];
    }
    LF_WSTAMP(5);
    // transient positions (member strips): lane = (row-block of the round, transient position, column)
    const int nT = lf_sgpr(pd[31]);
    if(ntrb > 0 && nT > 0)
    {
      // (small integers: the quotients by way of float reciprocals, exact here, instead of two software divisions)
      const int nTw = nT*wj;
      const float inv_nTw = 1.0f/(float)nTw, inv_wj = 1.0f/(float)wj;
      const int e = (int)(((float)lane + 0.5f)*inv_nTw), rem = lane - e*nTw;
      const int tq = (int)(((float)rem + 0.5f)*inv_wj), j = rem - tq*wj;
      const int E = lf_sgpr((int)(64.5f*inv_nTw));
      const int kT = pd[16 + min(tq, 14)];
      const int dumpT = (int)(Us - P) + ntri;
      // (two rounds' row-block records on their way together: record -> rows is a dependent LDS round trip)
      for(int i0 = 0; i0 < ntrb; i0 += 2*E)
      {
        const bool on0 = e < E && i0 + e < ntrb, on1 = e < E && i0 + E + e < ntrb;
        const int q0 = on0 ? i0 + e : 0, q1 = on1 ? i0 + E + e : 0;
        const int rb0 = (flags & 2) ? tlist + q0 : (int)l16[tlist + q0], rb1 = (flags & 2) ? tlist + q1 : (int)l16[tlist + q1];
        const int f0 = fib[rb0*16 + kT], f1 = fib[rb1*16 + kT];
        int sl0, sl1, h0, h1;
        if(hh > 0) { sl0 = (a_slot0 + q0*hh)*LF_RS; sl1 = (a_slot0 + q1*hh)*LF_RS; h0 = hh; h1 = hh; }       // (no record to wait for)
        else
        {
          const uint32_t hd0 = rbh[rb0], hd1 = rbh[rb1];
          sl0 = (hd0 & 0xFFFF)*LF_RS; h0 = (hd0 >> 16) & 0xFF; sl1 = (hd1 & 0xFFFF)*LF_RS; h1 = (hd1 >> 16) & 0xFF;
        }
        const double* pa = R + kT; const double* pb = R + kj + j;
        double acc0 = 0.0, acc1 = 0.0;
        if(h0 == 2 && h1 == 2)
        {
          const double a00 = pa[sl0], b00 = pb[sl0], a01 = pa[sl0 + LF_RS], b01 = pb[sl0 + LF_RS];
          const double a10 = pa[sl1], b10 = pb[sl1], a11 = pa[sl1 + LF_RS], b11 = pb[sl1 + LF_RS];
          acc0 = fma(a00, b00, acc0); acc0 = fma(a01, b01, acc0);
          acc1 = fma(a10, b10, acc1); acc1 = fma(a11, b11, acc1);
        }
        else
        {
          for(int r = 0; r < h0; r++) acc0 = fma(pa[sl0 + r*LF_RS], pb[sl0 + r*LF_RS], acc0);
          for(int r = 0; r < h1; r++) acc1 = fma(pa[sl1 + r*LF_RS], pb[sl1 + r*LF_RS], acc1);
        }
        P[on0 ? f0 + (col0 + j)*ldp : dumpT] = acc0;
        P[on1 ? f1 + (col0 + j)*ldp : dumpT] = acc1;
      }
    }
  }
  LF_WSTAMP(6);
  lf_barrier();
  LF_WSTAMP(7);
  LF_STAMP(2);
  // split strips / a rider's shares: the partial strips in order (update matrix and Jt*x record only); beside them the
  // right-hand side row and lambda (panel, member blocks)
  if(lf.ncomb > 0)
  {
    const LfComb* cb = reinterpret_cast<const LfComb*>(B + lf.o_comb);
    for(int c = wv; c < lf.ncomb; c += NW)
    {
      const LfComb C = cb[c];
      const LfTask* T = tk + C.task;
      const int col0 = T->col0, kj = T->kj, wj = T->wj;
      const int k = lane >> 2, pdk = T->pd[k];
      for(int j0 = 0; j0 < wj; j0 += 4)
      {
        const int j = j0 + (lane & 3);
        double acc = 0.0;
        for(int q = 0; q < C.nscr; q++) acc += Sc[(C.scr0 + q)*128 + k*8 + min(j, 7)];
        lf_put(acc, 1, pdk, k, j, wj, kj, col0, P, ldp, Dg, Us, mb, jtp_rec);
      }
    }
  }
  if(tid >= NT - 64 && tid - (NT - 64) < w)
  {
    const int c = tid - (NT - 64);
    if(x) s_jo[c] = P[(mb - 1) + c*ldp];             // (Jt*x of the leaf's own columns: stored with the panel)
    else if(rhs) P[(mb - 1) + c*ldp] = rhs[pvar];
  }
  // the members beside them (a wave of their own: the diagonal blocks were complete with the strip tasks)
  for(int m = tid; m <= lf.nbd; m += NT) s_mcol[m] = (m < lf.nbd) ? m*lf.bdw : w;
  if(tid >= NT/2 && tid - NT/2 < 64)
    for(int m = tid - NT/2; m < lf.nbd; m += 64)
    {
      const int c0 = m*lf.bdw;
      int badcol;
      switch(lf.bdw)
      {
        case 1: badcol = lf_factor_member<1>(Dg, c0, s_rdiag, lambda); break;
        case 2: badcol = lf_factor_member<2>(Dg, c0, s_rdiag, lambda); break;
        case 3: badcol = lf_factor_member<3>(Dg, c0, s_rdiag, lambda); break;
        default: badcol = lf_factor_member<4>(Dg, c0, s_rdiag, lambda); break;
      }
      if(badcol >= 0) atomicMin(&sbad, lf.col0 + c0 + badcol);
    }
  lf_barrier();
  LF_STAMP(3);
  // (PF: rows, scratch and schedule are done with -- the next leaf's rows and schedule into their place, under the rest)
  if(more) pf_issue(leaf + gridDim.x, tid, lane, wv);
  if(tid == 0 && sbad != 0x7fffffff) atomicMax(lf_word, 0x7fffffff - sbad);       // (0: every pivot of every leaf was positive)
  // the rows below against every member's block: thread = (row, group of members) -- the members do not couple
  double* Pb = P - w;
  {
    const int rp = (mb + 63) & ~63, ng = max(1, NT/rp);
    const int gi = tid/rp, i = tid - gi*rp;
    if(i < mb && gi < ng)
      for(int m = gi; m < lf.nbd; m += ng)
      {
        const int c0 = m*lf.bdw;
        switch(lf.bdw)
        {
          case 1: bd_solve_row_c<1, 4>(Pb, ldp, w + i, c0, s_rdiag, Dg); break;
          case 2: bd_solve_row_c<2, 4>(Pb, ldp, w + i, c0, s_rdiag, Dg); break;
          case 3: bd_solve_row_c<3, 4>(Pb, ldp, w + i, c0, s_rdiag, Dg); break;
          default: bd_solve_row_c<4, 4>(Pb, ldp, w + i, c0, s_rdiag, Dg); break;
        }
      }
    lf_barrier();
  }
  LF_STAMP(4);
  // ---- 4. U = B B' on the matrix cores; what leaves is U - (direct contributions)
  {
    const int T = (mb + 15) >> 4;
    constexpr int SY_G = 1;                       // tiles per wave and round (register budget of 16 waves with the next leaf's rows in flight)
    const int ntiles = T*(T + 1)/2;
    const int rounds = (ntiles + NW*SY_G - 1)/(NW*SY_G), nchunks = rounds*NW;
    for(int c = wv; c < nchunks; c += NW)
    {
      const int t0 = (int)((unsigned)(c*ntiles)/(unsigned)nchunks), t1 = (int)((unsigned)((c + 1)*ntiles)/(unsigned)nchunks);
      switch(t1 - t0)
      {
        case 1: factor_tail_tiles<1>(Pb, ldp, w, mb, T, t0, Us, 3, false, false, lane, 0, false, mb, P, 0); break;
        default: break;
      }
    }
    lf_barrier();
  }
  LF_STAMP(5);
  // ---- 5. out: the update matrix, the rows below the member blocks
  {
    if(x && Jt_x && tid >= NT - 64 && tid - (NT - 64) < w) Jt_x[pvar] = s_jo[tid - (NT - 64)];
    // the factored member blocks (lower triangles) into the panel's top block
    for(int e = tid; e < w*lf.bdw; e += NT)
    {
      const int c = e/lf.bdw, q = e - c*lf.bdw, cm = c % lf.bdw;       // column c of the leaf, entry q of its row in the block
      if(q <= cm) G[c + (size_t)(c - cm + q)*nrows] = Dg[c*4 + q];
    }
    if(x && jtp) for(int e = tid; e < mb - 1; e += NT) jtp[lf.jtp + e] = s_jt[e];
    double* Ug = uscr + lf.u_off;
    for(int e = tid; e < ntri; e += NT) Ug[e] = Us[e];
    const int cp_rows = min(NT, (mb + 63) & ~63), cp_ng = NT/cp_rows, cp_g = tid/cp_rows;
    for(int i = tid - cp_g*cp_rows; i < mb && cp_g < cp_ng; i += cp_rows)
    {
      double* gp = G + (w + i);
      for(int j0 = cp_g; j0 < w; j0 += 8*cp_ng)
      {
        double v[8];
#pragma unroll
        for(int u = 0; u < 8; u++) v[u] = (j0 + u*cp_ng < w) ? P[i + (j0 + u*cp_ng)*ldp] : 0.0;
#pragma unroll
        for(int u = 0; u < 8; u++) if(j0 + u*cp_ng < w) gp[(size_t)(j0 + u*cp_ng)*nrows] = v[u];
      }
    }
  }
  LF_STAMP(6);
  lf_barrier();                                   // (every wave is done with the LDS: the next leaf may have it)
  leaf += gridDim.x;
  } while(PF && leaf < nleaf);
}
} // namespace

// set-up: uploads, the record buffer, the kernel's LDS attribute.  Leaves Y->lf_on = false (the separate
// kernels stay) where the schedule is not there or the level's launch parameters do not match.
int sparse_leaf_setup(dlg_backend* b)
{
  SparseSym* Y = b->sym;
  const SymHost& H = Y->H;
  Y->lf_on = false;
  if(!H.lf_ok || b->sharded() || H.part_nranks > 1 || H.lf_leaf.empty()) return DLG_OK;
  if(!(Y->fac_leaf.size() > 0 && Y->fac_leaf[0] && Y->syrk_fused[0])) return DLG_OK;
  if((int)H.lf_leaf.size() != H.fw_lvl_ptr[1] - H.fw_lvl_ptr[0]) return DLG_OK;
  if(!(H.uw_lvl_ptr[1] - H.uw_lvl_ptr[0] > 0 && H.upd_syrk[0] && Y->upd_nw[0] > 0)) return DLG_OK;     // (the update gather carries the leaves' pivot word)
  DLG_CHECK(upload(Y->lf_leaf, H.lf_leaf)); Y->allocs.push_back(Y->lf_leaf);
  {
    // (+ a pass of the workgroup behind the last blob: its threads read 16 bytes each from the schedule's start, whatever its size)
    std::vector<uint8_t> bl(H.lf_blob);
    bl.resize(bl.size() + 16*LF_NT, 0);
    DLG_CHECK(upload(Y->lf_blob, bl)); Y->allocs.push_back(Y->lf_blob);
  }
  DLG_HIP(hipMalloc(&Y->lf_jtp, sizeof(double)*(size_t)std::max<int64_t>(1, H.lf_jtp_size))); Y->allocs.push_back(Y->lf_jtp);
  DLG_CHECK(upload(Y->lf_jf_ptr, H.lf_jf_ptr)); Y->allocs.push_back(Y->lf_jf_ptr);
  DLG_CHECK(upload(Y->lf_jf_ent, H.lf_jf_ent)); Y->allocs.push_back(Y->lf_jf_ent);
  DLG_CHECK(upload(Y->lf_jf_var0, H.lf_jf_var0)); Y->allocs.push_back(Y->lf_jf_var0);
  DLG_CHECK(upload(Y->lf_jf_w, H.lf_jf_w)); Y->allocs.push_back(Y->lf_jf_w);
  DLG_CHECK(upload(Y->lf_jf_short, H.lf_jf_short)); Y->allocs.push_back(Y->lf_jf_short);
  {
    std::vector<int> rec(4*H.lf_jf_long.size());
    for(size_t k = 0; k < H.lf_jf_long.size(); k++)
    { const int v = H.lf_jf_long[k]; rec[4*k] = H.lf_jf_ptr[v]; rec[4*k+1] = H.lf_jf_ptr[v+1]; rec[4*k+2] = H.lf_jf_var0[v]; rec[4*k+3] = H.lf_jf_w[v]; }
    DLG_CHECK(upload(Y->lf_jf_long, rec)); Y->allocs.push_back(Y->lf_jf_long);
    const size_t nl = std::max<size_t>(1, H.lf_jf_long.size());
    DLG_HIP(hipMalloc(&Y->lf_jf_lpart, sizeof(double)*16*JFL_SEG*nl)); Y->allocs.push_back(Y->lf_jf_lpart);
    DLG_HIP(hipMalloc(&Y->lf_jf_lcnt, sizeof(int)*nl)); Y->allocs.push_back(Y->lf_jf_lcnt);
    DLG_HIP(hipMemset(Y->lf_jf_lcnt, 0, sizeof(int)*nl));
  }
  {
    std::vector<char> col(H.lf_col.begin(), H.lf_col.end());
    DLG_CHECK(upload(Y->lf_col, col)); Y->allocs.push_back(Y->lf_col);
  }
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_leaf_front<false>), hipFuncAttributeMaxDynamicSharedMemorySize, H.lf_lds));
  if(H.lf_pf_lds > 0)
    DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_leaf_front<true>), hipFuncAttributeMaxDynamicSharedMemorySize, H.lf_pf_lds));
  Y->lf_on = true;
  return DLG_OK;
}

// the launch.  x / Jt_x: Jt*x beside JtJ (the evaluation's pass); else rhs (or nothing) fills the right-hand-side
// row of the leaves' panels.  Lx: the (zeroed) panel buffer; its tail word takes the leaves' pivot flag.
int sparse_leaf_front(dlg_backend* b, const double* Jv, double* Lx, const double* x, double* Jt_x, const double* rhs, double lambda)
{
  SparseSym* Y = b->sym;
  const SymHost& H = Y->H;
  const int n = (int)H.lf_leaf.size();
  // (the persistent form -- a workgroup per CU, the next leaf copied into LDS under the current one -- where its one LDS
  // layout fits and the copy engine's 32-bit offsets reach; DOGLEG_AMD_LF_NO_PF=1: a workgroup per leaf)
  static const bool no_pf = getenv("DOGLEG_AMD_LF_NO_PF") != nullptr;
  static const int wg_env = getenv("DOGLEG_AMD_LF_WGS") ? atoi(getenv("DOGLEG_AMD_LF_WGS")) : 0;
  const size_t jvb = (size_t)Y->nnz_loc*8, blb = H.lf_blob.size() + 16*LF_NT;
  const int nwg_pf = std::max(1, wg_env > 0 ? wg_env : b->ncu);
  const bool pf = !no_pf && H.lf_pf_lds > 0 && H.lf_stride > 0 && jvb < ((size_t)1 << 32) - 65536 && blb < ((size_t)1 << 32) - 65536 && n > nwg_pf;      // (more leaves than workgroups: something to fetch ahead)
  int* word = reinterpret_cast<int*>(Lx + H.lx_size);
  if(pf)
  {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_leaf_front<true>), dim3(nwg_pf), dim3(LF_NT), H.lf_pf_lds, b->stream, Y->lf_leaf, n, Y->lf_blob, H.lf_stride, H.lf_smax, H.lf_tb,
                       H.lf_pf_b, H.lf_pf_pud, (unsigned)jvb, (unsigned)blb, Y->perm, Jv, x, rhs, lambda, Lx, Y->uscr, Jt_x, Y->lf_jtp, word);
  }
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_leaf_front<false>), dim3(n), dim3(LF_NT), H.lf_lds, b->stream, Y->lf_leaf, n, Y->lf_blob, H.lf_stride, H.lf_smax, H.lf_tb,
                       0, 0, 0u, 0u, Y->perm, Jv, x, rhs, lambda, Lx, Y->uscr, Jt_x, Y->lf_jtp, word);
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}

#ifdef DLG_LF_PROFILE
extern "C" void dlg_lf_profile_dump(int nleaf)
{
  std::vector<long long> h(4096*8);
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_lf_prof), sizeof(long long)*h.size());
  long long t0 = 0, tend = 0; double mean[6] = {0, 0, 0, 0, 0, 0}; int n = 0;
  for(int g = 0; g < 4096 && g < nleaf; g++)
  {
    const long long* q = &h[g*8];
    if(q[6] == 0) continue;
    n++;
    if(t0 == 0 || q[0] < t0) t0 = q[0];
    tend = std::max(tend, q[6]);
    for(int k = 0; k < 6; k++) mean[k] += (double)(q[k+1] - q[k]);
  }
  if(n == 0) return;
  {
    std::vector<long long> hw(16*64);
    hipMemcpyFromSymbol(hw.data(), HIP_SYMBOL(g_lf_wave), sizeof(long long)*hw.size());
    for(int wv = 0; wv < 16; wv++)
    {
      fprintf(stderr, "wave %2d:", wv);
      const long long b0 = hw[wv*64] & 0xffffffffffffffLL;
      for(int q = 0; q < 60 && hw[wv*64 + q]; q++) fprintf(stderr, " %lld:%lld", hw[wv*64 + q] >> 56, (hw[wv*64 + q] & 0xffffffffffffffLL) - b0);
      fprintf(stderr, "\n");
    }
  }
  fprintf(stderr, "leaf fronts: %d workgroups recorded, span %lld (10 ns) | mean: stage %.0f tasks %.0f combine+members %.0f rows %.0f tail %.0f store %.0f\n",
          n, tend - t0, mean[0]/n, mean[1]/n, mean[2]/n, mean[3]/n, mean[4]/n, mean[5]/n);
}
#endif
