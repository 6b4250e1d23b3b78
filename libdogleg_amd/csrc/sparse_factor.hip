// sparse_factor.hip -- K5: level-scheduled supernodal Cholesky on gfx950,
// replaces cholmod_factorize[_p] (dogleg.c:659-664).  Per elimination-tree level: panel
// factorisation (k_factor_level), then the updates of the ancestors by the level's panels.
#include "sparse_internal.h"
#include "panel_factor.h"
#include "factor_tail.h"

namespace {
// -DDLG_FL_PROFILE: phase clocks of workgroup 0 of every factor launch (tools only)
#ifdef DLG_FL_PROFILE
constexpr int FL_PROF_WG = 4096;       // workgroups per level whose phase clocks are kept
__device__ long long g_fl_prof[32*FL_PROF_WG*8];
__device__ long long g_fl_add[FL_PROF_WG*4];          // the children's adds of the one-launch region: see mf_add_children
#define FL_ADD_STAMP(k) do { if(HANDOFF && threadIdx.x == 0 && blockIdx.x < FL_PROF_WG) g_fl_add[blockIdx.x*4 + (k)] = wall_clock64(); } while(0)
#define FL_STAMP(k) do { if(threadIdx.x == 0 && blockIdx.x < FL_PROF_WG) g_fl_prof[((prof_lvl & 31)*FL_PROF_WG + blockIdx.x)*8 + (k)] = wall_clock64(); } while(0)      // 100 MHz, one clock for the chip
#else
#define FL_STAMP(k)
#define FL_ADD_STAMP(k)
#endif

// multifrontal region: add the update matrices of the children of a supernode into its LDS
// panel P (entries whose column is one of the supernode's own columns) and into its own update
// matrix Wt (the others; zeroed here).  The region stores NEGATED update matrices W = -U, so
// both destinations are plain additions.  The symbolic phase lists the destination of every
// entry of every child (mf_dst: element offset from P; bit 15: offset into a Wt kept in HBM),
// padded to whole batches with a scratch slot, so the kernel is a pure stream: a low-occupancy
// wave pays ~4 cycles per instruction of any kind, and a dependent global load costs
// microseconds up here -- all loads of a batch (MF_SLOTS entries per thread) are issued
// together.  One child at a time: inside a child no two entries share a destination, so the
// sums are in child order.
template <int NT, bool UT_LDS, bool HANDOFF = false>
__device__ __forceinline__ void mf_add_children(double* P, double* Wt, int ntri, int nch, int ch0,
                                                MfChild rc, const MfChild* __restrict__ mf_rec,
                                                const double* uscr, const uint16_t* __restrict__ mf_dst, int tid,
                                                int* pr_flag = nullptr, int pr_epoch = 0, DlgHandoff ho = DlgHandoff{nullptr, 0, 0}, int z0 = 0)
{
  constexpr int MF_SLOTS = (NT >= 512) ? 20 : 16;      // 512 threads: up to 10240 entries (a 139-row update matrix) in one round
  const int lane = tid & 63;
  // (z0 .. ntri: the entries this workgroup sums -- all of them, or a replica's own stretch of an update matrix kept in HBM)
  for(int e = z0 + tid; e < ntri; e += NT) Wt[e] = 0.0;
  __syncthreads();
  for(int k = 0; k < nch; k++)
  {
    if(k > 0 && (k & 63) == 0) rc = mf_rec[ch0 + k + min(lane, nch - k - 1)];
    const int npad = __builtin_amdgcn_readlane(rc.npad, k & 63);
    const int64_t uo = ((int64_t)__builtin_amdgcn_readlane((int)(rc.u_off >> 32), k & 63) << 32) |
                       (uint32_t)__builtin_amdgcn_readlane((int)rc.u_off, k & 63);
    const int64_t dof = ((int64_t)__builtin_amdgcn_readlane((int)(rc.dst_off >> 32), k & 63) << 32) |
                        (uint32_t)__builtin_amdgcn_readlane((int)rc.dst_off, k & 63);
    const double* Wc = uscr + uo + tid;
    const uint16_t* D = mf_dst + dof + tid;
    // the destinations of the first round are static data (cold in HBM, ~2 us away): on their way BEFORE the
    // wait for the child, so that behind its flag only the update matrix itself is fetched
    unsigned d[MF_SLOTS];
#pragma unroll
    for(int u = 0; u < MF_SLOTS; u++) d[u] = (u*NT < npad) ? D[u*NT] : 0u;
    if(HANDOFF)
    {
      // persistent top region: wait for THIS child's flags only (the children of this launch; the others
      // finished with the launch before) -- one flag per replica of the child, each of which hands over its
      // tile columns of the update matrix.  The symbolic phase lists the children by expected time of
      // arrival, the latest last: the early ones are added while the late one is still at work, and the
      // order of the sums stays what the list says.
      const int cr = __builtin_amdgcn_readlane(rc.rsv, k & 63);
      const int ci = cr & 0xfffff, cn = cr >> 20;          // (cr < 0: not in this launch, cn < 0)
      if(tid < cn)
      {
        int spins = 0;
        while(__hip_atomic_load(pr_flag + ci + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != pr_epoch + ho.skew)
        {
          __builtin_amdgcn_s_sleep(1);
          if(++spins > ho.spins) { atomicOr(ho.status, DLG_HANDOFF_FACTOR); break; }      // a child that never arrives: report (its own status word: not a pivot), do not hang
        }
      }
      __syncthreads();
      if(k == 0) FL_ADD_STAMP(0);
      if(k == nch - 1) FL_ADD_STAMP(2);
    }
    for(int e0 = 0; e0 < npad; e0 += NT*MF_SLOTS)      // npad is a multiple of 1024: whole rounds of NT
    {
      double v[MF_SLOTS], old[MF_SLOTS];
      if(e0 > 0)
      {
#pragma unroll
        for(int u = 0; u < MF_SLOTS; u++) if(e0 + u*NT < npad) d[u] = D[e0 + u*NT];
      }
#pragma unroll
      for(int u = 0; u < MF_SLOTS; u++)
        if(e0 + u*NT < npad)
        {
          // persistent top region: the child wrote these bytes in this launch (write-through stores,
          // see k_factor_level) -- read them around this CU's L1
          // (a global_ load: the hand-off is not measured for flat_ ones)
          typedef const __attribute__((address_space(1))) double* gptr_t;
          v[u] = HANDOFF ? __hip_atomic_load((gptr_t)(Wc + e0 + u*NT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : Wc[e0 + u*NT];
        }
      if(UT_LDS)
      {
#pragma unroll
        for(int u = 0; u < MF_SLOTS; u++) if(e0 + u*NT < npad) old[u] = P[d[u]];
#pragma unroll
        for(int u = 0; u < MF_SLOTS; u++) if(e0 + u*NT < npad) P[d[u]] = old[u] + v[u];
      }
      else
      {
        // the update matrix lives in HBM: each of its entries gets an atomic add (fire and forget,
        // no read round trip).  A child adds to an address at most once and the barrier orders the
        // children's instruction streams; the adds of one CU to one address reach L2 through the
        // same queue in that order (waiting for L2's acknowledgement before every barrier --
        // s_waitcnt vmcnt(0) -- was measured: it doubles this phase).  With sliced fronts no supernode of
        // config #4 comes here any more; the run-to-run bit tests that pin this path are
        // tests/test_scale_gpu.py::test_update_matrices_summed_in_hbm_are_bitwise_reproducible_config4
        // (slices off: 23 update matrices of the region summed here) and ..._config5.
#pragma unroll
        for(int u = 0; u < MF_SLOTS; u++) if(e0 + u*NT < npad) old[u] = P[(d[u] & 0x8000) ? 0 : d[u]];
#pragma unroll
        for(int u = 0; u < MF_SLOTS; u++)
          if(e0 + u*NT < npad)
          {
            if(d[u] & 0x8000) unsafeAtomicAdd(Wt + (d[u] & 0x7fff), v[u]);
            else P[d[u]] = old[u] + v[u];
          }
      }
    }
    __syncthreads();
    if(k == 0) FL_ADD_STAMP(1);
  }
}

// The usual case of the one-launch region -- two children, each update matrix one round of loads, everything
// staged in LDS -- with the second child's loads on their way while the first is being added.  Both children of
// a supernode on the critical chain arrive together; added one after the other they cost two memory round trips
// and a poll in between (2.5 + 1.3 + 2.4 us measured).  Here every thread, once the first child's loads are out,
// reads the second child's flags itself: if they are up (a thread's own finding: its loads come after its own
// look at the flags, nobody else's), its half of that child's entries is fetched while it adds the first child's.
// The order of the sums is the list's, as before: first child, workgroup barrier, second child.
template <int NT>
__device__ __forceinline__ void mf_add_two_pipelined(double* P, double* Wt, int nzero, MfChild rc, const double* uscr,
                                                     const uint16_t* __restrict__ mf_dst, int tid,
                                                     int* pr_flag, int pr_epoch, DlgHandoff ho)
{
  constexpr int H = 10;                    // entries per thread and half (two halves = one round of 20*NT)
  constexpr bool HANDOFF = true;           // (the profile build's stamps ask)
  (void)HANDOFF;
  typedef const __attribute__((address_space(1))) double* gptr_t;
  for(int e = tid; e < nzero; e += NT) Wt[e] = 0.0;
  const int npA = __builtin_amdgcn_readlane(rc.npad, 0), npB = __builtin_amdgcn_readlane(rc.npad, 1);
  auto rl64 = [&](int64_t v, int l) { return ((int64_t)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, l); };
  const double* WA = uscr + rl64(rc.u_off, 0) + tid; const double* WB = uscr + rl64(rc.u_off, 1) + tid;
  const uint16_t* DA = mf_dst + rl64(rc.dst_off, 0) + tid; const uint16_t* DB = mf_dst + rl64(rc.dst_off, 1) + tid;
  const int crA = __builtin_amdgcn_readlane(rc.rsv, 0), crB = __builtin_amdgcn_readlane(rc.rsv, 1);
  const int ciA = crA & 0xfffff, cnA = crA >> 20, ciB = crB & 0xfffff, cnB = crB >> 20;
  const int want = pr_epoch + ho.skew;
  auto wait_flags = [&](int ci, int cn) {
    if(tid < cn)
    {
      int spins = 0;
      while(__hip_atomic_load(pr_flag + ci + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want)
      {
        __builtin_amdgcn_s_sleep(1);
        if(++spins > ho.spins) { atomicOr(ho.status, DLG_HANDOFF_FACTOR); break; }
      }
    } };
  auto load_child = [&](const double* W, int np, double (&x)[H], double (&y)[H]) {
#pragma unroll
    for(int u = 0; u < H; u++) x[u] = (u*NT < np) ? __hip_atomic_load((gptr_t)(W + u*NT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
    for(int u = 0; u < H; u++) y[u] = ((H + u)*NT < np) ? __hip_atomic_load((gptr_t)(W + (H + u)*NT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
  };
  auto add_half = [&](const unsigned* d, int np, int u0, const double (&v)[H]) {
    double old[H];
#pragma unroll
    for(int u = 0; u < H; u++) old[u] = P[d[u0 + u]];
#pragma unroll
    for(int u = 0; u < H; u++) if((u0 + u)*NT < np) P[d[u0 + u]] = old[u] + v[u];
  };
  // destinations of both children: static data, on their way before any wait
  unsigned dA[2*H], dB[2*H];
#pragma unroll
  for(int u = 0; u < 2*H; u++) { dA[u] = (u*NT < npA) ? DA[u*NT] : 0u; dB[u] = (u*NT < npB) ? DB[u*NT] : 0u; }
  // Wave 0 waits for the first child and looks at the second child's flags in the same loads (lanes 8 ..): where BOTH are
  // up when the first one is -- or one more look later: the children of a balanced tree finish together, and since the
  // short last blocks of the 66-column separators went to the vector pipe that is the rule at the top of config #4 --
  // all entries of both update matrices are fetched in ONE round of loads; one behind the other they are two round trips
  // (8 - 9 us from the last flag to "children added" at the root, profiles/r05_top_of_tree_levels.txt).  The sums keep
  // their order either way: first child, barrier, second child.
  __shared__ int s_both;
  if(tid < 64)
  {
    int spins = 0;
    const bool la = tid < cnA, lb = tid >= 8 && tid < 8 + cnB;
    bool ball = false;
    for(;;)
    {
      const int va = la ? __hip_atomic_load(pr_flag + ciA + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
      const int vb = lb ? __hip_atomic_load(pr_flag + ciB + (tid - 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
      ball = __all(vb == want);
      if(__all(va == want)) break;
      __builtin_amdgcn_s_sleep(1);
      if(++spins > ho.spins) { if(tid == 0) atomicOr(ho.status, DLG_HANDOFF_FACTOR); ball = false; break; }
    }
#ifndef DLG_ADD_NO_GRACE
    if(!ball && cnB > 0 && spins <= ho.spins)
    {
      // (one more look: a second child a few hundred nanoseconds behind the first is worth that wait -- it saves a round trip)
      const int vb = lb ? __hip_atomic_load(pr_flag + ciB + (tid - 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
      ball = __all(vb == want);
    }
#endif
#ifdef DLG_ADD_NO_BOTH
    ball = false;
#endif
    if(tid == 0) s_both = (ball && cnB > 0) ? 1 : 0;
  }
  __syncthreads();                         // (also: Wt is zero everywhere)
  FL_ADD_STAMP(0);
  double vX[H], vY[H], bX[H], bY[H];
  if(s_both)
  {
    load_child(WA, npA, vX, vY);
    load_child(WB, npB, bX, bY);
    add_half(dA, npA, 0, vX); add_half(dA, npA, H, vY);
    __syncthreads();                       // the first child is in, in every thread's entries
    FL_ADD_STAMP(1);
    FL_ADD_STAMP(2);
#ifdef DLG_FL_PROFILE
    if(threadIdx.x == 0 && blockIdx.x < FL_PROF_WG) g_fl_add[blockIdx.x*4 + 3] = 2000 + 10*cnB;
#endif
    add_half(dB, npB, 0, bX); add_half(dB, npB, H, bY);
    __syncthreads();
    return;
  }
  // The second child is still at work: the first child's entries, and behind their first half every thread's own look at
  // the second child's flags -- if they are up (a thread's own finding: its loads come after its own look at the flags,
  // nobody else's), its half of that child's entries is fetched while it adds the first child's second half.
  load_child(WA, npA, vX, vY);
  add_half(dA, npA, 0, vX);
  int fl[8];
#pragma unroll
  for(int q = 0; q < 8; q++) fl[q] = (q < cnB) ? __hip_atomic_load(pr_flag + ciB + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
  add_half(dA, npA, H, vY);
  bool bup = true;
#pragma unroll
  for(int q = 0; q < 8; q++) bup = bup && fl[q] == want;
  if(bup) load_child(WB, npB, bX, bY);
  __syncthreads();                         // the first child is in, in every thread's entries
  FL_ADD_STAMP(1);
#ifdef DLG_FL_PROFILE
  if(threadIdx.x == 0 && blockIdx.x < FL_PROF_WG) g_fl_add[blockIdx.x*4 + 3] = 1000 + (bup ? 1 : 0) + 10*cnB + 100*(fl[0] == want);
#endif
  if(!bup) wait_flags(ciB, cnB);           // (a polling lane that saw them up itself has nothing to wait for)
  __syncthreads();
  FL_ADD_STAMP(2);
  if(!bup) load_child(WB, npB, bX, bY);
  add_half(dB, npB, 0, bX); add_half(dB, npB, H, bY);
  __syncthreads();
}

// factor one supernode panel per workgroup in LDS (column-major, even leading dimension).
// One workgroup per work item (FwItem) = (supernode, slice [r0,r1) of its below rows): the LDS
// panel holds the w x w top block plus the slice (up to ~160 KB); slices of one supernode factor
// the top block redundantly (identical arithmetic), slice 0 publishes it (top_scr, k_copy_top).
// The arithmetic is in panel_factor.h:
//   * regular panels, >= 256 threads: panel_factor_mfma -- per 8 columns wave 0 brings the
//     diagonal row tile up to date on the matrix cores, factors the 8x8 block in registers and
//     publishes it while the other waves update the remaining row tiles; barrier; every thread
//     solves its row; barrier;
//   * 128 threads: panel_factor (same steps, all waves factor the block redundantly);
//   * sibling-merged leaves (block-diagonal top): no sweep at all; unsliced ones in a compact
//     layout that never stages the (almost empty) top block (bd_compact_*), sliced ones through
//     panel_factor_blockdiag.
// mode 1 / 2: the (unsliced) panel's update matrix U = B B' comes straight out of LDS afterwards;
// mode 2 (multifrontal region) first adds the children's update matrices (mf_add_children), leaves
// them in U as well and stores W = -U.  U is staged behind the panel in LDS when both fit.
// The panel is copied in and out thread = (row, column group), CP_FLIGHT columns in flight.
// LEAF: every work item of the launch is an unsliced block-diagonal panel in the compact layout (a level
// of merged leaves) -- the paths of the other kinds of panel are compiled out, which brings the kernel
// under 128 registers: 4 workgroups per CU instead of 3.
template <int NT, bool LEAF = false>
__global__ void __launch_bounds__(NT, LEAF ? 4 : 1) k_factor_level(const FwItem* __restrict__ items,
                                                     const MfChild* __restrict__ mf_rec,
                                                     const int* __restrict__ sn_bd_col,
                                                     const uint16_t* __restrict__ mf_dst,
                                                     double* __restrict__ Lx,
                                                     double* __restrict__ top_scr,
                                                     int* __restrict__ info,
                                                     double* uscr, int mode,
                                                     int* pr_flag, int pr_epoch, DlgHandoff ho, int64_t pr_acc)
{
  extern __shared__ __attribute__((aligned(16))) double P[];
  __shared__ int sbad, s_skip;
  constexpr int DS = LEAF ? 4 : 8;             // doubles per column of the factored member blocks (compact layout)
  constexpr int CP_FLIGHT = LEAF ? 16 : 32;
#ifdef DLG_FL_PROFILE
  const int prof_lvl = mode >> 8;
  FL_STAMP(0);
#endif
  const bool stage_leaf_u = (mode & 4) != 0;    // childless supernodes may stage U in LDS too (host: no occupancy loss)
  const bool b16 = (mode & 16) != 0;            // panel_factor_b16 (blocks of 16, the diagonal tile in registers)
  mode &= 3;
  const FwItem it = items[blockIdx.x];
  const int r0 = it.r0, w = it.w, nrows = it.nrows;
  double* G = Lx + it.lx;
  const int tid = threadIdx.x, lane = tid & 63;
  const int nloc = w + (it.r1 - r0);          // rows held by this workgroup
  const int shift = r0;                        // local row i >= w  <->  panel row i + shift
  const int64_t top = it.top;
  const int mb = nloc - w, ntri = mb*(mb + 1)/2;
  // compact layout for unsliced block-diagonal panels (the merged point leaves): the top block is
  // almost all zeros and is never staged -- LDS holds the rows below it (Pb[i + j*ldp], i >= w) and
  // the factored member blocks (Dg), see panel_factor.h
  const bool cmp = it.nbd > 0 && top < 0;
  const int ldp = cmp ? ((mb + 1) & ~1) : ((nloc + 1) & ~1);
  double* Pb = cmp ? P - w : P;
  const int row0c = cmp ? w : 0;               // first row that is staged
  const bool has_u = mode != 0 && top < 0;
  // the update matrix is staged in LDS where it fits: behind the panel, its last columns in the unused
  // strict upper triangle of the top block if need be (it.jsp, sym_w_split)
  // (sliced: a replica of the one-launch region that keeps only its own columns of the update matrix -- the
  // packed entries [eA, eB) -- behind the panel; the host sized the LDS for it)
  const bool sliced = !LEAF && it.sliced != 0;
  const bool u_lds = has_u && (sliced || ((it.nch > 0 || stage_leaf_u) &&
                     (cmp ? (size_t)(ldp*w + ntri + 1 + DS*w)*sizeof(double) <= (size_t)FAC_LDS_BUDGET : it.jsp >= 0)));   // + the scratch slot of mf_dst
  const int usp = (u_lds && !cmp && !sliced) ? it.jsp : mb;          // first column kept up there (mb: none)
  const int sl_pad = sliced ? (it.eA & 1) : 0;            // (keeps packed index and LDS index of one parity: 16-byte copies)
  const int nlin = sliced ? it.eB - it.eA + sl_pad : usp*mb - usp*(usp - 1)/2;              // doubles behind the panel
  double* Ug = has_u ? uscr + it.u_off : nullptr;
  double* Us = P + ldp*w;                                 // what is behind the panel ...
  const int ush = sliced ? sl_pad - it.eA : 0;            // ... holds packed index e at Us[e + ush]
  double* Dg = Us + (u_lds ? nlin + 1 : 0);
  // (LEAF: members of at most 4 columns, supernodes of at most 64 -- sparse_factor_setup: less LDS per
  // workgroup, which is what lets a fourth one onto the CU)
  __shared__ int s_mcol[LEAF ? 68 : 260];
  __shared__ double s_rdiag[LEAF ? 64 : 256];
  MfChild rc = {0, 0, 0, 0};
  const bool mf_acc = !LEAF && mode == 2 && it.nch > 0;
  if(mf_acc) rc = mf_rec[it.ch0 + min(lane, it.nch - 1)];     // on its way during the panel copy
  // (s_skip: a launch before this one found a non-positive pivot -- the factorisation is going to be
  // thrown away by the lambda loop, dogleg.c:656-677: nothing to do here but to let the parent go on)
  if(tid == 0) { sbad = 0x7fffffff; s_skip = __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0x7fffffff; }
  // one-launch region: its last workgroup (dispatched last: all of them are on their CUs) opens the gate
  // behind which the second stream holds the Cauchy step's pass over J (the word behind the items' flags;
  // a matter of timing only -- that pass depends on nothing this launch computes)
  if(pr_flag && tid == 0 && blockIdx.x == gridDim.x - 1)
    __hip_atomic_store(pr_flag + gridDim.x, pr_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if(cmp)
  {
    __syncthreads();
    // (a workgroup of a factorisation that is not going to be used -- failed pivot in an earlier launch, abandoned
    // (sparse_abandon_enqueued), or found doomed by the look at the diagonal (sparse_factorize) -- leaves before it loads
    // anything and, above all, before the members are factored IN PLACE below: the panels stay the assembly's)
    if(s_skip)
    {
      if(pr_flag && tid == 0) __hip_atomic_store(pr_flag + blockIdx.x, pr_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    bd_compact_members<NT, DS>(G, nrows, w, tid, sn_bd_col + it.bd0, it.nbd, it.bdw, s_mcol, s_rdiag, Dg, &sbad, it.col0);
  }
  // LEAF: the rows below the top block are ONE linear image in LDS (column j at P + j ldp, ldp even): they come in by
  // loads that write LDS themselves (16 bytes a lane, a wave-instruction fills 128 consecutive doubles of the image, the
  // lane's source address is its own) -- all of a panel's loads are in flight at once and no register holds a value.
  // Through registers, sixteen columns a thread at a time, the copy was two dependent round trips and 9.8 of a leaf
  // workgroup's 26.8 us (tools/prof_factor.sh).  (A pad row reads the first entry of the next column: never used.)
  // (thread = (panel row, column group) of the copies through registers: rows padded to whole waves, the remaining threads
  // take further columns)
  const int cp_rows = min(NT, (nloc - row0c + 63) & ~63), cp_ng = NT/cp_rows, cp_g = tid/cp_rows;
  // The same for a whole unsliced panel (rows 0 .. nloc of every column, column j at P + j ldp): the first level of the
  // one-launch region waits for nothing but this copy (8.2 us of its 28, profiles/r05_top_of_tree_levels.txt).
#ifndef DLG_LEAF_NO_GLDS
  if((LEAF && cmp) || (!LEAF && !cmp && !sliced && shift == 0))
  {
    const int ntot = ldp*w;
    const double* g0 = cmp ? G + (w + shift) : G;
    for(int k = tid >> 6; 128*k < ntot; k += NT/64)
    {
      const int e = 128*k + 2*lane;
      if(e < ntot)
      {
        const int j = e / ldp, r = e - j*ldp;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g0 + (size_t)j*nrows + r),
                                         (__attribute__((address_space(3))) void*)(P + 128*k), 16, 0, 0);
      }
    }
  }
  else
#endif
  {
  for(int i = row0c + tid - cp_g*cp_rows; i < nloc && cp_g < cp_ng; i += cp_rows)
  {
    const double* gp = G + (i < w ? i : i + shift);
    for(int j0 = cp_g; j0 < w; j0 += CP_FLIGHT*cp_ng)
    {
      double v[CP_FLIGHT];
#pragma unroll
      for(int u = 0; u < CP_FLIGHT; u++) v[u] = (j0 + u*cp_ng < w) ? gp[(size_t)(j0 + u*cp_ng)*nrows] : 0.0;
#pragma unroll
      for(int u = 0; u < CP_FLIGHT; u++) if(j0 + u*cp_ng < w) Pb[i + (j0 + u*cp_ng)*ldp] = v[u];
    }
  }
  }
  // (persistent top region: the panel is in LDS already; the children's flags are awaited one by one in
  // mf_add_children)
  __syncthreads();
  FL_STAMP(1);
  // Replicas of a supernode all read its panel from HBM and ONE of them writes the factored panel back: the last one
  // (highest workgroup index: the others were dispatched before it), and only after every other replica has said that
  // its copy is in LDS -- a replica that finds no CU until late (a chip shared with other launches) would otherwise
  // read a panel that is factored already.  The word: -epoch = "panel read", epoch = "done" (what the parent waits for).
  const int nrep = (pr_flag && it.rsv2 > 1) ? it.rsv2 : 1;
  if(nrep > 1 && it.rep < nrep - 1 && tid == 0) __hip_atomic_store(pr_flag + blockIdx.x, -pr_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if(s_skip)
  {
    if(pr_flag && tid == 0) __hip_atomic_store(pr_flag + blockIdx.x, pr_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if(mf_acc && usp < mb)
    for(int jw = usp + (tid >> 6); jw < mb; jw += NT/64)
      for(int i = lane; i < mb - jw; i += 64) P[(mb - jw)*ldp + i] = 0.0;      // (the barrier is in mf_add_children)
  if(mf_acc)
  {
    // (two children, one round of loads each: the second child's loads overlap the first child's adds)
    const bool two = NT >= 512 && it.nch == 2 && __builtin_amdgcn_readlane(rc.npad, 0) <= 20*NT && __builtin_amdgcn_readlane(rc.npad, 1) <= 20*NT &&
                     (__builtin_amdgcn_readlane(rc.rsv, 0) >> 20) <= 8 && (__builtin_amdgcn_readlane(rc.rsv, 1) >> 20) <= 8;
    if(pr_flag && u_lds && two) mf_add_two_pipelined<NT>(P, Us, nlin, rc, uscr, mf_dst, tid, pr_flag, pr_epoch, ho);
    else if(pr_flag && u_lds) mf_add_children<NT, true, true>(P, Us, nlin, it.nch, it.ch0, rc, mf_rec, uscr, mf_dst, tid, pr_flag, pr_epoch, ho);
    // (replicas of a supernode whose update matrix is summed in HBM: each owns the packed entries [eA, eB) -- its tile
    // columns --, its destination lists drop everything else of the children that is not a panel entry)
    else if(pr_flag)     mf_add_children<NT, false, true>(P, Ug + pr_acc, (nrep > 1 && !sliced) ? it.eB : ntri, it.nch, it.ch0, rc, mf_rec, uscr, mf_dst, tid, pr_flag, pr_epoch, ho,
                                                          (nrep > 1 && !sliced) ? it.eA : 0);
    else if(u_lds)  mf_add_children<NT, true >(P, Us, nlin, it.nch, it.ch0, rc, mf_rec, uscr, mf_dst, tid);
    else            mf_add_children<NT, false>(P, Ug, ntri, it.nch, it.ch0, rc, mf_rec, uscr, mf_dst, tid);
  }
  FL_STAMP(2);
  if(cmp) bd_compact_rows<NT, DS>(Pb, ldp, nloc, w, tid, it.nbd, s_mcol, s_rdiag, Dg, it.bdw);
  else if(LEAF) { }
  else if(it.nbd > 0) panel_factor_blockdiag<NT>(P, ldp, nloc, w, tid, sn_bd_col + it.bd0, it.nbd, &sbad, it.col0, s_mcol, s_rdiag);
  else if(NT >= 256 && b16 && nloc <= 16*PF_B16_MAXT) panel_factor_b16<(NT >= 256 ? NT : 256)>(P, ldp, nloc, w, tid, &sbad, it.col0);
  else if(NT >= 256) panel_factor_mfma<(NT >= 256 ? NT : 256)>(P, ldp, nloc, w, tid, &sbad, it.col0);
  else         panel_factor<NT, true, true>(P, ldp, nloc, w, tid, &sbad, it.col0);
  FL_STAMP(3);
  if(r0 == 0 && it.rep == 0 && tid == 0 && sbad != 0x7fffffff) atomicMin(info, sbad);
  // U = B B' (mode 2: W = children's sum - B B'): lower 16x16 tiles on the matrix cores, both
  // operands read from the panel (factor_tail_tiles above).  The lower triangle of tiles is dealt
  // out in column-major tile order, in chunks of consecutive tiles: every round gives each wave one
  // chunk, all chunks of a round within one tile of each other -- the two waves of a SIMD share its
  // matrix core, so the tail is only as fast as the busiest SIMD.
  if(has_u)
  {
    const int T = (mb + 15) >> 4;
    const int wv = tid >> 6;
    double* Ud = u_lds ? Us : Ug;              // in place behind the panel (+ ush), or straight to the scratch
    constexpr int NWV = NT/64;
    constexpr int SY_G = (NT >= 512) ? 6 : 4;
    constexpr int TKD = LEAF ? 2 : 4;          // k-steps of operands in flight (factor_tail.h; the leaf instantiation has 128 registers)
    // (a replica of the one-launch region forms the tile columns [tj0, tj1) only: tiles [tlo, thi) in
    // column-major tile order; every tile has its own accumulator chain, so who forms it changes no bit)
    const int tjA = min(it.tj0, T), tjB = min(it.tj1, T);
    const int tlo = tjA*T - tjA*(tjA - 1)/2, thi = tjB*T - tjB*(tjB - 1)/2;
    const int ntiles = thi - tlo;
    const int rounds = (ntiles + NWV*SY_G - 1)/(NWV*SY_G), nchunks = rounds*NWV;
    const bool w_hbm = mf_acc && !u_lds;
    const bool st_wt = pr_flag != nullptr && !u_lds;
    const int ush_t = u_lds ? ush : 0;
    const int64_t acc_shift = st_wt ? pr_acc : 0;
    for(int c = wv; c < nchunks; c += NWV)
    {
      // (32-bit: c*ntiles < 2^31 by far; a 64-bit division is some hundred instructions in front of the products)
      const int t0 = tlo + (int)((unsigned)(c*ntiles)/(unsigned)nchunks), t1 = tlo + (int)((unsigned)((c + 1)*ntiles)/(unsigned)nchunks);
      switch(t1 - t0)
      {
        case 1: factor_tail_tiles<1, TKD>(Pb, ldp, w, mb, T, t0, Ud, mode, w_hbm, mf_acc, lane, acc_shift, st_wt, usp, P, ush_t); break;
        case 2: factor_tail_tiles<2, TKD>(Pb, ldp, w, mb, T, t0, Ud, mode, w_hbm, mf_acc, lane, acc_shift, st_wt, usp, P, ush_t); break;
        case 3: factor_tail_tiles<3, TKD>(Pb, ldp, w, mb, T, t0, Ud, mode, w_hbm, mf_acc, lane, acc_shift, st_wt, usp, P, ush_t); break;
        case 4: factor_tail_tiles<4, TKD>(Pb, ldp, w, mb, T, t0, Ud, mode, w_hbm, mf_acc, lane, acc_shift, st_wt, usp, P, ush_t); break;
        case 5: if(SY_G >= 5) factor_tail_tiles<(SY_G >= 5 ? 5 : 1), TKD>(Pb, ldp, w, mb, T, t0, Ud, mode, w_hbm, mf_acc, lane, acc_shift, st_wt, usp, P, ush_t); break;
        case 6: if(SY_G >= 6) factor_tail_tiles<(SY_G >= 6 ? 6 : 1), TKD>(Pb, ldp, w, mb, T, t0, Ud, mode, w_hbm, mf_acc, lane, acc_shift, st_wt, usp, P, ush_t); break;
        default: break;
      }
    }
    if(u_lds) __syncthreads();
  }
  FL_STAMP(4);
  if(u_lds)
  {
    typedef __attribute__((address_space(1))) double* gwptr_t;
    // the part behind the panel.  Persistent top region: the parent reads W in this same launch --
    // 16-byte write-through stores (the slots are 16-byte aligned and padded to an even length),
    // drained by every wave before the flag goes up
    // (a replica: the columns [jA, jB) of its tile columns -- a contiguous stretch [eA, eB) of the packed triangle)
    const int jA = min(16*it.tj0, mb), jB = min(it.tj1, 1 << 20) >= ((mb + 15) >> 4) ? mb : 16*it.tj1;
    const int jAl = min(jA, usp), jBl = min(jB, usp);
    const int eA = jAl*mb - jAl*(jAl - 1)/2, eB = jBl*mb - jBl*(jBl - 1)/2;
    if(pr_flag)
    {
      const int eA2 = (eA + 1) & ~1, eB2 = eB & ~1;
      for(int e = eA2 + 2*tid; e < eB2; e += 2*NT)
      {
        const dlg_v2d v2 = *reinterpret_cast<const dlg_v2d*>(Us + (e + ush));
        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(Ug + e), "v"(v2) : "memory");
      }
      if(tid == 0 && eA < eA2 && eA < eB) __hip_atomic_store((gwptr_t)(Ug + eA), Us[eA + ush], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if(tid == 1 && eB2 < eB && eB2 >= eA2) __hip_atomic_store((gwptr_t)(Ug + eB2), Us[eB2 + ush], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    else for(int e = eA + tid; e < eB; e += NT) Ug[e] = Us[e + ush];
    // the columns kept in the top block's upper triangle
    for(int jw = max(usp, jA) + (tid >> 6); jw < jB; jw += NT/64)
    {
      const double* src = P + (mb - jw)*ldp;
      double* dst = Ug + tri_col(jw, mb) + jw;
      for(int i = lane; i < mb - jw; i += 64)
      {
        if(pr_flag) __hip_atomic_store((gwptr_t)(dst + i), src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else dst[i] = src[i];
      }
    }
  }
  if(pr_flag)
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if(tid == 0) __hip_atomic_store(pr_flag + blockIdx.x, pr_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  FL_STAMP(7);
  const bool store_panel = it.rep == nrep - 1;       // the panel is the last replica's to store
  if(nrep > 1 && store_panel)
  {
    if(tid < nrep - 1)
    {
      int spins = 0;
      for(;;)
      {
        const int v = __hip_atomic_load(pr_flag + ((int)blockIdx.x - (nrep - 1) + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if(v == pr_epoch || v == -pr_epoch) break;
        __builtin_amdgcn_s_sleep(1);
        if(++spins > ho.spins) { atomicOr(ho.status, DLG_HANDOFF_FACTOR); break; }
      }
    }
    __syncthreads();
  }
  for(int i = row0c + tid - cp_g*cp_rows; i < nloc && cp_g < cp_ng && store_panel; i += cp_rows)
  {
    // rows below the top block go back to the panel; the top block too unless the supernode is
    // cut into slices (then slice 0 parks it in top_scr, see k_copy_top) or it was never staged
    double* gp; size_t gs;
    if(i >= w)        { gp = G + (i + shift); gs = (size_t)nrows; }
    else if(top < 0)  { gp = G + i; gs = (size_t)nrows; }
    else if(r0 == 0)  { gp = top_scr + top + i; gs = (size_t)w; }
    else continue;
    // (where the top block's upper triangle held part of W, only its lower triangle goes back)
    const int jend = (usp < mb && i < w) ? i + 1 : w;
    for(int j0 = cp_g; j0 < jend; j0 += 16*cp_ng)
    {
      double v[16];
#pragma unroll
      for(int u = 0; u < 16; u++) v[u] = (j0 + u*cp_ng < jend) ? Pb[i + (j0 + u*cp_ng)*ldp] : 0.0;
#pragma unroll
      for(int u = 0; u < 16; u++) if(j0 + u*cp_ng < jend) gp[(size_t)(j0 + u*cp_ng)*gs] = v[u];
    }
  }
  FL_STAMP(5);
#ifdef DLG_FL_PROFILE
  if(threadIdx.x == 0 && blockIdx.x < FL_PROF_WG)
    g_fl_prof[((prof_lvl & 31)*FL_PROF_WG + blockIdx.x)*8 + 6] = (long long)w | ((long long)nrows << 12) | ((long long)it.nch << 24) | ((long long)(u_lds ? 1 : 0) << 36) | ((long long)(has_u ? 1 : 0) << 37) | ((long long)(it.r1 - it.r0) << 40) | ((long long)(it.rep & 15) << 52) | ((long long)(it.pad & 63) << 56);
#endif
}
// publish the top blocks of the multi-slice supernodes
__global__ void __launch_bounds__(TPB) k_copy_top(const int* __restrict__ ms_sn, const int* __restrict__ sn_c0,
                                                  const int* __restrict__ sn_rowptr,
                                                  const int64_t* __restrict__ sn_lx,
                                                  const int64_t* __restrict__ sn_top,
                                                  double* __restrict__ Lx,
                                                  const double* __restrict__ top_scr, const int* __restrict__ info)
{
  // (a factorisation that broke down or was found doomed: the scratch holds an EARLIER factorisation's top blocks -- they
  // must not land in panels that sparse_assemble may take over as the assembly left them)
  if(*info != 0x7fffffff) return;
  const int s = ms_sn[blockIdx.x];
  const int w = sn_c0[s+1] - sn_c0[s];
  const int nrows = sn_rowptr[s+1] - sn_rowptr[s];
  double* G = Lx + sn_lx[s];
  const double* T = top_scr + sn_top[s];
  for(int e = threadIdx.x; e < w*w; e += TPB) { const int j = e / w, i = e - j*w; G[i + (size_t)j*nrows] = T[e]; }
}

// cooperative variant of the update for heavy sources (wide panels): the whole
// workgroup works on one sub-task at a time, thread per source row, the nc x wd
// block of the source rows that sit in the target columns staged in LDS.  The
// target panel is updated in HBM; a barrier orders consecutive sub-tasks.
__global__ void __launch_bounds__(TPB) k_update_coop(int unit0, const int* __restrict__ uw_item,
                                                     const int* __restrict__ uw_s0,
                                                     const int* __restrict__ uw_s1,
                                                     const int64_t* __restrict__ uw_part,
                                                     const int* __restrict__ ui_t,
                                                     const int* __restrict__ ui_col,
                                                     const int* __restrict__ ui_nc,
                                                     const SymSub* __restrict__ usub,
                                                     const int* __restrict__ relpos,
                                                     const int* __restrict__ sn_rowptr,
                                                     const int64_t* __restrict__ sn_lx,
                                                     double* __restrict__ Lx,
                                                     double* __restrict__ upart, const int* __restrict__ info)
{
  // (a factorisation that has broken down -- or was found doomed by the look at the diagonal, sparse_host.hip -- is not
  // worth finishing, and the panels stay what the assembly left: sparse_assemble takes them over for the next lambda)
  if(*info != 0x7fffffff) return;

  __shared__ __attribute__((aligned(16))) double Bs[256*8];
  const int unit = unit0 + blockIdx.x;
  const int item = uw_item[unit];
  const int t = ui_t[item], col = ui_col[item], nc = ui_nc[item];
  const int nrows_t = sn_rowptr[t+1] - sn_rowptr[t];
  const int s0 = uw_s0[unit], s1 = uw_s1[unit];
  const int64_t part = uw_part[unit];
  double* dst = (part < 0) ? (Lx + sn_lx[t] + (int64_t)col*nrows_t) : (upart + part);
  const double sgn = (part < 0) ? -1.0 : 1.0;
  const int tid = threadIdx.x;
  if(part >= 0)
  {
    for(int e = tid; e < nrows_t*nc; e += TPB) dst[e] = 0.0;
    __syncthreads();
  }
  for(int st = s0; st < s1; st++)
  {
    const SymSub U = usub[st];
    const double* Ld = Lx + U.src;
    const int* rel = relpos + U.rel;
    const int ld = U.nrows_d;
    batched_copy<TPB, 8>(U.wd*8, tid,
                         [&](int e) { const int c = e & 7, q = e >> 3; return (c < nc) ? Ld[c + (size_t)q*ld] : 0.0; },
                         [&](int e, double v) { Bs[e] = v; });
    __syncthreads();
    for(int i = tid; i < U.m; i += TPB)
    {
      double sacc[8];
#pragma unroll
      for(int c = 0; c < 8; c++) sacc[c] = 0.0;
#pragma unroll 4
      for(int q = 0; q < U.wd; q++)
      {
        const double ai = Ld[i + (size_t)q*ld];
        const double2 b0 = *reinterpret_cast<const double2*>(&Bs[q*8]);
        const double2 b1 = *reinterpret_cast<const double2*>(&Bs[q*8 + 2]);
        const double2 b2 = *reinterpret_cast<const double2*>(&Bs[q*8 + 4]);
        const double2 b3 = *reinterpret_cast<const double2*>(&Bs[q*8 + 6]);
        sacc[0] += ai*b0.x; sacc[1] += ai*b0.y; sacc[2] += ai*b1.x; sacc[3] += ai*b1.y;
        sacc[4] += ai*b2.x; sacc[5] += ai*b2.y; sacc[6] += ai*b3.x; sacc[7] += ai*b3.y;
      }
      const int cmax = (i < nc - 1) ? i : nc - 1;
      const int r = rel[i];
#pragma unroll
      for(int c = 0; c < 8; c++) if(c <= cmax) dst[r + c*nrows_t] += sgn*sacc[c];
    }
    __syncthreads();
  }
}

// ---- two-phase update of a level with many small sources -------------------------------
// Phase 1: U_d = B_d B_d' for every source d of the level (B_d = the mb rows below its diagonal
// block, wd columns), packed lower triangle (tri_col), into the scratch.
// One workgroup per source: B_d is staged in LDS once (k-major, zero padded to whole tiles),
// the lower 16x16 tiles are produced by v_mfma_f64_16x16x4_f64 with both operands read from
// LDS; a wave owns whole tile columns so the B operand is read once per k-step.
template <int NT>
__global__ void __launch_bounds__(NT) k_update_syrk(const int* __restrict__ lvl_sn,
                                                    const int* __restrict__ sn_c0,
                                                    const int* __restrict__ sn_rowptr,
                                                    const int64_t* __restrict__ sn_lx,
                                                    const int64_t* __restrict__ u_off,
                                                    const double* __restrict__ Lx,
                                                    double* __restrict__ uscr, int KC)
{
  extern __shared__ __attribute__((aligned(16))) double Bs[];
  constexpr int NW = NT/64, TPW = 8;            // waves, tiles per wave (kept in registers)
  const int d = lvl_sn[blockIdx.x];
  const int wd = sn_c0[d+1] - sn_c0[d], nrows = sn_rowptr[d+1] - sn_rowptr[d], mb = nrows - wd;
  const double* Ld = Lx + sn_lx[d] + wd;
  double* U = uscr + u_off[d];
  const int T = (mb + 15) >> 4, MB16 = T*16;
  const int LDB = ((mb + 31)/32)*32 + 16, K4 = (wd + 3) & ~3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int jn = lane & 15, kq = lane >> 4;
  // this wave's tiles: flat index over the lower tiles (column by column), round robin
  const int ntiles = T*(T + 1)/2;
  int aoff[TPW], boff[TPW];
#pragma unroll
  for(int q = 0; q < TPW; q++)
  {
    int rem = w + q*NW, tj = 0;
    if(rem < ntiles) { while(rem >= T - tj) { rem -= T - tj; tj++; } }
    else { rem = 0; tj = 0; }
    aoff[q] = 16*(tj + rem); boff[q] = 16*tj;
  }
  const int nmine = (ntiles - w + NW - 1)/NW;           // tiles of this wave (<= TPW by construction)
  dlg_v4d c4[TPW];
#pragma unroll
  for(int q = 0; q < TPW; q++) c4[q] = (dlg_v4d){0.0, 0.0, 0.0, 0.0};
  for(int k0 = 0; k0 < K4; k0 += KC)
  {
    const int kc = min(KC, K4 - k0);
    if(k0 > 0) __syncthreads();
    batched_copy<NT, 8>(kc*MB16, tid,
                        [&](int e) { const int k = k0 + e / MB16, i = e % MB16; return (k < wd && i < mb) ? Ld[i + (size_t)k*nrows] : 0.0; },
                        [&](int e, double v) { const int k = e / MB16, i = e - k*MB16; Bs[k*LDB + i] = v; });
    __syncthreads();
    const double* base = Bs + kq*LDB + jn;
#pragma unroll 2
    for(int kk = 0; kk < kc; kk += 4)
    {
#pragma unroll
      for(int q = 0; q < TPW; q++)
        if(q < nmine)
          c4[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(base[kk*LDB + aoff[q]], base[kk*LDB + boff[q]], c4[q], 0, 0, 0);
    }
  }
#pragma unroll
  for(int q = 0; q < TPW; q++)
    if(q < nmine)
    {
      const int j = boff[q] + jn;
#pragma unroll
      for(int r = 0; r < 4; r++)
      {
        const int i = aoff[q] + kq + 4*r;
        if(i < mb && j <= i) U[tri_col(j, mb) + i] = c4[q][r];
      }
    }
}
// Phase 2: a work unit (chunk of the sub-tasks of one target column block) adds the column
// blocks of the U_d it is fed from into wave-private LDS slabs (waves take sub-tasks round
// robin), sums the slabs in wave order and applies / stores the result like the other update
// kernels.
constexpr int GATHER_FLIGHT = 4;        // sub-tasks whose loads are in flight together
template <int NT>
__global__ void __launch_bounds__(NT) k_update_gather(int unit0, const GatherUnit* __restrict__ units,
                                                       const SymSub* __restrict__ usub,
                                                       const int64_t* __restrict__ usub_u,
                                                       const int* __restrict__ relpos,
                                                       double* __restrict__ Lx,
                                                       double* __restrict__ upart,
                                                       const double* __restrict__ uscr, int nw,
                                                       int* __restrict__ info,
                                                       const int* __restrict__ fin_gate = nullptr, int fin_epoch = 0, int* fin_status = nullptr)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ int s_skip;
  // (fin on the side, sparse_assemble.hip: the partial-sum stages of the target panels may still run on the second
  // stream -- the word is fetched now and looked at in front of the first access to a target panel)
  int fin_seen = fin_epoch;
  if(fin_gate && threadIdx.x == 0) fin_seen = __hip_atomic_load(fin_gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if(threadIdx.x == 0)
  {
    s_skip = *info != 0x7fffffff;      // a failed factorisation is not worth finishing
  }
  // (one flat record: the chain unit -> item -> target supernode -> its rows cost three dependent loads
  // in front of the first barrier)
  const GatherUnit U0 = units[unit0 + blockIdx.x];
  const int nc = U0.nc, nrows_t = U0.nrows_t, s0 = U0.s0, s1 = U0.s1;
  double* Lt = Lx + U0.lt;
  const int64_t part = U0.part;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slab = nrows_t*nc;
  for(int e = tid; e < slab*nw; e += NT) lds[e] = 0.0;
  __syncthreads();
  if(s_skip) return;
  if(w < nw)
  {
    double* acc = lds + (size_t)w*slab;
    // the records of this wave's sub-tasks (s0 + w, s0 + w + nw, ...): lane l fetches record l of
    // the current batch of 64 with one load per field, the loop broadcasts them with readlane;
    // GATHER_FLIGHT sub-tasks are in flight at a time
    const int nmine = (s1 - s0 - w + nw - 1)/nw;
    for(int k0 = 0; k0 < nmine; k0 += 64)
    {
      const int stl = s0 + w + (k0 + min(lane, nmine - k0 - 1))*nw;
      const SymSub R = usub[stl];
      const int64_t ru = usub_u[stl];
      const int r_rel = R.rel, r_mb = R.nrows_d - R.wd, r_m = R.m;
      const int r_ulo = (int)(uint32_t)ru, r_uhi = (int)(ru >> 32);
      const int nb = min(64, nmine - k0);
      for(int k = 0; k < nb; k += GATHER_FLIGHT)
      {
        int relv[GATHER_FLIGHT], mbv[GATHER_FLIGHT], mv[GATHER_FLIGHT], mmax = 0;
        const double* Uv[GATHER_FLIGHT];
#pragma unroll
        for(int g = 0; g < GATHER_FLIGHT; g++)
        {
          const int kg = min(k + g, nb - 1);
          relv[g] = __builtin_amdgcn_readlane(r_rel, kg);
          mbv[g] = __builtin_amdgcn_readlane(r_mb, kg);
          mv[g] = (k + g < nb) ? __builtin_amdgcn_readlane(r_m, kg) : 0;
          Uv[g] = uscr + (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(r_uhi, kg) << 32) | (uint32_t)__builtin_amdgcn_readlane(r_ulo, kg));
          mmax = max(mmax, mv[g]);
        }
        int mv2[GATHER_FLIGHT];          // rows of the sub-block (>= 1), also its packed column stride
#pragma unroll
        for(int g = 0; g < GATHER_FLIGHT; g++) mv2[g] = max(mv[g], 1);
        for(int i0 = 0; i0 < mmax; i0 += 64)
        {
          const int i = i0 + lane;
          const int cmax = (i < nc - 1) ? i : nc - 1;
          int ig[GATHER_FLIGHT], rg[GATHER_FLIGHT];
          double v[GATHER_FLIGHT][8];
#pragma unroll
          for(int g = 0; g < GATHER_FLIGHT; g++) { ig[g] = min(i, max(mv[g], 1) - 1); rg[g] = relpos[relv[g] + ig[g]]; }
#pragma unroll
          for(int g = 0; g < GATHER_FLIGHT; g++)
#pragma unroll
            for(int c = 0; c < 8; c++)
              if(c < nc) { const int cc = min(min(c, cmax), ig[g]); v[g][c] = Uv[g][ig[g] + cc*(mv2[g] - 1) - cc*(cc - 1)/2]; }   // packed lower triangle, column cc of the sub-block
#pragma unroll
          for(int g = 0; g < GATHER_FLIGHT; g++)
#pragma unroll
            for(int c = 0; c < 8; c++)
              if(c < nc) { if(i < mv[g] && c <= cmax) acc[rg[g] + c*nrows_t] += v[g][c]; }
        }
      }
    }
  }
  if(fin_gate && tid == 0)
    for(int spins = 0; fin_seen != fin_epoch; spins++)
    {
      if(spins > (1 << 21)) { atomicOr(fin_status, DLG_HANDOFF_FACTOR); break; }       // (never in order: reported, not hung)
      __builtin_amdgcn_s_sleep(8);
      fin_seen = __hip_atomic_load(fin_gate, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    }
  __syncthreads();
  // the slab goes out eight entries per thread at a time: the reads of the panel are in flight together
  for(int e0 = tid; e0 < slab; e0 += 8*NT)
  {
    double old[8], tot[8];
#pragma unroll
    for(int u = 0; u < 8; u++) old[u] = (part < 0 && e0 + u*NT < slab) ? Lt[e0 + u*NT] : 0.0;
#pragma unroll
    for(int u = 0; u < 8; u++)
    {
      tot[u] = 0.0;
      if(e0 + u*NT < slab) for(int k = 0; k < nw; k++) tot[u] += lds[(size_t)k*slab + e0 + u*NT];
    }
#pragma unroll
    for(int u = 0; u < 8; u++)
      if(e0 + u*NT < slab) { if(part < 0) Lt[e0 + u*NT] = old[u] - tot[u]; else upart[part + e0 + u*NT] = tot[u]; }
  }
}

// fp64-MFMA variant of the update for heavy sources.  The unit's work is cut into pieces
// (sub-task, 16*UPD_TILES source rows); wave w of the nw active waves takes pieces w, w+nw, ... and
// accumulates into its private LDS slab (nrows_t x nc), so no barrier orders the pieces; the
// slabs are summed in wave order at the end.  One piece:
//   C[i][j] = sum_k Ld[i][k] * Ld[j][k],   i = the piece's source rows (MFMA row tiles of 16), j < nc <= 8
// with v_mfma_f64_16x16x4_f64: A[m][k] = Ld[row tile][4 columns], B[k][n] = Ld[n][4 columns]
// (n >= nc: zero); both operands are read straight from the source panel (column-major:
// 16 consecutive rows per column are one 128-byte segment).
constexpr int UPD_TILES = 6;          // MFMA row tiles (16 source rows each) per piece
__global__ void __launch_bounds__(TPB) k_update_mfma(int unit0, const int* __restrict__ uw_item,
                                                     const int* __restrict__ uw_s0,
                                                     const int* __restrict__ uw_s1,
                                                     const int64_t* __restrict__ uw_part,
                                                     const int* __restrict__ ui_t,
                                                     const int* __restrict__ ui_col,
                                                     const int* __restrict__ ui_nc,
                                                     const SymSub* __restrict__ usub,
                                                     const int* __restrict__ relpos,
                                                     const int* __restrict__ sn_rowptr,
                                                     const int64_t* __restrict__ sn_lx,
                                                     double* __restrict__ Lx,
                                                     double* __restrict__ upart, int nw, const int* __restrict__ info)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  // (a factorisation that has broken down -- or was found doomed by the look at the diagonal, sparse_host.hip -- is not
  // worth finishing, and the panels stay what the assembly left: sparse_assemble takes them over for the next lambda)
  if(*info != 0x7fffffff) return;

  const int unit = unit0 + blockIdx.x;
  const int item = uw_item[unit];
  const int t = ui_t[item], col = ui_col[item], nc = ui_nc[item];
  const int nrows_t = sn_rowptr[t+1] - sn_rowptr[t];
  double* Lt = Lx + sn_lx[t] + (int64_t)col*nrows_t;
  const int s0 = uw_s0[unit], s1 = uw_s1[unit];
  const int64_t part = uw_part[unit];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slab = nrows_t*nc;
  for(int e = tid; e < slab*nw; e += TPB) lds[e] = 0.0;
  __syncthreads();
  if(w < nw)
  {
    double* acc = lds + (size_t)w*slab;
    const int jn = lane & 15, kq = lane >> 4;
    const int jc = min(jn, nc - 1);
    int piece = 0;
    for(int st = s0; st < s1; st++)
    {
      const SymSub U = usub[st];
      const double* Ld = Lx + U.src;
      const int* rel = relpos + U.rel;
      const int ld = U.nrows_d, wd = U.wd, m = U.m;
      for(int mt0 = 0; mt0 < m; mt0 += 16*UPD_TILES, piece++)
      {
        if(piece % nw != w) continue;
        const int ntile = min(UPD_TILES, (m - mt0 + 15) >> 4);
        const int relv0 = rel[min(mt0 + lane, m - 1)], relv1 = rel[min(mt0 + 64 + lane, m - 1)];
        dlg_v4d c4[UPD_TILES];
#pragma unroll
        for(int q = 0; q < UPD_TILES; q++) c4[q] = (dlg_v4d){0.0, 0.0, 0.0, 0.0};
        int ia[UPD_TILES];
        bool va[UPD_TILES];
#pragma unroll
        for(int q = 0; q < UPD_TILES; q++) { const int i = mt0 + 16*q + jn; va[q] = i < m; ia[q] = min(i, m - 1); }
        // 16 source columns (4 MFMA k-steps) per round: all loads of a round are issued before
        // its first product; rows / columns past the end are clamped and zeroed
        for(int kk = 0; kk < wd; kk += 16)
        {
          double a[4][UPD_TILES], b[4];
#pragma unroll
          for(int h = 0; h < 4; h++)
          {
            const int k = kk + 4*h + kq;
            const size_t co = (size_t)min(k, wd - 1)*ld;
            b[h] = Ld[jc + co];
#pragma unroll
            for(int q = 0; q < UPD_TILES; q++) a[h][q] = Ld[ia[q] + co];
          }
#pragma unroll
          for(int h = 0; h < 4; h++)
          {
            const bool kok = kk + 4*h + kq < wd;
            const double bv = (kok && jn < nc) ? b[h] : 0.0;
#pragma unroll
            for(int q = 0; q < UPD_TILES; q++)
              if(q < ntile) c4[q] = __builtin_amdgcn_mfma_f64_16x16x4f64((kok && va[q]) ? a[h][q] : 0.0, bv, c4[q], 0, 0, 0);
          }
        }
        // D[i'][j]: this lane holds rows i' = kq + 4r of every tile, column j = jn
#pragma unroll
        for(int q = 0; q < UPD_TILES; q++)
          if(q < ntile)
          {
#pragma unroll
            for(int r = 0; r < 4; r++)
            {
              const int il = 16*q + kq + 4*r, i = mt0 + il;
              const int rr = il < 64 ? __builtin_amdgcn_ds_bpermute(4*il, relv0) : __builtin_amdgcn_ds_bpermute(4*(il - 64), relv1);
              const int cmax = (i < nc - 1) ? i : nc - 1;
              if(i < m && jn <= cmax) acc[rr + jn*nrows_t] += c4[q][r];
            }
          }
      }
    }
  }
  __syncthreads();
  for(int e = tid; e < slab; e += TPB)
  {
    double tot = 0.0;
    for(int k = 0; k < nw; k++) tot += lds[(size_t)k*slab + e];
    if(part < 0) Lt[e] -= tot; else upart[part + e] = tot;
  }
}

// apply the updates of all source supernodes of one level to their ancestors.
// One workgroup per work unit = a chunk of the sub-tasks of one item
// (target supernode t, one var-block of its columns).  nw waves each own a
// private LDS slab (nrows_t x nc) and walk the unit's sub-tasks round-robin;
// the slabs are summed in wave order.  A single-chunk item subtracts the sum
// from the target panel; a multi-chunk item (e.g. a dense last block that every
// supernode updates) stores it as a partial slab for k_update_fin.
// nw == 0: the slab does not fit LDS; the workgroup accumulates in HBM.
__global__ void __launch_bounds__(TPB) k_update_level(int unit0, const int* __restrict__ uw_item,
                                                      const int* __restrict__ uw_s0,
                                                      const int* __restrict__ uw_s1,
                                                      const int64_t* __restrict__ uw_part,
                                                      const int* __restrict__ ui_t,
                                                      const int* __restrict__ ui_col,
                                                      const int* __restrict__ ui_nc,
                                                      const SymSub* __restrict__ usub,
                                                      const int* __restrict__ relpos,
                                                      const int* __restrict__ sn_rowptr,
                                                      const int64_t* __restrict__ sn_lx,
                                                      double* __restrict__ Lx,
                                                      double* __restrict__ upart, int nw, const int* __restrict__ info)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  // (a factorisation that has broken down -- or was found doomed by the look at the diagonal, sparse_host.hip -- is not
  // worth finishing, and the panels stay what the assembly left: sparse_assemble takes them over for the next lambda)
  if(*info != 0x7fffffff) return;

  const int unit = unit0 + blockIdx.x;
  const int item = uw_item[unit];
  const int t = ui_t[item], col = ui_col[item], nc = ui_nc[item];
  const int nrows_t = sn_rowptr[t+1] - sn_rowptr[t];
  double* Lt = Lx + sn_lx[t] + (int64_t)col*nrows_t;
  const int s0 = uw_s0[unit], s1 = uw_s1[unit];
  const int64_t part = uw_part[unit];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int slab = nrows_t*nc;

  if(nw > 0)
  {
    for(int e = tid; e < slab*nw; e += TPB) lds[e] = 0.0;
    __syncthreads();
    if(w < nw && s0 + w < s1)
    {
      double* acc = lds + (size_t)w*slab;
      SymSub cur = usub[s0 + w];
      for(int st = s0 + w; st < s1; st += nw)
      {
        const SymSub U = cur;
        if(st + nw < s1) cur = usub[st + nw];          // prefetch the next record
        const double* Ld = Lx + U.src;
        const int* rel = relpos + U.rel;
        const int ld = U.nrows_d;
        for(int i = lane; i < U.m; i += 64)
        {
          const int cmax = (i < nc - 1) ? i : nc - 1;
          double sacc[8];
#pragma unroll
          for(int c = 0; c < 8; c++) sacc[c] = 0.0;
          for(int q = 0; q < U.wd; q++)
          {
            const double ai = Ld[i + (size_t)q*ld];
#pragma unroll
            for(int c = 0; c < 8; c++) if(c <= cmax) sacc[c] += ai*Ld[c + (size_t)q*ld];
          }
          const int r = rel[i];
#pragma unroll
          for(int c = 0; c < 8; c++) if(c <= cmax) acc[r + c*nrows_t] += sacc[c];
        }
      }
    }
    __syncthreads();
    for(int e = tid; e < slab; e += TPB)
    {
      double tot = 0.0;
      for(int k = 0; k < nw; k++) tot += lds[(size_t)k*slab + e];
      if(part < 0) Lt[e] -= tot; else upart[part + e] = tot;
    }
  }
  else
  {
    double* dst = (part < 0) ? Lt : upart + part;
    const double sgn = (part < 0) ? -1.0 : 1.0;
    if(part >= 0) { for(int e = tid; e < slab; e += TPB) dst[e] = 0.0; __syncthreads(); }
    for(int st = s0; st < s1; st++)
    {
      const SymSub U = usub[st];
      const double* Ld = Lx + U.src;
      const int* rel = relpos + U.rel;
      const int ld = U.nrows_d;
      for(int i = tid; i < U.m; i += TPB)
      {
        const int cmax = (i < nc - 1) ? i : nc - 1;
        const int r = rel[i];
        for(int c = 0; c <= cmax; c++)
        {
          double sacc = 0.0;
          for(int q = 0; q < U.wd; q++) sacc += Ld[i + (size_t)q*ld]*Ld[c + (size_t)q*ld];
          dst[r + c*nrows_t] += sgn*sacc;
        }
      }
      __syncthreads();
    }
  }
}
// sum the partial slabs of a multi-chunk item and apply them.  grid.y workgroups share the
// elements of a slab in chunks of E; within a workgroup the partials are dealt to G = 256/E
// thread groups, each keeping four loads in flight.  The summation order is fixed by (G, n)
// alone, so the result is deterministic.
__global__ void __launch_bounds__(TPB) k_update_fin(int f0, const int* __restrict__ uf_item,
                                                    const int* __restrict__ uf_n,
                                                    const int64_t* __restrict__ uf_off,
                                                    const int* __restrict__ ui_t,
                                                    const int* __restrict__ ui_col,
                                                    const int* __restrict__ ui_nc,
                                                    const int* __restrict__ sn_rowptr,
                                                    const int64_t* __restrict__ sn_lx,
                                                    double* __restrict__ Lx,
                                                    const double* __restrict__ upart, const int* __restrict__ info)
{
  __shared__ double sh[TPB];
  // (a factorisation that has broken down -- or was found doomed by the look at the diagonal, sparse_host.hip -- is not
  // worth finishing, and the panels stay what the assembly left: sparse_assemble takes them over for the next lambda)
  if(*info != 0x7fffffff) return;

  const int f = f0 + blockIdx.x;
  const int item = uf_item[f], n = uf_n[f];
  const int t = ui_t[item], col = ui_col[item], nc = ui_nc[item];
  const int nrows_t = sn_rowptr[t+1] - sn_rowptr[t];
  double* Lt = Lx + sn_lx[t] + (int64_t)col*nrows_t;
  const int slab = nrows_t*nc;
  const double* src = upart + uf_off[f];
  const int E = (slab >= 32) ? 32 : 8;
  const int G = TPB/E;
  const int el = threadIdx.x % E, g = threadIdx.x / E;
  for(int ebase = blockIdx.y*E; ebase < slab; ebase += gridDim.y*E)
  {
    const int e = ebase + el;
    double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
    if(e < slab)
    {
      const double* q = src + e;
      int k = g;
      for(; k + 3*G < n; k += 4*G)
      {
        t0 += q[(size_t)k*slab];         t1 += q[(size_t)(k + G)*slab];
        t2 += q[(size_t)(k + 2*G)*slab]; t3 += q[(size_t)(k + 3*G)*slab];
      }
      for(; k < n; k += G) t0 += q[(size_t)k*slab];
    }
    __syncthreads();
    sh[threadIdx.x] = (t0 + t1) + (t2 + t3);
    __syncthreads();
    if(g == 0 && e < slab)
    {
      double sacc = 0.0;
      for(int k = 0; k < G; k++) sacc += sh[k*E + el];
      Lt[e] -= sacc;
    }
  }
}

} // namespace

static int env_int_host(const char* n, int d) { const char* v = getenv(n); return v ? atoi(v) : d; }
// per-level launch parameters of the factor and update kernels
// plan_only: everything but the HIP calls (uploads, allocations, function attributes) -- the schedule of the
// one-launch region is left in Y->pr_*_h for dlg_sparse_region_probe (host-only checks, no GPU)
int sparse_factor_setup(dlg_backend* b, bool plan_only)
{
  SparseSym* Y = b->sym;
  const SymHost& H = Y->H;
  Y->fac_b16 = true;
  Y->fac_lds.assign(H.nlevels, 0); Y->fac_nt.assign(H.nlevels, 512); Y->upd_coop.assign(H.nlevels, 0);
  Y->upd_lds.assign(H.nlevels, 0); Y->upd_nw.assign(H.nlevels, 0);
  Y->syrk_lds.assign(H.nlevels, 0); Y->syrk_nt.assign(H.nlevels, 256); Y->syrk_kc.assign(H.nlevels, 4);
  Y->syrk_fused.assign(H.nlevels, 0); Y->fin_ny.assign(H.nlevels, 1); Y->fac_stage.assign(H.nlevels, 0);
  // levels whose work items are all unsliced block-diagonal panels (merged leaves) outside the
  // multifrontal region, with members of at most 4 columns and at most 64 columns in all: the lean
  // instantiation (k_factor_level<256, true>)
  Y->fac_leaf.assign(H.nlevels, 0);
  for(int l = 0; l < H.nlevels && !getenv("DOGLEG_AMD_NO_LEAF_KERNEL"); l++)
  {
    bool all = H.fw_lvl_ptr[l+1] > H.fw_lvl_ptr[l] && l < H.mf_level0;
    long maxr = 0;
    for(int i = H.fw_lvl_ptr[l]; i < H.fw_lvl_ptr[l+1] && all; i++)
    {
      const FwItem& it = H.fw_item[i];
      if(!(it.nbd > 0 && it.top < 0 && it.bdw > 0 && it.bdw <= 4 && it.w <= 64)) all = false;
      maxr = std::max(maxr, (long)it.w + (it.r1 - it.r0));
    }
    // (the instantiation exists for 256 threads: the block size the level gets below)
    const int nt = (maxr <= 128) ? 128 : (maxr <= 256 ? 256 : 512);
    Y->fac_leaf[l] = (all && nt == 256) ? 1 : 0;
  }
  for(int l = 0; l < H.nlevels; l++)
  {
    long maxp = 0, maxw = 0, maxr = 0;
    for(int i = H.lvl_ptr[l]; i < H.lvl_ptr[l+1]; i++)
    {
      const int s = H.lvl_sn[i];
      const long wv = H.sn_c0[s+1] - H.sn_c0[s];
      if(wv > maxw) maxw = wv;
    }
    for(int i = H.fw_lvl_ptr[l]; i < H.fw_lvl_ptr[l+1]; i++)
    {
      const int s = H.fw_sn[i];
      const long wv = H.sn_c0[s+1] - H.sn_c0[s];
      const long nloc = wv + (H.fw_r1[i] - H.fw_r0[i]);
      // even leading dimension in LDS; unsliced block-diagonal tops (merged leaves) use the compact
      // layout that never stages the top block (k_factor_level: cmp)
      const bool cmp = H.sn_bd_ptr[s+1] > H.sn_bd_ptr[s] && H.sn_top[s] < 0;
      const long mbl = nloc - wv;
      const long p = cmp ? ((mbl + 1) & ~1L)*wv + (Y->fac_leaf[l] ? 4 : 8)*wv + 1 : ((nloc + 1) & ~1L)*wv;
      if(p > maxp) maxp = p;
      if(nloc > maxr) maxr = nloc;
    }
    Y->fac_nt[l] = (maxr <= 128) ? 128 : (maxr <= 256 ? 256 : 512);
    if(l >= H.mf_level0) Y->fac_nt[l] = env_int_host("DOGLEG_AMD_MF_NT", 512);
    Y->upd_coop[l] = (maxw > 8) ? 1 : 0;           // heavy sources: matrix-core / cooperative update kernels
    if(maxp*8 > FAC_LDS_BUDGET) { dlg_set_error("internal error: a factor slice does not fit LDS (%ld doubles)", maxp); return DLG_ERR_ARG; }
    Y->fac_lds[l] = (int)(maxp*8);
    long maxslab = 0;
    for(int it = H.ui_lvl_ptr[l]; it < H.ui_lvl_ptr[l+1]; it++)
    {
      const int t = H.ui_t[it];
      const long sl = (long)(H.sn_rowptr[t+1] - H.sn_rowptr[t])*H.ui_nc[it];
      if(sl > maxslab) maxslab = sl;
    }
    long finslab = 0;
    for(int f = H.uf_lvl_ptr[l]; f < H.uf_lvl_ptr[l+1]; f++)
    {
      const int it = H.uf_item[f], t = H.ui_t[it];
      finslab = std::max(finslab, (long)(H.sn_rowptr[t+1] - H.sn_rowptr[t])*H.ui_nc[it]);
    }
    Y->fin_ny[l] = (int)std::min(64L, std::max(1L, (finslab + 31)/32));
    int nw = 0;
    if(maxslab > 0) { nw = (int)(LDS_BUDGET/(maxslab*8)); if(nw > 4) nw = 4; }
    Y->upd_nw[l] = nw;
    Y->upd_lds[l] = (int)(maxslab*8*nw);
    // heavy sources: 2 = matrix-core update kernel (needs the target slabs in LDS), 1 = cooperative
    // kernel accumulating in HBM (also selectable with DOGLEG_AMD_NO_UPDATE_MFMA for testing)
    if(Y->upd_coop[l] && nw > 0 && !getenv("DOGLEG_AMD_NO_UPDATE_MFMA")) Y->upd_coop[l] = 2;
    if(H.upd_syrk[l])
    {
      long ldbmax = 0, k4max = 0, tmax = 0;
      for(int i = H.lvl_ptr[l]; i < H.lvl_ptr[l+1]; i++)
      {
        const int d = H.lvl_sn[i];
        const long wd = H.sn_c0[d+1] - H.sn_c0[d], mb = H.sn_rowptr[d+1] - H.sn_rowptr[d] - wd;
        ldbmax = std::max(ldbmax, ((mb + 31)/32)*32 + 16); k4max = std::max(k4max, (wd + 3)/4*4);
        tmax = std::max(tmax, (mb + 15)/16);
      }
      long kc = (65536/(ldbmax*8)) & ~3L;            // source columns staged per round (<= 64 KB of LDS)
      if(kc > k4max) kc = k4max;
      Y->syrk_kc[l] = (int)kc;
      Y->syrk_lds[l] = (int)(kc*ldbmax*8);
      Y->syrk_nt[l] = (tmax*(tmax + 1)/2 <= 32) ? 256 : 1024;
      // no supernode of the level is cut into slices: the factor kernel forms the U_d itself
      bool unsliced = true;
      for(int i = H.lvl_ptr[l]; i < H.lvl_ptr[l+1]; i++) if(H.sn_top[H.lvl_sn[i]] >= 0) unsliced = false;
      Y->syrk_fused[l] = (unsliced && nw > 0 && !getenv("DOGLEG_AMD_NO_SYRK_FUSE")) ? 1 : 0;
    }
    if(l >= H.mf_level0 || Y->syrk_fused[l])
    {
      // room for the update matrix behind the panel where both fit (same rule as the kernel and,
      // for supernodes with children, as the symbolic phase).  Childless supernodes only stage it
      // if that does not cost the level a resident workgroup per CU.
      long with_children = Y->fac_lds[l], leaves = Y->fac_lds[l];
      for(int i = H.lvl_ptr[l]; i < H.lvl_ptr[l+1]; i++)
      {
        const int s = H.lvl_sn[i];
        const long wv = H.sn_c0[s+1] - H.sn_c0[s], nr = H.sn_rowptr[s+1] - H.sn_rowptr[s], mb = nr - wv;
        const bool cmp = H.sn_bd_ptr[s+1] > H.sn_bd_ptr[s] && H.sn_top[s] < 0;
        const long pan = cmp ? ((mb + 1) & ~1L)*wv + (Y->fac_leaf[l] ? 4 : 8)*wv : ((nr + 1) & ~1L)*wv;
        const int jsp = cmp ? (int)mb : sym_w_split(wv, nr);      // the kernel's rule (sym_w_split: part of W may sit in the top block's upper triangle)
        const long need = (pan + (jsp >= 0 ? sym_w_linear(mb, jsp) : mb*(mb + 1)/2) + 1)*8;
        const long need0 = (pan + 1)*8;          // at least the scratch slot behind the panel
        const long want = (need <= FAC_LDS_BUDGET && jsp >= 0) ? need : need0;
        const bool has_children = l >= H.mf_level0 && H.mf_cptr[s+1] > H.mf_cptr[s];
        if(has_children) with_children = std::max(with_children, want); else leaves = std::max(leaves, want);
      }
      const long base = std::max((long)Y->fac_lds[l], with_children);
      const long per_cu0 = 163840/(base + 3584), per_cu1 = 163840/(std::max(base, leaves) + 3584);
      Y->fac_stage[l] = (per_cu1 == per_cu0) ? 1 : 0;
      Y->fac_lds[l] = (int)(Y->fac_stage[l] ? std::max(base, leaves) : base);
    }
    if(getenv("DOGLEG_AMD_SYM_DEBUG"))
      fprintf(stderr, "factor level %d: %d supernodes, dynamic LDS %d bytes, update matrices staged %d, leaf instantiation %d, gather: %d waves, %d bytes\n",
              l, H.lvl_ptr[l+1] - H.lvl_ptr[l], Y->fac_lds[l], (int)Y->fac_stage[l], (int)Y->fac_leaf[l], Y->upd_nw[l], Y->upd_lds[l]);
  }
  if(!Y->uw_flat && !H.uw_item.empty() && !plan_only)
  {
    std::vector<GatherUnit> fl(H.uw_item.size());
    for(size_t u = 0; u < fl.size(); u++)
    {
      const int item = H.uw_item[u], t = H.ui_t[item];
      const int nr = H.sn_rowptr[t+1] - H.sn_rowptr[t];
      fl[u].lt = H.sn_lx[t] + (int64_t)H.ui_col[item]*nr; fl[u].part = H.uw_part[u];
      fl[u].s0 = H.uw_s0[u]; fl[u].s1 = H.uw_s1[u]; fl[u].nrows_t = nr; fl[u].nc = H.ui_nc[item];
    }
    DLG_CHECK(upload(Y->uw_flat, fl)); Y->allocs.push_back(Y->uw_flat);
  }
  // Persistent top region: the last levels of the multifrontal region hold a few supernodes each and
  // every one of them waits for the one before -- each kernel boundary costs the launch gap, a cold
  // panel load and the store of the panel before the next level may start.  They go out as ONE
  // launch, workgroups in level order (a workgroup only ever waits for lower-numbered ones, so
  // in-order dispatch cannot deadlock): a workgroup stages its panel at once, waits for its
  // children's flags, and raises its own flag as soon as its update matrix is out -- before its
  // panel goes back to HBM.  Conditions: unsliced supernodes, update matrices staged in LDS (their
  // hand-off is the write-through store of that LDS copy), one block size, no update units.
  // A subtree partition has TWO such regions (round 4): the levels of the rank's own subtrees up to the cut
  // (PrRegion lo: measured per level launch they cost a rank 35 - 40 us a level, profiles/r04_scaling_projection.md)
  // and, behind the sum over the ranks, the replicated levels above it (PrRegion top, as on one rank).
  Y->pr_level0 = H.nlevels; Y->pr2_level0 = H.nlevels; Y->pr2_level1 = -1;
  const bool dbg = getenv("DOGLEG_AMD_TIMING") != nullptr;
  bool acc_any = false;
  // levels [lo, hi] (as far down from hi as the conditions hold) -> region R; returns its first level (> hi: none)
  auto build_region = [&](int lo_min, int hi, int& r_level0, int& r_lds, int& r_stage, int& r_nwg,
                          std::vector<FwItem>& items, std::vector<MfChild>& rec, std::vector<uint16_t>& dst) -> int
  {
    r_level0 = H.nlevels; r_nwg = 0;
    if(getenv("DOGLEG_AMD_NO_PERSIST") || H.nlevels < 2 || hi < lo_min) return DLG_OK;
    // (a workgroup of the region fills a CU; the cap counts SUPERNODES: twice the CUs -- with replicas the launch holds more
    // workgroups than the chip has CUs anyway, they are dispatched in order and only wait for lower-numbered ones)
    const int ncu = b->ncu;
    // (supernodes; since the replicas of round 3 a region holds more workgroups than the chip has CUs anyway.  Round 5: five
    // times the CUs -- config #5's levels 1 - 3 (269 + 251 + 128 supernodes) were one launch each in front of a region of
    // 406: K5 2.33 -> 2.07 ms with all of them in it, 227 -> 234 steps/s.  The backward solve keeps twice the CUs: its
    // region runs from the root DOWN, the populous levels last, and those are faster as launches of their own.)
    const int cap = env_int_host("DOGLEG_AMD_PERSIST_MAX", 5*ncu);
    int total = 0, l0 = hi + 1, lds = 0, stage = 1;
    const int nt = Y->fac_nt[hi];
    for(int l = hi; l >= std::max(1, lo_min); l--)
    {
      const int n = H.fw_lvl_ptr[l+1] - H.fw_lvl_ptr[l];
      if(l < H.mf_level0) break;
      if(n == 0 || total + n > cap || Y->fac_nt[l] != nt || Y->fac_lds[l] <= 0) break;
      if(H.uw_lvl_ptr[l+1] > H.uw_lvl_ptr[l] || H.uf_lvl_ptr[l+1] > H.uf_lvl_ptr[l]) break;
      bool ok = true;
      for(int i = H.fw_lvl_ptr[l]; i < H.fw_lvl_ptr[l+1] && ok; i++)
      {
        const FwItem& it = H.fw_item[i];
        if(it.top >= 0 || it.r0 != 0 || it.nbd > 0) ok = false;
      }
      if(!ok) break;
      stage = stage && Y->fac_stage[l]; total += n; l0 = l; lds = std::max(lds, Y->fac_lds[l]);
    }
    if(hi + 1 - l0 < 2) return DLG_OK;
    r_level0 = l0; r_lds = lds; r_stage = stage;
    // Replicas: the upper levels hold fewer supernodes than the chip has CUs, and behind the panel sweep of a
    // supernode sits its update matrix W = (children) - B B', a third of the level's critical path on ONE CU.
    // A supernode of such a level is given to several workgroups: each stages the panel, adds the children
    // and runs the sweep -- the same instructions on the same data, so the same bits, and nothing to hand
    // over between them --, then forms and publishes only its share of W's tile columns; the parent waits
    // for all of them.  A level gets replicas while it and its neighbour level still fit the chip together
    // (a workgroup that finds no CU starts late and pays its panel load on the critical path).
    {
      const int rmax = std::max(1, std::min(8, env_int_host("DOGLEG_AMD_FRONT_REPLICAS", 8)));
      const int fill = ncu/2, fill0 = ncu/2;
      const bool slice_ok = !getenv("DOGLEG_AMD_NO_FRONT_SLICES");
      std::vector<int> first(H.fw_item.size(), -1), count(H.fw_item.size(), 0);
      // the region's own children records and destination lists: those of the symbolic phase (whole update
      // matrix behind the panel), and behind them the lists of the replicas that keep a slice of it
      rec = H.mf_rec; dst = H.mf_dst;
      auto lin = [](long j, long mb) { return j*mb - j*(j - 1)/2; };       // packed index of (j, j)
      long lds_need = lds;
      // Workgroups are dispatched in index order and a level holds more of them than CUs are free when its turn comes
      // (config #4: 230 + 134 + 128 + ... on 256 CUs): the ones that find a CU late should be the ones with slack.  A
      // level's supernodes are listed by the estimated time their subtree is done, the latest first (a static estimate:
      // columns swept in blocks of 16, rows, a constant for the children's sum and the hand-off) -- a parent whose
      // children are the level's slowest is on its CU, panel staged, when they arrive.  Levels stay in order (a
      // workgroup only waits for lower-numbered ones); no arithmetic depends on the order.
      std::vector<double> est(H.fw_item.size(), 0.0);
      std::vector<int> item_of_sn(H.nsn, -1);
      for(int l = r_level0; l <= hi; l++)
        for(int i = H.fw_lvl_ptr[l]; i < H.fw_lvl_ptr[l+1]; i++) item_of_sn[H.fw_sn[i]] = i;
      for(int l = r_level0; l <= hi; l++)
      {
        const int n = H.fw_lvl_ptr[l+1] - H.fw_lvl_ptr[l];
        std::vector<int> order(n);
        for(int k = 0; k < n; k++) order[k] = H.fw_lvl_ptr[l] + k;
        for(int i : order)
        {
          const FwItem& fi = H.fw_item[i];
          double e0 = 0.0;
          for(int k = 0; k < fi.nch; k++) { const int ci = item_of_sn[H.mf_child[fi.ch0 + k]]; if(ci >= 0) e0 = std::max(e0, est[ci]); }
          est[i] = e0 + 3.0*((fi.w + 15)/16) + 0.02*fi.nrows + 9.0;
        }
        std::stable_sort(order.begin(), order.end(), [&](int a, int b2) { return est[a] > est[b2]; });
        // (the first level of the region is its most populous one, and the Cauchy step's pass over J runs beside
        // it: replicas there cost more in CUs than they save -- only what does not fit LDS whole is sliced)
        const int rl = std::max(1, std::min(rmax, (l == r_level0 ? fill0 : fill)/std::max(n, 1)));
        // (the first level's supernodes whose panel is at least 70 % of the
        // level's largest get a second workgroup for their update matrix, while CUs are left)
        const int l1_pct = (l == r_level0 && rl == 1) ? 70 : 0;
        long l1_max = 0; int l1_left = std::max(0, ncu - n);
        if(l1_pct > 0) for(int i = H.fw_lvl_ptr[l]; i < H.fw_lvl_ptr[l+1]; i++) l1_max = std::max(l1_max, (long)H.fw_item[i].nrows*H.fw_item[i].w);
        for(int i : order)
        {
          FwItem it = H.fw_item[i];
          const int mb = it.nrows - it.w, T = (mb + 15) >> 4;
          int rl_i = rl;
          if(l1_pct > 0 && l1_left > 0 && (long)it.nrows*it.w*100 >= l1_max*l1_pct) { rl_i = 2; l1_left--; }
          const long ldp = (it.nrows + 1) & ~1L, pan = ldp*it.w, room = FAC_LDS_BUDGET/8 - pan - 2;
          const bool has_w = mb > 0 && it.u_off >= 0;
          // an update matrix that does not fit LDS whole (it.jsp < 0) fits in slices: two replicas at least
          int want = has_w ? std::min(rl_i, T) : 1;
          if(has_w && it.jsp < 0 && slice_ok) want = std::max(want, std::min(2, T));
          std::vector<int> cut;
          int nrep = 1;
          for(; want <= std::min(8, std::max(T, 1)); want++)
          {
            // contiguous tile columns, tile counts T - t, the largest share as small as possible
            for(int cap2 = (T*(T + 1)/2 + want - 1)/std::max(want, 1); ; cap2++)
            {
              cut.assign(1, 0);
              int load = 0;
              for(int t = 0; t < T; t++)
              {
                if(load > 0 && load + (T - t) > cap2) { cut.push_back(t); load = 0; }
                load += T - t;
              }
              cut.push_back(T);
              if((int)cut.size() - 1 <= want) break;
            }
            nrep = std::max(1, (int)cut.size() - 1);
            if(nrep == 1 || !slice_ok) break;
            bool fits = true;
            for(int r = 0; r < nrep; r++)
            {
              const long jA = std::min<long>(16L*cut[r], mb), jB = std::min<long>(16L*cut[r+1], mb);
              if(lin(jB, mb) - lin(jA, mb) + 1 > room) fits = false;
            }
            if(fits) break;
            nrep = 1;                                   // (more, narrower slices)
          }
          bool hbm_rep = false;
          if(nrep == 1 || !slice_ok)
          {
            // one workgroup (or replicas that each stage the whole update matrix: DOGLEG_AMD_NO_FRONT_SLICES)
            if(it.jsp < 0)
            {
              nrep = 1;
              if(has_w && it.nch > 0) acc_any = true;
              // Round 6: an update matrix that is summed in HBM (a panel that fills LDS by itself: config #5's fronts of
              // 250 - 300 rows x 60 - 66 columns) gets replicas too -- each sweeps the panel and OWNS a stretch of tile
              // columns of the update matrix in HBM: it sums the children's entries of that stretch only (its own destination
              // lists drop the others), forms B B' of those tiles and publishes them.  One workgroup formed all 120 tiles of
              // such a front, 30 - 46 us of every level on config #5's critical path (profiles/r05_top_of_tree_levels_config5.txt).
              const int want_h = (has_w && slice_ok) ? std::min(std::min(rl_i, T), 8) : 1;      // (at least 2 / 3 / 4 of them also on the populous levels: 336 / 330 / 326 steps/s against 338)
              if(want_h > 1)
              {
                for(int cap2 = (T*(T + 1)/2 + want_h - 1)/want_h; ; cap2++)
                {
                  cut.assign(1, 0);
                  int load = 0;
                  for(int t = 0; t < T; t++)
                  {
                    if(load > 0 && load + (T - t) > cap2) { cut.push_back(t); load = 0; }
                    load += T - t;
                  }
                  cut.push_back(T);
                  if((int)cut.size() - 1 <= want_h) break;
                }
                nrep = std::max(1, (int)cut.size() - 1);
                hbm_rep = nrep > 1;
              }
            }
            else if(nrep > 1) { /* whole-W replicas: the kernel's column-range copy-out */ }
          }
          first[i] = (int)items.size(); count[i] = nrep;
          for(int r = 0; r < nrep; r++)
          {
            it = H.fw_item[i];
            it.rep = r; it.tj0 = (nrep == 1) ? 0 : cut[r]; it.tj1 = (nrep == 1 || r == nrep - 1) ? (1 << 20) : cut[r+1];
            it.rsv2 = nrep;      // (the LAST replica stores the panel, once the others have read it: k_factor_level)
            it.pad = l;          // (the level: for the profile build's dump)
            if(hbm_rep)
            {
              // (not `sliced`: nothing of the update matrix is in LDS; [eA, eB) is the stretch of the packed triangle this
              // replica zeroes, sums and forms -- k_factor_level)
              const long jA = std::min<long>(16L*it.tj0, mb), jB = (r == nrep - 1) ? mb : std::min<long>(16L*it.tj1, mb);
              const long eA = lin(jA, mb), eB = lin(jB, mb);
              it.sliced = 0; it.eA = (int)eA; it.eB = (int)eB;
              lds_need = std::max(lds_need, (pan + 2)*8);
              const int ch0_new = (int)rec.size();
              for(int k = 0; k < it.nch; k++)
              {
                MfChild rc = H.mf_rec[it.ch0 + k];
                const int c = H.mf_child[it.ch0 + k];
                const int mc = (H.sn_rowptr[c+1] - H.sn_rowptr[c]) - (H.sn_c0[c+1] - H.sn_c0[c]);
                const int* map = &H.relpos[H.sn_prel[c]];
                const int nc = mc*(mc + 1)/2;
                rc.dst_off = (int64_t)dst.size();
                rc.rsv = (rc.rsv >= 0 && first[rc.rsv] >= 0) ? (first[rc.rsv] | (count[rc.rsv] << 20)) : -1;
                dst.reserve(dst.size() + rc.npad);
                for(int j = 0; j < mc; j++)
                  for(int q = j; q < mc; q++)
                  {
                    const long fi = map[q], fj = map[j], jw = fj - it.w;
                    const long d = (fj < it.w) ? fi + fj*ldp
                                 : (jw >= jA && jw < jB) ? (0x8000 | (lin(jw, mb) + (fi - fj))) : pan;      // pan: the scratch slot
                    dst.push_back((uint16_t)d);
                  }
                for(int e = nc; e < rc.npad; e++) dst.push_back((uint16_t)pan);
                rec.push_back(rc);
              }
              it.ch0 = ch0_new;
            }
            else if(nrep > 1 && slice_ok)
            {
              const long jA = std::min<long>(16L*it.tj0, mb), jB = (r == nrep - 1) ? mb : std::min<long>(16L*it.tj1, mb);
              const long eA = lin(jA, mb), eB = lin(jB, mb), slp = eA & 1;
              it.sliced = 1; it.eA = (int)eA; it.eB = (int)eB;
              lds_need = std::max(lds_need, (pan + slp + (eB - eA) + 2)*8);
              const long wt = pan + slp, trash = wt + (eB - eA);
              const int ch0_new = (int)rec.size();
              for(int k = 0; k < it.nch; k++)
              {
                MfChild rc = H.mf_rec[it.ch0 + k];
                const int c = H.mf_child[it.ch0 + k];
                const int mc = (H.sn_rowptr[c+1] - H.sn_rowptr[c]) - (H.sn_c0[c+1] - H.sn_c0[c]);
                const int* map = &H.relpos[H.sn_prel[c]];
                const int nc = mc*(mc + 1)/2;
                rc.dst_off = (int64_t)dst.size();
                rc.rsv = (rc.rsv >= 0 && first[rc.rsv] >= 0) ? (first[rc.rsv] | (count[rc.rsv] << 20)) : -1;
                dst.reserve(dst.size() + rc.npad);
                for(int j = 0; j < mc; j++)
                  for(int q = j; q < mc; q++)
                  {
                    const long fi = map[q], fj = map[j], jw = fj - it.w;
                    const long d = (fj < it.w) ? fi + fj*ldp
                                 : (jw >= jA && jw < jB) ? wt + (lin(jw, mb) - eA) + (fi - fj) : trash;
                    dst.push_back((uint16_t)d);
                  }
                for(int e = nc; e < rc.npad; e++) dst.push_back((uint16_t)trash);
                rec.push_back(rc);
              }
              it.ch0 = ch0_new;
            }
            items.push_back(it);
          }
        }
      }
      // (the records of the symbolic phase: the children's workgroups in this launch)
      for(size_t k = 0; k < H.mf_rec.size(); k++)
        rec[k].rsv = (rec[k].rsv >= 0 && first[rec[k].rsv] >= 0) ? (first[rec[k].rsv] | (count[rec[k].rsv] << 20)) : -1;
      r_nwg = (int)items.size();
      r_lds = (int)std::max<long>(r_lds, lds_need);
    }
    if(dbg)
      fprintf(stderr, "libdogleg_amd: one-launch region of the factorisation: levels %d..%d of %d (%d supernodes, %d workgroups, %d bytes of LDS, multifrontal from level %d)\n",
              r_level0, hi, H.nlevels, total, r_nwg, r_lds, H.mf_level0);
    return DLG_OK;
  };
  {
    const bool part = H.part_nranks > 1;
    std::vector<FwItem> items; std::vector<MfChild> rec; std::vector<uint16_t> dst;
    DLG_CHECK(build_region(part ? H.cut_level + 1 : 1, H.nlevels - 1, Y->pr_level0, Y->pr_lds, Y->pr_stage, Y->pr_nwg, items, rec, dst));
    if(Y->pr_level0 < H.nlevels)
    {
      if(plan_only) { Y->pr_item_h.swap(items); Y->pr_rec_h.swap(rec); Y->pr_dst_h.swap(dst); }
      else
      {
        if(!Y->pr_item) { DLG_CHECK(upload(Y->pr_item, items)); Y->allocs.push_back(Y->pr_item); }
        if(!Y->pr_rec)  { DLG_CHECK(upload(Y->pr_rec, rec));   Y->allocs.push_back(Y->pr_rec); }
        if(!Y->pr_dst)  { DLG_CHECK(upload(Y->pr_dst, dst));   Y->allocs.push_back(Y->pr_dst); }
      }
    }
    // the rank's own levels up to the cut (a partition whose cut leaves at least two multifrontal levels below it)
    if(part && H.cut_level >= 2 && !plan_only && !getenv("DOGLEG_AMD_NO_LOWER_REGION"))
    {
      std::vector<FwItem> items2; std::vector<MfChild> rec2; std::vector<uint16_t> dst2;
      DLG_CHECK(build_region(1, H.cut_level, Y->pr2_level0, Y->pr2_lds, Y->pr2_stage, Y->pr2_nwg, items2, rec2, dst2));
      if(Y->pr2_level0 <= H.cut_level)
      {
        Y->pr2_level1 = H.cut_level;
        if(!Y->pr2_item) { DLG_CHECK(upload(Y->pr2_item, items2)); Y->allocs.push_back(Y->pr2_item); }
        if(!Y->pr2_rec)  { DLG_CHECK(upload(Y->pr2_rec, rec2));   Y->allocs.push_back(Y->pr2_rec); }
        if(!Y->pr2_dst)  { DLG_CHECK(upload(Y->pr2_dst, dst2));   Y->allocs.push_back(Y->pr2_dst); }
      }
      else Y->pr2_level0 = H.nlevels;
    }
    // update matrices of a region that do not fit LDS are summed (atomics) in a shadow of the scratch,
    // so that the slot the parent reads only ever sees write-through stores
    if(acc_any && !Y->pr_acc && !plan_only)
    {
      DLG_HIP(hipMalloc(&Y->pr_acc, sizeof(double)*(size_t)std::max<int64_t>(1, H.uscr_size)));
      Y->allocs.push_back(Y->pr_acc);
    }
  }
  if(plan_only) return DLG_OK;
  if(Y->pr_level0 < H.nlevels && !Y->fac_flag)
  {
    DLG_HIP(hipMalloc(&Y->fac_flag, sizeof(int)*((size_t)Y->pr_nwg + 1)));         // one per workgroup of the region + the fork gate
    Y->allocs.push_back(Y->fac_flag);
    DLG_HIP(hipMemsetAsync(Y->fac_flag, 0, sizeof(int)*((size_t)Y->pr_nwg + 1), b->stream));
    Y->fac_epoch = 0;
  }
  if(Y->pr2_level0 < H.nlevels && !Y->fac2_flag)
  {
    DLG_HIP(hipMalloc(&Y->fac2_flag, sizeof(int)*((size_t)Y->pr2_nwg + 1)));
    Y->allocs.push_back(Y->fac2_flag);
    DLG_HIP(hipMemsetAsync(Y->fac2_flag, 0, sizeof(int)*((size_t)Y->pr2_nwg + 1), b->stream));
    Y->fac2_epoch = 0;
  }
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_factor_level<128>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, FAC_LDS_BUDGET));
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_factor_level<256>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, FAC_LDS_BUDGET));
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_factor_level<256, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, FAC_LDS_BUDGET));
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_factor_level<512>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, FAC_LDS_BUDGET));
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_update_level),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET));
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_update_syrk<256>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET));
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_update_syrk<1024>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET));
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_update_gather<TPB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET));
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_update_mfma),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET));
  return DLG_OK;
}

#ifdef DLG_FL_PROFILE
// per level: the workgroup that finishes last (its phases) and the mean over the recorded workgroups
extern "C" void dlg_fl_profile_dump(int nlevels)
{
  std::vector<long long> h(32*FL_PROF_WG*8);
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_fl_prof), sizeof(long long)*h.size());
  for(int l = 0; l < nlevels && l < 32; l++)
  {
    int nwg = 0, last = -1; long long t0 = 0, tend = 0; double mean[5] = {0, 0, 0, 0, 0};
    for(int g = 0; g < FL_PROF_WG; g++)
    {
      const long long* q = &h[(l*FL_PROF_WG + g)*8];
      if(q[5] == 0) continue;
      nwg++;
      if(t0 == 0 || q[0] < t0) t0 = q[0];
      if(q[5] > tend) { tend = q[5]; last = g; }
      for(int k = 0; k < 5; k++) mean[k] += (double)(q[k+1] - q[k]);
    }
    if(nwg == 0) continue;
    // a persistent launch: the timeline of its last workgroups (the top of the tree), 10 ns units
    for(int g = FL_PROF_WG - 1, shown = 0; g >= 0 && shown < (getenv("DLG_FL_DUMP_ALL") ? FL_PROF_WG : 20); g--)
    {
      const long long* q = &h[(l*FL_PROF_WG + g)*8];
      if(q[5] == 0 || q[7] == 0) continue;
      shown++;
      {
        std::vector<long long> ha(FL_PROF_WG*4);
        static bool got = false; static std::vector<long long> hadd;
        if(!got) { hadd.resize(FL_PROF_WG*4); hipMemcpyFromSymbol(hadd.data(), HIP_SYMBOL(g_fl_add), sizeof(long long)*hadd.size()); got = true; }
        if(hadd[g*4 + 0]) fprintf(stderr, "   add %3d: child0 seen %6lld child0 added %6lld last child seen %6lld dbg %lld\n", g, hadd[g*4] - t0, hadd[g*4 + 1] - t0, hadd[g*4 + 2] - t0, hadd[g*4 + 3]);
      }
      fprintf(stderr, "   wg %3d (w %3lld rows %4lld nch %lld u_lds %lld lvl %lld rep %lld): start %6lld flag up %6lld panel in %6lld added %6lld factored %6lld tail %6lld flag+stored %6lld\n",
              g, q[6] & 4095, (q[6] >> 12) & 4095, (q[6] >> 24) & 4095, (q[6] >> 36) & 1, (q[6] >> 56) & 63, (q[6] >> 52) & 15, q[0] - t0, q[7] - t0, q[1] - t0, q[2] - t0, q[3] - t0, q[4] - t0, q[5] - t0);
    }
    const long long* q = &h[(l*FL_PROF_WG + last)*8];
    fprintf(stderr, "level %2d: %3d wg, span %7lld | last wg %3d (w %3lld rows %4lld slice %4lld nch %2lld u_lds %lld has_u %lld): start +%6lld load %6lld add %6lld factor %6lld tail %6lld store %6lld | mean: load %6.0f add %6.0f factor %6.0f tail %6.0f store %6.0f\n",
            l, nwg, tend - t0, last, q[6] & 4095, (q[6] >> 12) & 4095, (q[6] >> 40) & 4095, (q[6] >> 24) & 4095, (q[6] >> 36) & 1, (q[6] >> 37) & 1,
            q[0] - t0, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4],
            mean[0]/nwg, mean[1]/nwg, mean[2]/nwg, mean[3]/nwg, mean[4]/nwg);
  }
  std::vector<long long> z(h.size(), 0);
  hipMemcpyToSymbol(HIP_SYMBOL(g_fl_prof), z.data(), sizeof(long long)*z.size());
}
#endif
// K5: level-scheduled supernodal Cholesky (launches only; the caller reads the pivot flag)
int sparse_factor_levels(dlg_backend* b, int part)
{
  SparseSym* Y = b->sym;
  const SymHost& H = Y->H;
  hipStream_t st = b->stream;
  // A factorisation enqueued ahead of the caller's decision (backend.hip, step_prepare) comes in two parts where its
  // first launch is a level of its own: part 1 = that launch (the leaf level), part 2 = everything behind it.  Same
  // launches in the same order on the same stream as part 0.
  const bool split = Y->pr_level0 > 0 && H.nlevels >= 2 && H.part_nranks <= 1 && H.fw_lvl_ptr[1] > H.fw_lvl_ptr[0];
  if(part == 2 && !split) return DLG_OK;
  if(part != 2)
  {
  if(H.part_nranks > 1 && H.cut_level < 0) DLG_CHECK(sparse_partition_reduce(b));     // nothing below the cut
  // (fin on the side, sparse_assemble.hip: the partial-sum stages of the ancestors' panels may still be on the second
  // stream -- level 0 does not touch those panels; whatever follows it does)
  if(Y->pr_level0 == 0 || H.nlevels < 2) DLG_CHECK(sparse_fin_side_gate(b));
  }
  for(int l = 0; l < H.nlevels; l++)
  {
    const bool launched_before = part == 2 && l == 0;      // (the level's factor launch is on the stream: part 1)

    const int n = H.fw_lvl_ptr[l+1] - H.fw_lvl_ptr[l];
    // from the first level that cannot fill the chip on, the factorisation is latency-bound:
    // independent work (the Cauchy step's pass over J) may run beside it
    const bool gate_here = (l == Y->pr_level0 && Y->fac_flag) || (l == Y->pr2_level0 && Y->fac2_flag);      // (no event on this stream: the launch opens a gate)
    if(l > 0 && n < 256 && !gate_here) dlg_fork_point(b);
    if(l == Y->pr2_level0 && Y->fac2_flag)
    {
      // a subtree partition: the rank's own levels up to the cut in one launch (sparse_factor_setup, region "lo")
      const int np = Y->pr2_nwg;
      const int fmode = 2 + 4*Y->pr2_stage + (Y->fac_b16 ? 16 : 0) + 256*(l & 31);
      int* fl = Y->fac2_flag; const int ep = ++Y->fac2_epoch;
      if(l > 0 && n < 256) dlg_fork_gate(b, fl + np, ep);
      const int64_t pacc = Y->pr_acc ? (int64_t)(Y->pr_acc - Y->uscr) : 0;
      const DlgHandoff ho = dlg_handoff(b, 1 << 21);
      {
        DlgRegionTurn turn(b);            // (held for the launch only: the sum over the ranks behind it blocks in the caller's hook)
        if(Y->fac_nt[l] == 128)
          hipLaunchKernelGGL(HIP_KERNEL_NAME(k_factor_level<128>), dim3(np), dim3(128), Y->pr2_lds, st,
                             Y->pr2_item, Y->pr2_rec, Y->sn_bd_col, Y->pr2_dst, Y->Lx, Y->top_scr, Y->d_info, Y->uscr, fmode, fl, ep, ho, pacc);
        else if(Y->fac_nt[l] == 256)
          hipLaunchKernelGGL(HIP_KERNEL_NAME(k_factor_level<256>), dim3(np), dim3(256), Y->pr2_lds, st,
                             Y->pr2_item, Y->pr2_rec, Y->sn_bd_col, Y->pr2_dst, Y->Lx, Y->top_scr, Y->d_info, Y->uscr, fmode, fl, ep, ho, pacc);
        else
          hipLaunchKernelGGL(HIP_KERNEL_NAME(k_factor_level<512>), dim3(np), dim3(512), Y->pr2_lds, st,
                             Y->pr2_item, Y->pr2_rec, Y->sn_bd_col, Y->pr2_dst, Y->Lx, Y->top_scr, Y->d_info, Y->uscr, fmode, fl, ep, ho, pacc);
      }
      l = Y->pr2_level1;                                 // (= the cut: everything below it is done)
      if(H.part_nranks > 1 && l == H.cut_level) { DLG_LAUNCH_CHECK(); DLG_CHECK(sparse_partition_reduce(b)); }
      continue;
    }
    if(l == Y->pr_level0)
    {
      // the persistent top region: every remaining level in one launch (sparse_factor_setup)
      const int np = Y->pr_nwg;           // the region's own work items (replicas) and children records
      const int fmode = 2 + 4*Y->pr_stage + (Y->fac_b16 ? 16 : 0) + 256*(l & 31);
      int* fl = Y->fac_flag; const int ep = ++Y->fac_epoch;
      // (the gate goes up with the region's LAST workgroup -- the top of the tree, on the chip once the populous levels below
      // it have drained --, however many supernodes the region's first level has: with `n < 256` asked for here too, config #5
      // -- 285 supernodes on level 1 -- never forked, and its Cauchy pass ran beside the backward solve instead)
      if(gate_here && l > 0) dlg_fork_gate(b, fl + np, ep);
      const int64_t pacc = Y->pr_acc ? (int64_t)(Y->pr_acc - Y->uscr) : 0;
      const DlgHandoff ho = dlg_handoff(b, 1 << 21);
      DlgRegionTurn turn(b);
      if(Y->fac_nt[l] == 128)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_factor_level<128>), dim3(np), dim3(128), Y->pr_lds, st,
                           Y->pr_item, Y->pr_rec, Y->sn_bd_col, Y->pr_dst, Y->Lx, Y->top_scr, Y->d_info, Y->uscr, fmode, fl, ep, ho, pacc);
      else if(Y->fac_nt[l] == 256)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_factor_level<256>), dim3(np), dim3(256), Y->pr_lds, st,
                           Y->pr_item, Y->pr_rec, Y->sn_bd_col, Y->pr_dst, Y->Lx, Y->top_scr, Y->d_info, Y->uscr, fmode, fl, ep, ho, pacc);
      else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_factor_level<512>), dim3(np), dim3(512), Y->pr_lds, st,
                           Y->pr_item, Y->pr_rec, Y->sn_bd_col, Y->pr_dst, Y->Lx, Y->top_scr, Y->d_info, Y->uscr, fmode, fl, ep, ho, pacc);
      break;
    }
    if(n > 0 && !launched_before)
    {
      const int o = H.fw_lvl_ptr[l];
      const int sweep_bits = (Y->fac_b16 ? 16 : 0);
      const int fmode = ((l >= H.mf_level0) ? 2 : Y->syrk_fused[l]) + 4*Y->fac_stage[l] + sweep_bits + 256*(l & 31);
      if(Y->fac_nt[l] == 128)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_factor_level<128>), dim3(n), dim3(128), Y->fac_lds[l], st,
                           Y->fw_item + o, Y->mf_rec, Y->sn_bd_col, Y->mf_dst, Y->Lx, Y->top_scr, Y->d_info, Y->uscr, fmode, (int*)nullptr, 0, DlgHandoff{nullptr, 0, 0}, (int64_t)0);
      else if(Y->fac_nt[l] == 256 && Y->fac_leaf[l])
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_factor_level<256, true>), dim3(n), dim3(256), Y->fac_lds[l], st,
                           Y->fw_item + o, Y->mf_rec, Y->sn_bd_col, Y->mf_dst, Y->Lx, Y->top_scr, Y->d_info, Y->uscr, fmode, (int*)nullptr, 0, DlgHandoff{nullptr, 0, 0}, (int64_t)0);
      else if(Y->fac_nt[l] == 256)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_factor_level<256>), dim3(n), dim3(256), Y->fac_lds[l], st,
                           Y->fw_item + o, Y->mf_rec, Y->sn_bd_col, Y->mf_dst, Y->Lx, Y->top_scr, Y->d_info, Y->uscr, fmode, (int*)nullptr, 0, DlgHandoff{nullptr, 0, 0}, (int64_t)0);
      else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_factor_level<512>), dim3(n), dim3(512), Y->fac_lds[l], st,
                           Y->fw_item + o, Y->mf_rec, Y->sn_bd_col, Y->mf_dst, Y->Lx, Y->top_scr, Y->d_info, Y->uscr, fmode, (int*)nullptr, 0, DlgHandoff{nullptr, 0, 0}, (int64_t)0);
    }
    if(part == 1 && split && l == 0) { Y->fac_pending = true; DLG_LAUNCH_CHECK(); return DLG_OK; }
    // (behind the leaf level's factor kernel, in front of its updates: the gather kernel looks at the word itself,
    // in front of anything else a one-wave kernel waits for it)
    const int nu = H.uw_lvl_ptr[l+1] - H.uw_lvl_ptr[l];
    const bool gather_next = nu > 0 && H.upd_syrk[l] && Y->upd_nw[l] > 0;
    const int fin_ep = (l == 0 && gather_next) ? Y->fin_side_owed : 0;
    if(fin_ep) Y->fin_side_owed = 0;
    if(l == 0) DLG_CHECK(sparse_fin_side_gate(b));
    if(gather_next)
    {
      const int ns = H.xl_ptr[l+1] - H.xl_ptr[l];
      if(Y->syrk_fused[l] || ns == 0) { /* phase 1 was done by the factor kernel */ }
      else if(Y->syrk_nt[l] == 256)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_update_syrk<256>), dim3(ns), dim3(256), Y->syrk_lds[l], st,
                           Y->xl_sn + H.xl_ptr[l], Y->sn_c0, Y->sn_rowptr, Y->sn_lx, Y->u_off, Y->Lx, Y->uscr,
                           Y->syrk_kc[l]);
      else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_update_syrk<1024>), dim3(ns), dim3(1024), Y->syrk_lds[l], st,
                           Y->xl_sn + H.xl_ptr[l], Y->sn_c0, Y->sn_rowptr, Y->sn_lx, Y->u_off, Y->Lx, Y->uscr,
                           Y->syrk_kc[l]);
      // (one wave per unit, all units resident at once, was measured: 78 us against 48 with up to four
      // waves sharing a unit's sub-tasks)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_update_gather<TPB>), dim3(nu), dim3(TPB), Y->upd_lds[l], st, H.uw_lvl_ptr[l],
                         Y->uw_flat, Y->usub, Y->usub_u, Y->relpos, Y->Lx, Y->upart, Y->uscr,
                         Y->upd_nw[l], Y->d_info,
                         fin_ep ? (const int*)(Y->fin_flag + 1) : (const int*)nullptr, fin_ep,
                         reinterpret_cast<int*>(b->d_scal + (dlg_backend::NSCAL - 2)));
    }
    else if(nu > 0 && Y->upd_coop[l] == 2)
      hipLaunchKernelGGL(k_update_mfma, dim3(nu), dim3(TPB), Y->upd_lds[l], st, H.uw_lvl_ptr[l],
                         Y->uw_item, Y->uw_s0, Y->uw_s1, Y->uw_part, Y->ui_t, Y->ui_col, Y->ui_nc,
                         Y->usub, Y->relpos, Y->sn_rowptr, Y->sn_lx, Y->Lx, Y->upart, Y->upd_nw[l], (const int*)Y->d_info);
    else if(nu > 0 && Y->upd_coop[l])
      hipLaunchKernelGGL(k_update_coop, dim3(nu), dim3(TPB), 0, st, H.uw_lvl_ptr[l],
                         Y->uw_item, Y->uw_s0, Y->uw_s1, Y->uw_part, Y->ui_t, Y->ui_col, Y->ui_nc,
                         Y->usub, Y->relpos, Y->sn_rowptr, Y->sn_lx, Y->Lx, Y->upart, (const int*)Y->d_info);
    else if(nu > 0)
      hipLaunchKernelGGL(k_update_level, dim3(nu), dim3(TPB), Y->upd_lds[l], st, H.uw_lvl_ptr[l],
                         Y->uw_item, Y->uw_s0, Y->uw_s1, Y->uw_part, Y->ui_t, Y->ui_col, Y->ui_nc,
                         Y->usub, Y->relpos, Y->sn_rowptr, Y->sn_lx, Y->Lx, Y->upart, Y->upd_nw[l], (const int*)Y->d_info);
    const int nfz = H.uf_lvl_ptr[l+1] - H.uf_lvl_ptr[l];
    if(nfz > 0)
      hipLaunchKernelGGL(k_update_fin, dim3(nfz, Y->fin_ny[l]), dim3(TPB), 0, st, H.uf_lvl_ptr[l], Y->uf_item, Y->uf_n,
                         Y->uf_off, Y->ui_t, Y->ui_col, Y->ui_nc, Y->sn_rowptr, Y->sn_lx, Y->Lx,
                         Y->upart, (const int*)Y->d_info);
    // subtree partition: everything below the cut is done -- the sum over the ranks, then the replicated top
    if(H.part_nranks > 1 && l == H.cut_level) { DLG_LAUNCH_CHECK(); DLG_CHECK(sparse_partition_reduce(b)); }
  }
  if(!H.ms_sn.empty())
    hipLaunchKernelGGL(k_copy_top, dim3((unsigned)H.ms_sn.size()), dim3(TPB), 0, st, Y->ms_sn, Y->sn_c0,
                       Y->sn_rowptr, Y->sn_lx, Y->sn_top, Y->Lx, Y->top_scr, (const int*)Y->d_info);
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}

// Host only (no GPU): the schedule of the one-launch region of the factorisation for a pattern as a chip with
// `ncu` compute units would get it -- and a check of everything the kernel takes on trust: every destination
// inside the workgroup's LDS, the slices of a supernode's replicas covering its update matrix exactly once,
// children listed before their parents.  stats = {first level, supernodes, workgroups, LDS bytes, sliced
// workgroups, supernodes whose update matrix stays in HBM}.  Returns DLG_ERR_STATE with a message on a violation.
extern "C" int dlg_sparse_region_probe(int N, int M, const int* colptr, const int* rowidx, int ncu, long* stats, int nstats)
{
  dlg_backend b;
  b.type = DLG_SPARSE; b.N = N; b.M = M; b.nnz = colptr[M]; b.ncu = ncu; b.row0 = 0; b.row1 = M; b.mloc = M;
  SparseSym Y;
  b.sym = &Y;
  char err[512];
  if(sym_analyze(Y.H, N, M, colptr, rowidx, 0, M, err, sizeof(err))) { b.sym = nullptr; dlg_set_error("symbolic analysis: %s", err); return DLG_ERR_ARG; }
  const int rc = sparse_factor_setup(&b, true);
  b.sym = nullptr;
  if(rc != DLG_OK) return rc;
  const SymHost& H = Y.H;
  long nsliced = 0, nhbm = 0;
  auto fail = [&](const char* what, size_t g) { dlg_set_error("one-launch region: %s (workgroup %zu)", what, g); return DLG_ERR_STATE; };
  std::vector<long> covered;
  for(size_t g = 0; g < Y.pr_item_h.size(); g++)
  {
    const FwItem& it = Y.pr_item_h[g];
    const long mb = it.nrows - it.w, ldp = (it.nrows + 1) & ~1L, pan = ldp*it.w, ntri = mb*(mb + 1)/2;
    const long T = (mb + 15) >> 4;
    if(it.rep == 0) covered.assign((size_t)std::max<long>(T, 1), 0);
    for(long t = std::min<long>(it.tj0, T); t < std::min<long>(it.tj1, T); t++) covered[(size_t)t]++;
    long lds_end;
    if(it.sliced)
    {
      nsliced++;
      const long jA = std::min<long>(16L*it.tj0, mb), jB = it.tj1 >= T ? mb : std::min<long>(16L*it.tj1, mb);
      if(it.eA != jA*mb - jA*(jA - 1)/2 || it.eB != jB*mb - jB*(jB - 1)/2) return fail("slice bounds do not match its tile columns", g);
      lds_end = pan + (it.eA & 1) + (it.eB - it.eA) + 1;             // + the scratch slot
    }
    else
    {
      if(it.jsp < 0 && mb > 0 && it.u_off >= 0)
      {
        if(it.rep == 0) nhbm++;
        // (replicas of an update matrix that is summed in HBM own the packed entries [eA, eB): their tile columns)
        if(it.rsv2 > 1)
        {
          const long jA = std::min<long>(16L*it.tj0, mb), jB = it.tj1 >= T ? mb : std::min<long>(16L*it.tj1, mb);
          if(it.eA != jA*mb - jA*(jA - 1)/2 || it.eB != jB*mb - jB*(jB - 1)/2) return fail("an HBM replica's stretch does not match its tile columns", g);
        }
        else if(it.tj0 != 0 || it.tj1 < T) return fail("a single workgroup that does not form the whole update matrix", g);
      }
      // (the kernel stages the whole update matrix behind the panel when the supernode has children, or -- childless --
      // when the launch stages those too)
      const bool staged = it.jsp >= 0 && mb > 0 && it.u_off >= 0 && (it.nch > 0 || Y.pr_stage);
      lds_end = pan + (staged ? sym_w_linear(mb, it.jsp) : 0) + 1;
    }
    if(lds_end*8 > Y.pr_lds) return fail("its LDS need exceeds the launch's", g);
    if(Y.pr_lds > FAC_LDS_BUDGET) return fail("the launch's LDS exceeds the budget", g);
    for(int k = 0; k < it.nch; k++)
    {
      if((size_t)(it.ch0 + k) >= Y.pr_rec_h.size()) return fail("children record out of range", g);
      const MfChild& rc2 = Y.pr_rec_h[it.ch0 + k];
      if(rc2.rsv >= 0)
      {
        const long ci = rc2.rsv & 0xfffff, cn = rc2.rsv >> 20;
        if(cn < 1 || ci + cn > (long)g) return fail("a child's workgroups do not precede their parent", g);
      }
      if(rc2.dst_off < 0 || (size_t)(rc2.dst_off + rc2.npad) > Y.pr_dst_h.size()) return fail("destination list out of range", g);
      if(rc2.u_off < 0 || rc2.u_off + rc2.npad > H.uscr_size) return fail("a child's update matrix outside the scratch", g);
      for(long e = 0; e < rc2.npad; e++)
      {
        const long d = Y.pr_dst_h[(size_t)(rc2.dst_off + e)];
        if(it.sliced || it.jsp >= 0) { if(d >= pan + (it.sliced ? (it.eA & 1) + (it.eB - it.eA) : sym_w_linear(mb, it.jsp)) + 1) return fail("a destination behind the workgroup's LDS", g); }
        else if(!(d & 0x8000) ? d >= pan + 1 : (d & 0x7fff) > ntri) return fail("a destination outside panel / update matrix", g);
        else if((d & 0x8000) && it.rsv2 > 1 && ((d & 0x7fff) < it.eA || (d & 0x7fff) >= it.eB)) return fail("an HBM replica adds to an entry it does not own", g);
      }
    }
    const bool last = g + 1 == Y.pr_item_h.size() || Y.pr_item_h[g + 1].rep == 0;
    if(last && mb > 0 && it.u_off >= 0)
      for(long t = 0; t < T; t++) if(covered[(size_t)t] != 1) return fail("a tile column of the update matrix is not formed exactly once", g);
  }
  const long v[] = { (long)Y.pr_level0, (long)(H.fw_lvl_ptr[H.nlevels] - (Y.pr_level0 < H.nlevels ? H.fw_lvl_ptr[Y.pr_level0] : H.fw_lvl_ptr[H.nlevels])),
                     (long)Y.pr_nwg, (long)Y.pr_lds, nsliced, nhbm };
  for(int i = 0; i < nstats && i < 6; i++) stats[i] = v[i];
  return DLG_OK;
}
