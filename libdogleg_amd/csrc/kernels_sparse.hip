// kernels_sparse.hip -- DOGLEG_SPARSE hot path on gfx950.
//
//   K1  Jt_x = Jt*x            replaces mul_spmatrix_densevector        (dogleg.c:249-261)
//   K3/K8  |J v|^2             replaces norm2_mul_spmatrix_t_densevector (dogleg.c:262-281)
//   K4  JtJ assembly           (CHOLMOD forms A*A' internally: dogleg.c:659-664)
//   K5  supernodal Cholesky    replaces cholmod_factorize[_p]            (dogleg.c:659-664)
//   K6  triangular solves      replaces cholmod_solve(CHOLMOD_A)         (dogleg.c:853)
//
// Everything is gather-based and atomics-free: every output (a JtJ block, a
// Jt_x block, a panel column range, a right-hand-side row) has exactly one
// owner wave/workgroup that sums its contributions in a fixed order, so the
// results are bitwise reproducible run to run.  The schedules come from the
// host symbolic phase (sparse_symbolic.cpp).
//
// HBM layout: Jacobian values stay in the callback's CSC order (one H2D DMA);
// the factor is a set of dense column-major supernode panels in one buffer Lx;
// JtJ is assembled straight into those panels (no separate JtJ array).
#include "dlg_internal.h"
#include "sparse_symbolic.h"
#include "panel_factor.h"

namespace {

constexpr int TPB = 256;
constexpr int LDS_BUDGET = 147456;      // bytes of dynamic LDS we allow a workgroup
constexpr int FAC_LDS_BUDGET = 163840 - 3584;  // the panel factorisation takes (almost) all 160 KB of a CU; the rest is its static LDS

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for(int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// copy `total` doubles with U independent global loads in flight per thread before
// the first dependent store (a plain load->store loop keeps ONE load in flight and
// pays a full memory latency per iteration)
template <int NT, int U, class LoadF, class StoreF>
__device__ __forceinline__ void batched_copy(int total, int tid, LoadF ld, StoreF st)
{
  for(int base = 0; base < total; base += U*NT)
  {
    double v[U];
#pragma unroll
    for(int u = 0; u < U; u++) { const int idx = base + u*NT + tid; v[u] = (idx < total) ? ld(idx) : 0.0; }
#pragma unroll
    for(int u = 0; u < U; u++) { const int idx = base + u*NT + tid; if(idx < total) st(idx, v[u]); }
  }
}

template <class T> int upload(T*& dev, const std::vector<T>& h)
{
  dev = nullptr;
  const size_t bytes = sizeof(T)*(h.size() ? h.size() : 1);
  DLG_HIP(hipMalloc(&dev, bytes));
  if(!h.empty()) DLG_HIP(hipMemcpy(dev, h.data(), sizeof(T)*h.size(), hipMemcpyHostToDevice));
  return DLG_OK;
}

} // namespace

struct SparseSym
{
  SymHost H;
  // schedules on the device
  int *sn_c0 = nullptr, *sn_rowptr = nullptr, *sn_rows = nullptr, *sn_scr = nullptr, *lvl_sn = nullptr;
  int64_t *sn_lx = nullptr, *diagpos = nullptr;
  int *ui_t = nullptr, *ui_col = nullptr, *ui_nc = nullptr, *ui_ptr = nullptr;
  SymSub* usub = nullptr; int* relpos = nullptr;
  int *uw_item = nullptr, *uw_s0 = nullptr, *uw_s1 = nullptr, *uf_item = nullptr, *uf_n = nullptr;
  int64_t *uw_part = nullptr, *uf_off = nullptr;
  double* upart = nullptr; double* uscr = nullptr; int64_t *u_off = nullptr, *usub_u = nullptr;
  SymOutBlock* oblk = nullptr; SymContrib* contrib = nullptr;
  SymTask *jtx_task = nullptr;
  int *jtx_fin_ptr = nullptr, *jtx_fin_blk = nullptr;
  AsmRho* asm_rho = nullptr; AsmPair* asm_pair = nullptr; AsmSlot* asm_slot = nullptr;
  AsmBatch* asm_batch = nullptr; AsmTask* asm_ctask = nullptr; AsmFin* asm_cfin = nullptr;
  AsmShape* asm_shape = nullptr; AsmKG* asm_kg = nullptr; AsmMTask* asm_mtask = nullptr; int* asm_tdest = nullptr;
  AsmFin2* asm_fin2 = nullptr; int64_t* asm_fin2_list = nullptr; AsmRun* asm_run = nullptr; int* asm_pdest = nullptr;
  int *rl_ptr = nullptr, *rl_pos = nullptr, *perm = nullptr, *col_sn = nullptr;
  int *fw_sn = nullptr, *fw_r0 = nullptr, *fw_r1 = nullptr, *ms_sn = nullptr; int64_t* sn_top = nullptr;
  int *sn_bd_ptr = nullptr, *sn_bd_col = nullptr;
  double* top_scr = nullptr;
  const double* aug_rhs = nullptr;        // rhs the augmented rows of the current factor were built from
  int *Jp = nullptr, *Ji = nullptr;       // rank-local pattern (row pointers rebased to 0)
  int *nv_chunk = nullptr; int n_nv_chunks = 0;   // row runs of <= NV_CHUNK non-zeros for |Jv|^2
  // numeric buffers
  double *Lx = nullptr, *scr = nullptr, *ywork = nullptr, *asm_part = nullptr, *jtx_part = nullptr;
  int *d_info = nullptr, *h_info = nullptr;
  size_t nnz_loc = 0;
  // per-level launch parameters
  std::vector<int> fac_lds;     // bytes of LDS for the factor kernel of a level (0: panels stay in HBM)
  std::vector<int> upd_lds, upd_nw, slv_lds, bwd_lds, fac_nt, upd_coop, syrk_lds, syrk_nt, syrk_kc, bwd_nt;
  std::vector<void*> allocs;
};

namespace {

// ------------------------------------------------------------ K4 assembly ---
// Column-block centric JtJ assembly.  One wave per task; a task owns a group of
// output blocks (I,J) of one column block J ("slots", accumulated in LDS) and
// walks the row-blocks containing J in batches.  Per batch the Jacobian rows
// are staged in LDS once (coalesced segment copies) and every row-block then
// feeds all its slots at once: lanes = flattened (pair, a, b),
//     acc[slot(I)][a][b] += sum_k J[k][offI+a] * J[k][offJ+b].
// Single-task groups store straight into the supernode panels; groups split
// over several tasks (very long lists: a block every row touches) store
// partial accumulators that k_assemble_fin adds in task order.
// Everything a wave shares goes through LDS in program order (same wave), so
// no barriers are needed and waves of a workgroup are independent.
constexpr int ASM_STAGE = 512, ASM_ACC = 256, ASM_RHO = 32, ASM_PAIRS = 256;

struct AsmWaveLds
{
  double  stage[ASM_STAGE];
  double  acc[ASM_ACC];
  AsmRho  rho[ASM_RHO + 1];
  AsmPair pair[ASM_PAIRS];
};

__global__ void __launch_bounds__(TPB) k_assemble(const AsmTask* __restrict__ tasks, int ntasks,
                                                  const AsmBatch* __restrict__ batches,
                                                  const AsmRho* __restrict__ rho,
                                                  const AsmPair* __restrict__ pairs,
                                                  const AsmSlot* __restrict__ slots,
                                                  const double* __restrict__ vals,
                                                  double* __restrict__ Lx, double* __restrict__ part)
{
  __shared__ __attribute__((aligned(16))) AsmWaveLds sh[TPB/64];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x*(TPB/64) + w);
  if(wid >= ntasks) return;
  AsmWaveLds& S = sh[w];
  const AsmTask T = tasks[wid];
  const int nJ = T.nJ;
  for(int e = lane; e < T.acc_size; e += 64) S.acc[e] = 0.0;
  // uniform tasks (every row-block has the same sequence of block sizes, <= 128 products):
  // the lane -> (pair ordinal, a, b) map is computed once for the whole task
  const bool uniform = T.pad != 0;
  int uj[2] = {-1, -1}, ua[2] = {0, 0}, ub[2] = {0, 0}, uidx[2] = {0, 0};
  if(uniform && T.batch0 < T.batch1)
  {
    const int rfirst = batches[T.batch0].rho0;
    const int q0 = rho[rfirst].pair0, q1 = rho[rfirst + 1].pair0;
#pragma unroll
    for(int u = 0; u < 2; u++)
    {
      const int tgt = lane + 64*u;
      int cum = 0;
      for(int p = q0; p < q1; p++)
      {
        const int ni = (pairs[p].acc_nI >> 12) + 1, n = ni*nJ;
        if(tgt >= cum && tgt < cum + n)
        {
          const int idx = tgt - cum;
          uj[u] = p - q0; uidx[u] = idx; ub[u] = idx / ni; ua[u] = idx - ub[u]*ni;
        }
        cum += n;
      }
    }
  }
  for(int bt = T.batch0; bt < T.batch1; bt++)
  {
    const AsmBatch B = batches[bt];
    const int nr = B.rho1 - B.rho0;
    // (a) the batch's row-block records (+ the one that closes the last pair list): one load
    if(lane <= nr) S.rho[lane] = rho[B.rho0 + lane];
    __builtin_amdgcn_wave_barrier();
    const int P0 = S.rho[0].pair0, np = S.rho[nr].pair0 - P0;
    // (b) pairs + values of the whole batch, all loads issued before the first LDS store
    {
      AsmPair pv[ASM_PAIRS/64];
#pragma unroll
      for(int u = 0; u < ASM_PAIRS/64; u++) if(lane + 64*u < np) pv[u] = pairs[P0 + lane + 64*u];
      for(int rb = 0; rb < nr; rb += 16)
      {
        double v[16];
#pragma unroll
        for(int u = 0; u < 16; u++)
        {
          v[u] = 0.0;
          if(rb + u < nr)
          {
            const AsmRho R = S.rho[rb + u];
            if(R.stage_off != 0xFFFF && lane < R.nrows*R.len) v[u] = vals[R.base + lane];
          }
        }
#pragma unroll
        for(int u = 0; u < 16; u++)
          if(rb + u < nr)
          {
            const AsmRho R = S.rho[rb + u];
            if(R.stage_off != 0xFFFF && lane < R.nrows*R.len) S.stage[R.stage_off + lane] = v[u];
          }
      }
      // segments longer than one wave-load (rare: long rows)
      for(int r = 0; r < nr; r++)
      {
        const AsmRho R = S.rho[r];
        if(R.stage_off == 0xFFFF) continue;
        const int cnt = R.nrows*R.len;
        for(int e = 64 + lane; e < cnt; e += 64) S.stage[R.stage_off + e] = vals[R.base + e];
      }
#pragma unroll
      for(int u = 0; u < ASM_PAIRS/64; u++) if(lane + 64*u < np) S.pair[lane + 64*u] = pv[u];
    }
    __builtin_amdgcn_wave_barrier();
    // (c) every row-block feeds all its slots at once
    if(uniform)
    {
      for(int r = 0; r < nr; r++)
      {
        const AsmRho R = S.rho[r];
        const int p0 = R.pair0 - P0;
#pragma unroll
        for(int u = 0; u < 2; u++)
        {
          if(uj[u] < 0) continue;
          const AsmPair P = S.pair[p0 + uj[u]];
          double sum = 0.0;
          if(R.stage_off != 0xFFFF)
          {
            const double* row = S.stage + R.stage_off;
            for(int k = 0; k < R.nrows; k++, row += R.len) sum += row[P.offI + ua[u]]*row[R.offJ + ub[u]];
          }
          else
          {
            const double* row = vals + R.base;
            for(int k = 0; k < R.nrows; k++, row += R.len) sum += row[P.offI + ua[u]]*row[R.offJ + ub[u]];
          }
          S.acc[(P.acc_nI & 0xFFF) + uidx[u]] += sum;
        }
      }
    }
    else
    for(int r = 0; r < nr; r++)
    {
      const AsmRho R = S.rho[r];
      const int p0 = R.pair0 - P0, p1 = S.rho[r+1].pair0 - P0;
      int total = 0;
      for(int p = p0; p < p1; p++) total += ((S.pair[p].acc_nI >> 12) + 1)*nJ;
      for(int base = 0; base < total; base += 64)
      {
        const int tgt = base + lane;
        int cum = 0, idx = -1, nI = 1, offI = 0, accoff = 0;
        for(int p = p0; p < p1; p++)
        {
          const AsmPair P = S.pair[p];
          const int ni = (P.acc_nI >> 12) + 1, n = ni*nJ;
          if(tgt >= cum && tgt < cum + n) { idx = tgt - cum; nI = ni; offI = P.offI; accoff = P.acc_nI & 0xFFF; }
          cum += n;
        }
        if(idx >= 0)
        {
          const int b = idx / nI, a = idx - b*nI;
          double sum = 0.0;
          if(R.stage_off != 0xFFFF)
          {
            const double* row = S.stage + R.stage_off;
            for(int k = 0; k < R.nrows; k++, row += R.len) sum += row[offI + a]*row[R.offJ + b];
          }
          else
          {
            const double* row = vals + R.base;
            for(int k = 0; k < R.nrows; k++, row += R.len) sum += row[offI + a]*row[R.offJ + b];
          }
          S.acc[accoff + idx] += sum;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  // write out
  for(int sidx = 0; sidx < T.nslots; sidx++)
  {
    const AsmSlot SL = slots[T.slot0 + sidx];
    const int n = SL.nI*nJ;
    for(int idx = lane; idx < n; idx += 64)
    {
      const double v = S.acc[SL.accoff + idx];
      if(T.part < 0)
      {
        const int b = idx / SL.nI, a = idx - b*SL.nI;
        if(!SL.diag || a >= b) Lx[SL.dest + a + (int64_t)b*SL.ld] = v;
      }
      else part[T.part + SL.accoff + idx] = v;
    }
  }
}
// MFMA assembly of the column blocks whose row-blocks all share one layout (AsmShape).
// One wave per task, no LDS.  Per k-group (4 Jacobian rows) a lane gathers three values
// of its row k = lane>>4: the persistent and the transient output row m = lane&15
// (A operands, A[m][k]) and the column-block entry b = (lane&15) % nJ (B operand,
// B[k][n]); v_mfma_f64_16x16x4_f64 then gives D[m][n] += sum_k A[m][k] B[k][n].
//   persistent:  n = b,            every row of the task sums into the same D
//   transient:   n = slot*nJ + b,  B is masked to the rows of row-block `slot`, so each
//                row-block of the k-group gets its own columns; D is stored and cleared
typedef double dlg_v4d __attribute__((ext_vector_type(4)));
#ifndef DLG_ASM_U
#define DLG_ASM_U 4
#endif
constexpr int ASM_U = DLG_ASM_U;
// One wave per RUN = consecutive tasks of one shape whose k-groups are contiguous: the shape's
// lane constants are loaded once and the k-group stream is software-pipelined across the
// task boundaries (bit 12 of a k-group's meta: last of its task -> store the persistent blocks,
// move on to the next task record, which was fetched ahead).
// Per k-group the wave copies the needed window of its 4 rows into a wave-private LDS tile with
// one coalesced load (lane = (row, column)), then every lane picks its A/B operands from the
// tile: the vector-memory pipe sees one load per k-group instead of one per operand.  The
// k-group records are wave-uniform (scalar loads), fetched one iteration ahead.  All loads are
// unconditional with clamped addresses (absent rows read element 0 and are zeroed afterwards):
// the loop body is straight-line code, the only branches are uniform.
template <bool HAS_T>
__device__ __forceinline__ void asm_mfma_run(const AsmRun& R, const AsmMTask* __restrict__ tasks,
                                             const AsmShape* __restrict__ SH,
                                             const AsmKG* __restrict__ kgs, const int* __restrict__ tdest,
                                             const int* __restrict__ pdest, const double* __restrict__ vals,
                                             double* __restrict__ Lx, double* __restrict__ part, int lane,
                                             double* __restrict__ tile, int LEN)
{
  const int m = lane & 15, kq = lane >> 4;
  const int nJ = SH->nJ, MP = SH->MP, MT = SH->MT, nT = SH->nT, nJr = SH->nJr;
  const int col0 = SH->col0, ncopy = SH->ncopy, dslot = SH->dslot, rslot = SH->rslot;
  const int bs = m / nJ, bb = m - bs*nJ;
  // Tile columns this lane reads its operands from.  Column ZC of every tile row is zero: lanes
  // without a persistent / transient row, B columns outside the product and rows of another
  // row-block slot read it instead of being masked afterwards.
  const int ZC = LEN - 2;
  const bool pn = m < nJ + nJr;
  int pc = SH->pcol[m];            pc = pc >= 0 ? pc : ZC;
  int tc = HAS_T ? SH->tcol[m] : -1; tc = tc >= 0 ? tc : ZC;
  const int bcol = SH->offJ + bb;   // B column of the transient product (this lane's slot only)
  // B column of the persistent product: J's columns, then the rider's (if any)
  const int bcolP = m < nJ ? SH->offJ + m : (pn ? SH->offR + (m - nJ) : ZC);
  // rows m' = kq + 4r of D this lane holds: transient (ordinal, row in block), persistent
  // (slot ordinal, row in block, offset in a partial, rows of the block)
  // (packed: these are only needed when something is stored)
  uint32_t pk1[4], pk2[4];
#pragma unroll
  for(int r = 0; r < 4; r++)
  {
    const int mm = kq + 4*r;
    const uint32_t tjv = HAS_T ? SH->tj[mm] : 0xFF, tav = HAS_T ? SH->ta[mm] : 0;
    const uint32_t psv = mm < MP ? SH->pslot[mm] : 0xFF;
    pk1[r] = tjv | tav << 8 | psv << 16 | (uint32_t)SH->pa[mm] << 24;
    pk2[r] = (uint32_t)SH->paccoff[mm] | (uint32_t)SH->pnI[mm] << 8;
  }
#define TJ(r)  (int)(pk1[r] & 0xFF)
#define TA(r)  (int)((pk1[r] >> 8) & 0xFF)
#define PS(r)  (int)((pk1[r] >> 16) & 0xFF)
#define PA(r)  (int)(pk1[r] >> 24)
#define PAO(r) (int)(pk2[r] & 0xFF)
#define PNI(r) (int)(pk2[r] >> 8)
  dlg_v4d accP = {0.0, 0.0, 0.0, 0.0}, accT = {0.0, 0.0, 0.0, 0.0};
  // task records: current + next (fetched ahead, as ONE vector load each: lane l holds dword l of
  // the record, fields are broadcast with readlane; vector loads return in order, so prefetches
  // overlap with the rest -- scalar loads would share a counter with the LDS traffic);
  // persistent destinations of both (16 per task)
  const int tlast = R.task1 - 1;
  int tix = R.task0;
  auto task_fetch = [&](int t) { return reinterpret_cast<const int*>(tasks + min(t, tlast))[min(lane, 11)]; };
  int tcv = task_fetch(tix), tnv = task_fetch(tix + 1);
  int pdc = pdest[16*(int64_t)min(tix, tlast) + m], pdn = pdest[16*(int64_t)min(tix + 1, tlast) + m];
  int64_t Tpart, Trpart, colT, colP;     // current task: partial offsets; Lx offset of this lane's column
  auto task_unpack = [&](int v) {
    const int ld = __builtin_amdgcn_readlane(v, 4);
    const int64_t panel = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(v, 7) << 32) | (uint32_t)__builtin_amdgcn_readlane(v, 6));
    Tpart  = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(v, 9) << 32) | (uint32_t)__builtin_amdgcn_readlane(v, 8));
    Trpart = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(v, 11) << 32) | (uint32_t)__builtin_amdgcn_readlane(v, 10));
    colT = panel + (int64_t)bb*ld; colP = panel + (int64_t)m*ld; };
  task_unpack(tcv);
  // k-group records of one iteration: ASM_U*6 dwords, one vector load, prefetched one iteration ahead
  static_assert(ASM_U*6 <= 64, "k-group records of an iteration must fit one wave load");
  const int kglast = R.kg1 - 1;
  const int krec = min(lane, ASM_U*6 - 1)/6, kw = min(lane, ASM_U*6 - 1) - 6*krec;
  auto kg_fetch = [&](int kg0) { return reinterpret_cast<const int*>(kgs + min(kg0 + krec, kglast))[kw]; };
  int gnv = kg_fetch(R.kg0);
  double* myrow = tile + kq*LEN;
#pragma unroll
  for(int u = 0; u < ASM_U; u++) myrow[u*4*LEN + ZC] = 0.0;
  for(int kg = R.kg0; kg < R.kg1; kg += ASM_U)
  {
    const int gv = gnv;
    uint32_t meta[ASM_U];
    int td[ASM_U];
#pragma unroll
    for(int u = 0; u < ASM_U; u++) meta[u] = kg + u <= kglast ? (uint32_t)__builtin_amdgcn_readlane(gv, 6*u + 5) : 0u;
    // (a) one coalesced copy of the rows' windows into the tile (absent rows: zeros)
    for(int c0 = 0; c0 < ncopy; c0 += 16)
    {
      double v[ASM_U];
      int b[ASM_U];
#pragma unroll
      for(int u = 0; u < ASM_U; u++)
      {
        b[u] = __builtin_amdgcn_ds_bpermute(4*(6*u + kq), gv);
        if(kg + u > kglast) b[u] = -1;
        v[u] = vals[max(b[u], 0) + col0 + min(c0 + m, ncopy - 1)];
      }
#pragma unroll
      for(int u = 0; u < ASM_U; u++) myrow[u*4*LEN + c0 + m] = b[u] >= 0 ? v[u] : 0.0;
    }
    if(HAS_T)
    {
#pragma unroll
      for(int u = 0; u < ASM_U; u++)
      {
        // transient destinations of this k-group: entry `lane` of its list (slot-major)
        const int nent = (int)((meta[u] >> 8) & 7)*nT;
        td[u] = tdest[lane < nent ? __builtin_amdgcn_readlane(gv, 6*u + 4) + lane : 0];
      }
    }
    gnv = kg_fetch(kg + ASM_U);
    __builtin_amdgcn_wave_barrier();
    // (b) operands from the tile, products
#pragma unroll
    for(int u = 0; u < ASM_U; u++)
    {
      const double* row = myrow + u*4*LEN;
      accP = __builtin_amdgcn_mfma_f64_16x16x4f64(row[pc], row[bcolP], accP, 0, 0, 0);
      if(HAS_T)
      {
        const int myslot = (meta[u] >> (2*kq)) & 3;
        accT = __builtin_amdgcn_mfma_f64_16x16x4f64(row[tc], row[bs == myslot ? bcol : ZC], accT, 0, 0, 0);
        if(meta[u] & (1u << 11))
        {
          const bool mine = bs < (int)((meta[u] >> 8) & 7);
#pragma unroll
          for(int r = 0; r < 4; r++)
            if(4*r < MT)
            {
              const int ro = __builtin_amdgcn_ds_bpermute(4*(bs*nT + TJ(r)), td[u]);
              if(mine && TJ(r) != 0xFF) Lx[colT + (ro + TA(r))] = accT[r];
            }
          accT = (dlg_v4d){0.0, 0.0, 0.0, 0.0};
        }
      }
      if(meta[u] & (1u << 12))           // end of a task: its persistent blocks
      {
#pragma unroll
        for(int r = 0; r < 4; r++)
          if(4*r < MP)
          {
            const int ro = __builtin_amdgcn_ds_bpermute(4*PS(r), pdc);
            if(PS(r) != 0xFF)
            {
              if(m < nJ)
              {
                if(Tpart < 0) { if(PS(r) != dslot || PA(r) >= m) Lx[colP + (ro + PA(r))] = accP[r]; }
                else part[Tpart + (PAO(r) + m*PNI(r))] = accP[r];
              }
              else if(pn && PS(r) == rslot) part[Trpart + ((m - nJ)*nJr + PA(r))] = accP[r];
            }
          }
        accP = (dlg_v4d){0.0, 0.0, 0.0, 0.0};
        tix++;
        task_unpack(tnv); pdc = pdn;
        tnv = task_fetch(tix + 1);
        pdn = pdest[16*(int64_t)min(tix + 1, tlast) + m];
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
#undef TJ
#undef TA
#undef PS
#undef PA
#undef PAO
#undef PNI
}
#ifdef DLG_ASM_WPE
#define ASM_WPE_ATTR __attribute__((amdgpu_waves_per_eu(DLG_ASM_WPE, DLG_ASM_WPE)))
#else
#define ASM_WPE_ATTR
#endif
__global__ void __launch_bounds__(TPB) ASM_WPE_ATTR k_assemble_mfma(const AsmRun* __restrict__ runs, int nruns,
                                                       const AsmMTask* __restrict__ tasks,
                                                       const AsmKG* __restrict__ kgs,
                                                       const AsmShape* __restrict__ shapes,
                                                       const int* __restrict__ tdest, const int* __restrict__ pdest,
                                                       const double* __restrict__ vals,
                                                       double* __restrict__ Lx, double* __restrict__ part, int LEN)
{
  extern __shared__ double asm_tiles[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x*(TPB/64) + (threadIdx.x >> 6));
  if(wid >= nruns) return;
  const AsmRun R = runs[wid];
  const AsmShape* SH = shapes + tasks[R.task0].shape;
  double* tile = asm_tiles + (threadIdx.x >> 6)*(ASM_U*4*LEN);
  if(SH->MT > 0) asm_mfma_run<true>(R, tasks, SH, kgs, tdest, pdest, vals, Lx, part, lane, tile, LEN);
  else           asm_mfma_run<false>(R, tasks, SH, kgs, tdest, pdest, vals, Lx, part, lane, tile, LEN);
}
// persistent blocks written by several MFMA tasks: fixed-order sum of the listed partials.
// k_assemble_fin2_short: one wave per block (lists of <= 32 partials);
// k_assemble_fin2_long: one 1024-thread workgroup per block, 16 waves stride over the list
__device__ __forceinline__ void fin2_store(const AsmFin2& F, int e, double v, double* __restrict__ Lx,
                                           double* __restrict__ part)
{
  if(F.to_part) { part[F.dest + e] = v; return; }
  const int b = e / F.nI, a = e - b*F.nI;
  if(!F.diag || a >= b) Lx[F.dest + a + (int64_t)b*F.ld] = v;
}
__global__ void __launch_bounds__(TPB) k_assemble_fin2_short(const AsmFin2* __restrict__ fins, int nfins,
                                                             const int64_t* __restrict__ list,
                                                             double* part, double* __restrict__ Lx)
{
  const int lane = threadIdx.x & 63;
  const int f = __builtin_amdgcn_readfirstlane(blockIdx.x*(TPB/64) + (threadIdx.x >> 6));
  if(f >= nfins) return;
  const AsmFin2 F = fins[f];
  if(lane >= F.nI*F.nJ) return;
  double s = 0.0;
  // 8 partials in flight, added in list order
  for(int k0 = 0; k0 < F.nlist; k0 += 8)
  {
    int64_t off[8];
    double v[8];
#pragma unroll
    for(int u = 0; u < 8; u++) off[u] = list[F.list0 + min(k0 + u, F.nlist - 1)];
#pragma unroll
    for(int u = 0; u < 8; u++) v[u] = part[off[u] + lane];
#pragma unroll
    for(int u = 0; u < 8; u++) if(k0 + u < F.nlist) s += v[u];
  }
  fin2_store(F, lane, s, Lx, part);
}
__global__ void __launch_bounds__(1024) k_assemble_fin2_long(const AsmFin2* __restrict__ fins,
                                                             const int64_t* __restrict__ list,
                                                             double* part, double* __restrict__ Lx)
{
  __shared__ double sh[1024];
  const AsmFin2 F = fins[blockIdx.x];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  double s = 0.0;
  if(lane < F.nI*F.nJ)
    for(int k0 = g; k0 < F.nlist; k0 += 64)
    {
      int64_t off[4];
      double v[4];
#pragma unroll
      for(int u = 0; u < 4; u++) off[u] = list[F.list0 + min(k0 + 16*u, F.nlist - 1)];
#pragma unroll
      for(int u = 0; u < 4; u++) v[u] = part[off[u] + lane];
#pragma unroll
      for(int u = 0; u < 4; u++) if(k0 + 16*u < F.nlist) s += v[u];
    }
  sh[threadIdx.x] = s;
  __syncthreads();
  if(g == 0 && lane < F.nI*F.nJ)
  {
    double tot = 0.0;
    for(int k = 0; k < 16; k++) tot += sh[k*64 + lane];
    fin2_store(F, lane, tot, Lx, part);
  }
}
// add the partial accumulators of a multi-task group in task order: one 1024-thread
// workgroup per group, 16 lanes-groups stride over the partials, fixed-order reduce
__global__ void __launch_bounds__(1024) k_assemble_fin(const AsmFin* __restrict__ fins,
                                                       const AsmSlot* __restrict__ slots,
                                                       const double* __restrict__ part,
                                                       double* __restrict__ Lx)
{
  __shared__ double sh[1024];
  const AsmFin F = fins[blockIdx.x];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  for(int ebase = 0; ebase < F.acc_size; ebase += 64)
  {
    const int e = ebase + lane;
    double s = 0.0;
    if(e < F.acc_size)
      for(int k = g; k < F.nparts; k += 16) s += part[F.part0 + (int64_t)k*F.acc_size + e];
    __syncthreads();
    sh[threadIdx.x] = s;
    __syncthreads();
    if(g == 0 && e < F.acc_size)
    {
      double tot = 0.0;
      for(int k = 0; k < 16; k++) tot += sh[k*64 + lane];
      // which slot holds accumulator e?
      for(int sidx = 0; sidx < F.nslots; sidx++)
      {
        const AsmSlot SL = slots[F.slot0 + sidx];
        const int n = SL.nI*F.nJ;
        if(e >= SL.accoff && e < SL.accoff + n)
        {
          const int idx = e - SL.accoff;
          const int b = idx / SL.nI, a = idx - b*SL.nI;
          if(!SL.diag || a >= b) Lx[SL.dest + a + (int64_t)b*SL.ld] = tot;
        }
      }
    }
  }
}
__global__ void __launch_bounds__(TPB) k_add_lambda(double* __restrict__ Lx,
                                                    const int64_t* __restrict__ diagpos, int n,
                                                    double lambda)
{
  const int i = blockIdx.x*TPB + threadIdx.x;
  if(i < n) Lx[diagpos[i]] += lambda;
}

// augmented row: panel(last row, column k) = rhs[perm[k]]
__global__ void __launch_bounds__(TPB) k_set_aug_row(double* __restrict__ Lx, const int* __restrict__ col_sn,
                                                     const int* __restrict__ sn_c0,
                                                     const int* __restrict__ sn_rowptr,
                                                     const int64_t* __restrict__ sn_lx,
                                                     const int* __restrict__ perm,
                                                     const double* __restrict__ rhs, int n)
{
  const int k = blockIdx.x*TPB + threadIdx.x;
  if(k >= n) return;
  const int s = col_sn[k];
  const int nrows = sn_rowptr[s+1] - sn_rowptr[s];
  Lx[sn_lx[s] + (nrows - 1) + (int64_t)(k - sn_c0[s])*nrows] = rhs[perm[k]];
}

// ------------------------------------------------------------------ K1 ------
// one wave per task over the diagonal block of a var-block I:
//   Jt_x[I] = sum_{row-blocks containing I} sum_k J[k][offI + a] * x[r0 + k]
// lanes = (a, j): j strides over the contributions; fixed-order LDS reduction.
__global__ void __launch_bounds__(TPB) k_jtx(const SymTask* __restrict__ tasks, int ntasks,
                                             const SymOutBlock* __restrict__ oblk,
                                             const SymContrib* __restrict__ contrib,
                                             const double* __restrict__ vals,
                                             const double* __restrict__ x,
                                             double* __restrict__ jtx, double* __restrict__ part)
{
  __shared__ double sh[TPB];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wid = blockIdx.x*(TPB/64) + w;
  double acc = 0.0;
  int nI = 1, J = 1, a = 0, j = 0;
  SymTask T = {0, 0, 0, -1};
  SymOutBlock B; B.nI = 1; B.var0 = 0;
  const bool live = wid < ntasks;
  if(live)
  {
    T = tasks[wid];
    B = oblk[T.blk];
    nI = B.nI; J = 64/nI; a = lane % nI; j = lane / nI;
    if(j < J)
      for(int c = T.c0 + j; c < T.c1; c += J)
      {
        const SymContrib C = contrib[c];
        const double* row = vals + C.base + C.offI + a;
        for(int k = 0; k < C.nrows; k++, row += C.len) acc += row[0]*x[C.r0 + k];
      }
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  if(live && lane < nI)
  {
    double s = 0.0;
    for(int jj = 0; jj < J; jj++) s += sh[w*64 + lane + jj*nI];
    if(T.part < 0) jtx[B.var0 + lane] = s;
    else part[(size_t)T.part*8 + lane] = s;
  }
}
// one workgroup per multi-chunk var-block: 32 groups x 8 scalars
__global__ void __launch_bounds__(TPB) k_jtx_fin(const int* __restrict__ fin_ptr,
                                                 const int* __restrict__ fin_blk, int nfin,
                                                 const SymOutBlock* __restrict__ oblk,
                                                 const double* __restrict__ part,
                                                 double* __restrict__ jtx)
{
  __shared__ double sh[TPB];
  const int f = blockIdx.x;
  const int a = threadIdx.x & 7, g = threadIdx.x >> 3;
  const SymOutBlock B = oblk[fin_blk[f]];
  double s = 0.0;
  for(int p = fin_ptr[f] + g; p < fin_ptr[f+1]; p += 32) s += part[(size_t)p*8 + a];
  sh[threadIdx.x] = s;
  __syncthreads();
  if(g == 0 && a < B.nI)
  {
    double tot = 0.0;
    for(int k = 0; k < 32; k++) tot += sh[k*8 + a];
    jtx[B.var0 + a] = tot;
  }
}

// --------------------------------------------------------------- K3 / K8 ---
// |J v|^2, "CSR-stream": a workgroup owns a run of consecutive measurement rows
// holding <= NV_CHUNK non-zeros.  The values and indices of the run are read
// fully coalesced (a thread-per-row loop strides by the row length and
// over-fetched 12x in rocprof), the products vals*v[idx] are parked in LDS, and
// each thread then sums the products of one row.  Deterministic: fixed
// row->thread map, ordered partials.
constexpr int NV_CHUNK = 2048;

__global__ void __launch_bounds__(TPB) k_norm2_Jv(const int* __restrict__ chunk_row,
                                                  const int* __restrict__ Jp,
                                                  const int* __restrict__ Ji,
                                                  const double* __restrict__ vals,
                                                  const double* __restrict__ v,
                                                  double* __restrict__ part)
{
  __shared__ double prod[NV_CHUNK];
  __shared__ double sh[4];
  const int r0 = chunk_row[blockIdx.x], r1 = chunk_row[blockIdx.x + 1];
  const int q0 = Jp[r0], n = Jp[r1] - q0;
  const int tid = threadIdx.x;
  double acc = 0.0;
  if(n <= NV_CHUNK)
  {
    for(int base = 0; base < n; base += 8*TPB)
    {
      double pv[8]; int pi[8];
#pragma unroll
      for(int u = 0; u < 8; u++) { const int e = base + u*TPB + tid; pi[u] = (e < n) ? Ji[q0 + e] : 0; pv[u] = (e < n) ? vals[q0 + e] : 0.0; }
#pragma unroll
      for(int u = 0; u < 8; u++) { const int e = base + u*TPB + tid; if(e < n) prod[e] = pv[u]*v[pi[u]]; }
    }
    __syncthreads();
    for(int r = r0 + tid; r < r1; r += TPB)
    {
      const int a = Jp[r] - q0, bq = Jp[r+1] - q0;
      double d = 0.0;
      for(int q = a; q < bq; q++) d += prod[q];
      acc += d*d;
    }
  }
  else
  {
    // a single row longer than the chunk: the whole workgroup reduces it
    double d = 0.0;
    for(int e = tid; e < n; e += TPB) d += vals[q0 + e]*v[Ji[q0 + e]];
    d = wave_sum(d);
    if((tid & 63) == 0) sh[tid >> 6] = d;
    __syncthreads();
    if(tid == 0) { const double t = (sh[0] + sh[1]) + (sh[2] + sh[3]); acc = t*t; }
    __syncthreads();
  }
  acc = wave_sum(acc);
  if((tid & 63) == 0) sh[tid >> 6] = acc;
  __syncthreads();
  if(tid == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// ------------------------------------------------------------------ K5 ------
// factor one supernode panel per workgroup: thread-per-row, left-looking over
// column blocks of 8.
//   (1) every thread brings the 8 block-column entries of its row(s) up to date
//       against all previous columns: per previous column one own LDS read and
//       the 8 entries of the block rows as 4 broadcast ds_read_b128 -> 8 FMAs;
//   (2) barrier; every thread factors the 8x8 diagonal block redundantly in
//       registers (no broadcast step, no extra barrier on the critical path);
//   (3) barrier; forward substitution of the thread's row against the 8x8 factor.
// 3 barriers per 8 columns instead of 2 per column, ~1.6 LDS reads per FMA
// instead of 3.  Panel in LDS with an even leading dimension (16-B aligned
// broadcast reads).  USE_LDS == false: panels larger than the LDS budget are
// factored in place in HBM with the same code path (slow, rare).
// One workgroup per work item = (supernode, slice [r0,r1) of its below rows): the LDS
// panel holds the w x w top block plus the slice; slices of one supernode factor the top
// block redundantly (identical arithmetic), slice 0 publishes it.
template <int NT>
__global__ void __launch_bounds__(NT) k_factor_level(const int* __restrict__ fw_sn,
                                                     const int* __restrict__ fw_r0,
                                                     const int* __restrict__ fw_r1,
                                                     const int* __restrict__ sn_c0,
                                                     const int* __restrict__ sn_rowptr,
                                                     const int64_t* __restrict__ sn_lx,
                                                     const int64_t* __restrict__ sn_top,
                                                     const int* __restrict__ sn_bd_ptr,
                                                     const int* __restrict__ sn_bd_col,
                                                     double* __restrict__ Lx,
                                                     double* __restrict__ top_scr,
                                                     int* __restrict__ info)
{
  extern __shared__ __attribute__((aligned(16))) double P[];
  __shared__ int sbad;
  const int s = fw_sn[blockIdx.x], r0 = fw_r0[blockIdx.x], r1 = fw_r1[blockIdx.x];
  const int w = sn_c0[s+1] - sn_c0[s];
  const int nrows = sn_rowptr[s+1] - sn_rowptr[s];
  double* G = Lx + sn_lx[s];
  const int tid = threadIdx.x;
  const int nloc = w + (r1 - r0);             // rows held by this workgroup
  const int ldp = (nloc + 1) & ~1;
  const int shift = r0;                        // local row i >= w  <->  panel row i + shift
  if(tid == 0) sbad = 0x7fffffff;
  // thread = panel row, 16 columns in flight (no index arithmetic per element)
  for(int i = tid; i < nloc; i += NT)
  {
    const double* gp = G + (i < w ? i : i + shift);
    for(int j0 = 0; j0 < w; j0 += 16)
    {
      double v[16];
#pragma unroll
      for(int u = 0; u < 16; u++) v[u] = (j0 + u < w) ? gp[(size_t)(j0 + u)*nrows] : 0.0;
#pragma unroll
      for(int u = 0; u < 16; u++) if(j0 + u < w) P[i + (j0 + u)*ldp] = v[u];
    }
  }
  __syncthreads();
  const int nmem = sn_bd_ptr[s+1] - sn_bd_ptr[s];
  if(nmem > 0) panel_factor_blockdiag<NT>(P, ldp, nloc, w, tid, sn_bd_col + sn_bd_ptr[s], nmem, &sbad, sn_c0[s]);
  else if(NT >= 256) panel_factor_mfma<(NT >= 256 ? NT : 256)>(P, ldp, nloc, w, tid, &sbad, sn_c0[s]);
  else         panel_factor<NT, true, true>(P, ldp, nloc, w, tid, &sbad, sn_c0[s]);
  const int64_t top = sn_top[s];
  if(r0 == 0 && tid == 0 && sbad != 0x7fffffff) atomicMin(info, sbad);
  for(int i = tid; i < nloc; i += NT)
  {
    // rows below the top block go back to the panel; the top block too unless the supernode is
    // cut into slices (then slice 0 parks it in top_scr, see k_copy_top)
    double* gp; size_t gs;
    if(i >= w)        { gp = G + (i + shift); gs = (size_t)nrows; }
    else if(top < 0)  { gp = G + i; gs = (size_t)nrows; }
    else if(r0 == 0)  { gp = top_scr + top + i; gs = (size_t)w; }
    else continue;
    for(int j0 = 0; j0 < w; j0 += 16)
    {
      double v[16];
#pragma unroll
      for(int u = 0; u < 16; u++) v[u] = (j0 + u < w) ? P[i + (j0 + u)*ldp] : 0.0;
#pragma unroll
      for(int u = 0; u < 16; u++) if(j0 + u < w) gp[(size_t)(j0 + u)*gs] = v[u];
    }
  }
}
// publish the top blocks of the multi-slice supernodes
__global__ void __launch_bounds__(TPB) k_copy_top(const int* __restrict__ ms_sn, const int* __restrict__ sn_c0,
                                                  const int* __restrict__ sn_rowptr,
                                                  const int64_t* __restrict__ sn_lx,
                                                  const int64_t* __restrict__ sn_top,
                                                  double* __restrict__ Lx,
                                                  const double* __restrict__ top_scr)
{
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
                                                     double* __restrict__ upart)
{
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
// block, wd columns), lower triangle, column-major with leading dimension mb, into the scratch.
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
        if(i < mb && j <= i) U[i + (size_t)j*mb] = c4[q][r];
      }
    }
}
// Phase 2: a work unit (chunk of the sub-tasks of one target column block) adds the column
// blocks of the U_d it is fed from into wave-private LDS slabs (waves take sub-tasks round
// robin), sums the slabs in wave order and applies / stores the result like the other update
// kernels.
__global__ void __launch_bounds__(TPB) k_update_gather(int unit0, const int* __restrict__ uw_item,
                                                       const int* __restrict__ uw_s0,
                                                       const int* __restrict__ uw_s1,
                                                       const int64_t* __restrict__ uw_part,
                                                       const int* __restrict__ ui_t,
                                                       const int* __restrict__ ui_col,
                                                       const int* __restrict__ ui_nc,
                                                       const SymSub* __restrict__ usub,
                                                       const int64_t* __restrict__ usub_u,
                                                       const int* __restrict__ relpos,
                                                       const int* __restrict__ sn_rowptr,
                                                       const int64_t* __restrict__ sn_lx,
                                                       double* __restrict__ Lx,
                                                       double* __restrict__ upart,
                                                       const double* __restrict__ uscr, int nw)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
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
    // the records of this wave's sub-tasks (s0 + w, s0 + w + nw, ...): lane l fetches record l of
    // the current batch of 64 with one load per field, the loop broadcasts them with readlane;
    // two sub-tasks are in flight at a time
    const int nmine = (s1 - s0 - w + nw - 1)/nw;
    for(int k0 = 0; k0 < nmine; k0 += 64)
    {
      const int stl = s0 + w + (k0 + min(lane, nmine - k0 - 1))*nw;
      const SymSub R = usub[stl];
      const int64_t ru = usub_u[stl];
      const int r_rel = R.rel, r_mb = R.nrows_d - R.wd, r_m = R.m;
      const int r_ulo = (int)(uint32_t)ru, r_uhi = (int)(ru >> 32);
      const int nb = min(64, nmine - k0);
      for(int k = 0; k < nb; k += 2)
      {
        const int ka = k, kb = min(k + 1, nb - 1);
        const bool two = k + 1 < nb;
        const int relA = __builtin_amdgcn_readlane(r_rel, ka), relB = __builtin_amdgcn_readlane(r_rel, kb);
        const int mbA = __builtin_amdgcn_readlane(r_mb, ka), mbB = __builtin_amdgcn_readlane(r_mb, kb);
        const int mA = __builtin_amdgcn_readlane(r_m, ka), mB = two ? __builtin_amdgcn_readlane(r_m, kb) : 0;
        const double* UA = uscr + (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(r_uhi, ka) << 32) | (uint32_t)__builtin_amdgcn_readlane(r_ulo, ka));
        const double* UB = uscr + (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(r_uhi, kb) << 32) | (uint32_t)__builtin_amdgcn_readlane(r_ulo, kb));
        for(int i0 = 0; i0 < max(mA, mB); i0 += 64)
        {
          const int i = i0 + lane;
          const int cmax = (i < nc - 1) ? i : nc - 1;
          const int iA = min(i, mA - 1), iB = min(i, max(mB, 1) - 1);
          const int rA = relpos[relA + iA], rB = relpos[relB + iB];
          double vA[8], vB[8];
#pragma unroll
          for(int c = 0; c < 8; c++)
            if(c < nc)
            {
              const int cc = min(c, cmax);          // entries above the diagonal are never written
              vA[c] = UA[iA + (size_t)cc*mbA]; vB[c] = UB[iB + (size_t)cc*mbB];
            }
#pragma unroll
          for(int c = 0; c < 8; c++)
            if(c < nc)
            {
              if(i < mA && c <= cmax) acc[rA + c*nrows_t] += vA[c];
            }
#pragma unroll
          for(int c = 0; c < 8; c++)
            if(c < nc)
            {
              if(i < mB && c <= cmax) acc[rB + c*nrows_t] += vB[c];
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
                                                     double* __restrict__ upart, int nw)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
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
                                                      double* __restrict__ upart, int nw)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
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
// sum the partial slabs of a multi-chunk item and apply them: the slab elements
// are spread over the lanes, the partials over 256/64 = 4 (or, for small slabs,
// up to 32) groups; fixed-order LDS reduction keeps the result deterministic
__global__ void __launch_bounds__(TPB) k_update_fin(int f0, const int* __restrict__ uf_item,
                                                    const int* __restrict__ uf_n,
                                                    const int64_t* __restrict__ uf_off,
                                                    const int* __restrict__ ui_t,
                                                    const int* __restrict__ ui_col,
                                                    const int* __restrict__ ui_nc,
                                                    const int* __restrict__ sn_rowptr,
                                                    const int64_t* __restrict__ sn_lx,
                                                    double* __restrict__ Lx,
                                                    const double* __restrict__ upart)
{
  __shared__ double sh[TPB];
  const int f = f0 + blockIdx.x;
  const int item = uf_item[f], n = uf_n[f];
  const int t = ui_t[item], col = ui_col[item], nc = ui_nc[item];
  const int nrows_t = sn_rowptr[t+1] - sn_rowptr[t];
  double* Lt = Lx + sn_lx[t] + (int64_t)col*nrows_t;
  const int slab = nrows_t*nc;
  const double* src = upart + uf_off[f];
  // E lanes per element-chunk, G groups over the partials
  const int E = (slab >= 128) ? 256 : (slab >= 64 ? 64 : (slab >= 32 ? 32 : 8));
  const int G = TPB/E;
  const int el = threadIdx.x % E, g = threadIdx.x / E;
  for(int ebase = 0; ebase < slab; ebase += E)
  {
    const int e = ebase + el;
    double tot = 0.0;
    if(e < slab) for(int k = g; k < n; k += G) tot += src[(size_t)k*slab + e];
    __syncthreads();
    sh[threadIdx.x] = tot;
    __syncthreads();
    if(g == 0 && e < slab)
    {
      double sacc = 0.0;
      for(int k = 0; k < G; k++) sacc += sh[k*E + el];
      Lt[e] -= sacc;
    }
  }
}

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
                                                            double* __restrict__ out, int use_aug)
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
  for(int i = tid; i < r; i += BWD_NT) xb[i] = ywork[rows[w + i]];
  for(int e = tid; e < nblk*64; e += BWD_NT)
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

// ================================================================ host side ==
int sparse_create(dlg_backend* b) { (void)b; return DLG_OK; }

size_t sparse_local_nnz(const dlg_backend* b) { return b->sym ? b->sym->nnz_loc : (size_t)b->nnz; }

void sparse_destroy(dlg_backend* b)
{
  SparseSym* Y = b->sym;
  if(!Y) return;
  for(void* p : Y->allocs) if(p) (void)hipFree(p);
  if(Y->h_info) (void)hipHostFree(Y->h_info);
  delete Y;
  b->sym = nullptr;
}

#define UP(field) do { DLG_CHECK(upload(Y->field, H.field)); Y->allocs.push_back(Y->field); } while(0)

int sparse_set_pattern(dlg_backend* b, const int* colptr, const int* rowidx)
{
  if(b->sym) { dlg_set_error("the sparsity pattern was already set"); return DLG_ERR_STATE; }
  if(colptr[b->M] != b->nnz)
  { dlg_set_error("Jt has %d entries but the backend was created for NJnnz = %d", colptr[b->M], b->nnz); return DLG_ERR_ARG; }
  SparseSym* Y = new (std::nothrow) SparseSym();
  if(!Y) { dlg_set_error("out of host memory"); return DLG_ERR_NOMEM; }
  b->sym = Y;
  char err[512];
  if(sym_analyze(Y->H, b->N, b->M, colptr, rowidx, b->row0, b->row1, err, sizeof(err)))
  { dlg_set_error("symbolic analysis: %s", err); return DLG_ERR_ARG; }
  SymHost& H = Y->H;
  UP(sn_c0); UP(sn_rowptr); UP(sn_rows); UP(sn_scr); UP(lvl_sn); UP(sn_lx); UP(diagpos);
  UP(ui_t); UP(ui_col); UP(ui_nc); UP(ui_ptr); UP(usub); UP(relpos); UP(u_off); UP(usub_u);
  UP(uw_item); UP(uw_s0); UP(uw_s1); UP(uw_part); UP(uf_item); UP(uf_n); UP(uf_off);
  UP(oblk); UP(contrib); UP(jtx_task); UP(jtx_fin_ptr); UP(jtx_fin_blk);
  UP(asm_rho); UP(asm_pair); UP(asm_slot); UP(asm_batch); UP(asm_ctask); UP(asm_cfin);
  UP(asm_shape); UP(asm_kg); UP(asm_mtask); UP(asm_tdest); UP(asm_fin2); UP(asm_fin2_list); UP(asm_run); UP(asm_pdest);
  UP(rl_ptr); UP(rl_pos); UP(perm); UP(col_sn); UP(fw_sn); UP(fw_r0); UP(fw_r1); UP(ms_sn); UP(sn_top); UP(sn_bd_ptr); UP(sn_bd_col);
  // rank-local pattern for the row-wise kernels
  {
    const int mloc = b->row1 - b->row0;
    const int q0 = colptr[b->row0], q1 = colptr[b->row1];
    Y->nnz_loc = (size_t)(q1 - q0);
    std::vector<int> jp(mloc + 1), ji(rowidx + q0, rowidx + q1);
    for(int r = 0; r <= mloc; r++) jp[r] = colptr[b->row0 + r] - q0;
    DLG_CHECK(upload(Y->Jp, jp)); Y->allocs.push_back(Y->Jp);
    DLG_CHECK(upload(Y->Ji, ji)); Y->allocs.push_back(Y->Ji);
    std::vector<int> ch; ch.push_back(0);
    for(int r = 0; r < mloc;)
    {
      int e = r;
      while(e < mloc && e - r < TPB && jp[e+1] - jp[r] <= NV_CHUNK) e++;
      if(e == r) e = r + 1;                       // one row longer than a chunk
      ch.push_back(e); r = e;
    }
    Y->n_nv_chunks = (int)ch.size() - 1;
    DLG_CHECK(upload(Y->nv_chunk, ch)); Y->allocs.push_back(Y->nv_chunk);
  }
  auto dalloc = [&](double*& p, size_t n) -> int {
    DLG_HIP(hipMalloc(&p, sizeof(double)*(n ? n : 1))); Y->allocs.push_back(p); return DLG_OK; };
  DLG_CHECK(dalloc(Y->Lx, (size_t)H.lx_size));
  DLG_CHECK(dalloc(Y->scr, (size_t)H.scr_size));
  DLG_CHECK(dalloc(Y->ywork, (size_t)H.N));
  DLG_CHECK(dalloc(Y->upart, (size_t)H.upart_size));
  DLG_CHECK(dalloc(Y->uscr, (size_t)H.uscr_size));
  DLG_CHECK(dalloc(Y->top_scr, (size_t)H.top_size));
  DLG_CHECK(dalloc(Y->asm_part, (size_t)H.asm_part_size));
  DLG_CHECK(dalloc(Y->jtx_part, (size_t)H.jtx_nparts*8));
  DLG_HIP(hipMalloc(&Y->d_info, sizeof(int))); Y->allocs.push_back(Y->d_info);
  DLG_HIP(hipHostMalloc(&Y->h_info, sizeof(int)));

  // per-level launch parameters
  Y->fac_lds.assign(H.nlevels, 0); Y->upd_lds.assign(H.nlevels, 0); Y->upd_nw.assign(H.nlevels, 0); Y->syrk_lds.assign(H.nlevels, 0); Y->bwd_nt.assign(H.nlevels, 512); Y->syrk_nt.assign(H.nlevels, 256); Y->syrk_kc.assign(H.nlevels, 4);
  Y->slv_lds.assign(H.nlevels, 0); Y->bwd_lds.assign(H.nlevels, 0); Y->fac_nt.assign(H.nlevels, 512); Y->upd_coop.assign(H.nlevels, 0);
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
      const long p = ((nloc + 1) & ~1L)*wv;        // even leading dimension in LDS
      if(p > maxp) maxp = p;
      if(nloc > maxr) maxr = nloc;
    }
    Y->fac_nt[l] = (maxr <= 128) ? 128 : (maxr <= 256 ? 256 : 512);
    Y->upd_coop[l] = (maxw > 8) ? 1 : 0;           // heavy sources: cooperative update kernel
    if(maxp*8 > FAC_LDS_BUDGET) { dlg_set_error("internal error: a factor slice does not fit LDS (%ld doubles)", maxp); return DLG_ERR_ARG; }
    Y->fac_lds[l] = (int)(maxp*8);
    Y->slv_lds[l] = (int)(maxw*(maxw | 1)*8);
    {
      long mb = 0;
      for(int i = H.lvl_ptr[l]; i < H.lvl_ptr[l+1]; i++)
      {
        const int s = H.lvl_sn[i];
        const long wv = H.sn_c0[s+1] - H.sn_c0[s], nr = H.sn_rowptr[s+1] - H.sn_rowptr[s];
        const long need = (nr - wv + 2) + 256 + ((wv + 7)/8)*64 + 16;   // xb, xs, diagonal blocks, rhs
        if(need > mb) mb = need;
      }
      if(mb*8 > LDS_BUDGET) { dlg_set_error("supernode too large for the backward-solve kernel (%ld doubles)", mb); return DLG_ERR_ARG; }
      Y->bwd_lds[l] = (int)(mb*8);
      Y->bwd_nt[l] = (maxw <= 128 && H.lvl_ptr[l+1] - H.lvl_ptr[l] >= 512) ? 256 : 512;
    }
    if(Y->slv_lds[l] > LDS_BUDGET) { dlg_set_error("supernode of width %ld is too wide for the solve kernels", maxw); return DLG_ERR_ARG; }
    long maxslab = 0;
    for(int it = H.ui_lvl_ptr[l]; it < H.ui_lvl_ptr[l+1]; it++)
    {
      const int t = H.ui_t[it];
      const long sl = (long)(H.sn_rowptr[t+1] - H.sn_rowptr[t])*H.ui_nc[it];
      if(sl > maxslab) maxslab = sl;
    }
    int nw = 0;
    if(maxslab > 0) { nw = (int)(LDS_BUDGET/(maxslab*8)); if(nw > 4) nw = 4; }
    Y->upd_nw[l] = nw;
    Y->upd_lds[l] = (int)(maxslab*8*nw);
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
    }
  }
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_factor_level<128>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, FAC_LDS_BUDGET));
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_factor_level<512>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, FAC_LDS_BUDGET));
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_factor_level<256>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, FAC_LDS_BUDGET));
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_update_level),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET));
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_update_syrk<256>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET));
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_update_syrk<1024>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET));
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_update_gather),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET));
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_update_mfma),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET));
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_fwd_level),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET));
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_bwd_level<256>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET));
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_bwd_level<512>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET));
  return DLG_OK;
}

extern "C" int dlg_sparse_stats(dlg_backend_t* b, long* nnz_JtJ_lower, long* nnz_L, int* n_supernodes,
                                int* n_levels, double* factor_flops)
{
  if(!b || !b->sym) { dlg_set_error("no symbolic analysis yet"); return DLG_ERR_STATE; }
  const SymHost& H = b->sym->H;
  if(nnz_JtJ_lower) *nnz_JtJ_lower = (long)H.nnz_JtJ_lower;
  if(nnz_L) *nnz_L = (long)H.nnz_L;
  if(n_supernodes) *n_supernodes = H.nsn;
  if(n_levels) *n_levels = H.nlevels;
  if(factor_flops) *factor_flops = H.factor_flops;
  return DLG_OK;
}

// K1
int sparse_eval(dlg_backend* b, int s)
{
  SparseSym* Y = b->sym;
  if(!Y) { dlg_set_error("dlg_sparse_set_pattern must be called first"); return DLG_ERR_STATE; }
  DlgSlot& S = b->slot[s];
  const SymHost& H = Y->H;
  DLG_HIP(hipMemsetAsync(S.Jt_x, 0, sizeof(double)*(size_t)b->N, b->stream));
  const int nt = (int)H.jtx_task.size();
  if(nt > 0)
    hipLaunchKernelGGL(k_jtx, dim3(dlg_cdiv(nt, TPB/64)), dim3(TPB), 0, b->stream, Y->jtx_task, nt,
                       Y->oblk, Y->contrib, S.Jin(), S.xin(), S.Jt_x, Y->jtx_part);
  const int nf = (int)H.jtx_fin_blk.size();
  if(nf > 0)
    hipLaunchKernelGGL(k_jtx_fin, dim3(nf), dim3(TPB), 0, b->stream,
                       Y->jtx_fin_ptr, Y->jtx_fin_blk, nf, Y->oblk, Y->jtx_part, S.Jt_x);
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}

// K3 / K8
int sparse_norm2_Jv(dlg_backend* b, int s, const double* v, double* out_dev)
{
  SparseSym* Y = b->sym;
  if(!Y) { dlg_set_error("dlg_sparse_set_pattern must be called first"); return DLG_ERR_STATE; }
  DlgSlot& S = b->slot[s];
  const int g = Y->n_nv_chunks;
  if(g == 0) { DLG_HIP(hipMemsetAsync(out_dev, 0, sizeof(double), b->stream)); return DLG_OK; }
  DLG_CHECK(dlg_ensure_partials(b, 4096 + (size_t)g));
  double* part = b->d_part + 4096;
  hipLaunchKernelGGL(k_norm2_Jv, dim3(g), dim3(TPB), 0, b->stream, Y->nv_chunk, Y->Jp, Y->Ji, S.Jin(), v, part);
  DLG_LAUNCH_CHECK();
  return k_reduce_sum(b, part, g, out_dev);
}

// K4 + K5
int sparse_factorize(dlg_backend* b, int s, double lambda, int* ok)
{
  SparseSym* Y = b->sym;
  if(!Y) { dlg_set_error("dlg_sparse_set_pattern must be called first"); return DLG_ERR_STATE; }
  DlgSlot& S = b->slot[s];
  const SymHost& H = Y->H;
  hipStream_t st = b->stream;
  // --- K4: JtJ straight into the supernode panels
  {
    DlgProfScope pt(b, DLG_PROF_K4_TOTAL);
    DLG_HIP(hipMemsetAsync(Y->Lx, 0, sizeof(double)*(size_t)H.lx_size, st));
    const int nt = (int)H.asm_ctask.size(), nmt = (int)H.asm_mtask.size();
    if(nt > 0 || nmt > 0)
    {
      DlgProfScope pk(b, DLG_PROF_K4_KERNEL);
      if(nmt > 0)
      {
        const int nruns = (int)H.asm_run.size();
        hipLaunchKernelGGL(k_assemble_mfma, dim3(dlg_cdiv(nruns, TPB/64)), dim3(TPB),
                           sizeof(double)*(TPB/64)*ASM_U*4*H.asm_lds_len, st, Y->asm_run, nruns, Y->asm_mtask,
                           Y->asm_kg, Y->asm_shape, Y->asm_tdest, Y->asm_pdest, S.Jin(), Y->Lx, Y->asm_part,
                           H.asm_lds_len);
      }
      if(nt > 0)
        hipLaunchKernelGGL(k_assemble, dim3(dlg_cdiv(nt, TPB/64)), dim3(TPB), 0, st, Y->asm_ctask, nt,
                           Y->asm_batch, Y->asm_rho, Y->asm_pair, Y->asm_slot, S.Jin(), Y->Lx, Y->asm_part);
    }
    for(size_t q = 0; q + 2 < H.fin2_stage.size(); q += 3)
    {
      const int f0 = H.fin2_stage[q], ns = H.fin2_stage[q+1], nl = H.fin2_stage[q+2];
      if(ns > 0)
        hipLaunchKernelGGL(k_assemble_fin2_short, dim3(dlg_cdiv(ns, TPB/64)), dim3(TPB), 0, st, Y->asm_fin2 + f0, ns,
                           Y->asm_fin2_list, Y->asm_part, Y->Lx);
      if(nl > 0)
        hipLaunchKernelGGL(k_assemble_fin2_long, dim3(nl), dim3(1024), 0, st, Y->asm_fin2 + f0 + ns,
                           Y->asm_fin2_list, Y->asm_part, Y->Lx);
    }
    const int nf = (int)H.asm_cfin.size();
    if(nf > 0)
      hipLaunchKernelGGL(k_assemble_fin, dim3(nf), dim3(1024), 0, st, Y->asm_cfin, Y->asm_slot,
                         Y->asm_part, Y->Lx);
    DLG_LAUNCH_CHECK();
  }
  // rows are sharded: sum the partial JtJ of all ranks before factorising
  DLG_CHECK(dlg_allreduce_dev(b, Y->Lx, (size_t)H.lx_size));
  if(lambda != 0.0)
    hipLaunchKernelGGL(k_add_lambda, dim3(dlg_cdiv(H.N, TPB)), dim3(TPB), 0, st, Y->Lx, Y->diagpos, H.N,
                       lambda);
  // the right-hand side rides along as the last row of every panel: y = L^-1 P Jt_x falls out
  Y->aug_rhs = nullptr;
  if(S.have_Jtx)
  {
    hipLaunchKernelGGL(k_set_aug_row, dim3(dlg_cdiv(H.N, TPB)), dim3(TPB), 0, st, Y->Lx, Y->col_sn, Y->sn_c0,
                       Y->sn_rowptr, Y->sn_lx, Y->perm, S.Jt_x, H.N);
    Y->aug_rhs = S.Jt_x;
  }
  // --- K5: level-scheduled supernodal Cholesky
  DlgProfScope pf(b, DLG_PROF_K5_FACTOR);
  *Y->h_info = 0x7fffffff;
  DLG_HIP(hipMemcpyAsync(Y->d_info, Y->h_info, sizeof(int), hipMemcpyHostToDevice, st));
  for(int l = 0; l < H.nlevels; l++)
  {
    const int n = H.fw_lvl_ptr[l+1] - H.fw_lvl_ptr[l];
    if(n > 0)
    {
      const int o = H.fw_lvl_ptr[l];
      if(Y->fac_nt[l] == 128)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_factor_level<128>), dim3(n), dim3(128), Y->fac_lds[l], st,
                           Y->fw_sn + o, Y->fw_r0 + o, Y->fw_r1 + o, Y->sn_c0, Y->sn_rowptr, Y->sn_lx,
                           Y->sn_top, Y->sn_bd_ptr, Y->sn_bd_col, Y->Lx, Y->top_scr, Y->d_info);
      else if(Y->fac_nt[l] == 256)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_factor_level<256>), dim3(n), dim3(256), Y->fac_lds[l], st,
                           Y->fw_sn + o, Y->fw_r0 + o, Y->fw_r1 + o, Y->sn_c0, Y->sn_rowptr, Y->sn_lx,
                           Y->sn_top, Y->sn_bd_ptr, Y->sn_bd_col, Y->Lx, Y->top_scr, Y->d_info);
      else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_factor_level<512>), dim3(n), dim3(512), Y->fac_lds[l], st,
                           Y->fw_sn + o, Y->fw_r0 + o, Y->fw_r1 + o, Y->sn_c0, Y->sn_rowptr, Y->sn_lx,
                           Y->sn_top, Y->sn_bd_ptr, Y->sn_bd_col, Y->Lx, Y->top_scr, Y->d_info);
    }
    const int nu = H.uw_lvl_ptr[l+1] - H.uw_lvl_ptr[l];
    if(nu > 0 && H.upd_syrk[l] && Y->upd_nw[l] > 0)
    {
      const int ns = H.lvl_ptr[l+1] - H.lvl_ptr[l];
      if(Y->syrk_nt[l] == 256)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_update_syrk<256>), dim3(ns), dim3(256), Y->syrk_lds[l], st,
                           Y->lvl_sn + H.lvl_ptr[l], Y->sn_c0, Y->sn_rowptr, Y->sn_lx, Y->u_off, Y->Lx, Y->uscr,
                           Y->syrk_kc[l]);
      else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_update_syrk<1024>), dim3(ns), dim3(1024), Y->syrk_lds[l], st,
                           Y->lvl_sn + H.lvl_ptr[l], Y->sn_c0, Y->sn_rowptr, Y->sn_lx, Y->u_off, Y->Lx, Y->uscr,
                           Y->syrk_kc[l]);
      hipLaunchKernelGGL(k_update_gather, dim3(nu), dim3(TPB), Y->upd_lds[l], st, H.uw_lvl_ptr[l],
                         Y->uw_item, Y->uw_s0, Y->uw_s1, Y->uw_part, Y->ui_t, Y->ui_col, Y->ui_nc,
                         Y->usub, Y->usub_u, Y->relpos, Y->sn_rowptr, Y->sn_lx, Y->Lx, Y->upart, Y->uscr,
                         Y->upd_nw[l]);
    }
    else if(nu > 0 && Y->upd_coop[l] && Y->upd_nw[l] > 0 && !getenv("DOGLEG_AMD_NO_UPDATE_MFMA"))
      hipLaunchKernelGGL(k_update_mfma, dim3(nu), dim3(TPB), Y->upd_lds[l], st, H.uw_lvl_ptr[l],
                         Y->uw_item, Y->uw_s0, Y->uw_s1, Y->uw_part, Y->ui_t, Y->ui_col, Y->ui_nc,
                         Y->usub, Y->relpos, Y->sn_rowptr, Y->sn_lx, Y->Lx, Y->upart, Y->upd_nw[l]);
    else if(nu > 0 && Y->upd_coop[l])
      hipLaunchKernelGGL(k_update_coop, dim3(nu), dim3(TPB), 0, st, H.uw_lvl_ptr[l],
                         Y->uw_item, Y->uw_s0, Y->uw_s1, Y->uw_part, Y->ui_t, Y->ui_col, Y->ui_nc,
                         Y->usub, Y->relpos, Y->sn_rowptr, Y->sn_lx, Y->Lx, Y->upart);
    else if(nu > 0)
      hipLaunchKernelGGL(k_update_level, dim3(nu), dim3(TPB), Y->upd_lds[l], st, H.uw_lvl_ptr[l],
                         Y->uw_item, Y->uw_s0, Y->uw_s1, Y->uw_part, Y->ui_t, Y->ui_col, Y->ui_nc,
                         Y->usub, Y->relpos, Y->sn_rowptr, Y->sn_lx, Y->Lx, Y->upart, Y->upd_nw[l]);
    const int nfz = H.uf_lvl_ptr[l+1] - H.uf_lvl_ptr[l];
    if(nfz > 0)
      hipLaunchKernelGGL(k_update_fin, dim3(nfz), dim3(TPB), 0, st, H.uf_lvl_ptr[l], Y->uf_item, Y->uf_n,
                         Y->uf_off, Y->ui_t, Y->ui_col, Y->ui_nc, Y->sn_rowptr, Y->sn_lx, Y->Lx,
                         Y->upart);
  }
  if(!H.ms_sn.empty())
    hipLaunchKernelGGL(k_copy_top, dim3((unsigned)H.ms_sn.size()), dim3(TPB), 0, st, Y->ms_sn, Y->sn_c0,
                       Y->sn_rowptr, Y->sn_lx, Y->sn_top, Y->Lx, Y->top_scr);
  DLG_LAUNCH_CHECK();
  DLG_HIP(hipMemcpyAsync(Y->h_info, Y->d_info, sizeof(int), hipMemcpyDeviceToHost, st));
  if(pf.e) { dlg_prof_end(b, pf.id, pf.e); pf.e = nullptr; }
  DLG_HIP(hipStreamSynchronize(st));
  *ok = (*Y->h_info == 0x7fffffff);
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
                         Y->ywork, out, use_aug);
    else if(n > 0)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_solve_bwd_level<512>), dim3(n), dim3(512), Y->bwd_lds[l], st,
                         Y->lvl_sn + H.lvl_ptr[l], Y->sn_c0, Y->sn_rowptr, Y->sn_rows, Y->sn_lx, Y->perm, Y->Lx,
                         Y->ywork, out, use_aug);
  }
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}

// host-only: run the symbolic phase on a pattern and report its statistics
// (no GPU needed; used by the CPU test-suite and by tools/)
extern "C" int dlg_sparse_symbolic_probe(int N, int M, const int* colptr, const int* rowidx, int row0,
                                         int row1, long* stats, int nstats, int* perm_out)
{
  SymHost H;
  char err[512];
  if(sym_analyze(H, N, M, colptr, rowidx, row0, row1, err, sizeof(err)))
  { dlg_set_error("symbolic analysis: %s", err); return DLG_ERR_ARG; }
  const long v[] = { (long)H.nvb, (long)H.nsn, (long)H.nlevels, (long)H.nnz_JtJ_lower, (long)H.nnz_L,
                     (long)H.lx_size, (long)H.factor_flops, (long)H.max_panel, (long)H.asm_ctask.size(),
                     (long)H.ui_t.size(), (long)H.relpos.size(), (long)H.oblk.size(),
                     (long)H.contrib.size(), (long)H.usub.size(), (long)H.scr_size,
                     (long)H.jtx_task.size(), (long)H.asm_mtask.size(), (long)H.asm_kg.size(),
                     (long)H.asm_shape.size() };
  for(int i = 0; i < nstats && i < (int)(sizeof(v)/sizeof(v[0])); i++) stats[i] = v[i];
  if(perm_out) memcpy(perm_out, H.perm.data(), sizeof(int)*(size_t)N);
  return DLG_OK;
}
