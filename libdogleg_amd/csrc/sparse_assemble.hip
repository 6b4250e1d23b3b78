// sparse_assemble.hip -- the kernels that read the Jacobian (DOGLEG_SPARSE, gfx950):
//   K1  Jt_x = Jt*x            replaces mul_spmatrix_densevector        (dogleg.c:249-261)
//   K3/K8  |J v|^2             replaces norm2_mul_spmatrix_t_densevector (dogleg.c:262-281)
//   K4  JtJ assembly           (CHOLMOD forms A*A' internally: dogleg.c:659-664)
// See sparse_internal.h for the layout and the determinism rules.
#include "sparse_internal.h"

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
#ifdef DLG_ASM_PROFILE
// (tools only: cycles a wave spends [shape class][0 head of an iteration up to the tile writes, 1 the rest, 2 iterations, 3 waves])
__device__ unsigned long long g_asm_prof[2][4];
#endif
// (-DDLG_ASM_NO_PREFETCH: the values of an iteration's rows fetched inside it -- tools/variant_lib.sh A/B)
#ifndef DLG_ASM_NO_PREFETCH
#define DLG_ASM_PREFETCH 1
#endif
#ifndef DLG_ASM_U
#define DLG_ASM_U 4
#endif
constexpr int ASM_U = DLG_ASM_U;
// (a run's k-groups are a multiple of ASM_KG_ALIGN, sparse_symbolic.cpp: with the default unroll no k-group of an iteration
// lies past the end of its run, and nothing in the loop asks)
constexpr bool ASM_RUN_ALIGNED = ASM_KG_ALIGN % ASM_U == 0;
constexpr int ASM_TBUF = 16*17 + 64;    // doubles of that square plus the four rows' Jt*x sums of a task's end (te_ok)
constexpr int ASM_TLD = 17;             // doubles a row of the wave-private square the transient product turns through (asm_mfma_run: ts_ok)
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
// JTX: the wave also forms Jt*x of its tasks' rows -- the B operand of the persistent product IS
// J(row, column of the task's block / of the rider), so one multiply-add per k-group with x(row)
// gives the task's share of (Jt x)[block]; it leaves a 16-double record per task (jtp), summed per
// var-block by k_jtx_fin2_* in task order.  K1's own pass over J is not needed then.
template <bool HAS_T, int CLEN, bool JTX, bool XT = false, bool TS = false>
__device__ __forceinline__ void asm_mfma_run(const AsmRun& R, const AsmMTask* __restrict__ tasks,
                                             const AsmShape* __restrict__ SH,
                                             const AsmKG* __restrict__ kgs, const int* __restrict__ tdest,
                                             const int* __restrict__ pdest, const double* __restrict__ vals,
                                             double* __restrict__ Lx, double* __restrict__ part, int lane,
                                             double* __restrict__ tile, int LEN_rt,
                                             const double* __restrict__ xvec, double* __restrict__ jtp,
                                             double* __restrict__ jtx_out, double* __restrict__ tbuf, uint32_t trash_off,
                                             const uint32_t* __restrict__ pent)
{
  constexpr int KD = ASM_KG_DW;
  // CLEN > 0: the tile row stride is a compile-time constant (the usual 16-column window), so the
  // LDS offsets of the unrolled k-groups become instruction immediates
  const int LEN = CLEN > 0 ? CLEN : LEN_rt;
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
  // XT, the transient blocks of a k-group: the 16 x 16 product turns through a wave-private LDS square (ASM_TLD doubles
  // a row) and leaves ONE entry per lane, consecutive lanes = consecutive rows of one destination column -- entry L is
  // (slot, column of J, transient row) in that order.  As the product comes out of the matrix core a lane holds
  // rows kq + 4 r of column m: four masked stores a lane, each with a row offset of its own fetched across the wave,
  // 37 of the ~100 instructions of such a k-group, for 54 of 256 entries (config #4).  ts_ok: the shape's entries fit one
  // round of 64 lanes (else the four stores, below).  A property of the whole schedule, checked by the host (TS): the two
  // forms in one kernel would keep the waits for the prefetched values as blind as before.
  constexpr bool ts_ok = HAS_T && XT && TS;
  int ts_rd = 0, ts_td4 = 0, ts_bs = 99, ts_cb = 0;      // LDS index read; 4 * (entry of td holding the row offset); slot; ta + bb * ld needs ld: (ta, bb) packed
  if(ts_ok)
  {
    int nrt = 0;
    for(int q = 0; q < 16; q++) if(SH->tj[q] != 0xFF) nrt++;
    // (XT: a k-group's destinations came in its record -- two at most --, so it has at most 2 / nT row-block slots)
    const int smax = min((int)SH->smax, max(1, 2/max(nT, 1)));
    if(lane < smax*nJ*nrt)
    {
      const int q = lane / nrt, i = lane - q*nrt;
      const int sbs = q / nJ, sbb = q - sbs*nJ;
      int mmL = 0, seen = 0;
      for(int qq = 0; qq < 16; qq++) if(SH->tj[qq] != 0xFF) { if(seen == i) mmL = qq; seen++; }
      ts_rd = mmL*ASM_TLD + sbs*nJ + sbb;
      ts_td4 = 4*(sbs*nT + SH->tj[mmL]);
      ts_bs = sbs;
      ts_cb = (int)SH->ta[mmL] | sbb << 8;
    }
  }
  // TS, the END of a task: its persistent blocks and its Jt*x record leave the same way -- the accumulator and the four
  // rows' Jt*x sums through the wave's LDS square, ONE entry a lane (the table of the shape: sparse_host.hip, asm_pent),
  // one unconditional store.  In four masked rounds with nested cases (a block of J's panel or its partial, the rider's
  // partial, the lower triangle of (J, J) only) plus two shuffles and two masked stores for Jt*x the end of a task was
  // ~70 instructions -- every third k-group of config #4's point tasks.  te_ok: the shape's entries fit 64 lanes.
  // (a property of the whole schedule, with ts_ok: the host found every shape's entries in at most two rounds of 64 lanes;
  // the second round's words -- the camera blocks' 81 entries -- are fetched at the task's end, such tasks are long)
  constexpr bool te_ok = ts_ok && JTX;
  uint32_t te_w0 = 0;
  bool te_two = false;
  if(te_ok)
  {
    te_w0 = pent[lane];
    te_two = __builtin_amdgcn_readfirstlane((int)pent[64]) != 0;
  }
  double* const tb_wr = tbuf + (kq*ASM_TLD + m);
  dlg_v4d accP = {0.0, 0.0, 0.0, 0.0}, accT = {0.0, 0.0, 0.0, 0.0};
  // task records: current + next (fetched ahead, as ONE vector load each: lane l holds dword l of
  // the record, fields are broadcast with readlane; vector loads return in order, so prefetches
  // overlap with the rest -- scalar loads would share a counter with the LDS traffic);
  // persistent destinations of both (16 per task)
  const int tlast = R.task1 - 1;
  int tix = R.task0;
  auto task_fetch = [&](int t) { return reinterpret_cast<const int*>(tasks + min(t, tlast))[min(lane, ASM_MTASK_DW - 1)]; };
  int tcv = task_fetch(tix), tnv = task_fetch(tix + 1);
  int pdc = pdest[16*(int64_t)min(tix, tlast) + m], pdn = pdest[16*(int64_t)min(tix + 1, tlast) + m];
  int64_t Tpart, Trpart, colT, colP;     // current task: partial offsets; Lx offset of this lane's column
  uint32_t colTs = 0;                    // (ts_ok: Lx offset of the entry this lane stores, its row-block's offset aside)
  int64_t Tpanel = 0;                    // (te_ok: the task's panel)
  int Tjvar = -1;                        // ... first variable of J if the task writes Jt*x itself
  int Tld = 0;                           // ... rows of J's panel
  auto task_unpack = [&](int v) {
    if(JTX) Tjvar = __builtin_amdgcn_readlane(v, 12);
    const int ld = __builtin_amdgcn_readlane(v, 4);
    Tld = ld;
    const int64_t panel = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(v, 7) << 32) | (uint32_t)__builtin_amdgcn_readlane(v, 6));
    Tpart  = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(v, 9) << 32) | (uint32_t)__builtin_amdgcn_readlane(v, 8));
    Trpart = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(v, 11) << 32) | (uint32_t)__builtin_amdgcn_readlane(v, 10));
    colT = panel + (int64_t)bb*ld; colP = panel + (int64_t)m*ld; Tpanel = panel;
    if(ts_ok) colTs = (uint32_t)panel + (uint32_t)((ts_cb >> 8)*ld + (ts_cb & 0xFF)); };
  task_unpack(tcv);
  // k-group records of one iteration: ASM_U*KD dwords, one vector load, prefetched one iteration ahead
  static_assert(ASM_U*KD <= 64, "k-group records of an iteration must fit one wave load");
  const int kglast = R.kg1 - 1;
  const int krec = min(lane, ASM_U*KD - 1)/KD, kw = min(lane, ASM_U*KD - 1) - KD*krec;
  double jacc = 0.0;
  auto kg_fetch = [&](int kg0) { return reinterpret_cast<const int*>(kgs + min(kg0 + krec, kglast))[kw]; };
  int gnv = kg_fetch(R.kg0);
  // (XT: every shape of the schedule has a window of at most 15 columns -- the host checks --, so that x of a
  // row can ride in column 15 of its tile row)
#ifdef DLG_ASM_PREFETCH
  constexpr bool x_in_tile = JTX && CLEN == 18 && XT;
#else
  constexpr bool x_in_tile = false;
#endif
#ifdef DLG_ASM_PREFETCH
  // the values of the first 16 columns of an iteration's rows are fetched one iteration ahead (their
  // records two ahead): the copy into the tile finds them in registers
  int gnn;
  double vpre[ASM_U]; int bpre[ASM_U];
  // (JTX, a window of at most 15 columns: lane 15 of every row fetches x(row) instead of a duplicate of the
  // window's last column -- it lands in column 15 of the tile row, and the Jt*x product reads it from there:
  // no gather of x between the records and the products, no load of its own)
  // (one fetch across the wave per row: lane 15 of an x_in_tile row asks for the record's x row, the others for the row's
  // first value -- the word it asks for and the base it adds to are the lane's own constants)
  const int vf_sel = 4*(kq + ((x_in_tile && m == 15) ? 6 : 0));
  const double* const vf_base = (x_in_tile && m == 15) ? xvec : vals + (col0 + min(m, ncopy - 1));
  auto vals_fetch = [&](int rec, int kgi) {
#pragma unroll
    for(int u = 0; u < ASM_U; u++)
    {
      bpre[u] = __builtin_amdgcn_ds_bpermute(vf_sel + 4*KD*u, rec);
      if(!ASM_RUN_ALIGNED && kgi + u > kglast && !(x_in_tile && m == 15)) bpre[u] = -1;
      const double* src = vf_base + (uint32_t)max(bpre[u], 0);
#ifdef DLG_ASM_NO_VLOAD
      vpre[u] = (double)(((long)src >> 3) & 7);
#else
      vpre[u] = *src;
#endif
    } };
  vals_fetch(gnv, R.kg0);
  // Loads and stores share ONE in-order counter, and a wait for the prefetched values is written as "at most N younger
  // operations outstanding".  At the loop's head the compiler has to be right for both ways in: over the back edge the
  // prefetch is followed by the next records' load and (TS) the iteration's four transient stores, from here by nothing -- it
  // took the smaller N, and every iteration's waits drained that iteration's stores (s_waitcnt vmcnt(3) .. (0) in front of
  // the copies of the prefetched rows: profiles/r06_experiments.md section 21).  So the way in from here issues the same
  // operations behind the prefetch as an iteration does: the records' load, and four stores into the words behind the panels.
  gnn = kg_fetch(R.kg0 + ASM_U);
  __builtin_amdgcn_sched_barrier(0);       // (behind the loads above, as in an iteration)
  if(HAS_T && XT && TS)
  {
#pragma unroll
    for(int u = 0; u < ASM_U; u++)
      *reinterpret_cast<double*>(reinterpret_cast<char*>(Lx) + ((trash_off + (uint32_t)((lane + u) & 7)) << 3)) = 0.0;
    asm volatile("" ::: "memory");
  }
#else
  int gnn_unused = 0; (void)gnn_unused;
#endif
  double* myrow = tile + kq*LEN;
#pragma unroll
  for(int u = 0; u < ASM_U; u++) myrow[u*4*LEN + ZC] = 0.0;
  int kg = R.kg0;                       // (a run has at least one k-group: no test in front of the first iteration, the prefetch above stays above)
#ifdef DLG_ASM_PROFILE
  unsigned long long pf_head = 0, pf_rest = 0, pf_it = 0, pf_t0 = clock64();
#endif
  do
  {
#ifdef DLG_ASM_PROFILE
    pf_t0 = clock64();
#endif
    const int gv = gnv;
    uint32_t meta[ASM_U];
    int td[ASM_U];
#pragma unroll
    for(int u = 0; u < ASM_U; u++) meta[u] = (ASM_RUN_ALIGNED || kg + u <= kglast) ? (uint32_t)__builtin_amdgcn_readlane(gv, KD*u + 5) : 0u;
    double xv[ASM_U];
    if(JTX && !x_in_tile)
    {
#pragma unroll
      for(int u = 0; u < ASM_U; u++) xv[u] = xvec[__builtin_amdgcn_ds_bpermute(4*(KD*u + 6 + kq), gv)];    // (rows past the end: row 0, times zeros)
    }
    // (a) one coalesced copy of the rows' windows into the tile (absent rows: zeros)
#ifdef DLG_ASM_PREFETCH
    {
      double v[ASM_U]; int b[ASM_U];
#pragma unroll
      for(int u = 0; u < ASM_U; u++) { v[u] = vpre[u]; b[u] = bpre[u]; }
      vals_fetch(gnn, kg + ASM_U);                    // the next iteration's, on their way during this one
#pragma unroll
      for(int u = 0; u < ASM_U; u++) myrow[u*4*LEN + m] = b[u] >= 0 ? v[u] : 0.0;
    }
    for(int c0 = 16; c0 < ncopy; c0 += 16)
#else
    for(int c0 = 0; c0 < ncopy; c0 += 16)
#endif
    {
      double v[ASM_U];
      int b[ASM_U];
#pragma unroll
      for(int u = 0; u < ASM_U; u++)
      {
        b[u] = __builtin_amdgcn_ds_bpermute(4*(KD*u + kq), gv);
        if(!ASM_RUN_ALIGNED && kg + u > kglast) b[u] = -1;
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
        // (at most two destinations: they came with the record)
        const int nent = (int)((meta[u] >> 8) & 7)*nT;
        // (XT: the host found every k-group's destinations in its record -- no load of asm_tdest at all, so that
        // nothing between the products and the stores waits on the memory counter: a wait there drains the value
        // prefetch of the next iteration, the loads complete in order)
        if(XT) td[u] = __builtin_amdgcn_ds_bpermute(4*(KD*u + 10 + min(lane, 1)), gv);
        else td[u] = tdest[lane < nent ? __builtin_amdgcn_readlane(gv, KD*u + 4) + lane : 0];
      }
    }
#ifdef DLG_ASM_PREFETCH
    gnv = gnn; gnn = kg_fetch(kg + 2*ASM_U);
#ifdef DLG_ASM_PROFILE
    __builtin_amdgcn_wave_barrier();
    { const unsigned long long t1 = clock64(); pf_head += t1 - pf_t0; pf_t0 = t1; }
#endif
#else
    gnv = kg_fetch(kg + ASM_U);
#endif
    __builtin_amdgcn_wave_barrier();
    // (b) operands from the tile, products
#pragma unroll
    for(int u = 0; u < ASM_U; u++)
    {
      const double* row = myrow + u*4*LEN;
      const double bP = row[bcolP];
      accP = __builtin_amdgcn_mfma_f64_16x16x4f64(row[pc], bP, accP, 0, 0, 0);
      if(JTX) jacc += (x_in_tile ? row[15] : xv[u])*bP;
      if(HAS_T && XT)
      {
        // (XT: every k-group with row-block slots closes them -- the host checked, asm_td_inline --: the transient
        // product starts from zero and is stored at once; a k-group past the end has no slot, so nothing is `mine`.
        // The four row offsets are fetched together, in front of the product: one LDS round trip, not four behind it;
        // the panels end below 4 GB (checked too): 32-bit offsets from the scalar base)
        const int myslot = (meta[u] >> (2*kq)) & 3;
        const bool mine = bs < (int)((meta[u] >> 8) & 7);
        const dlg_v4d t4 = __builtin_amdgcn_mfma_f64_16x16x4f64(row[tc], row[bs == myslot ? bcol : ZC], (dlg_v4d){0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
        if(ts_ok)
        {
          const int ro1 = __builtin_amdgcn_ds_bpermute(ts_td4, td[u]);
#pragma unroll
          for(int r = 0; r < 4; r++) tb_wr[4*r*ASM_TLD] = t4[r];
          // (no barrier of the wave's own: the LDS takes a wave's accesses in order, and the compiler keeps a load behind
          // stores it may alias -- the products of the next k-group need not wait for this round trip)
          const double tv = tbuf[ts_rd];
          // (EVERY lane stores, the ones without an entry into the words behind the panels: a store under a branch is one
          // the compiler cannot count, and the waits for the prefetched values -- one counter, in order, stores included --
          // then assume the fewest and wait for stores they need not wait for)
          const uint32_t off = (ts_bs < (int)((meta[u] >> 8) & 7)) ? colTs + (uint32_t)ro1 : trash_off + (uint32_t)(lane & 7);
          *reinterpret_cast<double*>(reinterpret_cast<char*>(Lx) + (off << 3)) = tv;
        }
        else
        {
          int ro[4];
#pragma unroll
          for(int r = 0; r < 4; r++) ro[r] = __builtin_amdgcn_ds_bpermute(4*(bs*nT + TJ(r)), td[u]);
#pragma unroll
          for(int r = 0; r < 4; r++)
          {
#ifndef DLG_ASM_NO_TSTORE                     // (tools/variant_lib.sh: the kernel without its transient stores)
            if(mine && TJ(r) != 0xFF)
#else
            if(mine && TJ(r) != 0xFF && t4[r] == 1.2345e300)
#endif
              *reinterpret_cast<double*>(reinterpret_cast<char*>(Lx) + (((uint32_t)colT + (uint32_t)(ro[r] + TA(r))) << 3)) = t4[r];
          }
        }
      }
      else if(HAS_T)
      {
        // (the general form: a row-block of more than four rows spans two k-groups, the first of which stores nothing)
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
        if(te_ok)
        {
          double* const tbj = tbuf + 16*ASM_TLD;               // [4][16] the rows' Jt*x sums
#pragma unroll
          for(int r = 0; r < 4; r++) tb_wr[4*r*ASM_TLD] = accP[r];
          tbj[16*kq + m] = jacc;
          for(int rd = 0; rd < (te_two ? 2 : 1); rd++)
          {
            const uint32_t te_w = rd == 0 ? te_w0 : pent[64 + lane];
            const int te_kind = te_w & 3, te_n = (te_w >> 6) & 15;
            const int ro = __builtin_amdgcn_ds_bpermute((int)((te_w >> 10) & 15) << 2, pdc);
            double val = tbuf[(te_kind == 3 ? 0 : (int)((te_w >> 2) & 15)*ASM_TLD) + (te_kind == 3 ? 0 : te_n)];
            const int te_c = te_kind == 3 ? te_n : 0;
            // the four rows of the k-groups: (0 + 1) + (2 + 3), as the shuffles added them
            const double tj = (tbj[te_c] + tbj[16 + te_c]) + (tbj[32 + te_c] + tbj[48 + te_c]);
            const int pa = (int)((te_w >> 14) & 15);
            double* dst = Lx + (size_t)trash_off + (lane & 7);
            if(te_kind == 1)
            {
              if(Tpart < 0) { if(!(te_w >> 30 & 1)) dst = Lx + (Tpanel + (int64_t)te_n*Tld + (ro + pa)); }
              else dst = part + (Tpart + ((int)((te_w >> 18) & 255) + te_n*(int)((te_w >> 26) & 15)));
            }
            else if(te_kind == 2) dst = part + (Trpart + ((te_n - nJ)*nJr + pa));
            else if(te_kind == 3)
            {
              val = tj;
              dst = (Tjvar >= 0 && te_n < nJ) ? jtx_out + (Tjvar + te_n) : jtp + (16*(int64_t)tix + te_n);
            }
            *dst = val;
          }
        }
        else
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
        if(JTX)
        {
          // the four rows of the k-groups sit in the four lane groups: (0 + 1) + (2 + 3)
          double t = jacc + __shfl_xor(jacc, 16);
          t += __shfl_xor(t, 32);
          if(kq == 0)
          {
            if(Tjvar >= 0 && m < nJ) jtx_out[Tjvar + m] = t;     // the only task of its block: (Jt x)[J] is complete
            else jtp[16*(int64_t)tix + m] = t;                   // (a rider on board still leaves its part of the record)
          }
        }
        }
        accP = (dlg_v4d){0.0, 0.0, 0.0, 0.0};
        jacc = 0.0;
        tix++;
        task_unpack(tnv); pdc = pdn;
        tnv = task_fetch(tix + 1);
        pdn = pdest[16*(int64_t)min(tix + 1, tlast) + m];
      }
    }
    __builtin_amdgcn_wave_barrier();
#ifdef DLG_ASM_PROFILE
    { const unsigned long long t1 = clock64(); pf_rest += t1 - pf_t0; pf_it++; }
#endif
    kg += ASM_U;
  }
  while(kg < R.kg1);
#ifdef DLG_ASM_PROFILE
  if(lane == 0)
  {
    const int c = HAS_T ? 1 : 0;
    atomicAdd(&g_asm_prof[c][0], pf_head); atomicAdd(&g_asm_prof[c][1], pf_rest); atomicAdd(&g_asm_prof[c][2], pf_it); atomicAdd(&g_asm_prof[c][3], 1ull);
  }
#endif
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
template <int CLEN, bool JTX, bool XT = false, bool TS = false>
__global__ void __launch_bounds__(TPB) ASM_WPE_ATTR k_assemble_mfma(const AsmRun* __restrict__ runs, int nruns,
                                                       const AsmMTask* __restrict__ tasks,
                                                       const AsmKG* __restrict__ kgs,
                                                       const AsmShape* __restrict__ shapes,
                                                       const int* __restrict__ tdest, const int* __restrict__ pdest,
                                                       const double* __restrict__ vals,
                                                       double* __restrict__ Lx, double* __restrict__ part, int LEN,
                                                       const double* __restrict__ xvec, double* __restrict__ jtp,
                                                       double* __restrict__ jtx_out, int only_shape, uint32_t trash_off,
                                                       const uint32_t* __restrict__ pent)
{
  extern __shared__ double asm_tiles[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x*(TPB/64) + (threadIdx.x >> 6));
  if(wid >= nruns) return;
  const AsmRun R = runs[wid];
  if(only_shape >= 0 && tasks[R.task0].shape != only_shape) return;      // (tools/k4_split.py: the time of one kind of task)
  const AsmShape* SH = shapes + tasks[R.task0].shape;
  double* tile = asm_tiles + (threadIdx.x >> 6)*(ASM_U*4*LEN);
  // (XT: the launch sized the LDS for a 16-row square a wave behind the tiles)
  double* tbuf = TS ? asm_tiles + (TPB/64)*(ASM_U*4*LEN) + (threadIdx.x >> 6)*ASM_TBUF : nullptr;
  const uint32_t* pe = pent + 128*tasks[R.task0].shape;
  if(SH->MT > 0) asm_mfma_run<true, CLEN, JTX, XT, TS>(R, tasks, SH, kgs, tdest, pdest, vals, Lx, part,  lane, tile, LEN, xvec, jtp, jtx_out, tbuf, trash_off, pe);
  else           asm_mfma_run<false, CLEN, JTX, XT, TS>(R, tasks, SH, kgs, tdest, pdest, vals, Lx, part, lane, tile, LEN, xvec, jtp, jtx_out, tbuf, trash_off, pe);
}
// Jt*x from the records the assembly kernel left (JTX): var-block v = blks[...] sums its list in order.
// short lists: 16 threads per var-block (thread = entry of the block); long ones (a dense block that
// every task carries as a rider): a workgroup per var-block, 64 strided sub-sums, then those in order.
__global__ void __launch_bounds__(TPB) k_jtx_fin2_short(const int* __restrict__ blks, int nblk,
                                                        const int* __restrict__ jf_ptr, const int* __restrict__ jf_ent,
                                                        const int* __restrict__ var0, const int* __restrict__ wv,
                                                        const double* __restrict__ jtp, double* __restrict__ Jt_x)
{
  const int g = blockIdx.x*(TPB/16) + (threadIdx.x >> 4), a = threadIdx.x & 15;
  if(g >= nblk) return;
  const int v = blks[g];
  if(a >= wv[v]) return;
  double sum = 0.0;
  for(int e = jf_ptr[v]; e < jf_ptr[v+1]; e++) sum += jtp[jf_ent[e] + a];
  Jt_x[var0[v] + a] = sum;
}
struct JfAug { double* Lx; const int64_t* augpos; const int* perm; const int64_t* aug_of_var; const char* listed; int n; int* info;
               int* flag; int flag_epoch; int nlist_blocks; };
__global__ void __launch_bounds__(1024) k_jtx_fin2_long(const int* __restrict__ blks,
                                                        const int* __restrict__ jf_ptr, const int* __restrict__ jf_ent,
                                                        const int* __restrict__ var0, const int* __restrict__ wv,
                                                        const double* __restrict__ jtp, double* __restrict__ Jt_x,
                                                        double* __restrict__ segpart, int* __restrict__ segcnt, JfAug aug)
{
  __shared__ double sh[128*16];
  __shared__ int s_last;
  // (aug.Lx: the launch also sets the augmented row of every panel -- the right-hand side Jt*x of the point just
  // evaluated -- and re-arms the pivot flag: the blocks behind the lists' store the columns whose Jt*x the assembly kernel
  // wrote itself, the workgroup that finishes the one list the columns of its block; it then raises the word that tells
  // the second stream that Jt*x is final.  One kernel less between the assembly and the leaf level.)
  if((int)blockIdx.x >= aug.nlist_blocks)
  {
    const int k = ((int)blockIdx.x - aug.nlist_blocks)*1024 + (int)threadIdx.x;
    if(k == 0) *aug.info = 0x7fffffff;
    if(k < aug.n) { const int var = aug.perm[k]; if(!aug.listed[var]) aug.Lx[aug.augpos[k]] = Jt_x[var]; }
    return;
  }
  // (blks: flat records {list begin, list end, first variable, width} -- sparse_set_pattern)
  // JFL_SEG workgroups per list, each sums a contiguous segment (one workgroup read the 60 000 records of
  // config #4's dense block at 0.6 TB/s: 11 us on the path of every evaluation); the last one to arrive
  // adds the segment sums in segment order -- the order of the sums does not depend on who that is.
  const int blk = blockIdx.x / JFL_SEG, seg = blockIdx.x % JFL_SEG;
  const int4 rec = reinterpret_cast<const int4*>(blks)[blk];
  const int w = rec.w;
  // groups of 8 lanes for blocks of up to 8 variables (128 sub-sums), of 16 otherwise (64)
  const int gw = (w <= 8) ? 8 : 16, ng = 1024/gw;
  const int a = threadIdx.x & (gw - 1), g = threadIdx.x/gw;
  const int len = rec.y - rec.x;
  const int e0 = rec.x + (int)((long)len*seg/JFL_SEG), e1 = rec.x + (int)((long)len*(seg + 1)/JFL_SEG);
  double sum = 0.0;
  if(a < w)
    for(int e = e0 + g; e < e1; e += ng*16)
    {
      // sixteen records in flight (their indices first), added in list order
      int ix[16]; double t[16];
#pragma unroll
      for(int u = 0; u < 16; u++) ix[u] = jf_ent[min(e + ng*u, e1 - 1)];
#pragma unroll
      for(int u = 0; u < 16; u++) t[u] = jtp[ix[u] + a];
#pragma unroll
      for(int u = 0; u < 16; u++) sum += (e + ng*u < e1) ? t[u] : 0.0;
    }
  sh[g*16 + a] = sum;
  __syncthreads();
  typedef __attribute__((address_space(1))) double* gdp_t;
  double* mine = segpart + ((size_t)blk*JFL_SEG + seg)*16;
  if(g == 0 && a < w)
  {
    double tot = 0.0;
    for(int k = 0; k < ng; k++) tot += sh[k*16 + a];
    // the hand-off of MI355X_MICROARCH.md: write-through payload, drained, then the counter
    __hip_atomic_store((gdp_t)(mine + a), tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if(threadIdx.x == 0) s_last = (atomicAdd(segcnt + blk, 1) == JFL_SEG - 1);
  __syncthreads();
  if(!s_last) return;
  if(threadIdx.x < w)
  {
    double tot = 0.0;
    for(int k = 0; k < JFL_SEG; k++)
      tot += __hip_atomic_load((gdp_t)(segpart + ((size_t)blk*JFL_SEG + k)*16 + threadIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Jt_x[rec.z + threadIdx.x] = tot;
    if(aug.Lx) aug.Lx[aug.aug_of_var[rec.z + threadIdx.x]] = tot;
  }
  if(threadIdx.x == 0) __hip_atomic_store(segcnt + blk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // for the next launch (stream order)
  if(aug.flag)
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if(threadIdx.x == 0) __hip_atomic_store(aug.flag, aug.flag_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
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
// (the blocks past the fins' own, if any: the augmented row of the point just evaluated, k_set_aug_row's
// work riding in this launch -- it touches the last row of the panels only, the fins never do)
struct AugRowArgs { const int64_t* augpos; const int* perm; const double* rhs; int n; int* info; };
__global__ void __launch_bounds__(TPB) k_assemble_fin2_short(const AsmFin2* __restrict__ fins, int nfins,
                                                             const int64_t* __restrict__ list,
                                                             double* part, double* __restrict__ Lx, AugRowArgs aug)
{
  const int nfb = (nfins + TPB/64 - 1)/(TPB/64);
  if((int)blockIdx.x >= nfb)
  {
    const int k = (blockIdx.x - nfb)*TPB + threadIdx.x;
    if(k == 0) *aug.info = 0x7fffffff;                  // re-arm the pivot flag of the factorisation that follows
    if(k < aug.n) Lx[aug.augpos[k]] = aug.rhs[aug.perm[k]];       // (a store: see k_set_aug_row; this launch carries it on a single rank only)
    return;
  }
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
// `phase` selects the columns by the owner of their supernode (subtree partition): -1 all of them,
// 0 the columns of supernodes below the cut, 1 the columns of the replicated supernodes above it
__device__ __forceinline__ bool phase_has(const int* __restrict__ sn_owner, int s, int phase)
{ return phase < 0 || (phase == 0) == (sn_owner[s] >= 0); }
__global__ void __launch_bounds__(TPB) k_add_lambda(double* __restrict__ Lx,
                                                    const int64_t* __restrict__ diagpos, int n,
                                                    double lambda, const int* __restrict__ col_sn,
                                                    const int* __restrict__ sn_owner, int phase)
{
  const int i = blockIdx.x*TPB + threadIdx.x;
  if(i < n && phase_has(sn_owner, col_sn[i], phase)) Lx[diagpos[i]] += lambda;
}

// augmented row: panel(last row, column k) += rhs[perm[k]]   (the row is zero after the assembly;
// above the cut of a subtree partition it already carries the updates from below).  augpos[k] = the
// entry's offset in Lx (precomputed: one dependent load instead of five)
__global__ void __launch_bounds__(TPB) k_set_aug_row(double* __restrict__ Lx, const int* __restrict__ col_sn,
                                                     const int64_t* __restrict__ augpos,
                                                     const int* __restrict__ perm,
                                                     const double* __restrict__ rhs, int n,
                                                     int* __restrict__ info,
                                                     const int* __restrict__ sn_owner, int phase)
{
  const int k = blockIdx.x*TPB + threadIdx.x;
  if(k == 0 && phase <= 0) *info = 0x7fffffff;          // re-arm the pivot flag of the factorisation that follows
  if(k >= n) return;
  if(phase >= 0 && !phase_has(sn_owner, col_sn[k], phase)) return;
  // (single rank: nothing but this kernel ever writes a last row between two assemblies, and a partially cleared
  // buffer -- clear_panels -- holds the previous factorisation's there; the phases of a partition add behind the sum over the ranks)
  if(phase < 0) Lx[augpos[k]] = rhs[perm[k]]; else Lx[augpos[k]] += rhs[perm[k]];
}

// the same for a freshly cleared buffer whose last rows nothing has touched (fin on the side: this kernel sits on the
// critical stream between the Jt*x sums and the leaf level): plain stores, two columns a thread, their loads in flight
// together -- two dependent round trips instead of three
__global__ void __launch_bounds__(TPB) k_store_aug_row(double* __restrict__ Lx, const int64_t* __restrict__ augpos,
                                                       const int* __restrict__ perm, const double* __restrict__ rhs, int n,
                                                       int* __restrict__ info, int* flag, int flag_epoch)
{
  const int k = blockIdx.x*TPB + threadIdx.x, k2 = k + gridDim.x*TPB;
  if(k == 0) { __hip_atomic_store(flag, flag_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); *info = 0x7fffffff; }
  const int p0 = perm[min(k, n - 1)], p1 = perm[min(k2, n - 1)];
  const int64_t a0 = augpos[min(k, n - 1)], a1 = augpos[min(k2, n - 1)];
  const double v0 = rhs[p0], v1 = rhs[p1];
  if(k < n) Lx[a0] = v0;
  if(k2 < n) Lx[a1] = v1;
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
  SymTask T = {0, 0, 0, -1, 0, 1};
  const bool live = wid < ntasks;
  if(live)
  {
    T = tasks[wid];
    nI = T.nI; J = 64/nI; a = lane % nI; j = lane / nI;
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
    if(T.part < 0) jtx[T.var0 + lane] = s;
    else part[(size_t)T.part*8 + lane] = s;
  }
}
// one workgroup per multi-chunk var-block: NT/8 groups x 8 scalars, four loads in flight per
// thread; 256 threads for the usual lists, 1024 for the few long ones (a dense block that every
// row touches has thousands of partials)
template <int NT>
__global__ void __launch_bounds__(NT) k_jtx_fin(const int* __restrict__ flist,
                                                const int* __restrict__ fin_ptr,
                                                const int* __restrict__ fin_blk,
                                                const SymOutBlock* __restrict__ oblk,
                                                const double* __restrict__ part,
                                                double* __restrict__ jtx)
{
  constexpr int G = NT/8;
  __shared__ double sh[NT];
  const int f = flist[blockIdx.x];
  const int a = threadIdx.x & 7, g = threadIdx.x >> 3;
  const SymOutBlock B = oblk[fin_blk[f]];
  const int p1 = fin_ptr[f+1];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int p = fin_ptr[f] + g;
  for(; p + 3*G < p1; p += 4*G)
  {
    s0 += part[(size_t)p*8 + a]; s1 += part[(size_t)(p + G)*8 + a];
    s2 += part[(size_t)(p + 2*G)*8 + a]; s3 += part[(size_t)(p + 3*G)*8 + a];
  }
  for(; p < p1; p += G) s0 += part[(size_t)p*8 + a];
  sh[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if(g == 0 && a < B.nI)
  {
    double tot = 0.0;
    for(int k = 0; k < G; k++) tot += sh[k*8 + a];
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

__global__ void __launch_bounds__(TPB) k_norm2_Jv(const int* __restrict__ chunk_row,
                                                  const int* __restrict__ Jp,
                                                  const int* __restrict__ Ji,
                                                  const double* __restrict__ vals,
                                                  const double* __restrict__ v,
                                                  double* __restrict__ part, const int* __restrict__ info,
                                                  const double* __restrict__ kind, int nnz,
                                                  const double* __restrict__ dsc, double* __restrict__ hsc, int nsc,
                                                  const double* __restrict__ psrc, double* __restrict__ pdst, int pn,
                                                  const double* __restrict__ skipf)
{
  __shared__ __attribute__((aligned(16))) double prod[NV_CHUNK + 4];
  __shared__ double sh[4];
  // the last kernel of dlg_take_step: the device scalars of the step (written by the kernels before this
  // one) go to the page-locked host array with it -- no copy kernel behind it
  if(dsc && blockIdx.x == 0 && (int)threadIdx.x < nsc) hsc[threadIdx.x] = dsc[threadIdx.x];
  // ... and p_new, a slice per workgroup, to its page-locked destination (dlg_take_step)
  if(psrc)
  {
    const int per = (pn + (int)gridDim.x - 1)/(int)gridDim.x, i0 = (int)blockIdx.x*per, i1 = min(i0 + per, pn);
    for(int i = i0 + (int)threadIdx.x; i < i1; i += TPB) pdst[i] = psrc[i];
  }
  // K8 behind a speculative factorisation (dlg_take_step): a step built on a failed factorisation is
  // never used -- unless it is the Cauchy step to the edge of the trust region, which needs no factor
  if(info && *info != 0x7fffffff && (int)*kind != DLG_KIND_CAUCHY_TO_EDGE) { if(threadIdx.x == 0) part[blockIdx.x] = 0.0; return; }
  // ... and K8 of a step whose |J step|^2 the host takes from the solved system (the step kernel's word, k_part_take_step)
  if(skipf && *skipf != 0.0) return;
  const int4 rec = reinterpret_cast<const int4*>(chunk_row)[blockIdx.x];      // {r0, r1, q0, n}
  const int r0 = rec.x, r1 = rec.y, q0 = rec.z, n = rec.w;
  const int tid = threadIdx.x;
  double acc = 0.0;
  if(n <= NV_CHUNK)
  {
    // the run from the 4-element boundary below its start, four non-zeros per thread and load: one 16-byte
    // load of indices, two of values (the up to three elements in front belong to the row before: their
    // products are formed and never summed; past the end of the arrays the last thread goes one by one)
    const int pad = q0 & 3, qa = q0 - pad, nn = n + pad;
    for(int base = 0; base < nn; base += 8*TPB)
    {
      int4 ix[2]; double2 va[2], vb[2];
#pragma unroll
      for(int u = 0; u < 2; u++)
      {
        const int e = base + 4*(tid + u*TPB);
        const bool whole = e < nn && qa + e + 4 <= nnz;
        ix[u] = whole ? *reinterpret_cast<const int4*>(Ji + qa + e) : make_int4(0, 0, 0, 0);
        va[u] = whole ? *reinterpret_cast<const double2*>(vals + qa + e) : make_double2(0.0, 0.0);
        vb[u] = whole ? *reinterpret_cast<const double2*>(vals + qa + e + 2) : make_double2(0.0, 0.0);
      }
#pragma unroll
      for(int u = 0; u < 2; u++)
      {
        const int e = base + 4*(tid + u*TPB);
        if(e < nn)
        {
          if(qa + e + 4 <= nnz)
          {
            const double p0 = va[u].x*v[ix[u].x], p1 = va[u].y*v[ix[u].y], p2 = vb[u].x*v[ix[u].z], p3 = vb[u].y*v[ix[u].w];
            *reinterpret_cast<double2*>(prod + e) = make_double2(p0, p1);
            *reinterpret_cast<double2*>(prod + e + 2) = make_double2(p2, p3);
          }
          else
            for(int k = 0; k < 4; k++) if(qa + e + k < nnz) prod[e + k] = vals[qa + e + k]*v[Ji[qa + e + k]];
        }
      }
    }
    __syncthreads();
    for(int r = r0 + tid; r < r1; r += TPB)
    {
      const int a = Jp[r] - qa, bq = Jp[r+1] - qa;
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

} // namespace

// K1
int sparse_eval(dlg_backend* b, int s)
{
  SparseSym* Y = b->sym;
  if(!Y) { dlg_set_error("dlg_sparse_set_pattern must be called first"); return DLG_ERR_STATE; }
  DlgSlot& S = b->slot[s];
  const SymHost& H = Y->H;
  if(!H.jtx_covers_all) DLG_HIP(hipMemsetAsync(S.Jt_x, 0, sizeof(double)*(size_t)b->N, b->stream));   // var-blocks without rows
  const int nt = (int)H.jtx_task.size();
  if(nt > 0)
    hipLaunchKernelGGL(k_jtx, dim3(dlg_cdiv(nt, TPB/64)), dim3(TPB), 0, b->stream, Y->jtx_task, nt,
                       Y->oblk, Y->contrib, S.Jin(), S.xin(), S.Jt_x, Y->jtx_part);
  if(Y->n_fin_short > 0)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_jtx_fin<256>), dim3(Y->n_fin_short), dim3(256), 0, b->stream, Y->jtx_fin_short,
                       Y->jtx_fin_ptr, Y->jtx_fin_blk, Y->oblk, Y->jtx_part, S.Jt_x);
  if(Y->n_fin_long > 0)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_jtx_fin<1024>), dim3(Y->n_fin_long), dim3(1024), 0, b->stream, Y->jtx_fin_long,
                       Y->jtx_fin_ptr, Y->jtx_fin_blk, Y->oblk, Y->jtx_part, S.Jt_x);
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}

// K3 / K8
int sparse_norm2_Jv(dlg_backend* b, int s, const double* v, double* out_dev, const double* kind_if_factor_failed)
{
  SparseSym* Y = b->sym;
  if(!Y) { dlg_set_error("dlg_sparse_set_pattern must be called first"); return DLG_ERR_STATE; }
  DlgSlot& S = b->slot[s];
  const int g = Y->n_nv_chunks;
  if(g == 0) { DLG_HIP(hipMemsetAsync(out_dev, 0, sizeof(double), b->stream)); return DLG_OK; }
  // K8 inside dlg_take_step: only the host reads the sum -- the workgroups' partial sums go straight to
  // pinned host memory and are added there behind the step's one synchronisation (no second-stage launch)
  // (tail_mode without a kind: K8 of a step from cached vectors, dlg_step -- no factorisation of this call to look at)
  if(kind_if_factor_failed || b->tail_mode)
    if(double* hp = b->tail_mode ? dlg_tail_partials(b, g) : dlg_host_partials(b, out_dev, g, 1, 0, 1))
    {
      const bool fold = b->fold_scal > 0 && b->fold_scal <= TPB && b->h_scal;
      if(!fold) b->attach_stop = nullptr;
      DLG_LAUNCH_LAST(b, k_norm2_Jv, dim3(g), dim3(TPB), 0, b->stream, Y->nv_chunk, Y->Jp, Y->Ji, S.Jin(), v, hp,
                      kind_if_factor_failed ? (const int*)Y->d_info : (const int*)nullptr, kind_if_factor_failed, (int)Y->nnz_loc,
                      fold ? (const double*)b->d_scal : (const double*)nullptr, b->h_scal, (int)b->fold_scal,
                      (fold || b->tail_mode) ? b->fold_p_src : (const double*)nullptr, b->fold_p_dst, (int)b->N, b->k8_skip);
      DLG_LAUNCH_CHECK();
      if(fold) b->scal_copied = true;
      if(fold || b->tail_mode) b->p_copied = b->fold_p_src != nullptr;
      return DLG_OK;
    }
  DLG_CHECK(dlg_ensure_partials(b, 5120 + (size_t)g));
  double* part = b->d_part + 5120;          // behind the regions of the vector reductions (kernels_vec.hip)
  hipLaunchKernelGGL(k_norm2_Jv, dim3(g), dim3(TPB), 0, b->stream, Y->nv_chunk, Y->Jp, Y->Ji, S.Jin(), v, part,
                     kind_if_factor_failed ? (const int*)Y->d_info : (const int*)nullptr, kind_if_factor_failed, (int)Y->nnz_loc,
                     (const double*)nullptr, (double*)nullptr, 0, (const double*)nullptr, (double*)nullptr, 0, (const double*)nullptr);
  DLG_LAUNCH_CHECK();
  return k_reduce_sum(b, part, g, out_dev);
}

// the assembly launches: JtJ of the local rows (values Jv) into the zeroed panel buffer
static int assemble_fin_launch(dlg_backend* b, double* Lx, const double* aug_rhs = nullptr, bool* aug_done = nullptr);
// the partial-sum stages of an assembly whose caller wanted Jt*x first (assemble_launch, defer_fin)
int sparse_assemble_finish(dlg_backend* b)
{
  SparseSym* Y = b->sym;
  if(!Y || !Y->fin_pending_Lx) return DLG_OK;
  double* Lx = Y->fin_pending_Lx;
  Y->fin_pending_Lx = nullptr;
  // the right-hand side of the Gauss-Newton system (Jt*x of the point just evaluated) goes into the
  // augmented row now, and the pivot flag is re-armed: nothing left to launch between the caller's
  // decision to factorise and the first factor kernel.  It rides in the first partial-sum launch.
  bool aug_done = false;
  DLG_CHECK(assemble_fin_launch(b, Lx, Y->fin_pending_rhs, &aug_done));
  if(Y->fin_pending_rhs)
  {
    const SymHost& H = Y->H;
    if(!aug_done)
      hipLaunchKernelGGL(k_set_aug_row, dim3(dlg_cdiv(H.N, TPB)), dim3(TPB), 0, b->stream, Lx, Y->col_sn, Y->augpos,
                         Y->perm, Y->fin_pending_rhs, H.N, Y->d_info, Y->sn_owner, -1);
    DLG_LAUNCH_CHECK();
    Y->spec_aug_rhs = Y->fin_pending_rhs; Y->info_clean = true;
    Y->fin_pending_rhs = nullptr;
  }
  return DLG_OK;
}
// ---- fin on the side ------------------------------------------------------------------------------------------
// An evaluation whose point is factorised at once (backend.hip, step_prepare) had, between the assembly kernel and the
// leaf level of the factorisation, on ONE stream: the Jt*x record sums, the norm kernel the host waits for, a gap behind
// it (a kernel somebody listens to holds the next dispatch back), and the two partial-sum stages of JtJ -- 35 us of
// small kernels of which the leaf level needs only Jt*x (for the augmented row).  The stages write blocks of the
// ANCESTORS' panels only (checked when the schedules are set up: every block a stage stores lies in a supernode above
// level 0), so they and the norm kernel go to the second stream:
//   main stream:    ... Jt*x sums | augmented row (+ flag A) | leaf level | [gate: flag B] update gather | region ...
//   second stream:  gate: flag A | norms (the host's event rides on them) | stage 1 | stage 2 | flag B
// The gates are one-wave kernels polling a word (no event packets on the critical stream: an event between two
// kernels costs 5 - 8 us there); flag B is long up when the main stream asks.  Whoever touches the ancestors' panels
// or the partial-sum buffers next on the main stream asks first (sparse_fin_side_gate: lambda on the diagonal, the
// first update kernel, another assembly).  Same kernels, same sums, same bits.
__global__ void k_fin_flag(int* flag, int epoch)
{
  if(threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
bool sparse_fin_side_ok(const dlg_backend* b)
{
  const SparseSym* Y = b->sym;
  return Y && Y->fin_side_sched_ok && Y->fin_flag && b->aux_stream && Y->fin_pending_Lx && (Y->fin_pending_rhs || Y->aug_fused_epoch) &&
         !Y->fin_main && b->stream != b->aux_stream;
}
int sparse_fin_side_begin(dlg_backend* b)
{
  SparseSym* Y = b->sym;
  const SymHost& H = Y->H;
  DLG_CHECK(sparse_fin_side_gate(b));                  // (stages of an evaluation nobody factorised: done with the buffers first)
  if(Y->aug_fused_epoch)
  {
    // the Jt*x sums set the augmented rows and raise the word themselves (assemble_launch): nothing to launch here
    const int ep = Y->aug_fused_epoch;
    Y->aug_fused_epoch = 0;
    DLG_CHECK(dlg_gate_wait(b, b->aux_stream, Y->fin_flag, ep, true));
    Y->fin_main = b->stream;
    b->stream = b->aux_stream;
    return DLG_OK;
  }
  const int ep = ++Y->fin_epoch;
  // main stream, behind the Jt*x sums: the augmented row of every panel (the stages never touch a last row), the pivot
  // flag re-armed -- and the word that tells the second stream that Jt*x is final
  hipLaunchKernelGGL(k_store_aug_row, dim3(dlg_cdiv(H.N, 2*TPB)), dim3(TPB), 0, b->stream, Y->fin_pending_Lx, Y->augpos,
                     Y->perm, Y->fin_pending_rhs, H.N, Y->d_info, Y->fin_flag, ep);
  DLG_LAUNCH_CHECK();
  Y->spec_aug_rhs = Y->fin_pending_rhs; Y->info_clean = true;
  Y->fin_pending_rhs = nullptr;                        // (sparse_assemble_finish: the stages only)
  DLG_CHECK(dlg_gate_wait(b, b->aux_stream, Y->fin_flag, ep, true));
  Y->fin_main = b->stream;
  b->stream = b->aux_stream;
  return DLG_OK;
}
int sparse_fin_side_end(dlg_backend* b)
{
  SparseSym* Y = b->sym;
  if(!Y || !Y->fin_main) return DLG_OK;
  hipStream_t side = b->stream;
  b->stream = Y->fin_main; Y->fin_main = nullptr;
  hipLaunchKernelGGL(k_fin_flag, dim3(1), dim3(64), 0, side, Y->fin_flag + 1, Y->fin_epoch);
  Y->fin_side_owed = Y->fin_epoch;
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}
int sparse_fin_side_gate(dlg_backend* b)
{
  SparseSym* Y = b->sym;
  if(!Y || !Y->fin_side_owed) return DLG_OK;
  const int ep = Y->fin_side_owed;
  Y->fin_side_owed = 0;
  return dlg_gate_wait(b, Y->fin_main ? Y->fin_main : b->stream, Y->fin_flag + 1, ep, true);
}
// ---- clearing a panel buffer in front of an assembly: all of it, or (sparse_host.hip, clr_partial_ok) only what is
// not a merged leaf's panel once the buffer's leaves are known to be clean
__global__ void __launch_bounds__(TPB) k_clear_ranges(double* __restrict__ Lx, const int64_t* __restrict__ off,
                                                      const int64_t* __restrict__ len)
{
  double* d = Lx + off[blockIdx.y];
  const int64_t n = len[blockIdx.y];
  // (16-byte stores where the range allows: the head up to an even index one by one)
  const int64_t head = ((reinterpret_cast<uintptr_t>(d) >> 3) & 1) ? 1 : 0;
  if(blockIdx.x == 0 && threadIdx.x == 0 && head && n > 0) d[0] = 0.0;
  double2* d2 = reinterpret_cast<double2*>(d + head);
  const int64_t n2 = (n - head) >> 1;
  for(int64_t i = blockIdx.x*(int64_t)TPB + threadIdx.x; i < n2; i += (int64_t)gridDim.x*TPB) d2[i] = make_double2(0.0, 0.0);
  if(blockIdx.x == 0 && threadIdx.x == 0 && ((n - head) & 1)) d[n - 1] = 0.0;
}
void sparse_mark_unclean(dlg_backend* b) { if(b->sym) b->sym->lz_ok[0] = b->sym->lz_ok[1] = nullptr; }
static int clear_panels(dlg_backend* b, double* Lx, hipStream_t st)
{
  SparseSym* Y = b->sym;
  const SymHost& H = Y->H;
  const bool known = Lx == Y->lz_ok[0] || Lx == Y->lz_ok[1];
  if(known && Y->clr_partial_ok && !b->sharded() && H.part_nranks <= 1)
  {
    hipLaunchKernelGGL(k_clear_ranges, dim3(64, Y->n_clr), dim3(TPB), 0, st, Lx, Y->clr_off, Y->clr_len);
    DLG_LAUNCH_CHECK();
    return DLG_OK;
  }
  DLG_HIP(hipMemsetAsync(Lx, 0, sizeof(double)*((size_t)H.lx_size + 8), st));
  if(!known) { if(!Y->lz_ok[0]) Y->lz_ok[0] = Lx; else if(!Y->lz_ok[1]) Y->lz_ok[1] = Lx; else { Y->lz_ok[0] = Lx; Y->lz_ok[1] = nullptr; } }
  return DLG_OK;
}
#define ASM_LAUNCH(kernel, grid, block, shm, st, ...) \
  do { hipEvent_t e0 = nullptr, e1 = nullptr; \
       if(timed_single && dlg_prof_pair(b, DLG_PROF_K4_KERNEL, &e0, &e1)) hipExtLaunchKernelGGL(kernel, grid, block, shm, st, e0, e1, 0, __VA_ARGS__); \
       else { DlgProfScope pk1(b, DLG_PROF_K4_KERNEL, timed_single && !b->ext_events); hipLaunchKernelGGL(kernel, grid, block, shm, st, __VA_ARGS__); } } while(0)
static int assemble_launch(dlg_backend* b, const double* Jv, double* Lx = nullptr, const double* xvec = nullptr, double* Jt_x = nullptr,
                           bool zeroed = false, bool defer_fin = false)
{
  SparseSym* Y = b->sym;
  const SymHost& H = Y->H;
  hipStream_t st = b->stream;
  DLG_CHECK(sparse_assemble_finish(b));             // (an earlier assembly's partial sums live in the buffers this one fills)
  DLG_CHECK(sparse_fin_side_gate(b));               // (... and stages still running on the second stream read them)
  if(!Lx) Lx = Y->Lx;
  if(!zeroed) DLG_CHECK(clear_panels(b, Lx, st));
  const int nt = (int)H.asm_ctask.size(), nmt = (int)H.asm_mtask.size();
  if(nt > 0 || nmt > 0)
  {
    // (one kernel: its time stamps ride on the launch)
    const bool timed_single = nmt > 0 && nt == 0;
    DlgProfScope pk(b, DLG_PROF_K4_KERNEL, !timed_single);
    if(nmt > 0)
    {
      const int nruns = (int)H.asm_run.size();
      const double* nox = nullptr; double* nojt = nullptr;
      static const int only_shape = getenv("DLG_ASM_ONLY_SHAPE") ? atoi(getenv("DLG_ASM_ONLY_SHAPE")) : -1;    // tools only
      // (x rides in the tile rows where every shape leaves column 15 free)
      bool xt = H.asm_lds_len == 18;
      for(const AsmShape& sh : H.asm_shape) if(sh.ncopy > 15) xt = false;
      if(xt) xt = H.asm_td_inline;                  // ... and every k-group carries its transient destinations
      if(xt) xt = (uint64_t)H.lx_size + (1u << 20) < (1ull << 29);      // ... and the panels end below 4 GB (32-bit store offsets)
      // (TS: every shape's transient entries of a k-group fit one round of 64 lanes -- asm_mfma_run, ts_ok)
      bool ts = xt;
      for(const AsmShape& sh : H.asm_shape)
        if(sh.MT > 0)
        {
          int nrt = 0;
          for(int q = 0; q < 16; q++) if(sh.tj[q] != 0xFF) nrt++;
          const int smax = std::min((int)sh.smax, std::max(1, 2/std::max((int)sh.nT, 1)));
          if(!(nrt > 0 && smax*sh.nJ*nrt <= 64 && smax*sh.nJ <= 16)) ts = false;
        }
      if(H.asm_ts_off || !Y->asm_pent_ok) ts = false;
      if(H.asm_lds_len == 18 && xvec && xt && ts)
        ASM_LAUNCH((k_assemble_mfma<18, true, true, true>), dim3(dlg_cdiv(nruns, TPB/64)), dim3(TPB),
                           sizeof(double)*(TPB/64)*(ASM_U*4*18 + ASM_TBUF), st, Y->asm_run, nruns, Y->asm_mtask,
                           Y->asm_kg, Y->asm_shape, Y->asm_tdest, Y->asm_pdest, Jv, Lx, Y->asm_part, 18, xvec, Y->jtp, Jt_x, only_shape, (uint32_t)H.lx_size, Y->asm_pent);
      else if(H.asm_lds_len == 18 && xvec && xt)
        ASM_LAUNCH((k_assemble_mfma<18, true, true>), dim3(dlg_cdiv(nruns, TPB/64)), dim3(TPB),
                           sizeof(double)*(TPB/64)*ASM_U*4*18, st, Y->asm_run, nruns, Y->asm_mtask,
                           Y->asm_kg, Y->asm_shape, Y->asm_tdest, Y->asm_pdest, Jv, Lx, Y->asm_part, 18, xvec, Y->jtp, Jt_x, only_shape, (uint32_t)H.lx_size, Y->asm_pent);
      else if(H.asm_lds_len == 18 && xvec)
        ASM_LAUNCH((k_assemble_mfma<18, true>), dim3(dlg_cdiv(nruns, TPB/64)), dim3(TPB),
                           sizeof(double)*(TPB/64)*ASM_U*4*18, st, Y->asm_run, nruns, Y->asm_mtask,
                           Y->asm_kg, Y->asm_shape, Y->asm_tdest, Y->asm_pdest, Jv, Lx, Y->asm_part, 18, xvec, Y->jtp, Jt_x, only_shape, (uint32_t)H.lx_size, Y->asm_pent);
      else if(xvec)
        ASM_LAUNCH((k_assemble_mfma<0, true>), dim3(dlg_cdiv(nruns, TPB/64)), dim3(TPB),
                           sizeof(double)*(TPB/64)*ASM_U*4*H.asm_lds_len, st, Y->asm_run, nruns, Y->asm_mtask,
                           Y->asm_kg, Y->asm_shape, Y->asm_tdest, Y->asm_pdest, Jv, Lx, Y->asm_part,
                           H.asm_lds_len, xvec, Y->jtp, Jt_x, only_shape, (uint32_t)H.lx_size, Y->asm_pent);
      else if(H.asm_lds_len == 18)
        ASM_LAUNCH((k_assemble_mfma<18, false>), dim3(dlg_cdiv(nruns, TPB/64)), dim3(TPB),
                           sizeof(double)*(TPB/64)*ASM_U*4*18, st, Y->asm_run, nruns, Y->asm_mtask,
                           Y->asm_kg, Y->asm_shape, Y->asm_tdest, Y->asm_pdest, Jv, Lx, Y->asm_part, 18, nox, nojt, nojt, only_shape, (uint32_t)H.lx_size, Y->asm_pent);
      else
        ASM_LAUNCH((k_assemble_mfma<0, false>), dim3(dlg_cdiv(nruns, TPB/64)), dim3(TPB),
                           sizeof(double)*(TPB/64)*ASM_U*4*H.asm_lds_len, st, Y->asm_run, nruns, Y->asm_mtask,
                           Y->asm_kg, Y->asm_shape, Y->asm_tdest, Y->asm_pdest, Jv, Lx, Y->asm_part,
                           H.asm_lds_len, nox, nojt, nojt, only_shape, (uint32_t)H.lx_size, Y->asm_pent);
    }
    if(nt > 0)
      hipLaunchKernelGGL(k_assemble, dim3(dlg_cdiv(nt, TPB/64)), dim3(TPB), 0, st, Y->asm_ctask, nt,
                         Y->asm_batch, Y->asm_rho, Y->asm_pair, Y->asm_slot, Jv, Lx, Y->asm_part);
  }
  if(xvec)
  {
    // Jt*x from the task records (first: it is what the caller waits for)
    const int ns = (int)H.jf_short.size(), nl = (int)H.jf_long.size();
    if(ns > 0)
      hipLaunchKernelGGL(k_jtx_fin2_short, dim3(dlg_cdiv(ns, TPB/16)), dim3(TPB), 0, st, Y->jf_short, ns, Y->jf_ptr,
                         Y->jf_ent, Y->jf_var0, Y->jf_w, Y->jtp, Jt_x);
    // (an evaluation whose point may be factorised at once: the one long list's launch also sets the augmented rows,
    // re-arms the pivot flag and raises the word for the second stream -- fin on the side needs no kernel of its own)
    const bool fuse_aug = defer_fin && ns == 0 && nl == 1 && Y->aug_of_var && Y->fin_flag && !b->sharded() && H.part_nranks <= 1;
    Y->aug_fused_epoch = 0;
    if(nl > 0)
    {
      JfAug ja{nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, nl*JFL_SEG};
      int extra = 0;
      if(fuse_aug)
      {
        Y->aug_fused_epoch = ++Y->fin_epoch;
        ja = JfAug{Lx, Y->augpos, Y->perm, Y->aug_of_var, Y->jf_listed, H.N, Y->d_info, Y->fin_flag, Y->aug_fused_epoch, nl*JFL_SEG};
        extra = dlg_cdiv(H.N, 1024);
      }
      hipLaunchKernelGGL(k_jtx_fin2_long, dim3(nl*JFL_SEG + extra), dim3(1024), 0, st, Y->jf_long, Y->jf_ptr, Y->jf_ent,
                         Y->jf_var0, Y->jf_w, Y->jtp, Jt_x, Y->jf_lpart, Y->jf_lcnt, ja);
    }
    if(fuse_aug)
    {
      Y->fin_pending_Lx = Lx; Y->fin_pending_rhs = nullptr;      // (the stages only: the augmented row is set ...)
      Y->spec_aug_fused = Jt_x;                                   // (... sparse_eval_assemble records that behind this call)
      DLG_LAUNCH_CHECK();
      return DLG_OK;
    }
  }
  if(defer_fin) { Y->fin_pending_Lx = Lx; Y->fin_pending_rhs = Jt_x; DLG_LAUNCH_CHECK(); return DLG_OK; }
  return assemble_fin_launch(b, Lx);
}
static int assemble_fin_launch(dlg_backend* b, double* Lx, const double* aug_rhs, bool* aug_done)
{
  SparseSym* Y = b->sym;
  const SymHost& H = Y->H;
  hipStream_t st = b->stream;
  if(aug_done) *aug_done = false;
  for(size_t q = 0; q + 2 < H.fin2_stage.size(); q += 3)
  {
    const int f0 = H.fin2_stage[q], ns = H.fin2_stage[q+1], nl = H.fin2_stage[q+2];
    if(ns > 0)
    {
      // the first launch of short lists also sets the augmented row, if the caller has one to set
      AugRowArgs aug = {nullptr, nullptr, nullptr, 0, nullptr};
      int gaug = 0;
      if(aug_rhs && aug_done && !*aug_done)
      {
        aug = AugRowArgs{Y->augpos, Y->perm, aug_rhs, H.N, Y->d_info};
        gaug = dlg_cdiv(H.N, TPB); *aug_done = true;
      }
      hipLaunchKernelGGL(k_assemble_fin2_short, dim3(dlg_cdiv(ns, TPB/64) + gaug), dim3(TPB), 0, st, Y->asm_fin2 + f0, ns,
                         Y->asm_fin2_list, Y->asm_part, Lx, aug);
    }
    if(nl > 0)
      hipLaunchKernelGGL(k_assemble_fin2_long, dim3(nl), dim3(1024), 0, st, Y->asm_fin2 + f0 + ns,
                         Y->asm_fin2_list, Y->asm_part, Lx);
  }
  const int nf = (int)H.asm_cfin.size();
  if(nf > 0)
    hipLaunchKernelGGL(k_assemble_fin, dim3(nf), dim3(1024), 0, st, Y->asm_cfin, Y->asm_slot,
                       Y->asm_part, Lx);
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}

// Sharded rows: the partial JtJ of all ranks are summed before the factorisation.  Only the
// structural non-zeros of JtJ travel (half of the panel buffer is fill and padding): the first
// call finds them -- the assembly of an all-ones Jacobian, summed over the ranks once, is non-zero
// exactly there on every rank -- and from then on the entries are packed, all-reduced, unpacked.
namespace {
__global__ void __launch_bounds__(TPB) k_fill(double* __restrict__ a, size_t n, double v)
{ for(size_t i = blockIdx.x*(size_t)TPB + threadIdx.x; i < n; i += (size_t)gridDim.x*TPB) a[i] = v; }
__global__ void __launch_bounds__(TPB) k_pack(const double* __restrict__ Lx, const uint32_t* __restrict__ idx, size_t n,
                                              double* __restrict__ buf)
{ for(size_t i = blockIdx.x*(size_t)TPB + threadIdx.x; i < n; i += (size_t)gridDim.x*TPB) buf[i] = Lx[idx[i]]; }
__global__ void __launch_bounds__(TPB) k_unpack(double* __restrict__ Lx, const uint32_t* __restrict__ idx, size_t n,
                                                const double* __restrict__ buf)
{ for(size_t i = blockIdx.x*(size_t)TPB + threadIdx.x; i < n; i += (size_t)gridDim.x*TPB) Lx[idx[i]] = buf[i]; }
}
static int allreduce_panels(dlg_backend* b, bool* again)
{
  *again = false;
  SparseSym* Y = b->sym;
  const SymHost& H = Y->H;
  hipStream_t st = b->stream;
  if(!b->sharded() || b->part_requested) return DLG_OK;      // (subtree partition: sparse_partition_reduce, at the cut)
  if((size_t)H.lx_size >= ((size_t)1 << 32))
    return dlg_allreduce_dev(b, Y->Lx, (size_t)H.lx_size);
  if(!Y->ar_idx)
  {
    // (the caller's assembled panels are overwritten here: it assembles again afterwards)
    double* ones = nullptr;
    const size_t nv = Y->nnz_loc ? Y->nnz_loc : 1;
    DLG_HIP(hipMalloc(&ones, sizeof(double)*nv));
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(TPB), 0, st, ones, nv, 1.0);
    int rc = assemble_launch(b, ones);
    if(rc == DLG_OK) rc = dlg_allreduce_dev(b, Y->Lx, (size_t)H.lx_size);
    std::vector<double> h((size_t)H.lx_size);
    if(rc == DLG_OK && (hipMemcpyAsync(h.data(), Y->Lx, sizeof(double)*h.size(), hipMemcpyDeviceToHost, st) != hipSuccess ||
                        hipStreamSynchronize(st) != hipSuccess))
    { dlg_set_error("all-reduce index: download failed"); rc = DLG_ERR_HIP; }
    (void)hipFree(ones);
    DLG_CHECK(rc);
    std::vector<uint32_t> idx;
    idx.reserve((size_t)H.nnz_JtJ_lower + 16);
    for(size_t i = 0; i < h.size(); i++) if(h[i] != 0.0) idx.push_back((uint32_t)i);
    Y->ar_n = idx.size();
    DLG_CHECK(upload(Y->ar_idx, idx)); Y->allocs.push_back(Y->ar_idx);
    DLG_HIP(hipMalloc(&Y->ar_buf, sizeof(double)*(Y->ar_n ? Y->ar_n : 1))); Y->allocs.push_back(Y->ar_buf);
    *again = true;                              // tell the caller to assemble again
    return DLG_OK;
  }
  hipLaunchKernelGGL(k_pack, dim3(2048), dim3(TPB), 0, st, Y->Lx, Y->ar_idx, Y->ar_n, Y->ar_buf);
  DLG_LAUNCH_CHECK();
  DLG_CHECK(dlg_allreduce_dev(b, Y->ar_buf, Y->ar_n));
  hipLaunchKernelGGL(k_unpack, dim3(2048), dim3(TPB), 0, st, Y->Lx, Y->ar_idx, Y->ar_n, Y->ar_buf);
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}

// K4: JtJ straight into the supernode panels, summed over the ranks, + lambda, + augmented row
int sparse_assemble(dlg_backend* b, int s, double lambda)
{
  SparseSym* Y = b->sym;
  DlgSlot& S = b->slot[s];
  const SymHost& H = Y->H;
  hipStream_t st = b->stream;
  // a speculative assembly may be in flight on the second stream (sparse_assemble_speculative): it
  // shares the partial-sum buffers with any other assembly and, if it is this slot's, it IS the assembly
  if(Y->spec_inflight)
  {
    DLG_HIP(hipStreamWaitEvent(st, Y->ev_spec, 0));
    Y->spec_inflight = false;
  }
  DLG_CHECK(sparse_assemble_finish(b));
  // The attempt before this one was stopped by the look at the diagonal (sparse_host.hip, sparse_note_breakdown): its panels
  // are the assembly's still -- same slot, same inputs: they get the difference of the lambdas, nothing is assembled.
  if(Y->intact_Lx && Y->intact_Lx == Y->Lx && Y->intact_slot == s && Y->intact_J == S.Jin() && S.have_Jtx && Y->aug_rhs == S.Jt_x &&
     !b->sharded() && H.part_nranks <= 1)
  {
    const double dl = lambda - Y->intact_lambda;
    Y->intact_Lx = nullptr;
    if(dl != 0.0)
    {
      DLG_CHECK(sparse_fin_side_gate(b));
      hipLaunchKernelGGL(k_add_lambda, dim3(dlg_cdiv(H.N, TPB)), dim3(TPB), 0, st, Y->Lx, Y->diagpos, H.N,
                         dl, Y->col_sn, Y->sn_owner, -1);
      DLG_LAUNCH_CHECK();
    }
    Y->info_armed = false;            // (sparse_factorize arms the pivot word with a copy)
    return DLG_OK;
  }
  Y->intact_Lx = nullptr;
  bool adopted = false;
  if(Y->spec_valid && Y->spec_slot == s && Y->spec_J == S.Jin())
  {
    std::swap(Y->Lx, Y->Lx_spec);           // the panels assembled beside K1 become the factor's panels
    Y->spec_valid = false;
    adopted = true;
    // the other buffer (the previous factor) is free from here on: sparse_zero_spare clears it behind the
    // step's last fetch (step_finish), the next evaluation finds it zeroed
    Y->spare_zeroed = false; Y->spare_dirty = true;
  }
  else
  {
    DlgProfScope pt(b, DLG_PROF_K4_TOTAL);
    Y->spec_valid = Y->spec_valid && !(Y->spec_slot == s);      // (a second buffer that did not fit is of no further use)
    DLG_CHECK(assemble_launch(b, S.Jin()));
  }
  // contiguous row sharding: sum the partial JtJ of all ranks before factorising
  {
    bool again = false;
    DLG_CHECK(allreduce_panels(b, &again));
    if(again)
    {
      DLG_CHECK(assemble_launch(b, S.Jin()));
      DLG_CHECK(allreduce_panels(b, &again));
    }
  }
  // subtree partition: the replicated panels above the cut get lambda and their right-hand side after
  // the sum over the ranks (sparse_partition_reduce); everything else here
  const int phase = H.part_nranks > 1 ? 0 : -1;
  if(lambda != 0.0) DLG_CHECK(sparse_fin_side_gate(b));      // (lambda goes onto diagonal entries the stages on the second stream store)
  if(lambda != 0.0)
    hipLaunchKernelGGL(k_add_lambda, dim3(dlg_cdiv(H.N, TPB)), dim3(TPB), 0, st, Y->Lx, Y->diagpos, H.N,
                       lambda, Y->col_sn, Y->sn_owner, phase);
  // the right-hand side rides along as the last row of every panel: y = L^-1 P Jt_x falls out
  Y->aug_rhs = nullptr;
  Y->info_armed = false;
  if(S.have_Jtx && adopted && Y->spec_aug_rhs == S.Jt_x)
  {
    // the adopted panels carry their right-hand side already (sparse_assemble_finish)
    Y->aug_rhs = S.Jt_x;
    Y->info_armed = Y->info_clean;
  }
  else if(S.have_Jtx)
  {
    hipLaunchKernelGGL(k_set_aug_row, dim3(dlg_cdiv(H.N, TPB)), dim3(TPB), 0, st, Y->Lx, Y->col_sn, Y->augpos,
                       Y->perm, S.Jt_x, H.N, Y->d_info, Y->sn_owner, phase);
    Y->aug_rhs = S.Jt_x;
    Y->info_armed = true;
  }
  else sparse_mark_unclean(b);       // (no right-hand side: the leaves' last rows keep what they held -- not something to build on)
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}

// ---- subtree partition: the one sum over the ranks inside a factorisation ----------------------
// Between the last level below the cut and the first one above it: the panels of the replicated
// supernodes (assembled entries of every rank's rows + the updates of every rank's subtrees, all
// partial sums) and the update matrices that cross the cut are packed, summed (RCCL on the stream,
// or the hook) and unpacked; then the replicated panels get lambda and their right-hand side, once.
namespace {
__global__ void __launch_bounds__(TPB) k_red_pack(const int64_t* __restrict__ off, const int* __restrict__ len,
                                                  const int* __restrict__ kind, const int64_t* __restrict__ dst,
                                                  const double* __restrict__ Lx, const double* __restrict__ uscr,
                                                  double* __restrict__ buf, size_t total, const int* __restrict__ info)
{
  const int sg = blockIdx.y;
  const double* src = (kind[sg] == 0 ? Lx : uscr) + off[sg];
  double* d = buf + dst[sg];
  const int n = len[sg];
  const bool zero = kind[sg] == 2;              // another rank's update matrix: nothing from here
  for(int i = blockIdx.x*TPB + threadIdx.x; i < n; i += gridDim.x*TPB) d[i] = zero ? 0.0 : src[i];
  // the pivot flag of the levels below the cut rides along: a rank sees only its own subtrees' pivots,
  // but every rank must take the same decision (and raise lambda together)
  if(sg == 0 && blockIdx.x == 0 && threadIdx.x == 0) buf[total] = (*info != 0x7fffffff) ? 1.0 : 0.0;
}
__global__ void __launch_bounds__(TPB) k_red_unpack(const int64_t* __restrict__ off, const int* __restrict__ len,
                                                    const int* __restrict__ kind, const int64_t* __restrict__ dst,
                                                    double* __restrict__ Lx, double* __restrict__ uscr,
                                                    const double* __restrict__ buf, size_t total, int* __restrict__ info)
{
  const int sg = blockIdx.y;
  double* d = (kind[sg] == 0 ? Lx : uscr) + off[sg];
  const double* src = buf + dst[sg];
  const int n = len[sg];
  for(int i = blockIdx.x*TPB + threadIdx.x; i < n; i += gridDim.x*TPB) d[i] = src[i];
  if(sg == 0 && blockIdx.x == 0 && threadIdx.x == 0 && buf[total] > 0.0) atomicMin(info, 0);
}
}
int sparse_partition_reduce(dlg_backend* b)
{
  SparseSym* Y = b->sym;
  const SymHost& H = Y->H;
  hipStream_t st = b->stream;
  if(H.part_nranks <= 1) return DLG_OK;
  if(Y->n_red_seg > 0)
  {
    hipLaunchKernelGGL(k_red_pack, dim3(16, Y->n_red_seg), dim3(TPB), 0, st, Y->red_off, Y->red_len, Y->red_kind,
                       Y->red_dst, Y->Lx, Y->uscr, Y->red_buf, Y->red_n, Y->d_info);
    DLG_LAUNCH_CHECK();
    DLG_CHECK(dlg_allreduce_dev(b, Y->red_buf, Y->red_n + 1));
    hipLaunchKernelGGL(k_red_unpack, dim3(16, Y->n_red_seg), dim3(TPB), 0, st, Y->red_off, Y->red_len, Y->red_kind,
                       Y->red_dst, Y->Lx, Y->uscr, Y->red_buf, Y->red_n, Y->d_info);
  }
  if(Y->cur_lambda != 0.0)
    hipLaunchKernelGGL(k_add_lambda, dim3(dlg_cdiv(H.N, TPB)), dim3(TPB), 0, st, Y->Lx, Y->diagpos, H.N,
                       Y->cur_lambda, Y->col_sn, Y->sn_owner, 1);
  if(Y->aug_rhs)
    hipLaunchKernelGGL(k_set_aug_row, dim3(dlg_cdiv(H.N, TPB)), dim3(TPB), 0, st, Y->Lx, Y->col_sn, Y->augpos,
                       Y->perm, Y->aug_rhs, H.N, Y->d_info, Y->sn_owner, 1);
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}

// ---- speculative assembly (K4 beside K1) -----------------------------------------------------------
// The assembly needs nothing but J, so it can start as soon as J is on the device: when the caller of
// dlg_point_eval expects the point to be factorised (the driver does once steps need the
// Gauss-Newton step), the assembly of the slot's J runs on the second stream, into a second panel
// buffer, while Jt*x and the norms run on the main stream and the host looks at the gradient.  A
// later sparse_assemble of the same slot adopts the buffer (pointer swap); an unused one is
// dropped.  Same kernels, same sums: the panels are bit-identical to the in-line assembly.
int sparse_assemble_speculative(dlg_backend* b, int s)
{
  SparseSym* Y = b->sym;
  if(!Y || !b->aux_stream || b->sharded() || Y->H.part_nranks > 1) return DLG_OK;
  const SymHost& H = Y->H;
  DlgSlot& S = b->slot[s];
  if(!Y->Lx_spec)
  {
    DLG_HIP(hipMalloc(&Y->Lx_spec, sizeof(double)*((size_t)H.lx_size + 8))); Y->allocs.push_back(Y->Lx_spec);
    DLG_HIP(hipEventCreateWithFlags(&Y->ev_spec, hipEventDisableTiming));
    DLG_HIP(hipEventCreateWithFlags(&Y->ev_spec_fork, hipEventDisableTiming));
  }
  // behind everything enqueued so far on the main stream (the upload of J, a factorisation that still
  // reads the partial-sum buffers), and behind an earlier speculative assembly (same stream)
  DLG_HIP(hipEventRecord(Y->ev_spec_fork, b->stream));
  DLG_HIP(hipStreamWaitEvent(b->aux_stream, Y->ev_spec_fork, 0));
  hipStream_t main_stream = b->stream;
  b->stream = b->aux_stream;
  int rc;
  {
    DlgProfScope pt(b, DLG_PROF_K4_TOTAL);
    Y->spare_zeroed = false; Y->spare_dirty = false; Y->spec_aug_rhs = nullptr;
    rc = assemble_launch(b, S.Jin(), Y->Lx_spec);
  }
  b->stream = main_stream;
  DLG_CHECK(rc);
  DLG_HIP(hipEventRecord(Y->ev_spec, b->aux_stream));
  Y->spec_inflight = true; Y->spec_valid = true; Y->spec_slot = s; Y->spec_J = S.Jin();
  return DLG_OK;
}
// K1 + K4 in one pass over J: the assembly kernel forms Jt*x of its rows as a by-product (JTX), the
// panels go to the second panel buffer and are adopted by the factorisation of this slot's point
// (sparse_assemble) like a speculative assembly -- only that nothing is speculative about the pass
// over J: the gradient is needed at every evaluation.  *done = 0 if the schedule cannot (a column
// block assembled by the LDS kernel, sharded rows): the caller runs K1.
int sparse_eval_assemble(dlg_backend* b, int s, int* done)
{
  *done = 0;
  SparseSym* Y = b->sym;
  if(!Y) { dlg_set_error("dlg_sparse_set_pattern must be called first"); return DLG_ERR_STATE; }
  const SymHost& H = Y->H;
  if(!H.asm_jtx_ok || !Y->jtp) return DLG_OK;
  // (sharded rows / subtree partition: the rank's rows give its share of Jt*x and of JtJ as the separate
  // kernels would; the sums over the ranks follow where they always did)
  DlgSlot& S = b->slot[s];
  hipStream_t st = b->stream;
  if(!Y->Lx_spec)
  {
    DLG_HIP(hipMalloc(&Y->Lx_spec, sizeof(double)*((size_t)H.lx_size + 8))); Y->allocs.push_back(Y->Lx_spec);
    DLG_HIP(hipEventCreateWithFlags(&Y->ev_spec, hipEventDisableTiming));
    DLG_HIP(hipEventCreateWithFlags(&Y->ev_spec_fork, hipEventDisableTiming));
  }
  if(Y->spec_inflight) { DLG_HIP(hipStreamWaitEvent(st, Y->ev_spec, 0)); Y->spec_inflight = false; }    // (shares the partial-sum buffers)
  if(!H.jtx_covers_all) DLG_HIP(hipMemsetAsync(S.Jt_x, 0, sizeof(double)*(size_t)b->N, st));          // var-blocks without rows
  const bool zeroed = Y->spare_zeroed && Y->spare_stream == st;      // (cleared on this very stream: sparse_zero_spare)
  Y->spare_zeroed = false; Y->spare_dirty = false;
  {
    DlgProfScope pt(b, DLG_PROF_K4_TOTAL);
    // (the partial-sum stages of JtJ wait until the caller has Jt*x on its way to the host: sparse_assemble_finish)
    DLG_CHECK(assemble_launch(b, S.Jin(), Y->Lx_spec, S.xin(), S.Jt_x, zeroed, true));
    if(b->sharded()) Y->fin_pending_rhs = nullptr;     // Jt*x is not summed over the ranks yet: the augmented row waits for the factorisation
  }
  Y->spec_valid = true; Y->spec_slot = s; Y->spec_J = S.Jin(); Y->spec_aug_rhs = nullptr;
  if(Y->spec_aug_fused) { Y->spec_aug_rhs = Y->spec_aug_fused; Y->info_clean = true; Y->spec_aug_fused = nullptr; }    // (set with the Jt*x sums: assemble_launch)
  *done = 1;
  return DLG_OK;
}
// the panel buffer that a factorisation just left behind (sparse_assemble swapped it out) is cleared
// behind the point the host waits for (step_finish): in stream order before the next assembly
// (ordered_for: the stream whose later work is ordered behind this clear by other means -- the Cauchy step's join, when
// the clear runs on the second stream beside the factorisation: cauchy_fork_enqueue)
int sparse_zero_spare(dlg_backend* b, hipStream_t ordered_for)
{
  SparseSym* Y = b->sym;
  if(!Y || !Y->spare_dirty || !Y->Lx_spec) return DLG_OK;
  // the spare buffer holds a factor somebody may still turn to (sparse_hold_factor: the one a factorisation
  // enqueued ahead of the caller's decision displaced): it is cleared once that question is settled
  if(Y->held_Lx && Y->held_Lx == Y->Lx_spec) return DLG_OK;
  Y->spare_dirty = false;
  DLG_CHECK(clear_panels(b, Y->Lx_spec, b->stream));
  Y->spare_zeroed = true; Y->spare_stream = ordered_for ? ordered_for : b->stream;
  return DLG_OK;
}
// the second panel buffer holds the assembly of slot s's point with the Jacobian values at J (sparse_eval_assemble)
bool sparse_spec_is(const dlg_backend* b, int s, const double* J)
{
  const SparseSym* Y = b->sym;
  return Y && Y->spec_valid && Y->spec_slot == s && Y->spec_J == J;
}
void sparse_spec_invalidate(dlg_backend* b, int s)
{
  SparseSym* Y = b->sym;
  if(Y && Y->spec_valid && Y->spec_slot == s) Y->spec_valid = false;
  if(Y && Y->intact_slot == s) Y->intact_Lx = nullptr;      // (new inputs: panels assembled from the old ones are nobody's)
}

#ifdef DLG_ASM_PROFILE
extern "C" void dlg_asm_profile_dump()
{
  unsigned long long h[2][4];
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(h, HIP_SYMBOL(g_asm_prof), sizeof(h));
  for(int c = 0; c < 2; c++)
    if(h[c][3])
      fprintf(stderr, "assembly kernel, %s tasks: %llu waves, %llu iterations, per iteration %.0f clocks at the head (prefetch consumed, next issued, tile written) + %.0f in the products and stores\n",
              c ? "point (transient)" : "camera", h[c][3], h[c][2], (double)h[c][0]/h[c][2], (double)h[c][1]/h[c][2]);
  memset(h, 0, sizeof(h));
  hipMemcpyToSymbol(HIP_SYMBOL(g_asm_prof), h, sizeof(h));
}
#endif
