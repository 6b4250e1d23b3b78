// panel_factor.h -- in-workgroup Cholesky of a dense column-major panel
// (nrows x w, nrows >= w): the w x w top block is factored, the rows below are
// solved against it.  Shared by the sparse supernode kernel and the dense
// blocked potrf (diagonal blocks).
//
// Variants, all left-looking over column blocks of 8 with thread = panel row:
//   panel_factor            (1) every thread brings the 8 block-column entries of its row(s) up
//                           to date against all previous columns (own read + the 8 block-row
//                           entries as 4 broadcast 16-byte reads -> 8 FMAs; with MFMA_SWEEP the
//                           matrix cores do this step, see panel_mfma_tiles); (2) barrier; every
//                           thread factors the 8x8 diagonal block redundantly in registers
//                           (right-looking, reciprocal square roots); (3) barrier; forward
//                           substitution of the thread's row against the 8x8 factor.
//   panel_factor_mfma       the same steps with a dedicated diagonal wave (>= 4 waves): the
//                           block factorisation leaves the other waves' path, 2 barriers.
//   panel_factor_blockdiag  block-diagonal top (merged sibling leaves): no sweep at all;
//   bd_compact_members/rows the same for unsliced panels without staging the top block (end of file).
// ALIGNED16: P is 16-byte aligned and ldp is even (LDS panels) -> double2 reads.
// On a non-positive pivot the column index (col0 + j) is min-reduced into *info
// and the pivot is replaced by 1 so that the sweep finishes without NaN storms.
#pragma once
#include <hip/hip_runtime.h>

// phase time stamps for tools/micro/bench_panel.hip (compiled out everywhere else)
#ifndef DLG_PF_STAMP
#define DLG_PF_DECL
#define DLG_PF_STAMP(i)
#define DLG_PF_PIN(x)
#define DLG_PF_DONE
#endif
#ifndef DLG_PF_EVT           // tools/micro/bench_panel: time stamps of the hand-overs between the waves of panel_factor_b16
#define DLG_PF_EVT(J, e)
#endif
#ifndef DLG_PF_WLOG          // tools/micro/bench_panel -DBP_WLOG: every wave's own log of (block, tile, event, clock)
#define DLG_PF_WLOG(J, t, e)
#endif

// 1/sqrt(d) for d > 0: hardware v_rsq_f64 seed (about 1e-8 relative) + two Newton
// steps (relative error ~1e-16).  The pivot sqrt(d) = d * rsqrt(d) and its reciprocal
// come out of one short dependency chain instead of an IEEE sqrt followed by an IEEE
// divide.
__device__ __forceinline__ double dlg_rsqrt(double d)
{
  double y = __builtin_amdgcn_rsq(d);
  y = y*(1.5 - 0.5*d*y*y);
  y = y*(1.5 - 0.5*d*y*y);
  return y;
}

// MFMA_SWEEP (LDS panels, NT a multiple of 64): the bulk of step (1) is done by the matrix
// cores.  Every 16 columns the rows kb.. are cut into tiles of 16, the waves take tiles round
// robin and form  C[16 x 16] = L[tile rows][0:kb] * L[kb:kb+16][0:kb]'  (v_mfma_f64_16x16x4_f64,
// both operands read from the LDS panel: 16 consecutive rows of 4 columns) and subtract it in
// place; the second 8-column half of a 16-block is completed the same way against the first.
typedef double dlg_pf_v4d __attribute__((ext_vector_type(4)));
// one wave: tiles t0, t0 + ts, t0 + 2 ts, ... (< ntile) of the rows kb.., two at a time if PAIR
template <bool PAIR>
__device__ __forceinline__ void panel_mfma_tiles(double* P, int ldp, int nrows, int kb, int nb, int kbeg,
                                                 int lane, int t0, int ts, int ntile)
{
  const int mm = lane & 15, kq = lane >> 4;
  const bool bvalid = mm < nb;
  const double* bp = P + kb + (bvalid ? mm : 0) + kq*ldp;
  double* cp = P + (kb + (bvalid ? mm : 0))*ldp;
  for(int t = t0; t < ntile; t += (PAIR ? 2 : 1)*ts)
  {
    const int r0 = kb + 16*t, r1 = r0 + 16*ts;
    const bool two = PAIR && t + ts < ntile;
    const double* ap0 = P + min(r0 + mm, nrows - 1) + kq*ldp;
    const double* ap1 = P + min(r1 + mm, nrows - 1) + kq*ldp;
    // the accumulators start from the current entries and the B operand is negated, so the
    // result C - L L' only has to be written back
    dlg_pf_v4d acc0, acc1;
#pragma unroll
    for(int r = 0; r < 4; r++)
    {
      acc0[r] = cp[min(r0 + kq + 4*r, nrows - 1)];
      if(PAIR) acc1[r] = cp[min(r1 + kq + 4*r, nrows - 1)];
    }
    // 8 panel columns (2 k-steps) per iteration, operands of the next iteration in flight
    double b0 = bp[kbeg*ldp], b1 = bp[(kbeg + 4)*ldp], a00 = ap0[kbeg*ldp], a01 = ap0[(kbeg + 4)*ldp];
    double a10 = 0.0, a11 = 0.0;
    if(PAIR) { a10 = ap1[kbeg*ldp]; a11 = ap1[(kbeg + 4)*ldp]; }
    for(int k0 = kbeg; k0 < kb; k0 += 8)
    {
      const int kn = (k0 + 8 < kb) ? k0 + 8 : k0;
      const double nb0 = bp[kn*ldp], nb1 = bp[(kn + 4)*ldp];
      const double na00 = ap0[kn*ldp], na01 = ap0[(kn + 4)*ldp];
      double na10 = 0.0, na11 = 0.0;
      if(PAIR) { na10 = ap1[kn*ldp]; na11 = ap1[(kn + 4)*ldp]; }
      __builtin_amdgcn_sched_barrier(0);        // keep the prefetch ahead of the products
      const double vb0 = bvalid ? -b0 : 0.0, vb1 = bvalid ? -b1 : 0.0;
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a00, vb0, acc0, 0, 0, 0);
      if(PAIR) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a10, vb0, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a01, vb1, acc0, 0, 0, 0);
      if(PAIR) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a11, vb1, acc1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      b0 = nb0; b1 = nb1; a00 = na00; a01 = na01; a10 = na10; a11 = na11;
    }
#pragma unroll
    for(int r = 0; r < 4; r++)
    {
      const int row0 = r0 + kq + 4*r, row1 = r1 + kq + 4*r;
      // (never the strict upper triangle: the factor kernel may keep part of an update matrix there)
      if(bvalid && row0 < nrows && row0 >= kb + mm) cp[row0] = acc0[r];
      if(PAIR) if(bvalid && two && row1 < nrows) cp[row1] = acc1[r];
    }
  }
}
template <int NT>
__device__ __forceinline__ void panel_mfma_sweep(double* P, int ldp, int nrows, int kb, int nb, int kbeg, int tid)
{
  constexpr int NW = NT/64;
  panel_mfma_tiles<true>(P, ldp, nrows, kb, nb, kbeg, tid & 63, tid >> 6, NW, (nrows - kb + 15) >> 4);
  DLG_PF_STAMP(0);
  __syncthreads();
  DLG_PF_STAMP(1);
}
// ---- pieces of the 8-column step, shared by the variants below
// the 8x8 diagonal block (lower triangle) from an LDS panel: pairs of rows are 16-byte aligned
__device__ __forceinline__ void pf_load_block(double (&D)[8][8], const double* P, int ldp, int kb, int nb)
{
#pragma unroll
  for(int q = 0; q < 8; q++)
#pragma unroll
    for(int c2 = (q & ~1); c2 < 8; c2 += 2)
    {
      const double2 v = *reinterpret_cast<const double2*>(&P[(kb + c2) + (kb + q)*ldp]);
      D[c2][q] = v.x; D[c2 + 1][q] = v.y;
    }
  if(nb < 8)
  {
#pragma unroll
    for(int c = 0; c < 8; c++)
#pragma unroll
      for(int q = 0; q <= c; q++) if(c >= nb) D[c][q] = (c == q) ? 1.0 : 0.0;
  }
}
// right-looking Cholesky of the block in registers: once column c is scaled the trailing block
// is updated at once, so the next pivot only waits for one multiply-add after the reciprocal
// square root.  Returns the first non-positive pivot (or -1); that pivot is replaced by 1.
__device__ __forceinline__ int pf_factor_block(double (&D)[8][8], double (&Dinv)[8])
{
  int badcol = -1;
#pragma unroll
  for(int c = 0; c < 8; c++)
  {
    double d = D[c][c];
    if(!(d > 0.0)) { if(badcol < 0) badcol = c; d = 1.0; }
    const double inv = dlg_rsqrt(d);
    Dinv[c] = inv;
#pragma unroll
    for(int i = c + 1; i < 8; i++) D[i][c] *= inv;
#pragma unroll
    for(int j = c + 1; j < 8; j++)
#pragma unroll
      for(int i = j; i < 8; i++) D[i][j] -= D[i][c]*D[j][c];
    D[c][c] = d*inv;
  }
  return badcol;
}
// The diagonal wave holds the factored block in EVERY lane (the factorisation is the same instruction
// stream in all of them).  Write-back without a select chain: element e of the 44 (see pf_block_lane_addr)
// is moved into lane e of one register with only that lane enabled -- 44 x (s_mov exec + v_mov_b64) --
// and one ds_write_b64 of 44 lanes stores them all.  (The select chain it replaces kept 36 lane-index
// compares in spilled SGPRs, ~1400 clocks per block; 44 single-lane LDS stores cost as much: a lone
// wave issues an LDS store every ~30 clocks.  tools/micro/bench_pfchain.)  exec is restored inside the asm.
__device__ __forceinline__ double pf_block_to_lanes(const double (&D)[8][8], const double (&Dinv)[8])
{
  unsigned long long sv;
  double t = 0.0;
  asm volatile(
    "s_mov_b64 %[sv], exec\n\t"
    "s_mov_b32 exec_hi, 0\n\t"
    "s_mov_b32 exec_lo, 0x1\n\t"
    "v_mov_b64 %[t], %[d0]\n\t"
    "s_mov_b32 exec_lo, 0x2\n\t"
    "v_mov_b64 %[t], %[d1]\n\t"
    "s_mov_b32 exec_lo, 0x4\n\t"
    "v_mov_b64 %[t], %[d2]\n\t"
    "s_mov_b32 exec_lo, 0x8\n\t"
    "v_mov_b64 %[t], %[d3]\n\t"
    "s_mov_b32 exec_lo, 0x10\n\t"
    "v_mov_b64 %[t], %[d4]\n\t"
    "s_mov_b32 exec_lo, 0x20\n\t"
    "v_mov_b64 %[t], %[d5]\n\t"
    "s_mov_b32 exec_lo, 0x40\n\t"
    "v_mov_b64 %[t], %[d6]\n\t"
    "s_mov_b32 exec_lo, 0x80\n\t"
    "v_mov_b64 %[t], %[d7]\n\t"
    "s_mov_b32 exec_lo, 0x100\n\t"
    "v_mov_b64 %[t], %[d8]\n\t"
    "s_mov_b32 exec_lo, 0x200\n\t"
    "v_mov_b64 %[t], %[d9]\n\t"
    "s_mov_b32 exec_lo, 0x400\n\t"
    "v_mov_b64 %[t], %[d10]\n\t"
    "s_mov_b32 exec_lo, 0x800\n\t"
    "v_mov_b64 %[t], %[d11]\n\t"
    "s_mov_b32 exec_lo, 0x1000\n\t"
    "v_mov_b64 %[t], %[d12]\n\t"
    "s_mov_b32 exec_lo, 0x2000\n\t"
    "v_mov_b64 %[t], %[d13]\n\t"
    "s_mov_b32 exec_lo, 0x4000\n\t"
    "v_mov_b64 %[t], %[d14]\n\t"
    "s_mov_b32 exec_lo, 0x8000\n\t"
    "v_mov_b64 %[t], %[d15]\n\t"
    "s_mov_b32 exec_lo, 0x10000\n\t"
    "v_mov_b64 %[t], %[d16]\n\t"
    "s_mov_b32 exec_lo, 0x20000\n\t"
    "v_mov_b64 %[t], %[d17]\n\t"
    "s_mov_b32 exec_lo, 0x40000\n\t"
    "v_mov_b64 %[t], %[d18]\n\t"
    "s_mov_b32 exec_lo, 0x80000\n\t"
    "v_mov_b64 %[t], %[d19]\n\t"
    "s_mov_b32 exec_lo, 0x100000\n\t"
    "v_mov_b64 %[t], %[d20]\n\t"
    "s_mov_b32 exec_lo, 0x200000\n\t"
    "v_mov_b64 %[t], %[d21]\n\t"
    "s_mov_b32 exec_lo, 0x400000\n\t"
    "v_mov_b64 %[t], %[d22]\n\t"
    "s_mov_b32 exec_lo, 0x800000\n\t"
    "v_mov_b64 %[t], %[d23]\n\t"
    "s_mov_b32 exec_lo, 0x1000000\n\t"
    "v_mov_b64 %[t], %[d24]\n\t"
    "s_mov_b32 exec_lo, 0x2000000\n\t"
    "v_mov_b64 %[t], %[d25]\n\t"
    "s_mov_b32 exec_lo, 0x4000000\n\t"
    "v_mov_b64 %[t], %[d26]\n\t"
    "s_mov_b32 exec_lo, 0x8000000\n\t"
    "v_mov_b64 %[t], %[d27]\n\t"
    "s_mov_b64 exec, %[sv]"
    : [sv] "=&s"(sv), [t] "+v"(t)
    : [d0] "v"(D[0][0]), [d1] "v"(D[1][0]), [d2] "v"(D[1][1]), [d3] "v"(D[2][0]), [d4] "v"(D[2][1]), [d5] "v"(D[2][2]), [d6] "v"(D[3][0]), [d7] "v"(D[3][1]), [d8] "v"(D[3][2]), [d9] "v"(D[3][3]), [d10] "v"(D[4][0]), [d11] "v"(D[4][1]), [d12] "v"(D[4][2]), [d13] "v"(D[4][3]), [d14] "v"(D[4][4]), [d15] "v"(D[5][0]), [d16] "v"(D[5][1]), [d17] "v"(D[5][2]), [d18] "v"(D[5][3]), [d19] "v"(D[5][4]), [d20] "v"(D[5][5]), [d21] "v"(D[6][0]), [d22] "v"(D[6][1]), [d23] "v"(D[6][2]), [d24] "v"(D[6][3]), [d25] "v"(D[6][4]), [d26] "v"(D[6][5]), [d27] "v"(D[6][6]));
  asm volatile(
    "s_mov_b64 %[sv], exec\n\t"
    "s_mov_b32 exec_hi, 0\n\t"
    "s_mov_b32 exec_lo, 0x10000000\n\t"
    "v_mov_b64 %[t], %[d0]\n\t"
    "s_mov_b32 exec_lo, 0x20000000\n\t"
    "v_mov_b64 %[t], %[d1]\n\t"
    "s_mov_b32 exec_lo, 0x40000000\n\t"
    "v_mov_b64 %[t], %[d2]\n\t"
    "s_mov_b32 exec_lo, 0x80000000\n\t"
    "v_mov_b64 %[t], %[d3]\n\t"
    "s_mov_b32 exec_lo, 0\n\t"
    "s_mov_b32 exec_hi, 0x1\n\t"
    "v_mov_b64 %[t], %[d4]\n\t"
    "s_mov_b32 exec_hi, 0x2\n\t"
    "v_mov_b64 %[t], %[d5]\n\t"
    "s_mov_b32 exec_hi, 0x4\n\t"
    "v_mov_b64 %[t], %[d6]\n\t"
    "s_mov_b32 exec_hi, 0x8\n\t"
    "v_mov_b64 %[t], %[d7]\n\t"
    "s_mov_b32 exec_hi, 0x10\n\t"
    "v_mov_b64 %[t], %[d8]\n\t"
    "s_mov_b32 exec_hi, 0x20\n\t"
    "v_mov_b64 %[t], %[d9]\n\t"
    "s_mov_b32 exec_hi, 0x40\n\t"
    "v_mov_b64 %[t], %[d10]\n\t"
    "s_mov_b32 exec_hi, 0x80\n\t"
    "v_mov_b64 %[t], %[d11]\n\t"
    "s_mov_b32 exec_hi, 0x100\n\t"
    "v_mov_b64 %[t], %[d12]\n\t"
    "s_mov_b32 exec_hi, 0x200\n\t"
    "v_mov_b64 %[t], %[d13]\n\t"
    "s_mov_b32 exec_hi, 0x400\n\t"
    "v_mov_b64 %[t], %[d14]\n\t"
    "s_mov_b32 exec_hi, 0x800\n\t"
    "v_mov_b64 %[t], %[d15]\n\t"
    "s_mov_b64 exec, %[sv]"
    : [sv] "=&s"(sv), [t] "+v"(t)
    : [d0] "v"(D[7][0]), [d1] "v"(D[7][1]), [d2] "v"(D[7][2]), [d3] "v"(D[7][3]), [d4] "v"(D[7][4]), [d5] "v"(D[7][5]), [d6] "v"(D[7][6]), [d7] "v"(D[7][7]), [d8] "v"(Dinv[0]), [d9] "v"(Dinv[1]), [d10] "v"(Dinv[2]), [d11] "v"(Dinv[3]), [d12] "v"(Dinv[4]), [d13] "v"(Dinv[5]), [d14] "v"(Dinv[6]), [d15] "v"(Dinv[7]));
  return t;
}
// LDS byte address of the element lane `lane` holds after pf_block_to_lanes: (c, q), q <= c, in row-major
// order of the lower triangle for lanes 0..35, the reciprocal pivots dinv[0..7] for lanes 36..43
__device__ __forceinline__ unsigned pf_block_lane_addr(const double* P, int ldp, int kb, const double* dinv, int lane)
{
  const int c = (lane >= 1) + (lane >= 3) + (lane >= 6) + (lane >= 10) + (lane >= 15) + (lane >= 21) + (lane >= 28);
  const int q = lane - ((c*(c + 1)) >> 1);
  const double* p = (lane < 36) ? P + (kb + c) + (kb + q)*ldp : dinv + ((lane - 36) & 7);
  return (unsigned)(size_t)(const __attribute__((address_space(3))) double*)p;
}
// row r below a factored 8x8 block: x <- x * inv(D)' in place (same operations, same order, for FULL or not)
template <bool FULL>
__device__ __forceinline__ void pf_solve_row_lds(double* P, int ldp, int r, int kb, int nb, const double (&D)[8][8],
                                                 const double (&Dinv)[8])
{
  double x[8];
#pragma unroll
  for(int c = 0; c < 8; c++) x[c] = (FULL || c < nb) ? P[r + (kb + c)*ldp] : 0.0;
#pragma unroll
  for(int c = 0; c < 8; c++)
  {
    double v = x[c];
#pragma unroll
    for(int q = 0; q < c; q++) v -= x[q]*D[c][q];
    x[c] = v*Dinv[c];
  }
#pragma unroll
  for(int c = 0; c < 8; c++) if(FULL || c < nb) P[r + (kb + c)*ldp] = x[c];
}
// MFMA panel factorisation with a dedicated diagonal wave (LDS panels, NT >= 256).
// Per 8 columns, two barriers instead of three and the block factorisation off the other
// waves' path:
//   wave 0:   MFMA update of the first row tile (it holds the diagonal block), factors the
//             8x8 block in registers, writes it back and publishes the reciprocal pivots;
//   waves 1+: MFMA update of the remaining row tiles meanwhile;
//   barrier;  every thread fetches the factored block and solves its row; barrier.
template <int NT>
__device__ __forceinline__ void panel_factor_mfma(double* P, int ldp, int nrows, int w, int tid,
                                                  int* __restrict__ info, int col0)
{
  static_assert(NT >= 256, "panel_factor_mfma needs at least 4 waves");
  constexpr int NW = NT/64;
  __shared__ double s_dinv[8];
  const int lane = tid & 63, wv = tid >> 6;
  DLG_PF_DECL
  for(int kb = 0; kb < w; kb += 8)
  {
    const int nb = (w - kb < 8) ? w - kb : 8;
    const int ntile = (nrows - kb + 15) >> 4;
    const int nb16 = ((kb & 15) == 0) ? min(16, w - kb) : nb, kbeg = ((kb & 15) == 0) ? 0 : kb - 8;
    double D[8][8], Dinv[8];
    if(wv == 0)
    {
      if(kb > 0) panel_mfma_tiles<false>(P, ldp, nrows, kb, nb16, kbeg, lane, 0, ntile, ntile);
      DLG_PF_STAMP(2);
      pf_load_block(D, P, ldp, kb, nb);
      DLG_PF_STAMP(3);
      const int badcol = pf_factor_block(D, Dinv);
      DLG_PF_PIN(D[7][7]); DLG_PF_PIN(D[7][6]); DLG_PF_PIN(Dinv[7]);
      DLG_PF_STAMP(6);
      if(badcol >= 0 && tid == 0) atomicMin(info, col0 + kb + badcol);
      if(nb == 8)
      {
        const double t = pf_block_to_lanes(D, Dinv);
        if(lane < 44) *(__attribute__((address_space(3))) double*)(size_t)pf_block_lane_addr(P, ldp, kb, s_dinv, lane) = t;
      }
      else
      {
        // the last, partial block: thread (c, q) keeps element (c, q) (compile-time indices: D lives in registers)
        const int c = tid >> 3, q = tid & 7;
        double v = 0.0, dv = 0.0;
#pragma unroll
        for(int cc = 0; cc < 8; cc++)
        {
#pragma unroll
          for(int qq = 0; qq <= cc; qq++) v = (cc == c && qq == q) ? D[cc][qq] : v;
          dv = (cc == q) ? Dinv[cc] : dv;
        }
        if(q <= c && c < nb) P[(kb + c) + (kb + q)*ldp] = v;
        if(c == 0) s_dinv[q] = dv;
      }
    }
    else if(kb > 0)
    {
      // with 8 waves, wave 4 shares its SIMD with the diagonal wave and the two contend for instruction
      // issue: it stays out of the tile updates, the diagonal wave's chain has SIMD 0 to itself
      // (tools/micro/bench_ahead: 52.8k -> 50.1k cycles for a 187 x 60 panel)
#ifdef DLG_PF_EXPERIMENT_TILE_WAVES   // tools/micro only: 0 = nobody updates (timing of the diagonal wave alone), 3 = waves 1-3
      if(DLG_PF_EXPERIMENT_TILE_WAVES == 3) { if(wv < 4) panel_mfma_tiles<true>(P, ldp, nrows, kb, nb16, kbeg, lane, wv, 3, ntile); }
#else
      if(NW == 8) { if(wv != 4) panel_mfma_tiles<true>(P, ldp, nrows, kb, nb16, kbeg, lane, wv < 4 ? wv : wv - 1, NW - 2, ntile); }
      else panel_mfma_tiles<true>(P, ldp, nrows, kb, nb16, kbeg, lane, wv, NW - 1, ntile);
#endif
    }
    DLG_PF_STAMP(0);
    __syncthreads();
    DLG_PF_STAMP(1);
    if(kb + nb + (tid & ~63) < nrows)
    {
      if(wv != 0)
      {
        pf_load_block(D, P, ldp, kb, nb);
#pragma unroll
        for(int c = 0; c < 8; c++) Dinv[c] = s_dinv[c];
      }
      // a full block of 8 columns (all but the last of a panel) is straight-line code: 8 loads, the
      // 36-operation chain, 8 stores; the partial one tests every column against nb
      if(nb == 8)
        for(int r = kb + 8 + tid; r < nrows; r += NT) pf_solve_row_lds<true>(P, ldp, r, kb, 8, D, Dinv);
      else
        for(int r = kb + nb + tid; r < nrows; r += NT) pf_solve_row_lds<false>(P, ldp, r, kb, nb, D, Dinv);
    }
    DLG_PF_STAMP(4);
    __syncthreads();
    DLG_PF_STAMP(5);
  }
  DLG_PF_DONE
}
// ---- hand-overs between the waves of a workgroup inside a sweep (panel_factor_b16) ----------------------------
__device__ __forceinline__ void pf_wait(int* flag, int need)
{
#ifdef DLG_PF_BOUNDED_WAIT      // tools/micro: a protocol error shows as wrong numbers, not as a hung GPU
  int spins = 0;
  while(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(1);
#else
  while(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
#endif
}
// order the LDS traffic of ONE wave across a hand-over between its lanes (no instruction: LDS
// operations of a wave are executed in order; this only stops the compiler from moving them)
__device__ __forceinline__ void pf_wave_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// ---- panel factorisation by blocks of 16 columns, the diagonal tile in registers (LDS panels, NT >= 256) -------
// The sweeps above pay, per 8 columns, a chain of dependent steps that each cross LDS: tile update -> block load ->
// 8x8 factorisation in every lane -> write-back -> barrier -> row solves -> barrier (about 6.2k clocks on a 100 x 66
// panel, tools/micro/bench_panel).  Here wave 0 keeps the 16 x 16 diagonal tile T in the accumulators of
// v_mfma_f64_16x16x4 and never leaves its registers inside a block:
//   * T is symmetric, so register s of the tile (rows 4s .. 4s + 3 in the C/D layout) is at the same time
//     "lane (j, k) = T[j][4s + k]", the A/B operand layout of the four columns 4s .. 4s + 3;
//   * step s: the 4 x 4 pivot block comes out of the tile by v_readlane, is factored and inverted in uniform
//     registers (M = D^-1, 4 reciprocal square roots in a row -- the only serial part);
//   * X' = M * src'  -- ONE MFMA with M (padded) as A and the source register as B -- leaves the four finished
//     columns of L for all 16 rows of the tile in register 0 of the result, again in the operand layout;
//   * T -= X X'  (one MFMA, both operands that register) brings the remaining columns up to date;
//   * the rows of L_JJ^-1 fall out the same way from an identity tile that rides along (two more MFMAs, off the chain)
//     and are published in LDS.
// The other waves own the row tiles below (tile t -> wave 1 + (t - 1) mod (NW - 1)): per block they bring their tile
// up to date against all columns before the block (left-looking, as in the sweeps above, but TRANSPOSED so that the
// accumulators are at once the B operands of the next product), wait for L_JJ^-1 and multiply with it: four MFMAs, no
// row-by-row substitution.  Hand-offs are LDS words (wdone: blocks whose inverse is published; tdone[t]: blocks row
// tile t has finished), no workgroup barrier inside the sweep.  The diagonal tile of the next block belongs to the
// wave that finishes first in every block (the tiles are taken in increasing order).
constexpr int PF_B16_WS = 17;          // row stride of the published inverse (doubles; 16 would put a column on one bank)
constexpr int PF_B16_MAXT = 32;        // row tiles: nrows <= 512
#define PF_NEG_A 1                     // v_mfma_f64: the BLGP field holds the NEG bits, bit 0 negates A (no VALU negation between the products)
// W: L_JJ^-1 of the last two blocks; A / E: the NEXT diagonal tile's rows, brought up to date by their wave and handed
// to wave 0 (A: the 16 columns of the current block, transposed; E: the diagonal tile itself against all columns before
// the block); wdone / adone / tdone: the hand-off words
struct pf_b16_lds { double W[2][16*PF_B16_WS]; double A[4*64]; double E[4*64]; int wdone; int adone; int tdone[PF_B16_MAXT]; };

__device__ __forceinline__ double pf_readlane64(double v, int l)
{
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// the diagonal tile of the block at kb (nb columns) as the panel holds it: both triangles from the lower one, identity
// where the block is short; register r, lane (n, kq) = T[kq + 4 r][n]
__device__ __forceinline__ dlg_pf_v4d pf_b16_diag_tile(const double* P, int ldp, int kb, int nb, int mm, int kq)
{
  dlg_pf_v4d U;
#pragma unroll
  for(int r = 0; r < 4; r++)
  {
    const int row = kq + 4*r, hi = max(row, mm), lo = min(row, mm);
    const bool valid = hi < nb;
    const double v = P[(kb + (valid ? hi : 0)) + (kb + (valid ? lo : 0))*ldp];
    U[r] = valid ? v : (row == mm ? 1.0 : 0.0);
  }
  return U;
}
// 1/sqrt(d) on the pivot chain: v_rsq_f64 seed and ONE third-order step  y (1 + e/2 + 3 e^2/8),  e = 1 - d y^2  (five
// dependent operations instead of the six of two Newton steps; the seed's 2^-23 goes to ~1e-21 before rounding)
__device__ __forceinline__ double pf_rsqrt3(double d)
{
  const double y = __builtin_amdgcn_rsq(d);
  const double e = __builtin_fma(-(d*y), y, 1.0);
  const double q = __builtin_fma(0.375, e, 0.5)*e;
  return __builtin_fma(y, q, y);
}
// A panel whose width is a few columns past a multiple of 16 (the 66-column separators of config #4: 4 x 16 + 2) used to
// pay a whole block step for those columns -- every row tile brought up to date against all columns before them on the
// matrix cores (16 products a tile for 2 of 16 columns), the next diagonal tile handed to wave 0 and back: 5.6 us of a
// 17.5 us sweep (profiles/r05_top_of_tree_levels.txt, 193 x 66 against 187 x 60).  They are finished on the vector pipe
// instead, by ALL threads, behind the sweep over the whole blocks (which treats their rows as rows below): thread =
// panel row, one pass over the row's finished entries -- own entry (consecutive lanes, consecutive addresses) times the
// pivot rows' entries (one 16-byte broadcast read for two of them) --, the little diagonal block factored by every
// thread from the pivot rows' sums (LDS, one barrier), then the row's own solve.  Columns finished that way:
constexpr int PF_VFIN_MAX = 4;          // (5 .. 8 columns: measured 0.4 - 0.7 us faster on 54 / 70 / 72-column panels, 0.6 us slower on a 40-column one: left to the matrix cores)
constexpr int PF_VFIN_SCR = 2*16*PF_B16_WS + 2*4*64;       // doubles of pf_b16_lds the finish may use (W, A, E: free behind the sweep)
// The waves share the pass two ways: a wave = (group of 64 rows, slice of the finished columns) -- 193 x 66: three groups x
// two slices --, every thread with two accumulator chains a column; the slices' partial sums meet in LDS (slice 0 adds
// them in slice order: the bits do not depend on who was first).  With one wave per row group and one chain the pass was a
// chain of 64 dependent multiply-adds behind their loads (5000 - 6600 clocks, tools/micro/bench_panel).
template <int NT, int NF>
__device__ __forceinline__ void pf_b16_vector_finish(double* P, int ldp, int nrows, int wm, int tid, double* scr,
                                                     int* __restrict__ info, int col0)
{
  // (the caller's barrier is behind us: every entry left of column wm is final; at most 512 rows: PF_B16_MAXT)
  constexpr int NWV = NT/64;
  constexpr int NFP = (NF + 1) & ~1;                            // accumulators kept per row (16-byte reads of the pivot rows)
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nlive = nrows - wm, G = (nlive + 63) >> 6;          // row groups (G <= NWV: the caller's condition)
  // slices of the columns: 1, 2 or 4 (shifts and compares only: an integer division costs more than a slice saves here)
  int ls = 0;
#ifndef DLG_VFIN_NOSLICE
  while(ls < 2 && (G << (ls + 1)) <= NWV && ((2 << ls) - 1)*G*64*NFP <= PF_VFIN_SCR - 4*NF*NF && (wm >> (ls + 1)) >= 8) ls++;
#endif
  const int S = 1 << ls;
  double* pivs = scr;                                           // [4][NF][NF]: the pivot rows' partial sums, slice by slice
  double* part = scr + 4*NF*NF;                                 // [S - 1][G*64][NFP]
  int g = wv, sl = 0;
  while(g >= G) { g -= G; sl++; }
  const bool in_pass = sl < S;
  const int i = wm + 64*g + lane;
  const bool live = i < nrows;
  double acc[NFP];
#pragma unroll
  for(int c = 0; c < NFP; c++) acc[c] = 0.0;
  if(in_pass)
  {
    const int ic = live ? i : nrows - 1;
    const int k0 = sl*(wm >> ls), k1 = k0 + (wm >> ls);
    const double* own = P + ic;
    const double* piv = P + wm;                  // rows wm .. wm + NF - 1 (wm is a multiple of 16, ldp even: 16-byte aligned pairs)
    double a0[NFP], a1[NFP];
#pragma unroll
    for(int c = 0; c < NFP; c++) { a0[c] = (sl == 0 && c < NF) ? P[ic + (wm + (c < NF ? c : 0))*ldp] : 0.0; a1[c] = 0.0; }
#pragma unroll 8
    for(int k = k0; k < k1; k += 2)
    {
      const double x0 = own[k*ldp], x1 = own[(k + 1)*ldp];
#pragma unroll
      for(int c = 0; c < NFP; c += 2)
      {
        const double2 b0 = *reinterpret_cast<const double2*>(piv + k*ldp + c), b1 = *reinterpret_cast<const double2*>(piv + (k + 1)*ldp + c);
        a0[c] = __builtin_fma(-x0, b0.x, a0[c]); a0[c + 1] = __builtin_fma(-x0, b0.y, a0[c + 1]);
        a1[c] = __builtin_fma(-x1, b1.x, a1[c]); a1[c + 1] = __builtin_fma(-x1, b1.y, a1[c + 1]);
      }
    }
#pragma unroll
    for(int c = 0; c < NFP; c++) acc[c] = a0[c] + a1[c];
    if(sl > 0)
    {
#pragma unroll
      for(int c = 0; c < NFP; c++) part[((sl - 1)*G*64 + g*64 + lane)*NFP + c] = acc[c];
    }
    // the pivot rows' sums, slice by slice: row wm + r holds the r-th row of the little diagonal block (its lower triangle counts)
    if(g == 0 && lane < NF)
    {
#pragma unroll
      for(int c = 0; c < NF; c++) pivs[(sl*NF + lane)*NF + c] = acc[c];
    }
  }
  DLG_PF_STAMP(6);
  __syncthreads();
  DLG_PF_STAMP(7);
  if(in_pass && sl == 0)
  {
    // the other slices' shares, in slice order -- for the thread's own row and for the pivot rows (every thread forms
    // the little block's sums itself: the same additions in the same order as the pivot rows' own threads)
    double d[NF][NF];
#pragma unroll
    for(int r = 0; r < NF; r++)
#pragma unroll
      for(int c = 0; c < NF; c++) d[r][c] = (c <= r) ? pivs[r*NF + c] : 0.0;
#pragma unroll
    for(int q = 1; q < 4; q++)
      if(q < S)
      {
#pragma unroll
        for(int c = 0; c < NFP; c++) acc[c] += part[((q - 1)*G*64 + g*64 + lane)*NFP + c];
#pragma unroll
        for(int r = 0; r < NF; r++)
#pragma unroll
          for(int c = 0; c < NF; c++) if(c <= r) d[r][c] += pivs[(q*NF + r)*NF + c];
      }
    DLG_PF_STAMP(8);
    // every thread of slice 0 factors the little block itself (at most four reciprocal square roots in a row) ...
    double inv[NF];
    int bad = 0x7fffffff;
#pragma unroll
    for(int c = 0; c < NF; c++)
    {
      double dd = d[c][c];
#pragma unroll
      for(int q = 0; q < c; q++) dd = __builtin_fma(-d[c][q], d[c][q], dd);
      if(!(dd > 0.0)) { bad = min(bad, wm + c); dd = 1.0; }
      inv[c] = pf_rsqrt3(dd);
#pragma unroll
      for(int r = c + 1; r < NF; r++)
      {
        double v = d[r][c];
#pragma unroll
        for(int q = 0; q < c; q++) v = __builtin_fma(-d[r][q], d[c][q], v);
        d[r][c] = v*inv[c];
      }
    }
    DLG_PF_STAMP(9);
    // ... and solves its own row against it (a pivot row: its entries up to the diagonal, which is 1 / inv)
    double x[NF];
#pragma unroll
    for(int c = 0; c < NF; c++)
    {
      double v = acc[c];
#pragma unroll
      for(int q = 0; q < c; q++) v = __builtin_fma(-x[q], d[c][q], v);
      x[c] = v*inv[c];
      if(live && i >= wm + c) P[i + (wm + c)*ldp] = x[c];          // (never above the diagonal: part of an update matrix may live there)
    }
    if(tid == 0 && bad != 0x7fffffff) atomicMin(info, col0 + bad);
    DLG_PF_STAMP(10);
  }
  __syncthreads();
  DLG_PF_STAMP(11);
}
// One to THREE row tiles of a helper wave in block J (tiles t1, t1 + dt, t1 + 2 dt -- all strictly below the next diagonal
// tile): brought up to date against the columns before the block, multiplied with L_JJ^-T, stored.  Round 6: a wave with
// two tiles used to finish one before it began the other -- every k-step four LDS reads, a round trip, two MFMAs -- and
// such waves, not wave 0's pivot chain, set the pace of a 187 x 60 sweep (they ran a whole block behind:
// tools/micro/bench_panel -DBP_WLOG).  All tiles of the wave in ONE loop: the diagonal tile's rows (the A operand) are
// read once for all, 2 NTL MFMAs hide a round trip instead of two.  Every tile keeps its own accumulator chain with the
// same products in the same order: the same bits as one tile at a time.  foreign: the tiles belonged to other waves in the
// blocks before this one: their rows are waited for.  (Tried with it, round 6: the LAST block's tiles dealt out anew over all
// eight waves, wave 4 and -- behind its last step -- wave 0 included: wave 4's products slow wave 0's chain by more than
// the end of the sweep gains, 18.9 against 18.5 us on 187 x 60, 22.0 against 21.1 on 193 x 66.  Not kept.)
template <int NTL>
__device__ __forceinline__ void pf_b16_normal_tiles(double* P, int ldp, int nrows, pf_b16_lds& S, int J, int kb, int nb, int t1, int dt,
                                                    int lane, int mm, int kq, bool foreign = false, bool own_first = false)
{
  // (own_first: the first tile is the block's own tile below a short top block -- only its rows past the block are stored,
  // and it is nobody's tile to wait for)
  const dlg_pf_v4d zero4 = {0.0, 0.0, 0.0, 0.0};
  int r0[NTL], rowc[NTL];
#pragma unroll
  for(int q = 0; q < NTL; q++) { r0[q] = 16*(t1 + q*dt); rowc[q] = min(r0[q] + mm, nrows - 1); }
  DLG_PF_WLOG(J, t1, 0);
  if(foreign && J > 0)
  {
#pragma unroll
    for(int q = 0; q < NTL; q++) pf_wait(&S.tdone[t1 + q*dt], J);
  }
  dlg_pf_v4d acc[NTL];
#pragma unroll
  for(int r = 0; r < 4; r++)
  {
    const int col = kq + 4*r, cc = kb + (col < nb ? col : 0);
#pragma unroll
    for(int q = 0; q < NTL; q++) { const double v = P[rowc[q] + cc*ldp]; acc[q][r] = (col < nb) ? v : 0.0; }
  }
  if(J > 0)
  {
    pf_wait(&S.tdone[J], J);
    const double* ap = P + kb + (mm < nb ? mm : 0) + kq*ldp;        // rows of the diagonal tile
    double a0 = ap[0], a1 = ap[4*ldp], b0[NTL], b1[NTL];
#pragma unroll
    for(int q = 0; q < NTL; q++) { const double* bp = P + rowc[q] + kq*ldp; b0[q] = bp[0]; b1[q] = bp[4*ldp]; }
    for(int k0 = 0; k0 < kb; k0 += 8)
    {
      const int kn = (k0 + 8 < kb) ? k0 + 8 : k0;
      const double na0 = ap[kn*ldp], na1 = ap[(kn + 4)*ldp];
      double nb0[NTL], nb1[NTL];
#pragma unroll
      for(int q = 0; q < NTL; q++) { const double* bp = P + rowc[q] + kq*ldp; nb0[q] = bp[kn*ldp]; nb1[q] = bp[(kn + 4)*ldp]; }
      __builtin_amdgcn_sched_barrier(0);
      const double va0 = (mm < nb) ? a0 : 0.0, va1 = (mm < nb) ? a1 : 0.0;
#pragma unroll
      for(int q = 0; q < NTL; q++) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(va0, b0[q], acc[q], 0, 0, PF_NEG_A);
#pragma unroll
      for(int q = 0; q < NTL; q++) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(va1, b1[q], acc[q], 0, 0, PF_NEG_A);
      __builtin_amdgcn_sched_barrier(0);
      a0 = na0; a1 = na1;
#pragma unroll
      for(int q = 0; q < NTL; q++) { b0[q] = nb0[q]; b1[q] = nb1[q]; }
    }
  }
  DLG_PF_WLOG(J, t1, 1);
  // times L_JJ^-T: Y'[i][j] = sum_c W[i][c] S[r0 + j][kb + c]
  pf_wait(&S.wdone, J + 1);
  DLG_PF_WLOG(J, t1, 2);
  const double* Wr = S.W[J & 1] + mm*PF_B16_WS + kq;
  const double w0 = Wr[0], w1 = Wr[4], w2 = Wr[8], w3 = Wr[12];
  dlg_pf_v4d y0[NTL], y1[NTL];
#pragma unroll
  for(int q = 0; q < NTL; q++) { y0[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(w0, acc[q][0], zero4, 0, 0, 0); y1[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(w2, acc[q][2], zero4, 0, 0, 0); }
#pragma unroll
  for(int q = 0; q < NTL; q++) { y0[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(w1, acc[q][1], y0[q], 0, 0, 0); y1[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(w3, acc[q][3], y1[q], 0, 0, 0); }
#pragma unroll
  for(int r = 0; r < 4; r++)
  {
    const int col = kq + 4*r;
#pragma unroll
    for(int q = 0; q < NTL; q++)
      if(r0[q] + mm < nrows && col < nb && !(q == 0 && own_first && r0[q] + mm < kb + nb)) P[(r0[q] + mm) + (kb + col)*ldp] = y0[q][r] + y1[q][r];
  }
  pf_wave_sync();
  if(lane == 0)
  {
#pragma unroll
    for(int q = 0; q < NTL; q++) if(!(q == 0 && own_first)) __hip_atomic_store(&S.tdone[t1 + q*dt], J + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  DLG_PF_WLOG(J, t1, 3);
}
// a wave's tiles t, t + dt, ... < tend of block J, three / two / one at a time
__device__ __forceinline__ void pf_b16_tile_run(double* P, int ldp, int nrows, pf_b16_lds& S, int J, int kb, int nb, int t, int dt, int tend,
                                                int lane, int mm, int kq, bool foreign, bool own_first = false)
{
  while(t < tend)
  {
#ifndef DLG_PF_NO_PAIRS
    if(t + 2*dt < tend) { pf_b16_normal_tiles<3>(P, ldp, nrows, S, J, kb, nb, t, dt, lane, mm, kq, foreign, own_first); t += 3*dt; own_first = false; continue; }
    if(t + dt < tend)   { pf_b16_normal_tiles<2>(P, ldp, nrows, S, J, kb, nb, t, dt, lane, mm, kq, foreign, own_first); t += 2*dt; own_first = false; continue; }
#endif
    pf_b16_normal_tiles<1>(P, ldp, nrows, S, J, kb, nb, t, dt, lane, mm, kq, foreign, own_first); t += dt; own_first = false;
  }
}
template <int NT>
__device__ __forceinline__ void panel_factor_b16(double* P, int ldp, int nrows, int w_all, int tid,
                                                 int* __restrict__ info, int col0)
{
  static_assert(NT >= 256, "panel_factor_b16 needs at least 4 waves");
  // (a short last block goes to the vector pipe: the sweep below sees a panel of w columns whose other columns' rows are rows below)
#ifdef DLG_PF_NO_VFIN
  const int nfin = 0;
#else
  const int nfin = ((w_all & 15) != 0 && (w_all & 15) <= PF_VFIN_MAX && w_all > 16 && (nrows - (w_all & ~15) + 63)/64 <= NT/64) ? (w_all & 15) : 0;
#endif
  const int w = w_all - nfin;
  // (with 8 waves, wave 4 shares its SIMD with wave 0: an fp64 MFMA holds the SIMD for its 64 clocks, so that wave
  // takes no tiles -- wave 0's chain has SIMD 0 to itself)
  constexpr int NW = NT/64, NH = (NW == 8) ? 6 : NW - 1;
  __shared__ pf_b16_lds S;
  const int lane = tid & 63, wv = tid >> 6;
  const int mm = lane & 15, kq = lane >> 4;
  const int ntr = (nrows + 15) >> 4, nblk = (w + 15) >> 4;
  if(tid == 0) { S.wdone = 0; S.adone = 0; }
  if(tid < PF_B16_MAXT) S.tdone[tid] = 0;
  __syncthreads();
  DLG_PF_DECL
  const dlg_pf_v4d zero4 = {0.0, 0.0, 0.0, 0.0};
  if(wv == 0)
  {
    int bad = 0x7fffffff;
    dlg_pf_v4d U = pf_b16_diag_tile(P, ldp, 0, min(16, w), mm, kq);
    int tv_ahead = 0;
    for(int J = 0; J < nblk; J++)
    {
      const int kb = 16*J, nb = min(16, w - kb);
      dlg_pf_v4d G;
#pragma unroll
      for(int r = 0; r < 4; r++) G[r] = (kq + 4*r == mm) ? 1.0 : 0.0;
      if(J >= 2)
      {
        // the buffer of the inverse published two blocks ago is written again: every tile must be done with it (the
        // words were fetched at the end of the last block; only a tile that was late then is polled)
        if(!__all(tv_ahead >= J - 1))
          for(int spins = 0; spins < (1 << 20); spins++)          // (bounded: a wave that never reports shows as a wrong factor)
          {
            const bool mine = lane < ntr && lane > J - 2;
            const int v = mine ? __hip_atomic_load(&S.tdone[mine ? lane : 0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) : 0x7fffffff;
            if(__all(v >= J - 1)) break;
            __builtin_amdgcn_s_sleep(1);
          }
      }
      DLG_PF_STAMP(0);
      DLG_PF_EVT(J, 0);
      double* Wb = S.W[J & 1];
#pragma unroll
      for(int s = 0; s < 4; s++)
      {
        if(4*s >= nb)
        {
          Wb[(4*s + kq)*PF_B16_WS + mm] = (4*s + kq == mm) ? 1.0 : 0.0;
          continue;
        }
        const double src = U[s];
        // the 4 x 4 pivot block: T[4s + q][4s + k] sits in lane (j = 4s + q, k) = 16 k + 4 s + q
        const double a00 = pf_readlane64(src, 4*s), a10 = pf_readlane64(src, 4*s + 1), a20 = pf_readlane64(src, 4*s + 2), a30 = pf_readlane64(src, 4*s + 3);
        const double a11 = pf_readlane64(src, 16 + 4*s + 1), a21 = pf_readlane64(src, 16 + 4*s + 2), a31 = pf_readlane64(src, 16 + 4*s + 3);
        const double a22 = pf_readlane64(src, 32 + 4*s + 2), a32 = pf_readlane64(src, 32 + 4*s + 3), a33 = pf_readlane64(src, 48 + 4*s + 3);
        double d0 = a00; if(!(d0 > 0.0)) { bad = min(bad, kb + 4*s); d0 = 1.0; }
        const double i0 = pf_rsqrt3(d0);
        const double l10 = a10*i0, l20 = a20*i0, l30 = a30*i0;
        double d1 = a11 - l10*l10; if(!(d1 > 0.0)) { bad = min(bad, kb + 4*s + 1); d1 = 1.0; }
        const double i1 = pf_rsqrt3(d1);
        const double l21 = (a21 - l20*l10)*i1, l31 = (a31 - l30*l10)*i1;
        double d2 = a22 - l20*l20 - l21*l21; if(!(d2 > 0.0)) { bad = min(bad, kb + 4*s + 2); d2 = 1.0; }
        const double i2 = pf_rsqrt3(d2);
        const double l32 = (a32 - l30*l20 - l31*l21)*i2;
        double d3 = a33 - l30*l30 - l31*l31 - l32*l32; if(!(d3 > 0.0)) { bad = min(bad, kb + 4*s + 3); d3 = 1.0; }
        const double i3 = pf_rsqrt3(d3);
        // M = D^-1 (lower)
        const double m10 = -(l10*i0)*i1, m21 = -(l21*i1)*i2, m32 = -(l32*i2)*i3;
        const double m20 = -(l20*i0 + l21*m10)*i2, m31 = -(l31*i1 + l32*m21)*i3;
        const double m30 = -(l30*i0 + l31*m10 + l32*m20)*i3;
        // A operand: lane (i, k) = M[i][k] (rows 4 .. 15 of the operand are zero)
        double aop = 0.0;       // (the values that come out last go in last)
        aop = (lane == 0) ? i0 : aop;   aop = (lane == 1) ? m10 : aop;  aop = (lane == 17) ? i1 : aop;
        aop = (lane == 2) ? m20 : aop;  aop = (lane == 18) ? m21 : aop; aop = (lane == 34) ? i2 : aop;
        aop = (lane == 3) ? m30 : aop;  aop = (lane == 19) ? m31 : aop; aop = (lane == 35) ? m32 : aop; aop = (lane == 51) ? i3 : aop;
        const dlg_pf_v4d Xv = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, src, zero4, 0, 0, 0);
        const double X0 = Xv[0];                    // lane (j, k) = L[kb + j][kb + 4s + k]
        if(s < 3) U = __builtin_amdgcn_mfma_f64_16x16x4f64(X0, X0, U, 0, 0, PF_NEG_A);
        if(mm >= 4*s + kq && mm < nb && 4*s + kq < nb) P[(kb + mm) + (kb + 4*s + kq)*ldp] = X0;
        // the rows 4s .. 4s + 3 of L_JJ^-1 from the identity tile
        const dlg_pf_v4d Wv = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, G[s], zero4, 0, 0, 0);
        const double W0 = Wv[0];                    // lane (j, k) = W[4s + k][j]
        Wb[(4*s + kq)*PF_B16_WS + mm] = W0;
        if(s < 3) G = __builtin_amdgcn_mfma_f64_16x16x4f64(X0, W0, G, 0, 0, PF_NEG_A);
      }
      pf_wave_sync();
      if(lane == 0) __hip_atomic_store(&S.wdone, J + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      DLG_PF_STAMP(1);
      DLG_PF_EVT(J, 1);
      DLG_PF_WLOG(J, 0, 5);
      if(J + 1 < nblk)
      {
        // the next diagonal tile: its rows in the columns of this block times L_JJ^-T, here, in registers -- the
        // product IS the operand of the tile's own update
        const int r1 = kb + 16, nb1 = min(16, w - r1);
        const double* Wr = Wb + mm*PF_B16_WS + kq;
        const double w0 = Wr[0], w1 = Wr[4], w2 = Wr[8], w3 = Wr[12];
        pf_wait(&S.adone, J + 1);
        DLG_PF_STAMP(3);
        DLG_PF_EVT(J, 2);
        const double s0 = S.A[lane], s1 = S.A[64 + lane], s2 = S.A[128 + lane], s3 = S.A[192 + lane];
#pragma unroll
        for(int r = 0; r < 4; r++) U[r] = S.E[64*r + lane];
        // (an fp64 MFMA is issue-bound, 72 clocks apiece whatever the dependences: one chain each)
        dlg_pf_v4d y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(w0, s0, zero4, 0, 0, 0);
        y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(w1, s1, y0, 0, 0, 0);
        y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(w2, s2, y0, 0, 0, 0);
        y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(w3, s3, y0, 0, 0, 0);
#pragma unroll
        for(int r = 0; r < 4; r++)
        {
          const double yr = y0[r];                // lane (n, kq) = L[r1 + n][kb + 4 r + kq]
          if(r1 + mm < nrows) P[(r1 + mm) + (kb + kq + 4*r)*ldp] = yr;
          const double ym = (mm < nb1) ? yr : 0.0;          // (rows below the top block are not part of the next diagonal tile)
          U = __builtin_amdgcn_mfma_f64_16x16x4f64(ym, ym, U, 0, 0, PF_NEG_A);
        }
        pf_wave_sync();
        if(lane == 0) __hip_atomic_store(&S.tdone[J + 1], J + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        DLG_PF_EVT(J, 3);
        DLG_PF_WLOG(J, J + 1, 6);
        // (for the check in front of the next block's first store into the other buffer: tiles > J - 1 at >= J)
        tv_ahead = (lane < ntr && lane > J - 1) ? __hip_atomic_load(&S.tdone[lane], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) : 0x7fffffff;
      }
      DLG_PF_STAMP(2);
    }
    if(bad != 0x7fffffff && lane == 0) atomicMin(info, col0 + bad);
  }
  else if(!(NW == 8 && wv == 4))
  {
    const int h = (NW == 8 && wv > 4) ? wv - 2 : wv - 1;
    for(int J = 0; J < nblk; J++)
    {
      const int kb = 16*J, nb = min(16, w - kb);
      // The wave's tiles of this block in increasing order.  The special ones -- the block's own tile below a short top block,
      // the next diagonal tile (handed to wave 0 half-finished: it is waited for) -- one at a time, as before; the others two
      // at a time (pf_b16_normal_tiles)
      int t = (h + 1) % NH;
      while(t < ntr && t < J) t += NH;
      for(; t < ntr; )
      {
        const bool own = t == J && nb < 16 && nrows > kb + nb;
        const bool next = t == J + 1 && J + 1 < nblk;        // the next diagonal tile: wave 0 finishes it
        if(t == J && !own) { t += NH; continue; }
        if(!next)
        {
          pf_b16_tile_run(P, ldp, nrows, S, J, kb, nb, t, NH, ntr, lane, mm, kq, false, own);
          break;
        }
        const int r0 = 16*t, rowc = min(r0 + mm, nrows - 1);
        const int nb1 = min(16, w - r0);
        if(next) { DLG_PF_EVT(J, 4); }
        DLG_PF_WLOG(J, t, 0);
        // the tile TRANSPOSED: register r, lane (n, kq) = S[r0 + n][kb + kq + 4 r]
        dlg_pf_v4d acc, E = zero4;
#pragma unroll
        for(int r = 0; r < 4; r++)
        {
          const int col = kq + 4*r;
          const double v = P[rowc + (kb + (col < nb ? col : 0))*ldp];
          acc[r] = (col < nb) ? v : 0.0;
        }
        if(next) E = pf_b16_diag_tile(P, ldp, r0, nb1, mm, kq);
        if(J > 0)
        {
          pf_wait(&S.tdone[J], J);
          if(next) { DLG_PF_EVT(J, 5); }
          const double* ap = P + kb + (mm < nb ? mm : 0) + kq*ldp;        // rows of the diagonal tile
          const double* bp = P + rowc + kq*ldp;                           // rows of this tile
          double a0 = ap[0], b0 = bp[0], a1 = ap[4*ldp], b1 = bp[4*ldp];
          for(int k0 = 0; k0 < kb; k0 += 8)
          {
            const int kn = (k0 + 8 < kb) ? k0 + 8 : k0;
            const double na0 = ap[kn*ldp], nb0 = bp[kn*ldp], na1 = ap[(kn + 4)*ldp], nb1v = bp[(kn + 4)*ldp];
            __builtin_amdgcn_sched_barrier(0);
            const double va0 = (mm < nb) ? a0 : 0.0, va1 = (mm < nb) ? a1 : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va0, b0, acc, 0, 0, PF_NEG_A);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va1, b1, acc, 0, 0, PF_NEG_A);
            if(next)
            {
              const double e0 = (mm < nb1) ? b0 : 0.0, e1 = (mm < nb1) ? b1 : 0.0;
              E = __builtin_amdgcn_mfma_f64_16x16x4f64(e0, e0, E, 0, 0, PF_NEG_A);
              E = __builtin_amdgcn_mfma_f64_16x16x4f64(e1, e1, E, 0, 0, PF_NEG_A);
            }
            __builtin_amdgcn_sched_barrier(0);
            a0 = na0; b0 = nb0; a1 = na1; b1 = nb1v;
          }
        }
        DLG_PF_WLOG(J, t, 1);
        if(next)
        {
#pragma unroll
          for(int r = 0; r < 4; r++) { S.A[64*r + lane] = acc[r]; S.E[64*r + lane] = E[r]; }
          pf_wave_sync();
          if(lane == 0) __hip_atomic_store(&S.adone, J + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          DLG_PF_EVT(J, 6);
          DLG_PF_WLOG(J, t, 4);
          t += NH;
          continue;
        }
        // times L_JJ^-T: Y'[i][j] = sum_c W[i][c] S[r0 + j][kb + c]
        pf_wait(&S.wdone, J + 1);
        DLG_PF_WLOG(J, t, 2);
        const double* Wr = S.W[J & 1] + mm*PF_B16_WS + kq;
        const double w0 = Wr[0], w1 = Wr[4], w2 = Wr[8], w3 = Wr[12];
        dlg_pf_v4d y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(w0, acc[0], zero4, 0, 0, 0);
        dlg_pf_v4d y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(w2, acc[2], zero4, 0, 0, 0);
        y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(w1, acc[1], y0, 0, 0, 0);
        y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(w3, acc[3], y1, 0, 0, 0);
#pragma unroll
        for(int r = 0; r < 4; r++)
        {
          const int col = kq + 4*r;
          if(r0 + mm < nrows && col < nb && (!own || r0 + mm >= kb + nb)) P[(r0 + mm) + (kb + col)*ldp] = y0[r] + y1[r];
        }
        pf_wave_sync();
        if(lane == 0 && !own) __hip_atomic_store(&S.tdone[t], J + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        DLG_PF_WLOG(J, t, 3);
        t += NH;
      }
    }
  }
  __syncthreads();
  DLG_PF_STAMP(5);
  switch(nfin)
  {
    case 1: pf_b16_vector_finish<NT, 1>(P, ldp, nrows, w, tid, &S.W[0][0], info, col0); break;
    case 2: pf_b16_vector_finish<NT, 2>(P, ldp, nrows, w, tid, &S.W[0][0], info, col0); break;
    case 3: pf_b16_vector_finish<NT, 3>(P, ldp, nrows, w, tid, &S.W[0][0], info, col0); break;
    case 4: pf_b16_vector_finish<NT, 4>(P, ldp, nrows, w, tid, &S.W[0][0], info, col0); break;
    default: break;
  }
  DLG_PF_DONE
}
template <int NT, bool ALIGNED16, bool MFMA_SWEEP = false>
__device__ __forceinline__ void panel_factor(double* P, int ldp, int nrows, int w, int tid,
                                             int* __restrict__ info, int col0)
{
  DLG_PF_DECL
  for(int kb = 0; kb < w; kb += 8)
  {
    const int nb = (w - kb < 8) ? w - kb : 8;
    // MFMA sweep: at the start of a 16-column block both 8-column halves are brought up to date
    // against all columns before the block; the second half then only lacks the first half
    constexpr int ks = 0;
    if(MFMA_SWEEP)
    {
      if(kb > 0)
      {
        if((kb & 15) == 0) panel_mfma_sweep<NT>(P, ldp, nrows, kb, min(16, w - kb), 0, tid);
        else               panel_mfma_sweep<NT>(P, ldp, nrows, kb, nb, kb - 8, tid);
      }
    }
    else
    for(int r = kb + tid; r < nrows; r += NT)
    {
      double x[8];
#pragma unroll
      for(int c = 0; c < 8; c++) x[c] = (c < nb) ? P[r + (kb + c)*ldp] : 0.0;
#pragma unroll 4
      for(int k = ks; k < kb; k++)
      {
        const double a = P[r + k*ldp];
        const double* bp = P + kb + k*ldp;
        if(ALIGNED16)
        {
          const double2 b0 = *reinterpret_cast<const double2*>(bp);
          const double2 b1 = *reinterpret_cast<const double2*>(bp + 2);
          const double2 b2 = *reinterpret_cast<const double2*>(bp + 4);
          const double2 b3 = *reinterpret_cast<const double2*>(bp + 6);
          x[0] -= a*b0.x; x[1] -= a*b0.y; x[2] -= a*b1.x; x[3] -= a*b1.y;
          x[4] -= a*b2.x; x[5] -= a*b2.y; x[6] -= a*b3.x; x[7] -= a*b3.y;
        }
        else
        {
#pragma unroll
          for(int c = 0; c < 8; c++) if(c < nb) x[c] -= a*bp[c];
        }
      }
#pragma unroll
      for(int c = 0; c < 8; c++) if(c < nb) P[r + (kb + c)*ldp] = x[c];
    }
    if(!MFMA_SWEEP) { DLG_PF_STAMP(0); }
    __syncthreads();
    DLG_PF_STAMP(1);
    double D[8][8];
    if(ALIGNED16)
    {
      // column q, rows q..7 of the block: pairs of rows are 16-byte aligned (kb, ldp even)
#pragma unroll
      for(int q = 0; q < 8; q++)
#pragma unroll
        for(int c2 = (q & ~1); c2 < 8; c2 += 2)
        {
          const double2 v = *reinterpret_cast<const double2*>(&P[(kb + c2) + (kb + q)*ldp]);
          D[c2][q] = v.x; D[c2 + 1][q] = v.y;
        }
      if(nb < 8)
      {
#pragma unroll
        for(int c = 0; c < 8; c++)
#pragma unroll
          for(int q = 0; q <= c; q++) if(c >= nb) D[c][q] = (c == q) ? 1.0 : 0.0;
      }
    }
    else
    {
#pragma unroll
      for(int c = 0; c < 8; c++)
#pragma unroll
        for(int q = 0; q <= c; q++)
          D[c][q] = (c < nb) ? P[(kb + c) + (kb + q)*ldp] : ((c == q) ? 1.0 : 0.0);
    }
    bool bad = false; int badcol = 0;
    double Dinv[8];
    // waves without rows below the block only keep the barriers (wave 0 writes the block back)
    const bool need = (tid < 64) || (kb + nb + (tid & ~63) < nrows);
    if(need)
    // right-looking: once column c is scaled the trailing block is updated at once, so the next
    // pivot only waits for one multiply-add after the reciprocal square root
#pragma unroll
    for(int c = 0; c < 8; c++)
    {
      double d = D[c][c];
      if(!(d > 0.0)) { if(!bad) { bad = true; badcol = c; } d = 1.0; }
      const double inv = dlg_rsqrt(d);
      Dinv[c] = inv;
#pragma unroll
      for(int i = c + 1; i < 8; i++) D[i][c] *= inv;
#pragma unroll
      for(int j = c + 1; j < 8; j++)
#pragma unroll
        for(int i = j; i < 8; i++) D[i][j] -= D[i][c]*D[j][c];
      D[c][c] = d*inv;
    }
    if(bad && tid == 0) atomicMin(info, col0 + kb + badcol);
    DLG_PF_STAMP(2);
    __syncthreads();
    DLG_PF_STAMP(3);
    // the factored block goes back through the first 64 threads: thread (c, q) keeps element
    // (c, q) (selected with compile-time indices: D lives in registers)
    if(tid < 64)
    {
      const int c = tid >> 3, q = tid & 7;
      double v = 0.0;
#pragma unroll
      for(int cc = 0; cc < 8; cc++)
#pragma unroll
        for(int qq = 0; qq <= cc; qq++) v = (cc == c && qq == q) ? D[cc][qq] : v;
      if(q <= c && c < nb) P[(kb + c) + (kb + q)*ldp] = v;
    }
    for(int r = kb + nb + tid; r < nrows; r += NT)
    {
      double x[8];
#pragma unroll
      for(int c = 0; c < 8; c++) x[c] = (c < nb) ? P[r + (kb + c)*ldp] : 0.0;
#pragma unroll
      for(int c = 0; c < 8; c++)
      {
        double v = x[c];
#pragma unroll
        for(int q = 0; q < c; q++) v -= x[q]*D[c][q];
        x[c] = v*Dinv[c];
      }
#pragma unroll
      for(int c = 0; c < 8; c++) if(c < nb) P[r + (kb + c)*ldp] = x[c];
    }
    DLG_PF_STAMP(4);
    __syncthreads();
    DLG_PF_STAMP(5);
  }
  DLG_PF_DONE
}

// Panel whose w x w top block is block diagonal (a supernode made of independent
// sibling leaves, e.g. the points seen by one set of cameras): member m owns columns
// [mcol[m], mcol[m+1]) (at most 8).  The members do not couple, so there is no sweep:
//   A: thread m factors the diagonal block of member m in registers (<= 8x8);
//   B: after one barrier every below row solves against each member's block.
// member block of NB columns (compile-time width: no predicated 8-wide code for 3-wide points)
template <int NB>
__device__ __forceinline__ int bd_factor_member(double* P, int ldp, int c0, double* rdiag)
{
  double D[NB][NB];
#pragma unroll
  for(int c = 0; c < NB; c++)
#pragma unroll
    for(int q = 0; q <= c; q++) D[c][q] = P[(c0 + c) + (c0 + q)*ldp];
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
    for(int q = 0; q <= c; q++) P[(c0 + c) + (c0 + q)*ldp] = D[c][q];
  return badcol;
}
template <int NB>
__device__ __forceinline__ void bd_solve_row(double* P, int ldp, int r, int c0, const double* rdiag)
{
  double x[NB], Lm[NB][NB], rd[NB];
#pragma unroll
  for(int c = 0; c < NB; c++)
  {
    x[c] = P[r + (c0 + c)*ldp]; rd[c] = rdiag[c0 + c];
#pragma unroll
    for(int q = 0; q < c; q++) Lm[c][q] = P[(c0 + c) + (c0 + q)*ldp];
  }
#pragma unroll
  for(int c = 0; c < NB; c++)
  {
    double v = x[c];
#pragma unroll
    for(int q = 0; q < c; q++) v -= x[q]*Lm[c][q];
    x[c] = v*rd[c];
  }
#pragma unroll
  for(int c = 0; c < NB; c++) P[r + (c0 + c)*ldp] = x[c];
}
template <int NT>
__device__ __forceinline__ void panel_factor_blockdiag(double* P, int ldp, int nrows, int w, int tid,
                                                       const int* __restrict__ mcol_g, int nmem,
                                                       int* __restrict__ info, int col0,
                                                       int* mcol, double* rdiag)
{
  // member boundaries and reciprocal pivots live in LDS (mcol[260], rdiag[256], the caller's):
  // phase B walks them for every row
  for(int m = tid; m <= nmem; m += NT) mcol[m] = (m < nmem) ? mcol_g[m] : w;
  __syncthreads();
  for(int m = tid; m < nmem; m += NT)
  {
    const int c0 = mcol[m], nb = mcol[m+1] - c0;
    int badcol;
    switch(nb)
    {
      case 1: badcol = bd_factor_member<1>(P, ldp, c0, rdiag); break;
      case 2: badcol = bd_factor_member<2>(P, ldp, c0, rdiag); break;
      case 3: badcol = bd_factor_member<3>(P, ldp, c0, rdiag); break;
      case 4: badcol = bd_factor_member<4>(P, ldp, c0, rdiag); break;
      case 5: badcol = bd_factor_member<5>(P, ldp, c0, rdiag); break;
      case 6: badcol = bd_factor_member<6>(P, ldp, c0, rdiag); break;
      case 7: badcol = bd_factor_member<7>(P, ldp, c0, rdiag); break;
      default: badcol = bd_factor_member<8>(P, ldp, c0, rdiag); break;
    }
    if(badcol >= 0) atomicMin(info, col0 + c0 + badcol);
  }
  __syncthreads();
  for(int r = w + tid; r < nrows; r += NT)
  {
    for(int m = 0; m < nmem; m++)
    {
      const int c0 = mcol[m], nb = mcol[m+1] - c0;        // uniform over the workgroup
      switch(nb)
      {
        case 1: bd_solve_row<1>(P, ldp, r, c0, rdiag); break;
        case 2: bd_solve_row<2>(P, ldp, r, c0, rdiag); break;
        case 3: bd_solve_row<3>(P, ldp, r, c0, rdiag); break;
        case 4: bd_solve_row<4>(P, ldp, r, c0, rdiag); break;
        case 5: bd_solve_row<5>(P, ldp, r, c0, rdiag); break;
        case 6: bd_solve_row<6>(P, ldp, r, c0, rdiag); break;
        case 7: bd_solve_row<7>(P, ldp, r, c0, rdiag); break;
        default: bd_solve_row<8>(P, ldp, r, c0, rdiag); break;
      }
    }
  }
  __syncthreads();
}

// ---- compact variant for unsliced block-diagonal panels (the merged point leaves) -------------
// The top block of such a panel is block diagonal, i.e. almost all zeros: it is never staged.
// LDS holds the rows below it only (Pb[i + j*ldp] for i >= w: Pb = base - w) plus the factored
// member blocks (Dg[(c0 + c)*8 + q]); a member's thread reads its little block straight from the
// panel in HBM (G, leading dimension ldg), factors it in registers and writes it back.
// (DS: doubles kept per column of the factored member blocks in Dg: 8, or 4 where no member is wider than 4)
template <int NB, int DS = 8>
__device__ __forceinline__ int bd_factor_member_g(double* __restrict__ G, int ldg, int c0, double* rdiag,
                                                  double* __restrict__ Dg)
{
  double D[NB][NB];
#pragma unroll
  for(int c = 0; c < NB; c++)
#pragma unroll
    for(int q = 0; q <= c; q++) D[c][q] = G[(c0 + c) + (size_t)(c0 + q)*ldg];
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
    for(int q = 0; q <= c; q++) { G[(c0 + c) + (size_t)(c0 + q)*ldg] = D[c][q]; if(q < DS) Dg[(c0 + c)*DS + q] = D[c][q]; }
  return badcol;
}
template <int NB, int DS = 8>
__device__ __forceinline__ void bd_solve_row_c(double* Pb, int ldp, int r, int c0, const double* rdiag,
                                               const double* __restrict__ Dg)
{
  double x[NB], Lm[NB][NB], rd[NB];
#pragma unroll
  for(int c = 0; c < NB; c++)
  {
    x[c] = Pb[r + (c0 + c)*ldp]; rd[c] = rdiag[c0 + c];
#pragma unroll
    for(int q = 0; q < c; q++) Lm[c][q] = Dg[(c0 + c)*DS + q];
  }
#pragma unroll
  for(int c = 0; c < NB; c++)
  {
    double v = x[c];
#pragma unroll
    for(int q = 0; q < c; q++) v -= x[q]*Lm[c][q];
    x[c] = v*rd[c];
  }
#pragma unroll
  for(int c = 0; c < NB; c++) Pb[r + (c0 + c)*ldp] = x[c];
}
// members first (their global loads are on their way while the caller copies the rows below into
// LDS: call bd_compact_members before that copy's barrier), then the rows
template <int NT, int DS = 8>
__device__ __forceinline__ void bd_compact_members(double* __restrict__ G, int ldg, int w, int tid,
                                                   const int* __restrict__ mcol_g, int nmem, int bdw,
                                                   int* mcol, double* rdiag, double* __restrict__ Dg,
                                                   int* __restrict__ info, int col0)
{
  for(int m = tid; m <= nmem; m += NT) mcol[m] = (m < nmem) ? (bdw > 0 ? m*bdw : mcol_g[m]) : w;
  for(int m = tid; m < nmem; m += NT)
  {
    const int c0 = bdw > 0 ? m*bdw : mcol_g[m];
    const int nb = (m + 1 < nmem ? (bdw > 0 ? (m + 1)*bdw : mcol_g[m + 1]) : w) - c0;
    int badcol;
    switch(nb)
    {
      case 1: badcol = bd_factor_member_g<1, DS>(G, ldg, c0, rdiag, Dg); break;
      case 2: badcol = bd_factor_member_g<2, DS>(G, ldg, c0, rdiag, Dg); break;
      case 3: badcol = bd_factor_member_g<3, DS>(G, ldg, c0, rdiag, Dg); break;
      case 4: badcol = bd_factor_member_g<4, DS>(G, ldg, c0, rdiag, Dg); break;
      case 5: badcol = bd_factor_member_g<5, DS>(G, ldg, c0, rdiag, Dg); break;
      case 6: badcol = bd_factor_member_g<6, DS>(G, ldg, c0, rdiag, Dg); break;
      case 7: badcol = bd_factor_member_g<7, DS>(G, ldg, c0, rdiag, Dg); break;
      default: badcol = bd_factor_member_g<8, DS>(G, ldg, c0, rdiag, Dg); break;
    }
    if(badcol >= 0) atomicMin(info, col0 + c0 + badcol);
  }
}
template <int NT, int NB, int DS>
__device__ __forceinline__ void bd_compact_rows_flat(double* Pb, int ldp, int nr, int w, int tid, int nmem,
                                                     const double* rdiag, const double* __restrict__ Dg)
{
  // thread = (row, member) pairs tid, tid + NT, ... of the nr x nmem of them, member-major (a wave's lanes on consecutive
  // rows of a column): one division, then steps of NT = q nr + rem
  const int q = NT / nr, rem = NT - q*nr;
  int m = tid / nr, r = tid - m*nr;
  while(m < nmem)
  {
    bd_solve_row_c<NB, DS>(Pb, ldp, w + r, m*NB, rdiag, Dg);
    r += rem; m += q;
    if(r >= nr) { r -= nr; m++; }
  }
}
// bdw > 0 with w == nmem * bdw: every member has bdw columns (merged point leaves) -- the (row, member) pairs are
// independent and are dealt to ALL threads (thread = row left 73 of a leaf workgroup's 256 threads with 18 members each
// in turn: 6.1 of its 26.8 us, tools/prof_factor.sh); the same arithmetic per pair, the same bits
template <int NT, int DS = 8>
__device__ __forceinline__ void bd_compact_rows(double* Pb, int ldp, int nrows, int w, int tid, int nmem,
                                                const int* mcol, const double* rdiag, const double* __restrict__ Dg, int bdw = 0)
{
#ifndef DLG_BD_ROWS_BY_THREAD
  if(bdw > 0 && bdw <= 4 && w == nmem*bdw && nrows > w)
  {
    const int nr = nrows - w;
    switch(bdw)
    {
      case 1: bd_compact_rows_flat<NT, 1, DS>(Pb, ldp, nr, w, tid, nmem, rdiag, Dg); break;
      case 2: bd_compact_rows_flat<NT, 2, DS>(Pb, ldp, nr, w, tid, nmem, rdiag, Dg); break;
      case 3: bd_compact_rows_flat<NT, 3, DS>(Pb, ldp, nr, w, tid, nmem, rdiag, Dg); break;
      default: bd_compact_rows_flat<NT, 4, DS>(Pb, ldp, nr, w, tid, nmem, rdiag, Dg); break;
    }
    __syncthreads();
    return;
  }
#endif
  for(int r = w + tid; r < nrows; r += NT)
  {
    for(int m = 0; m < nmem; m++)
    {
      const int c0 = mcol[m], nb = mcol[m+1] - c0;        // uniform over the workgroup
      switch(nb)
      {
        case 1: bd_solve_row_c<1, DS>(Pb, ldp, r, c0, rdiag, Dg); break;
        case 2: bd_solve_row_c<2, DS>(Pb, ldp, r, c0, rdiag, Dg); break;
        case 3: bd_solve_row_c<3, DS>(Pb, ldp, r, c0, rdiag, Dg); break;
        case 4: bd_solve_row_c<4, DS>(Pb, ldp, r, c0, rdiag, Dg); break;
        case 5: bd_solve_row_c<5, DS>(Pb, ldp, r, c0, rdiag, Dg); break;
        case 6: bd_solve_row_c<6, DS>(Pb, ldp, r, c0, rdiag, Dg); break;
        case 7: bd_solve_row_c<7, DS>(Pb, ldp, r, c0, rdiag, Dg); break;
        default: bd_solve_row_c<8, DS>(Pb, ldp, r, c0, rdiag, Dg); break;
      }
    }
  }
  __syncthreads();
}

