// sparse_factor.hip -- K5: level-scheduled supernodal Cholesky on gfx950,
// replaces cholmod_factorize[_p] (dogleg.c:659-664).  Per elimination-tree level: panel
// factorisation (k_factor_level), then the updates of the ancestors by the level's panels.
#include "sparse_internal.h"
#include "panel_factor.h"

namespace {
typedef double dlg_v4d __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ K5 ------
// factor one supernode panel per workgroup in LDS (column-major, even leading dimension).
// One workgroup per work item = (supernode, slice [r0,r1) of its below rows): the LDS panel
// holds the w x w top block plus the slice (up to ~160 KB); slices of one supernode factor the
// top block redundantly (identical arithmetic), slice 0 publishes it (top_scr, k_copy_top).
// The arithmetic is in panel_factor.h:
//   * regular panels, >= 256 threads: panel_factor_mfma -- per 8 columns wave 0 brings the
//     diagonal row tile up to date on the matrix cores, factors the 8x8 block in registers and
//     publishes it while the other waves update the remaining row tiles; barrier; every thread
//     solves its row; barrier;
//   * 128 threads: panel_factor (same steps, all waves factor the block redundantly);
//   * sibling-merged leaves (block-diagonal top): panel_factor_blockdiag, no sweep at all.
// The panel is copied in and out thread-per-row, 16 columns in flight.
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
                                                     int* __restrict__ info,
                                                     const int64_t* __restrict__ u_off,
                                                     double* __restrict__ uscr, int fuse_syrk)
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
  // two-phase update, phase 1 fused: this workgroup holds all the below rows of the (unsliced)
  // panel in LDS, so U = B B' (B = rows w.. of the factored panel) comes straight out of it:
  // lower 16x16 tiles round robin over the waves, both MFMA operands read from the panel
  if(fuse_syrk && top < 0)
  {
    const int mb = nloc - w, T = (mb + 15) >> 4, ntiles = T*(T + 1)/2;
    double* U = uscr + u_off[s];
    const int lane = tid & 63, wv = tid >> 6;
    const int jn = lane & 15, kq = lane >> 4;
    for(int idx = wv; idx < ntiles; idx += NT/64)
    {
      int rem = idx, tj = 0;
      while(rem >= T - tj) { rem -= T - tj; tj++; }
      const int ti = tj + rem;
      const int ra = w + min(16*ti + jn, mb - 1), rb = w + min(16*tj + jn, mb - 1);
      const bool va = 16*ti + jn < mb, vb = 16*tj + jn < mb;
      dlg_v4d c4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 2
      for(int kk = 0; kk < w; kk += 4)
      {
        const int k = kk + kq;
        const bool kok = k < w;
        const int kc = kok ? k : w - 1;
        const double a = P[ra + kc*ldp], bv = P[rb + kc*ldp];
        c4 = __builtin_amdgcn_mfma_f64_16x16x4f64((kok && va) ? a : 0.0, (kok && vb) ? bv : 0.0, c4, 0, 0, 0);
      }
      const int j = 16*tj + jn;
#pragma unroll
      for(int r = 0; r < 4; r++)
      {
        const int i = 16*ti + kq + 4*r;
        if(i < mb && j <= i) U[i + (size_t)j*mb] = c4[r];
      }
    }
  }
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
constexpr int GATHER_FLIGHT = 4;        // sub-tasks whose loads are in flight together
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
              if(c < nc) v[g][c] = Uv[g][ig[g] + (size_t)min(c, cmax)*mbv[g]];   // entries above the diagonal are never written
#pragma unroll
          for(int g = 0; g < GATHER_FLIGHT; g++)
#pragma unroll
            for(int c = 0; c < 8; c++)
              if(c < nc) { if(i < mv[g] && c <= cmax) acc[rg[g] + c*nrows_t] += v[g][c]; }
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

} // namespace

// per-level launch parameters of the factor and update kernels
int sparse_factor_setup(dlg_backend* b)
{
  SparseSym* Y = b->sym;
  const SymHost& H = Y->H;
  Y->fac_lds.assign(H.nlevels, 0); Y->fac_nt.assign(H.nlevels, 512); Y->upd_coop.assign(H.nlevels, 0);
  Y->upd_lds.assign(H.nlevels, 0); Y->upd_nw.assign(H.nlevels, 0);
  Y->syrk_lds.assign(H.nlevels, 0); Y->syrk_nt.assign(H.nlevels, 256); Y->syrk_kc.assign(H.nlevels, 4);
  Y->syrk_fused.assign(H.nlevels, 0);
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
  }
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_factor_level<128>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, FAC_LDS_BUDGET));
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_factor_level<256>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, FAC_LDS_BUDGET));
  DLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_factor_level<512>),
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
  return DLG_OK;
}

// K5: level-scheduled supernodal Cholesky (launches only; the caller reads the pivot flag)
int sparse_factor_levels(dlg_backend* b)
{
  SparseSym* Y = b->sym;
  const SymHost& H = Y->H;
  hipStream_t st = b->stream;
  for(int l = 0; l < H.nlevels; l++)
  {
    const int n = H.fw_lvl_ptr[l+1] - H.fw_lvl_ptr[l];
    if(n > 0)
    {
      const int o = H.fw_lvl_ptr[l];
      if(Y->fac_nt[l] == 128)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_factor_level<128>), dim3(n), dim3(128), Y->fac_lds[l], st,
                           Y->fw_sn + o, Y->fw_r0 + o, Y->fw_r1 + o, Y->sn_c0, Y->sn_rowptr, Y->sn_lx,
                           Y->sn_top, Y->sn_bd_ptr, Y->sn_bd_col, Y->Lx, Y->top_scr, Y->d_info,
                           Y->u_off, Y->uscr, Y->syrk_fused[l]);
      else if(Y->fac_nt[l] == 256)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_factor_level<256>), dim3(n), dim3(256), Y->fac_lds[l], st,
                           Y->fw_sn + o, Y->fw_r0 + o, Y->fw_r1 + o, Y->sn_c0, Y->sn_rowptr, Y->sn_lx,
                           Y->sn_top, Y->sn_bd_ptr, Y->sn_bd_col, Y->Lx, Y->top_scr, Y->d_info,
                           Y->u_off, Y->uscr, Y->syrk_fused[l]);
      else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_factor_level<512>), dim3(n), dim3(512), Y->fac_lds[l], st,
                           Y->fw_sn + o, Y->fw_r0 + o, Y->fw_r1 + o, Y->sn_c0, Y->sn_rowptr, Y->sn_lx,
                           Y->sn_top, Y->sn_bd_ptr, Y->sn_bd_col, Y->Lx, Y->top_scr, Y->d_info,
                           Y->u_off, Y->uscr, Y->syrk_fused[l]);
    }
    const int nu = H.uw_lvl_ptr[l+1] - H.uw_lvl_ptr[l];
    if(nu > 0 && H.upd_syrk[l] && Y->upd_nw[l] > 0)
    {
      const int ns = H.lvl_ptr[l+1] - H.lvl_ptr[l];
      if(Y->syrk_fused[l]) { /* phase 1 was done by the factor kernel */ }
      else if(Y->syrk_nt[l] == 256)
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
    else if(nu > 0 && Y->upd_coop[l] == 2)
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
  return DLG_OK;
}
