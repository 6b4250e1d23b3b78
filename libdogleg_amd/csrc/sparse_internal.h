// sparse_internal.h -- shared by the DOGLEG_SPARSE translation units:
//   sparse_assemble.hip   K1 Jt*x, K3/K8 |Jv|^2, K4 JtJ assembly into the supernode panels
//   sparse_factor.hip     K5 level-scheduled supernodal Cholesky (panel factorisation + updates)
//   sparse_solve.hip      K6 triangular solves
//   sparse_host.hip       pattern set-up (symbolic phase, uploads, buffers), orchestration
// Kernels are launched from the file that defines them (no relocatable device code).
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
#pragma once
#include "dlg_internal.h"
#include "sparse_symbolic.h"

namespace {
constexpr int JFL_SEG = 16;             // workgroups per long Jt*x list (k_jtx_fin2_long)
constexpr int NV_CHUNK = 2048;          // non-zeros per workgroup of the |Jv|^2 kernel

constexpr int TPB = 256;
constexpr int LDS_BUDGET = 147456;      // bytes of dynamic LDS we allow a workgroup
constexpr int FAC_LDS_BUDGET = SYM_FAC_LDS_BUDGET;  // the panel factorisation takes (almost) all 160 KB of a CU; the rest is its static LDS

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

// flat record of a supernode for the backward solve (same order as lvl_sn)
// one work unit of k_update_gather, flat (everything the kernel needs before its first barrier behind ONE load)
struct GatherUnit { int64_t lt, part; int s0, s1, nrows_t, nc; };
struct SolveItem { int c0, w, nrows, rowoff; int64_t lx; int bd0, nbd; int pflag, rsv; };    // pflag: the parent's workgroup in the persistent backward launch (-1: none); rsv: pre-multiplied block sweep

struct SparseSym
{
  SymHost H;
  uint64_t pat_key = 0;         // hash of (sizes, rows of the rank, partition, pattern, schedule knobs): dlg_sparse_pattern_matches
  // schedules on the device
  int *sn_c0 = nullptr, *sn_rowptr = nullptr, *sn_rows = nullptr, *sn_scr = nullptr, *lvl_sn = nullptr;
  int64_t *sn_lx = nullptr, *diagpos = nullptr;
  int *ui_t = nullptr, *ui_col = nullptr, *ui_nc = nullptr, *ui_ptr = nullptr;
  SymSub* usub = nullptr; int* relpos = nullptr;
  int *uw_item = nullptr, *uw_s0 = nullptr, *uw_s1 = nullptr, *uf_item = nullptr, *uf_n = nullptr;
  int64_t *uw_part = nullptr, *uf_off = nullptr;
  GatherUnit* uw_flat = nullptr;          // [work units] flat records for k_update_gather
  double* upart = nullptr; double* uscr = nullptr; int64_t *u_off = nullptr, *usub_u = nullptr;
  SolveItem* slv_item = nullptr; FwItem* fw_item = nullptr; MfChild* mf_rec = nullptr; uint16_t* mf_dst = nullptr;
  SymOutBlock* oblk = nullptr; SymContrib* contrib = nullptr;
  SymTask *jtx_task = nullptr;
  int *jtx_fin_ptr = nullptr, *jtx_fin_blk = nullptr;
  int *jtx_fin_short = nullptr, *jtx_fin_long = nullptr; int n_fin_short = 0, n_fin_long = 0;   // entries by list length
  AsmRho* asm_rho = nullptr; AsmPair* asm_pair = nullptr; AsmSlot* asm_slot = nullptr;
  AsmBatch* asm_batch = nullptr; AsmTask* asm_ctask = nullptr; AsmFin* asm_cfin = nullptr;
  bool asm_pent_ok = false;
  uint32_t* asm_pent = nullptr;       // [shape][128] what the end of a task stores, one entry a lane (sparse_host.hip; asm_mfma_run: TS)
  AsmShape* asm_shape = nullptr; AsmKG* asm_kg = nullptr; AsmMTask* asm_mtask = nullptr; int* asm_tdest = nullptr;
  int *jf_ptr = nullptr, *jf_ent = nullptr, *jf_var0 = nullptr, *jf_w = nullptr, *jf_short = nullptr, *jf_long = nullptr;
  double* jf_lpart = nullptr; int* jf_lcnt = nullptr;      // k_jtx_fin2_long: [long block][JFL_SEG][16] segment sums, arrival counters
  const double* fin_pending_rhs = nullptr; const double* spec_aug_rhs = nullptr; bool info_clean = false;   // augmented row set at evaluation time
  double* fin_pending_Lx = nullptr;      // assembly whose partial-sum stages are still to be launched
  bool spare_dirty = false, spare_zeroed = false; hipStream_t spare_stream = nullptr;   // sparse_zero_spare
  double* jtp = nullptr;        // [16 per MFMA task] Jt*x records of the assembly kernel (sparse_eval_assemble)
  AsmFin2* asm_fin2 = nullptr; int64_t* asm_fin2_list = nullptr; AsmRun* asm_run = nullptr; int* asm_pdest = nullptr;
  int *rl_ptr = nullptr, *rl_pos = nullptr, *perm = nullptr, *col_sn = nullptr;
  int *fw_sn = nullptr, *fw_r0 = nullptr, *fw_r1 = nullptr, *ms_sn = nullptr; int64_t* sn_top = nullptr;
  int *sn_bd_ptr = nullptr, *sn_bd_col = nullptr;
  double* top_scr = nullptr;
  const double* aug_rhs = nullptr;        // rhs the augmented rows of the current factor were built from
  int *Jp = nullptr, *Ji = nullptr;       // rank-local pattern (row pointers rebased to 0)
  int *gat_rows = nullptr, *gat_src = nullptr;   // dlg_point_gather_device: the rank's rows, first value of each in the full value array
  int *nv_chunk = nullptr; int n_nv_chunks = 0;   // row runs of <= NV_CHUNK non-zeros for |Jv|^2
  // numeric buffers
  double *Lx = nullptr, *scr = nullptr, *ywork = nullptr, *asm_part = nullptr, *jtx_part = nullptr;
  int *d_info = nullptr, *h_info = nullptr;   // pivot flag: inside the backend's scalar block (device / pinned host)
  bool info_armed = false;                    // the assembly re-armed the flag (k_set_aug_row)
  // A factorisation at lambda = 0 that broke down before (the reference's lambda loop, dogleg.c:656-677; its lambda is sticky, a
  // caller that starts over from 0 every step pays the failed attempt every step): from then on a factorisation at lambda = 0
  // looks at the diagonal entries of the leaves' columns first (final once the assembly kernel is through) -- one that is not
  // positive is a pivot that is not (pivot <= diagonal entry), the doomed launches find the pivot word lowered and return.
  bool zero_fail_seen = false; int64_t* leaf_diag = nullptr; int* leaf_diag_col = nullptr; int n_leaf_diag = 0;
  // ... and a factorisation stopped by that look has stored nothing: its panels (intact_Lx, of slot intact_slot's inputs
  // intact_J, assembled with intact_lambda) are what the next attempt of the lambda loop factors (sparse_assemble)
  double* intact_Lx = nullptr; const double* intact_J = nullptr; int intact_slot = -1; double intact_lambda = 0.0;
  const double* fac_J = nullptr; int fac_slot = -1;      // the inputs of the last sparse_factorize
  bool fac_pending = false;                   // sparse_factor_levels(b, 1) launched the leaf level only: part 2 is owed (backend.hip, step_prepare)
  size_t nnz_loc = 0;
  // sharded rows: positions of the structural non-zeros of JtJ in Lx (what the all-reduce carries)
  uint32_t* ar_idx = nullptr; double* ar_buf = nullptr; size_t ar_n = 0;
  // per-level launch parameters
  std::vector<int> fac_lds;     // bytes of LDS for the factor kernel of a level (0: panels stay in HBM)
  std::vector<int> upd_lds, upd_nw, slv_lds, bwd_lds, fac_nt, upd_coop, syrk_lds, syrk_nt, syrk_kc, bwd_nt, syrk_fused, fin_ny, fac_stage, bwd_top, bwd_bd, bwd_pmx, fac_leaf;
  // subtree partition (part_nranks > 1): what is summed over the ranks between the last level below
  // the cut and the first one above it -- segments of Lx (panels above the cut) and of uscr (update
  // matrices that cross the cut), packed into red_buf --, and the 0/1 mask that makes the solution
  // a sum over the ranks (own subtrees everywhere, the replicated top on rank 0 only)
  int64_t* red_off = nullptr; int* red_len = nullptr; int* red_kind = nullptr; int64_t* red_dst = nullptr;
  int n_red_seg = 0; size_t red_n = 0; double* red_buf = nullptr;
  double* colmask = nullptr; int* sn_owner = nullptr;
  int64_t* augpos = nullptr;            // Lx offset of the augmented-row entry of every column
  int *xl_sn = nullptr;
  double* held_Lx = nullptr; const double* held_aug = nullptr; double held_lambda = 0.0;   // sparse_hold_factor
  double cur_lambda = 0.0;              // of the factorisation being enqueued (the top panels get it after the sum)
  double *ms_scr = nullptr, *ms_y = nullptr; int ms_lds_f = 0, ms_lds_b = 0;     // multi-right-hand-side solves (sparse_multi.hip)
  // speculative assembly beside K1 (sparse_assemble_speculative): second panel buffer, its state
  double* Lx_spec = nullptr; hipEvent_t ev_spec = nullptr, ev_spec_fork = nullptr;
  bool spec_inflight = false, spec_valid = false; int spec_slot = -1; const double* spec_J = nullptr;
  int64_t touch_off = 0, touch_n = 0;    // stretch of Lx with the leaf panels (sparse_touch_factor)
  int bw_level0 = 1 << 30, bw_lds = 0, bw_n = 0;   // persistent top region of the backward solve (sparse_solve_setup)
  SolveItem* slv_item_pr = nullptr; int* bwd_flag = nullptr; int bwd_epoch = 0; double* bwd_xh = nullptr;   // (bwd_xh: x of the region as its own arrival signal, two sets)
  int pr_stage = 0; double* pr_acc = nullptr;   // ... childless supernodes stage their update matrix; shadow scratch for the ones kept in HBM
  int pr_level0 = 1 << 30, pr_lds = 0;   // persistent top region of the factorisation: first level, LDS bytes (sparse_factor_setup)
  int* fac_flag = nullptr; int fac_epoch = 0;   // one flag per workgroup of the region: the epoch of the launch that finished it
  FwItem* pr_item = nullptr; MfChild* pr_rec = nullptr; uint16_t* pr_dst = nullptr; int pr_nwg = 0;
  // a subtree partition's second one-launch region: the rank's own levels pr2_level0 .. pr2_level1 (= the cut)
  int pr2_level0 = 1 << 30, pr2_level1 = -1, pr2_lds = 0, pr2_stage = 0, pr2_nwg = 0; int* fac2_flag = nullptr; int fac2_epoch = 0;
  FwItem* pr2_item = nullptr; MfChild* pr2_rec = nullptr; uint16_t* pr2_dst = nullptr;
  std::vector<FwItem> pr_item_h; std::vector<MfChild> pr_rec_h; std::vector<uint16_t> pr_dst_h;   // ... on the host (plan-only set-up: dlg_sparse_region_probe)   // the region's work items (supernode x replica) and its copy of the children records
  // fin on the side: flags [A: Jt*x final / augmented row on its way, B: partial-sum stages done], their epoch, the
  // epoch the main stream still has to wait for (0: nothing owed), whether the schedule allows it at all
  // partial clears (sparse_assemble.hip, clear_panels): the ranges of a panel buffer outside the merged leaves' panels,
  // and the buffers whose leaves are known to hold nothing but zeros outside the structure of JtJ
  int64_t* clr_off = nullptr; int64_t* clr_len = nullptr; int n_clr = 0; bool clr_partial_ok = false; double* lz_ok[2] = {nullptr, nullptr};
  int64_t* aug_of_var = nullptr; char* jf_listed = nullptr;      // by variable: where its augmented-row entry lies; its Jt*x comes from a record list
  int aug_fused_epoch = 0; const double* spec_aug_fused = nullptr;   // the Jt*x sums of the last evaluation set the augmented rows (epoch of their word)
  int* fin_flag = nullptr; int fin_epoch = 0, fin_side_owed = 0; bool fin_side_sched_ok = false; hipStream_t fin_main = nullptr;
  bool fac_b16 = false;         // panel_factor_b16 where the panel has at most 512 rows
  int bwd_xb_cap = 12288;       // below rows of a supernode staged in LDS by the backward solve
  double* diag_mm = nullptr; int n_diag_mm = 0;   // [supernode][min, max] of the diagonal of L (k_solve_bwd_level), read by the step kernels
  std::vector<void*> allocs;
};

// ---- per-file host entry points
int sparse_assemble(dlg_backend* b, int s, double lambda);   // K4 (+ all-reduce, lambda, augmented row)
int sparse_partition_reduce(dlg_backend* b);
int sparse_assemble_speculative(dlg_backend* b, int s);     // K4 on the second stream, beside K1
int sparse_eval_assemble(dlg_backend* b, int s, int* done);   // K1 + K4 in one pass over J
int sparse_assemble_finish(dlg_backend* b);                   // ... its JtJ partial-sum stages (deferred behind the fetch of Jt*x)
int sparse_zero_spare(dlg_backend* b, hipStream_t ordered_for);                        // clear the swapped-out panel buffer behind the step's fetch
void sparse_spec_invalidate(dlg_backend* b, int s);                 // subtree partition: the sum over the ranks at the cut
int sparse_factor_setup(dlg_backend* b, bool plan_only = false);   // per-level launch parameters of K5
int sparse_factor_levels(dlg_backend* b, int part = 0);      // K5 launches (no synchronisation); part 1: the leaf level only where the rest can follow later (fac_pending), part 2: that rest
int sparse_solve_setup(dlg_backend* b);                      // per-level launch parameters of K6
