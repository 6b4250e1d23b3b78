// dlg_internal.h -- shared declarations of the HIP backend (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdarg>
#include <cmath>
#include <vector>
#include <string>
#include "../../include/dlg_backend.h"

// ---------------------------------------------------------------- errors ----
void dlg_set_error(const char* fmt, ...);

#define DLG_HIP(call)                                                              \
  do {                                                                             \
    hipError_t e__ = (call);                                                       \
    if(e__ != hipSuccess) {                                                        \
      dlg_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call,                  \
                    hipGetErrorString(e__));                                       \
      return DLG_ERR_HIP;                                                          \
    }                                                                              \
  } while(0)

#define DLG_CHECK(call)                                                            \
  do { int rc__ = (call); if(rc__ != DLG_OK) return rc__; } while(0)

#define DLG_LAUNCH_CHECK()  DLG_HIP(hipGetLastError())

// ------------------------------------------------------------- geometry ----
static inline int dlg_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

struct SparseSym;      // host+device symbolic data of the sparse path (sparse_symbolic.h)

struct DlgSlot
{
  double* p      = nullptr;   // [N]
  double* x      = nullptr;   // [M] (rank-local rows on a sharded rank)
  double* J      = nullptr;   // dense: [M][N]; sparse: values[nnz]; products: JtJ
  double* Jt_x   = nullptr;   // [N]
  double* cauchy = nullptr;   // [N]
  double* gn     = nullptr;   // [N]
  double* step   = nullptr;   // [N] step_to_here
  const double* x_bound = nullptr;   // device-resident inputs (dlg_point_bind_device)
  const double* J_bound = nullptr;
  double  norm2_x = 0, norm2_cauchy = 0, norm2_gn = 0, norm2_jtx = 0;
  // the expected improvement from the solved system (backend.hip, ident_norm2_Jstep): |J Jt_x|^2 of the Cauchy step, <Jt_x, gn>,
  // and whether the factor the Gauss-Newton step came from allows it (pivot ratio)
  double  Jg2 = 0, g_dot_gn = 0, a_dot_gn = 0, ident_lam = 0; bool ident_ok = false;      // (ident_norm2_Jstep: <Jt x, gn>, <cauchy, gn>, the lambda gn was solved at)
  bool    have_inputs = false, have_Jtx = false, have_cauchy = false, have_gn = false;
  const double* xin() const { return x_bound ? x_bound : x; }
  const double* Jin() const { return J_bound ? J_bound : J; }
};

struct dlg_backend
{
  int type = 0, N = 0, M = 0, nnz = 0, flags = 0, device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // side stream + events: p_new travels to the host while the expected improvement is computed (dlg_step)
  hipStream_t copy_stream = nullptr;
  hipEvent_t  ev_step = nullptr, ev_copy = nullptr, ev_fetch = nullptr;
  // second compute stream: the Cauchy step (one pass over J, independent of the factorisation) runs
  // beside the latency-bound upper levels of the elimination tree / the potrf chain, which leave
  // most of the chip idle.  The factorisation records ev_fork where that phase begins
  // (want_fork -> fork_recorded), the caller joins with ev_join before the step is formed.
  hipStream_t aux_stream = nullptr;
  hipEvent_t  ev_fork = nullptr, ev_join = nullptr;
  // the join without an event: the second stream raises d_join to join_epoch behind the Cauchy step, the first kernel of the
  // main stream that reads that step (k_negate_interp1) polls the word itself -- a wait for an event of another stream costs
  // the main stream ~6 us between two kernels even when the event is long complete
  int* d_join = nullptr; int join_epoch = 0, join_pending = 0;
  hipEvent_t ev_region = nullptr;    // behind this backend's last one-launch region (DlgRegionTurn)
  bool turn_registered = false;
  bool want_fork = false, fork_recorded = false, overlap = true;
  bool fuse_eval = true;      // ... in the pass that forms Jt*x where the schedule allows (DOGLEG_AMD_NO_FUSED_EVAL: second stream instead)
  bool speculate = false;     // dlg_backend_set_speculation: assemble JtJ beside Jt*x at every dlg_point_eval
  bool presolve = false;      // ... and enqueue K5 + K6 behind it (step_prepare) for dlg_take_step to pick up
  bool pre_rejected = false;  // the last point whose factorisation was enqueued ahead was rejected: the next evaluation enqueues nothing ahead (a step taken from a fresh point clears it)
  bool factor_ahead = false;  // (around sparse_factorize in step_prepare) only what covers the host's round trip: the leaf level
  bool pre_split = false;     // the prepared factorisation stopped behind its leaf level: dlg_take_step enqueues the rest and the solve
  bool prof_cont = false;     // the next timed scope continues a phase that was counted already (a factorisation in two parts)
  int  pre_slot = -1, pre_held = -1; double pre_lambda = 0.0, pre_hint = 0.0; bool pre_hint_valid = false, pre_hint_input = false;   // prepared slot; slot whose factor it displaced
  DlgSlot slot[2];

  // scalar return path: kernels write d_scal, one D2H into pinned h_scal
  double* d_scal = nullptr;
  double* h_scal = nullptr;
  const double* fold_p_src = nullptr; double* fold_p_dst = nullptr; bool p_copied = false;   // ... and p_new to a page-locked destination
  int* fork_gate = nullptr; int fork_gate_epoch = 0;      // dlg_fork_gate
  int fold_scal = 0; bool scal_copied = false;   // dlg_take_step: its last kernel (K8-sparse) copies d_scal to h_scal itself
  // The expected improvement's pass over J (K8) behind the host's decision point (dlg_backend_set_defer_tail): the value is
  // first used after the NEXT evaluation (dogleg.c:1427 -- takeStepFrom only needs max|step| to tell "done", 1289-1296), so
  // dlg_take_step's synchronisation rides on the step kernel (fold_scal_k7: it takes the scalars to the host), K8 follows on
  // the same stream while the host is on its way back, its partial sums (and p_new) land in page-locked memory and
  // dlg_step_tail adds them up -- behind a wait of its own only if the host has not waited for anything enqueued behind K8
  // since (sync_mark against tail_mark: the evaluation of the trial point is such a wait).
  bool defer_tail = false, tail_pending = false, tail_mode = false;
  int fold_scal_k7 = 0, tail_nb = 0;
  double tail_inner = 0.0, tail_value = 0.0;
  double* h_tail = nullptr; int h_tail_cap = 0;
  unsigned long sync_mark = 0, tail_mark = 0;
  bool factor_doomed = false; // sparse_factorize found the factorisation doomed at its look at the diagonal and enqueued nothing else (the lambda loops go on at once)
  bool kout_host = false;     // (dlg_step behind the decision point: k_interpolate's k straight into the page-locked scalars)
  // The expected improvement WITHOUT its pass over J (K8): with (JtJ + lambda I) gn = -Jt_x solved, |J step|^2 of all three
  // kinds of step is a combination of N-vector dot products (ident_norm2_Jstep).  The step kernel decides on the device
  // (the factor's pivot ratio small enough -- d_scal[IDENT_SLOT] = 1, [IDENT_SLOT + 1] = the ratio, [IDENT_SLOT + 4] = <cauchy - gn, cauchy>) and the
  // pass over J that is on the stream behind it returns at once (k8_skip); the host reads the same word and forms the value.
  // DOGLEG_AMD_EI_JPASS=1: always the pass over J.
  static constexpr int IDENT_SLOT = 3, GB_SLOT = 13;     // free slots of dlg_take_step's scalar block
  static constexpr double IDENT_RATIO_MAX = 212.0;       // (max L_ii / min L_ii)^2 * 2.2e-16 <= 1e-11
  bool ident_launched = false, ident_predict = false; const double* k8_skip = nullptr;      // ident_predict: the last step's pass over J was let go by the device
  bool ei_from_system = false; double pivot_ratio = NAN;   // dlg_backend_ei_source: how the last value handed out was formed
  int ei_flip = 0, ei_count = 0;                           // DOGLEG_AMD_DEBUG_EI_FLIP (test hook)
  bool tail_ident = false, tail_no_fold = false; double tail_nJs = 0.0;     // (tail_no_fold: the K8 on the stream carries no p_new)
  bool p_side_pending = false;          // p_new of a step behind the decision point is on its way on the copy stream (ev_copy)
  // Work the CALLER has for the stream that does not depend on a step's scalars (dlg_backend_set_between): enqueued from
  // inside dlg_take_step / dlg_step, between the step's last launch and the host's wait for it -- the device model's
  // kernels for the trial point, the first pass over the next point's J (dlg_point_eval_early).  The host's turn-round
  // behind the step (event wake-up, scalars, the evaluation's first launch: ~20 us) then hides behind that work.
  dlg_between_fn between_fn = nullptr; void* between_cookie = nullptr; bool between_armed = false, between_ran = false, between_redone = false;
  int early_slot = -1; const double* early_x = nullptr; const double* early_J = nullptr;      // dlg_point_eval_early: what dlg_point_eval finds enqueued
  static constexpr int NSCAL = 16;
  // hand-offs inside the one-launch regions (flags between workgroups): a wait that gives up raises a bit in
  // the status word (slot NSCAL - 2 of the scalar block, so it travels with every fetch of the scalars);
  // the host turns it into DLG_ERR_STATE (dlg_check_handoff).  handoff_skew != 0 (DOGLEG_AMD_DEBUG_HANDOFF_TIMEOUT,
  // read at create time): the waits look for an epoch that never comes -- the test hook of that path.
  int handoff_spins = 0, handoff_skew = 0;
  // environment knobs that steer per-step paths, read ONCE when the backend is created (dlg_backend_create)
  struct Knobs
  {
    bool potrf_steps = false, trsv_steps = false, no_abandon = false, ei_jpass = false, no_between = false, no_k8_predict = false;
    int touch_wg = 512;
  } knobs;
  int ncu = 256;              // compute units of b->device

  // reduction partials
  double* d_part = nullptr;
  size_t  part_cap = 0;       // in doubles
  double* d_gnpart = nullptr; // [1024] partials of |gn|^2 kept on the device (dlg_take_step)
  // Reductions whose result only the host reads skip their one-workgroup second stage: the
  // partials are written straight into page-locked host memory and summed (in index order) by
  // dlg_resolve_pending() after the synchronisation that the host needs anyway.  Off when an
  // all-reduce hook wants the results on the device.
  struct PendingFinal { size_t off; int nb, nsum, nmax, dst, stride; };
  static constexpr size_t HPART_CAP = 16384;
  double* h_part = nullptr;
  size_t  h_part_used = 0;
  std::vector<PendingFinal> pending;
  bool    host_finals = true;

  // pinned staging for p_new D2H / uploads of small vectors
  double* h_vec = nullptr;

  // dense / products
  double* G = nullptr;        // N x N column-major, lower triangle = factor
  double* Linv = nullptr;     // inverses of the 64x64 diagonal blocks of the factor
  int* potrf_flag = nullptr; int potrf_epoch = 0; int* trsv_flag = nullptr; double* trsv_y = nullptr; double* trsv_x = nullptr; int trsv_epoch = 0;   // hand-off flags of the one-launch factorisation ([T*T] + 1 for the step form)
  double* slabs = nullptr;    // split-K partial slabs for the SYRK
  size_t  slabs_bytes = 0;
  int*    d_info = nullptr;
  int*    h_info = nullptr;
  bool    defer_factor_sync = false;   // dlg_gauss_newton: the *_factorize calls enqueue only; the pivot
                                       // flag is read after the solve's synchronisation
  double* d_work = nullptr;   // N-vector scratch
  double* d_solve_scr = nullptr; size_t solve_scr_cap = 0;   // scratch of the post-solve entry points (backend.hip: solve_scratch)

  // sparse
  SparseSym* sym = nullptr;

  // multi-GPU: measurement rows are the sharded unit.  Contiguous rows [row0, row1) (dense; sparse
  // "row sharding": every rank factors everything, the whole JtJ is summed), or the subtree
  // partition of the sparse path (part_nranks > 1: the rank's rows are chosen by the symbolic phase,
  // only the top of the elimination tree is summed).  Sums over the ranks: RCCL on the backend's
  // stream (rccl_comm: no host in between), or the caller's hook (host-synchronous fallback).
  int row0 = 0, row1 = 0;     // owned measurement rows (contiguous mode)
  int mloc = 0;               // measurement rows held by this rank
  int part_rank = 0, part_nranks = 1;
  bool part_requested = false;          // dlg_backend_set_partition was called (also with nranks == 1)
  dlg_allreduce_fn allreduce = nullptr;
  void* allreduce_cookie = nullptr;
  void* rccl_comm = nullptr;  // ncclComm_t
  bool  rccl_owned = false;
  double* d_red = nullptr;    // fused reduce buffer [Jt_x | norm2_x | ...]
  // subtree partition: the Cauchy step's pass over the rank's rows runs on the second stream beside the
  // factorisation and leaves |J g|^2 of those rows in fold_scalar; the sum over the ranks is made together
  // with the solution's (sparse_solve -> fold_result = where the sum is), k_cauchy_finish follows on the main stream
  const double* fold_scalar = nullptr; const double* fold_result = nullptr; double* fold_cauchy_out = nullptr;
  // measurement only (dlg_backend_set_noop_comm): the backend behaves as one rank of several -- partition, reduce
  // buffers, device-side finals, every collective's place in the stream -- but a sum over the ranks returns at once
  // (the numbers are this rank's partial sums: timing, not results)
  bool noop_comm = false;
  bool sharded() const { return allreduce != nullptr || rccl_comm != nullptr || noop_comm; }

  // optional per-phase timing with HIP events on b->stream (dlg_backend_set_profiling)
  bool profiling = false;
  hipEvent_t attach_stop = nullptr; bool stop_attached = false, ext_events = true;      // DLG_LAUNCH_LAST
  int prof_every = 1; unsigned prof_tick[DLG_PROF_COUNT] = {};      // every n-th occurrence of a timed phase carries events
  unsigned prof_mask = 0;     // the phases that are timed (bit = DLG_PROF_*)
  struct ProfPair { hipEvent_t a, b; int id; bool cond; bool cont = false; };
  std::vector<ProfPair> prof_pending;
  std::vector<hipEvent_t> prof_pool;
  double prof_ms[DLG_PROF_COUNT] = {0};
  long   prof_n[DLG_PROF_COUNT] = {0};
  // Launches that return after their first barrier when the factorisation they belong to has failed (the lambda
  // path: the rest of K5, K6, K8) are timed "conditionally" (prof_cond set by the caller around them): their
  // times wait in prof_att_* until the attempt's outcome is known (dlg_prof_commit) and are then counted as
  // full launches (prof_ms / prof_n) or as early returns (prof_early_*) -- a per-launch average over both
  // would report bandwidths no kernel reaches (7 TB/s on config #5).
  bool   prof_cond = false;
  double prof_att_ms[DLG_PROF_COUNT] = {0};   long prof_att_n[DLG_PROF_COUNT] = {0};
  double prof_early_ms[DLG_PROF_COUNT] = {0}; long prof_early_n[DLG_PROF_COUNT] = {0};

  int factor_slot = -1;       // slot whose JtJ the stored factor belongs to (-1: none)
};

// profiling helpers: bracket a phase with events; resolved at the next stream sync
hipEvent_t dlg_prof_begin(dlg_backend* b);
void dlg_prof_end(dlg_backend* b, int id, hipEvent_t start);
void dlg_prof_resolve(dlg_backend* b);
void dlg_prof_commit(dlg_backend* b, bool attempt_succeeded);
struct DlgProfCond       // launches inside the scope return early if the factorisation of this attempt fails
{
  dlg_backend* b; bool was;
  explicit DlgProfCond(dlg_backend* b_) : b(b_), was(b_->prof_cond) { b_->prof_cond = true; }
  ~DlgProfCond() { b->prof_cond = was; }
};
struct DlgProfScope
{
  dlg_backend* b; int id; hipEvent_t e;
  DlgProfScope(dlg_backend* b_, int id_, bool enable = true) : b(b_), id(id_), e((enable && (b_->prof_mask >> id_ & 1u) && b_->prof_tick[id_]++ % b_->prof_every == 0) ? dlg_prof_begin(b_) : nullptr) {}
  ~DlgProfScope() { if(e) dlg_prof_end(b, id, e); }
};
// One kernel as a timed phase: the two events ride on the launch itself (hipExtLaunchKernelGGL: the dispatch's own
// start / end time stamps) -- no event records, i.e. no barrier packets and no idle microseconds, around it.
bool dlg_prof_pair(dlg_backend* b, int id, hipEvent_t* e0, hipEvent_t* e1);
#define DLG_LAUNCH_TIMED(b_, id_, kernel, grid, block, shm, st, ...) \
  do { hipEvent_t dlg_e0 = nullptr, dlg_e1 = nullptr; \
       if(dlg_prof_pair(b_, id_, &dlg_e0, &dlg_e1)) hipExtLaunchKernelGGL(kernel, grid, block, shm, st, dlg_e0, dlg_e1, 0, __VA_ARGS__); \
       else hipLaunchKernelGGL(kernel, grid, block, shm, st, __VA_ARGS__); } while(0)
// The event the host is going to wait for, attached to the last kernel of what it waits for (attach_stop set by the
// caller around the launch; stop_attached: the launch took it): no record behind the kernel.
#define DLG_LAUNCH_LAST(b_, kernel, grid, block, shm, st, ...) \
  do { if((b_)->attach_stop) { hipExtLaunchKernelGGL(kernel, grid, block, shm, st, (hipEvent_t)nullptr, (b_)->attach_stop, 0, __VA_ARGS__); (b_)->stop_attached = true; } \
       else hipLaunchKernelGGL(kernel, grid, block, shm, st, __VA_ARGS__); } while(0)

// rows of the measurement vector owned by this rank
static inline int dlg_mloc(const dlg_backend* b) { return b->mloc; }
// sum-all-reduce over ranks of `count` doubles at device address buf (no-op single rank)
int dlg_allreduce_dev(dlg_backend* b, double* buf, size_t count);

// the factorisation calls this where its latency-bound phase begins
static inline void dlg_fork_point(dlg_backend* b)
{
  if(b->want_fork && !b->fork_recorded && hipEventRecord(b->ev_fork, b->stream) == hipSuccess) b->fork_recorded = true;
}

// ... or, where that phase is ONE launch: the word that launch sets to `epoch` once all its workgroups are
// dispatched (no event on the main stream; the second stream holds a waiting kernel in front of its work)
static inline void dlg_fork_gate(dlg_backend* b, int* gate, int epoch)
{
  if(b->want_fork && !b->fork_recorded) { b->fork_gate = gate; b->fork_gate_epoch = epoch; b->fork_recorded = true; }
}

// fetch the first n scalars of d_scal to the host (synchronises the stream)
int dlg_fetch_scalars(dlg_backend* b, int n);

// what a kernel of a one-launch region needs to report a wait that gave up
struct DlgHandoff { int* status; int spins; int skew; };
enum { DLG_HANDOFF_FACTOR = 1, DLG_HANDOFF_SOLVE = 2, DLG_HANDOFF_POTRF = 4, DLG_HANDOFF_TRSV = 8, DLG_HANDOFF_TRSM = 16 };
static inline DlgHandoff dlg_handoff(const dlg_backend* b, int spins)
{
  DlgHandoff h;
  h.status = reinterpret_cast<int*>(b->d_scal + (dlg_backend::NSCAL - 2));
  h.spins = b->handoff_spins > 0 ? b->handoff_spins : spins; h.skew = b->handoff_skew;
  return h;
}
// One-launch regions (workgroups that wait for each other inside a launch) of DIFFERENT backends on one device take
// turns: two such launches at once can hold each other's CUs with waiting workgroups whose partners find no room --
// every one of them gives up after its 2^21 polls and the step fails with DLG_ERR_STATE (tools/stress_concurrent.py).
// With one backend on the device (one process per GPU: the rule) this is one atomic load.  Around the launch:
//   { DlgRegionTurn turn(b); hipLaunchKernelGGL(...region...); }
struct DlgRegionTurn { dlg_backend* b; bool on; explicit DlgRegionTurn(dlg_backend* b); ~DlgRegionTurn(); };
// after a synchronisation that brought the scalar block to the host: DLG_ERR_STATE if a hand-off timed out
int dlg_check_handoff(dlg_backend* b);

// --------------------------------------------------------- kernels_vec.hip --
// out[0] = sum x[i]^2 ; out[1] = max |x[i]|   (deterministic two-stage)
int k_norm2_absmax(dlg_backend* b, const double* x, int n, double* out2);
// ... of two vectors behind one launch
int k_norm2_absmax_pair(dlg_backend* b, const double* x1, int n1, double* out1, const double* x2, int n2, double* out2,
                        bool* on_host = nullptr);
// out[0] = <x,y>
int k_inner(dlg_backend* b, const double* x, const double* y, int n, double* out);
// Cauchy finish: g2 = |g|^2 (host), Jg2 = *Jg2_dev; k = -g2/Jg2;
// cauchy = k*g ; out[0] = k*k*g2
int k_cauchy_finish(dlg_backend* b, const double* g, double g2, const double* Jg2_dev, double* cauchy,
                    int n, double* out);
// step = s * v ; p_new = p + step ; out[0] = max|step|
int k_scaled_step(dlg_backend* b, const double* v, double s, const double* p, double* step,
                  double* p_new, int n, double* out_absmax);
// interpolation (dogleg.c:964-987): out = {norm2_step, k, max|step|}
int k_interpolate(dlg_backend* b, const double* a, const double* bb, double norm2a,
                  double trustregion, const double* p, double* step, double* p_new, int n,
                  double* out3);
// gn = -u ; out[0] = norm2(gn)
int k_negate_norm2(dlg_backend* b, double* v, int n, double* out);
// step chosen on the device (dogleg.c:1192-1256): out_n2_max[0] = |step|^2, [2] = max|step|; out3 = {kind, k, |gn|^2}
int k_negate_interp1(dlg_backend* b, double* gn, const double* cauchy, int n, double* gnpart, int* nb, const double* mm = nullptr, int nmm = 0, long mm_stride = 2);
int k_take_step(dlg_backend* b, const double* cauchy, const double* gn, const double* gnpart, int nbg,
                const double* n2c_dev, double trustregion, const double* p, double* step, double* p_new, int n,
                double* out_n2_max, double* out3, const double* Jtx, double* out_inner,
                double* out_gb = nullptr, double* ident_out = nullptr, bool have_mm = false, bool ident_gn = false, double ratio_max = 0.0);
// generic deterministic final reduction of `np` partials (sum) into out[0]
int k_reduce_sum(dlg_backend* b, const double* partials, int np, double* out);
// region of b->h_part for (nsum + nmax) x nb partials whose results go to h_scal[out - d_scal + k*stride],
// or nullptr: the result is wanted on the device / no room -> the caller launches the second stage
double* dlg_host_partials(dlg_backend* b, const double* out, int nb, int nsum, int nmax, int stride);
double* dlg_tail_partials(dlg_backend* b, int nb);      // page-locked room for the partial sums of a K8 behind the decision point (dlg_step_tail adds them)
void dlg_resolve_pending(dlg_backend* b);
int dlg_ensure_partials(dlg_backend* b, size_t ndoubles);

// ------------------------------------------------------- kernels_dense.hip --
int dense_create(dlg_backend* b);
// dense_diag.hip: factor the 64x64 diagonal block at kb and form its inverse (one workgroup)
void dense_launch_potrf_diag(hipStream_t st, double* A, int lda, int kb, int nb, int* info_dev, double* Linv);
// ... and the rows below it in the same launch (flag: one device int, epoch: a value no earlier launch used)
// the whole dense factorisation in one launch (a workgroup per 64 x 64 tile; flags: T*T device ints, T = ceil(n/64))
void dense_launch_potrf_tiles(hipStream_t st, double* A, int lda, int n, int* info_dev, double* Linv, int* flags, int epoch, const DlgHandoff& ho, int* gate = nullptr);
// both triangular solves of (L L') x = rhs in one launch (a workgroup per 64 rows; flags: 2*T device ints; Y: n doubles of scratch)
void dense_launch_trsv_tiles(hipStream_t st, const double* A, int lda, int n, const double* Linv, const double* rhs,
                             double* Yh, double* X, double* Xh, int epoch, const DlgHandoff& ho);
void dense_trsv_arm(hipStream_t st, double* Yh, double* Xh, size_t n_each);
void dense_launch_potrf_diag_trsm(hipStream_t st, double* A, int lda, int kb, int nb, int n, int* info_dev, double* Linv,
                                  int* flag, int epoch, const DlgHandoff& ho);
void dense_destroy(dlg_backend* b);
int dense_eval(dlg_backend* b, int slot);                       // K1
int dense_norm2_Jv(dlg_backend* b, int slot, const double* v, double* out_dev); // K3/K8
int dense_factorize(dlg_backend* b, int slot, double lambda, int* ok);          // K4+K5
bool dense_factor_ok(const dlg_backend* b);      // pivot flag of the last factorisation (after a sync)
int dense_solve(dlg_backend* b, const double* rhs, double* out);                // K6 (no negate)
int products_quadform(dlg_backend* b, int slot, const double* v, double* out_dev);
int products_factorize(dlg_backend* b, int slot, double lambda, int* ok);

// ------------------- sparse_host.hip / sparse_assemble.hip / sparse_solve.hip --
int sparse_create(dlg_backend* b);
size_t sparse_local_nnz(const dlg_backend* b);   // J values held by this rank
void sparse_destroy(dlg_backend* b);
void sparse_reset(dlg_backend* b);
int sparse_set_pattern(dlg_backend* b, const int* colptr, const int* rowidx);
int sparse_eval(dlg_backend* b, int slot);                      // K1
int sparse_assemble_speculative(dlg_backend* b, int s);         // K4 beside K1 (second stream, second panel buffer)
int sparse_eval_assemble(dlg_backend* b, int s, int* done);      // K1 + K4 in one pass over J (the assembly kernel forms Jt*x too)
int sparse_assemble_finish(dlg_backend* b);                      // ... the deferred partial-sum stages of that JtJ
int sparse_touch_factor(dlg_backend* b, hipStream_t st);         // second stream: pull the leaf panels into the Infinity Cache
int sparse_zero_spare(dlg_backend* b, hipStream_t ordered_for = nullptr);                           // clear the panel buffer the factorisation left behind (behind the step's fetch)
int sparse_abandon_enqueued(dlg_backend* b);                     // the launches of a factorisation + solve enqueued ahead (step_prepare) return early from here on
void sparse_spec_invalidate(dlg_backend* b, int s);
bool sparse_spec_is(const dlg_backend* b, int s, const double* J);     // the second panel buffer holds slot s's assembly from the values at J
// K3/K8; kind_if_factor_failed (device scalar holding the kind of step, or null): the pass is skipped when the
// factorisation on the stream failed and the step is not the Cauchy step to the edge
int sparse_norm2_Jv(dlg_backend* b, int slot, const double* v, double* out_dev, const double* kind_if_factor_failed = nullptr);
int sparse_factorize(dlg_backend* b, int slot, double lambda, int* ok);          // K4+K5 (b->factor_ahead: K5 up to the leaf level only, sparse_factorize_rest owes the rest)
int sparse_factorize_rest(dlg_backend* b, bool* was_pending);
bool sparse_factor_pending(const dlg_backend* b);                     // the levels above the leaves of a factorisation enqueued ahead
int sparse_norm2_chunks(const dlg_backend* b);
int dense_norm2_chunks(const dlg_backend* b);
bool sparse_factor_ok(const dlg_backend* b);     // pivot flag of the last factorisation (after a sync)
int sparse_solve(dlg_backend* b, const double* rhs, double* out);                // K6
void sparse_hold_factor(dlg_backend* b);
bool sparse_would_look(const dlg_backend* b, double lambda);      // sparse_factorize(lambda) would look at the diagonal first (and ask the host at once)
bool sparse_note_breakdown(dlg_backend* b);           // a factorisation broke down (the host knows): sparse_factorize looks at the diagonal first from now on (lambda = 0); true: stopped by that look, the panels are the assembly's still
void sparse_mark_unclean(dlg_backend* b);             // the next clear of either panel buffer is a full one (clear_panels)
// the partial-sum stages of an evaluation-time assembly on the SECOND stream (sparse_assemble.hip, "fin on the side")
bool sparse_fin_side_ok(const dlg_backend* b);
int  sparse_fin_side_begin(dlg_backend* b);      // main: augmented row + flag A; second stream: waits for flag A; b->stream := second stream
int  sparse_fin_side_end(dlg_backend* b);        // second stream: flag B behind the partial-sum stages; b->stream := main stream
int  sparse_fin_side_gate(dlg_backend* b);       // main: wait for flag B if stages are still owed (no-op otherwise)
// one wave that holds stream `st` until *gate == epoch (bounded: it is a matter of ordering work the chip has
// long finished in every measured case; a gate that never opens is reported through the hand-off status word)
int  dlg_gate_wait(dlg_backend* b, hipStream_t st, const int* gate, int epoch, bool report);
void sparse_release_held(dlg_backend* b);
int sparse_restore_factor(dlg_backend* b, bool* restored, bool rearm = true);
double sparse_current_lambda(const dlg_backend* b);                                 // of the last factorisation enqueued
const double* sparse_pivot_minmax(const dlg_backend* b, int* n);                    // [supernode][min, max] of diag(L) left by the last backward solve
// blocked multi-right-hand-side solves (sparse_multi.hip / kernels_dense.hip): MR = 16 right-hand sides
// interleaved [N][MR] (element (variable k, rhs c) at k*MR + c), solved in place, original order
int sparse_multi_width_ok(const dlg_backend* b);
int sparse_multi_rhs();
int sparse_solve_multi(dlg_backend* b, double* d_il);
int dense_solve_multi(dlg_backend* b, double* d_il);
int multi_cols_to_interleaved(dlg_backend* b, const double* d_cols, int ncols, double* d_il);
int multi_interleaved_to_cols(dlg_backend* b, const double* d_il, int ncols, double* d_cols);
int sparse_jt_chunk_interleaved(dlg_backend* b, int s, int row0, int ncols, double* d_il);
int dense_jt_chunk_interleaved(dlg_backend* b, int s, int row0, int ncols, double* d_il);
