/* dlg_backend.h -- the thin C-ABI between the host trust-region driver and the
 * gfx950 HIP kernels.
 *
 * The reference has no seam here: its driver (dogleg.c:1172-1476) calls the
 * linear-algebra routines (dogleg.c:529-1165) as `static` functions in one
 * translation unit.  This header is the boundary a reference maintainer would
 * bind to put those routines on an MI355X; each entry point names the
 * reference routine it replaces.  Plain C types only: pointers, sizes, ints,
 * doubles; every function returns DLG_OK (0) or an error code and never
 * exits the process; dlg_last_error() describes the most recent failure on
 * the calling thread.
 *
 * Data residency: x / J / Jt_x / update vectors / the Cholesky factor live in
 * HBM for the life of the backend object.  Per trial step only the Jacobian
 * VALUES and x cross host->device (the user callback runs on the host,
 * dogleg.c:1016-1022) and p_new plus a few scalars come back.
 *
 * Two operating-point slots (0 and 1) mirror ctx->beforeStep / ctx->afterStep
 * (dogleg.h:178-182); the driver swaps slot ids instead of pointers.
 */
#ifndef DLG_BACKEND_H
#define DLG_BACKEND_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dlg_backend dlg_backend_t;

enum
{
  DLG_OK           = 0,
  DLG_ERR_HIP      = 1,   /* a HIP runtime call or kernel launch failed      */
  DLG_ERR_ARG      = 2,   /* bad argument                                    */
  DLG_ERR_STATE    = 3,   /* op called before its inputs exist               */
  DLG_ERR_NOMEM    = 4,
  DLG_ERR_NODEVICE = 5,   /* no usable gfx950 device                         */
  DLG_ERR_COMM     = 6    /* the all-reduce hook reported failure            */
};

/* solve types: same values as dogleg_solve_type_t (dogleg.h:158-161) */
enum { DLG_DENSE = 0, DLG_SPARSE = 1, DLG_DENSE_PRODUCTS = 2 };
/* flags */
enum { DLG_FLAG_JTJ_PACKED = 1, DLG_FLAG_JTJ_UPPER = 2 };
/* step kinds for dlg_make_step (the three branches of dogleg.c:1192-1256) */
enum { DLG_KIND_CAUCHY_TO_EDGE = 0, DLG_KIND_GAUSSNEWTON = 1, DLG_KIND_INTERPOLATED = 2 };
/* vectors that can be downloaded from a slot */
enum
{
  DLG_VEC_P = 0, DLG_VEC_X = 1, DLG_VEC_JTX = 2, DLG_VEC_CAUCHY = 3,
  DLG_VEC_GN = 4, DLG_VEC_STEP = 5, DLG_VEC_J = 6,
  /* dlg_point_device_ptr only: the slot's OWN x / J buffers (what a device-side evaluation writes
   * before dlg_point_bind_device), whatever is bound at the moment */
  DLG_VEC_X_OWN = 7, DLG_VEC_J_OWN = 8
};

const char* dlg_last_error(void);
int         dlg_device_count(void);

/* ---- lifecycle: replaces allocOperatingPoint x2 + factorization storage
 * (dogleg.c:1479-1562, 1707-1728).  device < 0: current device. ------------- */
int  dlg_backend_create(dlg_backend_t** out, int solve_type, int Nstate, int Nmeas,
                        int NJnnz, int flags, int device);
void dlg_backend_destroy(dlg_backend_t* b);
/* run on a caller-owned HIP stream (hipStream_t passed as void*) */
int  dlg_backend_set_stream(dlg_backend_t* b, void* hip_stream);
void* dlg_backend_get_stream(dlg_backend_t* b);

/* ---- multi-GPU: measurement rows are the sharded unit (every J-dependent quantity is a sum over
 * rows: dogleg.c:253-260, 269-278, 712-714).  One backend per rank (= per GPU).
 *
 * Who holds which rows:
 *   dlg_backend_set_partition(rank, nranks)   sparse: the SUBTREE PARTITION.  dlg_sparse_set_pattern
 *       (given the FULL pattern on every rank) cuts the elimination tree above the highest level with
 *       >= nranks supernodes; every subtree below the cut belongs to one rank, which holds the rows
 *       whose first-eliminated variable lies in it (dlg_partition_rows) and assembles, factors and
 *       solves that subtree alone.  Per factorisation only the panels above the cut and the update
 *       matrices crossing it are summed over the ranks (dlg_partition_stats: a few MB on the
 *       bundle-adjustment configurations, against the whole JtJ), the solution is completed by one
 *       sum over N doubles.  The replicated top of the tree is computed identically on every rank.
 *   dlg_backend_set_shard(row0, row1, fn, cookie)   contiguous rows [row0, row1): the dense path (JtJ
 *       summed, factorisation replicated) and the simple sparse variant (the whole assembled JtJ
 *       summed, every rank factors everything).  fn may be NULL when RCCL is used.
 * How the sums over the ranks are made:
 *   dlg_backend_init_rccl / dlg_backend_set_rccl   RCCL (ncclAllReduce, fp64 sum) enqueued on the
 *       backend's own stream: no host synchronisation per collective, every op of this header keeps
 *       its number of synchronisations.  librccl.so is loaded on demand.
 *   the hook (dlg_backend_set_shard / dlg_backend_set_allreduce)   a sum-all-reduce over `count`
 *       doubles at device address `buf` supplied by the caller (MPI, torch.distributed ...); the
 *       backend synchronises its stream before every call.  Fallback, and what the single-GPU tests
 *       use to run several logical ranks on one device. */
typedef int (*dlg_allreduce_fn)(void* buf, size_t count, void* cookie);
int  dlg_backend_set_shard(dlg_backend_t* b, int row0, int row1,
                           dlg_allreduce_fn fn, void* cookie);
int  dlg_backend_set_allreduce(dlg_backend_t* b, dlg_allreduce_fn fn, void* cookie);
int  dlg_backend_set_partition(dlg_backend_t* b, int rank, int nranks);   /* before dlg_sparse_set_pattern */
/* after dlg_sparse_set_pattern: the measurement rows this rank holds (ascending); dlg_point_upload /
 * dlg_point_bind_device expect x and the values of Jt for exactly these rows, in this order */
int  dlg_partition_rows(dlg_backend_t* b, int* nrows, const int** rows);
/* x and the values of Jt for ALL rows are on the device (a model evaluated there, colptr = the full pattern's
 * column pointers on the host): gather this rank's rows into the slot's own buffers, in the order of
 * dlg_partition_rows, and bind them as the slot's inputs (a kernel on the backend's stream, no host copy) */
int  dlg_point_gather_device(dlg_backend_t* b, int slot, const double* x_full_dev, const double* J_full_dev,
                             const int* colptr_host);
/* stats[] = {cut level, supernodes above the cut, supernodes of this rank, rows of this rank,
 * doubles summed per factorisation (panels above the cut + update matrices crossing it),
 * doubles of the whole panel buffer, non-zeros of this rank} */
int  dlg_partition_stats(dlg_backend_t* b, long* stats, int nstats);
/* host-only (no GPU): the partition of a pattern over nranks ranks as rank `rank` sees it.
 * row_owner[M] (may be NULL) receives 1 for the rows of this rank, 0 otherwise; stats as above */
int  dlg_sparse_partition_probe(int N, int M, const int* colptr, const int* rowidx, int rank, int nranks,
                                long* stats, int nstats, char* row_owner);
/* RCCL: rank 0 creates the id (128 bytes) and distributes it by any means; every rank then joins */
int  dlg_rccl_unique_id(void* out128);
int  dlg_backend_init_rccl(dlg_backend_t* b, int rank, int nranks, const void* unique_id128);
int  dlg_backend_set_rccl(dlg_backend_t* b, void* nccl_comm);            /* adopt a caller-owned ncclComm_t */
int  dlg_backend_share_rccl(dlg_backend_t* b, dlg_backend_t* owner);     /* ... the one another backend of this process made (it keeps owning it) */
int  dlg_backend_comm_size(dlg_backend_t* b, int* nranks);               /* what RCCL reports (1 without RCCL) */
int  dlg_backend_has_rccl(dlg_backend_t* b);
/* MEASUREMENT ONLY (bench.py --logical-ranks): one rank of a partition on a device of its own with every sum over
 * the ranks skipped -- the rank's compute time per phase, its numbers meaningless (partial sums). */
int  dlg_backend_set_noop_comm(dlg_backend_t* b, int on);                             /* 1: the sums over the ranks are ncclAllReduce calls on the backend's stream */

/* ---- nsteps trial steps of a fresh operating point in one call: bind resident inputs (ncopy copies of (x, J)
 * on the device, rotated from first_copy), dlg_point_eval, dlg_take_step from lambda0 -- the sequence of the
 * driver's takeStepFrom for a point with nothing cached, with the driver's host overhead and its two host
 * synchronisations per step (what bench.py times).  out9 (may be NULL): {|x|^2, |cauchy|^2, |gn|^2, k,
 * |step|^2, expected improvement, max|Jt x|, max|step|, lambda} of the last step; kind_out: its DLG_KIND_*. */
int  dlg_run_steps(dlg_backend_t* b, int from, int to, int nsteps, int ncopy, const double* const* x_dev,
                   const double* const* J_dev, int first_copy, double trustregion, double lambda0,
                   double* out9, int* kind_out);

/* ---- one backend for a series of solves (what the dogleg.h driver does between dogleg_optimize* calls: the
 * reference allocates and frees everything per solve, dogleg.c:1479-1562, 1694-1750 -- there a solve takes
 * seconds, here the set-up would be the solve).  dlg_backend_reset forgets the operating points and the held
 * factor and keeps every buffer, stream, the uploaded pattern and its schedules; dlg_sparse_pattern_matches
 * tells whether a pattern is the one the backend is set up for (exact: hash + compare), dlg_sparse_drop_pattern
 * makes room for another one. */
int  dlg_backend_reset(dlg_backend_t* b);
int  dlg_backend_device(dlg_backend_t* b);                               /* the HIP device index the backend lives on */
int  dlg_sparse_pattern_matches(dlg_backend_t* b, const int* colptr, const int* rowidx);
int  dlg_sparse_drop_pattern(dlg_backend_t* b);

/* ---- sparse pattern: replaces cholmod_analyze (dogleg.c:650-654).  The
 * pattern is assumed constant for the life of the solve, as the reference
 * assumes (dogleg.c:648-649).  colptr[Nmeas+1], rowidx[NJnnz] on the host
 * (the FULL pattern, also on a sharded rank).  Runs the host symbolic phase
 * (block detection, ordering, supernodes, schedules) and uploads it. */
int  dlg_sparse_set_pattern(dlg_backend_t* b, const int* colptr, const int* rowidx);
/* statistics of the symbolic phase; any pointer may be NULL */
int  dlg_sparse_stats(dlg_backend_t* b, long* nnz_JtJ_lower, long* nnz_L,
                      int* n_supernodes, int* n_levels, double* factor_flops);

/* host-only (no GPU): the schedule of the one-launch region of the factorisation as a chip with `ncu` compute
 * units would get it, checked for everything the kernel takes on trust (destinations inside LDS, replicas'
 * slices covering an update matrix exactly once, children before parents).  stats[] = {first level of the
 * region, supernodes, workgroups, LDS bytes, workgroups that keep a slice of their update matrix, supernodes
 * whose update matrix is summed in HBM} */
int  dlg_sparse_region_probe(int N, int M, const int* colptr, const int* rowidx, int ncu, long* stats, int nstats);

/* launch schedule of the factorisation: the number of levels of the supernodal
 * elimination tree; the first level of the persistent top region (all levels from
 * there on are ONE launch whose workgroups hand their update matrices to their
 * parents through flags) or -1 if there is none; the workgroups of that launch */
int  dlg_sparse_schedule(dlg_backend_t* b, int* n_levels, int* persist_level0, int* persist_items);

/* host-only symbolic phase on a pattern (no GPU): stats[] = {var-blocks,
 * supernodes, levels, nnz(tril JtJ), nnz(L), panel doubles, factor flops, max
 * panel, assembly tasks, update items, relpos entries, output blocks,
 * contributions, update sub-tasks, solve scratch, Jt*x tasks}; perm_out[N] may
 * be NULL */
int  dlg_sparse_symbolic_probe(int N, int M, const int* colptr, const int* rowidx, int row0,
                               int row1, long* stats, int nstats, int* perm_out);

/* ---- operating point inputs (host buffers; the callback's outputs).
 * K1 = computeCallbackOperatingPoint after the callback (dogleg.c:1024-1082):
 * Jt_x = Jt*x, norm2_x, and max_i |Jt_x[i]| for the gradient test. --------- */
int  dlg_point_set_p(dlg_backend_t* b, int slot, const double* p_host);
int  dlg_point_upload(dlg_backend_t* b, int slot, const double* x_host,
                      const double* J_host);           /* sparse: values[nnz]; dense: J[M][N] */
int  dlg_point_upload_products(dlg_backend_t* b, int slot, double norm2x,
                               const double* Jtx_host, const double* JtJ_host);
/* device-resident inputs (no PCIe): x and J already in HBM */
int  dlg_point_bind_device(dlg_backend_t* b, int slot, const double* x_dev,
                           const double* J_dev);
int  dlg_point_eval(dlg_backend_t* b, int slot, double* norm2_x, double* Jtx_absmax);
/* Assembly at evaluation time (sparse): while it is on, every dlg_point_eval also assembles the
 * slot's JtJ into a second panel buffer -- in the SAME pass over J that forms Jt*x where the
 * assembly schedule allows (the assembly kernel's B operand times x; DOGLEG_AMD_NO_FUSED_EVAL or
 * an irregular pattern: on a second stream beside the Jt*x kernel) --, and a following
 * dlg_factorize / dlg_gauss_newton / dlg_take_step of the same slot and inputs adopts it instead of
 * assembling again.  For callers that expect the evaluated point to be factorised (the driver turns
 * it on once a step has needed the Gauss-Newton step: an accepted point is factorised next).  JtJ
 * is the same bit for bit; in the one-pass form Jt*x is summed in another (fixed) order, i.e. it
 * differs from the stand-alone kernel's by rounding.  An assembly that is not used is dropped.
 * In the one-pass form (single rank) the evaluation also ENQUEUES the factorisation and the Gauss-Newton
 * solve of the slot, at the lambda the next step is expected to ask for (the one the last step ended with;
 * a caller that was seen to start over from its own value: that value), behind the point the host waits
 * for: dlg_take_step from that slot at that lambda picks them up.  Any other use leaves every result as
 * it would have been: a step from the other slot, or any call that uses the factor held before, gets
 * that factor back (its panels stay in the second buffer until the next step), another lambda
 * factorises again.  DOGLEG_AMD_NO_PRESOLVE=1 keeps the evaluation to the assembly. */
int  dlg_backend_set_speculation(dlg_backend_t* b, int on);
/* Work for the stream between a step and the host's wait for it.  dlg_take_step / dlg_step enqueue the step, then WAIT
 * for its scalars; the evaluation of the trial point that follows (dogleg.c:1410: computeCallbackOperatingPoint) is
 * enqueued when the host is back -- with a device-side model the chip idles meanwhile (event wake-up, the scalars, the
 * evaluation's first launch: ~20 us of a 0.6 ms step on config #4 once the expected improvement lost its pass over J).
 * dlg_backend_set_between(fn, cookie): the next dlg_take_step / dlg_step calls fn(cookie) ONCE, between the step's last
 * launch and its wait.  fn may enqueue, on the backend's stream, work that needs no scalar of the step: the device model's
 * kernels for the trial point (p of slot `to` is final in stream order) and dlg_point_eval_early -- the first pass over
 * the trial point's J (K1 + K4: Jt*x records and JtJ into the second panel buffer), exactly what dlg_point_eval would
 * launch first; dlg_point_eval of the same slot with the same inputs bound then finds it enqueued and goes on from there.
 * Nothing else of the API may be called from fn.  A step that had to be made again (its factorisation broke down: the
 * lambda loop, dogleg.c:656-677) drops that pass -- p_new changed --: dlg_backend_between_redone says so, the caller
 * evaluates as if fn had never run.  A trial step that turns out to end the solve (dogleg.c:1289-1296, 1403-1408) has
 * evaluated one point for nothing.  One call arms one step; fn == NULL disarms. */
typedef void (*dlg_between_fn)(void* cookie);
int  dlg_backend_set_between(dlg_backend_t* b, dlg_between_fn fn, void* cookie);
int  dlg_backend_between_redone(dlg_backend_t* b);      /* 1: the last step was made again behind fn -- what fn enqueued is void */
/* from inside fn only (sparse, single rank, dlg_backend_set_speculation on): *done = 1 if the pass was enqueued */
int  dlg_point_eval_early(dlg_backend_t* b, int slot, const double* x_dev, const double* J_dev, int* done);
/* The expected improvement behind the decision point.  takeStepFrom (dogleg.c:1172-1297) computes the expected improvement
 * with the step (1258-1269); runOptimizer looks at it twice: `expectedImprovement < 0.0` in front of the evaluation of the trial
 * point (dogleg.c:1403-1408: stop, the step is not applied; takeStepFrom's own -1 for max|step| below the threshold, 1289-1296,
 * is the usual way there) and as rho's denominator behind it (1410-1427).  With this on (single rank), dlg_take_step returns
 * NaN in its place (out7[6]) as soon as the step's scalars are on the host; the pass over J that forms |J step|^2 (K8) is on
 * the backend's stream behind the step, running while the host is on its way back and enqueues what comes next -- the model's
 * kernels, the evaluation of the trial point --, and dlg_step_tail waits for it and returns the value (sparse: bit for bit the
 * one dlg_take_step would have returned; dense: to rounding, the host adds the partial sums in index order).  A caller that
 * defers the value evaluates the trial point BEFORE it can make the first of the two tests, and makes it then (driver.hip,
 * run_optimizer: a computed value below zero still stops the solve with the step not applied; the one extra evaluation is
 * discarded).  A page-locked p_new_host is complete when dlg_step_tail returns, not before.  Until then the caller leaves
 * the J arrays of the slot the step was taken from alone (binding other arrays to the slot is fine).  dlg_step_tail with
 * nothing outstanding returns the value of the last step, whichever way it was formed; dlg_step_tail_pending says whether a
 * value is outstanding (NaN in out7[6] is how dlg_take_step says so, but a computed NaN looks the same).
 *
 * The value itself needs no pass over J where the Gauss-Newton system (JtJ + lambda I) gn = -Jt x was solved with a factor
 * whose pivots span less than 212x ((max L_ii / min L_ii)^2 eps <= 1e-11): |J gn|^2 = -<Jt x, gn> - lambda |gn|^2,
 * <J cauchy, J gn> = -<cauchy, Jt x> - lambda <cauchy, gn>, |J cauchy|^2 is the Cauchy step's own scalar (backend.hip:
 * ident_norm2_Jstep; to rounding the number of computeExpectedImprovement, dogleg.c:1085-1165).  The step kernel decides on
 * the device, the pass over J that is on the stream returns at once.  A wider pivot range (config #5: 3.5e6 at
 * lambda = 1e-10), several ranks: the pass over J, as before.
 * DOGLEG_AMD_EI_JPASS=1: always the pass over J. */
int  dlg_backend_set_defer_tail(dlg_backend_t* b, int on);
int  dlg_step_tail_pending(dlg_backend_t* b);
/* how the last expected improvement handed out was formed: *from_solved_system = 1 without a pass over J; *pivot_ratio =
 * max L_ii / min L_ii of the factor behind the last dlg_take_step (NaN: not looked at -- dense, several ranks, the knob) */
int  dlg_backend_ei_source(dlg_backend_t* b, int* from_solved_system, double* pivot_ratio);
int  dlg_step_tail(dlg_backend_t* b, double* expected_improvement);
/* measurement only (tools/rccl_floor.py): average enqueue-to-completion time, in microseconds, of `iters` all-reduces of
 * `count` doubles on the backend's stream through the communicator it holds */
int  dlg_backend_time_allreduce(dlg_backend_t* b, size_t count, int iters, double* us_each);

/* ---- K3: compute_updateCauchy (dogleg.c:529-617) -------------------------- */
int  dlg_cauchy(dlg_backend_t* b, int slot, double* norm2_updateCauchy);

/* ---- K4+K5: one pass of the loop body of dogleg_computeJtJfactorization
 * (dogleg.c:656-677 / 699-816): assemble JtJ + lambda*I and factorise.
 * *ok = 1 on success, 0 if not positive definite (the host raises lambda). -- */
int  dlg_factorize(dlg_backend_t* b, int slot, double lambda, int* ok);

/* ---- K6: compute_updateGN (dogleg.c:822-908): updateGN = -(JtJ)^-1 Jt_x ---- */
int  dlg_solve_gn(dlg_backend_t* b, int slot, double* norm2_updateGN);

/* ---- K4+K5+K6 fused: compute_updateGN as the reference runs it (dogleg.c:822-908 calls
 * dogleg_computeJtJfactorization first, dogleg.c:825): factorise with *lambda_io, raising it by the
 * reference's schedule (0 -> 1e-10 -> x10, dogleg.c:138,671-672,812-813) until the factorisation
 * succeeds, and solve.  One host synchronisation per attempt instead of two (dlg_factorize +
 * dlg_solve_gn).  On return *lambda_io is the lambda that worked. ---------------------------- */
int  dlg_gauss_newton(dlg_backend_t* b, int slot, double* lambda_io, double* norm2_updateGN);

/* ---- K3 + K4+K5+K6 behind one synchronisation: dlg_cauchy issued in front of dlg_gauss_newton.
 * takeStepFrom (dogleg.c:1186-1256) needs the Cauchy step first and the Gauss-Newton step whenever
 * the Cauchy step ends inside the trust region; a caller that expects that (the driver does once a
 * step needed both) saves a host round trip.  Same results as the two calls. ------------------- */
int  dlg_cauchy_gauss_newton(dlg_backend_t* b, int slot, double* lambda_io, double* norm2_updateCauchy,
                             double* norm2_updateGN);

/* ---- K7 + the vector part of takeStepFrom (dogleg.c:1192-1259,1289-1291):
 * form the step of the given kind from slot `from`, store it as
 * step_to_here of slot `to`, set p[to] = p[from] + step, copy p[to] to
 * p_new_host (may be NULL).  norm2_step follows the reference's reporting
 * (unscaled Cauchy length for DLG_KIND_CAUCHY_TO_EDGE). -------------------- */
int  dlg_make_step(dlg_backend_t* b, int from, int to, int kind, double trustregion,
                   double* norm2_step, double* k_cauchy_to_gn, double* step_absmax,
                   double* p_new_host);

/* ---- K8: computeExpectedImprovement (dogleg.c:1085-1165) for the step held
 * in slot `to`, evaluated with J / Jt_x of slot `from` ---------------------- */
int  dlg_expected_improvement(dlg_backend_t* b, int from, int to, double* out);

/* ---- K7 + K8 behind one synchronisation: dlg_make_step followed by dlg_expected_improvement,
 * as takeStepFrom issues them back to back (dogleg.c:1192-1269) ------------------------------- */
int  dlg_step(dlg_backend_t* b, int from, int to, int kind, double trustregion,
              double* norm2_step, double* k_cauchy_to_gn, double* step_absmax,
              double* expected_improvement, double* p_new_host);

/* ---- K3 .. K8 behind ONE synchronisation: takeStepFrom (dogleg.c:1172-1297) for a point with
 * nothing cached -- Cauchy step, factorise + solve (lambda loop as in dlg_gauss_newton), the choice
 * of the kind of step (made on the device with the reference's comparisons, dogleg.c:1192-1256),
 * the step, its expected improvement and p_new.
 * out7 = {|cauchy|^2, |gn|^2, kind, |step|^2 (reference reporting), k_cauchy_to_gn, max|step|,
 * expected improvement}.  The Gauss-Newton step is computed speculatively; if the Cauchy step is the
 * one taken (kind == DLG_KIND_CAUCHY_TO_EDGE) it is discarded with its factorisation, *lambda_io is
 * NOT modified (the reference does not factorise on that branch, dogleg.c:1192-1211) and
 * out7[1] = NaN. --------------------------------------------------------------------------- */
int  dlg_take_step(dlg_backend_t* b, int from, int to, double trustregion, double* lambda_io,
                   double* out7, double* p_new_host);

/* ---- post-solve reuse of the factor: (JtJ + lambda I) u = rhs for nrhs right-hand
 * sides (host, N doubles each, one after the other) with the factorisation held for `slot` -- what a
 * libdogleg user does with cholmod_solve / dpotrs on ctx->factorization (dogleg.h:304-310; the
 * reference's own user is its outlier / confidence code, dogleg.c:1831-1921).  The factor stays on
 * the device. --------------------------------------------------------------------------------- */
int  dlg_solve_with_factor(dlg_backend_t* b, int slot, const double* rhs_host, double* out_host, int nrhs);
/* the same, BLOCKED: 16 right-hand sides per pass over the factor (the factor is read once per block,
 * the products with its off-diagonal part run on the matrix cores): cholmod_solve / dpptrs with a
 * block of right-hand sides.  rhs / out: N x nrhs column-major (ld = N), host. */
int  dlg_solve_multi(dlg_backend_t* b, int slot, const double* rhs_host, double* out_host, int nrhs);
/* out (N x (row1 - row0), column-major, host) = inv(JtJ + lambda I) * Jt[:, row0:row1] with the factor held
 * for `slot` and the slot's Jacobian on the device: pseudoinverse_J_dense / pseudoinverse_J_sparse of
 * the reference (dogleg.c:1831-1921), built on the blocked solve */
int  dlg_pseudoinverse_chunk(dlg_backend_t* b, int slot, int row0, int row1, double* out_host);

/* ---- downloads (returnContext, tests) -------------------------------------- */
int  dlg_point_download(dlg_backend_t* b, int slot, int which, double* host, size_t n);
/* dense factor in the reference's layout (packed as dpptrf('L') leaves it, or
 * full N*N for unpacked products): dogleg.h:192-194 */
int  dlg_factor_download_dense(dlg_backend_t* b, double* host, size_t n);
/* device address of a slot vector (DLG_VEC_*), for zero-copy harnesses */
void* dlg_point_device_ptr(dlg_backend_t* b, int slot, int which);

/* ---- stand-alone kernels exposed for parity tests and micro-benchmarks ----
 * C (n x n, column-major, ldc) lower triangle += A^T-style rank-K update:
 * C[i,j] += alpha * sum_k A[i + k*lda] * A[j + k*lda]   (i >= j)
 * with device pointers.  This is the fp64-MFMA kernel behind K4-dense and the
 * trailing updates of K5-dense. */
int  dlg_kernel_syrk_lower(void* hip_stream, double* C_dev, int ldc, const double* A_dev,
                           int lda, int n, int K, double alpha, double* workspace_dev,
                           size_t workspace_bytes);
/* in-place blocked Cholesky of the lower triangle (column-major). info_dev: int */
int  dlg_kernel_potrf_lower(void* hip_stream, double* A_dev, int lda, int n, int* info_dev);
/* f64 MFMA issue-rate probe: returns achieved TFLOP/s */
int  dlg_probe_mfma_f64(double* tflops);
int  dlg_probe_mfma_f64_clock(double* tflops, double* clock3);   /* + {shader MHz during the loop, clocks per MFMA and wave, per SIMD} */
int  dlg_probe_mfma_f64_waves(int waves_per_simd, double* tflops, double* clock3);   /* the same with 1 .. 8 waves a SIMD, one resident round */
int  dlg_probe_hbm_copy(double* gbytes_per_s);

/* ---- per-phase GPU timing with HIP events on the backend's stream (bench.py's
 * roofline numbers come from here).  Phases: */
enum
{
  DLG_PROF_K1_JTX = 0,        /* Jt*x kernels                                   */
  DLG_PROF_K3K8_NORM2JV = 1,  /* |J v|^2 kernels (Cauchy + expected improvement) */
  DLG_PROF_K4_KERNEL = 2,     /* the JtJ assembly kernel alone (sparse: k_assemble;
                                 dense: the MFMA SYRK kernel)                    */
  DLG_PROF_K4_TOTAL = 3,      /* assembly incl. memset / partial-sum finalize    */
  DLG_PROF_K5_FACTOR = 4,     /* numeric Cholesky                                */
  DLG_PROF_K6_SOLVE = 5,      /* triangular solves                               */
  DLG_PROF_K7_STEP = 6,       /* step formation                                  */
  DLG_PROF_VEC = 7,           /* other O(N) reductions                           */
  DLG_PROF_COUNT = 8
};
/* on = 0: off; 1: every phase; else: only the phases p with bit (p + 1) of `on` set, e.g.
 * 2 << DLG_PROF_K4_KERNEL (two event records per step instead of twenty).  Bits 16-23, if not zero: only every
 * n-th occurrence of a phase carries events (a kernel whose completion somebody listens to holds the next
 * dispatch back by ~5 us: sampling keeps the timed loop close to the untimed one).  Clears the counters. */
int  dlg_backend_set_profiling(dlg_backend_t* b, int on);
int  dlg_backend_get_profile(dlg_backend_t* b, double* ms_total, long* launches, int n);
/* the launches that returned after their first barrier because the factorisation they belong to had failed (the
 * lambda path, dogleg.c:670-673: what is left of K5, K6, K8 of the attempt that is thrown away) -- counted
 * here and NOT in dlg_backend_get_profile, whose per-launch averages are over launches that ran in full */
int  dlg_backend_get_profile_early(dlg_backend_t* b, double* ms_total, long* launches, int n);

/* ---- raw device-memory helpers for harnesses that hold inputs in HBM without
 * a framework (tests, bench): thin wrappers of hipMalloc/hipMemcpy/hipFree --- */
void* dlg_mem_alloc(size_t bytes);
void  dlg_mem_free(void* dev);
int   dlg_mem_upload(void* dev, const void* host, size_t bytes);
int   dlg_mem_download(void* host, const void* dev, size_t bytes);
int   dlg_mem_zero(void* dev, size_t bytes);
/* page-locked host memory (hipHostMalloc): a p_new_host / download destination allocated here is
 * written by the DMA engine directly, pageable memory goes through a staging buffer */
void* dlg_host_alloc(size_t bytes);
void  dlg_host_free(void* host);
int   dlg_device_sync(void);

/* ---- trial trace sink used by the dogleg_optimize* entry points on the
 * calling thread (include/dlg_trace.h); NULL disables ---------------------- */
struct dlg_trace_s;
void dlg_set_trace(void* dlg_trace_t_ptr);

#ifdef __cplusplus
}
#endif
#endif
