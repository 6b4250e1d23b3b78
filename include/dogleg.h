/* dogleg.h -- public API of the MI355X-native dog-leg solver.
 *
 * Drop-in for libdogleg's dogleg.h (reference: /root/reference/dogleg.h):
 * identical function names, argument order, callback contracts, struct and
 * field names, so a program written against libdogleg re-links against
 * libdogleg_amd.so unchanged.  The trust-region control flow runs on the
 * host; every per-iteration linear-algebra op (Jt*x, |J v|^2, JtJ assembly,
 * Cholesky factor + solve, dog-leg interpolation) runs in HIP kernels on
 * gfx950 through the C-ABI declared in dlg_backend.h.
 *
 * What is NOT provided (out of the hot-path scope):
 * the experimental outlier / confidence API.
 *
 * Binary layout note: like the reference (dogleg.h:166-210) the context embeds
 * a cholmod_common by value as its first member, so the *binary* layout
 * depends on the CHOLMOD headers in use; source compatibility is the goal.
 */
#ifndef DOGLEG_AMD_DOGLEG_H
#define DOGLEG_AMD_DOGLEG_H

#include <stddef.h>
#include <stdbool.h>
#include "dogleg_cholmod_compat.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- user callbacks (reference dogleg.h:11-45) --------------------------- */

/* sparse: fill x[Nmeas] and Jt (CSC, Nstate rows x Nmeas cols; column r holds
 * d x[r] / d p, row indices ascending).  Jt->p/i/x are int/int/double arrays
 * owned by the library. */
typedef void (dogleg_callback_t)(const double* p, double* x,
                                 cholmod_sparse* Jt, void* cookie);

/* dense: fill x[Nmeas] and J[Nmeas][Nstate] (row-major) */
typedef void (dogleg_callback_dense_t)(const double* p, double* x,
                                       double* J, void* cookie);

/* dense products: the callback reduces over the measurements itself and
 * returns norm2(x), Jt*x and JtJ (full N*N, or one packed triangle) */
typedef void (dogleg_callback_dense_products_t)(const double* p,
                                                double* norm2x, double* xtJ,
                                                double* JtJ, void* cookie);

/* ---- one operating point (reference dogleg.h:48-105) ---------------------- */
typedef struct
{
  double* p;                       /* always valid */
  double* x;
  double  norm2_x;
  union
  {
    cholmod_sparse* Jt;            /* DOGLEG_SPARSE         */
    double*         J_dense;       /* DOGLEG_DENSE, [Nmeas][Nstate] */
    double*         JtJ;           /* DOGLEG_DENSE_PRODUCTS */
  };
  double* Jt_x;

  /* cached steps: a rejected trial is retried from these */
  double* updateCauchy;
  union
  {
    cholmod_dense* updateGN_cholmoddense;
    double*        updateGN_dense;
  };
  double norm2_updateCauchy, norm2_updateGN;

  union
  {
    int dummy_bits[3];
    struct
    {
      bool have_updateCauchy          : 1;
      bool have_updateGN              : 1;
      bool have_factorization         : 1;
      bool have_x                     : 1;
      bool have_J                     : 1;
      bool have_Jtx                   : 1;
      bool have_JtJ                   : 1;
      bool have_step_to_here          : 1;
      bool didStepToEdgeOfTrustRegion : 1;
    };
  };

  double* step_to_here;
  double  norm2_step_to_here;
} dogleg_operatingPoint_t;

/* ---- parameters (reference dogleg.h:107-153) ----------------------------- */
#define DOGLEG_DEBUG_VNLOG_BIT 30
#define DOGLEG_DEBUG_VNLOG     (1 << DOGLEG_DEBUG_VNLOG_BIT)

typedef struct
{
  int max_iterations;
  union
  {
    int dogleg_debug;              /* legacy view of the bits below */
    struct
    {
      bool debug       : 1;
      bool JtJ_packed  : 1;        /* dense-products: LAPACK packed triangle */
      bool JtJ_upper   : 1;        /* ... row-major upper if set             */
      int  dummy       : DOGLEG_DEBUG_VNLOG_BIT - 3;
      bool debug_vnlog : 1;        /* lands on bit DOGLEG_DEBUG_VNLOG_BIT    */
    };
  };

  double trustregion0;

  double trustregion_decrease_factor;
  double trustregion_decrease_threshold;
  double trustregion_increase_factor;
  double trustregion_increase_threshold;

  /* termination thresholds */
  double Jt_x_threshold;
  double update_threshold;
  double trustregion_threshold;
} dogleg_parameters2_t;

#ifndef __cplusplus
_Static_assert(offsetof(dogleg_parameters2_t, trustregion0) == 2 * sizeof(int),
               "dogleg_parameters2_t layout differs from libdogleg");
#else
static_assert(offsetof(dogleg_parameters2_t, trustregion0) == 2 * sizeof(int),
              "dogleg_parameters2_t layout differs from libdogleg");
#endif

typedef enum
{
  DOGLEG_DENSE          = 0,
  DOGLEG_SPARSE         = 1,
  DOGLEG_DENSE_PRODUCTS = 2
} dogleg_solve_type_t;

/* ---- solver context (reference dogleg.h:166-210) -------------------------- */
typedef struct
{
  cholmod_common common;

  union
  {
    dogleg_callback_t*                f;
    dogleg_callback_dense_t*          f_dense;
    dogleg_callback_dense_products_t* f_dense_products;
  };
  void* cookie;

  dogleg_operatingPoint_t* beforeStep;  /* current point between steps      */
  dogleg_operatingPoint_t* afterStep;   /* scratch point while trying a step */

  union
  {
    cholmod_factor* factorization;       /* sparse: handle to the GPU factor  */
    double*         factorization_dense; /* dense: packed factor, host mirror */
  };

  double lambda;                         /* sticky diagonal damping           */

  dogleg_solve_type_t solve_type;
  int Nstate, Nmeasurements;

  const dogleg_parameters2_t* parameters;
} dogleg_solverContext_t;

/* ---- parameter handling (reference dogleg.h:214-257, dogleg.c:117-181) ---- */
void dogleg_getDefaultParameters(dogleg_parameters2_t* parameters);
void dogleg_setMaxIterations(int n);
void dogleg_setTrustregionUpdateParameters(double downFactor, double downThreshold,
                                           double upFactor,   double upThreshold);
void dogleg_setDebug(int debug);
void dogleg_setInitialTrustregion(double t);
void dogleg_setThresholds(double Jt_x, double update, double trustregion);

/* ---- solves (reference dogleg.h:278-302, dogleg.c:1633-1818) --------------
 * p: in = initial estimate, out = optimum.  Return norm2(x) at the optimum, or
 * a negative number on error.  parameters == NULL selects the process-global
 * set edited by the dogleg_set*() functions.  A non-NULL returnContext
 * receives the solver state; release it with dogleg_freeContext(). */
double dogleg_optimize(double* p, unsigned int Nstate,
                       unsigned int Nmeas, unsigned int NJnnz,
                       dogleg_callback_t* f, void* cookie,
                       dogleg_solverContext_t** returnContext);
double dogleg_optimize2(double* p, unsigned int Nstate,
                        unsigned int Nmeas, unsigned int NJnnz,
                        dogleg_callback_t* f, void* cookie,
                        const dogleg_parameters2_t* parameters,
                        dogleg_solverContext_t** returnContext);
double dogleg_optimize_dense(double* p, unsigned int Nstate, unsigned int Nmeas,
                             dogleg_callback_dense_t* f, void* cookie,
                             dogleg_solverContext_t** returnContext);
double dogleg_optimize_dense2(double* p, unsigned int Nstate, unsigned int Nmeas,
                              dogleg_callback_dense_t* f, void* cookie,
                              const dogleg_parameters2_t* parameters,
                              dogleg_solverContext_t** returnContext);
double dogleg_optimize_dense_products(double* p, unsigned int Nstate,
                                      dogleg_callback_dense_products_t* f, void* cookie,
                                      const dogleg_parameters2_t* parameters,
                                      dogleg_solverContext_t** returnContext);

/* make sure ctx holds the Cholesky factor of JtJ at `point`
 * (reference dogleg.h:304-310, dogleg.c:634-820) */
bool dogleg_computeJtJfactorization(dogleg_operatingPoint_t* point,
                                    dogleg_solverContext_t* ctx);

void dogleg_freeContext(dogleg_solverContext_t** ctx);

/* ---- extensions of this implementation (nothing below exists in the reference) -------------
 *
 * Device-side evaluation.  The reference evaluates the user's model on the host at
 * every trial point (computeCallbackOperatingPoint, dogleg.c:1016-1022); here that means the
 * Jacobian values cross PCIe once per evaluation, which is the end-to-end bottleneck once the
 * kernels are fast.  A model that already lives on the GPU supplies a device callback instead:
 *   p_dev     in : Nstate doubles, device memory
 *   x_dev     out: Nmeas doubles, device memory (library-owned)
 *   J_dev     out: sparse: the NJnnz values of Jt in the order of the pattern (column r of Jt =
 *                  gradient of measurement r); dense: J[Nmeas][Nstate] row-major.  Device memory.
 *   hip_stream   : the hipStream_t (as void*) the callback must enqueue its kernels on; the
 *                  library orders its own work behind it on the same stream and the callback
 *                  must not synchronise.
 * dogleg_optimize_device2: the sparsity pattern of Jt is constant over a solve (as the reference
 * assumes, dogleg.c:648-649), so it is given once, on the host (Jt_colptr[Nmeas+1],
 * Jt_rowidx[NJnnz], row indices ascending within a column).  NJnnz == 0 and NULL pattern
 * pointers select the dense path.  Everything else -- p in/out on the host, return value, the
 * iterate sequence, returnContext -- is as dogleg_optimize2 / dogleg_optimize_dense2, except that
 * the Jacobian values of a returned context stay on the device (point->Jt->x / J_dense == NULL;
 * dlg_point_download(dogleg_amd_backend(ctx), slot, DLG_VEC_J, ...) fetches them). */
typedef void (dogleg_callback_device_t)(const double* p_dev, double* x_dev, double* J_dev,
                                        void* hip_stream, void* cookie);
double dogleg_optimize_device2(double* p, unsigned int Nstate,
                               unsigned int Nmeas, unsigned int NJnnz,
                               const int* Jt_colptr, const int* Jt_rowidx,
                               dogleg_callback_device_t* f, void* cookie,
                               const dogleg_parameters2_t* parameters,
                               dogleg_solverContext_t** returnContext);

/* Multi-GPU (the reference is single-threaded CPU code; its row sums dogleg.c:253-260, 269-278, 712-714 are
 * what is split).  One process per GPU; EVERY rank calls dogleg_optimize* / dogleg_optimize_device2 with the
 * same arguments and the same callback.  The callback keeps its contract -- it evaluates ALL measurement
 * rows at the p it is given (every rank sees the same p: the sums over the ranks leave identical bits
 * everywhere) --; the library takes from it the rows of its rank (sparse: those the subtree partition of
 * the elimination tree assigns, include/dlg_backend.h; dense: a contiguous slice), assembles / factors /
 * solves its share and sums Jt*x, the top of the tree (dense: JtJ), the solution and two scalars per step
 * over the ranks.  p, the return value, the iterate sequence and a returned context are the same on every
 * rank (the vectors of a returned context are complete; dogleg_amd_rank tells a rank which one it is).
 * dense-products solves have no rows to split: every rank does all of it (replicas).
 *
 * How a solve finds its communicator, in this order:
 *   dogleg_amd_set_communicator   per calling thread, before the solve: rank, ranks, the GPU (device index,
 *       -1: the current one) and the 128-byte RCCL id rank 0 made with dogleg_amd_rccl_unique_id and handed
 *       to the others by any means (MPI_Bcast, a file ...).  The sums are ncclAllReduce calls on the
 *       solve's own stream.
 *   dogleg_amd_set_allreduce      ... with the caller's sum-all-reduce over `count` doubles at a device
 *       address instead of RCCL (MPI, or the in-process sum the single-GPU tests use); host-synchronous.
 *   the environment               DOGLEG_AMD_WORLD_SIZE > 1, DOGLEG_AMD_RANK, DOGLEG_AMD_LOCAL_RANK (the GPU,
 *       default: the rank), DOGLEG_AMD_RCCL_ID_FILE (a path all ranks see: rank 0 writes the id there,
 *       the others wait for it; DOGLEG_AMD_RUN_ID names the launch, see dogleg_amd_id_file_publish): a program that was only re-linked against this library, started once
 *       per GPU by a launcher, needs no source change.  The communicator is made once per process.
 * All return 0 on success, -1 on bad arguments. */
typedef int (*dogleg_amd_allreduce_t)(void* buf_dev, size_t count, void* cookie);
int  dogleg_amd_set_communicator(int rank, int nranks, int device, const void* rccl_unique_id128);
int  dogleg_amd_set_allreduce(int rank, int nranks, int device, dogleg_amd_allreduce_t fn, void* cookie);
void dogleg_amd_clear_communicator(void);
int  dogleg_amd_rccl_unique_id(void* out128);
int  dogleg_amd_rank(const dogleg_solverContext_t* ctx, int* nranks);
/* The id file of the environment contract, for launchers that hand the id round themselves: rank 0 publishes
 * the 128-byte id (tmp + rename: never seen half-written), the others wait for a complete file that carries
 * the same run id (any string that names the launch; the environment contract uses DOGLEG_AMD_RUN_ID, else
 * TORCHELASTIC_RUN_ID, else ""), so that a file an earlier launch left at the path is not taken for this
 * launch's.  Under one run id the path must be fresh for each launch.  Host only, no GPU call.  0 / -1. */
int  dogleg_amd_id_file_publish(const char* path, const void* id128, const char* run_id);
int  dogleg_amd_id_file_wait(const char* path, void* id128_out, const char* run_id, int timeout_ms);

/* Between solves the library keeps one idle backend (device buffers, the uploaded sparsity pattern and its
 * schedules) and the page-locked host buffers of the operating points: a program that solves many problems
 * of one shape -- the same scene, new measurements -- pays allocations, uploads and the symbolic analysis
 * once (the reference redoes all of it per solve, dogleg.c:1479-1562, 1633-1753: its solves take seconds).
 * dogleg_amd_release_cache gives everything back; DOGLEG_AMD_NO_BACKEND_CACHE=1 never keeps anything. */
void dogleg_amd_release_cache(void);
/* measurement: where the calling thread's last solve spent its wall time, if DOGLEG_AMD_TIMING=1 was set for it (the same
 * numbers it printed on stderr): ms7 / calls7 = {pattern, model callback, inputs to the backend, dlg_point_eval,
 * dlg_take_step + dlg_step, trace records, run_optimizer as a whole}.  Returns the number of entries (7). */
int  dogleg_amd_last_solve_timing(double* ms7, int* calls7);

/* the device backend (include/dlg_backend.h) behind a returned context, and the backend slot of
 * one of its operating points: what dlg_solve_with_factor / dlg_solve_multi /
 * dlg_pseudoinverse_chunk / dlg_point_download need to work with the factor and the vectors that
 * stay on the device after a solve (the reference hands out a cholmod_factor / LAPACK factor
 * for the same purpose, dogleg.h:185-194) */
struct dlg_backend;
struct dlg_backend* dogleg_amd_backend(dogleg_solverContext_t* ctx);
int dogleg_amd_point_slot(dogleg_solverContext_t* ctx, const dogleg_operatingPoint_t* point);

/* gradient check of a callback (reference dogleg.h:312-322): prints, for variable `var`, the
 * reported d x[i] / d p[var] next to a central difference, one line per measurement, as a
 * vnlog-style table on stdout.  Host only. */
void dogleg_testGradient(unsigned int var, const double* p0,
                         unsigned int Nstate, unsigned int Nmeas, unsigned int NJnnz,
                         dogleg_callback_t* f, void* cookie);
void dogleg_testGradient_dense(unsigned int var, const double* p0,
                               unsigned int Nstate, unsigned int Nmeas,
                               dogleg_callback_dense_t* f, void* cookie);
void dogleg_testGradient_dense_products(unsigned int var, const double* p0,
                                        unsigned int Nstate, unsigned int Nmeas,
                                        dogleg_callback_dense_products_t* f, void* cookie);

#ifdef __cplusplus
}
#endif
#endif
