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
