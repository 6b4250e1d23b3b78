/* dogleg_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, single-threaded restatement of libdogleg's algorithm
 * (/root/reference/dogleg.c) used ONLY as the checker in tests/, in
 * __graft_entry__.smoke() and as bench.py's cpu_baseline leg.  Nothing in
 * libdogleg_amd/ may call into this file.
 *
 * Parity pinning status: the reference itself cannot be built in this image
 * (it needs <cholmod.h>/libcholmod, which are absent; writing a stand-in is
 * not allowed), so there is no oracle/_ref.  The oracle is pinned against
 *   (1) the reference's own test assertions (sample.c:424-458, run for all
 *       four modes of check.sh:11-14), and
 *   (2) the known-answer trace of the bundled sample problem recorded in
 *       SURVEY.md Appendix B / 8c (tests/golden/sample_trace.json).
 * For the sparse path the arithmetic of the factorisation lives in CHOLMOD
 * (third-party, unvendored, version unpinned: Makefile:23 `-lcholmod`); its
 * published algorithm (up-looking simplicial Cholesky of A*A' + beta*I under a
 * fill-reducing permutation, Davis 2006 / CHOLMOD user guide) is restated
 * here; parity at the 1e-10 level for that boundary is by mathematical
 * equivalence ("parity unpinned" by the reference's own tests beyond 5e-2).
 */
#ifndef DOGLEG_ORACLE_H
#define DOGLEG_ORACLE_H

#include "../include/dogleg.h"
#include "../include/dlg_trace.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- whole solves; same contracts as dogleg.h ---------------------------- */
double orc_optimize_sparse(double* p, unsigned int Nstate, unsigned int Nmeas,
                           unsigned int NJnnz, dogleg_callback_t* f, void* cookie,
                           const dogleg_parameters2_t* parameters, dlg_trace_t* trace);
double orc_optimize_dense(double* p, unsigned int Nstate, unsigned int Nmeas,
                          dogleg_callback_dense_t* f, void* cookie,
                          const dogleg_parameters2_t* parameters, dlg_trace_t* trace);
double orc_optimize_dense_products(double* p, unsigned int Nstate,
                                   dogleg_callback_dense_products_t* f, void* cookie,
                                   const dogleg_parameters2_t* parameters, dlg_trace_t* trace);
void   orc_default_parameters(dogleg_parameters2_t* parameters);

/* ---- primitives (dogleg.c:190-347), exposed for per-kernel parity tests --- */
double orc_norm2(const double* x, unsigned int n);
double orc_inner(const double* x, const double* y, unsigned int n);
/* dest[nrow] = Jt * x ; Jt CSC with ncol columns */
void   orc_spmv_Jt_x(double* dest, int nrow, int ncol, const int* Jp, const int* Ji,
                     const double* Jx, const double* x);
/* norm2(J v) */
double orc_norm2_J_v(int ncol, const int* Jp, const int* Ji, const double* Jx, const double* v);
void   orc_dense_Jt_x(double* dest, const double* J, const double* x, int Nrows, int Ncols);
double orc_dense_norm2_J_v(const double* J, const double* v, int Nrows, int Ncols);
double orc_xt_Apacked_upper_x(const double* x, const double* A, int N);
double orc_xt_A_x(const double* x, const double* A, int N);
/* JtJ (row-major packed upper) += sum_r j_r j_r' ; caller zeroes */
void   orc_dense_JtJ_packed_upper(double* JtJ, const double* J, int Nrows, int Ncols);

/* LAPACK restatements on the reference's storage conventions.
 * packed 'L' (column-major lower == row-major upper). return info (0 = ok,
 * k>0: leading minor k not positive definite) */
int    orc_dpptrf_L(int n, double* ap);
void   orc_dpptrs_L(int n, const double* ap, double* b);
int    orc_dpotrf_L(int n, double* a, int lda);
void   orc_dpotrs_L(int n, const double* a, int lda, double* b);

/* ---- sparse Cholesky of Jt*J + beta*I (CHOLMOD stand-in) ------------------ */
typedef struct orc_sparse_factor orc_sparse_factor_t;
orc_sparse_factor_t* orc_sparse_analyze(int Nstate, int Nmeas, const int* Jp, const int* Ji);
/* returns minor (== Nstate on success, else index of the failed pivot in
 * elimination order) */
long   orc_sparse_factorize(orc_sparse_factor_t* F, const int* Jp, const int* Ji,
                            const double* Jx, double beta);
void   orc_sparse_solve(const orc_sparse_factor_t* F, const double* b, double* x);
long   orc_sparse_nnzL(const orc_sparse_factor_t* F);
double orc_sparse_flops(const orc_sparse_factor_t* F);
void   orc_sparse_free(orc_sparse_factor_t* F);

/* ---- one full trial step on fixed inputs: the unit bench.py times as the CPU
 * baseline.  Same op sequence as a step that refactorises and interpolates
 * (dogleg.c:1025-1027, 529-617, 634-820, 822-908, 927-998, 1259, 1085-1165):
 * Jt_x, norm2_x, Cauchy, JtJ + Cholesky, GN solve, dog-leg interpolation at
 * trustregion = mean(|cauchy|,|gn|), p_new, expected improvement.
 * work: 5*N doubles.  out[8] = {norm2_x, norm2_cauchy, norm2_gn, k, norm2_step,
 * expected_improvement, max|Jt_x|, max|step|}.  Returns 0, or >0 if JtJ is not
 * positive definite. */
int orc_step_sparse(orc_sparse_factor_t* F, int N, int M, const int* Jp, const int* Ji,
                    const double* Jx, const double* x, const double* p, double lambda,
                    double* work, double* out);
/* dense: dfac = N(N+1)/2 doubles scratch */
int orc_step_dense(int N, int M, const double* J, const double* x, const double* p, double lambda,
                   double* dfac, double* work, double* out);

#ifdef __cplusplus
}
#endif
#endif
