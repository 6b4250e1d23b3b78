/* context_harness.c -- test helper (compiled by tests/test_context_gpu.py with gcc):
 * drives the returnContext part of the public API as a C user would
 * (reference dogleg.h:269-276, 304-310, 324-328): solve with returnContext, read the
 * operating point through the context, ask for the factorisation of JtJ at that point
 * (dogleg_computeJtJfactorization), free the context.  Prints what it sees as
 * "key v0 v1 ..." lines for the Python side to check.  usage: context_harness dense|sparse */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <dogleg.h>

typedef struct synth_s synth_t;
synth_t* synth_ba_create(int Nc, int Np, int Nobs, int g, uint64_t seed, double eps, double noise,
                         double p0_spread, double scale_decades, int n_zero_cols);
synth_t* synth_dense_create(int M, int N, uint64_t seed, double eps, double noise, double p0_spread);
void synth_free(synth_t* S);
int  synth_nstate(const synth_t* S);
int  synth_nmeas (const synth_t* S);
int  synth_nnz   (const synth_t* S);
void synth_p0    (const synth_t* S, double* out);
void synth_cb_sparse(const double* p, double* x, cholmod_sparse* Jt, void* cookie);
void synth_cb_dense(const double* p, double* x, double* J, void* cookie);

static void dump(const char* key, const double* v, int n)
{
  printf("%s", key);
  for(int i = 0; i < n; i++) printf(" %a", v[i]);
  printf("\n");
}

int main(int argc, char** argv)
{
  if(argc < 2) return 2;
  const int dense = strcmp(argv[1], "dense") == 0;
  synth_t* S = dense ? synth_dense_create(120, 10, 7, 0.3, 0.01, 0.5)
                     : synth_ba_create(4, 20, 60, 6, 2, 0.4, 0.01, 0.8, 0.0, 0);
  const int N = synth_nstate(S), M = synth_nmeas(S), nnz = synth_nnz(S);
  double* p = malloc(sizeof(double)*N);
  synth_p0(S, p);
  dogleg_parameters2_t prm;
  dogleg_getDefaultParameters(&prm);
  prm.max_iterations = 4;
  dogleg_solverContext_t* ctx = NULL;
  const double r = dense ? dogleg_optimize_dense2(p, N, M, &synth_cb_dense, S, &prm, &ctx)
                         : dogleg_optimize2(p, N, M, nnz, &synth_cb_sparse, S, &prm, &ctx);
  if(r < 0 || !ctx) { printf("FAILED solve\n"); return 1; }
  printf("dims %d %d %d\n", N, M, nnz);
  printf("result %a\n", r);
  dump("p_out", p, N);
  printf("ctx %d %d %d %a\n", (int)ctx->solve_type, ctx->Nstate, ctx->Nmeasurements, ctx->lambda);
  const dogleg_operatingPoint_t* pt = ctx->beforeStep;
  printf("flags %d %d %d\n", (int)pt->have_x, (int)pt->have_J, (int)pt->have_Jtx);
  printf("norm2_x %a\n", pt->norm2_x);
  dump("p", pt->p, N);
  dump("x", pt->x, M);
  dump("Jt_x", pt->Jt_x, N);
  if(dense) dump("J", pt->J_dense, M*N);
  else
  {
    printf("Jt_p"); for(int i = 0; i <= M; i++) printf(" %d", ((int*)pt->Jt->p)[i]); printf("\n");
    printf("Jt_i"); for(int i = 0; i < nnz; i++) printf(" %d", ((int*)pt->Jt->i)[i]); printf("\n");
    dump("Jt_x_vals", (double*)pt->Jt->x, nnz);
  }
  /* the factorisation of JtJ at the final point (dogleg.h:304-310) */
  if(!dogleg_computeJtJfactorization(ctx->beforeStep, ctx)) { printf("FAILED factorization\n"); return 1; }
  printf("have_factorization %d lambda %a\n", (int)ctx->beforeStep->have_factorization, ctx->lambda);
  if(dense) dump("factor_packed", ctx->factorization_dense, N*(N + 1)/2);
  else      printf("factor_handle %d\n", ctx->factorization != NULL);
  /* asking again is a no-op on a cached factor (dogleg.c:637) */
  if(!dogleg_computeJtJfactorization(ctx->beforeStep, ctx)) { printf("FAILED second factorization\n"); return 1; }
  dogleg_freeContext(&ctx);
  printf("freed %d\n", ctx == NULL);
  free(p);
  synth_free(S);
  return 0;
}
