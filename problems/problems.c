/* problems.c -- test/bench problem definitions (callbacks in the dogleg.h
 * callback contracts).  Test and benchmark plumbing: used by tests/, bench.py
 * and __graft_entry__.smoke(); the same callbacks are handed to the product
 * (libdogleg_amd.so) and to the CPU oracle.
 *
 *  1. "sample": the reference's bundled test problem (sample.c:22-80 data
 *     generation, sample.c:82-237 callbacks, sample.c:351-371 start point),
 *     re-stated: 6 parameters, 100 measurements on a 10x10 grid,
 *       x_i = p0 p1 X^2 + p1 p2 Y^2 + p2 X Y + p3 X + p4 Y + p5 - m_i
 *     with m_i simulated at p=(1..6) plus glibc random() noise after
 *     srandom(0).  It is the fixture of the reference's check.sh.
 *
 *  2. "ba": synthetic block-arrowhead (bundle-adjustment shaped) sparse
 *     problems, SURVEY.md 8d generator G(Nc,Np,Nobs,g,6,3,seed): g global +
 *     6/camera + 3/point parameters, 2 measurement rows per observation, 15
 *     non-zeros per row in ascending index order, banded co-visibility.
 *       u_r = sum_j a_rj (p_j - p*_j);  x_r = u_r + eps sin(u_r) - n_r
 *
 *  3. "dense": the same residual model with a dense M x N coefficient matrix.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include "../include/dogleg.h"

/* ======================================================================== */
/* 1. sample problem                                                         */
/* ======================================================================== */
#define SAMPLE_N 6
#define SAMPLE_GRID 10
#define SAMPLE_M (SAMPLE_GRID*SAMPLE_GRID)
static double smp_X[SAMPLE_M], smp_Y[SAMPLE_M], smp_meas[SAMPLE_M];
static const double smp_truth[SAMPLE_N] = {1., 2., 3., 4., 5., 6.};

static double smp_model(const double* p, double X, double Y)
{
  return p[0]*p[1]*X*X + p[1]*p[2]*Y*Y + p[2]*X*Y + p[3]*X + p[4]*Y + p[5];
}
static void smp_gradient(const double* p, double X, double Y, double* g)
{
  g[0] = p[1]*X*X;
  g[1] = p[0]*X*X + p[2]*Y*Y;
  g[2] = p[1]*Y*Y + X*Y;
  g[3] = X;
  g[4] = Y;
  g[5] = 1.0;
}

/* Reproduces the reference program's random stream: srandom(0), one random()
 * per measurement (noise in [-0.5,0.5)), then one per parameter for the start
 * point (random()/RAND_MAX - 0.1).  p0 receives the start point. */
void sample_init(double* p0)
{
  srandom(0);
  int k = 0;
  for(int ix = 0; ix < SAMPLE_GRID; ix++)
    for(int iy = 0; iy < SAMPLE_GRID; iy++, k++)
    {
      smp_X[k] = -10. + ix*2.0;
      smp_Y[k] = -10. + iy*2.0;
    }
  for(int i = 0; i < SAMPLE_M; i++)
  {
    const double X = smp_X[i], Y = smp_Y[i];
    smp_meas[i] =
      smp_truth[0]*smp_truth[1] * X*X +
      smp_truth[1]*smp_truth[2] * Y*Y +
      smp_truth[2] * X*Y +
      smp_truth[3] * X +
      smp_truth[4] * Y +
      smp_truth[5] +
      ((double)random() / (double)RAND_MAX - 0.5) * 1.0;
  }
  for(int i = 0; i < SAMPLE_N; i++)
    p0[i] = ((double)random() / (double)RAND_MAX - 0.1) * 1.0;
}
/* install measurements from a committed fixture instead of random() */
void sample_set_measurements(const double* m)
{
  int k = 0;
  for(int ix = 0; ix < SAMPLE_GRID; ix++)
    for(int iy = 0; iy < SAMPLE_GRID; iy++, k++)
    { smp_X[k] = -10. + ix*2.0; smp_Y[k] = -10. + iy*2.0; }
  memcpy(smp_meas, m, sizeof(smp_meas));
}
void sample_get_measurements(double* m) { memcpy(m, smp_meas, sizeof(smp_meas)); }

void sample_cb_sparse(const double* p, double* x, cholmod_sparse* Jt, void* cookie)
{
  (void)cookie;
  int* colptr = (int*)Jt->p; int* rowidx = (int*)Jt->i; double* val = (double*)Jt->x;
  int q = 0;
  for(int i = 0; i < SAMPLE_M; i++)
  {
    double g[SAMPLE_N];
    x[i] = smp_model(p, smp_X[i], smp_Y[i]) - smp_meas[i];
    smp_gradient(p, smp_X[i], smp_Y[i], g);
    colptr[i] = q;
    for(int j = 0; j < SAMPLE_N; j++, q++) { rowidx[q] = j; val[q] = g[j]; }
  }
  colptr[SAMPLE_M] = q;
}
void sample_cb_dense(const double* p, double* x, double* J, void* cookie)
{
  (void)cookie;
  for(int i = 0; i < SAMPLE_M; i++)
  {
    x[i] = smp_model(p, smp_X[i], smp_Y[i]) - smp_meas[i];
    smp_gradient(p, smp_X[i], smp_Y[i], &J[i*SAMPLE_N]);
  }
}
/* cookie = const dogleg_parameters2_t* (selects the JtJ layout), as sample.c does */
void sample_cb_products(const double* p, double* norm2x, double* xtJ, double* JtJ, void* cookie)
{
  const dogleg_parameters2_t* prm = (const dogleg_parameters2_t*)cookie;
  const int packed = prm->JtJ_packed, upper = prm->JtJ_upper;
  const int size = packed ? SAMPLE_N*(SAMPLE_N+1)/2 : SAMPLE_N*SAMPLE_N;
  *norm2x = 0.0;
  memset(xtJ, 0, sizeof(double)*SAMPLE_N);
  memset(JtJ, 0, sizeof(double)*size);
  for(int i = 0; i < SAMPLE_M; i++)
  {
    double g[SAMPLE_N];
    const double xi = smp_model(p, smp_X[i], smp_Y[i]) - smp_meas[i];
    smp_gradient(p, smp_X[i], smp_Y[i], g);
    *norm2x += xi*xi;
    for(int k = 0; k < SAMPLE_N; k++) xtJ[k] += xi*g[k];
    if(packed && upper)
    {
      int t = 0;
      for(int k = 0; k < SAMPLE_N; k++)
        for(int l = k; l < SAMPLE_N; l++, t++) JtJ[t] += g[k]*g[l];
    }
    else if(!packed)
    {
      for(int k = 0; k < SAMPLE_N; k++)
        for(int l = 0; l < SAMPLE_N; l++) JtJ[k*SAMPLE_N + l] += g[k]*g[l];
    }
    else
    {
      /* packed lower: [A B D C E F] */
      int t = 0;
      for(int k = 0; k < SAMPLE_N; k++)
        for(int l = 0; l <= k; l++, t++) JtJ[t] += g[k]*g[l];
    }
  }
}

/* ======================================================================== */
/* counter-based PRNG (splitmix64 finaliser)                                 */
/* ======================================================================== */
static inline uint64_t mix64(uint64_t z)
{
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
/* uniform in [-1,1) from (seed, stream, index) */
static inline double urand(uint64_t seed, uint64_t stream, uint64_t idx)
{
  const uint64_t h = mix64(mix64(seed ^ (stream*0xD6E8FEB86659FD93ull)) + idx);
  return (double)(h >> 11) * (2.0/9007199254740992.0) - 1.0;
}

/* ======================================================================== */
/* 2. sparse block-arrowhead problems                                        */
/* ======================================================================== */
typedef struct
{
  int kind;                 /* 0 = ba sparse, 1 = dense */
  int Nc, Np, Nobs, g;
  int N, M, nnz;
  uint64_t seed;
  double eps, noise, p0_spread;
  int*    Jp; int* Ji;      /* pattern (sparse) */
  double* a;                /* coefficients, nnz (sparse) */
  double* pstar;
  double* colscale;         /* per-variable scale (ill-conditioning), N */
  int neval;
} synth_t;

#define BC 6
#define BP 3

synth_t* synth_ba_create(int Nc, int Np, int Nobs, int g, uint64_t seed,
                         double eps, double noise, double p0_spread,
                         double scale_decades, int n_zero_cols)
{
  synth_t* S = calloc(1, sizeof(*S));
  S->kind = 0; S->Nc = Nc; S->Np = Np; S->Nobs = Nobs; S->g = g; S->seed = seed;
  S->eps = eps; S->noise = noise; S->p0_spread = p0_spread;
  S->N = g + BC*Nc + BP*Np;
  S->M = 2*Nobs;
  const int k_row = g + BC + BP;
  S->nnz = S->M * k_row;
  S->Jp = malloc(sizeof(int)*((size_t)S->M+1));
  S->Ji = malloc(sizeof(int)*(size_t)S->nnz);
  S->a  = malloc(sizeof(double)*(size_t)S->nnz);
  S->pstar    = malloc(sizeof(double)*(size_t)S->N);
  S->colscale = malloc(sizeof(double)*(size_t)S->N);
  for(int j = 0; j < S->N; j++)
  {
    S->pstar[j] = urand(seed, 1, (uint64_t)j);
    /* column scales spread log-uniformly over `scale_decades` decades */
    S->colscale[j] = (scale_decades > 0.0)
      ? pow(10.0, -scale_decades * 0.5*(urand(seed, 2, (uint64_t)j)+1.0)) : 1.0;
  }
  /* numerically-zero (but structurally present) columns force the lambda path */
  for(int t = 0; t < n_zero_cols; t++)
  {
    const int j = (int)((uint64_t)(mix64(seed*31 + (uint64_t)t) % (uint64_t)S->N));
    S->colscale[j] = 0.0;
  }
  const int K = (Nobs + Np - 1)/Np;           /* observations per point (max) */
  int q = 0;
  for(int o = 0; o < Nobs; o++)
  {
    const int pt  = o % Np;
    int cam = (int)(((long)pt*Nc)/Np) + o/Np - K/2;
    cam %= Nc; if(cam < 0) cam += Nc;
    for(int h = 0; h < 2; h++)
    {
      const int r = 2*o + h;
      S->Jp[r] = q;
      for(int j = 0; j < g;  j++) S->Ji[q++] = j;
      for(int j = 0; j < BC; j++) S->Ji[q++] = g + BC*cam + j;
      for(int j = 0; j < BP; j++) S->Ji[q++] = g + BC*Nc + BP*pt + j;
    }
  }
  S->Jp[S->M] = q;
  /* coefficients: globals are weak (they see every row), blocks O(1) */
  for(int r = 0; r < S->M; r++)
    for(int t = S->Jp[r]; t < S->Jp[r+1]; t++)
    {
      const int j = S->Ji[t];
      double v = urand(seed, 3, (uint64_t)t);
      if(j < g) v *= 0.05;
      S->a[t] = v * S->colscale[j];
    }
  return S;
}

synth_t* synth_dense_create(int M, int N, uint64_t seed, double eps, double noise, double p0_spread)
{
  synth_t* S = calloc(1, sizeof(*S));
  S->kind = 1; S->M = M; S->N = N; S->nnz = 0; S->seed = seed;
  S->eps = eps; S->noise = noise; S->p0_spread = p0_spread;
  S->pstar = malloc(sizeof(double)*(size_t)N);
  for(int j = 0; j < N; j++) S->pstar[j] = urand(seed, 1, (uint64_t)j);
  return S;
}

void synth_free(synth_t* S)
{
  if(!S) return;
  free(S->Jp); free(S->Ji); free(S->a); free(S->pstar); free(S->colscale); free(S);
}
int  synth_nstate(const synth_t* S) { return S->N; }
int  synth_nmeas (const synth_t* S) { return S->M; }
int  synth_nnz   (const synth_t* S) { return S->nnz; }
int  synth_neval (const synth_t* S) { return S->neval; }
void synth_pstar (const synth_t* S, double* out) { memcpy(out, S->pstar, sizeof(double)*(size_t)S->N); }
void synth_p0    (const synth_t* S, double* out)
{
  for(int j = 0; j < S->N; j++)
    out[j] = S->pstar[j] + S->p0_spread * urand(S->seed, 4, (uint64_t)j);
}
/* the coefficients a[nnz] of a ba problem (for the device twin, problems/device_problems.hip) */
void synth_coefs(const synth_t* S, double* out) { memcpy(out, S->a, sizeof(double)*(size_t)S->nnz); }
void synth_model(const synth_t* S, double* eps_noise2, uint64_t* seed)
{ eps_noise2[0] = S->eps; eps_noise2[1] = S->noise; *seed = S->seed; }
void synth_pattern(const synth_t* S, int* Jp, int* Ji)
{
  memcpy(Jp, S->Jp, sizeof(int)*((size_t)S->M+1));
  memcpy(Ji, S->Ji, sizeof(int)*(size_t)S->nnz);
}

/* values only: x[M], Jx[nnz] */
void synth_ba_eval(synth_t* S, const double* p, double* x, double* Jx)
{
  S->neval++;
  for(int r = 0; r < S->M; r++)
  {
    double u = 0.0;
    for(int t = S->Jp[r]; t < S->Jp[r+1]; t++) u += S->a[t]*(p[S->Ji[t]] - S->pstar[S->Ji[t]]);
    x[r] = u + S->eps*sin(u) - S->noise*urand(S->seed, 5, (uint64_t)r);
    const double d = 1.0 + S->eps*cos(u);
    for(int t = S->Jp[r]; t < S->Jp[r+1]; t++) Jx[t] = S->a[t]*d;
  }
}
/* dogleg_callback_t */
void synth_cb_sparse(const double* p, double* x, cholmod_sparse* Jt, void* cookie)
{
  synth_t* S = (synth_t*)cookie;
  memcpy(Jt->p, S->Jp, sizeof(int)*((size_t)S->M+1));
  memcpy(Jt->i, S->Ji, sizeof(int)*(size_t)S->nnz);
  synth_ba_eval(S, p, x, (double*)Jt->x);
}

static inline double dense_coef(const synth_t* S, int r, int j)
{
  return urand(S->seed, 3, (uint64_t)r*(uint64_t)S->N + (uint64_t)j) / sqrt((double)S->N);
}
/* dogleg_callback_dense_t */
void synth_cb_dense(const double* p, double* x, double* J, void* cookie)
{
  synth_t* S = (synth_t*)cookie;
  S->neval++;
  const int N = S->N;
  for(int r = 0; r < S->M; r++)
  {
    double* Jr = &J[(size_t)r*N];
    double u = 0.0;
    for(int j = 0; j < N; j++) { Jr[j] = dense_coef(S, r, j); u += Jr[j]*(p[j] - S->pstar[j]); }
    x[r] = u + S->eps*sin(u) - S->noise*urand(S->seed, 5, (uint64_t)r);
    const double d = 1.0 + S->eps*cos(u);
    for(int j = 0; j < N; j++) Jr[j] *= d;
  }
}
/* dogleg_callback_dense_products_t on the dense problem; cookie = synth_t*,
 * layout chosen by the two flags stored with synth_set_products_layout() */
static int g_prod_packed = 0, g_prod_upper = 0;
void synth_set_products_layout(int packed, int upper) { g_prod_packed = packed; g_prod_upper = upper; }
void synth_cb_products(const double* p, double* norm2x, double* xtJ, double* JtJ, void* cookie)
{
  synth_t* S = (synth_t*)cookie;
  S->neval++;
  const int N = S->N;
  const size_t size = g_prod_packed ? (size_t)N*(N+1)/2 : (size_t)N*N;
  double* Jr = malloc(sizeof(double)*(size_t)N);
  *norm2x = 0.0;
  memset(xtJ, 0, sizeof(double)*(size_t)N);
  memset(JtJ, 0, sizeof(double)*size);
  for(int r = 0; r < S->M; r++)
  {
    double u = 0.0;
    for(int j = 0; j < N; j++) { Jr[j] = dense_coef(S, r, j); u += Jr[j]*(p[j] - S->pstar[j]); }
    const double xr = u + S->eps*sin(u) - S->noise*urand(S->seed, 5, (uint64_t)r);
    const double d = 1.0 + S->eps*cos(u);
    for(int j = 0; j < N; j++) Jr[j] *= d;
    *norm2x += xr*xr;
    for(int j = 0; j < N; j++) xtJ[j] += xr*Jr[j];
    if(g_prod_packed && g_prod_upper)
    {
      size_t t = 0;
      for(int k = 0; k < N; k++) for(int l = k; l < N; l++, t++) JtJ[t] += Jr[k]*Jr[l];
    }
    else if(!g_prod_packed)
    {
      for(int k = 0; k < N; k++) for(int l = 0; l < N; l++) JtJ[(size_t)k*N + l] += Jr[k]*Jr[l];
    }
  }
  free(Jr);
}
