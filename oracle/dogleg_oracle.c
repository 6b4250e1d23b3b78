/* dogleg_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See dogleg_oracle.h for scope and pinning status.  Every function names the
 * reference lines (in /root/reference/dogleg.c unless stated) it restates.
 * Single-threaded on purpose: the reference has no threads.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "dogleg_oracle.h"

#define ORC_SAY(...) do { fprintf(stderr, "dogleg-oracle: " __VA_ARGS__); fprintf(stderr, "\n"); } while(0)

/* ======================================================================== */
/* primitives                                                                */
/* ======================================================================== */

/* dogleg.c:190-196 */
double orc_norm2(const double* x, unsigned int n)
{
  double acc = 0;
  for(unsigned int i = 0; i < n; i++) acc += x[i]*x[i];
  return acc;
}
/* dogleg.c:197-203 */
double orc_inner(const double* x, const double* y, unsigned int n)
{
  double acc = 0;
  for(unsigned int i = 0; i < n; i++) acc += x[i]*y[i];
  return acc;
}
/* dogleg.c:249-261: column r of Jt scaled by x[r], scattered into dest */
void orc_spmv_Jt_x(double* dest, int nrow, int ncol, const int* Jp, const int* Ji,
                   const double* Jx, const double* x)
{
  memset(dest, 0, sizeof(double)*(size_t)nrow);
  for(int r = 0; r < ncol; r++)
    for(int q = Jp[r]; q < Jp[r+1]; q++)
      dest[Ji[q]] += x[r] * Jx[q];
}
/* dogleg.c:262-281 */
double orc_norm2_J_v(int ncol, const int* Jp, const int* Ji, const double* Jx, const double* v)
{
  double acc = 0.0;
  for(int r = 0; r < ncol; r++)
  {
    double dot = 0.0;
    for(int q = Jp[r]; q < Jp[r+1]; q++) dot += v[Ji[q]] * Jx[q];
    acc += dot*dot;
  }
  return acc;
}
/* dogleg.c:284-292 (column-outer loop order kept: it fixes the summation order) */
void orc_dense_Jt_x(double* dest, const double* J, const double* x, int Nrows, int Ncols)
{
  memset(dest, 0, sizeof(double)*(size_t)Ncols);
  for(int c = 0; c < Ncols; c++)
    for(int r = 0; r < Nrows; r++)
      dest[c] += J[c + (size_t)r*Ncols] * x[r];
}
/* dogleg.c:293-306 */
double orc_dense_norm2_J_v(const double* J, const double* v, int Nrows, int Ncols)
{
  double acc = 0.0;
  for(int r = 0; r < Nrows; r++)
  {
    double dot = orc_inner(v, &J[(size_t)r*Ncols], Ncols);
    acc += dot*dot;
  }
  return acc;
}
/* dogleg.c:309-332 */
double orc_xt_Apacked_upper_x(const double* x, const double* A, int N)
{
  double s = 0.0;
  int k = 0;
  for(int i = 0; i < N; i++)
  {
    s += A[k++] * x[i] * x[i];
    for(int j = i+1; j < N; j++, k++)
      s += 2. * A[k] * x[j] * x[i];
  }
  return s;
}
/* dogleg.c:335-347 */
double orc_xt_A_x(const double* x, const double* A, int N)
{
  double s = 0.0;
  for(int i = 0; i < N; i++)
    for(int j = 0; j < N; j++)
      s += A[i*N + j] * x[i] * x[j];
  return s;
}
/* dogleg.c:214-220 applied to every row, as at dogleg.c:712-714 */
void orc_dense_JtJ_packed_upper(double* JtJ, const double* J, int Nrows, int Ncols)
{
  for(int r = 0; r < Nrows; r++)
  {
    const double* j = &J[(size_t)r*Ncols];
    size_t k = 0;
    for(int i1 = 0; i1 < Ncols; i1++)
      for(int i0 = i1; i0 < Ncols; i0++, k++)
        JtJ[k] += j[i0]*j[i1];
  }
}

/* ------------------------------------------------------------------------ */
/* LAPACK restatements (reference netlib algorithms; the reference links      */
/* -llapack: dogleg.c:621-631, Makefile:23)                                   */
/* ------------------------------------------------------------------------ */

/* DPPTRF uplo='L' on a column-major packed lower triangle (the reference's
 * row-major packed upper is the same bytes: comment at dogleg.c:788-790).
 * call site dogleg.c:782 */
int orc_dpptrf_L(int n, double* ap)
{
  size_t jj = 0;
  for(int j = 0; j < n; j++)
  {
    double ajj = ap[jj];
    if(ajj <= 0.0) return j+1;
    ajj = sqrt(ajj);
    ap[jj] = ajj;
    const int m = n-j-1;
    if(m > 0)
    {
      double* col = &ap[jj+1];
      const double inv = 1.0/ajj;
      for(int i = 0; i < m; i++) col[i] *= inv;
      /* DSPR lower, alpha=-1 on the trailing packed block */
      double* trail = &ap[jj + m + 1];
      size_t kk = 0;
      for(int c = 0; c < m; c++)
      {
        if(col[c] != 0.0)
        {
          const double t = -col[c];
          for(int i = c; i < m; i++) trail[kk + (i-c)] += col[i]*t;
        }
        kk += m-c;
      }
    }
    jj += m+1;
  }
  return 0;
}
/* DPPTRS uplo='L', nrhs=1: L y = b then L' x = y.  call site dogleg.c:875 */
void orc_dpptrs_L(int n, const double* ap, double* b)
{
  /* forward, DTPSV('L','N','N') */
  size_t kk = 0;
  for(int j = 0; j < n; j++)
  {
    if(b[j] != 0.0)
    {
      b[j] /= ap[kk];
      const double t = b[j];
      for(int i = j+1; i < n; i++) b[i] -= t*ap[kk + (i-j)];
    }
    kk += n-j;
  }
  /* backward, DTPSV('L','T','N') */
  kk = (size_t)n*(n+1)/2;
  for(int j = n-1; j >= 0; j--)
  {
    double t = b[j];
    kk -= n-j;                     /* start of column j */
    for(int i = n-1; i > j; i--) t -= ap[kk + (i-j)]*b[i];
    b[j] = t / ap[kk];
  }
}
/* DPOTF2 uplo='L' (column-major, lda). call site dogleg.c:801 */
int orc_dpotrf_L(int n, double* a, int lda)
{
#define A_(i,j) a[(size_t)(j)*lda + (i)]
  for(int j = 0; j < n; j++)
  {
    double ajj = A_(j,j);
    for(int k = 0; k < j; k++) ajj -= A_(j,k)*A_(j,k);
    if(ajj <= 0.0 || ajj != ajj) { A_(j,j) = ajj; return j+1; }
    ajj = sqrt(ajj);
    A_(j,j) = ajj;
    for(int i = j+1; i < n; i++)
    {
      double s = A_(i,j);
      for(int k = 0; k < j; k++) s -= A_(i,k)*A_(j,k);
      A_(i,j) = s/ajj;
    }
  }
  return 0;
}
/* DPOTRS uplo='L', nrhs=1.  call site dogleg.c:889 */
void orc_dpotrs_L(int n, const double* a, int lda, double* b)
{
  for(int j = 0; j < n; j++)
  {
    if(b[j] != 0.0)
    {
      b[j] /= A_(j,j);
      const double t = b[j];
      for(int i = j+1; i < n; i++) b[i] -= t*A_(i,j);
    }
  }
  for(int j = n-1; j >= 0; j--)
  {
    double t = b[j];
    for(int i = n-1; i > j; i--) t -= A_(i,j)*b[i];
    b[j] = t / A_(j,j);
  }
#undef A_
}

/* ======================================================================== */
/* sparse Cholesky of  Jt*J + beta*I   (stand-in for CHOLMOD, third-party,    */
/* absent from /root/reference; call sites dogleg.c:652,659,663,853).         */
/* Published algorithm restated: fill-reducing permutation, elimination tree, */
/* up-looking simplicial numeric factorisation where row k of A*A' is formed  */
/* on the fly from A and its transpose (CHOLMOD "rowfac" for an unsymmetric   */
/* A; Davis, "Direct Methods for Sparse Linear Systems", ch. 4).  LL' form:   */
/* a non-positive pivot reports minor = k (the reference only ever tests      */
/* minor == n, dogleg.c:667).                                                 */
/* ======================================================================== */
struct orc_sparse_factor
{
  int   n, m;
  int  *perm, *iperm;        /* perm[k] = original variable eliminated k-th      */
  int  *parent;              /* elimination tree (permuted indices)              */
  int  *Rp, *Rr, *Rq;        /* variable-major view of Jt: for variable i the    */
                             /* measurement rows Rr[] and positions Rq[] in Jx   */
  long *Lp; int *Li; double *Lx; int *Lnz;
  double flops;
  /* work */
  double *w; int *stack, *flag;
};

static int cmp_deg(const void* a, const void* b)
{
  const long* x = (const long*)a; const long* y = (const long*)b;
  if(x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
  return x[1] < y[1] ? -1 : (x[1] > y[1]);
}

/* pattern of row k (permuted) of C = P Jt J P', entries i < k, via marker.
 * out[] receives the permuted indices; returns the count */
static int row_pattern(const orc_sparse_factor_t* F, const int* Jp, const int* Ji,
                       int k, int* mark, int* out)
{
  int cnt = 0;
  const int v = F->perm[k];
  for(int a = F->Rp[v]; a < F->Rp[v+1]; a++)
  {
    const int r = F->Rr[a];
    for(int q = Jp[r]; q < Jp[r+1]; q++)
    {
      const int i = F->iperm[Ji[q]];
      if(i < k && mark[i] != k) { mark[i] = k; out[cnt++] = i; }
    }
  }
  return cnt;
}

orc_sparse_factor_t* orc_sparse_analyze(int n, int m, const int* Jp, const int* Ji)
{
  orc_sparse_factor_t* F = calloc(1, sizeof(*F));
  F->n = n; F->m = m;
  const int nnz = Jp[m];

  /* variable-major view */
  F->Rp = calloc((size_t)n+1, sizeof(int));
  F->Rr = malloc(sizeof(int)*(size_t)(nnz > 0 ? nnz : 1));
  F->Rq = malloc(sizeof(int)*(size_t)(nnz > 0 ? nnz : 1));
  for(int q = 0; q < nnz; q++) F->Rp[Ji[q]+1]++;
  for(int i = 0; i < n; i++) F->Rp[i+1] += F->Rp[i];
  {
    int* next = malloc(sizeof(int)*(size_t)(n > 0 ? n : 1));
    memcpy(next, F->Rp, sizeof(int)*(size_t)n);
    for(int r = 0; r < m; r++)
      for(int q = Jp[r]; q < Jp[r+1]; q++)
      { const int i = Ji[q]; F->Rr[next[i]] = r; F->Rq[next[i]] = q; next[i]++; }
    free(next);
  }

  /* fill-reducing permutation: variables sorted by their degree in the graph
   * of JtJ (a static minimum-degree ordering; CHOLMOD would run AMD/COLAMD --
   * any permutation gives the same solution up to rounding) */
  F->perm  = malloc(sizeof(int)*(size_t)(n > 0 ? n : 1));
  F->iperm = malloc(sizeof(int)*(size_t)(n > 0 ? n : 1));
  int* mark = malloc(sizeof(int)*(size_t)(n > 0 ? n : 1));
  {
    long* key = malloc(sizeof(long)*2*(size_t)(n > 0 ? n : 1));
    for(int i = 0; i < n; i++) mark[i] = -1;
    for(int v = 0; v < n; v++)
    {
      long deg = 0;
      for(int a = F->Rp[v]; a < F->Rp[v+1]; a++)
      {
        const int r = F->Rr[a];
        for(int q = Jp[r]; q < Jp[r+1]; q++)
          if(mark[Ji[q]] != v) { mark[Ji[q]] = v; deg++; }
      }
      key[2*v] = deg; key[2*v+1] = v;
    }
    qsort(key, (size_t)n, 2*sizeof(long), cmp_deg);
    for(int k = 0; k < n; k++) { F->perm[k] = (int)key[2*k+1]; F->iperm[F->perm[k]] = k; }
    free(key);
  }

  /* elimination tree of C (Liu's algorithm with path compression) */
  F->parent = malloc(sizeof(int)*(size_t)(n > 0 ? n : 1));
  int* anc  = malloc(sizeof(int)*(size_t)(n > 0 ? n : 1));
  int* pat  = malloc(sizeof(int)*(size_t)(n > 0 ? n : 1));
  for(int i = 0; i < n; i++) mark[i] = -1;
  for(int k = 0; k < n; k++)
  {
    F->parent[k] = -1; anc[k] = -1;
    const int cnt = row_pattern(F, Jp, Ji, k, mark, pat);
    for(int t = 0; t < cnt; t++)
    {
      int i = pat[t];
      while(i != -1 && i < k)
      {
        const int nxt = anc[i];
        anc[i] = k;
        if(nxt == -1) F->parent[i] = k;
        i = nxt;
      }
    }
  }

  /* column counts by a symbolic up-looking sweep */
  F->Lnz = calloc((size_t)(n > 0 ? n : 1), sizeof(int));
  F->flag = malloc(sizeof(int)*(size_t)(n > 0 ? n : 1));
  for(int i = 0; i < n; i++) { mark[i] = -1; F->flag[i] = -1; }
  for(int k = 0; k < n; k++)
  {
    const int cnt = row_pattern(F, Jp, Ji, k, mark, pat);
    F->flag[k] = k;
    F->Lnz[k]++;                               /* diagonal */
    for(int t = 0; t < cnt; t++)
      for(int i = pat[t]; F->flag[i] != k; i = F->parent[i])
      { F->flag[i] = k; F->Lnz[i]++; }
  }
  F->Lp = malloc(sizeof(long)*((size_t)n+1));
  F->Lp[0] = 0;
  F->flops = 0;
  for(int k = 0; k < n; k++)
  {
    F->Lp[k+1] = F->Lp[k] + F->Lnz[k];
    F->flops += (double)F->Lnz[k]*(double)F->Lnz[k];
  }
  F->Li = malloc(sizeof(int)   *(size_t)(F->Lp[n] > 0 ? F->Lp[n] : 1));
  F->Lx = malloc(sizeof(double)*(size_t)(F->Lp[n] > 0 ? F->Lp[n] : 1));
  F->w     = calloc((size_t)(n > 0 ? n : 1), sizeof(double));
  F->stack = malloc(sizeof(int)*(size_t)(n > 0 ? n : 1));
  free(anc); free(pat); free(mark);
  return F;
}

long orc_sparse_factorize(orc_sparse_factor_t* F, const int* Jp, const int* Ji,
                          const double* Jx, double beta)
{
  const int n = F->n;
  double* w = F->w;
  int* s = F->stack;
  int* flag = F->flag;
  for(int i = 0; i < n; i++) { F->Lnz[i] = 0; flag[i] = -1; w[i] = 0.0; }

  for(int k = 0; k < n; k++)
  {
    /* numeric row k of C into w[], and its etree reach into s[top..n) */
    int top = n;
    flag[k] = k;
    const int v = F->perm[k];
    double d = beta;
    for(int a = F->Rp[v]; a < F->Rp[v+1]; a++)
    {
      const int r = F->Rr[a];
      const double vk = Jx[F->Rq[a]];
      for(int q = Jp[r]; q < Jp[r+1]; q++)
      {
        const int i = F->iperm[Ji[q]];
        if(i > k) continue;
        if(i == k) { d += vk*Jx[q]; continue; }
        w[i] += vk*Jx[q];
        /* walk up the etree until a flagged node; push the path */
        int len = 0, j = i;
        while(flag[j] != k) { s[len++] = j; flag[j] = k; j = F->parent[j]; }
        while(len > 0) s[--top] = s[--len];
      }
    }
    /* sparse triangular solve along the reach (topological order) */
    for(; top < n; top++)
    {
      const int i = s[top];
      const long c0 = F->Lp[i];
      const double lki = w[i] / F->Lx[c0];
      w[i] = 0.0;
      const long c1 = c0 + F->Lnz[i];
      for(long q = c0+1; q < c1; q++) w[F->Li[q]] -= F->Lx[q]*lki;
      d -= lki*lki;
      F->Li[c1] = k; F->Lx[c1] = lki; F->Lnz[i]++;
    }
    if(d <= 0.0 || d != d)
    {
      /* not positive definite: clear the work vector and report */
      for(int i = 0; i < n; i++) w[i] = 0.0;
      return k;
    }
    const long ck = F->Lp[k];
    F->Li[ck] = k; F->Lx[ck] = sqrt(d); F->Lnz[k] = 1;
  }
  return n;
}

/* x = P' L^-T L^-1 P b   (cholmod_solve(CHOLMOD_A), dogleg.c:853) */
void orc_sparse_solve(const orc_sparse_factor_t* F, const double* b, double* x)
{
  const int n = F->n;
  double* y = malloc(sizeof(double)*(size_t)(n > 0 ? n : 1));
  for(int k = 0; k < n; k++) y[k] = b[F->perm[k]];
  for(int j = 0; j < n; j++)
  {
    const long c0 = F->Lp[j], c1 = F->Lp[j+1];
    y[j] /= F->Lx[c0];
    for(long q = c0+1; q < c1; q++) y[F->Li[q]] -= F->Lx[q]*y[j];
  }
  for(int j = n-1; j >= 0; j--)
  {
    const long c0 = F->Lp[j], c1 = F->Lp[j+1];
    for(long q = c0+1; q < c1; q++) y[j] -= F->Lx[q]*y[F->Li[q]];
    y[j] /= F->Lx[c0];
  }
  for(int k = 0; k < n; k++) x[F->perm[k]] = y[k];
  free(y);
}
long   orc_sparse_nnzL (const orc_sparse_factor_t* F) { return F->Lp[F->n]; }
double orc_sparse_flops(const orc_sparse_factor_t* F) { return F->flops; }
void orc_sparse_free(orc_sparse_factor_t* F)
{
  if(!F) return;
  free(F->perm); free(F->iperm); free(F->parent); free(F->Rp); free(F->Rr); free(F->Rq);
  free(F->Lp); free(F->Li); free(F->Lx); free(F->Lnz); free(F->w); free(F->stack); free(F->flag);
  free(F);
}

/* ======================================================================== */
/* solver state                                                              */
/* ======================================================================== */
typedef struct
{
  double *p, *x, *Jt_x, *updateCauchy, *updateGN, *step_to_here;
  /* sparse */ int *Jp, *Ji; double *Jx; cholmod_sparse Jt_view;
  /* dense  */ double *J_dense;
  /* prods  */ double *JtJ;
  double norm2_x, norm2_updateCauchy, norm2_updateGN, norm2_step_to_here;
  int have_updateCauchy, have_updateGN, have_factorization, have_step_to_here;
  int didStepToEdge;
} orc_point_t;

typedef struct
{
  dogleg_solve_type_t type;
  int N, M, nnz;
  dogleg_callback_t*                f;
  dogleg_callback_dense_t*          f_dense;
  dogleg_callback_dense_products_t* f_products;
  void* cookie;
  const dogleg_parameters2_t* prm;
  orc_point_t *before, *after;
  orc_sparse_factor_t* sfac;       /* sparse */
  double* dfac;                    /* dense / products */
  double lambda;
  dlg_trace_t* trace;
  dlg_trial_t  cur;                /* record being assembled */
  int ncallbacks;
} orc_ctx_t;

/* dogleg.c:117-128 */
void orc_default_parameters(dogleg_parameters2_t* q)
{
  memset(q, 0, sizeof(*q));
  q->max_iterations                 = 100;
  q->trustregion0                   = 1.0e3;
  q->trustregion_decrease_factor    = 0.1;
  q->trustregion_decrease_threshold = 0.25;
  q->trustregion_increase_factor    = 2;
  q->trustregion_increase_threshold = 0.75;
  q->Jt_x_threshold                 = 1e-8;
  q->update_threshold               = 1e-8;
  q->trustregion_threshold          = 1e-8;
}

static size_t jtj_size(const orc_ctx_t* c)
{
  const size_t N = (size_t)c->N;
  return (c->type == DOGLEG_DENSE || c->prm->JtJ_packed) ? N*(N+1)/2 : N*N;
}

/* dogleg.c:1479-1562 */
static orc_point_t* point_alloc(const orc_ctx_t* c)
{
  orc_point_t* pt = calloc(1, sizeof(*pt));
  const size_t N = (size_t)c->N, M = (size_t)c->M;
  pt->p            = calloc(N, sizeof(double));
  pt->Jt_x         = calloc(N, sizeof(double));
  pt->updateCauchy = calloc(N, sizeof(double));
  pt->updateGN     = calloc(N, sizeof(double));
  pt->step_to_here = calloc(N, sizeof(double));
  if(c->type != DOGLEG_DENSE_PRODUCTS) pt->x = calloc(M > 0 ? M : 1, sizeof(double));
  if(c->type == DOGLEG_SPARSE)
  {
    pt->Jp = calloc(M+1, sizeof(int));
    pt->Ji = calloc((size_t)c->nnz, sizeof(int));
    pt->Jx = calloc((size_t)c->nnz, sizeof(double));
    cholmod_sparse* v = &pt->Jt_view;
    memset(v, 0, sizeof(*v));
    v->nrow = N; v->ncol = M; v->nzmax = (size_t)c->nnz;
    v->p = pt->Jp; v->i = pt->Ji; v->x = pt->Jx;
    v->stype = 0; v->itype = CHOLMOD_INT; v->xtype = CHOLMOD_REAL; v->dtype = CHOLMOD_DOUBLE;
    v->sorted = 1; v->packed = 1;
  }
  else if(c->type == DOGLEG_DENSE)
    pt->J_dense = calloc(M*N > 0 ? M*N : 1, sizeof(double));
  else
    pt->JtJ = calloc(jtj_size(c), sizeof(double));
  return pt;
}
static void point_free(orc_point_t* pt)
{
  if(!pt) return;
  free(pt->p); free(pt->x); free(pt->Jt_x); free(pt->updateCauchy); free(pt->updateGN);
  free(pt->step_to_here); free(pt->Jp); free(pt->Ji); free(pt->Jx); free(pt->J_dense); free(pt->JtJ);
  free(pt);
}

/* dogleg.c:1004-1083 */
static int eval_point(int* converged, orc_point_t* pt, orc_ctx_t* c)
{
  pt->norm2_x = -1.;
  pt->have_updateCauchy = pt->have_updateGN = pt->have_factorization = 0;
  pt->have_step_to_here = 0; pt->didStepToEdge = 0;       /* memset of all bits, :1012 */
  c->ncallbacks++;
  switch(c->type)
  {
  case DOGLEG_SPARSE:
    c->f(pt->p, pt->x, &pt->Jt_view, c->cookie);
    orc_spmv_Jt_x(pt->Jt_x, c->N, c->M, pt->Jp, pt->Ji, pt->Jx, pt->x);
    pt->norm2_x = orc_norm2(pt->x, c->M);
    break;
  case DOGLEG_DENSE:
    c->f_dense(pt->p, pt->x, pt->J_dense, c->cookie);
    orc_dense_Jt_x(pt->Jt_x, pt->J_dense, pt->x, c->M, c->N);
    pt->norm2_x = orc_norm2(pt->x, c->M);
    break;
  default:
    c->f_products(pt->p, &pt->norm2_x, pt->Jt_x, pt->JtJ, c->cookie);
    break;
  }
  for(int i = 0; i < c->N; i++)
    if(fabs(pt->Jt_x[i]) > c->prm->Jt_x_threshold) { *converged = 0; return 1; }
  *converged = 1;
  return 1;
}

/* norm2(J v) in whichever representation the solve type holds
 * (dogleg.c:558-603 and 1095-1163 share this switch) */
static int norm2_Jv(double* out, const orc_point_t* pt, const double* v, const orc_ctx_t* c)
{
  switch(c->type)
  {
  case DOGLEG_SPARSE: *out = orc_norm2_J_v(c->M, pt->Jp, pt->Ji, pt->Jx, v); return 1;
  case DOGLEG_DENSE:  *out = orc_dense_norm2_J_v(pt->J_dense, v, c->M, c->N); return 1;
  default:
    if(c->prm->JtJ_packed && c->prm->JtJ_upper) { *out = orc_xt_Apacked_upper_x(v, pt->JtJ, c->N); return 1; }
    if(!c->prm->JtJ_packed)                     { *out = orc_xt_A_x(v, pt->JtJ, c->N); return 1; }
    ORC_SAY("only JtJ unpacked || (packed,upper) is supported");   /* :599, :1158 */
    return 0;
  }
}

/* dogleg.c:529-617 */
static int compute_cauchy(orc_point_t* pt, orc_ctx_t* c)
{
  if(!pt->have_updateCauchy)
  {
    pt->have_updateCauchy = 1;
    const double g2 = orc_norm2(pt->Jt_x, c->N);
    double Jg2;
    if(!norm2_Jv(&Jg2, pt, pt->Jt_x, c)) return 0;
    const double k = -g2 / Jg2;
    pt->norm2_updateCauchy = k*k * g2;
    for(int i = 0; i < c->N; i++) pt->updateCauchy[i] = k * pt->Jt_x[i];
  }
  c->cur.norm2_cauchy = pt->norm2_updateCauchy;
  return 1;
}

/* dogleg.c:634-820 */
static int compute_factorization(orc_point_t* pt, orc_ctx_t* c)
{
  if(pt->have_factorization) return 1;
  const int N = c->N;
  if(c->type == DOGLEG_SPARSE)
  {
    if(c->sfac == NULL) c->sfac = orc_sparse_analyze(N, c->M, pt->Jp, pt->Ji);   /* :650-654 */
    while(1)
    {
      const long minor = orc_sparse_factorize(c->sfac, pt->Jp, pt->Ji, pt->Jx, c->lambda);
      if(minor == N) break;                                                       /* :667 */
      c->lambda = (c->lambda == 0.0) ? 1e-10 : c->lambda*10.0;                    /* :671-672 */
      if(!isfinite(c->lambda)) { ORC_SAY("lambda overflow"); return 0; }
    }
  }
  else
  {
    while(1)
    {
      int info;
      if(c->type == DOGLEG_DENSE)
      {
        memset(c->dfac, 0, sizeof(double)*(size_t)N*(N+1)/2);                     /* :709-711 */
        orc_dense_JtJ_packed_upper(c->dfac, pt->J_dense, c->M, N);                /* :712-714 */
        if(c->lambda > 0.0)                                                       /* :715-723 */
        {
          size_t k = 0;
          for(int i1 = 0; i1 < N; i1++) { c->dfac[k] += c->lambda; k += N-i1; }
        }
        info = orc_dpptrf_L(N, c->dfac);                                          /* :782 */
      }
      else
      {
        memcpy(c->dfac, pt->JtJ, sizeof(double)*jtj_size(c));                     /* :746-748 */
        if(c->lambda > 0.0)
        {
          if(c->prm->JtJ_packed)
          {
            size_t k = 0;
            for(int i1 = 0; i1 < N; i1++)
            { c->dfac[k] += c->lambda; k += c->prm->JtJ_upper ? (size_t)(N-i1) : (size_t)(i1+2); }
          }
          else
            for(int i1 = 0; i1 < N; i1++) c->dfac[(size_t)i1*(N+1)] += c->lambda;
        }
        if(c->prm->JtJ_packed)
        {
          if(!c->prm->JtJ_upper) { ORC_SAY("packed-lower JtJ unsupported"); return 0; }
          info = orc_dpptrf_L(N, c->dfac);                                        /* :787-795 */
        }
        else
          info = orc_dpotrf_L(N, c->dfac, N);                                     /* :801 */
      }
      if(info == 0) break;
      c->lambda = (c->lambda == 0.0) ? 1e-10 : c->lambda*10.0;                    /* :812-813 */
      if(!isfinite(c->lambda)) { ORC_SAY("lambda overflow"); return 0; }
    }
  }
  pt->have_factorization = 1;
  return 1;
}

/* dogleg.c:822-908 */
static int compute_gn(orc_point_t* pt, orc_ctx_t* c)
{
  if(!pt->have_updateGN)
  {
    if(!compute_factorization(pt, c)) return 0;
    const int N = c->N;
    if(c->type == DOGLEG_SPARSE)
      orc_sparse_solve(c->sfac, pt->Jt_x, pt->updateGN);
    else
    {
      memcpy(pt->updateGN, pt->Jt_x, sizeof(double)*(size_t)N);
      if(c->type == DOGLEG_DENSE || (c->prm->JtJ_packed && c->prm->JtJ_upper))
        orc_dpptrs_L(N, c->dfac, pt->updateGN);
      else if(!c->prm->JtJ_packed)
        orc_dpotrs_L(N, c->dfac, N, pt->updateGN);
      else { ORC_SAY("packed-lower JtJ unsupported"); return 0; }
    }
    for(int i = 0; i < N; i++) pt->updateGN[i] *= -1.0;
    pt->norm2_updateGN = orc_norm2(pt->updateGN, N);
    pt->have_updateGN = 1;
  }
  c->cur.norm2_gn = pt->norm2_updateGN;
  return 1;
}

/* dogleg.c:927-998 */
static int compute_interpolated(double* step, double* norm2_step, orc_point_t* pt,
                                double trustregion, orc_ctx_t* c)
{
  const double dsq = trustregion*trustregion;
  const double norm2a = pt->norm2_updateCauchy;
  const double *a = pt->updateCauchy, *b = pt->updateGN;
  double l2 = 0.0, neg_c = 0.0;
  for(int i = 0; i < c->N; i++)
  {
    const double d = a[i] - b[i];
    l2    += d*d;
    neg_c += d*a[i];
  }
  double disc = neg_c*neg_c - l2*(norm2a - dsq);
  if(disc < 0.0) disc = 0.0;
  const double k = (neg_c + sqrt(disc))/l2;
  *norm2_step = 0.0;
  for(int i = 0; i < c->N; i++)
  {
    step[i] = a[i] + k*(b[i] - a[i]);
    *norm2_step += step[i]*step[i];
  }
  c->cur.k_cauchy_to_gn = k;
  return 1;
}

/* dogleg.c:1085-1165 */
static int expected_improvement(double* out, const double* step, const orc_point_t* pt,
                                const orc_ctx_t* c)
{
  double Js2;
  if(!norm2_Jv(&Js2, pt, step, c)) return 0;
  *out = -2.0*orc_inner(pt->Jt_x, step, c->N) - Js2;
  return 1;
}

/* dogleg.c:1172-1297 */
static int take_step(double* expectedImprovement, double* p_new, double* step,
                     double* norm2_step, orc_point_t* from, double trustregion, orc_ctx_t* c)
{
  c->cur.trustregion_before = trustregion;
  c->cur.norm2x_before      = from->norm2_x;
  if(!compute_cauchy(from, c)) return 0;

  if(from->norm2_updateCauchy >= trustregion*trustregion)
  {
    c->cur.step_type = DLG_STEP_CAUCHY;
    *norm2_step = from->norm2_updateCauchy;                       /* unscaled, :1200 */
    const double s = trustregion / sqrt(from->norm2_updateCauchy);
    for(int i = 0; i < c->N; i++) step[i] = s * from->updateCauchy[i];
    from->didStepToEdge = 1;
  }
  else
  {
    if(!compute_gn(from, c)) return 0;
    if(from->norm2_updateGN <= trustregion*trustregion)
    {
      c->cur.step_type = DLG_STEP_GAUSSNEWTON;
      *norm2_step = from->norm2_updateGN;
      memcpy(step, from->updateGN, sizeof(double)*(size_t)c->N);
      from->didStepToEdge = 0;
    }
    else
    {
      if(!compute_interpolated(step, norm2_step, from, trustregion, c)) return 0;
      from->didStepToEdge = 1;
      c->cur.step_type = DLG_STEP_INTERPOLATED;
    }
  }
  for(int i = 0; i < c->N; i++) p_new[i] = from->p[i] + step[i];
  if(!expected_improvement(expectedImprovement, step, from, c)) return 0;
  /* the diagnostics record the computed value (dogleg.c:1267-1269), also for the terminal step
     whose return value is replaced by -1 below (dogleg.c:1289-1296) */
  c->cur.expected_improvement = *expectedImprovement;
  c->cur.norm2_step = *norm2_step;
  c->cur.did_step_to_edge = from->didStepToEdge;

  for(int i = 0; i < c->N; i++)
    if(fabs(step[i]) > c->prm->update_threshold) return 1;
  *expectedImprovement = -1.0;
  return 1;
}

/* dogleg.c:1303-1356 */
static int evaluate_step(int* accept, double* trustregion, const orc_point_t* before,
                         const orc_point_t* after, double expectedImprovement, orc_ctx_t* c)
{
  const double observed = before->norm2_x - after->norm2_x;
  const double rho = observed / expectedImprovement;
  c->cur.observed_improvement = observed;
  c->cur.rho = rho;
  if(rho < c->prm->trustregion_decrease_threshold)
  {
    if(!before->didStepToEdge)
    {
      if(!before->have_updateGN) { ORC_SAY("updateGN missing: bug"); return 0; }
      *trustregion = sqrt(before->norm2_updateGN);
    }
    *trustregion *= c->prm->trustregion_decrease_factor;
  }
  else if(rho > c->prm->trustregion_increase_threshold && before->didStepToEdge)
    *trustregion *= c->prm->trustregion_increase_factor;
  c->cur.trustregion_after = *trustregion;
  *accept = (rho > 0.0);
  return 1;
}

static void trace_reset(orc_ctx_t* c)
{
  dlg_trial_t* t = &c->cur;
  memset(t, 0, sizeof(*t));
  t->norm2x_after = t->norm2_cauchy = t->norm2_gn = t->k_cauchy_to_gn = NAN;
  t->observed_improvement = t->rho = t->trustregion_after = NAN;
}
static void trace_emit(orc_ctx_t* c, int iteration, int accepted)
{
  dlg_trace_t* tr = c->trace;
  c->cur.iteration = iteration;
  c->cur.accepted  = accepted;
  c->cur.lambda    = c->lambda;
  if(tr)
  {
    if(tr->ntrials < tr->capacity)
    {
      tr->trials[tr->ntrials] = c->cur;
      if(tr->p_trial) memcpy(&tr->p_trial[(size_t)tr->ntrials*c->N], c->after->p,            sizeof(double)*(size_t)c->N);
      if(tr->step)    memcpy(&tr->step   [(size_t)tr->ntrials*c->N], c->after->step_to_here, sizeof(double)*(size_t)c->N);
    }
    tr->ntrials++;
  }
  trace_reset(c);
}

/* dogleg.c:1359-1476 */
static int run_optimizer(orc_ctx_t* c)
{
  double trustregion = c->prm->trustregion0;
  int stepCount = 0;
  int converged;
  trace_reset(c);
  if(!eval_point(&converged, c->before, c)) return -1;
  if(converged) return stepCount;

  while(stepCount < c->prm->max_iterations)
  {
    while(1)
    {
      c->after->have_step_to_here = 0;
      double expectedImprovement;
      if(!take_step(&expectedImprovement, c->after->p, c->after->step_to_here,
                    &c->after->norm2_step_to_here, c->before, trustregion, c))
        return -1;
      c->after->have_step_to_here = 1;

      if(expectedImprovement < 0.0) { trace_emit(c, stepCount, 2); return stepCount; }   /* :1403-1408 */

      int afterZeroGradient;
      if(!eval_point(&afterZeroGradient, c->after, c)) return -1;
      /* eval_point clears the flags of `after`, including have_step_to_here
       * (the reference's memset at :1012 does the same) */
      c->cur.norm2x_after = c->after->norm2_x;

      int accept;
      if(!evaluate_step(&accept, &trustregion, c->before, c->after, expectedImprovement, c))
        return -1;

      if(accept)
      {
        trace_emit(c, stepCount, 1);
        stepCount++;
        orc_point_t* t = c->after; c->after = c->before; c->before = t;
        if(afterZeroGradient) return stepCount;
        break;
      }
      trace_emit(c, stepCount, 0);
      if(trustregion < c->prm->trustregion_threshold) return stepCount;
    }
  }
  return stepCount;
}

/* dogleg.c:1633-1753 */
static double optimize_common(double* p, unsigned int N, unsigned int M, unsigned int nnz,
                              dogleg_callback_t* f, dogleg_callback_dense_t* fd,
                              dogleg_callback_dense_products_t* fp, void* cookie,
                              const dogleg_parameters2_t* prm, dlg_trace_t* trace)
{
  static dogleg_parameters2_t defaults; static int defaults_set = 0;
  if(!defaults_set) { orc_default_parameters(&defaults); defaults_set = 1; }
  orc_ctx_t C; memset(&C, 0, sizeof(C));
  C.N = (int)N; C.M = (int)M; C.nnz = (int)nnz; C.cookie = cookie;
  C.prm = prm ? prm : &defaults; C.trace = trace;
  if(f)       { C.type = DOGLEG_SPARSE; C.f = f;         if(nnz == 0) return -1.0; }
  else if(fd) { C.type = DOGLEG_DENSE;  C.f_dense = fd;  if(nnz != 0) return -1.0; }
  else if(fp) { C.type = DOGLEG_DENSE_PRODUCTS; C.f_products = fp; if(nnz != 0) return -1.0; }
  else return -1.0;
  if(trace) { trace->ntrials = 0; trace->ncallbacks = 0; trace->nstate = (int)N; }

  if(C.type != DOGLEG_SPARSE) C.dfac = calloc(jtj_size(&C), sizeof(double));
  C.before = point_alloc(&C);
  C.after  = point_alloc(&C);
  memcpy(C.before->p, p, sizeof(double)*N);

  const int numsteps = run_optimizer(&C);
  double result = C.before->norm2_x;
  if(numsteps < 0) result = -1.0;
  else memcpy(p, C.before->p, sizeof(double)*N);
  if(trace) trace->ncallbacks = C.ncallbacks;

  point_free(C.before); point_free(C.after);
  orc_sparse_free(C.sfac); free(C.dfac);
  return result;
}

double orc_optimize_sparse(double* p, unsigned int Nstate, unsigned int Nmeas, unsigned int NJnnz,
                           dogleg_callback_t* f, void* cookie,
                           const dogleg_parameters2_t* parameters, dlg_trace_t* trace)
{
  if(NJnnz == 0) return -1.0;                                   /* :1762-1766 */
  return optimize_common(p, Nstate, Nmeas, NJnnz, f, NULL, NULL, cookie, parameters, trace);
}
double orc_optimize_dense(double* p, unsigned int Nstate, unsigned int Nmeas,
                          dogleg_callback_dense_t* f, void* cookie,
                          const dogleg_parameters2_t* parameters, dlg_trace_t* trace)
{
  return optimize_common(p, Nstate, Nmeas, 0, NULL, f, NULL, cookie, parameters, trace);
}
double orc_optimize_dense_products(double* p, unsigned int Nstate,
                                   dogleg_callback_dense_products_t* f, void* cookie,
                                   const dogleg_parameters2_t* parameters, dlg_trace_t* trace)
{
  return optimize_common(p, Nstate, 0, 0, NULL, NULL, f, cookie, parameters, trace);
}

/* ======================================================================== */
/* fixed-input trial step (CPU baseline unit for bench.py)                   */
/* ======================================================================== */
static void step_tail(int N, const double* g, const double* cauchy, double n2c, double* gn,
                      const double* p, double* step, double* pnew, double* out)
{
  for(int i = 0; i < N; i++) gn[i] *= -1.0;
  const double n2gn = orc_norm2(gn, N);
  const double tr = 0.5*(sqrt(n2c) + sqrt(n2gn));
  const double dsq = tr*tr;
  double l2 = 0.0, neg_c = 0.0;
  for(int i = 0; i < N; i++) { const double d = cauchy[i] - gn[i]; l2 += d*d; neg_c += d*cauchy[i]; }
  double disc = neg_c*neg_c - l2*(n2c - dsq);
  if(disc < 0.0) disc = 0.0;
  const double k = (neg_c + sqrt(disc))/l2;
  double n2s = 0.0, amax = 0.0, gmax = 0.0;
  for(int i = 0; i < N; i++)
  {
    step[i] = cauchy[i] + k*(gn[i] - cauchy[i]);
    n2s += step[i]*step[i];
    pnew[i] = p[i] + step[i];
    if(fabs(step[i]) > amax) amax = fabs(step[i]);
    if(fabs(g[i]) > gmax) gmax = fabs(g[i]);
  }
  out[2] = n2gn; out[3] = k; out[4] = n2s; out[6] = gmax; out[7] = amax;
}

int orc_step_sparse(orc_sparse_factor_t* F, int N, int M, const int* Jp, const int* Ji,
                    const double* Jx, const double* x, const double* p, double lambda,
                    double* work, double* out)
{
  double *g = work, *cauchy = work + N, *gn = work + 2*(size_t)N, *step = work + 3*(size_t)N,
         *pnew = work + 4*(size_t)N;
  orc_spmv_Jt_x(g, N, M, Jp, Ji, Jx, x);
  out[0] = orc_norm2(x, M);
  const double g2 = orc_norm2(g, N);
  const double Jg2 = orc_norm2_J_v(M, Jp, Ji, Jx, g);
  const double kc = -g2/Jg2;
  out[1] = kc*kc*g2;
  for(int i = 0; i < N; i++) cauchy[i] = kc*g[i];
  if(orc_sparse_factorize(F, Jp, Ji, Jx, lambda) != N) return 1;
  orc_sparse_solve(F, g, gn);
  step_tail(N, g, cauchy, out[1], gn, p, step, pnew, out);
  out[5] = -2.0*orc_inner(g, step, N) - orc_norm2_J_v(M, Jp, Ji, Jx, step);
  return 0;
}

int orc_step_dense(int N, int M, const double* J, const double* x, const double* p, double lambda,
                   double* dfac, double* work, double* out)
{
  double *g = work, *cauchy = work + N, *gn = work + 2*(size_t)N, *step = work + 3*(size_t)N,
         *pnew = work + 4*(size_t)N;
  orc_dense_Jt_x(g, J, x, M, N);
  out[0] = orc_norm2(x, M);
  const double g2 = orc_norm2(g, N);
  const double Jg2 = orc_dense_norm2_J_v(J, g, M, N);
  const double kc = -g2/Jg2;
  out[1] = kc*kc*g2;
  for(int i = 0; i < N; i++) cauchy[i] = kc*g[i];
  memset(dfac, 0, sizeof(double)*(size_t)N*(N+1)/2);
  orc_dense_JtJ_packed_upper(dfac, J, M, N);
  if(lambda > 0.0) { size_t k = 0; for(int i1 = 0; i1 < N; i1++) { dfac[k] += lambda; k += N-i1; } }
  if(orc_dpptrf_L(N, dfac) != 0) return 1;
  memcpy(gn, g, sizeof(double)*(size_t)N);
  orc_dpptrs_L(N, dfac, gn);
  step_tail(N, g, cauchy, out[1], gn, p, step, pnew, out);
  out[5] = -2.0*orc_inner(g, step, N) - orc_dense_norm2_J_v(J, step, M, N);
  return 0;
}
