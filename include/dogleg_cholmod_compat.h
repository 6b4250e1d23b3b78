/* dogleg_cholmod_compat.h
 *
 * The sparse callback of the dogleg API receives its Jacobian as a
 * `cholmod_sparse*` (reference: dogleg.h:11-20, sample.c:82-125 shows the
 * fields a callback touches: ->p, ->i, ->x reinterpreted as int/int/double
 * arrays).  This backend does not use CHOLMOD at all (the factorisation runs
 * on the GPU), so when <cholmod.h> is not installed we provide the few types
 * the API surface needs, with the same leading field order as SuiteSparse
 * CHOLMOD 2.x-5.x so that user callbacks compile unchanged.
 *
 * When the real header is available it is used instead, and everything here
 * disappears.
 */
#ifndef DOGLEG_CHOLMOD_COMPAT_H
#define DOGLEG_CHOLMOD_COMPAT_H

#if defined(__has_include)
#  if __has_include(<cholmod.h>) && !defined(DOGLEG_FORCE_CHOLMOD_COMPAT)
#    include <cholmod.h>
#    define DOGLEG_HAVE_REAL_CHOLMOD 1
#  endif
#endif

#ifndef DOGLEG_HAVE_REAL_CHOLMOD

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* values of the xtype / dtype / itype tags, as CHOLMOD defines them */
#define CHOLMOD_PATTERN 0
#define CHOLMOD_REAL    1
#define CHOLMOD_DOUBLE  0
#define CHOLMOD_INT     0

/* compressed-column sparse matrix. For the dogleg callback: nrow = Nstate,
 * ncol = Nmeasurements, column r = gradient of measurement r, 32-bit indices,
 * sorted, packed, stype = 0 (unsymmetric). */
typedef struct cholmod_sparse_struct
{
  size_t nrow, ncol, nzmax;
  void  *p;    /* int[ncol+1] column pointers   */
  void  *i;    /* int[nzmax]  row indices       */
  void  *nz;   /* unused (packed)               */
  void  *x;    /* double[nzmax] values          */
  void  *z;    /* unused (real)                 */
  int    stype, itype, xtype, dtype, sorted, packed;
} cholmod_sparse;

/* dense column-major matrix; the solver exposes updateGN through one */
typedef struct cholmod_dense_struct
{
  size_t nrow, ncol, nzmax, d;
  void  *x, *z;
  int    xtype, dtype;
} cholmod_dense;

/* Placeholders: the context struct of the API names these types.  They carry
 * no CHOLMOD state here; `factorization` of a sparse solve is an opaque handle
 * to the GPU supernodal factor. */
typedef struct cholmod_factor_struct
{
  size_t n, minor;      /* minor == n  <=>  last factorisation succeeded */
  void  *backend;       /* opaque */
} cholmod_factor;

typedef struct cholmod_common_struct
{
  int    supernodal;    /* kept for source compatibility; ignored */
  int    status;
  void  *reserved[6];
} cholmod_common;

#ifdef __cplusplus
}
#endif

#endif /* !DOGLEG_HAVE_REAL_CHOLMOD */
#endif /* DOGLEG_CHOLMOD_COMPAT_H */
