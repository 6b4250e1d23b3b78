// compile-only: the host side of the dogleg.h drop-in against a <cholmod.h> with CHOLMOD's public
// struct shapes (tests/c/cholmod_shape/cholmod.h): include/dogleg.h must pick the "real" header,
// and nothing in the driver may rely on fields that exist only in dogleg_cholmod_compat.h.
#include <cholmod.h>
#include "../../include/dogleg.h"
#ifndef DOGLEG_HAVE_REAL_CHOLMOD
#error "include/dogleg.h did not pick up <cholmod.h>"
#endif
// what a user callback does with the types (reference sample.c:82-125)
static void cb(const double* p, double* x, cholmod_sparse* Jt, void* cookie)
{
  (void)p; (void)cookie;
  int* Jp = (int*)Jt->p; int* Ji = (int*)Jt->i; double* Jx = (double*)Jt->x;
  Jp[0] = 0; Ji[0] = 0; Jx[0] = 1.0; x[0] = 0.0; Jp[1] = 1;
}
int use(dogleg_solverContext_t* ctx)
{
  dogleg_callback_t* f = cb; (void)f;
  // the reference's post-solve idiom: ctx->factorization is a cholmod_factor*, its public fields are readable
  return ctx->factorization && ctx->factorization->minor == ctx->factorization->n ? (int)ctx->beforeStep->Jt->nrow : -1;
}
