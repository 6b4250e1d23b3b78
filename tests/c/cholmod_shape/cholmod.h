/* TEST FIXTURE, not a CHOLMOD replacement: a header that declares the PUBLIC type and field names of
 * SuiteSparse CHOLMOD's cholmod.h (cholmod_sparse, cholmod_dense, cholmod_factor, cholmod_common and
 * the xtype/dtype/itype tags) in CHOLMOD's own field order, with the many fields this repository never
 * touches present as well.  tests/test_library_cpu.py compiles the host side of libdogleg_amd against
 * it with the include path set so that <cholmod.h> resolves here: what must keep compiling when the
 * drop-in is built where SuiteSparse is installed (include/dogleg_cholmod_compat.h then steps aside).
 * Nothing links against or runs with this file. */
#ifndef CHOLMOD_H_SHAPE_FIXTURE
#define CHOLMOD_H_SHAPE_FIXTURE
#include <stddef.h>
#include <stdint.h>
#define CHOLMOD_PATTERN 0
#define CHOLMOD_REAL 1
#define CHOLMOD_COMPLEX 2
#define CHOLMOD_ZOMPLEX 3
#define CHOLMOD_DOUBLE 0
#define CHOLMOD_SINGLE 4
#define CHOLMOD_INT 0
#define CHOLMOD_LONG 2
#define CHOLMOD_A 0
#define CHOLMOD_MAIN_VERSION 5
#define CHOLMOD_SUB_VERSION 0
#define CHOLMOD_VER_CODE(main, sub) ((main) * 1000 + (sub))
#define CHOLMOD_VERSION CHOLMOD_VER_CODE(CHOLMOD_MAIN_VERSION, CHOLMOD_SUB_VERSION)

typedef struct cholmod_sparse_struct
{
  size_t nrow, ncol, nzmax;
  void *p, *i, *nz, *x, *z;
  int stype, itype, xtype, dtype, sorted, packed;
} cholmod_sparse;

typedef struct cholmod_dense_struct
{
  size_t nrow, ncol, nzmax, d;
  void *x, *z;
  int xtype, dtype;
} cholmod_dense;

typedef struct cholmod_factor_struct
{
  size_t n, minor;
  void *Perm, *ColCount, *IPerm;
  size_t nzmax;
  void *p, *i, *x, *z, *nz, *next, *prev;
  size_t nsuper, ssize, xsize, maxcsize, maxesize;
  void *super, *pi, *px, *s;
  int ordering, is_ll, is_super, is_monotonic, itype, xtype, dtype, useGPU;
} cholmod_factor;

typedef struct cholmod_common_struct
{
  double dbound, grow0, grow1;
  size_t grow2, maxrank;
  double supernodal_switch;
  int supernodal, final_asis, final_super, final_ll, final_pack, final_monotonic, final_resymbol;
  double zrelax[3];
  size_t nrelax[3];
  int prefer_zomplex, prefer_upper, quick_return_if_not_posdef, prefer_binary, print, precise, try_catch;
  void (*error_handler)(int status, const char* file, int line, const char* message);
  int nmethods, current, selected;
  char other[2048];                 /* the rest of cholmod_common (methods, workspace, statistics ...) */
  int itype, dtype, no_workspace_reallocate, status;
} cholmod_common;
#endif
