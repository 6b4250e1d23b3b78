// gradtest.cpp -- dogleg_testGradient{,_dense,_dense_products}: the developer tool of the
// reference (dogleg.h:312-322, dogleg.c:352-522).  Host only: two callback evaluations half a
// step on either side of p0[var]; per measurement the reported gradient (mean of the two
// Jacobians) is printed next to the central difference, as a vnlog-style text table on stdout:
//   # ivar imeasurement gradient_reported gradient_observed error error_relative
// No linear algebra, no GPU.
#include "../../include/dogleg.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {
constexpr double kDelta = 1e-6;               // reference GRADTEST_DELTA, dogleg.c:352

// d x[meas] / d p[var] in a CSC Jt (column = measurement, rows = variables); absent entry = 0
double entry(const cholmod_sparse* Jt, unsigned var, unsigned meas)
{
  const int* cp = static_cast<const int*>(Jt->p);
  const int* ri = static_cast<const int*>(Jt->i);
  const double* v = static_cast<const double*>(Jt->x);
  for(int k = cp[meas]; k < cp[meas + 1]; k++) if((unsigned)ri[k] == var) return v[k];
  return 0.0;
}

struct HostJt                                   // a cholmod_sparse over plain host arrays
{
  std::vector<int> p, i; std::vector<double> x; cholmod_sparse A;
  HostJt(unsigned N, unsigned M, unsigned nnz) : p(M + 1, 0), i(nnz, 0), x(nnz, 0.0)
  {
    memset(&A, 0, sizeof(A));
    A.nrow = N; A.ncol = M; A.nzmax = nnz; A.p = p.data(); A.i = i.data(); A.x = x.data();
    A.sorted = 1; A.packed = 1;
    A.itype = 0 /* CHOLMOD_INT */; A.xtype = 1 /* CHOLMOD_REAL */; A.dtype = 0 /* CHOLMOD_DOUBLE */;
  }
};

void report(unsigned var, unsigned Nmeas, const double* x0, const double* x1,
            double (*reported)(unsigned, unsigned, const void*, const void*, unsigned), const void* J0,
            const void* J1, unsigned Nstate)
{
  printf("# ivar imeasurement gradient_reported gradient_observed error error_relative\n");
  for(unsigned m = 0; m < Nmeas; m++)
  {
    const double observed = (x1[m] - x0[m])/kDelta;
    const double rep = reported(var, m, J0, J1, Nstate);
    const double sum = fabs(rep) + fabs(observed), err = fabs(rep - observed);
    printf("%d %d %.6g %.6g %.6g %.6g\n", (int)var, (int)m, rep, observed, err, sum == 0.0 ? 0.0 : err/(sum/2.0));
  }
}
} // namespace

extern "C" void dogleg_testGradient(unsigned int var, const double* p0, unsigned int Nstate,
                                    unsigned int Nmeas, unsigned int NJnnz, dogleg_callback_t* f, void* cookie)
{
  if(NJnnz == 0) { fprintf(stderr, "libdogleg_amd: dogleg_testGradient needs NJnnz > 0\n"); return; }
  if(!f || !p0 || var >= Nstate) { fprintf(stderr, "libdogleg_amd: dogleg_testGradient: bad arguments\n"); return; }
  std::vector<double> p(p0, p0 + Nstate), x0(Nmeas), x1(Nmeas);
  HostJt J0(Nstate, Nmeas, NJnnz), J1(Nstate, Nmeas, NJnnz);
  p[var] = p0[var] - kDelta/2.0; f(p.data(), x0.data(), &J0.A, cookie);
  p[var] = p0[var] + kDelta/2.0; f(p.data(), x1.data(), &J1.A, cookie);
  report(var, Nmeas, x0.data(), x1.data(),
         [](unsigned v, unsigned m, const void* a, const void* b, unsigned) {
           return 0.5*(entry(static_cast<const cholmod_sparse*>(a), v, m) + entry(static_cast<const cholmod_sparse*>(b), v, m)); },
         &J0.A, &J1.A, Nstate);
}

extern "C" void dogleg_testGradient_dense(unsigned int var, const double* p0, unsigned int Nstate,
                                          unsigned int Nmeas, dogleg_callback_dense_t* f, void* cookie)
{
  if(!f || !p0 || var >= Nstate) { fprintf(stderr, "libdogleg_amd: dogleg_testGradient_dense: bad arguments\n"); return; }
  std::vector<double> p(p0, p0 + Nstate), x0(Nmeas), x1(Nmeas);
  std::vector<double> J0((size_t)Nmeas*Nstate), J1((size_t)Nmeas*Nstate);
  p[var] = p0[var] - kDelta/2.0; f(p.data(), x0.data(), J0.data(), cookie);
  p[var] = p0[var] + kDelta/2.0; f(p.data(), x1.data(), J1.data(), cookie);
  report(var, Nmeas, x0.data(), x1.data(),
         [](unsigned v, unsigned m, const void* a, const void* b, unsigned N) {
           return 0.5*(static_cast<const double*>(a)[v + (size_t)m*N] + static_cast<const double*>(b)[v + (size_t)m*N]); },
         J0.data(), J1.data(), Nstate);
}

// the reference never finished this variant either (it prints a message and exits, dogleg.c:441-447);
// here it reports and returns
extern "C" void dogleg_testGradient_dense_products(unsigned int, const double*, unsigned int, unsigned int,
                                                   dogleg_callback_dense_products_t*, void*)
{
  fprintf(stderr, "libdogleg_amd: dogleg_testGradient_dense_products is not implemented (nor is it in the reference)\n");
}
