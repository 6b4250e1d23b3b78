// accuracy of the v_rsq_f64 seed and of one / two Newton steps on it (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double* d, double* y0, double* y1, double* y2, int n)
{
  int i = blockIdx.x*blockDim.x + threadIdx.x; if(i >= n) return;
  double x = d[i];
  double y = __builtin_amdgcn_rsq(x); y0[i] = y;
  y = y*(1.5 - 0.5*x*y*y); y1[i] = y;
  y = y*(1.5 - 0.5*x*y*y); y2[i] = y;
}
int main()
{
  const int n = 1 << 22;
  std::vector<double> h(n), a(n), b(n), c(n);
  std::mt19937_64 g(7); std::uniform_real_distribution<double> u(0.0, 1.0);
  for(int i = 0; i < n; i++) h[i] = (1.0 + u(g))*std::pow(2.0, (double)((int)(u(g)*120) - 60));
  double *d, *y0, *y1, *y2; hipMalloc(&d, n*8); hipMalloc(&y0, n*8); hipMalloc(&y1, n*8); hipMalloc(&y2, n*8);
  hipMemcpy(d, h.data(), n*8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n/256), dim3(256), 0, 0, d, y0, y1, y2, n);
  hipMemcpy(a.data(), y0, n*8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), y1, n*8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), y2, n*8, hipMemcpyDeviceToHost);
  long double e0 = 0, e1 = 0, e2 = 0;
  for(int i = 0; i < n; i++)
  {
    const long double ex = 1.0L/sqrtl((long double)h[i]);
    e0 = fmaxl(e0, fabsl(a[i] - ex)/ex); e1 = fmaxl(e1, fabsl(b[i] - ex)/ex); e2 = fmaxl(e2, fabsl(c[i] - ex)/ex);
  }
  printf("max relative error: seed %.3Le  one step %.3Le  two steps %.3Le  (eps/2 = 1.11e-16)\n", e0, e1, e2);
  return 0;
}
