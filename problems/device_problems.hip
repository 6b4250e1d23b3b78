// device_problems.hip -- the synthetic test/bench problems of problems.c evaluated ON THE GPU,
// in the dogleg_callback_device_t contract (include/dogleg.h, dogleg_optimize_device2).
// Test and benchmark plumbing like problems.c: the same residual model, the same operation order
// per measurement row (compiled with -ffp-contract=off, as gcc compiles problems.c for x86-64), so
// the host callback handed to the CPU oracle and this device callback handed to the product
// describe the same function up to the last bits of sin / cos.
//
//   ba:     u_r = sum_t a_t (p[i_t] - p*[i_t]);  x_r = u_r + eps sin(u_r) - noise n_r;
//           J_t = a_t (1 + eps cos(u_r))                                   (problems.c synth_ba_eval)
//   dense:  the same with coefficients urand(seed, 3, r N + j) / sqrt(N)    (problems.c synth_cb_dense)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>

namespace {
__host__ __device__ inline uint64_t mix64(uint64_t z)
{
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__host__ __device__ inline double urand(uint64_t seed, uint64_t stream, uint64_t idx)
{
  const uint64_t h = mix64(mix64(seed ^ (stream*0xD6E8FEB86659FD93ull)) + idx);
  return (double)(h >> 11) * (2.0/9007199254740992.0) - 1.0;
}

struct DevProblem
{
  int kind, N, M, nnz;
  uint64_t seed;
  double eps, noise;
  int *Jp, *Ji;
  double *a, *pstar;
  int neval;
};

// A workgroup takes BA_RPB consecutive rows: their entries are one contiguous run of Jt's arrays, read with lane = entry
// (the products a_t (p[i_t] - p*[i_t]) and the coefficients parked in LDS), then one thread per row sums ITS products in index
// order -- the order of the host's loop, so the bits are the host's -- and leaves the row's factor behind for the second
// coalesced pass that writes the Jacobian values.  (One thread per row with a row's 15 entries 120 bytes apart read
// 300 MB at 0.5 TB/s: 0.65 ms an evaluation, more than a whole trial step of the solver: profiles/r05_e2e.md.)
constexpr int BA_RPB = 64, BA_CAP = 2048;
__global__ void __launch_bounds__(256) k_ba_eval(int M, const int* __restrict__ Jp, const int* __restrict__ Ji,
                                                 const double* __restrict__ a, const double* __restrict__ pstar,
                                                 const double* __restrict__ p, double eps, double noise,
                                                 uint64_t seed, double* __restrict__ x, double* __restrict__ Jx)
{
  __shared__ double s_prod[BA_CAP], s_a[BA_CAP];
  const int r0 = blockIdx.x*BA_RPB, r1 = min(M, r0 + BA_RPB);
  const int e0 = Jp[r0], e1 = Jp[r1], n = e1 - e0;
  if(n > BA_CAP)
  {
    // (rows too long for the staging buffers: one thread per row)
    const int r = r0 + threadIdx.x;
    if(threadIdx.x < BA_RPB && r < r1)
    {
      const int t0 = Jp[r], t1 = Jp[r+1];
      double u = 0.0;
      for(int t = t0; t < t1; t++) { const int j = Ji[t]; u += a[t]*(p[j] - pstar[j]); }
      x[r] = u + eps*sin(u) - noise*urand(seed, 5, (uint64_t)r);
      const double d = 1.0 + eps*cos(u);
      for(int t = t0; t < t1; t++) Jx[t] = a[t]*d;
    }
    return;
  }
  for(int e = threadIdx.x; e < n; e += 256)
  {
    const int j = Ji[e0 + e];
    const double av = a[e0 + e];
    s_a[e] = av;
    s_prod[e] = av*(p[j] - pstar[j]);
  }
  __syncthreads();
  const int r = r0 + threadIdx.x;
  if(threadIdx.x < BA_RPB && r < r1)
  {
    const int t0 = Jp[r] - e0, t1 = Jp[r+1] - e0;
    double u = 0.0;
    for(int t = t0; t < t1; t++) u += s_prod[t];
    x[r] = u + eps*sin(u) - noise*urand(seed, 5, (uint64_t)r);
    const double d = 1.0 + eps*cos(u);
    for(int t = t0; t < t1; t++) s_prod[t] = d;
  }
  __syncthreads();
  for(int e = threadIdx.x; e < n; e += 256) Jx[e0 + e] = s_a[e]*s_prod[e];
}
// one workgroup per row; the coefficients are generated in parallel, u is summed in index order
__global__ void __launch_bounds__(256) k_dense_eval(int M, int N, const double* __restrict__ pstar,
                                                    const double* __restrict__ p, double eps, double noise,
                                                    uint64_t seed, double* __restrict__ x, double* __restrict__ J)
{
  extern __shared__ double sh[];           // N products + 1
  const int r = blockIdx.x;
  double* Jr = J + (size_t)r*N;
  const double sq = sqrt((double)N);
  for(int j = threadIdx.x; j < N; j += 256)
  {
    const double c = urand(seed, 3, (uint64_t)r*(uint64_t)N + (uint64_t)j) / sq;
    Jr[j] = c;
    sh[j] = c*(p[j] - pstar[j]);
  }
  __syncthreads();
  if(threadIdx.x == 0)
  {
    double u = 0.0;
    for(int j = 0; j < N; j++) u += sh[j];
    x[r] = u + eps*sin(u) - noise*urand(seed, 5, (uint64_t)r);
    sh[N] = 1.0 + eps*cos(u);
  }
  __syncthreads();
  const double d = sh[N];
  for(int j = threadIdx.x; j < N; j += 256) Jr[j] *= d;
}

template <class T> T* to_device(const T* h, size_t n)
{
  T* d = nullptr;
  if(hipMalloc(&d, sizeof(T)*(n ? n : 1)) != hipSuccess) return nullptr;
  if(n && hipMemcpy(d, h, sizeof(T)*n, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
  return d;
}
} // namespace

extern "C" {

// sparse ba problem: the host arrays of a problems.c synth_t (pattern, coefficients, p*)
void* synth_dev_create_ba(int N, int M, int nnz, const int* Jp, const int* Ji, const double* a,
                          const double* pstar, double eps, double noise, uint64_t seed)
{
  DevProblem* P = (DevProblem*)calloc(1, sizeof(DevProblem));
  if(!P) return nullptr;
  P->kind = 0; P->N = N; P->M = M; P->nnz = nnz; P->seed = seed; P->eps = eps; P->noise = noise;
  P->Jp = to_device(Jp, (size_t)M + 1); P->Ji = to_device(Ji, (size_t)nnz);
  P->a = to_device(a, (size_t)nnz); P->pstar = to_device(pstar, (size_t)N);
  if(!P->Jp || !P->Ji || !P->a || !P->pstar) { fprintf(stderr, "synth_dev_create_ba: device allocation failed\n"); free(P); return nullptr; }
  return P;
}
void* synth_dev_create_dense(int N, int M, const double* pstar, double eps, double noise, uint64_t seed)
{
  DevProblem* P = (DevProblem*)calloc(1, sizeof(DevProblem));
  if(!P) return nullptr;
  P->kind = 1; P->N = N; P->M = M; P->seed = seed; P->eps = eps; P->noise = noise;
  P->pstar = to_device(pstar, (size_t)N);
  if(!P->pstar) { free(P); return nullptr; }
  return P;
}
void synth_dev_free(void* h)
{
  DevProblem* P = (DevProblem*)h;
  if(!P) return;
  (void)hipFree(P->Jp); (void)hipFree(P->Ji); (void)hipFree(P->a); (void)hipFree(P->pstar);
  free(P);
}
int synth_dev_neval(void* h) { return ((DevProblem*)h)->neval; }

// dogleg_callback_device_t
void synth_cb_device(const double* p_dev, double* x_dev, double* J_dev, void* hip_stream, void* cookie)
{
  DevProblem* P = (DevProblem*)cookie;
  hipStream_t st = (hipStream_t)hip_stream;
  P->neval++;
  if(P->kind == 0)
    hipLaunchKernelGGL(k_ba_eval, dim3((P->M + BA_RPB - 1)/BA_RPB), dim3(256), 0, st, P->M, P->Jp, P->Ji, P->a, P->pstar,
                       p_dev, P->eps, P->noise, P->seed, x_dev, J_dev);
  else
    hipLaunchKernelGGL(k_dense_eval, dim3(P->M), dim3(256), sizeof(double)*((size_t)P->N + 1), st, P->M, P->N,
                       P->pstar, p_dev, P->eps, P->noise, P->seed, x_dev, J_dev);
}

} // extern "C"
