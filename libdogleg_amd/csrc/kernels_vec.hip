// kernels_vec.hip -- O(N) vector kernels of the dog-leg step (K7 and the
// scalar reductions around K1/K3/K6/K8).  All reductions are two-stage and
// order-deterministic: stage 1 writes one partial per workgroup, stage 2 (one
// workgroup) combines them in index order, so results are bitwise
// reproducible run to run.  HBM-bound streaming kernels; 64-wide wavefront
// shuffles, no atomics.
//
// Reference loops replaced: norm2 (dogleg.c:190-196), inner (197-203),
// vec_copy_scaled (221-226), vec_add (228-233), vec_negate (241-245), the
// interpolation loops (964-987), the threshold scans (1073-1078, 1289-1291).
#include "dlg_internal.h"

namespace {

constexpr int TPB = 256;
constexpr int MAXB = 1024;

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for(int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
  for(int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  return v;
}
// block-wide sum/max; result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* sh)
{
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if(l == 0) sh[w] = v;
  __syncthreads();
  double r = 0;
  if(threadIdx.x == 0) for(int i = 0; i < (int)(blockDim.x >> 6); i++) r += sh[i];
  return r;
}
__device__ __forceinline__ double block_max(double v, double* sh)
{
  v = wave_max(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if(l == 0) sh[w] = v;
  __syncthreads();
  double r = 0;
  if(threadIdx.x == 0) for(int i = 0; i < (int)(blockDim.x >> 6); i++) r = fmax(r, sh[i]);
  return r;
}

__device__ __forceinline__ double block_min(double v, double* sh)
{
#pragma unroll
  for(int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_down(v, o, 64));
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if(l == 0) sh[w] = v;
  __syncthreads();
  double r = 1e300;
  if(threadIdx.x == 0) for(int i = 0; i < (int)(blockDim.x >> 6); i++) r = fmin(r, sh[i]);
  return r;
}

// Several block-wide reductions behind ONE pair of barriers, their shuffle chains interleaved: value k is combined with
// OPS[k] (0 sum, 1 max, 2 min) exactly as block_sum / block_max / block_min combine it -- the same operations in the same
// order, the same bits --; results valid in thread 0.  sh: 4 doubles per value.
template <int... OPS>
__device__ __forceinline__ void block_reduce(double (&v)[sizeof...(OPS)], double* sh)
{
  constexpr int K = sizeof...(OPS);
  constexpr int op[K] = { OPS... };
#pragma unroll
  for(int o = 32; o > 0; o >>= 1)
  {
    double t[K];
#pragma unroll
    for(int k = 0; k < K; k++) t[k] = __shfl_down(v[k], o, 64);
#pragma unroll
    for(int k = 0; k < K; k++) v[k] = op[k] == 0 ? v[k] + t[k] : (op[k] == 1 ? fmax(v[k], t[k]) : fmin(v[k], t[k]));
  }
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if(l == 0)
  {
#pragma unroll
    for(int k = 0; k < K; k++) sh[4*k + w] = v[k];
  }
  __syncthreads();
  if(threadIdx.x == 0)
  {
#pragma unroll
    for(int k = 0; k < K; k++)
    {
      double r = op[k] == 0 ? 0.0 : (op[k] == 1 ? 0.0 : 1e300);
      for(int i = 0; i < (int)(blockDim.x >> 6); i++) r = op[k] == 0 ? r + sh[4*k + i] : (op[k] == 1 ? fmax(r, sh[4*k + i]) : fmin(r, sh[4*k + i]));
      v[k] = r;
    }
  }
}

// partial layout: part[k*nb + blk] for output k
__global__ void __launch_bounds__(TPB) k_part_norm2_absmax(const double* __restrict__ x, int n,
                                                           double* __restrict__ part)
{
  __shared__ double sh[4];
  double s = 0, m = 0;
  for(int i = blockIdx.x*TPB + threadIdx.x; i < n; i += gridDim.x*TPB)
  { const double v = x[i]; s += v*v; m = fmax(m, fabs(v)); }
  const double S = block_sum(s, sh);
  const double Mx = block_max(m, sh);
  if(threadIdx.x == 0) { part[blockIdx.x] = S; part[gridDim.x + blockIdx.x] = Mx; }
}
// the same for two vectors in one launch: blocks [0, g1) take x1, the rest x2 -- each half sums exactly as
// a launch of its own would (same grid-stride order), one launch latency instead of two
__global__ void __launch_bounds__(TPB) k_part_norm2_absmax2(const double* __restrict__ x1, int n1, double* __restrict__ part1, int g1,
                                                            const double* __restrict__ x2, int n2, double* __restrict__ part2)
{
  __shared__ double sh[4];
  const bool first = (int)blockIdx.x < g1;
  const double* x = first ? x1 : x2;
  const int n = first ? n1 : n2, g = first ? g1 : (int)gridDim.x - g1, blk = first ? (int)blockIdx.x : (int)blockIdx.x - g1;
  double* part = first ? part1 : part2;
  double s = 0, m = 0;
  for(int i = blk*TPB + threadIdx.x; i < n; i += g*TPB)
  { const double v = x[i]; s += v*v; m = fmax(m, fabs(v)); }
  const double S = block_sum(s, sh);
  const double Mx = block_max(m, sh);
  if(threadIdx.x == 0) { part[blk] = S; part[g + blk] = Mx; }
}
__global__ void __launch_bounds__(TPB) k_part_inner(const double* __restrict__ x,
                                                    const double* __restrict__ y, int n,
                                                    double* __restrict__ part)
{
  __shared__ double sh[4];
  double s = 0;
  for(int i = blockIdx.x*TPB + threadIdx.x; i < n; i += gridDim.x*TPB) s += x[i]*y[i];
  const double S = block_sum(s, sh);
  if(threadIdx.x == 0) part[blockIdx.x] = S;
}
// final: nsum sums followed by nmax maxima, each over nb partials
__global__ void __launch_bounds__(TPB) k_final(const double* __restrict__ part, int nb, int nsum,
                                               int nmax, double* __restrict__ out, int ostride)
{
  __shared__ double sh[4];
  for(int k = 0; k < nsum + nmax; k++)
  {
    const double* pk = part + (size_t)k*nb;
    double v = 0;
    if(k < nsum) { for(int i = threadIdx.x; i < nb; i += TPB) v += pk[i]; v = block_sum(v, sh); }
    else         { for(int i = threadIdx.x; i < nb; i += TPB) v = fmax(v, pk[i]); v = block_max(v, sh); }
    if(threadIdx.x == 0) out[k*ostride] = v;
    __syncthreads();
  }
}

__global__ void __launch_bounds__(TPB) k_cauchy_scale(const double* __restrict__ g, double g2,
                                                      const double* __restrict__ Jg2p,
                                                      double* __restrict__ c, int n,
                                                      double* __restrict__ out)
{
  const double Jg2 = Jg2p[0];
  const double k = -g2 / Jg2;                       // dogleg.c:605
  for(int i = blockIdx.x*TPB + threadIdx.x; i < n; i += gridDim.x*TPB) c[i] = k*g[i];
  if(blockIdx.x == 0 && threadIdx.x == 0) out[0] = k*k*g2;   // dogleg.c:607
}

__global__ void __launch_bounds__(TPB) k_part_scaled_step(const double* __restrict__ v, double s,
                                                          const double* __restrict__ p,
                                                          double* __restrict__ step,
                                                          double* __restrict__ pnew, int n,
                                                          double* __restrict__ part)
{
  __shared__ double sh[4];
  double m = 0;
  for(int i = blockIdx.x*TPB + threadIdx.x; i < n; i += gridDim.x*TPB)
  {
    const double st = s*v[i];
    step[i] = st; pnew[i] = p[i] + st; m = fmax(m, fabs(st));
  }
  const double Mx = block_max(m, sh);
  if(threadIdx.x == 0) part[blockIdx.x] = Mx;
}

// interpolation pass 1: l2 = sum (a-b)^2, neg_c = sum (a-b) a   (dogleg.c:964-972)
__global__ void __launch_bounds__(TPB) k_part_interp1(const double* __restrict__ a,
                                                      const double* __restrict__ b, int n,
                                                      double* __restrict__ part)
{
  __shared__ double sh[4];
  double l2 = 0, nc = 0;
  for(int i = blockIdx.x*TPB + threadIdx.x; i < n; i += gridDim.x*TPB)
  { const double d = a[i] - b[i]; l2 += d*d; nc += d*a[i]; }
  const double L = block_sum(l2, sh);
  const double Cn = block_sum(nc, sh);
  if(threadIdx.x == 0) { part[blockIdx.x] = L; part[gridDim.x + blockIdx.x] = Cn; }
}
// pass 2: k from the reduced scalars (dogleg.c:974-980), step = a + k (b-a)
__global__ void __launch_bounds__(TPB) k_part_interp2(const double* __restrict__ a,
                                                      const double* __restrict__ b,
                                                      const double* __restrict__ part1, int nb1,
                                                      double norm2a, double dsq,
                                                      const double* __restrict__ p,
                                                      double* __restrict__ step,
                                                      double* __restrict__ pnew, int n,
                                                      double* __restrict__ part,
                                                      double* __restrict__ kout)
{
  __shared__ double sh[4];
  __shared__ double s_l2, s_negc;
  {
    // second stage of pass 1, redone by every workgroup in the same (index) order: no launch for it
    double v0 = 0, v1 = 0;
    for(int i = threadIdx.x; i < nb1; i += TPB) { v0 += part1[i]; v1 += part1[nb1 + i]; }
    v0 = block_sum(v0, sh); __syncthreads();
    v1 = block_sum(v1, sh);
    if(threadIdx.x == 0) { s_l2 = v0; s_negc = v1; }
    __syncthreads();
  }
  const double l2 = s_l2, neg_c = s_negc;
  double disc = neg_c*neg_c - l2*(norm2a - dsq);
  if(disc < 0.0) disc = 0.0;
  const double k = (neg_c + sqrt(disc))/l2;
  double s2 = 0, m = 0;
  for(int i = blockIdx.x*TPB + threadIdx.x; i < n; i += gridDim.x*TPB)
  {
    const double st = a[i] + k*(b[i] - a[i]);
    step[i] = st; pnew[i] = p[i] + st; s2 += st*st; m = fmax(m, fabs(st));
  }
  const double S = block_sum(s2, sh);
  const double Mx = block_max(m, sh);
  if(threadIdx.x == 0) { part[blockIdx.x] = S; part[gridDim.x + blockIdx.x] = Mx; }
  if(blockIdx.x == 0 && threadIdx.x == 0) kout[0] = k;
}

// takeStepFrom's choice of step (dogleg.c:1192-1256) made on the device, so that the Cauchy step,
// the Gauss-Newton step and the step itself can be issued behind ONE host synchronisation:
//   |cauchy|^2 >= D^2 -> cauchy scaled to the edge; else |gn|^2 <= D^2 -> gn; else interpolate.
// |gn|^2 = the partials of k_part_negate_norm2 summed in index order (as the host would);
// l2 / neg_c = the partials of k_part_interp1 (as k_part_interp2 sums them).
// out3 = {kind, k, |gn|^2}; the step's own partials (|step|^2, max|step|) go to `part`.
__global__ void __launch_bounds__(TPB) k_part_take_step(const double* __restrict__ a,
                                                        const double* __restrict__ b,
                                                        const double* __restrict__ part1, int nb1,
                                                        const double* __restrict__ gnpart, int nbg,
                                                        const double* __restrict__ n2c_dev, double trustregion,
                                                        const double* __restrict__ p,
                                                        double* __restrict__ step,
                                                        double* __restrict__ pnew, int n,
                                                        double* __restrict__ part,
                                                        double* __restrict__ out3,
                                                        const double* __restrict__ g,
                                                        double* __restrict__ gpart,
                                                        const double* __restrict__ dsc, double* __restrict__ hsc, int nsc,
                                                        double* __restrict__ gbpart, const double* __restrict__ mmpart, int nbm,
                                                        int ident_enable, double ratio_max, double* __restrict__ ident_out)
{
  __shared__ double sh[16];
  __shared__ double s_l2, s_negc, s_n2g, s_skip, s_ratio;
  __shared__ double s_gn[MAXB];
  // The expected improvement from the solved system (K8 without its pass over J, backend.hip: ident_value): allowed by
  // the host (lambda == 0, one rank) and by the factor itself -- the ratio of its largest to its smallest pivot, from
  // the pairs the backward solve left per supernode (mmpart: their partial minima / maxima, k_part_negate_interp1).
  // ident_out[0] = 1: the pass over J that follows on the stream returns at once (k_norm2_Jv: skipf); [1] = the ratio.
  double ratio = 0.0;
  if(ident_out && blockIdx.x == 0)
  {
    double lo = 1e300, hi = 0.0;
    for(int i = threadIdx.x; i < nbm; i += TPB) { lo = fmin(lo, mmpart[i]); hi = fmax(hi, mmpart[nbm + i]); }
    { double r2[2] = { lo, hi }; block_reduce<2, 1>(r2, sh); lo = r2[0]; hi = r2[1]; }
    if(threadIdx.x == 0) { ratio = (nbm > 0 && lo > 0.0) ? hi/lo : INFINITY; s_ratio = ratio; }
    __syncthreads();
  }
  {
    double v0 = 0, v1 = 0;
    for(int i = threadIdx.x; i < nb1; i += TPB) { v0 += part1[i]; v1 += part1[nb1 + i]; }
    for(int i = threadIdx.x; i < nbg; i += TPB) s_gn[i] = gnpart[i];      // one round of loads, summed in order below
    { double r2[2] = { v0, v1 }; block_reduce<0, 0>(r2, sh); v0 = r2[0]; v1 = r2[1]; }
    if(threadIdx.x == 0)
    {
      s_l2 = v0; s_negc = v1;
      // in index order, as the host sums them -- sixteen LDS reads in flight per trip: one read per add
      // (a dependent LDS round trip each) made this prologue most of the kernel
      double g2 = 0;
      int i = 0;
      for(; i + 16 <= nbg; i += 16)
      {
        double t[16];
#pragma unroll
        for(int u = 0; u < 16; u++) t[u] = s_gn[i + u];
#pragma unroll
        for(int u = 0; u < 16; u++) g2 += t[u];
      }
      for(; i < nbg; i++) g2 += s_gn[i];
      s_n2g = g2;
    }
    __syncthreads();
  }
  const double n2c = n2c_dev[0], n2g = s_n2g, dsq = trustregion*trustregion;
  const int kind = (n2c >= dsq) ? 0 : ((n2g <= dsq) ? 1 : 2);
  double k = 0.0, sc = 0.0;
  if(kind == 0) sc = trustregion / sqrt(n2c);                  // dogleg.c:1204-1207
  else if(kind == 2)
  {
    const double l2 = s_l2, neg_c = s_negc;                     // dogleg.c:974-980
    double disc = neg_c*neg_c - l2*(n2c - dsq);
    if(disc < 0.0) disc = 0.0;
    k = (neg_c + sqrt(disc))/l2;
  }
  double s2 = 0, m = 0, gs = 0, gb = 0;
  for(int i = blockIdx.x*TPB + threadIdx.x; i < n; i += gridDim.x*TPB)
  {
    const double bi = b[i], gi = g[i];
    const double st = (kind == 0) ? sc*a[i] : ((kind == 1) ? bi : a[i] + k*(bi - a[i]));
    step[i] = st; pnew[i] = p[i] + st; s2 += st*st; m = fmax(m, fabs(st));
    gs += gi*st;                                                 // <Jt x, step> for the expected improvement
    gb += gi*bi;                                                 // <Jt x, gn>: |J gn|^2 of the solved system
  }
  double r4[4] = { s2, m, gs, gb };
  block_reduce<0, 1, 0, 0>(r4, sh);
  const double S = r4[0], Mx = r4[1], Gs = r4[2], Gb = r4[3];
  if(threadIdx.x == 0) { part[blockIdx.x] = S; part[gridDim.x + blockIdx.x] = Mx; gpart[blockIdx.x] = Gs; if(gbpart) gbpart[blockIdx.x] = Gb; }
  if(blockIdx.x == 0 && threadIdx.x == 0)
  {
    out3[0] = (double)kind; out3[1] = (kind == 2) ? k : NAN; out3[2] = n2g;
    if(ident_out)
    {
      // (the Cauchy step to the edge needs no factor: its |J step|^2 is a multiple of K3's own scalar)
      const double skip = (kind == 0 || (ident_enable && ratio <= ratio_max)) ? 1.0 : 0.0;
      ident_out[0] = skip; ident_out[1] = ratio; s_skip = skip;
      ident_out[4] = s_negc;                                     // <cauchy - gn, cauchy>: with lambda > 0 the identity needs <cauchy, gn>
    }
  }
  // the step's last kernel on the main stream (K8 follows on the second one, dlg_step_tail): the device scalars of the
  // step -- written by the kernels before this one, and out3 from here -- go to the page-locked host array with it
  if(dsc && blockIdx.x == 0) __syncthreads();                    // (s_skip, s_ratio: thread 0's, above)
  if(dsc && blockIdx.x == 0 && (int)threadIdx.x < nsc)
  {
    const double* o3 = out3;
    const int t = threadIdx.x, d = (int)(o3 - dsc), f = ident_out ? (int)(ident_out - dsc) : -8;
    hsc[t] = (t == d) ? (double)kind : (t == d + 1) ? ((kind == 2) ? k : NAN) : (t == d + 2) ? n2g :
             (t == f) ? s_skip : (t == f + 1) ? s_ratio : (t == f + 4) ? s_negc : dsc[t];
  }
}

// gn = -u with the partials of |gn|^2, and pass 1 of the interpolation (l2, neg_c of cauchy vs. gn)
// in the same sweep: what dlg_take_step needs before it can choose the step
__global__ void __launch_bounds__(TPB) k_part_negate_interp1(double* __restrict__ v,
                                                             const double* __restrict__ a, int n,
                                                             double* __restrict__ gnpart,
                                                             double* __restrict__ part1,
                                                             const int* __restrict__ gate, int epoch, int* __restrict__ status,
                                                             const double* __restrict__ mm, int nmm, double* __restrict__ mmpart,
                                                             long mm_stride, int mm_hi)
{
  __shared__ double sh[12];
  // (the factor's smallest / largest pivot: a partial minimum / maximum per workgroup of the pairs the backward solve left
  // per supernode -- k_part_take_step finishes them)
  if(mmpart)
  {
    double lo = 1e300, hi = 0.0;
    for(int i = blockIdx.x*TPB + threadIdx.x; i < nmm; i += gridDim.x*TPB) { lo = fmin(lo, mm[i*mm_stride]); hi = fmax(hi, mm[i*mm_stride + mm_hi]); }
    { double r2[2] = { lo, hi }; block_reduce<2, 1>(r2, sh); lo = r2[0]; hi = r2[1]; }
    if(threadIdx.x == 0) { mmpart[blockIdx.x] = lo; mmpart[gridDim.x + blockIdx.x] = hi; }
    __syncthreads();
  }
  // (the Cauchy step a comes from the second stream: its word instead of an event between two kernels of this stream,
  // dlg_backend::d_join)
  if(gate)
  {
    if(threadIdx.x == 0)
      for(int spins = 0; __hip_atomic_load(gate, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - epoch < 0; spins++)
      {
        if(spins > (1 << 21)) { atomicOr(status, DLG_HANDOFF_FACTOR); break; }       // (never in order: reported, not hung)
        __builtin_amdgcn_s_sleep(8);
      }
    __syncthreads();
  }
  double s = 0, l2 = 0, nc = 0;
  for(int i = blockIdx.x*TPB + threadIdx.x; i < n; i += gridDim.x*TPB)
  {
    const double t = -v[i]; v[i] = t; s += t*t;
    const double ai = a[i], d = ai - t; l2 += d*d; nc += d*ai;
  }
  double r3[3] = { s, l2, nc };
  block_reduce<0, 0, 0>(r3, sh);
  const double S = r3[0], L = r3[1], Cn = r3[2];
  if(threadIdx.x == 0) { gnpart[blockIdx.x] = S; part1[blockIdx.x] = L; part1[gridDim.x + blockIdx.x] = Cn; }
}
__global__ void __launch_bounds__(TPB) k_part_negate_norm2(double* __restrict__ v, int n,
                                                           double* __restrict__ part)
{
  __shared__ double sh[4];
  double s = 0;
  for(int i = blockIdx.x*TPB + threadIdx.x; i < n; i += gridDim.x*TPB)
  { const double t = -v[i]; v[i] = t; s += t*t; }
  const double S = block_sum(s, sh);
  if(threadIdx.x == 0) part[blockIdx.x] = S;
}

inline int grid_for(int n) { int g = dlg_cdiv(n, TPB*2); if(g < 1) g = 1; if(g > MAXB) g = MAXB; return g; }

} // namespace

double* dlg_host_partials(dlg_backend* b, const double* out, int nb, int nsum, int nmax, int stride)
{
  if(!b->host_finals || !b->h_part) return nullptr;
  if(out < b->d_scal || out >= b->d_scal + dlg_backend::NSCAL) return nullptr;
  const size_t need = (size_t)(nsum + nmax)*nb;
  if(b->h_part_used + need > dlg_backend::HPART_CAP) return nullptr;
  double* region = b->h_part + b->h_part_used;
  b->pending.push_back({b->h_part_used, nb, nsum, nmax, (int)(out - b->d_scal), stride});
  b->h_part_used += need;
  return region;
}
double* dlg_tail_partials(dlg_backend* b, int nb)
{
  if(nb > b->h_tail_cap)
  {
    if(b->h_tail) (void)hipHostFree(b->h_tail);
    b->h_tail = nullptr; b->h_tail_cap = 0;
    if(hipHostMalloc((void**)&b->h_tail, sizeof(double)*(size_t)nb, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    b->h_tail_cap = nb;
  }
  b->tail_nb = nb;
  return b->h_tail;
}
void dlg_resolve_pending(dlg_backend* b)
{
  for(const dlg_backend::PendingFinal& f : b->pending)
    for(int k = 0; k < f.nsum + f.nmax; k++)
    {
      const double* pk = b->h_part + f.off + (size_t)k*f.nb;
      double v = 0;
      if(k < f.nsum) for(int i = 0; i < f.nb; i++) v += pk[i];
      else           for(int i = 0; i < f.nb; i++) v = fmax(v, pk[i]);
      b->h_scal[f.dst + k*f.stride] = v;
    }
  b->pending.clear();
  b->h_part_used = 0;
}

int dlg_ensure_partials(dlg_backend* b, size_t nd)
{
  if(nd <= b->part_cap) return DLG_OK;
  if(b->d_part) DLG_HIP(hipFree(b->d_part));
  b->d_part = nullptr; b->part_cap = 0;
  DLG_HIP(hipMalloc(&b->d_part, nd*sizeof(double)));
  b->part_cap = nd;
  return DLG_OK;
}

// long lists (the thousands of row-run partials of |J v|^2): 1024 threads, four loads in flight
__global__ void __launch_bounds__(1024) k_final_sum_long(const double* __restrict__ part, int nb, double* __restrict__ out)
{
  __shared__ double sh[16];
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  int i = threadIdx.x;
  for(; i + 3072 < nb; i += 4096) { s0 += part[i]; s1 += part[i + 1024]; s2 += part[i + 2048]; s3 += part[i + 3072]; }
  for(; i < nb; i += 1024) s0 += part[i];
  double v = wave_sum((s0 + s1) + (s2 + s3));
  if((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  if(threadIdx.x == 0) { double r = 0; for(int k = 0; k < 16; k++) r += sh[k]; out[0] = r; }
}
int k_reduce_sum(dlg_backend* b, const double* partials, int np, double* out)
{
  if(np > 2048)
  {
    hipLaunchKernelGGL(k_final_sum_long, dim3(1), dim3(1024), 0, b->stream, partials, np, out);
    DLG_LAUNCH_CHECK();
    return DLG_OK;
  }
  hipLaunchKernelGGL(k_final, dim3(1), dim3(TPB), 0, b->stream, partials, np, 1, 0, out, 1);
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}

int k_norm2_absmax(dlg_backend* b, const double* x, int n, double* out2)
{
  const int g = grid_for(n);
  DLG_CHECK(dlg_ensure_partials(b, 4*MAXB));
  if(double* hp = dlg_host_partials(b, out2, g, 1, 1, 1))
  { hipLaunchKernelGGL(k_part_norm2_absmax, dim3(g), dim3(TPB), 0, b->stream, x, n, hp); DLG_LAUNCH_CHECK(); return DLG_OK; }
  hipLaunchKernelGGL(k_part_norm2_absmax, dim3(g), dim3(TPB), 0, b->stream, x, n, b->d_part);
  hipLaunchKernelGGL(k_final, dim3(1), dim3(TPB), 0, b->stream, b->d_part, g, 1, 1, out2, 1);
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}
// (|x1|^2, max|x1|) -> out1[0..1] and (|x2|^2, max|x2|) -> out2[0..1] behind one launch where the second
// stages run on the host (dlg_host_partials); else two calls of k_norm2_absmax
int k_norm2_absmax_pair(dlg_backend* b, const double* x1, int n1, double* out1, const double* x2, int n2, double* out2,
                        bool* on_host)
{
  if(on_host) *on_host = false;
  const int g1 = grid_for(n1), g2 = grid_for(n2);
  if(b->host_finals && b->h_part && b->h_part_used + 2*(size_t)(g1 + g2) <= dlg_backend::HPART_CAP)
  {
    double* hp1 = dlg_host_partials(b, out1, g1, 1, 1, 1);
    double* hp2 = hp1 ? dlg_host_partials(b, out2, g2, 1, 1, 1) : nullptr;
    if(hp1 && hp2)
    {
      DLG_LAUNCH_LAST(b, k_part_norm2_absmax2, dim3(g1 + g2), dim3(TPB), 0, b->stream, x1, n1, hp1, g1, x2, n2, hp2);
      DLG_LAUNCH_CHECK();
      if(on_host) *on_host = true;             // all four scalars are summed on the host (dlg_resolve_pending)
      return DLG_OK;
    }
    if(hp1) { hipLaunchKernelGGL(k_part_norm2_absmax, dim3(g1), dim3(TPB), 0, b->stream, x1, n1, hp1); DLG_LAUNCH_CHECK(); return k_norm2_absmax(b, x2, n2, out2); }
  }
  DLG_CHECK(k_norm2_absmax(b, x1, n1, out1));
  return k_norm2_absmax(b, x2, n2, out2);
}
int k_inner(dlg_backend* b, const double* x, const double* y, int n, double* out)
{
  const int g = grid_for(n);
  DLG_CHECK(dlg_ensure_partials(b, 4*MAXB));
  if(double* hp = dlg_host_partials(b, out, g, 1, 0, 1))
  { DLG_LAUNCH_LAST(b, k_part_inner, dim3(g), dim3(TPB), 0, b->stream, x, y, n, hp); DLG_LAUNCH_CHECK(); return DLG_OK; }      // (attach_stop: the launch the host waits for, dlg_step)
  b->attach_stop = nullptr;
  hipLaunchKernelGGL(k_part_inner, dim3(g), dim3(TPB), 0, b->stream, x, y, n, b->d_part);
  hipLaunchKernelGGL(k_final, dim3(1), dim3(TPB), 0, b->stream, b->d_part, g, 1, 0, out, 1);
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}
int k_cauchy_finish(dlg_backend* b, const double* g, double g2, const double* Jg2_dev, double* cauchy, int n,
                    double* out)
{
  hipLaunchKernelGGL(k_cauchy_scale, dim3(grid_for(n)), dim3(TPB), 0, b->stream, g, g2, Jg2_dev, cauchy,
                     n, out);
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}
int k_scaled_step(dlg_backend* b, const double* v, double s, const double* p, double* step,
                  double* p_new, int n, double* out_absmax)
{
  const int g = grid_for(n);
  DLG_CHECK(dlg_ensure_partials(b, 4*MAXB));
  if(double* hp = dlg_host_partials(b, out_absmax, g, 0, 1, 1))
  {
    hipLaunchKernelGGL(k_part_scaled_step, dim3(g), dim3(TPB), 0, b->stream, v, s, p, step, p_new, n, hp);
    DLG_LAUNCH_CHECK();
    return DLG_OK;
  }
  hipLaunchKernelGGL(k_part_scaled_step, dim3(g), dim3(TPB), 0, b->stream, v, s, p, step, p_new, n,
                     b->d_part);
  hipLaunchKernelGGL(k_final, dim3(1), dim3(TPB), 0, b->stream, b->d_part, g, 0, 1, out_absmax, 1);
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}
int k_interpolate(dlg_backend* b, const double* a, const double* bb, double norm2a,
                  double trustregion, const double* p, double* step, double* p_new, int n,
                  double* out3)
{
  const int g = grid_for(n);
  DLG_CHECK(dlg_ensure_partials(b, 4*MAXB));
  // out3 = {norm2_step, k, absmax}.  Pass 1 leaves its partials in d_part[0 .. 2g); pass 2 sums them
  // itself and writes its own partials behind them (or into host memory: sum -> out3[0], max -> out3[2])
  hipLaunchKernelGGL(k_part_interp1, dim3(g), dim3(TPB), 0, b->stream, a, bb, n, b->d_part);
  double* part2 = b->d_part + 2*g;
  double* hp = dlg_host_partials(b, out3, g, 1, 1, 2);
  // (kout_host: the caller fetches no device scalars behind this step -- dlg_step with K8 behind the decision point --: k goes
  // straight to its place in the page-locked block)
  double* kout = (b->kout_host && hp && out3 >= b->d_scal && out3 + 3 <= b->d_scal + dlg_backend::NSCAL) ? b->h_scal + ((out3 + 1) - b->d_scal) : out3 + 1;
  hipLaunchKernelGGL(k_part_interp2, dim3(g), dim3(TPB), 0, b->stream, a, bb, b->d_part, g, norm2a,
                     trustregion*trustregion, p, step, p_new, n, hp ? hp : part2, kout);
  if(!hp) hipLaunchKernelGGL(k_final, dim3(1), dim3(TPB), 0, b->stream, part2, g, 1, 1, out3, 2);
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}
// gn = -u (in place); its |.|^2 partials -> gnpart, the interpolation's pass-1 partials -> d_part[0 .. 2g)
// (mm / nmm: the pairs [min, max] of the factor's pivots per supernode -- or, mm_stride != 2, the diagonal of a dense factor,
// one entry serving as both --, or null; their partials go to gnpart + 2*MAXB)
int k_negate_interp1(dlg_backend* b, double* gn, const double* cauchy, int n, double* gnpart, int* nb, const double* mm, int nmm, long mm_stride)
{
  const int g = grid_for(n);
  DLG_CHECK(dlg_ensure_partials(b, 5*MAXB));      // k_take_step's layout: no reallocation between the two
  const int ep = b->join_pending;
  b->join_pending = 0;
  hipLaunchKernelGGL(k_part_negate_interp1, dim3(g), dim3(TPB), 0, b->stream, gn, cauchy, n, gnpart, b->d_part,
                     ep ? (const int*)b->d_join : (const int*)nullptr, ep, reinterpret_cast<int*>(b->d_scal + (dlg_backend::NSCAL - 2)),
                     mm, nmm, mm ? gnpart + 2*MAXB : (double*)nullptr, mm_stride, mm_stride == 2 ? 1 : 0);
  DLG_LAUNCH_CHECK();
  *nb = g;
  return DLG_OK;
}
// after k_negate_interp1: the step (kind chosen on the device) and <Jtx, step> -> out_inner
// out_gb: <Jt x, gn>; ident_out (two device scalars inside d_scal, or null): {the pass over J may be skipped, pivot ratio},
// ident_gn: the Gauss-Newton system was solved at lambda = 0 (its identity holds), ratio_max: the largest pivot ratio trusted
int k_take_step(dlg_backend* b, const double* cauchy, const double* gn, const double* gnpart, int nbg,
                const double* n2c_dev, double trustregion, const double* p, double* step, double* p_new, int n,
                double* out_n2_max, double* out3, const double* Jtx, double* out_inner,
                double* out_gb, double* ident_out, bool have_mm, bool ident_gn, double ratio_max)
{
  const int g = grid_for(n);
  // d_part: [0, 2g) pass-1 partials (k_negate_interp1), [2g, 4g) |step|^2 and max|step|, and the
  // <Jt x, step> partials in a region of their own behind 4*MAXB (g can be MAXB)
  DLG_CHECK(dlg_ensure_partials(b, 5*MAXB));
  double* part2 = b->d_part + 2*g;
  double* hp = dlg_host_partials(b, out_n2_max, g, 1, 1, 2);
  double* hg = hp ? dlg_host_partials(b, out_inner, g, 1, 0, 1) : nullptr;
  double* gp = hg ? hg : b->d_part + 4*MAXB;
  // (<Jt x, gn> only where the host adds the partial sums: the identity is a single-rank matter)
  double* hb = (hg && out_gb) ? dlg_host_partials(b, out_gb, g, 1, 0, 1) : nullptr;
  if(!hb) ident_out = nullptr;
  // (fold_scal_k7: this is the launch the host waits for -- it takes the scalars along and carries the event)
  const bool fold = hp && hg && b->fold_scal_k7 > 0 && b->fold_scal_k7 <= TPB && b->h_scal && out3 >= b->d_scal && out3 + 3 <= b->d_scal + b->fold_scal_k7;
  if(!fold) b->attach_stop = nullptr;
  DLG_LAUNCH_LAST(b, k_part_take_step, dim3(g), dim3(TPB), 0, b->stream, cauchy, gn, (const double*)b->d_part, g, gnpart, nbg,
                  n2c_dev, trustregion, p, step, p_new, n, hp ? hp : part2, out3, Jtx, gp,
                  fold ? (const double*)b->d_scal : (const double*)nullptr, b->h_scal, fold ? b->fold_scal_k7 : 0,
                  hb, have_mm ? gnpart + 2*MAXB : (const double*)nullptr, have_mm ? nbg : 0, ident_gn ? 1 : 0, ratio_max, ident_out);
  b->ident_launched = ident_out != nullptr;
  if(fold) b->scal_copied = true;
  if(!hp) hipLaunchKernelGGL(k_final, dim3(1), dim3(TPB), 0, b->stream, part2, g, 1, 1, out_n2_max, 2);
  if(!hg) hipLaunchKernelGGL(k_final, dim3(1), dim3(TPB), 0, b->stream, gp, g, 1, 0, out_inner, 1);
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}
int k_negate_norm2(dlg_backend* b, double* v, int n, double* out)
{
  const int g = grid_for(n);
  DLG_CHECK(dlg_ensure_partials(b, 4*MAXB));
  if(double* hp = dlg_host_partials(b, out, g, 1, 0, 1))
  { hipLaunchKernelGGL(k_part_negate_norm2, dim3(g), dim3(TPB), 0, b->stream, v, n, hp); DLG_LAUNCH_CHECK(); return DLG_OK; }
  hipLaunchKernelGGL(k_part_negate_norm2, dim3(g), dim3(TPB), 0, b->stream, v, n, b->d_part);
  hipLaunchKernelGGL(k_final, dim3(1), dim3(TPB), 0, b->stream, b->d_part, g, 1, 0, out, 1);
  DLG_LAUNCH_CHECK();
  return DLG_OK;
}
