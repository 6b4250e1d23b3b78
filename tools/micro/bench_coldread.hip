// micro-benchmark (tools only): how long does ONE workgroup need to pull a cold chunk of
// global memory (written by the previous kernel) into registers / LDS?  Varies the chunk size,
// the number of concurrent workgroups, the spacing of the chunks and the threads per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_write(double* a, size_t n) { for(size_t i = blockIdx.x*(size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x*blockDim.x) a[i] = 1.0 + i; }
template <int FLIGHT>
__global__ void k_read(const double* a, size_t stride, int chunk, long long* out, double* sink)
{
  const double* g = a + blockIdx.x*stride;
  const long long t0 = clock64();
  double acc = 0.0;
  for(int base = 0; base < chunk; base += FLIGHT*blockDim.x)
  {
    double v[FLIGHT];
#pragma unroll
    for(int u = 0; u < FLIGHT; u++) { const int e = base + u*blockDim.x + threadIdx.x; v[u] = e < chunk ? g[e] : 0.0; }
#pragma unroll
    for(int u = 0; u < FLIGHT; u++) acc += v[u];
  }
  __syncthreads();
  const long long t1 = clock64();
  if(threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if(acc == 12345.678) sink[0] = acc;
}
int main(int argc, char** argv)
{
  const size_t N = (size_t)1 << 28;            // 2 GB of doubles
  double* a; long long* out; double* sink;
  hipMalloc(&a, N*8); hipMalloc(&out, 4096*8); hipMalloc(&sink, 8);
  const int chunks[] = {1024, 8192, 16384};     // doubles: 8 KB, 64 KB, 128 KB
  const int wgs[] = {1, 8, 64, 256};
  const size_t strides[] = {16384, 262144 + 1024};
  for(int nt : {256, 512, 1024})
    for(int chunk : chunks)
      for(int nwg : wgs)
        for(size_t stride : strides)
        {
          hipLaunchKernelGGL(k_write, dim3(1024), dim3(256), 0, 0, a, N);
          hipLaunchKernelGGL(k_read<16>, dim3(nwg), dim3(nt), 0, 0, a, stride, chunk, out, sink);
          hipDeviceSynchronize();
          std::vector<long long> h(nwg); hipMemcpy(h.data(), out, nwg*8, hipMemcpyDeviceToHost);
          long long mx = 0, sum = 0; for(long long x : h) { mx = x > mx ? x : mx; sum += x; }
          printf("nt %4d chunk %4d KB wgs %3d stride %7zu KB: block0 %6lld avg %6lld max %6lld cycles\n", nt, chunk/128, nwg, stride/128, h[0], sum/nwg, mx);
        }
  return 0;
}
