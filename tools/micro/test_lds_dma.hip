#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// each lane copies 16 bytes from global (per-lane offset) to LDS (lane-linear behind a wave-uniform base)
__global__ void k(const double* __restrict__ src, int nbytes, const int* __restrict__ offs, double* __restrict__ out)
{
  extern __shared__ __attribute__((aligned(16))) double L[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  const int off = offs[threadIdx.x];
  __attribute__((address_space(3))) void* lp = (__attribute__((address_space(3))) void*)(L + wv*128);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lp, 16, off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[threadIdx.x*2] = L[threadIdx.x*2]; out[threadIdx.x*2+1] = L[threadIdx.x*2+1];
}
int main()
{
  const int n = 4096;
  std::vector<double> h(n); for(int i = 0; i < n; i++) h[i] = i;
  double *d, *o; int* offs;
  hipMalloc(&d, n*8); hipMalloc(&o, 256*2*8); hipMalloc(&offs, 256*4);
  hipMemcpy(d, h.data(), n*8, hipMemcpyHostToDevice);
  std::vector<int> ho(256);
  for(int t = 0; t < 256; t++) ho[t] = 8*(15*(t/8)) + 16*(t%8);     // rows of 15 doubles, 8 lanes per row
  ho[255] = n*8 - 8;                                                  // straddles the end: second half out of range
  hipMemcpy(offs, ho.data(), 256*4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 4*128*8, 0, d, n*8, offs, o);
  std::vector<double> r(512);
  hipMemcpy(r.data(), o, 512*8, hipMemcpyDeviceToHost);
  int bad = 0;
  for(int t = 0; t < 255; t++) { const double e0 = ho[t]/8, e1 = ho[t]/8 + 1; if(r[2*t] != e0 || r[2*t+1] != e1) bad++; }
  printf("bad %d  last lane: %g %g (expect %d and 0)\n", bad, r[510], r[511], n - 1);
  return 0;
}
