// tools only: (1) which XCD a workgroup lands on (XCC_ID against its index), (2) the one-way latency of a word handed
// from one workgroup to another -- same XCD or another, through memory (sc1 stores / sc1 loads: what the library's
// hand-offs use) or through the XCD's own L2 (plain stores, sc0 loads: around L1 only).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o xcd_probe xcd_probe.hip ; run: ./xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15; }
__global__ void k_where(int* out) { extern __shared__ double pad[]; if(threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

typedef __attribute__((address_space(1))) unsigned long long* gu_t;
template <int MODE> __device__ __forceinline__ unsigned long long ld(unsigned long long* p)
{
  unsigned long long v;
  if(MODE == 0) v = __hip_atomic_load((gu_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // sc1: through memory
  else { asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); }   // around L1, into the XCD's L2
  return v;
}
template <int MODE> __device__ __forceinline__ void st(unsigned long long* p, unsigned long long v)
{
  if(MODE == 0) __hip_atomic_store((gu_t)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else { asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory"); }
}
// workgroups a and b play ping-pong on two words; everybody else leaves at once
template <int MODE>
__global__ void k_pingpong(unsigned long long* w, int a, int b, int iters, long long* out, int* xcc)
{
  extern __shared__ double pad[];
  const int me = blockIdx.x;
  if(threadIdx.x != 0 || (me != a && me != b)) return;
  xcc[me == a ? 0 : 1] = xcc_id();
  unsigned long long* mine = w + (me == a ? 0 : 16), *theirs = w + (me == a ? 16 : 0);
  const long long t0 = wall_clock64();
  for(int i = 1; i <= iters; i++)
  {
    if(me == a) st<MODE>(mine, (unsigned long long)i);
    int spins = 0;
    while(ld<MODE>(theirs) < (unsigned long long)i) { if(++spins > (1 << 22)) { out[2] = -1; return; } }
    if(me == b) st<MODE>(mine, (unsigned long long)i);
  }
  if(me == a) out[0] = wall_clock64() - t0;
}
int main()
{
  int* d; hipMalloc(&d, 4096*4);
  const int lds = 90*1024;          // one workgroup per CU
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_where), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(k_where, dim3(1024), dim3(64), lds, 0, d);
  std::vector<int> h(1024); hipMemcpy(h.data(), d, 1024*4, hipMemcpyDeviceToHost);
  int same = 0; for(int i = 0; i < 1024; i++) same += h[i] == (i & 7);
  printf("XCC_ID of workgroups 0..23:"); for(int i = 0; i < 24; i++) printf(" %d", h[i]); printf("\n%d of 1024 workgroups have XCC_ID == index mod 8\n", same);
  unsigned long long* w; hipMalloc(&w, 4096); long long* out; hipMalloc(&out, 64); int* xc; hipMalloc(&xc, 64);
  auto run = [&](int mode, int a, int b) {
    hipMemset(w, 0, 4096); hipMemset(out, 0, 64);
    const int iters = 2000;
    if(mode == 0) { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pingpong<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL(k_pingpong<0>, dim3(64), dim3(64), lds, 0, w, a, b, iters, out, xc); }
    else          { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pingpong<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL(k_pingpong<1>, dim3(64), dim3(64), lds, 0, w, a, b, iters, out, xc); }
    hipDeviceSynchronize();
    long long o[3]; int x[2]; hipMemcpy(o, out, 24, hipMemcpyDeviceToHost); hipMemcpy(x, xc, 8, hipMemcpyDeviceToHost);
    printf("  workgroups %2d (XCD %d) <-> %2d (XCD %d), %s: %s one way %.0f ns\n", a, x[0], b, x[1], mode == 0 ? "sc1 store / sc1 load (memory)" : "plain store / sc0 load (L2)      ",
           o[2] < 0 ? "TIMED OUT," : "", o[0]*10.0/(2.0*iters));
  };
  for(int rep = 0; rep < 2; rep++)
  {
    run(0, 0, 8); run(1, 0, 8); run(0, 0, 1); run(1, 0, 1); run(0, 3, 43); run(1, 3, 43);
  }
  return 0;
}
