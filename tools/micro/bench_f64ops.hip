// issue cost and dependent latency of the fp64 vector instructions the panel sweep is made of (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(double* out, long long* cyc, double a, double b)
{
  double x0 = a + threadIdx.x, x1 = a*2, x2 = a*3, x3 = a*4, x4 = a*5, x5 = a*6, x6 = a*7, x7 = a*8;
  int iv = threadIdx.x;
  __syncthreads();
  const long long t0 = clock64();
#pragma unroll 1
  for(int it = 0; it < 64; it++)
  {
    if(MODE == 0) {
#pragma unroll
      for(int u = 0; u < 32; u++) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x0) : "v"(b));
    } else if(MODE == 1) {
#pragma unroll
      for(int u = 0; u < 4; u++) {
        asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x0) : "v"(b)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x1) : "v"(b));
        asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x2) : "v"(b)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x3) : "v"(b));
        asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x4) : "v"(b)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x5) : "v"(b));
        asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x6) : "v"(b)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x7) : "v"(b));
      }
    } else if(MODE == 2) {
#pragma unroll
      for(int u = 0; u < 32; u++) asm volatile("v_rsq_f64 %0, %0" : "+v"(x0));
    } else if(MODE == 3) {
#pragma unroll
      for(int u = 0; u < 4; u++) {
        asm volatile("v_rsq_f64 %0, %0" : "+v"(x0)); asm volatile("v_rsq_f64 %0, %0" : "+v"(x1));
        asm volatile("v_rsq_f64 %0, %0" : "+v"(x2)); asm volatile("v_rsq_f64 %0, %0" : "+v"(x3));
        asm volatile("v_rsq_f64 %0, %0" : "+v"(x4)); asm volatile("v_rsq_f64 %0, %0" : "+v"(x5));
        asm volatile("v_rsq_f64 %0, %0" : "+v"(x6)); asm volatile("v_rsq_f64 %0, %0" : "+v"(x7));
      }
    } else if(MODE == 4) {
#pragma unroll
      for(int u = 0; u < 32; u++) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x0) : "v"(b));
    } else if(MODE == 5) {
#pragma unroll
      for(int u = 0; u < 32; u++) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(iv) : "v"(it) : "vcc");
    } else if(MODE == 6) {
#pragma unroll
      for(int u = 0; u < 4; u++) {
        asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x0) : "v"(b)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x1) : "v"(b));
        asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x2) : "v"(b)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x3) : "v"(b));
        asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x4) : "v"(b)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x5) : "v"(b));
        asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x6) : "v"(b)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x7) : "v"(b));
      }
    } else if(MODE == 7) {
#pragma unroll
      for(int u = 0; u < 32; u++) asm volatile("v_rcp_f64 %0, %0" : "+v"(x0));
    }
  }
  const long long t1 = clock64();
  out[threadIdx.x + blockIdx.x*blockDim.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + iv;
  if(threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char* name, int nt)
{
  double* o; long long* c; hipMalloc(&o, 8*1024); hipMalloc(&c, 8);
  long long h = 0;
  for(int r = 0; r < 2; r++) { hipLaunchKernelGGL(HIP_KERNEL_NAME(k<MODE>), dim3(1), dim3(nt), 0, 0, o, c, 1.0000001, 0.9999999); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); }
  printf("%-34s %4d threads: %.1f cycles per instruction\n", name, nt, (double)h/(64.0*32.0));
  hipFree(o); hipFree(c);
}
int main()
{
  for(int nt : {64, 512})
  {
    run<0>("v_fma_f64 dependent", nt); run<1>("v_fma_f64 8 independent chains", nt);
    run<4>("v_mul_f64 dependent", nt); run<6>("v_mul_f64 8 independent chains", nt);
    run<2>("v_rsq_f64 dependent", nt); run<3>("v_rsq_f64 8 independent chains", nt);
    run<7>("v_rcp_f64 dependent", nt); run<5>("v_cndmask_b32 dependent", nt);
  }
  return 0;
}
