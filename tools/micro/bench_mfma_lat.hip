// tools only: latency of dependent v_mfma_f64_16x16x4_f64 chains, of an MFMA result read by the VALU, of v_readlane
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(long long* out, double* sink, int mode)
{
  double a = threadIdx.x*0.001 + 1.0, b = 0.5;
  v4d acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
  long long t0 = clock64();
  if(mode == 0) { for(int i = 0; i < 64; i++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0); }
  if(mode == 1) { for(int i = 0; i < 64; i++) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0); } }
  if(mode == 2) { for(int i = 0; i < 64; i++) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0); a = acc[0]*0.5; } }        // MFMA -> VALU -> MFMA
  if(mode == 3) { for(int i = 0; i < 64; i++) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0); a = acc[0]; } }            // result as the next A operand
  if(mode == 4) { for(int i = 0; i < 64; i++) { a = a*1.0000001 + b; } }                                                                     // dependent FMA
  if(mode == 5) { for(int i = 0; i < 64; i++) { int lo = __builtin_amdgcn_readlane((int)__double2loint(a), 5); a = a*1.0000001 + (double)lo*1e-300; } }  // FMA -> readlane -> FMA
  if(mode == 6) { for(int i = 0; i < 64; i++) { a = __builtin_amdgcn_rsq(a) + 1.0; } }
  long long t1 = clock64();
  if(threadIdx.x == 0) out[0] = t1 - t0;
  sink[threadIdx.x] = acc[0] + acc[1] + acc2[2] + a;
}
int main()
{
  long long* o; double* s; hipMalloc(&o, 8); hipMalloc(&s, 8*64);
  const char* names[] = {"dependent MFMA chain", "two independent MFMA chains (per pair)", "MFMA -> v_mul -> MFMA (A operand)", "MFMA -> MFMA (result as A operand)", "dependent v_fma_f64", "fma -> readlane -> cvt -> fma", "rsq + add"};
  for(int m = 0; m < 7; m++)
  {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, s, m); hipDeviceSynchronize();
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, s, m); hipDeviceSynchronize();
    long long h; hipMemcpy(&h, o, 8, hipMemcpyDeviceToHost);
    printf("%-44s %.1f clocks per iteration\n", names[m], h/64.0);
  }
  return 0;
}
