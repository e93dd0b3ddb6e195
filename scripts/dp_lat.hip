// microbenchmark (one wave): dependent / independent v_fma_f64, v_rsq_f64, v_readlane
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *out, unsigned long long *t, int iters)
{
  double a = 1.0 + threadIdx.x * 1e-3, b = 0.999, c = 1e-3;
  double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
  unsigned long long t0, t1, t2, t3, t4;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int i = 0; i < iters; i++) { x0 = fma(x0, b, c); x0 = fma(x0, b, c); x0 = fma(x0, b, c); x0 = fma(x0, b, c); }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(x0) : "memory");
  for (int i = 0; i < iters; i++) { x0 = fma(x0, b, c); x1 = fma(x1, b, c); x2 = fma(x2, b, c); x3 = fma(x3, b, c); }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2) : "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "memory");
  for (int i = 0; i < iters; i++) { x0 = __builtin_amdgcn_rsq(x0 + 1.0); x0 = __builtin_amdgcn_rsq(x0 + 1.0); x0 = __builtin_amdgcn_rsq(x0 + 1.0); x0 = __builtin_amdgcn_rsq(x0 + 1.0); }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t3) : "v"(x0) : "memory");
  for (int i = 0; i < iters; i++) {
    int lo = __builtin_amdgcn_readlane(__double2loint(x1), 3), hi = __builtin_amdgcn_readlane(__double2hiint(x1), 3);
    x1 = fma(x1, __hiloint2double(hi, lo), c);
    lo = __builtin_amdgcn_readlane(__double2loint(x1), 5); hi = __builtin_amdgcn_readlane(__double2hiint(x1), 5);
    x1 = fma(x1, __hiloint2double(hi, lo), c);
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t4) : "v"(x1) : "memory");
  out[threadIdx.x] = x0 + x1 + x2 + x3;
  if (threadIdx.x == 0) { t[0] = t1 - t0; t[1] = t2 - t1; t[2] = t3 - t2; t[3] = t4 - t3; }
}
int main()
{
  double *o; unsigned long long *t, h[4];
  hipMalloc(&o, 64 * 8); hipMalloc(&t, 32);
  const int iters = 2000;
  for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, t, iters); hipDeviceSynchronize(); }
  hipMemcpy(h, t, 32, hipMemcpyDeviceToHost);
  printf("v_fma_f64: dependent %.1f cyc, independent %.1f cyc; rsq_f64(+add) dependent pair %.1f cyc; readlane x2 + fma dependent %.1f cyc\n",
         h[0] / (4.0 * iters), h[1] / (4.0 * iters), h[2] / (4.0 * iters), h[3] / (2.0 * iters));
  return 0;
}
