// microbenchmark: does a v_fma_f64 / v_readlane cost less when only 16 or 32 lanes of the wave are active?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *out, unsigned long long *t, int iters, int active)
{
  double b = 0.999, c = 1e-3;
  double x0 = 1.0 + threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  unsigned long long t0 = 0, t1 = 0;
  if ((int)threadIdx.x < active) {
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int i = 0; i < iters; i++) { x0 = fma(x0, b, c); x1 = fma(x1, b, c); x2 = fma(x2, b, c); x3 = fma(x3, b, c); }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "memory");
  }
  out[threadIdx.x] = x0 + x1 + x2 + x3;
  if (threadIdx.x == 0) t[0] = t1 - t0;
}
int main()
{
  double *o; unsigned long long *t, h;
  hipMalloc(&o, 64 * 8); hipMalloc(&t, 8);
  const int iters = 2000;
  for (int active : { 64, 48, 32, 16, 1 }) {
    for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, t, iters, active); hipDeviceSynchronize(); }
    hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("active lanes %2d: independent v_fma_f64 %.1f cycles each\n", active, h / (4.0 * iters));
  }
  return 0;
}
