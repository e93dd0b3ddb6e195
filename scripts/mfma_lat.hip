// microbenchmark: dependent-chain latency and independent throughput of v_mfma_f64_16x16x4_f64 (one wave)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(double *out, unsigned long long *t, int iters)
{
  double a = threadIdx.x * 0.001, b = 1.0 - threadIdx.x * 0.002;
  d4 acc = {0, 0, 0, 0}, acc2 = {1, 1, 1, 1}, acc3 = {2, 2, 2, 2}, acc4 = {3, 3, 3, 3};
  unsigned long long t0, t1, t2, t3;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int i = 0; i < iters; i++) { // dependent through the accumulator
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(acc) : "memory");
  for (int i = 0; i < iters; i++) { // dependent through an operand (result element feeds the next B operand)
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc2[0], acc2, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc2[1], acc2, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc2[2], acc2, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc2[3], acc2, 0, 0, 0);
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2) : "v"(acc2) : "memory");
  for (int i = 0; i < iters; i++) { // four independent accumulators
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
    acc4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc4, 0, 0, 0);
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t3) : "v"(acc), "v"(acc2), "v"(acc3), "v"(acc4) : "memory");
  out[threadIdx.x] = acc[0] + acc2[1] + acc3[2] + acc4[3];
  if (threadIdx.x == 0) { t[0] = t1 - t0; t[1] = t2 - t1; t[2] = t3 - t2; }
}
int main()
{
  double *o; unsigned long long *t, h[3];
  hipMalloc(&o, 64 * 8); hipMalloc(&t, 24);
  const int iters = 1000;
  for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, t, iters); hipDeviceSynchronize(); }
  hipMemcpy(h, t, 24, hipMemcpyDeviceToHost);
  printf("per MFMA f64 16x16x4: acc-dependent %.1f cyc, operand-dependent %.1f cyc, independent x4 %.1f cyc\n", h[0] / (4.0 * iters), h[1] / (4.0 * iters), h[2] / (4.0 * iters));
  return 0;
}
