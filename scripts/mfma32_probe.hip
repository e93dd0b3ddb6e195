// Development aid: what the fp32 (and fp64) matrix pipe sustains as a function of the independent accumulators per wave and the waves per SIMD
// (v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32), to tell a dependent-issue limit from a pipe limit in k32_update_mt.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma32_probe.hip -o scripts/mfma32_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
template <int NACC> __global__ __launch_bounds__(256) void k32x32(float *out, int iters, float a0)
{
  extern __shared__ float pad[];
  f16v c[NACC];
  for (int n = 0; n < NACC; ++n) for (int q = 0; q < 16; ++q) c[n][q] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = 1.0f - a;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int n = 0; n < NACC; ++n) c[n] = __builtin_amdgcn_mfma_f32_32x32x2f32((n & 1) ? a : b, (n & 2) ? a : b, c[n], 0, 0, 0);
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) s += c[n][n];
  if (iters < 0) pad[threadIdx.x] = s;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> __global__ __launch_bounds__(256) void k16x16(float *out, int iters, float a0)
{
  extern __shared__ float pad[];
  f4v c[NACC];
  for (int n = 0; n < NACC; ++n) for (int q = 0; q < 4; ++q) c[n][q] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = 1.0f - a;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int n = 0; n < NACC; ++n) c[n] = __builtin_amdgcn_mfma_f32_16x16x4f32((n & 1) ? a : b, (n & 2) ? a : b, c[n], 0, 0, 0);
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) s += c[n][n & 3];
  if (iters < 0) pad[threadIdx.x] = s;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
typedef double d4v __attribute__((ext_vector_type(4)));
template <int NACC> __global__ __launch_bounds__(256) void k64(float *out, int iters, float a0)
{ // v_mfma_f64_16x16x4_f64, NACC independent accumulators, operands rotated over four register pairs
  extern __shared__ float pad[];
  d4v c[NACC];
  for (int n = 0; n < NACC; ++n) for (int q = 0; q < 4; ++q) c[n][q] = 0.0;
  double a = a0 + threadIdx.x * 1e-6, b = 1.0 - a, a2 = a * 0.5, b2 = b * 0.5;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int n = 0; n < NACC; ++n) c[n] = __builtin_amdgcn_mfma_f64_16x16x4f64((n & 1) ? a : b, (n & 2) ? a2 : b2, c[n], 0, 0, 0);
  }
  double s = 0.0;
  for (int n = 0; n < NACC; ++n) s += c[n][n & 3];
  if (iters < 0) pad[threadIdx.x] = (float)s;
  out[blockIdx.x * 256 + threadIdx.x] = (float)s;
}
template <class K> static void run(const char *name, K kern, int nacc, double flop_per_mfma, int per_cu, float *d)
{
  const int lds = 160 * 1024 / per_cu - 512; // dynamic LDS that lets exactly per_cu workgroups share a CU
  hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int wgs = 256 * per_cu * 4, iters = 20000 / nacc;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds, 0, d, iters, 0.25f);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("%s, %d accumulator(s) per wave, %d waves per SIMD: %.1f TF/s\n", name, nacc, per_cu, (double)wgs * 4 * iters * nacc * flop_per_mfma / best * 1e-9);
}
int main()
{
  float *d; hipMalloc(&d, (size_t)256 * 8 * 4 * 256 * 4);
  for (int per_cu : { 1, 2, 4, 8 }) {
    run("32x32x2", k32x32<1>, 1, 4096.0, per_cu, d);
    run("32x32x2", k32x32<2>, 2, 4096.0, per_cu, d);
    run("32x32x2", k32x32<4>, 4, 4096.0, per_cu, d);
    run("16x16x4", k16x16<1>, 1, 2048.0, per_cu, d);
    run("16x16x4", k16x16<4>, 4, 2048.0, per_cu, d);
    run("fp64 16x16x4", k64<1>, 1, 2048.0, per_cu, d);
    run("fp64 16x16x4", k64<2>, 2, 2048.0, per_cu, d);
    run("fp64 16x16x4", k64<4>, 4, 2048.0, per_cu, d);
  }
  return 0;
}
