// Development aid: the fp64 MFMA rate the whole chip sustains (v_mfma_f64_16x16x4_f64, four independent accumulators per wave,
// 16 waves per CU, 1024 workgroups, operands in registers): what 78.6 TF/s (2.4 GHz) becomes under load.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_peak.hip -o scripts/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_peak(double *out, int iters, double a0)
{
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double a = a0 + threadIdx.x * 1e-9, b = 1.0 - a;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
// the same with eight accumulators per wave (dependency distance 8) and a clock read: cycles per MFMA as the wave sees them
__global__ __launch_bounds__(256) void k_peak8(double *out, int iters, double a0, unsigned long long *clk)
{
  d4 c[8];
  for (int q = 0; q < 8; ++q) c[q] = (d4){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = 1.0 - a;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < 8; ++q) c[q] = __builtin_amdgcn_mfma_f64_16x16x4f64((q & 1) ? a : b, (q & 2) ? a : b, c[q], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double sum = 0; for (int q = 0; q < 8; ++q) sum += c[q][q & 3];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
  // stamps of one wave of a workgroup in the MIDDLE of the grid (it runs beside a full chip): shader cycles (s_memtime) and the
  // 100 MHz real-time clock (s_memrealtime) over the loop: in-kernel clock = d(memtime) / d(memrealtime) x 100 MHz
  if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k_peak32(float *out, int iters, float a0)
{ // v_mfma_f32_32x32x2_f32: 4096 flop each, four accumulators per wave
  f16v c0, c1, c2, c3;
  for (int q = 0; q < 16; ++q) { c0[q] = 0.f; c1[q] = 0.f; c2[q] = 0.f; c3[q] = 0.f; }
  float a = a0 + threadIdx.x * 1e-6f, b = 1.0f - a;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main(int argc, char **argv)
{
  const int wgs = argc > 1 ? atoi(argv[1]) : 1024 * 4, iters = argc > 2 ? atoi(argv[2]) : 20000;
  double *d; hipMalloc(&d, (size_t)wgs * 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_peak, dim3(wgs), dim3(256), 0, 0, d, iters, 0.25);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)wgs * 4 * iters * 4 * 2048.0;
    printf("%d workgroups x 4 waves x %d x 4 MFMA: %.3f ms, %.1f TF/s fp64\n", wgs, iters, ms, flops / ms * 1e-9);
  }
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_peak32, dim3(wgs), dim3(256), 0, 0, (float *)d, iters, 0.25f);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)wgs * 4 * iters * 4 * 4096.0;
    printf("fp32 v_mfma_f32_32x32x2_f32, 4 accumulators: %.3f ms, %.1f TF/s\n", ms, flops / ms * 1e-9);
  }
  unsigned long long *dc; hipMalloc(&dc, 16);
  int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_peak8, 256, 0);
  printf("k_peak8: %d workgroups of 4 waves resident per CU (occupancy API) = %d waves per SIMD\n", occ, occ);
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_peak8, dim3(wgs), dim3(256), 0, 0, d, iters / 2, 0.25, dc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
    const double flops = (double)wgs * 4 * (iters / 2) * 8 * 2048.0, nm = (double)(iters / 2) * 8;
    const double ghz = (double)c[0] / (double)c[1] * 0.1; // shader cycles per 10 ns tick
    const double per_simd_ns = 1024.0 * 2048.0 / (flops / ms * 1e-6) * 1e0; // ns per MFMA and SIMD at the measured chip rate (1024 SIMDs)
    printf("8 accumulators: %.3f ms, %.1f TF/s fp64; one wave mid-grid: %.1f shader cycles (s_memtime) = %.1f ns (s_memrealtime) per MFMA of its own -> in-kernel clock %.3f GHz; "
           "chip rate = one MFMA per %.1f ns and SIMD = %.1f cycles at that clock\n", ms, flops / ms * 1e-9, (double)c[0] / nm, (double)c[1] * 10.0 / nm, ghz, per_simd_ns, per_simd_ns * ghz);
  }
  return 0;
}
