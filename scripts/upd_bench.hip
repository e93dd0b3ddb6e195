// microbenchmark: how fast can 11-12 waves of one CU run "tile -= P(i) P(j)^T" updates with operands in LDS?
//   hipcc --offload-arch=gfx950 -O3 scripts/upd_bench.hip -o scripts/upd_bench
// variants: 0 = one tile at a time (8 ds_read_b64 + 4 dependent MFMAs), 1 = two tiles sharing the column operand
// (12 reads + 8 MFMAs, two chains), 2 = one tile at a time, operands of the next tile loaded before the MFMAs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
#define NT 11
template <int VAR> __global__ __launch_bounds__(768) void k(double *out, unsigned long long *cyc, int iters, int nwaves)
{
  __shared__ double sS[17][256];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int t = threadIdx.x; t < 17 * 256; t += 768) sS[t >> 8][t & 255] = 1e-3 * (t % 97);
  __syncthreads();
  if (wave >= nwaves) return;
  d4 tile[NT];
#pragma unroll
  for (int s = 0; s < NT; ++s) tile[s] = (d4){ 0.0, 0.0, 0.0, 0.0 };
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int it = 0; it < iters; ++it) {
    int lp = lane;
    asm volatile("" : "+v"(lp));
    int ti = (it + wave) % 13 + 3, tj = (it * 5 + wave) % 3;
    asm volatile("" : "+s"(ti), "+s"(tj));
    if (VAR == 0) {
#pragma unroll
      for (int s = 0; s < NT; ++s) {
        d4 acc = tile[s];
        const double *pa = &sS[(tj + s) % 17][0], *pb = &sS[(ti + s) % 17][0];
#pragma unroll
        for (int st = 0; st < 4; ++st) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[st * 64 + lp], -pb[st * 64 + lp], acc, 0, 0, 0);
        tile[s] = acc;
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (VAR == 1) {
#pragma unroll
      for (int s = 0; s + 1 < NT; s += 2) {
        d4 a0 = tile[s], a1 = tile[s + 1];
        const double *pa = &sS[(tj + s) % 17][0], *pb0 = &sS[(ti + s) % 17][0], *pb1 = &sS[(ti + s + 1) % 17][0];
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          const double av = pa[st * 64 + lp];
          a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, -pb0[st * 64 + lp], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, -pb1[st * 64 + lp], a1, 0, 0, 0);
        }
        tile[s] = a0; tile[s + 1] = a1;
        __builtin_amdgcn_sched_barrier(0);
      }
      { // odd one
        d4 acc = tile[NT - 1];
        const double *pa = &sS[(tj + NT - 1) % 17][0], *pb = &sS[(ti + NT - 1) % 17][0];
#pragma unroll
        for (int st = 0; st < 4; ++st) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[st * 64 + lp], -pb[st * 64 + lp], acc, 0, 0, 0);
        tile[NT - 1] = acc;
      }
    } else {
      double xa[4], xb[4];
#pragma unroll
      for (int st = 0; st < 4; ++st) { xa[st] = sS[tj % 17][st * 64 + lp]; xb[st] = sS[ti % 17][st * 64 + lp]; }
#pragma unroll
      for (int s = 0; s < NT; ++s) {
        double na[4], nb[4];
        if (s + 1 < NT) {
#pragma unroll
          for (int st = 0; st < 4; ++st) { na[st] = sS[(tj + s + 1) % 17][st * 64 + lp]; nb[st] = sS[(ti + s + 1) % 17][st * 64 + lp]; }
        }
        d4 acc = tile[s];
#pragma unroll
        for (int st = 0; st < 4; ++st) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[st], -xb[st], acc, 0, 0, 0);
        tile[s] = acc;
        if (s + 1 < NT) {
#pragma unroll
          for (int st = 0; st < 4; ++st) { xa[st] = na[st]; xb[st] = nb[st]; }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  double sum = 0.0;
#pragma unroll
  for (int s = 0; s < NT; ++s) sum += tile[s][0] + tile[s][1] + tile[s][2] + tile[s][3];
  out[threadIdx.x] = sum;
  if (lane == 0) cyc[wave] = t1 - t0;
}
int main()
{
  double *o; unsigned long long *c, h[12];
  hipMalloc(&o, 768 * 8); hipMalloc(&c, 96);
  const int iters = 200;
  for (int var = 0; var < 3; var++)
    for (int nw : { 1, 4, 8, 12 }) {
      for (int r = 0; r < 2; r++) {
        if (var == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(768), 0, 0, o, c, iters, nw);
        if (var == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(768), 0, 0, o, c, iters, nw);
        if (var == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(768), 0, 0, o, c, iters, nw);
        hipDeviceSynchronize();
      }
      hipMemcpy(h, c, 96, hipMemcpyDeviceToHost);
      unsigned long long mx = 0; for (int w = 0; w < nw; w++) if (h[w] > mx) mx = h[w];
      const double per_tile_wave = (double)mx / (iters * NT);
      const double waves_per_simd = nw / 4.0 < 1 ? 1 : nw / 4.0;
      printf("variant %d, %2d waves: %.0f cycles per tile update per wave, %.0f per SIMD (MFMA floor 256)\n", var, nw, per_tile_wave, per_tile_wave / waves_per_simd);
    }
  return 0;
}
