// Development aid: the macro-tile update kernel on one large SYRK target (n x n lower, one source of depth k), TF/s
//   hipcc --offload-arch=gfx950 -O3 -Iinclude -Icholesky_amd/csrc scripts/mt_bench.hip -o scripts/mt_bench
#include "../cholesky_amd/csrc/chol_kernels.hip"
#if MKB >= 16
// the 128 x 128 / 8-wave instance of the macro-tile body: measured here only (slower than the product's 64 x 64: chol_kernels.hip)
__global__ __launch_bounds__(512) void k_update_mt128(double *__restrict__ base, const chol_upd_task *__restrict__ tasks,
                                                      const chol_upd_src *__restrict__ srcs, int ntask, int per_xcd)
{
  const int tid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if ((blockIdx.x >> 3) >= per_xcd || tid >= ntask) return;
  update_mt_body<128, 128, 4, 2>(base, tasks[tid], srcs);
}
__global__ __launch_bounds__(512) void k_update_mt128x64(double *__restrict__ base, const chol_upd_task *__restrict__ tasks,
                                                        const chol_upd_src *__restrict__ srcs, int ntask, int per_xcd)
{
  const int tid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if ((blockIdx.x >> 3) >= per_xcd || tid >= ntask) return;
  update_mt_body<128, 64, 4, 2>(base, tasks[tid], srcs);
}
static int chol_launch_update_mt128x64(double *base, const chol_upd_task *tasks, const chol_upd_src *srcs, int ntask, hipStream_t st)
{
  const int per_xcd = (ntask + 7) / 8;
  hipLaunchKernelGGL(k_update_mt128x64, dim3(per_xcd * 8), dim3(512), 0, st, base, tasks, srcs, ntask, per_xcd);
  return (int)hipGetLastError();
}
static int chol_launch_update_mt128(double *base, const chol_upd_task *tasks, const chol_upd_src *srcs, int ntask, hipStream_t st)
{
  const int per_xcd = (ntask + 7) / 8;
  hipLaunchKernelGGL(k_update_mt128, dim3(per_xcd * 8), dim3(512), 0, st, base, tasks, srcs, ntask, per_xcd);
  return (int)hipGetLastError();
}
#else
static int chol_launch_update_mt128(double *, const chol_upd_task *, const chol_upd_src *, int, hipStream_t) { return -1; }
static int chol_launch_update_mt128x64(double *, const chol_upd_task *, const chol_upd_src *, int, hipStream_t) { return -1; }
#endif
#include <cstdio>
#include <vector>
#include <cstring>
#include <algorithm>
#include <cmath>
int main(int argc, char **argv)
{
  const int n = argc > 1 ? atoi(argv[1]) : 8192, k = argc > 2 ? atoi(argv[2]) : 144;
  double *dC, *dX;
  hipMalloc(&dC, (size_t)n * n * 8); hipMalloc(&dX, (size_t)n * k * 8);
  hipMemset(dC, 0, (size_t)n * n * 8);
  std::vector<double> X((size_t)n * k);
  for (size_t i = 0; i < X.size(); i++) X[i] = 1e-3 * (double)(i % 977);
  hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice);
  std::vector<chol_upd_task> tasks;
  const int TSZ = argc > 4 ? atoi(argv[4]) : 64; // macro tile: 64 (k_update_mt), 128 (128 x 128) or 12864 (128 rows x 64 columns)
  const int TR = TSZ == 12864 ? 128 : TSZ, TC = TSZ == 12864 ? 64 : TSZ;
  const int ntr = (n + TR - 1) / TR, ntc = (n + TC - 1) / TC;
  const int BL = argc > 3 ? atoi(argv[3]) : 1; // tasks enumerated in BL x BL blocks of tiles
  for (int A0 = 0; A0 < ntr; A0 += BL)
   for (int B0 = 0; B0 < ntc; B0 += BL)
    for (int a = A0; a < std::min(A0 + BL, ntr); a++)
    for (int b = B0; b < std::min(B0 + BL, ntc); b++) {
      if (b * TC > a * TR + TR - 1) continue; // entirely above the diagonal
      chol_upd_task t = {};
      t.c_off = (int64_t)((uintptr_t)dC / 8) + a * TR + (int64_t)b * TC * n; t.ldc = n;
      t.mv = (short)std::min(TR, n - a * TR); t.nv = (short)std::min(TC, n - b * TC);
      t.lower = b * TC + TC - 1 > a * TR; // the diagonal passes through the tile
      t.src_begin = 0; t.src_end = 1; t.ar = a * TR; t.br = b * TC;
      tasks.push_back(t);
    }
  chol_upd_src src = { (int64_t)((uintptr_t)dX / 8), (int64_t)((uintptr_t)dX / 8), n, n, k, 0, 0, 0 };
  chol_upd_task *dt; chol_upd_src *ds;
  hipMalloc(&dt, tasks.size() * sizeof(chol_upd_task)); hipMalloc(&ds, sizeof src);
  hipMemcpy(dt, tasks.data(), tasks.size() * sizeof(chol_upd_task), hipMemcpyHostToDevice);
  hipMemcpy(ds, &src, sizeof src, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; rep++) {
    hipEventRecord(e0);
    if (TSZ == 128) chol_launch_update_mt128(nullptr, dt, ds, (int)tasks.size(), 0); else if (TSZ == 12864) chol_launch_update_mt128x64(nullptr, dt, ds, (int)tasks.size(), 0); else chol_launch_update_mt(nullptr, dt, ds, (int)tasks.size(), 0, 0);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flops = (double)tasks.size() * TR * TC * 2.0 * k;
  // a few entries against the host (row sums of X X^T)
  { std::vector<double> C((size_t)n * 4); hipMemcpy(C.data(), dC, C.size() * 8, hipMemcpyDeviceToHost); double err = 0; for (int c = 0; c < 4; c++) for (int r = c; r < n; r += 997) { double sum = 0; for (int q = 0; q < k; q++) sum += X[r + (size_t)q * n] * X[c + (size_t)q * n]; err = std::max(err, fabs(C[r + (size_t)c * n] + 5.0 * sum) / (1.0 + fabs(sum))); } printf("check (5 launches accumulate): rel err %.2e\n", err); }
  printf("TS=%d BL=%d n=%d k=%d: %zu macro tiles, %.3f ms, %.1f TF/s executed\n", TSZ, BL, n, k, tasks.size(), best, flops / best * 1e-9);
#ifdef MT_CLOCK
  { unsigned long long c[2]; hipMemcpyFromSymbol(c, HIP_SYMBOL(g_mt_clock), sizeof c);
    printf("   one workgroup mid-grid: %llu shader cycles in %.2f us on its tile -> in-kernel clock %.3f GHz; %.0f cycles per 16-deep chunk\n", c[0], c[1] * 0.01, (double)c[0] / (double)c[1] * 0.1, (double)c[0] / (k / 16)); }
#endif
  return 0;
}
