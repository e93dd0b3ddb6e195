// Development aid: the fp32 macro-tile update kernel (k32_update_mt, 64 x 64 tiles) on one large SYRK target (n x n lower, one
// source of depth k), TF/s.   hipcc --offload-arch=gfx950 -O3 -Iinclude -Icholesky_amd/csrc scripts/mt_bench32.hip -o scripts/mt_bench32
//   ./mt_bench32 [n] [k] [row offset of the source: 0 = 16-byte aligned rows, 1 = only 4-byte aligned]   (-DM32_NODMA / -DM32_NOBAR: timing diagnostics)
#include "../cholesky_amd/csrc/chol_kernels_f32.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
#include <cmath>
#include <climits>
int main(int argc, char **argv)
{
  const int n = argc > 1 ? atoi(argv[1]) : 8192, k = argc > 2 ? atoi(argv[2]) : 384, off = argc > 3 ? atoi(argv[3]) : 0;
  const int ld = n + off + 3; // a leading dimension that is no multiple of four
  float *dC, *dX;
  hipMalloc(&dC, (size_t)n * n * 4); hipMalloc(&dX, ((size_t)ld * k + 16) * 4);
  hipMemset(dC, 0, (size_t)n * n * 4);
  std::vector<float> X((size_t)ld * k + 16);
  for (size_t i = 0; i < X.size(); i++) X[i] = 1e-2f * (float)(i % 97) - 0.3f;
  hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
  std::vector<chol_upd_task> tasks;
  const int T = 64, nt = (n + T - 1) / T;
  for (int a = 0; a < nt; a++)
    for (int b = 0; b <= a; b++) {
      chol_upd_task t = {};
      t.c_off = (int64_t)((uintptr_t)dC / 4) + a * T + (int64_t)b * T * n; t.ldc = n;
      t.mv = (short)std::min(T, n - a * T); t.nv = (short)std::min(T, n - b * T);
      t.lower = a == b; t.src_begin = 0; t.src_end = 1; t.ar = a * T; t.br = b * T;
      tasks.push_back(t);
    }
  chol_upd_src src = { (int64_t)((uintptr_t)dX / 4) + off, (int64_t)((uintptr_t)dX / 4) + off, ld, ld, k, 0, 0, 0 };
  chol_upd_task *dt; chol_upd_src *ds;
  hipMalloc(&dt, tasks.size() * sizeof(chol_upd_task)); hipMalloc(&ds, sizeof src);
  hipMemcpy(dt, tasks.data(), tasks.size() * sizeof(chol_upd_task), hipMemcpyHostToDevice);
  hipMemcpy(ds, &src, sizeof src, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; rep++) {
    hipEventRecord(e0);
    chol32_launch_update_mt(nullptr, dt, ds, (int)tasks.size(), LLONG_MAX, 0);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double flops = 0;
  for (auto &t : tasks) flops += 2.0 * t.mv * t.nv * k;
  { // entries against the host
    std::vector<float> C((size_t)n * n); hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double err = 0; int upper = 0;
    for (int c : { 0, 1, 2, 3, n / 2 + 1, n - 5, n - 1 })
      for (int r = 0; r < n; r += 331) {
        if (r < c) { if (C[r + (size_t)c * n] != 0.f) upper++; continue; }
        double sum = 0; for (int q = 0; q < k; q++) sum += (double)X[off + r + (size_t)q * ld] * X[off + c + (size_t)q * ld];
        err = std::max(err, fabs(C[r + (size_t)c * n] + 5.0 * sum) / (1.0 + fabs(5.0 * sum)));
      }
    printf("check (5 launches accumulate): rel err %.2e, %d entries above the diagonal touched\n", err, upper);
  }
  printf("fp32 tile %d n=%d k=%d off=%d: %zu macro tiles, %.3f ms, %.1f TF/s\n", T, n, k, off, tasks.size(), best, flops / best * 1e-9);
#ifdef M32_CLOCK
  { unsigned long long c[2]; hipMemcpyFromSymbol(c, HIP_SYMBOL(g_m32_clock), sizeof c);
    printf("   one workgroup mid-grid: %llu shader cycles in %.2f us on its tile -> in-kernel clock %.3f GHz; %.0f cycles per 16-deep chunk\n", c[0], c[1] * 0.01, (double)c[0] / (double)c[1] * 0.1, (double)c[0] / (k / 16)); }
#endif
  return 0;
}
