// Development aid: the macro-tile update kernel on one large SYRK target (n x n lower, one source of depth k), TF/s
//   hipcc --offload-arch=gfx950 -O3 -Iinclude -Icholesky_amd/csrc scripts/mt_bench.hip -o scripts/mt_bench
#include "../cholesky_amd/csrc/chol_kernels.hip"
#include <cstdio>
#include <vector>
#include <cstring>
#include <algorithm>
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
  const int nt = (n + 63) / 64;
  const int BL = argc > 3 ? atoi(argv[3]) : 1; // tasks enumerated in BL x BL blocks of tiles
  for (int A0 = 0; A0 < nt; A0 += BL)
   for (int B0 = 0; B0 <= A0; B0 += BL)
    for (int a = A0; a < std::min(A0 + BL, nt); a++)
    for (int b = B0; b < std::min(B0 + BL, nt); b++) {
      if (b > a) continue;
      chol_upd_task t = {};
      t.c_off = (int64_t)((uintptr_t)dC / 8) + a * 64 + (int64_t)b * 64 * n; t.ldc = n;
      t.mv = (short)std::min(64, n - a * 64); t.nv = (short)std::min(64, n - b * 64);
      t.lower = a == b; t.src_begin = 0; t.src_end = 1; t.ar = a * 64; t.br = b * 64;
      tasks.push_back(t);
    }
  chol_upd_src src = { (int64_t)((uintptr_t)dX / 8), (int64_t)((uintptr_t)dX / 8), n, n, k, 0 };
  chol_upd_task *dt; chol_upd_src *ds;
  hipMalloc(&dt, tasks.size() * sizeof(chol_upd_task)); hipMalloc(&ds, sizeof src);
  hipMemcpy(dt, tasks.data(), tasks.size() * sizeof(chol_upd_task), hipMemcpyHostToDevice);
  hipMemcpy(ds, &src, sizeof src, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; rep++) {
    hipEventRecord(e0);
    chol_launch_update_mt(nullptr, dt, ds, (int)tasks.size(), 0);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flops = (double)tasks.size() * 64.0 * 64.0 * 2.0 * k;
  printf("BL=%d n=%d k=%d: %zu macro tiles, %.3f ms, %.1f TF/s executed\n", BL, n, k, tasks.size(), best, flops / best * 1e-9);
  return 0;
}
