// Diagnostic build: cycle stamps of one wave of k_trsm_rr (not part of the product)
//   hipcc --offload-arch=gfx950 -O3 -DCHOL_STAMPS -Iinclude -Icholesky_amd/csrc scripts/stamp_trsm.hip -o scripts/stamp_trsm
#include "../cholesky_amd/csrc/chol_kernels.hip"
#include <cstdio>
#include <vector>
int main(int argc, char **argv)
{
  int n = argc > 1 ? atoi(argv[1]) : 256, strips = argc > 2 ? atoi(argv[2]) : 64;
  std::vector<double> L((size_t)n * n, 0.0), B((size_t)strips * 16 * n, 1.0);
  for (int j = 0; j < n; j++) for (int i = j; i < n; i++) L[i + (size_t)j * n] = (i == j) ? 2.0 + 0.01 * i : 0.01 / (1.0 + i - j);
  double *dL, *dB, *dW; chol_trsm_desc *dd;
  hipMalloc(&dL, L.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dW, 64 * 256 * 8); hipMalloc(&dd, strips * sizeof(chol_trsm_desc));
  hipMemcpy(dL, L.data(), L.size() * 8, hipMemcpyHostToDevice);
  std::vector<chol_trsm_desc> d(strips);
  const int ldb = strips * 16;
  for (int s = 0; s < strips; s++) d[s] = { (int64_t)((uintptr_t)dL / 8), (int64_t)((uintptr_t)dW / 8), (int64_t)((uintptr_t)dB / 8) + s * 16, n, n, 16, ldb };
  hipMemcpy(dd, d.data(), strips * sizeof(chol_trsm_desc), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_dinv, dim3((n + 15) / 16), dim3(64), 0, 0, dL, n, n, dW);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; rep++) {
    hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_trsm_rr, dim3(strips), dim3(256), 0, 0, (double *)nullptr, (double *const *)nullptr, (const double *)nullptr, dd);
    hipEventRecord(e1);
    hipDeviceSynchronize();
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long st[16];
  hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof st);
  const char *names[8] = { "", "barrier wait", "chain or updates", "issue prefetch", "-", "", "", "" };
  int T = (n + 15) / 16;
  unsigned long long tot = 0;
  for (int i = 1; i < 4; i++) tot += st[i];
  printf("trsm n=%d strips=%d: kernel %.1f us; wave 1 of strip 0: %.1f cycles/step\n", n, strips, ms * 1e3, (double)tot / T);
  for (int i = 1; i < 4; i++) printf("  %-34s %10.1f cycles/step  %5.1f%%\n", names[i], (double)st[i] / T, 100.0 * st[i] / tot);
  return 0;
}
