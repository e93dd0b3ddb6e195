// Development aid: time k_trsm_w / k_trsm_rr on synthetic strips (not part of the product)
//   hipcc --offload-arch=gfx950 -O3 -Iinclude -Icholesky_amd/csrc scripts/time_trsm_w.hip -o scripts/time_trsm_w
#include "../cholesky_amd/csrc/chol_kernels.hip"
#include <cstdio>
#include <vector>
int main(int argc, char **argv)
{
  int n = argc > 1 ? atoi(argv[1]) : 128, strips = argc > 2 ? atoi(argv[2]) : 8, wpb = argc > 3 ? atoi(argv[3]) : 1;
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
  for (int which = 0; which < 2; which++) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
      hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL(k_trsm_w, dim3((strips + 3) / 4), dim3(256), 0, 0, (double *)nullptr, (double *const *)nullptr, (const double *)nullptr, dd, strips);
      else hipLaunchKernelGGL(k_trsm_rr, dim3(strips), dim3(256), 0, 0, (double *)nullptr, (double *const *)nullptr, (const double *)nullptr, dd);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%s n=%d strips=%d: %.1f us\n", which == 0 ? "k_trsm_w " : "k_trsm_rr", n, strips, best * 1e3);
#ifdef CHOL_STAMPS
    if (which == 0) {
      unsigned long long st[16];
      hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof st);
      printf("   wave 0 of strip group 0 (cycles): issue loads %llu, wait vmcnt %llu, barriers+fix %llu, solve %llu\n", st[1], st[2], st[3], st[4]);
    }
#endif
  }
  return 0;
}
