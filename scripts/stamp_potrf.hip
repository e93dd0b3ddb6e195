// Diagnostic build: where does the factor wave of k_potrf_rr spend its cycles?  (not part of the product)
//   hipcc --offload-arch=gfx950 -O3 -DCHOL_STAMPS -Iinclude -Icholesky_amd/csrc scripts/stamp_potrf.hip -o scripts/stamp_potrf
#include "../cholesky_amd/csrc/chol_kernels.hip"
#include <cstdio>
#include <vector>
#include <cmath>
int main(int argc, char **argv)
{
  int n = argc > 1 ? atoi(argv[1]) : 256;
  std::vector<double> A((size_t)n * n);
  for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) A[i + (size_t)j * n] = (i == j) ? 4.0 + n : 1.0 / (1.0 + abs(i - j));
  double *dA, *dW; int *dinfo; chol_potrf_desc *dd;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&dW, 64 * 256 * 8); hipMalloc(&dinfo, 8); hipMalloc(&dd, sizeof(chol_potrf_desc));
  chol_potrf_desc d = { 0, 0, n, n, 1, 0 };
  hipMemcpy(dd, &d, sizeof d, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; rep++) {
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemset(dinfo, 0, 8);
    hipLaunchKernelGGL(k_potrf_rr, dim3(1), dim3(RR_THREADS), 0, 0, dA, dW, dd, dinfo);
    hipDeviceSynchronize();
  }
  unsigned long long st[16];
  hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof st);
  const char *names[8] = { "", "convert to rows", "chol16", "linv4+publish L", "wait cUpd (tile waves)", "solve (k+1,k)+publish", "update (k+1,k+1)+store", "-" };
  int T = (n + 15) / 16;
  unsigned long long tot = 0;
  for (int i = 1; i < 8; i++) tot += st[i];
  printf("n=%d steps=%d total %.1f cycles/step (s_memtime ticks @100MHz? see ratio)\n", n, T, (double)tot / T);
  for (int i = 1; i < 8; i++) printf("  %-18s %10.1f ticks/step  %5.1f%%\n", names[i], (double)st[i] / T, 100.0 * st[i] / tot);
  const char *n2[8] = { "", "wait fL (factor)", "wait cUpd (others)", "solve loop", "cSol barrier", "wait fP", "update chain", "" };
  printf("tile wave 0:\n");
  unsigned long long tt = 0; for (int i = 1; i < 7; i++) tt += st[8 + i];
  for (int i = 1; i < 7; i++) printf("  %-22s %10.1f ticks/step %5.1f%%\n", n2[i], (double)st[8 + i] / T, 100.0 * st[8 + i] / tt);
  if (argc > 2) { // per-step timeline: start of each step relative to the factor wave's step 0, per wave
    static unsigned long long tr[12][20][8];
    hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_trace), sizeof tr);
    const unsigned long long t0 = tr[0][0][0];
    printf("step: factor[start chol16-done fL LAwait-done fP] | tile waves: fL-seen solve-done cSol-done upd-done (w0, w3 light, w10)\n");
    for (int k = 0; k < T && k < 20; k++) {
      printf("%2d: F %7lld %7lld %7lld %7lld %7lld |", k, (long long)(tr[0][k][0] - t0), (long long)(tr[0][k][2] - t0), (long long)(tr[0][k][3] - t0), (long long)(tr[0][k][4] - t0), (long long)(tr[0][k][5] - t0));
      const int ws[3] = { 1, 4, 11 };
      for (int j = 0; j < 3; j++) { const int w = ws[j]; printf(" [%7lld %7lld %7lld %7lld]", (long long)(tr[w][k][1] - t0), (long long)(tr[w][k][3] - t0), (long long)(tr[w][k][4] - t0), (long long)(tr[w][k][6] - t0)); }
      printf("\n");
    }
  }
  return 0;
}
