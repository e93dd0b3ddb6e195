// Diagnostic build: where does the factor wave of k_potrf_rr spend its cycles?  (not part of the product)
//   hipcc --offload-arch=gfx950 -O3 -DCHOL_STAMPS -Iinclude -Icholesky_amd/csrc scripts/stamp_potrf.hip -o scripts/stamp_potrf
#include "../cholesky_amd/csrc/chol_kernels.hip"
#include <cstdio>
#include <vector>
#include <cmath>
int main(int argc, char **argv)
{
  int n = argc > 1 ? atoi(argv[1]) : 256;
  const int band = argc > 3 ? atoi(argv[3]) : 0; // > 0: banded matrix (band in tiles) with the matching skyline in the descriptor
  std::vector<double> A((size_t)n * n);
  for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) A[i + (size_t)j * n] = (i == j) ? 4.0 + n : (band > 0 && abs(i / 16 - j / 16) > band) ? 0.0 : 1.0 / (1.0 + abs(i - j));
  double *dA, *dW; int *dinfo; chol_potrf_desc *dd;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&dW, 64 * 256 * 8); hipMalloc(&dinfo, 8); hipMalloc(&dd, sizeof(chol_potrf_desc));
  chol_potrf_desc d = { 0, 0, n, n, 1, 0 };
  if (band > 0) for (int i = 0; i < 24; i++) d.sky[i] = (unsigned char)(i > band ? i - band : 0);
  hipMemcpy(dd, &d, sizeof d, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 6; rep++) {
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemset(dinfo, 0, 8);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_potrf_rr, dim3(1), dim3(RR_THREADS), 0, 0, dA, dW, dd, dinfo);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  { // check against the host factorisation
    std::vector<double> L(A), G(A.size());
    hipMemcpy(G.data(), dA, A.size() * 8, hipMemcpyDeviceToHost);
    for (int j = 0; j < n; j++) {
      for (int k = 0; k < j; k++) { const double l = L[j + (size_t)k * n]; for (int i = j; i < n; i++) L[i + (size_t)j * n] -= L[i + (size_t)k * n] * l; }
      const double dj = sqrt(L[j + (size_t)j * n]);
      for (int i = j; i < n; i++) L[i + (size_t)j * n] /= dj;
    }
    double err = 0; for (int j = 0; j < n; j++) for (int i = j; i < n; i++) err = fmax(err, fabs(L[i + (size_t)j * n] - G[i + (size_t)j * n]));
    printf("n=%d band=%d: %.2f us (best of 6, one workgroup, events), max |L - L_host| = %.2e\n", n, band, best * 1e3, err);
  }
#ifdef CHOL_POLLS
  { unsigned long long pc[8];
    hipMemcpyFromSymbol(pc, HIP_SYMBOL(g_polls), sizeof pc);
    unsigned long long wp[12][8];
    hipMemcpyFromSymbol(wp, HIP_SYMBOL(g_wpolls), sizeof wp);
    printf("tile waves, poll iterations over the block (fL / cRaw / barrier / fP):");
    for (int w = 1; w < 12; w++) printf(" w%d %llu/%llu/%llu/%llu", w - 1, wp[w][0], wp[w][1], wp[w][2], wp[w][3]);
    printf("\n");
    printf("n=%d steps=%d: look-ahead tiles late in %llu steps (%llu polls of ~64 cycles + an LDS round trip each); waited for the tile waves' counter in %llu steps (%llu polls)\n", n, (n + 15) / 16, pc[0], pc[1], pc[2], pc[3]); }
#endif
#ifdef CHOL_STAMPS
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
  static unsigned long long tr[12][20][8];
  if (argc > 2) { // per-step timeline: start of each step relative to the factor wave's step 0, per wave
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
  if (argc > 2) { // inside the update walk: k==0 part | LDS slot | slots 8-10 | 4-7 | 0-3
    static unsigned long long wk[12][20][8];
    hipMemcpyFromSymbol(wk, HIP_SYMBOL(g_walk), sizeof wk);
    printf("update walk (cycles): col-1 part, LDS slot, slots 8-10, slots 4-7, slots 0-3   (waves 1, 4, 11)\n");
    for (int k = 0; k < T && k < 20; k++) {
      printf("%2d:", k);
      const int ws[3] = { 1, 4, 11 };
      for (int j = 0; j < 3; j++) { const int w = ws[j]; printf(" [%6lld %6lld %6lld %6lld %6lld]", (long long)(wk[w][k][0] - tr[w][k][5]), (long long)(wk[w][k][1] - wk[w][k][0]), (long long)(wk[w][k][3] - wk[w][k][4]), (long long)(wk[w][k][2] - wk[w][k][3]), (long long)(wk[w][k][5] - wk[w][k][2])); }
      printf("\n");
    }
  }
#endif
  return 0;
}
