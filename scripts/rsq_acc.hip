// accuracy of v_rsq_f64 / v_rcp_f64 seeds and of 1 / 2 Newton steps (relative error vs long double on host)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double *x, double *o, int n)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d = x[i];
  double y0 = __builtin_amdgcn_rsq(d);
  double h = 0.5 * d;
  double e = fma(-(h * y0), y0, 0.5);
  double y1 = fma(y0, e, y0);
  e = fma(-(h * y1), y1, 0.5);
  double y2 = fma(y1, e, y1);
  double z0 = __builtin_amdgcn_rcp(d);
  double f = fma(-d, z0, 1.0);
  double z1 = fma(z0, f, z0);
  o[5 * i] = y0; o[5 * i + 1] = y1; o[5 * i + 2] = y2; o[5 * i + 3] = z0; o[5 * i + 4] = z1;
}
int main()
{
  const int n = 1 << 20;
  std::vector<double> x(n), o(5 * n);
  srand(1);
  for (int i = 0; i < n; i++) x[i] = ldexp(1.0 + rand() / (double)RAND_MAX, (rand() % 40) - 20);
  double *dx, *dout;
  hipMalloc(&dx, n * 8); hipMalloc(&dout, 5 * n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  hipMemcpy(o.data(), dout, 5 * n * 8, hipMemcpyDeviceToHost);
  long double m[5] = { 0, 0, 0, 0, 0 };
  for (int i = 0; i < n; i++) {
    long double rs = 1.0L / sqrtl((long double)x[i]), rc = 1.0L / (long double)x[i];
    for (int j = 0; j < 3; j++) { long double e = fabsl((o[5 * i + j] - rs) / rs); if (e > m[j]) m[j] = e; }
    for (int j = 3; j < 5; j++) { long double e = fabsl((o[5 * i + j] - rc) / rc); if (e > m[j]) m[j] = e; }
  }
  printf("max rel err: rsq seed %.3Le, 1 NR %.3Le, 2 NR %.3Le | rcp seed %.3Le, 1 NR %.3Le  (eps = 1.1e-16)\n", m[0], m[1], m[2], m[3], m[4]);
  return 0;
}
