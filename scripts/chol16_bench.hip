// microbenchmark (one wave): the row-per-lane 16x16 Cholesky of the POTRF factor wave, as it is in
// chol_kernels.hip (column multipliers broadcast through SGPRs with v_readlane) against a DPP form
// (lead 4a): v_fmac_f64_dpp row_newbcast:k takes the multiplier from lane k of the lane's own
// 16-lane row, the identity-passenger rows (lanes 16-31, 48-63) get the tile's column through one
// v_permlane16_swap mirror per column step.  Prints cycles per 16x16 factorisation and the largest difference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define TS 16
__device__ __forceinline__ double readlane_f64(double v, int l)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rsqrt_nr(double d)
{
  const double y = __builtin_amdgcn_rsq(d);
  const double e = fma(-(d * y), y, 1.0);
  const double q = e * fma(0.375, e, 0.5);
  return fma(y, q, y);
}
// ---- variant A: as in the library
__device__ __forceinline__ void chol16_readlane(double (&a)[TS])
{
#pragma unroll
  for (int j = 0; j < TS; ++j) {
    const double d = readlane_f64(a[j], j);
    double akj[TS];
#pragma unroll
    for (int k = j + 1; k < TS; ++k) akj[k] = readlane_f64(a[j], k);
    const double rv = rsqrt_nr(d);
    const double r = readlane_f64(rv, 0);
    a[j] = a[j] * r;
    const double t = a[j] * r;
#pragma unroll
    for (int k = j + 1; k < TS; ++k) a[k] = fma(-t, akj[k], a[k]);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// ---- variant B: DPP multipliers; mirror of the tile's column into the passenger rows
template <int K, bool FIRST> __device__ __forceinline__ void fmac_bcast(double &acc, double m, double nt)
{ // acc += m[lane K of this lane's 16-lane row] * nt.  hipcc pads no hazards inside an asm statement: a VGPR written by the
  // VALU needs two wait states before a DPP instruction reads it as its shuffled operand -- the first use of m in a column
  // step carries them (m may have been produced by a compiler-inserted copy just ahead of the statement)
  if (FIRST) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(m), "v"(nt), "n"(K));
  else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(m), "v"(nt), "n"(K));
}
__device__ __forceinline__ double mirror_rows(double v)
{ // rows 1 and 3 (lanes 16-31, 48-63) <- rows 0 and 2
  int xl = __double2loint(v), xh = __double2hiint(v), yl = xl, yh = xh;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3" : "+v"(xl), "+v"(yl), "+v"(xh), "+v"(yh));
  return __hiloint2double(xh, xl);
}
template <int J> __device__ __forceinline__ void chol16_dpp_col(double (&a)[TS])
{
  const double d = readlane_f64(a[J], J);
  const double m = mirror_rows(a[J]); // unscaled column J of the tile in every row; off the rsqrt chain
  const double rv = rsqrt_nr(d);
  const double r = readlane_f64(rv, 0);
  a[J] = a[J] * r;
  const double nt = -(a[J] * r);
#define FM(K) if constexpr (K > J && K < TS) fmac_bcast<K, K == J + 1>(a[K], m, nt);
  FM(1) FM(2) FM(3) FM(4) FM(5) FM(6) FM(7) FM(8) FM(9) FM(10) FM(11) FM(12) FM(13) FM(14) FM(15)
#undef FM
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void chol16_dpp(double (&a)[TS])
{
  chol16_dpp_col<0>(a); chol16_dpp_col<1>(a); chol16_dpp_col<2>(a); chol16_dpp_col<3>(a);
  chol16_dpp_col<4>(a); chol16_dpp_col<5>(a); chol16_dpp_col<6>(a); chol16_dpp_col<7>(a);
  chol16_dpp_col<8>(a); chol16_dpp_col<9>(a); chol16_dpp_col<10>(a); chol16_dpp_col<11>(a);
  chol16_dpp_col<12>(a); chol16_dpp_col<13>(a); chol16_dpp_col<14>(a); chol16_dpp_col<15>(a);
}
// ---- variant C: DPP for the tile rows only, no passenger (what the inverse would then cost is not counted)
template <int J> __device__ __forceinline__ void chol16_dpp_nop_col(double (&a)[TS])
{
  const double d = readlane_f64(a[J], J);
  const double m = a[J];
  const double rv = rsqrt_nr(d);
  const double r = readlane_f64(rv, 0);
  a[J] = a[J] * r;
  const double nt = -(a[J] * r);
#define FM(K) if constexpr (K > J && K < TS) fmac_bcast<K, K == J + 1>(a[K], m, nt);
  FM(1) FM(2) FM(3) FM(4) FM(5) FM(6) FM(7) FM(8) FM(9) FM(10) FM(11) FM(12) FM(13) FM(14) FM(15)
#undef FM
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void chol16_dpp_nop(double (&a)[TS])
{
  chol16_dpp_nop_col<0>(a); chol16_dpp_nop_col<1>(a); chol16_dpp_nop_col<2>(a); chol16_dpp_nop_col<3>(a);
  chol16_dpp_nop_col<4>(a); chol16_dpp_nop_col<5>(a); chol16_dpp_nop_col<6>(a); chol16_dpp_nop_col<7>(a);
  chol16_dpp_nop_col<8>(a); chol16_dpp_nop_col<9>(a); chol16_dpp_nop_col<10>(a); chol16_dpp_nop_col<11>(a);
  chol16_dpp_nop_col<12>(a); chol16_dpp_nop_col<13>(a); chol16_dpp_nop_col<14>(a); chol16_dpp_nop_col<15>(a);
}

// ---- variant D: DPP multipliers; the tile's unscaled column goes to every 16-lane row through ds_bpermute_b32 (lane l reads lane l & 15),
//      issued ahead of the rsqrt chain that hides its round trip; the pivot's reciprocal root stays in a vector register
__device__ __forceinline__ double bperm_row0(double v, int addr)
{
  const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
template <int J> __device__ __forceinline__ void chol16_bp_col(double (&a)[TS], int addr)
{
  const double d = readlane_f64(a[J], J);
  const double m = bperm_row0(a[J], addr); // unscaled column J of the tile in every row; off the rsqrt chain
  const double rv = rsqrt_nr(d);
  a[J] = a[J] * rv;
  const double nt = -(a[J] * rv);
#define FM(K) if constexpr (K > J && K < TS) fmac_bcast<K, K == J + 1>(a[K], m, nt);
  FM(1) FM(2) FM(3) FM(4) FM(5) FM(6) FM(7) FM(8) FM(9) FM(10) FM(11) FM(12) FM(13) FM(14) FM(15)
#undef FM
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void chol16_bp(double (&a)[TS], int lane)
{
  const int addr = (lane & 15) << 2;
  chol16_bp_col<0>(a, addr); chol16_bp_col<1>(a, addr); chol16_bp_col<2>(a, addr); chol16_bp_col<3>(a, addr);
  chol16_bp_col<4>(a, addr); chol16_bp_col<5>(a, addr); chol16_bp_col<6>(a, addr); chol16_bp_col<7>(a, addr);
  chol16_bp_col<8>(a, addr); chol16_bp_col<9>(a, addr); chol16_bp_col<10>(a, addr); chol16_bp_col<11>(a, addr);
  chol16_bp_col<12>(a, addr); chol16_bp_col<13>(a, addr); chol16_bp_col<14>(a, addr); chol16_bp_col<15>(a, addr);
}
// ---- variant E: as A (readlane multipliers) with the pivot's reciprocal root kept in a vector register (no readlane of r)
__device__ __forceinline__ void chol16_readlane_v(double (&a)[TS])
{
#pragma unroll
  for (int j = 0; j < TS; ++j) {
    const double d = readlane_f64(a[j], j);
    double akj[TS];
#pragma unroll
    for (int k = j + 1; k < TS; ++k) akj[k] = readlane_f64(a[j], k);
    const double rv = rsqrt_nr(d);
    a[j] = a[j] * rv;
    const double t = a[j] * rv;
#pragma unroll
    for (int k = j + 1; k < TS; ++k) a[k] = fma(-t, akj[k], a[k]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int V> __global__ void k(const double *A, double *out, unsigned long long *t, int iters)
{
  __shared__ double s[2 * TS][TS + 1];
  const int lane = threadIdx.x;
  for (int e = lane; e < TS * TS; e += 64) { s[e / TS][e % TS] = A[e]; s[TS + e / TS][e % TS] = (e / TS == e % TS) ? 1.0 : 0.0; }
  __syncthreads();
  double a[TS];
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < TS; ++c) a[c] = s[lane & 31][c];
    if (V == 0) chol16_readlane(a); else if (V == 1) chol16_dpp(a); else if (V == 2) chol16_dpp_nop(a); else if (V == 3) chol16_bp(a, lane); else chol16_readlane_v(a);
    if (lane < 2 * TS) {
#pragma unroll
      for (int c = 0; c < TS; ++c) s[lane][c] = (it + 1 < iters) ? s[lane][c] : a[c]; // keeps the loop honest; the last pass stores the result
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (it + 1 < iters) { // restore the input
      for (int e = lane; e < TS * TS; e += 64) { s[e / TS][e % TS] = A[e]; s[TS + e / TS][e % TS] = (e / TS == e % TS) ? 1.0 : 0.0; }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  if (lane < 2 * TS)
    for (int c = 0; c < TS; ++c) out[lane * TS + c] = s[lane][c];
  if (lane == 0) t[0] = t1 - t0;
}
// the loop overhead alone (loads / restores, no factorisation)
__global__ void k_empty(const double *A, double *out, unsigned long long *t, int iters)
{
  __shared__ double s[2 * TS][TS + 1];
  const int lane = threadIdx.x;
  for (int e = lane; e < TS * TS; e += 64) { s[e / TS][e % TS] = A[e]; s[TS + e / TS][e % TS] = (e / TS == e % TS) ? 1.0 : 0.0; }
  __syncthreads();
  double a[TS];
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < TS; ++c) a[c] = s[lane & 31][c];
    if (lane < 2 * TS) {
#pragma unroll
      for (int c = 0; c < TS; ++c) s[lane][c] = (it + 1 < iters) ? s[lane][c] : a[c];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (it + 1 < iters) {
      for (int e = lane; e < TS * TS; e += 64) { s[e / TS][e % TS] = A[e]; s[TS + e / TS][e % TS] = (e / TS == e % TS) ? 1.0 : 0.0; }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  if (lane < 2 * TS)
    for (int c = 0; c < TS; ++c) out[lane * TS + c] = s[lane][c];
  if (lane == 0) t[0] = t1 - t0;
}
int main()
{
  double hA[TS * TS], hO[5][2 * TS * TS];
  for (int i = 0; i < TS; i++)
    for (int j = 0; j < TS; j++) hA[i * TS + j] = (i == j ? 6.0 + 0.1 * i : -1.0 / (1.0 + abs(i - j))); // row-major [row][col], SPD
  double *A, *o; unsigned long long *t, h;
  hipMalloc(&A, sizeof hA); hipMalloc(&o, sizeof hO[0]); hipMalloc(&t, 8);
  hipMemcpy(A, hA, sizeof hA, hipMemcpyHostToDevice);
  const int iters = 500;
  double cyc[6];
  for (int v = 0; v < 6; v++) {
    for (int r = 0; r < 2; r++) {
      if (v == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, A, o, t, iters);
      else if (v == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, A, o, t, iters);
      else if (v == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, A, o, t, iters);
      else if (v == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, A, o, t, iters);
      else if (v == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, A, o, t, iters);
      else hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, A, o, t, iters);
      hipDeviceSynchronize();
    }
    hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    if (v < 5) hipMemcpy(hO[v], o, sizeof hO[0], hipMemcpyDeviceToHost);
    cyc[v] = (double)h / iters;
  }
  { double dLd = 0, dId = 0, dLe = 0, dIe = 0;
    for (int r = 0; r < TS; r++) for (int c = 0; c < TS; c++) {
      if (c <= r) { dLd = fmax(dLd, fabs(hO[0][r * TS + c] - hO[3][r * TS + c])); dLe = fmax(dLe, fabs(hO[0][r * TS + c] - hO[4][r * TS + c])); }
      if (c >= r) { dId = fmax(dId, fabs(hO[0][(TS + r) * TS + c] - hO[3][(TS + r) * TS + c])); dIe = fmax(dIe, fabs(hO[0][(TS + r) * TS + c] - hO[4][(TS + r) * TS + c])); }
    }
    printf("ticks per tile (loop overhead %.0f subtracted): dpp + bpermute mirror %.0f (max |dL| %.1e, |dLinv| %.1e), readlane with vector r %.0f (max |dL| %.1e, |dLinv| %.1e)\n",
           cyc[5], cyc[3] - cyc[5], dLd, dId, cyc[4] - cyc[5], dLe, dIe);
    cyc[3] = cyc[5]; }
  double dL = 0, dI = 0, dLc = 0;
  for (int r = 0; r < TS; r++)
    for (int c = 0; c < TS; c++) {
      if (c <= r) { dL = fmax(dL, fabs(hO[0][r * TS + c] - hO[1][r * TS + c])); dLc = fmax(dLc, fabs(hO[0][r * TS + c] - hO[2][r * TS + c])); }
      if (c >= r) dI = fmax(dI, fabs(hO[0][(TS + r) * TS + c] - hO[1][(TS + r) * TS + c]));
    }
  printf("chol16 cycles per tile (loop overhead %.0f subtracted): readlane %.0f, dpp+mirror %.0f, dpp without passenger %.0f; max |dL| %.1e (dpp) %.1e (no passenger), max |dLinv| %.1e\n",
         cyc[3], cyc[0] - cyc[3], cyc[1] - cyc[3], cyc[2] - cyc[3], dL, dLc, dI);
  return 0;
}
