// HIP kernels of libcholamd for gfx950 (MI355X, CDNA4): batched POTRF / TRSM / SYRK+GEMM update of
// one tree level, the A scatter and the solve kernels.  fp64 throughout.
//
// Work descriptors (chol_plan.h) carry offsets in doubles relative to a base pointer, so one set
// of descriptors serves any arena; the BLAS-/task-level entry points pass base = nullptr and
// offsets = pointer / 8.
//
// MFMA use: v_mfma_f64_16x16x4_f64.  Operand maps (cdna_hip_programming.md section 3): lane l supplies
// A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; result register q of lane l is
// D[i = (l >> 4) + 4 q][j = l & 15].  All kernels here compute  acc(r, c) = sum_k X(r, k) * Y(c, k)
// for two column-major row panels X, Y by feeding Y as the MFMA "A" operand and X as the "B"
// operand, so that lane l ends up with acc(r = l & 15, c = (l >> 4) + 4 q): 16 consecutive lanes
// hold 16 consecutive rows of one column and the read-modify-write of the column-major target is
// made of 128-byte segments.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "chol_plan.h"
#include "chol_kernels.h"

typedef double d4 __attribute__((ext_vector_type(4)));

#define NB CHOL_NB

// ------------------------------------------------------------------------------------------------
// acc(r, c) += sum_{k < K} X[r + k ldx] * Y[c + k ldy],  r < mv, c < nv (rows beyond are read as 0)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ d4 rank_k_16x16(d4 acc, const double *__restrict__ X, int ldx, int mv,
                                           const double *__restrict__ Y, int ldy, int nv, int K, int lane)
{
  const int r = lane & 15, kq = lane >> 4;
  const bool vx = r < mv, vy = r < nv;
  const double *px = X + r + (int64_t)kq * ldx;
  const double *py = Y + r + (int64_t)kq * ldy;
  int k0 = 0;
  for (; k0 + 16 <= K; k0 += 16) { // four MFMAs per trip, eight loads in flight
    double x0 = vx ? px[0] : 0.0, y0 = vy ? py[0] : 0.0;
    double x1 = vx ? px[4 * (int64_t)ldx] : 0.0, y1 = vy ? py[4 * (int64_t)ldy] : 0.0;
    double x2 = vx ? px[8 * (int64_t)ldx] : 0.0, y2 = vy ? py[8 * (int64_t)ldy] : 0.0;
    double x3 = vx ? px[12 * (int64_t)ldx] : 0.0, y3 = vy ? py[12 * (int64_t)ldy] : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y0, x0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y1, x1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y2, x2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y3, x3, acc, 0, 0, 0);
    px += 16 * (int64_t)ldx; py += 16 * (int64_t)ldy;
  }
  for (; k0 < K; k0 += 4) {
    const bool vk = k0 + kq < K;
    double x = (vx && vk) ? px[0] : 0.0, y = (vy && vk) ? py[0] : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, acc, 0, 0, 0);
    px += 4 * (int64_t)ldx; py += 4 * (int64_t)ldy;
  }
  return acc;
}

// ------------------------------------------------------------------------------------------------
// A scatter (fill_block, mmat.rg:529-633): arena is zeroed by a memset node, then tril(A) lands
// ------------------------------------------------------------------------------------------------
__global__ void k_scatter(double *__restrict__ arena, const int64_t *__restrict__ dst, const double *__restrict__ val, int64_t nnz)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nnz; i += stride) arena[dst[i]] = val[i];
}

// ------------------------------------------------------------------------------------------------
// UPDATE: one wavefront per 16x16 output sub-tile of a target C tile; loops over the tile's sources
// in the reference's program order.  C <- C - sum_s A_s B_s^T   (cblas_dgemm NoTrans/Trans alpha=-1
// beta=1, blas.rg:139; cblas_dsyrk Lower alpha=-1 beta=1, blas.rg:187 for `lower` diagonal tiles)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_update(double *__restrict__ base, const chol_upd_task *__restrict__ tasks,
                                                const chol_upd_src *__restrict__ srcs, int ntask)
{
  const int lane = threadIdx.x & 63;
  const int tid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (tid >= ntask) return;
  const chol_upd_task t = tasks[tid];
  d4 acc = { 0.0, 0.0, 0.0, 0.0 };
  for (int s = t.src_begin; s < t.src_end; ++s) {
    const chol_upd_src sd = srcs[s];
    acc = rank_k_16x16(acc, base + sd.a_off + t.ar, sd.lda, t.mv, base + sd.b_off + t.br, sd.ldb, t.nv, sd.k, lane);
  }
  const int r = lane & 15;
  double *C = base + t.c_off + r;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = (lane >> 4) + 4 * q;
    if (r < t.mv && c < t.nv && (!t.lower || r >= c)) C[(int64_t)c * t.ldc] -= acc[q];
  }
}

// ------------------------------------------------------------------------------------------------
// 32x32 lower Cholesky by one wavefront: lane r (< 32; lanes 32..63 mirror them) holds row r in
// registers.  Column step j: every lane publishes its a(r, j) in LDS, reads the pivot d = a(j, j)
// back (broadcast), and applies the Schur update a(r, k) -= (a(r, j) / d) a(k, j) with a(k, j)
// broadcast from LDS -- one LDS round trip per column, no cross-lane register traffic.
// sR[j] receives 1 / L(j, j).  Returns the first failing column (1-based) or 0.
// ------------------------------------------------------------------------------------------------
// 1/sqrt(d) to fp64 accuracy: v_rsq_f64 seed + two Newton steps (6 dependent FMAs instead of the
// ~45-instruction sqrt + divide sequence on the critical path of every column)
__device__ __forceinline__ double rsqrt_nr(double d)
{
  double y = __builtin_amdgcn_rsq(d);
  const double h = 0.5 * d;
  double e = fma(-(h * y), y, 0.5);
  y = fma(y, e, y);
  e = fma(-(h * y), y, 0.5);
  y = fma(y, e, y);
  return y;
}

__device__ __forceinline__ int chol32_wave(double (&a)[NB], int lane, double *sCol, double *sR)
{
  int bad = 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    sCol[lane] = a[j];
    __builtin_amdgcn_wave_barrier();
    const double d = sCol[j];
    if (!(d > 0.0) && bad == 0) bad = j + 1;
    const double inv = rsqrt_nr(d);
    const double t = a[j] * (inv * inv);
#pragma unroll
    for (int k = j + 1; k < NB; ++k) a[k] = fma(-t, sCol[k], a[k]);
    a[j] = a[j] * inv; // lane j: d / sqrt(d) = L(j, j) to within an ulp
    sR[j] = inv;       // same value from every lane
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  return bad;
}

// x <- x * L^-T for one row x (registers) against the NB x NB lower factor in LDS (sD[row][col]);
// sR[j] = 1 / L[j][j].
__device__ __forceinline__ void row_solve32(double (&x)[NB], const double (*sD)[NB + 1], const double *sR)
{
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    x[j] = x[j] * sR[j];
#pragma unroll
    for (int k = j + 1; k < NB; ++k) x[k] = fma(-x[j], sD[k][j], x[k]);
  }
}

// ------------------------------------------------------------------------------------------------
// POTRF: one workgroup (256 threads) per separator; right-looking, NB = 32 columns per step.
//   1. wave 0 factors the diagonal block in registers, publishes it in LDS and global memory;
//   2. every thread solves rows of the panel below against it (substitution, L11 broadcast from
//      LDS); 32 more "virtual rows" = the identity give L11^-1, stored to the workspace for TRSM;
//   3. trailing update A22 -= P P^T with fp64 MFMA, 16x16 tiles of the lower triangle over the 4
//      waves.
// LAPACKE_dpotrf(ColMajor,'L',n,a,lda) semantics (blas.rg:71); info: first non-positive pivot.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_potrf(double *__restrict__ base, double *__restrict__ ws,
                                               const chol_potrf_desc *__restrict__ descs, int *__restrict__ info)
{
  __shared__ double sD[NB][NB + 1];
  __shared__ double sR[NB];
  __shared__ double sCol[NB];
  const chol_potrf_desc d = descs[blockIdx.x];
  double *A = base + d.a_off;
  double *W = ws + d.dinv_off;
  const int n = d.n, lda = d.lda;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  for (int j0 = 0; j0 < n; j0 += NB) {
    const int jb = min(NB, n - j0);
    // ---- 1. diagonal block: stage (identity-padded, upper part zeroed) in LDS, factor in wave 0
    for (int e = tid; e < NB * NB; e += 256) {
      const int r = e & (NB - 1), k = e >> 5;
      double v = (r == k) ? 1.0 : 0.0;
      if (r < jb && k <= r) v = A[(j0 + r) + (int64_t)(j0 + k) * lda];
      sD[r][k] = v;
    }
    __syncthreads();
    if (wave == 0) {
      double a[NB];
      const int r = lane & 31;
#pragma unroll
      for (int k = 0; k < NB; ++k) a[k] = sD[r][k];
      int bad = chol32_wave(a, r, sCol, sR);
#pragma unroll
      for (int k = 0; k < NB; ++k) sD[r][k] = a[k];
      if (bad && bad <= jb && lane == 0) {
        if (atomicCAS(&info[0], 0, j0 + bad) == 0) info[1] = d.sep;
      }
    }
    __syncthreads();
    for (int e = tid; e < NB * NB; e += 256) {
      const int r = e & (NB - 1), k = e >> 5;
      if (r < jb && k <= r) A[(j0 + r) + (int64_t)(j0 + k) * lda] = sD[r][k];
    }
    // ---- 2. panel rows below + identity rows (inverse)
    const int below = n - j0 - jb;
    for (int rr = tid; rr < below + NB; rr += 256) {
      double x[NB];
      if (rr < below) {
        const double *src = A + (j0 + jb + rr) + (int64_t)j0 * lda;
#pragma unroll
        for (int k = 0; k < NB; ++k) x[k] = (k < jb) ? src[(int64_t)k * lda] : 0.0;
      } else {
#pragma unroll
        for (int k = 0; k < NB; ++k) x[k] = (k == rr - below) ? 1.0 : 0.0;
      }
      {
        // keep the 528 LDS operands of the substitution from being hoisted out of the row loop
        // (they are loop invariant; hoisting them costs > 512 registers and spills)
        const double (*pD)[NB + 1] = sD; const double *pR = sR;
        asm volatile("" : "+v"(pD), "+v"(pR));
        row_solve32(x, pD, pR);
      }
      if (rr < below) {
        double *dst = A + (j0 + jb + rr) + (int64_t)j0 * lda;
#pragma unroll
        for (int k = 0; k < NB; ++k)
          if (k < jb) dst[(int64_t)k * lda] = x[k];
      } else {
        // row c of (I L^-T) = column c of L^-1: V(k, c) = Linv[k][c], col-major ld NB
        const int c = rr - below;
        double *dst = W + (int64_t)(j0 / NB) * NB * NB + (int64_t)c * NB;
#pragma unroll
        for (int k = 0; k < NB; ++k) dst[k] = x[k];
      }
    }
    __syncthreads();
    // ---- 3. trailing update (lower triangle of the (n - j0 - jb)^2 block)
    if (below > 0) {
      const int nt = (below + 15) / 16;
      const int ntiles = nt * (nt + 1) / 2;
      const double *P = A + (j0 + jb) + (int64_t)j0 * lda;
      double *T = A + (j0 + jb) + (int64_t)(j0 + jb) * lda;
      for (int t = wave; t < ntiles; t += 4) {
        // t -> (tr, tc) with tc <= tr
        int tr = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
        while (tr * (tr + 1) / 2 > t) --tr;
        while ((tr + 1) * (tr + 2) / 2 <= t) ++tr;
        const int tc = t - tr * (tr + 1) / 2;
        const int mv = min(16, below - tr * 16), nv = min(16, below - tc * 16);
        d4 acc = { 0.0, 0.0, 0.0, 0.0 };
        acc = rank_k_16x16(acc, P + tr * 16, lda, mv, P + tc * 16, lda, nv, jb, lane);
        const int r = lane & 15;
        double *C = T + (tr * 16 + r) + (int64_t)(tc * 16) * lda;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = (lane >> 4) + 4 * q;
          if (r < mv && c < nv && (tr != tc || r >= c)) C[(int64_t)c * lda] -= acc[q];
        }
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Inverses of the NB x NB diagonal blocks of an already factored L (for the BLAS-/task-level TRSM
// entry points, where L was not produced by k_potrf in the same call chain).  One wave per block.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_dinv(const double *__restrict__ Lp, int n, int ldl, double *__restrict__ W)
{
  __shared__ double sD[NB][NB + 1];
  __shared__ double sR[NB];
  const int j0 = blockIdx.x * NB, jb = min(NB, n - j0), lane = threadIdx.x;
  if (lane < NB) {
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      double v = (lane == k) ? 1.0 : 0.0;
      if (lane < jb && k <= lane) v = Lp[(j0 + lane) + (int64_t)(j0 + k) * ldl];
      sD[lane][k] = v;
      if (k == lane) sR[lane] = 1.0 / v;
    }
  }
  __syncthreads();
  if (lane < NB) {
    double x[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) x[k] = (k == lane) ? 1.0 : 0.0;
    row_solve32(x, sD, sR);
    double *dst = W + (int64_t)blockIdx.x * NB * NB + (int64_t)lane * NB;
#pragma unroll
    for (int k = 0; k < NB; ++k) dst[k] = x[k];
  }
}

// ------------------------------------------------------------------------------------------------
// TRSM: B <- B L^-T for a chunk of <= 32 rows (cblas_dtrsm Right/Lower/Trans/NonUnit alpha=1,
// blas.rg:99).  Blocked over 32-column blocks J:  X_J = (B_J - X_<J L[J,<J]^T) Linv_J^T, both
// products on fp64 MFMA; the 4 waves own the 2x2 grid of 16x16 tiles of the current block.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_trsm(double *__restrict__ base, const double *__restrict__ ws,
                                              const chol_trsm_desc *__restrict__ descs)
{
  __shared__ double sT[NB][CHOL_TRSM_ROWS + 1]; // sT[k][r]: T(r, k), column-major like the operands
  const chol_trsm_desc d = descs[blockIdx.x];
  const double *Lm = base + d.l_off;
  const double *W = ws + d.dinv_off;
  double *B = base + d.b_off;
  const int n = d.n, m = d.m, ldl = d.ldl, ldb = d.ldb;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tr = wave & 1, tc = wave >> 1; // this wave's 16x16 tile inside the 32x32 block
  const int mv = max(0, min(16, m - tr * 16));
  const int r = lane & 15;

  for (int j0 = 0; j0 < n; j0 += NB) {
    const int jb = min(NB, n - j0);
    const int nv = max(0, min(16, jb - tc * 16));
    // T = B_J - X_<J L[J,<J]^T
    d4 acc = { 0.0, 0.0, 0.0, 0.0 };
    if (mv > 0 && nv > 0)
      acc = rank_k_16x16(acc, B + tr * 16, ldb, mv, Lm + j0 + tc * 16, ldl, nv, j0, lane);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = (lane >> 4) + 4 * q;
      double v = 0.0;
      if (r < mv && c < nv) v = B[(tr * 16 + r) + (int64_t)(j0 + tc * 16 + c) * ldb] - acc[q];
      sT[tc * 16 + c][tr * 16 + r] = v;
    }
    __syncthreads();
    // X(r, c) = sum_k T(r, k) Linv(c, k), k < 32: operands from LDS (T) and workspace (Linv)
    {
      const double *V = W + (int64_t)(j0 / NB) * NB * NB; // V(c, k) at c + NB k
      d4 x = { 0.0, 0.0, 0.0, 0.0 };
      const int kq = lane >> 4;
#pragma unroll
      for (int k0 = 0; k0 < NB; k0 += 4) {
        const double tv = sT[k0 + kq][tr * 16 + r];
        const double lv = V[(tc * 16 + r) + (int64_t)(k0 + kq) * NB];
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(lv, tv, x, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = (lane >> 4) + 4 * q;
        if (r < mv && c < nv) B[(tr * 16 + r) + (int64_t)(j0 + tc * 16 + c) * ldb] = x[q];
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Solve phase (mmat.rg:1364-1495).  Vectors live in permuted order.
// ------------------------------------------------------------------------------------------------
__global__ void k_permute_in(const double *__restrict__ b, const int *__restrict__ perm, double *__restrict__ y, int n)
{ // fill_b, mmat.rg:769-783: y[pos] = b[perm[pos]]
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = b[perm[i]];
}
__global__ void k_permute_out(const double *__restrict__ y, const int *__restrict__ perm, double *__restrict__ x, int n)
{ // mmat.rg:1483-1491: x[perm[pos]] = y[pos]
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[perm[i]] = y[i];
}

// forward level step: for each separator s of the level (one workgroup each):
//   y_s <- L_ss^-1 y_s (cblas_dtrsv Lower/NoTrans, blas.rg:226), blocked by 32 with the panel GEMV
__global__ __launch_bounds__(256) void k_trsv_fwd(const double *__restrict__ base, const chol_trsv_desc *__restrict__ descs, double *__restrict__ y)
{
  __shared__ double sx[NB];
  const chol_trsv_desc d = descs[blockIdx.x];
  const double *Lm = base + d.a_off;
  double *x = y + d.x_off;
  const int n = d.n, lda = d.lda, tid = threadIdx.x;
  for (int j0 = 0; j0 < n; j0 += NB) {
    const int jb = min(NB, n - j0);
    if (tid < 64) { // one wave solves the 32x32 triangle, lane = row
      const int r = tid & 31;
      double v = (tid < jb) ? x[j0 + r] : 0.0;
      for (int j = 0; j < jb; ++j) {
        const double dj = Lm[(j0 + j) + (int64_t)(j0 + j) * lda];
        const double xj = __shfl(v, j, 64) / dj;
        if (r == j) v = xj;
        else if (r > j && r < jb) v -= xj * Lm[(j0 + r) + (int64_t)(j0 + j) * lda];
      }
      if (tid < jb) { x[j0 + r] = v; sx[r] = v; }
    }
    __syncthreads();
    for (int rr = j0 + jb + tid; rr < n; rr += 256) {
      double acc = 0.0;
      for (int k = 0; k < jb; ++k) acc += Lm[rr + (int64_t)(j0 + k) * lda] * sx[k];
      x[rr] -= acc;
    }
    __syncthreads();
  }
}

// y_t <- y_t - sum over sources A x   (cblas_dgemv NoTrans alpha=-1 beta=1, blas.rg:263), target-
// centric: one workgroup per 256 rows of a target separator, sources in program order.
__global__ __launch_bounds__(256) void k_gemv_fwd(const double *__restrict__ base, const chol_gemv_desc *__restrict__ descs,
                                                  const int *__restrict__ grp_start, const int *__restrict__ grp_rows, double *__restrict__ y)
{
  const int g = blockIdx.x;
  const int row0 = grp_rows[2 * g], y_off = grp_rows[2 * g + 1];
  const int r = row0 + threadIdx.x;
  double acc = 0.0;
  bool valid = false;
  for (int s = grp_start[g]; s < grp_start[g + 1]; ++s) {
    const chol_gemv_desc d = descs[s];
    if (r < d.m) {
      valid = true;
      const double *A = base + d.a_off + r;
      const double *x = y + d.x_off;
      for (int k = 0; k < d.n; ++k) acc += A[(int64_t)k * d.lda] * x[k];
    }
  }
  if (valid) y[y_off + r] -= acc;
}

// backward level step for separator s: y_s <- y_s - sum_anc A(anc,s)^T y_anc  (cblas_dgemv Trans),
// then y_s <- L_ss^-T y_s (cblas_dtrsv Lower/Trans).  One workgroup per separator.
__global__ __launch_bounds__(256) void k_bwd(const double *__restrict__ base, const chol_trsv_desc *__restrict__ descs,
                                             const chol_gemv_desc *__restrict__ gd, const int *__restrict__ gstart, double *__restrict__ y)
{
  __shared__ double sx[NB];
  __shared__ double red[256];
  const chol_trsv_desc d = descs[blockIdx.x];
  const double *Lm = base + d.a_off;
  double *x = y + d.x_off;
  const int n = d.n, lda = d.lda, tid = threadIdx.x;
  // gather from ancestors: column c of A^T x = dot(A[:, c], x_anc)
  for (int s = gstart[blockIdx.x]; s < gstart[blockIdx.x + 1]; ++s) {
    const chol_gemv_desc g = gd[s];
    const double *A = base + g.a_off;
    const double *xa = y + g.x_off;
    for (int c0 = 0; c0 < g.n; c0 += 4) { // 4 columns at a time, 64 threads per column
      const int c = c0 + (tid >> 6), l = tid & 63;
      double acc = 0.0;
      if (c < g.n)
        for (int i = l; i < g.m; i += 64) acc += A[i + (int64_t)c * g.lda] * xa[i];
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
      if (l == 0 && c < g.n) x[c] -= acc;
    }
    __syncthreads();
  }
  // L^T x = b, blocks from the bottom up
  const int nblk = (n + NB - 1) / NB;
  for (int jb_i = nblk - 1; jb_i >= 0; --jb_i) {
    const int j0 = jb_i * NB, jb = min(NB, n - j0);
    // x_J -= L[J+1.., J]^T x_{J+1..}: column c of the panel dotted with the solved tail
    {
      const int c = tid >> 3, l = tid & 7; // 32 columns x 8 threads
      double acc = 0.0;
      if (c < jb)
        for (int i = j0 + jb + l; i < n; i += 8) acc += Lm[i + (int64_t)(j0 + c) * lda] * x[i];
      red[tid] = acc;
      __syncthreads();
      if (tid < NB) {
        double s = 0.0;
        for (int q = 0; q < 8; ++q) s += red[tid * 8 + q];
        sx[tid] = (tid < jb) ? x[j0 + tid] - s : 0.0;
      }
      __syncthreads();
    }
    if (tid < 64) { // triangle solve, lane = column index, backwards
      const int r = tid & 31;
      double v = (r < jb) ? sx[r] : 0.0;
      for (int j = jb - 1; j >= 0; --j) {
        const double dj = Lm[(j0 + j) + (int64_t)(j0 + j) * lda];
        const double xj = __shfl(v, j, 64) / dj;
        if (r == j) v = xj;
        else if (r < j) v -= xj * Lm[(j0 + j) + (int64_t)(j0 + r) * lda];
      }
      if (tid < jb) x[j0 + r] = v;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// launchers (extern "C", called from chol_api.cpp)
// ------------------------------------------------------------------------------------------------
extern "C" {

int chol_launch_scatter(double *arena, const int64_t *dst, const double *val, int64_t nnz, hipStream_t st)
{
  if (nnz <= 0) return 0;
  int blocks = (int)((nnz + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_scatter, dim3(blocks), dim3(256), 0, st, arena, dst, val, nnz);
  return (int)hipGetLastError();
}
int chol_launch_potrf(double *base, double *ws, const chol_potrf_desc *descs, int n, int *info, hipStream_t st)
{
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_potrf, dim3(n), dim3(256), 0, st, base, ws, descs, info);
  return (int)hipGetLastError();
}
int chol_launch_dinv(const double *L, int n, int ldl, double *W, hipStream_t st)
{
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_dinv, dim3((n + NB - 1) / NB), dim3(64), 0, st, L, n, ldl, W);
  return (int)hipGetLastError();
}
int chol_launch_trsm(double *base, const double *ws, const chol_trsm_desc *descs, int n, hipStream_t st)
{
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_trsm, dim3(n), dim3(256), 0, st, base, ws, descs);
  return (int)hipGetLastError();
}
int chol_launch_update(double *base, const chol_upd_task *tasks, const chol_upd_src *srcs, int ntask, hipStream_t st)
{
  if (ntask <= 0) return 0;
  hipLaunchKernelGGL(k_update, dim3((ntask + 3) / 4), dim3(256), 0, st, base, tasks, srcs, ntask);
  return (int)hipGetLastError();
}
int chol_launch_permute(const double *in, const int *perm, double *out, int n, int inverse, hipStream_t st)
{
  if (n <= 0) return 0;
  if (inverse) hipLaunchKernelGGL(k_permute_out, dim3((n + 255) / 256), dim3(256), 0, st, in, perm, out, n);
  else hipLaunchKernelGGL(k_permute_in, dim3((n + 255) / 256), dim3(256), 0, st, in, perm, out, n);
  return (int)hipGetLastError();
}
int chol_launch_trsv_fwd(const double *base, const chol_trsv_desc *descs, int n, double *y, hipStream_t st)
{
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_trsv_fwd, dim3(n), dim3(256), 0, st, base, descs, y);
  return (int)hipGetLastError();
}
int chol_launch_gemv_fwd(const double *base, const chol_gemv_desc *descs, const int *grp_start, const int *grp_rows, int ngroups, double *y, hipStream_t st)
{
  if (ngroups <= 0) return 0;
  hipLaunchKernelGGL(k_gemv_fwd, dim3(ngroups), dim3(256), 0, st, base, descs, grp_start, grp_rows, y);
  return (int)hipGetLastError();
}
int chol_launch_bwd(const double *base, const chol_trsv_desc *descs, const chol_gemv_desc *gd, const int *gstart, int n, double *y, hipStream_t st)
{
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_bwd, dim3(n), dim3(256), 0, st, base, descs, gd, gstart, y);
  return (int)hipGetLastError();
}

} // extern "C"
