// fp32 factorisation kernels of libcholamd for gfx950 (BASELINE config 5: fp32 factor + fp64 iterative refinement;
// SURVEY 8 f4 -- not in the reference, whose only arithmetic is fp64 CBLAS).  Same work descriptors and the same
// panel arena layout as the fp64 path (offsets in ELEMENTS), the arena holding floats.  The fp32 path serves the large
// generated problems, where the flops are in the macro-tile update and the panels are HBM-sized: half the bytes, and
// v_mfma_f32_32x32x2 / 16x16x4 issue twice the fp64 rate.  The kernels here are throughput kernels (one launch per
// phase of a column-block step, pivots in blocks of at most CHOL32_MAXN columns factored out of LDS); the latency
// machinery of the fp64 path (register-resident pivots, fused launches) is not duplicated.
//
// MFMA operand maps (cdna_hip_programming.md section 3): v_mfma_f32_16x16x4_f32 lane l supplies A[i = l & 15][k = l >> 4],
// B[k = l >> 4][j = l & 15], result register q of lane l is D[i = 4 (l >> 4) + q][j = l & 15] -- NOT the fp64 map.  As in
// the fp64 kernels every product is  acc(r, c) = sum_k X(r, k) Y(c, k)  with Y fed as "A" and X as "B", so lane
// (r = l & 15, g = l >> 4) ends up with acc(r, c = 4 g + q): four consecutive columns per lane.  An accumulator tile is
// still a valid X operand of the next MFMA if the k index of step s is taken as 4 g + s (the register the lane holds)
// and the other operand is indexed with the same k: a permutation of the summation order, nothing else.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "chol_plan.h"
#include "chol_kernels.h"

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));
#define TS 16

__global__ void k32_scatter(float *__restrict__ arena, const int64_t *__restrict__ dst, const double *__restrict__ val, int64_t nnz)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nnz; i += stride) arena[dst[i]] = (float)val[i];
}

// acc(r, c) += sum_{k < K} X[r + k ldx] * Y[c + k ldy]; rows outside [r0, mv) / [c0, nv) read as 0
__device__ __forceinline__ f4 rank_k_16x16_f32(f4 acc, const float *__restrict__ X, int ldx, int mv, const float *__restrict__ Y, int ldy, int nv, int K,
                                               int lane, int r0 = 0, int c0 = 0)
{
  const int r = lane & 15, kq = lane >> 4;
  const bool vx = r >= r0 && r < mv, vy = r >= c0 && r < nv;
  const float *px = X + r + (int64_t)kq * ldx;
  const float *py = Y + r + (int64_t)kq * ldy;
  int k0 = 0;
  for (; k0 + 32 <= K; k0 += 32) {
    float x[8], y[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { x[u] = vx ? px[(int64_t)(4 * u) * ldx] : 0.f; y[u] = vy ? py[(int64_t)(4 * u) * ldy] : 0.f; }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(y[u], x[u], acc, 0, 0, 0);
    px += 32 * (int64_t)ldx; py += 32 * (int64_t)ldy;
  }
  for (; k0 < K; k0 += 4) {
    const bool vk = k0 + kq < K;
    const float x = (vx && vk) ? px[0] : 0.f, y = (vy && vk) ? py[0] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, acc, 0, 0, 0);
    px += 4 * (int64_t)ldx; py += 4 * (int64_t)ldy;
  }
  return acc;
}

// ------------------------------------------------------------------------------------------------
// 16x16 lower Cholesky, one row per lane (row = lane & 15), lanes 16-31 carry the rows of the identity through the same
// column operations and come out as the rows of L^-T (the explicit inverse the panel solves multiply with) -- the fp64
// factor wave's scheme (chol_kernels.hip, chol16_rows) in single precision.  v_rsq_f32 + one Newton step.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float readlane_f32(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ int chol16_rows_f32(float (&a)[TS])
{
  int bad = 0;
#pragma unroll
  for (int j = 0; j < TS; ++j) {
    const float d = readlane_f32(a[j], j);
    float akj[TS];
#pragma unroll
    for (int k = j + 1; k < TS; ++k) akj[k] = readlane_f32(a[j], k);
    if (!(d > 0.f) && bad == 0) bad = j + 1;
    float y = __builtin_amdgcn_rsqf(d);
    y = y * fmaf(-0.5f * d * y, y, 1.5f);
    const float r = readlane_f32(y, 0);
    a[j] = a[j] * r;
    const float t = a[j] * r;
#pragma unroll
    for (int k = j + 1; k < TS; ++k) a[k] = fmaf(-t, akj[k], a[k]);
  }
  return bad;
}

// ------------------------------------------------------------------------------------------------
// POTRF of one pivot block of at most CHOL32_MAXN columns: the block's lower triangle lives in LDS (column-major,
// leading dimension 144 floats: the four k-groups of an MFMA operand read land in disjoint banks), right-looking in
// 16-column steps -- wave 0 factors the diagonal tile (rows per lane, inverse from the identity passengers), the four
// waves solve the panel tiles with the explicit inverse (4 MFMAs each) and update the trailing tiles.
// Linv of every diagonal tile goes to the workspace as W[tile * 256 + k * 16 + c] = Linv(c, k) for the TRSM kernel.
// ------------------------------------------------------------------------------------------------
#define P32_LD 144
__global__ __launch_bounds__(256) void k32_potrf(float *__restrict__ base, float *__restrict__ ws, const chol_potrf_desc *__restrict__ descs, int *__restrict__ info)
{
  extern __shared__ float smem32[];
  float *sA = smem32;                       // [CHOL32_MAXN][P32_LD]
  float *sW = smem32 + CHOL32_MAXN * P32_LD; // [16][16]: sW[k * 16 + c] = Linv(c, k) of the current diagonal tile
  const chol_potrf_desc d = descs[blockIdx.x];
  float *A = base + d.a_off;
  float *W = ws + d.dinv_off;
  const int n = d.n, lda = d.lda;
  const int T = (n + TS - 1) / TS, np = T * TS;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r15 = lane & 15, g = lane >> 4;
  for (int c = wave; c < np; c += 4)
    for (int r = lane; r < np; r += 64) {
      float v = (r == c) ? 1.f : 0.f; // identity padding past n
      if (r < n && c < n) v = (c <= r) ? A[r + (int64_t)c * lda] : 0.f;
      sA[c * P32_LD + r] = v;
    }
  __syncthreads();
  for (int k = 0; k < T; ++k) {
    const int j0 = k * TS;
    if (wave == 0) {
      float a[TS];
#pragma unroll
      for (int c = 0; c < TS; ++c) a[c] = (lane & 16) ? ((lane & 15) == c ? 1.f : 0.f) : sA[(j0 + c) * P32_LD + j0 + r15];
      const int bad = chol16_rows_f32(a);
      if (bad && j0 + bad <= n && lane == 0) {
        if (atomicCAS(&info[0], 0, d.col0 + j0 + bad) == 0) info[1] = d.sep;
      }
      if (lane < TS) {
#pragma unroll
        for (int c = 0; c < TS; ++c) sA[(j0 + c) * P32_LD + j0 + lane] = (c <= lane) ? a[c] : 0.f;
      } else if (lane < 2 * TS) { // lane 16 + m holds row m of L^-T: a[c] = Linv(c, m)
#pragma unroll
        for (int c = 0; c < TS; ++c) { sW[(lane - TS) * TS + c] = a[c]; W[(int64_t)k * TS * TS + (lane - TS) * TS + c] = a[c]; }
      }
    }
    __syncthreads();
    // panel: X(r, c) = sum_k T(r, k) Linv(c, k)
    for (int i = k + 1 + wave; i < T; i += 4) {
      f4 x = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
      for (int s = 0; s < 4; ++s)
        x = __builtin_amdgcn_mfma_f32_16x16x4f32(sW[(4 * s + g) * TS + r15], sA[(j0 + 4 * s + g) * P32_LD + i * TS + r15], x, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) sA[(j0 + 4 * g + q) * P32_LD + i * TS + r15] = x[q];
    }
    __syncthreads();
    // trailing tiles (i, j), k < j <= i: T(i,j) -= P_i P_j^T
    const int nt = T - k - 1;
    for (int t = wave; t < nt * (nt + 1) / 2; t += 4) {
      int tj = 0, rem = t;
      while (rem >= nt - tj) { rem -= nt - tj; ++tj; }
      const int i = k + 1 + tj + rem, j = k + 1 + tj;
      f4 acc = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
      for (int s = 0; s < 4; ++s)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sA[(j0 + 4 * s + g) * P32_LD + j * TS + r15], sA[(j0 + 4 * s + g) * P32_LD + i * TS + r15], acc, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) sA[(j * TS + 4 * g + q) * P32_LD + i * TS + r15] -= acc[q];
    }
    __syncthreads();
  }
  for (int c = wave; c < n; c += 4)
    for (int r = c + lane; r < n; r += 64) A[r + (int64_t)c * lda] = sA[c * P32_LD + r];
}

// ------------------------------------------------------------------------------------------------
// TRSM: B <- B L^-T for strips of at most 16 rows against a pivot block of at most CHOL32_MAXN columns; one wave per
// strip, four strips per workgroup.  The strip lives in LDS ([col][row]); left-looking over the column tiles:
// T_J = B_J - sum_{K<J} X_K L(J,K)^T (X_K out of LDS, L from global / L2), X_J = T_J Linv(J,J)^T with T_J used as the
// MFMA operand straight from its accumulator registers (k index 4 g + s on both operands).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k32_trsm(float *__restrict__ base, const float *__restrict__ ws, const chol_trsm_desc *__restrict__ descs, int ndesc)
{
  __shared__ float sXall[4][CHOL32_MAXN * TS];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int id = blockIdx.x * 4 + wave;
  if (id >= ndesc) return;
  const chol_trsm_desc d = descs[id];
  if (d.m <= 0) return;
  float *sX = sXall[wave];
  const float *Lm = base + d.l_off;
  const float *W = ws + d.dinv_off;
  float *B = base + d.b_off;
  const int n = d.n, m = d.m, ldl = d.ldl, ldb = d.ldb;
  const int T = (n + TS - 1) / TS;
  const int r15 = lane & 15, g = lane >> 4;
  for (int c = g; c < T * TS; c += 4) sX[c * TS + r15] = (r15 < m && c < n) ? B[r15 + (int64_t)c * ldb] : 0.f;
  for (int J = 0; J < T; ++J) {
    f4 acc = { 0.f, 0.f, 0.f, 0.f };
    const int lrow = min(J * TS + r15, n - 1); // rows past n are clamped: they only reach output columns >= n, never stored
    for (int K = 0; K < J; ++K) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Lm[lrow + (int64_t)(K * TS + 4 * s + g) * ldl], sX[(K * TS + 4 * s + g) * TS + r15], acc, 0, 0, 0);
    }
    f4 t;
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = sX[(J * TS + 4 * g + q) * TS + r15] - acc[q];
    f4 x = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
    for (int s = 0; s < 4; ++s) x = __builtin_amdgcn_mfma_f32_16x16x4f32(W[(int64_t)J * TS * TS + (4 * g + s) * TS + r15], t[s], x, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = J * TS + 4 * g + q;
      sX[col * TS + r15] = x[q];
      if (r15 < m && col < n) B[r15 + (int64_t)col * ldb] = x[q];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// UPDATE, 16x16 tasks (target-centric, sources in program order, the four waves split K or the sources; deterministic):
// the fp64 k_update in single precision.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k32_update(float *__restrict__ base, const chol_upd_task *__restrict__ tasks, const chol_upd_src *__restrict__ srcs, int ntask, int per_xcd)
{
  __shared__ float sAcc[3][4][64];
  const int tid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per_xcd || tid >= ntask) return;
  const chol_upd_task t = tasks[tid];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  f4 acc = { 0.f, 0.f, 0.f, 0.f };
  const int nsrc = t.src_end - t.src_begin;
  if (nsrc >= 4) {
    for (int s = t.src_begin + wave; s < t.src_end; s += 4) {
      const chol_upd_src sd = srcs[s];
      const int r0 = sd.range & 255, r1 = sd.range ? (sd.range >> 8) & 255 : t.mv, c0 = (sd.range >> 16) & 255, c1 = sd.range ? (sd.range >> 24) & 255 : t.nv;
      acc = rank_k_16x16_f32(acc, base + sd.a_off + t.ar, sd.lda, r1, base + sd.b_off + t.br, sd.ldb, c1, sd.k, lane, r0, c0);
    }
  } else {
    for (int s = t.src_begin; s < t.src_end; ++s) {
      const chol_upd_src sd = srcs[s];
      const int kc = ((((sd.k + 3) >> 2) + 3) >> 2) << 2;
      const int k_lo = wave * kc;
      if (k_lo < sd.k) {
        const int kn = min(kc, sd.k - k_lo);
        const int r0 = sd.range & 255, r1 = sd.range ? (sd.range >> 8) & 255 : t.mv, c0 = (sd.range >> 16) & 255, c1 = sd.range ? (sd.range >> 24) & 255 : t.nv;
        acc = rank_k_16x16_f32(acc, base + sd.a_off + t.ar + (int64_t)k_lo * sd.lda, sd.lda, r1, base + sd.b_off + t.br + (int64_t)k_lo * sd.ldb, sd.ldb, c1, kn, lane, r0, c0);
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) sAcc[wave - 1][q][lane] = acc[q];
  }
  __syncthreads();
  if (wave == 0) {
    float *C = base + t.c_off;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = 4 * g + q;
      const float v = ((acc[q] + sAcc[0][q][lane]) + sAcc[1][q][lane]) + sAcc[2][q][lane];
      if (r < t.mv && c < t.nv && (!t.lower || r >= c)) C[r + (int64_t)c * t.ldc] -= v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// UPDATE, macro tiles: one workgroup per 64x64 block of a target, the four waves own its 32x32 quadrants, one
// v_mfma_f32_32x32x2 accumulator (16 registers) each.  Source panels staged through LDS in 16-deep K chunks ([k][row]
// images: an operand read is 32 consecutive floats per k), double buffered, the next chunk's global loads in flight
// during the MFMAs.  32x32x2 maps: lane l supplies A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31]; result register i
// of lane l is D[(i & 3) + 8 (i >> 2) + 4 (l >> 5)][l & 31]; with Y as "A" and X as "B": row = l & 31, column = that.
// ------------------------------------------------------------------------------------------------
#define MT 64
#define MKB 16
__global__ __launch_bounds__(256) void k32_update_mt(float *__restrict__ base, const chol_upd_task *__restrict__ tasks, const chol_upd_src *__restrict__ srcs, int ntask, int per_xcd)
{
  __shared__ float sA[2][MKB][MT];
  __shared__ float sB[2][MKB][MT];
  const int tt = threadIdx.x, lane = tt & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tt >> 6);
  const int tid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per_xcd || tid >= ntask) return;
  const chol_upd_task t = tasks[tid];
  const int l31 = lane & 31, h = lane >> 5;
  const int wr = wave & 1, wc = wave >> 1;
  const int srow = tt & 63, skq = tt >> 6;
  const bool sva = srow < t.mv, svb = srow < t.nv;
  f16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  int buf = 0;
  for (int s = t.src_begin; s < t.src_end; ++s) {
    const chol_upd_src sd = srcs[s];
    const float *A = base + sd.a_off + t.ar + srow;
    const float *Bp = base + sd.b_off + t.br + srow;
    const int K = sd.k;
    float ra[4], rb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = skq + 4 * i;
      ra[i] = (sva && k < K) ? A[(int64_t)k * sd.lda] : 0.f;
      rb[i] = (svb && k < K) ? Bp[(int64_t)k * sd.ldb] : 0.f;
    }
    for (int k0 = 0; k0 < K; k0 += MKB) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { sA[buf][skq + 4 * i][srow] = ra[i]; sB[buf][skq + 4 * i][srow] = rb[i]; }
      if (k0 + MKB < K) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int k = k0 + MKB + skq + 4 * i;
          ra[i] = (sva && k < K) ? A[(int64_t)k * sd.lda] : 0.f;
          rb[i] = (svb && k < K) ? Bp[(int64_t)k * sd.ldb] : 0.f;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // LDS-only barrier: the prefetch stays in flight
#pragma unroll
      for (int kk = 0; kk < MKB / 2; ++kk)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sB[buf][2 * kk + h][32 * wc + l31], sA[buf][2 * kk + h][32 * wr + l31], acc, 0, 0, 0);
      buf ^= 1;
    }
  }
  const int r = 32 * wr + l31;
  float *C = base + t.c_off + r;
  float cv[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = min(32 * wc + (i & 3) + 8 * (i >> 2) + 4 * h, t.nv - 1);
    cv[i] = *(const volatile float *)(base + t.c_off + min(r, t.mv - 1) + (int64_t)c * t.ldc);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = 32 * wc + (i & 3) + 8 * (i >> 2) + 4 * h;
    if (r < t.mv && c < t.nv && (!t.lower || r >= c)) C[(int64_t)c * t.ldc] = cv[i] - acc[i];
  }
}

// ------------------------------------------------------------------------------------------------
// Iterative refinement helpers (fp64): r = b - A x with A as a symmetric CSR in original dof order, per-workgroup
// partial sums of r^2 and b^2 (added on the host: deterministic), x += dx.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_residual_csr(const int64_t *__restrict__ ptr, const int *__restrict__ col, const double *__restrict__ val,
                                                      const double *__restrict__ b, const double *__restrict__ x, double *__restrict__ r, int n, double *__restrict__ partial)
{
  __shared__ double s2[2][4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  double ri = 0.0, bi = 0.0;
  if (i < n) {
    bi = b[i];
    double acc = bi;
    for (int64_t e = ptr[i]; e < ptr[i + 1]; ++e) acc = fma(-val[e], x[col[e]], acc);
    ri = acc;
    if (r) r[i] = ri;
  }
  double a = ri * ri, c = bi * bi;
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); c += __shfl_down(c, o, 64); }
  if ((threadIdx.x & 63) == 0) { s2[0][threadIdx.x >> 6] = a; s2[1][threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = ((s2[0][0] + s2[0][1]) + s2[0][2]) + s2[0][3];
    partial[2 * blockIdx.x + 1] = ((s2[1][0] + s2[1][1]) + s2[1][2]) + s2[1][3];
  }
}
__global__ void k_axpy1(double *__restrict__ x, const double *__restrict__ dx, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] += dx[i];
}

extern "C" {
int chol32_launch_scatter(float *arena, const int64_t *dst, const double *val, int64_t nnz, hipStream_t st)
{
  if (nnz <= 0) return 0;
  int blocks = (int)((nnz + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k32_scatter, dim3(blocks), dim3(256), 0, st, arena, dst, val, nnz);
  return (int)hipGetLastError();
}
int chol32_launch_potrf(float *base, float *ws, const chol_potrf_desc *descs, int n, int *info, hipStream_t st)
{
  if (n <= 0) return 0;
  static bool attr = false;
  const size_t lds = (size_t)(CHOL32_MAXN * P32_LD + TS * TS) * sizeof(float);
  if (!attr) { (void)hipFuncSetAttribute((const void *)k32_potrf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
  hipLaunchKernelGGL(k32_potrf, dim3(n), dim3(256), lds, st, base, ws, descs, info);
  return (int)hipGetLastError();
}
int chol32_launch_trsm(float *base, const float *ws, const chol_trsm_desc *descs, int n, hipStream_t st)
{
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k32_trsm, dim3((n + 3) / 4), dim3(256), 0, st, base, ws, descs, n);
  return (int)hipGetLastError();
}
int chol32_launch_update(float *base, const chol_upd_task *tasks, const chol_upd_src *srcs, int ntask, hipStream_t st)
{
  if (ntask <= 0) return 0;
  const int per_xcd = (ntask + 7) / 8;
  hipLaunchKernelGGL(k32_update, dim3(per_xcd * 8), dim3(256), 0, st, base, tasks, srcs, ntask, per_xcd);
  return (int)hipGetLastError();
}
int chol32_launch_update_mt(float *base, const chol_upd_task *tasks, const chol_upd_src *srcs, int ntask, hipStream_t st)
{
  if (ntask <= 0) return 0;
  const int per_xcd = (ntask + 7) / 8;
  hipLaunchKernelGGL(k32_update_mt, dim3(per_xcd * 8), dim3(256), 0, st, base, tasks, srcs, ntask, per_xcd);
  return (int)hipGetLastError();
}
int chol_launch_residual(const int64_t *ptr, const int *col, const double *val, const double *b, const double *x, double *r, int n, double *partial, hipStream_t st)
{
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_residual_csr, dim3((n + 255) / 256), dim3(256), 0, st, ptr, col, val, b, x, r, n, partial);
  return (int)hipGetLastError();
}
int chol_launch_axpy1(double *x, const double *dx, int n, hipStream_t st)
{
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_axpy1, dim3((n + 255) / 256), dim3(256), 0, st, x, dx, n);
  return (int)hipGetLastError();
}
} // extern "C"
