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
// UPDATE, macro tiles: one workgroup of four waves per 64x64 block of a target, every wave a 32x32 quadrant = 2 x 2 accumulators of
// v_mfma_f32_16x16x4_f32.  Source panels go through LDS as [k][row] images in 16-deep chunks: by LDS-DMA (global_load_lds_dwordx4,
// 1 KiB per wave instruction = four k-rows of 64 floats, no registers) into a ring of M32_STAGES stages behind a counted vmcnt, the
// chunk sequence running across the sources of the task -- the structure of the fp64 k_update_mt; 8 KB of LDS and ~60 registers let
// eight workgroups share a CU.  An edge tile takes the DMA path too when the rows it does not own lie inside the arena (their
// products are neither computed -- 16-blocks outside the tile are skipped -- nor stored); otherwise, and for the K tails (K mod 16),
// masked loads through registers.  Same task / source lists as the fp64 kernel, same program-order accumulation.
// (Round 2's first version staged 4-byte loads through registers into one 32x32x2 accumulator per wave: 67 TF/s on the 100^3 task
// lists.  A 128x128 tile -- 87 TF/s on one big SYRK -- lost on them: a third of the flops sit in partial tiles there.)
// ------------------------------------------------------------------------------------------------
#define M32 64
#define M32_KB 16
#ifndef M32_STAGES
#define M32_STAGES 2
#endif
#define M32_SRC_BATCH 32
/* LDS image of one 16-deep chunk of an operand: four PIECES (one per wave's DMA instruction: k-rows 4 w .. 4 w + 3, 256 floats) M32_PS
 * floats apart.  k-step kk of the MFMA loop gives lane group g the k-row kk of piece g (k = 4 g + kk): the four groups then read at
 * g * 272 + ..., i.e. from four different quarters of the banks.  With k = 4 kk + g out of an unpadded [k][64] image -- rounds 1 and 2 --
 * the four groups' addresses were 64 floats apart: the same bank, a four-way conflict on every ds_read_b32, and the LDS pipe as busy as
 * the matrix pipe (the loop without transfers and barriers stopped at 93 of 157 TF/s). */
#define M32_PS (4 * M32 + 16)
#define M32_IMG (4 * M32_PS)
/* timing diagnostics of scripts/mt_bench32.hip (wrong results): -DM32_NODMA leaves the operand transfers out, -DM32_NOBAR the ring's barriers */
#ifdef M32_NODMA
#define M32_DIAG_DMA(x_) ((void)0)
#else
#define M32_DIAG_DMA(x_) x_
#endif
#ifdef M32_NOBAR
#define M32_DIAG_BAR(x_) ((void)0)
#else
#define M32_DIAG_BAR(x_) x_
#endif
#ifndef M32_WPE
#define M32_WPE 8 /* waves per SIMD asked of the compiler (60 registers): 81 -> 85 TF/s on one big SYRK against the default's 6 */
#endif
__device__ __forceinline__ void lds_dma16_f32(const float *g, float *lds)
{ // lane l's 16 bytes at g land at lds + 16 l bytes; M0 saved and restored inside the statement (see lds_dma16, chol_kernels.hip)
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)lds);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}
// one 16-deep chunk out of the LDS images into the wave's quadrant.
#ifndef M32_MFMA16
// v_mfma_f32_32x32x2_f32: the quadrant is ONE 32 x 32 accumulator (16 registers, as many as the 2 x 2 of 16 x 16), eight MFMAs of 64 cycles
// per chunk.  In k-step s lanes 0-31 take k = s (pieces 0, 1), lanes 32-63 k = 8 + s (pieces 2, 3); a lane reads row xo32 + (lane & 31) of
// both operands.  (The 16x16x4 form issues at ~59 % of the matrix pipe's rate however it is fed -- the loop without transfers and barriers
// stops at 93 TF/s, with or without bank conflicts -- as the fp64 16x16x4 does at 64 %; the 32x32x2 form reaches 155.7 of 157 TF/s in
// scripts/mfma_peak.hip.)  nq: the quadrant has rows and columns inside the tile (wave uniform), else nothing is computed.
__device__ __forceinline__ void m32_chunk(f16 &acc, const float *sa, const float *sb, int lane, int xo32, int yo32, bool nq)
{
  if (!nq) return;
  const int h = lane >> 5;
  const float *pa = sa + 2 * h * M32_PS + xo32, *pb = sb + 2 * h * M32_PS + yo32;
  float x = pa[0], y = pb[0];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    float nx = 0.f, ny = 0.f;
#ifndef M32_NOLDS /* (timing diagnostic of scripts/mt_bench32.hip: the operands of the first k-step for all of them) */
    if (s + 1 < 8) { const int o = ((s + 1) >> 2) * M32_PS + ((s + 1) & 3) * M32; nx = pa[o]; ny = pb[o]; }
#else
    nx = x; ny = y;
#endif
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, acc, 0, 0, 0);
    x = nx; y = ny;
  }
}
#else
// (-DM32_MFMA16: the round-2 form) 2 x 2 accumulators of v_mfma_f32_16x16x4_f32; the operands of k-step kk + 1 are requested before the
// MFMAs of k-step kk are issued.  ni / nj: 16-row / 16-column blocks of the quadrant inside the tile (wave uniform)
__device__ __forceinline__ void m32_chunk(f4 (&acc)[2][2], const float *sa, const float *sb, int g, int xo, int yo, int ni, int nj)
{
  float x[2], y[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { x[i] = sa[g * M32_PS + xo + 16 * i]; y[i] = sb[g * M32_PS + yo + 16 * i]; }
#pragma unroll
  for (int kk = 0; kk < M32_KB / 4; ++kk) {
    float nx[2] = { 0.f, 0.f }, ny[2] = { 0.f, 0.f };
    if (kk + 1 < M32_KB / 4) {
#pragma unroll
      for (int i = 0; i < 2; ++i) { nx[i] = sa[g * M32_PS + (kk + 1) * M32 + xo + 16 * i]; ny[i] = sb[g * M32_PS + (kk + 1) * M32 + yo + 16 * i]; }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (i < ni && j < nj) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(y[j], x[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) { x[i] = nx[i]; y[i] = ny[i]; }
  }
}
#endif
#ifdef M32_CLOCK /* timing diagnostic of scripts/mt_bench32.hip: shader cycles (s_memtime) and 100 MHz ticks (s_memrealtime) one workgroup in the middle of the grid spends on its tile */
__device__ unsigned long long g_m32_clock[2];
#endif
__global__ __launch_bounds__(256, M32_WPE) void k32_update_mt(float *__restrict__ base, const chol_upd_task *__restrict__ tasks, const chol_upd_src *__restrict__ srcs, int ntask, int per_xcd, int64_t arena_elems)
{
#ifdef M32_CLOCK
  const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  __shared__ float sA[M32_STAGES][M32_IMG];
  __shared__ float sB[M32_STAGES][M32_IMG];
  __shared__ chol_upd_src sS[M32_SRC_BATCH];
  __shared__ int sOk;
  const int tt = threadIdx.x, lane = tt & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tt >> 6);
  const int tid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per_xcd || tid >= ntask) return;
  const chol_upd_task t = tasks[tid];
  const int wr = wave & 1, wc = wave >> 1;
#ifndef M32_MFMA16
  const int xo32 = 32 * wr + (lane & 31), yo32 = 32 * wc + (lane & 31);
  f16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  const bool nq = 32 * wr < t.mv && 32 * wc < t.nv && (!t.lower || t.ar + 32 * wr + 31 >= t.br + 32 * wc); // (a quadrant above the diagonal of a SYRK tile stores nothing)
#define M32_CHUNK(sa_, sb_) m32_chunk(acc, sa_, sb_, lane, xo32, yo32, nq)
#else
  const int r15 = lane & 15, g = lane >> 4;
  const int xo = 32 * wr + r15, yo = 32 * wc + r15;
  f4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f4){ 0.f, 0.f, 0.f, 0.f };
  const int ni = max(0, min(2, (t.mv - 32 * wr + 15) >> 4)), nj = max(0, min(2, (t.nv - 32 * wc + 15) >> 4));
#define M32_CHUNK(sa_, sb_) m32_chunk(acc, sa_, sb_, g, xo, yo, ni, nj)
#endif
  bool full = t.mv == M32 && t.nv == M32;
  if (!full && arena_elems > 0) { // an edge tile: may the DMA read 64 rows of every source?
    if (tt == 0) sOk = 1;
    __syncthreads();
    for (int s = t.src_begin + tt; s < t.src_end; s += 256) {
      const chol_upd_src sd = srcs[s];
      const int kf = (sd.k / M32_KB) * M32_KB;
      if (kf > 0 && (sd.a_off + t.ar + (M32 - 1) + (int64_t)(kf - 1) * sd.lda >= arena_elems || sd.b_off + t.br + (M32 - 1) + (int64_t)(kf - 1) * sd.ldb >= arena_elems)) sOk = 0;
    }
    __syncthreads();
    full = sOk != 0;
  }
  if (full) {
    for (int sb = t.src_begin; sb < t.src_end; sb += M32_SRC_BATCH) {
      const int ns = min(M32_SRC_BATCH, t.src_end - sb);
      __builtin_amdgcn_s_barrier(); // the previous batch is done with sS and with the ring
      if (tt < ns) sS[tt] = srcs[sb + tt];
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      int total = 0;
      for (int s = 0; s < ns; ++s) total += sS[s].k / M32_KB;
      // issue cursor: source `is`, `left` full chunks of it to go, the lane's addresses of the next chunk in registers -- the descriptor in LDS is
      // read once per source, not once per chunk (five LDS round trips in front of every chunk's MFMAs otherwise).
      // One DMA instruction per wave, chunk and operand: wave w moves k-rows 4 w .. 4 w + 3 (256 floats), lane l the four floats at 4 l
      int is = -1, left = 0, issued = 0;
      const float *pa = nullptr, *pb = nullptr;
      int64_t sta = 0, stb = 0;
      const int e_ = 256 * wave + 4 * lane;
#define M32_ISSUE()                                                                                                   \
      {                                                                                                               \
        while (left == 0) {                                                                                           \
          ++is;                                                                                                       \
          left = sS[is].k / M32_KB;                                                                                   \
          pa = base + sS[is].a_off + t.ar + e_ % M32 + (int64_t)(e_ / M32) * sS[is].lda;                              \
          pb = base + sS[is].b_off + t.br + e_ % M32 + (int64_t)(e_ / M32) * sS[is].ldb;                              \
          sta = (int64_t)M32_KB * sS[is].lda; stb = (int64_t)M32_KB * sS[is].ldb;                                     \
        }                                                                                                             \
        const int st_ = issued % M32_STAGES;                                                                          \
        M32_DIAG_DMA(lds_dma16_f32(pa, &sA[st_][0] + M32_PS * wave));                                                 \
        M32_DIAG_DMA(lds_dma16_f32(pb, &sB[st_][0] + M32_PS * wave));                                                 \
        pa += sta; pb += stb; --left; ++issued;                                                                       \
      }
      for (int i = 0; i < M32_STAGES - 1 && issued < total; ++i) M32_ISSUE();
      for (int c = 0; c < total; ++c) {
        if (issued - c - 1 >= M32_STAGES - 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (M32_STAGES - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        M32_DIAG_BAR(__builtin_amdgcn_s_barrier()); // every wave's part of chunk c is in LDS; every wave has left chunk c - 1
        if (issued < total) M32_ISSUE(); // into the stage chunk c - 1 occupied
        M32_CHUNK(&sA[c % M32_STAGES][0], &sB[c % M32_STAGES][0]);
      }
#undef M32_ISSUE
    }
    __builtin_amdgcn_s_barrier(); // the tail path below re-uses the stages
  }
  // ---- register-staged path: everything for edge tiles the DMA may not serve, the K tails (K mod 16 columns) of the sources otherwise.
  //      Thread tt stages row tt & 63 of both operands, k-columns (tt >> 6) + 4 i of the chunk
  {
    const int srow = tt & (M32 - 1), skq = tt >> 6;
    const bool sva = srow < t.mv, svb = srow < t.nv;
    int buf = 0;
    for (int s = t.src_begin; s < t.src_end; ++s) {
      const chol_upd_src sd = srcs[s];
      const int K = sd.k, kbeg = full ? (K / M32_KB) * M32_KB : 0;
      if (kbeg >= K) continue;
      const float *A = base + sd.a_off + t.ar + srow;
      const float *Bp = base + sd.b_off + t.br + srow;
      float ra[4], rb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = kbeg + skq + 4 * i;
        ra[i] = (sva && k < K) ? A[(int64_t)k * sd.lda] : 0.f;
        rb[i] = (svb && k < K) ? Bp[(int64_t)k * sd.ldb] : 0.f;
      }
      for (int k0 = kbeg; k0 < K; k0 += M32_KB) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { sA[buf][i * M32_PS + skq * M32 + srow] = ra[i]; sB[buf][i * M32_PS + skq * M32 + srow] = rb[i]; } // k = skq + 4 i: row skq of piece i
        if (k0 + M32_KB < K) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int k = k0 + M32_KB + skq + 4 * i;
            ra[i] = (sva && k < K) ? A[(int64_t)k * sd.lda] : 0.f;
            rb[i] = (svb && k < K) ? Bp[(int64_t)k * sd.ldb] : 0.f;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // LDS-only barrier: the prefetch stays in flight
        M32_CHUNK(&sA[buf][0], &sB[buf][0]);
        buf ^= 1;
      }
    }
  }
#ifndef M32_MFMA16
  // epilogue: register q of lane l is C(row l & 31, column 8 (q >> 2) + 4 (l >> 5) + (q & 3)) of the wave's quadrant (first MFMA operand = the
  // column side); every C value of the lane requested before the first is used (clamped addresses, masked stores)
  if (nq) {
    const int r = 32 * wr + (lane & 31);
    const float *Cr = base + t.c_off + min(r, t.mv - 1);
    float cv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int c = min(32 * wc + 8 * (q >> 2) + 4 * (lane >> 5) + (q & 3), t.nv - 1);
      cv[q] = *(const volatile float *)(Cr + (int64_t)c * t.ldc);
    }
    float *C = base + t.c_off + r;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int c = 32 * wc + 8 * (q >> 2) + 4 * (lane >> 5) + (q & 3);
      if (r < t.mv && c < t.nv && (!t.lower || t.ar + r >= t.br + c)) C[(int64_t)c * t.ldc] = cv[q] - acc[q];
    }
  }
#else
  // epilogue: every C value of the lane requested before the first is used (clamped addresses, masked stores); register q of lane
  // (r15, g) is C(row 16 i + r15, column 16 j + 4 g + q) of the wave's quadrant
  float cv[2][2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = min(32 * wr + 16 * i + r15, t.mv - 1), c = min(32 * wc + 16 * j + 4 * g + q, t.nv - 1);
        cv[i][j][q] = *(const volatile float *)(base + t.c_off + r + (int64_t)c * t.ldc);
      }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = 32 * wr + 16 * i + r15;
      float *C = base + t.c_off + r;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 32 * wc + 16 * j + 4 * g + q;
        if (r < t.mv && c < t.nv && (!t.lower || t.ar + r >= t.br + c)) C[(int64_t)c * t.ldc] = cv[i][j][q] - acc[i][j][q];
      }
    }
#endif
#undef M32_CHUNK
#ifdef M32_CLOCK
  if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) { g_m32_clock[0] = __builtin_amdgcn_s_memtime() - clk_t0; g_m32_clock[1] = __builtin_amdgcn_s_memrealtime() - clk_r0; }
#endif
}

// ------------------------------------------------------------------------------------------------
// TRSM, throughput form (the fp32 schedule's only TRSM phases: large problems): the fp64 k_trsm_wt at fp32.  One wave per 16-row
// strip, SIXTEEN strips per 1024-thread workgroup sharing one staged image of the pivot block (up to 128 columns: 36 L tiles + 8
// inverses, one LDS-DMA instruction per tile -- lane l moves rows 4 (l & 3) .. + 3 of column l >> 2, which is the [k][c] image the
// MFMA operand reads want, and the workspace layout of the inverses verbatim); the strip's column tiles stay in registers and are
// the next MFMA's operand straight out of their accumulators (k = 4 g + s on both operands, as in k32_trsm).  k32_trsm reads L from
// global memory inside its MFMA loop: 25.7 of the 333 ms of the 100^3 factorisation.
// ------------------------------------------------------------------------------------------------
#define T32_MAXT ((CHOL32_MAXN + TS - 1) / TS)
#define T32_WAVES CHOL32_TRSM_GROUP
template <int T> __device__ __forceinline__ void trsm32_solve(f4 (&tile)[T32_MAXT], const float *simg, float *__restrict__ B, int n, int ldb, bool vrow, int r15, int g)
{
#pragma unroll
  for (int J = 0; J < T; ++J) {
    const float *sd = simg + (J * T - J * (J - 1) / 2) * (TS * TS); // slot (J, J): the inverse
    f4 x = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
    for (int s = 0; s < 4; ++s) x = __builtin_amdgcn_mfma_f32_16x16x4f32(sd[(4 * g + s) * TS + r15], tile[J][s], x, 0, 0, 0);
    if (vrow) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = J * TS + 4 * g + q;
        if (col < n) B[r15 + (int64_t)col * ldb] = x[q];
      }
    }
    float nx[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { nx[s] = -x[s]; asm volatile("" : "+v"(nx[s])); } // negated once per step, not once per MFMA
#pragma unroll
    for (int j = J + 1; j < T; ++j) {
      const float *sl = sd + (j - J) * (TS * TS);
      f4 acc = tile[j];
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sl[(4 * g + s) * TS + r15], nx[s], acc, 0, 0, 0);
      tile[j] = acc;
    }
  }
}
__global__ __launch_bounds__(64 * T32_WAVES) void k32_trsm_wt(float *__restrict__ base, const float *__restrict__ ws, const chol_trsm_desc *__restrict__ descs, int ndesc)
{
  __shared__ float sT[T32_MAXT * (T32_MAXT + 1) / 2][TS * TS]; // slot(J2, J) = J T - J (J - 1) / 2 + (J2 - J): tile L(J2, J) as [k][c]; diagonal slots: Linv(J, J)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int id0 = blockIdx.x * T32_WAVES;
  const chol_trsm_desc d0 = descs[id0];
  const float *Lm = base + d0.l_off;
  const float *W = ws + d0.dinv_off;
  const int n = d0.n, ldl = d0.ldl;
  const int T = (n + TS - 1) / TS;
  const int r15 = lane & 15, g = lane >> 4;
  int64_t b_off = d0.b_off;
  int m = 0, ldb = d0.ldb;
  if (id0 + wave < ndesc) {
    const chol_trsm_desc d = descs[id0 + wave];
    b_off = d.b_off; m = d.m; ldb = d.ldb;
  }
  float *B = base + b_off;
  const bool vrow = r15 < m;
  { // stage: slot h goes to wave h mod 16.  Rows past n of the last row tile are whatever follows in the panel (finite): they only
    // reach output columns >= n, which are never stored, and meet the zeros of the padded inverse
    const int nslots = T * (T + 1) / 2;
    for (int h = wave; h < nslots; h += T32_WAVES) {
      int J = 0, rem = h;
      while (rem >= T - J) { rem -= T - J; ++J; }
      const int J2 = J + rem;
      const float *src = J2 == J ? W + (int64_t)J * TS * TS + 4 * lane
                                 : Lm + (J2 * TS + 4 * (lane & 3)) + (int64_t)(J * TS + (lane >> 2)) * ldl;
      lds_dma16_f32(src, &sT[h][0]);
    }
  }
  f4 tile[T32_MAXT];
  {
    const int rb = min(r15, max(m - 1, 0));
#pragma unroll
    for (int J = 0; J < T32_MAXT; ++J) {
#pragma unroll
      for (int q = 0; q < 4; ++q) tile[J][q] = B[rb + (int64_t)min(J * TS + 4 * g + q, n - 1) * ldb];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads(); // every wave's LDS-DMA has landed
  if (m <= 0) return;
#pragma unroll
  for (int J = 0; J < T32_MAXT; ++J) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v = tile[J][q];
      asm volatile("" : "+v"(v));
      tile[J][q] = (vrow && J * TS + 4 * g + q < n) ? v : 0.f;
    }
  }
  const float *const s0 = &sT[0][0];
  switch (T) {
  case 1: trsm32_solve<1>(tile, s0, B, n, ldb, vrow, r15, g); break;
  case 2: trsm32_solve<2>(tile, s0, B, n, ldb, vrow, r15, g); break;
  case 3: trsm32_solve<3>(tile, s0, B, n, ldb, vrow, r15, g); break;
  case 4: trsm32_solve<4>(tile, s0, B, n, ldb, vrow, r15, g); break;
  case 5: trsm32_solve<5>(tile, s0, B, n, ldb, vrow, r15, g); break;
  case 6: trsm32_solve<6>(tile, s0, B, n, ldb, vrow, r15, g); break;
  case 7: trsm32_solve<7>(tile, s0, B, n, ldb, vrow, r15, g); break;
  case 8: trsm32_solve<8>(tile, s0, B, n, ldb, vrow, r15, g); break;
  default: break;
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
int chol32_launch_trsm_wt(float *base, const float *ws, const chol_trsm_desc *descs, int n, hipStream_t st)
{ // every aligned group of CHOL32_TRSM_GROUP descriptors shares one pivot block (m = 0: placeholder)
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k32_trsm_wt, dim3((n + T32_WAVES - 1) / T32_WAVES), dim3(64 * T32_WAVES), 0, st, base, ws, descs, n);
  return (int)hipGetLastError();
}
int chol32_launch_update(float *base, const chol_upd_task *tasks, const chol_upd_src *srcs, int ntask, hipStream_t st)
{
  if (ntask <= 0) return 0;
  const int per_xcd = (ntask + 7) / 8;
  hipLaunchKernelGGL(k32_update, dim3(per_xcd * 8), dim3(256), 0, st, base, tasks, srcs, ntask, per_xcd);
  return (int)hipGetLastError();
}
int chol32_launch_update_mt(float *base, const chol_upd_task *tasks, const chol_upd_src *srcs, int ntask, int64_t arena_elems, hipStream_t st)
{ // arena_elems: floats in the arena behind `base` (0: unknown -- edge tiles then stage their operands through registers)
  if (ntask <= 0) return 0;
  const int per_xcd = (ntask + 7) / 8;
  hipLaunchKernelGGL(k32_update_mt, dim3(per_xcd * 8), dim3(256), 0, st, base, tasks, srcs, ntask, per_xcd, arena_elems);
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
