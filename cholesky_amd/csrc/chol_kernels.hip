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

// ================================================================================================
// Dense pivot kernels.  Diagonal blocks are TS = 16 wide (one fp64 MFMA tile); the inverse of every
// 16x16 diagonal block of L is written to the workspace as W[blk][k * 16 + c] = Linv(c, k), the
// layout the TRSM kernels read as the MFMA "Y" operand.
//
// Register-resident design for pivots up to CHOL_RR_MAXN = 272 (17 tiles): the MI355X register
// file (512 KB per CU) is the only on-chip memory that holds a 259 x 259 fp64 lower triangle
// (269 KB; LDS has 160 KB), so the trailing matrix lives in VGPRs as 16x16 tiles in MFMA
// accumulator layout, spread round-robin over 15 "tile" waves of a 1024-thread workgroup; wave 0
// is the "factor" wave (diagonal tile: Cholesky + inverse).  An accumulator tile is directly a
// valid "X" operand of the next MFMA (register q of lane l holds column (l >> 4) + 4 q = k-step q
// of the operand map), so the panel solve X = T Linv^T and the trailing update T -= P_i P_j^T need
// no data movement for T; P goes through LDS.
// ================================================================================================
#define TS 16
#define RR_MAXT 17
// In-kernel cycle stamps of the factor wave: diagnostic builds only (-DCHOL_STAMPS, scripts/stamp_potrf.hip)
#ifdef CHOL_STAMPS
__device__ unsigned long long g_stamps[16];
#define STAMP_DECL unsigned long long st_[8], acc_[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_[i]) :: "memory"); __builtin_amdgcn_sched_barrier(0); if ((i) > 0) acc_[i] += st_[i] - st_[(i) - 1]; } while (0)
#define STAMP_FLUSH do { if (lane == 0) for (int i_ = 0; i_ < 8; ++i_) g_stamps[i_] = acc_[i_]; } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH
#endif
#define RR_TILE_WAVES 15
#define RR_SLOTS 11 /* ceil(17 * 18 / 2 / 15) */

__device__ __forceinline__ double readlane_f64(double v, int l)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

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

// 16x16 lower Cholesky, one row per lane (row = lane & 15; the four 16-lane groups of the wave
// compute the same thing).  a[c] = A(row, c) on entry (c <= row used), L(row, c) on exit.
// inv[j] = 1 / L(j, j) (wave uniform).  Returns the first non-positive pivot (1-based) or 0.
__device__ __forceinline__ int chol16_rows(double (&a)[TS], double (&inv)[TS])
{
  int bad = 0;
#pragma unroll
  for (int j = 0; j < TS; ++j) {
    const double d = readlane_f64(a[j], j);
    double akj[TS];
#pragma unroll
    for (int k = j + 1; k < TS; ++k) akj[k] = readlane_f64(a[j], k); // unscaled column j, overlaps the rsqrt chain
    if (!(d > 0.0) && bad == 0) bad = j + 1;
    const double r = readlane_f64(rsqrt_nr(d), 0); // wave uniform: keep it in SGPRs
    inv[j] = r;
    const double t = a[j] * (r * r);
#pragma unroll
    for (int k = j + 1; k < TS; ++k) a[k] = fma(-t, akj[k], a[k]);
    a[j] = a[j] * r; // row j: d / sqrt(d)
    __builtin_amdgcn_sched_barrier(0); // bound the live range of the broadcast scalars to one column
  }
  return bad;
}

// Inverse of the lower factor held row-per-lane: on exit lane c holds column c of Linv,
// x[r] = Linv(r, c) (zero for r < c).
__device__ __forceinline__ void linv16_cols(const double (&a)[TS], const double (&inv)[TS], double (&x)[TS], int lane15)
{
#pragma unroll
  for (int r = 0; r < TS; ++r) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < r; ++k) {
      const double lrk = readlane_f64(a[k], r); // L(r, k)
      if (k & 1) s1 = fma(lrk, x[k], s1); else s0 = fma(lrk, x[k], s0);
    }
    const double v = -(s0 + s1) * inv[r];
    x[r] = (lane15 == r) ? inv[r] : v; // rows above the diagonal of this column come out as -0 * inv = 0
    __builtin_amdgcn_sched_barrier(0); // keep the 120 broadcast pairs from being hoisted together
  }
}

// tile index -> (i, j) of the column-major enumeration of the lower triangle of a T x T tile grid
__device__ __forceinline__ void tile_of_index(int idx, int T, int &ti, int &tj)
{
  int j = 0;
  while (j < T && idx >= T - j) { idx -= T - j; ++j; }
  if (j >= T) { ti = -1; tj = 1 << 20; } else { ti = j + idx; tj = j; }
}

__global__ __launch_bounds__(1024) void k_potrf_rr(double *__restrict__ base, double *__restrict__ ws,
                                                   const chol_potrf_desc *__restrict__ descs, int *__restrict__ info)
{
  __shared__ double sPanel[RR_MAXT][TS][TS]; // [tile i][k][r]: solved panel of the current step
  __shared__ double sDiag[TS][TS + 1];       // [r][c]: diagonal tile on its way to / from the factor wave
  __shared__ double sLinv[TS][TS + 1];       // [k][c] = Linv(c, k)
  const chol_potrf_desc d = descs[blockIdx.x];
  double *A = base + d.a_off;
  double *W = ws + d.dinv_off;
  const int n = d.n, lda = d.lda;
  const int T = (n + TS - 1) / TS;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r15 = lane & 15, g = lane >> 4;

  // (i, j) of every lower-triangle tile, column-major enumeration; slot s of tile wave w holds tile
  // s * RR_TILE_WAVES + w.  Kept in LDS (one broadcast read per use) instead of 22 live scalars.
  __shared__ unsigned short sIJ[RR_SLOTS * RR_TILE_WAVES + 16];
  for (int t = threadIdx.x; t < RR_SLOTS * RR_TILE_WAVES; t += 1024) {
    int ti, tj;
    tile_of_index(t, T, ti, tj);
    sIJ[t] = ti < 0 ? (unsigned short)0xffff : (unsigned short)(ti | (tj << 8));
  }
  __syncthreads();

  if (wave == 0) {
    // ------------------------------------------------------------------ factor wave
    __syncthreads(); // B0: tile (0,0) published
    STAMP_DECL;
    for (int k = 0; k < T; ++k) {
      double a[TS], inv[TS], x[TS];
      STAMP(0);
#pragma unroll
      for (int c = 0; c < TS; ++c) a[c] = sDiag[r15][c];
      STAMP(1);
      const int bad = chol16_rows(a, inv);
      STAMP(2);
      if (bad && k * TS + bad <= n && lane == 0) {
        if (atomicCAS(&info[0], 0, k * TS + bad) == 0) info[1] = d.sep;
      }
      linv16_cols(a, inv, x, r15);
      STAMP(3);
      if (lane < TS) {
#pragma unroll
        for (int c = 0; c < TS; ++c) sLinv[lane][c] = x[c]; // lane = column index kk: sLinv[kk][cc] = Linv(cc, kk)
      }
      STAMP(4);
      __syncthreads(); // B1: Linv(k) published
      STAMP(5);
      if (lane < TS) { // global stores overlap the tile waves' panel solve
        const int row = k * TS + lane;
        double *dst = A + row + (int64_t)(k * TS) * lda;
        double *wd = W + (int64_t)k * TS * TS + lane * TS;
#pragma unroll
        for (int c = 0; c < TS; ++c) wd[c] = x[c];
        if (row < n) {
#pragma unroll
          for (int c = 0; c < TS; ++c)
            if (c <= lane) dst[(int64_t)c * lda] = a[c]; // L(k,k), lower part
        }
      }
      __syncthreads(); // B2: panel solved
      STAMP(6);
      __syncthreads(); // B3: trailing update done, tile (k+1,k+1) published
      STAMP(7);
    }
    STAMP_FLUSH;
  } else {
    // ------------------------------------------------------------------ tile waves
    d4 tile[RR_SLOTS];
    const int w = wave - 1;
    int ijp[RR_SLOTS]; // packed (i | j << 8) per slot, wave uniform (11 scalars)
#pragma unroll
    for (int s = 0; s < RR_SLOTS; ++s) ijp[s] = __builtin_amdgcn_readfirstlane((int)sIJ[s * RR_TILE_WAVES + w]);
#define SLOT_IJ(s, ti_, tj_) \
    const int ti_ = (ijp[s] == 0xffff) ? -1 : (ijp[s] & 0xff), tj_ = (ijp[s] == 0xffff) ? (1 << 20) : (ijp[s] >> 8)
#pragma unroll
    for (int s = 0; s < RR_SLOTS; ++s) {
      SLOT_IJ(s, ti, tj);
      d4 v = { 0.0, 0.0, 0.0, 0.0 };
      if (ti >= 0) {
        const int row = ti * TS + r15;
        const bool rowok = row < n;
        const double *src = A + row + (int64_t)(tj * TS + g) * lda;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = g + 4 * q, col = tj * TS + c;
          double e = (row == col) ? 1.0 : 0.0; // identity padding of the last partial diagonal tile
          if (rowok && col < n) e = (ti > tj || c <= r15) ? src[(int64_t)(4 * q) * lda] : 0.0;
          v[q] = e;
        }
        if (ti == 0 && tj == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) sDiag[r15][g + 4 * q] = v[q];
        }
      }
      tile[s] = v;
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads(); // B0
    for (int k = 0; k < T; ++k) {
      __syncthreads(); // B1
      // ---- panel solve: X = T Linv^T for owned tiles (i, k), i > k
#pragma unroll
      for (int s = 0; s < RR_SLOTS; ++s) {
        SLOT_IJ(s, ti, tj);
        if (tj == k && ti > k) {
          d4 x = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
          for (int st = 0; st < 4; ++st) x = __builtin_amdgcn_mfma_f64_16x16x4f64(sLinv[4 * st + g][r15], tile[s][st], x, 0, 0, 0);
          tile[s] = x; // final values of L(i, k): stay in registers until the epilogue stores them
#pragma unroll
          for (int q = 0; q < 4; ++q) sPanel[ti][g + 4 * q][r15] = x[q];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads(); // B2
      // ---- trailing update: T(i, j) -= P_i P_j^T for owned tiles with j > k
#pragma unroll
      for (int s = 0; s < RR_SLOTS; ++s) {
        SLOT_IJ(s, ti, tj);
        if (tj > k && tj < T) {
          d4 acc = tile[s];
#pragma unroll
          for (int st = 0; st < 4; ++st)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sPanel[tj][4 * st + g][r15], -sPanel[ti][4 * st + g][r15], acc, 0, 0, 0);
          tile[s] = acc;
          if (tj == k + 1 && ti == k + 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) sDiag[r15][g + 4 * q] = acc[q];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads(); // B3
    }
    // epilogue: every off-diagonal tile now holds its block of L
#pragma unroll
    for (int s = 0; s < RR_SLOTS; ++s) {
      SLOT_IJ(s, ti, tj);
      if (ti > tj && ti >= 0) {
        const int row = ti * TS + r15;
        double *dst = A + row + (int64_t)(tj * TS + g) * lda;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (row < n) dst[(int64_t)(4 * q) * lda] = tile[s][q];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef SLOT_IJ
  }
}

// ------------------------------------------------------------------------------------------------
// POTRF for pivots larger than CHOL_RR_MAXN: same blocking, trailing matrix in global memory (L2).
// One workgroup of 256 threads per pivot.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_potrf_big(double *__restrict__ base, double *__restrict__ ws,
                                                   const chol_potrf_desc *__restrict__ descs, int *__restrict__ info)
{
  __shared__ double sLinv[TS][TS + 1];
  const chol_potrf_desc d = descs[blockIdx.x];
  double *A = base + d.a_off;
  double *W = ws + d.dinv_off;
  const int n = d.n, lda = d.lda;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r15 = lane & 15, g = lane >> 4;
  for (int k = 0, j0 = 0; j0 < n; ++k, j0 += TS) {
    const int below = n - j0 - TS; // rows under the diagonal tile (may be <= 0)
    if (wave == 0) {
      double a[TS], inv[TS], x[TS];
      const int row = j0 + r15;
#pragma unroll
      for (int c = 0; c < TS; ++c) {
        double v = (r15 == c) ? 1.0 : 0.0;
        if (row < n && j0 + c < n) v = (c <= r15) ? A[row + (int64_t)(j0 + c) * lda] : 0.0;
        a[c] = v;
      }
      const int bad = chol16_rows(a, inv);
      if (bad && j0 + bad <= n && lane == 0) {
        if (atomicCAS(&info[0], 0, j0 + bad) == 0) info[1] = d.sep;
      }
      linv16_cols(a, inv, x, r15);
      if (lane < TS) {
#pragma unroll
        for (int c = 0; c < TS; ++c) {
          if (row < n && c <= lane) A[row + (int64_t)(j0 + c) * lda] = a[c];
          sLinv[lane][c] = x[c];
          W[(int64_t)k * TS * TS + lane * TS + c] = x[c];
        }
      }
    }
    __syncthreads();
    // panel: 16-row tiles below the diagonal tile, X = T Linv^T
    const int nt = below > 0 ? (below + TS - 1) / TS : 0;
    for (int t = wave; t < nt; t += 4) {
      const int row = j0 + TS + t * TS + r15;
      d4 x = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const double tv = (row < n) ? A[row + (int64_t)(j0 + 4 * st + g) * lda] : 0.0;
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(sLinv[4 * st + g][r15], tv, x, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (row < n) A[row + (int64_t)(j0 + g + 4 * q) * lda] = x[q];
    }
    __syncthreads();
    // trailing update of the lower triangle
    if (nt > 0) {
      const int ntl = nt * (nt + 1) / 2;
      const double *P = A + (j0 + TS) + (int64_t)j0 * lda;
      double *Tm = A + (j0 + TS) + (int64_t)(j0 + TS) * lda;
      for (int t = wave; t < ntl; t += 4) {
        int tr = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
        while (tr * (tr + 1) / 2 > t) --tr;
        while ((tr + 1) * (tr + 2) / 2 <= t) ++tr;
        const int tc = t - tr * (tr + 1) / 2;
        const int mv = min(TS, below - tr * TS), nv = min(TS, below - tc * TS);
        d4 acc = { 0.0, 0.0, 0.0, 0.0 };
        acc = rank_k_16x16(acc, P + tr * TS, lda, mv, P + tc * TS, lda, nv, TS, lane);
        double *C = Tm + (tr * TS + r15) + (int64_t)(tc * TS) * lda;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = g + 4 * q;
          if (r15 < mv && c < nv && (tr != tc || r15 >= c)) C[(int64_t)c * lda] -= acc[q];
        }
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Inverses of the 16x16 diagonal blocks of an already factored L (BLAS-/task-level TRSM entry
// points, where L was not produced by a POTRF kernel of the same call chain).  One wave per block.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_dinv(const double *__restrict__ Lp, int n, int ldl, double *__restrict__ W)
{
  const int j0 = blockIdx.x * TS, lane = threadIdx.x, r15 = lane & 15;
  double a[TS], inv[TS], x[TS];
  const int row = j0 + r15;
#pragma unroll
  for (int c = 0; c < TS; ++c) {
    double v = (r15 == c) ? 1.0 : 0.0;
    if (row < n && j0 + c < n) v = (c <= r15) ? Lp[row + (int64_t)(j0 + c) * ldl] : 0.0;
    a[c] = v;
  }
#pragma unroll
  for (int j = 0; j < TS; ++j) inv[j] = readlane_f64(1.0 / readlane_f64(a[j], j), 0);
  linv16_cols(a, inv, x, r15);
  if (lane < TS) {
#pragma unroll
    for (int c = 0; c < TS; ++c) W[(int64_t)blockIdx.x * TS * TS + lane * TS + c] = x[c];
  }
}

// ------------------------------------------------------------------------------------------------
// TRSM: B <- B L^-T for a strip of <= 16 rows (cblas_dtrsm Right/Lower/Trans/NonUnit alpha=1,
// blas.rg:99), n <= CHOL_RR_MAXN.  The strip's 16x16 column tiles live in registers (accumulator
// layout), tile J owned by wave J mod 4.  Right-looking: the owner solves X_J = T_J Linv_J^T (the
// accumulator tile is the MFMA X operand as it stands), publishes X_J in LDS (double buffered, one
// barrier per step), and every wave applies T_J'' -= X_J L(J'', J)^T to its tiles J'' > J with the
// L tile streamed from global memory / L2.
// ------------------------------------------------------------------------------------------------
#define TRSM_SLOTS 5 /* ceil(17 / 4) */
__global__ __launch_bounds__(256) void k_trsm_rr(double *__restrict__ base, const double *__restrict__ ws,
                                                 const chol_trsm_desc *__restrict__ descs)
{
  __shared__ double sX[2][TS][TS]; // [buf][k][r]
  const chol_trsm_desc d = descs[blockIdx.x];
  const double *Lm = base + d.l_off;
  const double *W = ws + d.dinv_off;
  double *B = base + d.b_off;
  const int n = d.n, m = d.m, ldl = d.ldl, ldb = d.ldb;
  const int T = (n + TS - 1) / TS;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r15 = lane & 15, g = lane >> 4;
  const bool vrow = r15 < m;

  d4 tile[TRSM_SLOTS];
#pragma unroll
  for (int s = 0; s < TRSM_SLOTS; ++s) {
    const int J = wave + 4 * s;
    d4 v = { 0.0, 0.0, 0.0, 0.0 };
    if (J < T) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = J * TS + g + 4 * q;
        v[q] = (vrow && col < n) ? B[r15 + (int64_t)col * ldb] : 0.0;
      }
    }
    tile[s] = v;
  }
#pragma unroll
  for (int J = 0; J < RR_MAXT; ++J) {
    if (J < T) {
      // L tiles of this step's updates: issued ahead of the solve and the barrier that hide their latency
      double lpre[TRSM_SLOTS][4];
#pragma unroll
      for (int s = 0; s < TRSM_SLOTS; ++s) {
        const int J2 = wave + 4 * s;
        const int lrow = J2 * TS + r15;
        const bool ok = J2 > J && lrow < n;
        const double *lp = Lm + (ok ? lrow : 0) + (int64_t)(J * TS + g) * ldl;
#pragma unroll
        for (int st = 0; st < 4; ++st) lpre[s][st] = ok ? lp[(int64_t)(4 * st) * ldl] : 0.0;
      }
      if ((J & 3) == wave) { // solve
        const double *V = W + (int64_t)J * TS * TS;
        d4 x = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
        for (int st = 0; st < 4; ++st) x = __builtin_amdgcn_mfma_f64_16x16x4f64(V[(4 * st + g) * TS + r15], tile[J >> 2][st], x, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = g + 4 * q, col = J * TS + c;
          sX[J & 1][c][r15] = x[q];
          if (vrow && col < n) B[r15 + (int64_t)col * ldb] = x[q];
        }
      }
      __syncthreads();
#pragma unroll
      for (int s = 0; s < TRSM_SLOTS; ++s) {
        const int J2 = wave + 4 * s;
        if (J2 > J && J2 < T) {
          d4 acc = tile[s];
#pragma unroll
          for (int st = 0; st < 4; ++st)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(lpre[s][st], -sX[J & 1][4 * st + g][r15], acc, 0, 0, 0);
          tile[s] = acc;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// TRSM for pivots larger than CHOL_RR_MAXN: one independent wave per 16-row strip (4 strips per
// workgroup), left-looking from global memory: T_J = B_J - X_<J L(J,<J)^T, X_J = T_J Linv_J^T.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_trsm_big(double *__restrict__ base, const double *__restrict__ ws,
                                                  const chol_trsm_desc *__restrict__ descs, int ndesc)
{
  const int lane = threadIdx.x & 63;
  const int id = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (id >= ndesc) return;
  const chol_trsm_desc d = descs[id];
  const double *Lm = base + d.l_off;
  const double *W = ws + d.dinv_off;
  double *B = base + d.b_off;
  const int n = d.n, m = d.m, ldl = d.ldl, ldb = d.ldb;
  const int r15 = lane & 15, g = lane >> 4;
  for (int J = 0, j0 = 0; j0 < n; ++J, j0 += TS) {
    const int nv = min(TS, n - j0);
    d4 acc = { 0.0, 0.0, 0.0, 0.0 };
    acc = rank_k_16x16(acc, B, ldb, m, Lm + j0, ldl, nv, j0, lane);
    d4 t;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = g + 4 * q;
      t[q] = (r15 < m && c < nv) ? B[r15 + (int64_t)(j0 + c) * ldb] - acc[q] : 0.0;
    }
    const double *V = W + (int64_t)J * TS * TS;
    d4 x = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
    for (int st = 0; st < 4; ++st) x = __builtin_amdgcn_mfma_f64_16x16x4f64(V[(4 * st + g) * TS + r15], t[st], x, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = g + 4 * q;
      if (r15 < m && c < nv) B[r15 + (int64_t)(j0 + c) * ldb] = x[q];
    }
    __threadfence_block(); // the next column block reads these columns through other lanes of this wave
  }
}

#define NB 32 /* block width of the triangular solves below */
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
  __shared__ double sx[32];
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
  __shared__ double sx[32];
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
{ // pivots up to CHOL_RR_MAXN: register-resident kernel, one 1024-thread workgroup each
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_potrf_rr, dim3(n), dim3(1024), 0, st, base, ws, descs, info);
  return (int)hipGetLastError();
}
int chol_launch_potrf_big(double *base, double *ws, const chol_potrf_desc *descs, int n, int *info, hipStream_t st)
{
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_potrf_big, dim3(n), dim3(256), 0, st, base, ws, descs, info);
  return (int)hipGetLastError();
}
int chol_launch_dinv(const double *L, int n, int ldl, double *W, hipStream_t st)
{
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_dinv, dim3((n + TS - 1) / TS), dim3(64), 0, st, L, n, ldl, W);
  return (int)hipGetLastError();
}
int chol_launch_trsm(double *base, const double *ws, const chol_trsm_desc *descs, int n, hipStream_t st)
{ // strips of pivots up to CHOL_RR_MAXN
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_trsm_rr, dim3(n), dim3(256), 0, st, base, ws, descs);
  return (int)hipGetLastError();
}
int chol_launch_trsm_big(double *base, const double *ws, const chol_trsm_desc *descs, int n, hipStream_t st)
{
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_trsm_big, dim3((n + 3) / 4), dim3(256), 0, st, base, ws, descs, n);
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
