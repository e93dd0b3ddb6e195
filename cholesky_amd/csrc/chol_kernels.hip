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

// workgroup barrier that orders LDS traffic only (global loads / stores stay in flight across it)
__device__ __forceinline__ void lds_barrier()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}


// Data handed from one workgroup to another INSIDE a launch (fused POTRF+TRSM+update launch) is written with
// agent-scope stores (write-through, device-coherent) and read with agent-scope loads, so that it is seen across
// the XCDs' L2s while the kernel runs; PUB = false is a plain access.
template <bool PUB> __device__ __forceinline__ void gstore(double *p, double v)
{
  if (PUB) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool PUB> __device__ __forceinline__ double gload(const double *p)
{
  if (PUB) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}

// a bounded spin gave up: the factorisation is reported as failed.  The code is materialised HERE (opaque to the compiler): hoisted out of the
// many spin loops of the program launch it became a vector register that lived -- and was spilled -- across every role body
__device__ __forceinline__ void report_stall(int *info)
{
#ifdef STALL_PLAIN
  atomicCAS(&info[0], 0, CHOLAMD_ERR_STALL);
#else
  int e;
  asm volatile("v_mov_b32 %0, %1" : "=v"(e) : "i"(CHOLAMD_ERR_STALL));
  atomicCAS(&info[0], 0, e);
#endif
}
// wait until a progress word written by workgroups of the same launch reaches `target` (columns published by a
// pivot's POTRF workgroup; TRSM workgroups finished).  Producers have lower block indices than their consumers,
// are dispatched first and never wait for a consumer; the spin is bounded all the same: after about two seconds
// (every in-launch wait gives up after ~2 s of polling -- long enough for a GPU shared with other streams or processes,
// where a producer may simply not be resident yet) it gives up and reports through info (the factorisation then fails
// loudly, CHOLAMD_ERR_STALL, instead of hanging the GPU).
__device__ __forceinline__ int wait_progress(const int *progress, int target, int seen, int *info)
{
  if (seen >= target) return seen;
  int v = 0;
  for (int it = 0; it < (1 << 23); ++it) { // x ~0.25 us per poll
    v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (v >= target) return v;
#ifndef PROG_SLEEP
#define PROG_SLEEP 8
#endif
    __builtin_amdgcn_s_sleep(PROG_SLEEP);
  }
  if ((threadIdx.x & 63) == 0) report_stall(info); // no progress: the factorisation is reported as failed
  return target;
}
// ------------------------------------------------------------------------------------------------
// acc(r, c) += sum_{k < K} X[r + k ldx] * Y[c + k ldy],  r0 <= r < mv, c0 <= c < nv (rows outside are read as 0)
// ------------------------------------------------------------------------------------------------
template <bool PUB = false>
__device__ __forceinline__ d4 rank_k_16x16(d4 acc, const double *__restrict__ X, int ldx, int mv,
                                           const double *__restrict__ Y, int ldy, int nv, int K, int lane, int r0 = 0, int c0 = 0)
{ // rows [r0, mv) of X and [c0, nv) of Y take part
  const int r = lane & 15, kq = lane >> 4;
  const bool vx = r >= r0 && r < mv, vy = r >= c0 && r < nv;
  const double *px = X + r + (int64_t)kq * ldx;
  const double *py = Y + r + (int64_t)kq * ldy;
  int k0 = 0;
  if (PUB) {
    // in-launch hand-offs are read past the L2 of the producer's XCD: a memory round trip per batch.  Two batches of sixteen
    // loads are kept in flight (the next trip's loads are issued ahead of this trip's MFMAs)
    double x[8], y[8];
    if (K >= 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { x[u] = vx ? gload<PUB>(&px[(int64_t)(4 * u) * ldx]) : 0.0; y[u] = vy ? gload<PUB>(&py[(int64_t)(4 * u) * ldy]) : 0.0; }
    }
    for (; k0 + 32 <= K; k0 += 32) {
      double xn[8], yn[8];
      px += 32 * (int64_t)ldx; py += 32 * (int64_t)ldy;
      const bool more = k0 + 64 <= K;
      if (more) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { xn[u] = vx ? gload<PUB>(&px[(int64_t)(4 * u) * ldx]) : 0.0; yn[u] = vy ? gload<PUB>(&py[(int64_t)(4 * u) * ldy]) : 0.0; }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y[u], x[u], acc, 0, 0, 0);
      if (more) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { x[u] = xn[u]; y[u] = yn[u]; }
      }
    }
  } else
  for (; k0 + 32 <= K; k0 += 32) { // eight MFMAs per trip, sixteen loads in flight
    double x[8], y[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { x[u] = vx ? gload<PUB>(&px[(int64_t)(4 * u) * ldx]) : 0.0; y[u] = vy ? gload<PUB>(&py[(int64_t)(4 * u) * ldy]) : 0.0; }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y[u], x[u], acc, 0, 0, 0);
    px += 32 * (int64_t)ldx; py += 32 * (int64_t)ldy;
  }
  for (; k0 + 16 <= K; k0 += 16) { // four MFMAs per trip, eight loads in flight
    double x0 = vx ? gload<PUB>(&px[0]) : 0.0, y0 = vy ? gload<PUB>(&py[0]) : 0.0;
    double x1 = vx ? gload<PUB>(&px[4 * (int64_t)ldx]) : 0.0, y1 = vy ? gload<PUB>(&py[4 * (int64_t)ldy]) : 0.0;
    double x2 = vx ? gload<PUB>(&px[8 * (int64_t)ldx]) : 0.0, y2 = vy ? gload<PUB>(&py[8 * (int64_t)ldy]) : 0.0;
    double x3 = vx ? gload<PUB>(&px[12 * (int64_t)ldx]) : 0.0, y3 = vy ? gload<PUB>(&py[12 * (int64_t)ldy]) : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y0, x0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y1, x1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y2, x2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y3, x3, acc, 0, 0, 0);
    px += 16 * (int64_t)ldx; py += 16 * (int64_t)ldy;
  }
  for (; k0 < K; k0 += 4) {
    const bool vk = k0 + kq < K;
    double x = (vx && vk) ? gload<PUB>(&px[0]) : 0.0, y = (vy && vk) ? gload<PUB>(&py[0]) : 0.0;
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

// Staged waits of an extend-add job of the program launch (chol_upd_src.stage): the job's sources become readable one source pivot
// block after the other; a wave looks at the remaining waits in ONE poll (every lane one entry) and polls again only in front of a
// source whose stage does not hold yet.  The wave that polled reads the source with sc1 loads afterwards (the hand-off form of
// wait_list / the job-level waits).
struct stage_waits { const chol_wait *w; int n; const int *ctr; const int *ctr_total; int epoch; int *info; };
__device__ __forceinline__ int stages_holding(const stage_waits &sw, int lane)
{ // leading waits that hold
  int lead = 0;
  for (int b0 = 0; b0 < sw.n; b0 += 64) {
    bool ok = true;
    if (b0 + lane < sw.n) { const chol_wait wt = sw.w[b0 + lane]; ok = __hip_atomic_load(&sw.ctr[wt.ctr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= wt.value + sw.epoch * sw.ctr_total[wt.ctr]; }
    const unsigned long long notok = __ballot(!ok);
    if (notok) return lead + __builtin_ctzll(notok);
    lead = min(b0 + 64, sw.n);
  }
  return lead;
}
__device__ __forceinline__ int wait_stage(const stage_waits &sw, int need, int lane)
{
  int have = stages_holding(sw, lane);
  for (int it = 0; have < need; ++it) {
    if (it >= (1 << 22)) { if (lane == 0) report_stall(sw.info); return sw.n; } // x ~0.5 us per poll
#ifndef STAGE_SLEEP
#define STAGE_SLEEP 16
#endif
    __builtin_amdgcn_s_sleep(STAGE_SLEEP);
    have = stages_holding(sw, lane);
  }
  return have;
}
// ------------------------------------------------------------------------------------------------
// UPDATE: one workgroup (4 waves) per 16x16 output sub-tile of a target C tile.  The tile's sources
// are walked in the reference's program order; the four waves split every source's K range, or -- from
// four sources on -- the sources themselves (the tasks are latency bound: dependent descriptor ->
// operand -> MFMA chains, so 4x the chains in flight is 4x less time); partial sums meet in LDS and
// are added in a fixed order (deterministic).
//   C <- C - sum_s A_s B_s^T   (cblas_dgemm NoTrans/Trans alpha=-1 beta=1, blas.rg:139;
//                               cblas_dsyrk Lower alpha=-1 beta=1, blas.rg:187 for `lower` tiles)
// Task ids are remapped so that consecutive tasks (sub-tiles of one target, sharing their source
// panels) run on the same XCD and hit in its L2.
// ------------------------------------------------------------------------------------------------
// one task = one 16x16 output sub-tile, four waves (wave = 0..3 of the task's group); sAcc = the group's LDS slots.
// `live` = false: a placeholder that only keeps the workgroup's barrier count uniform.  `wait` (fused launch): the
// counter of finished TRSM workgroups and the value it must reach before the sources may be read -- the task
// descriptor and the C sub-tile are requested before that wait.
template <bool PUB>
__device__ __forceinline__ void update_task_body(double *__restrict__ base, const chol_upd_task t, const chol_upd_src *__restrict__ srcs, double (*sAcc)[4][64],
                                                 int wave, int lane, bool live, const int *wait, int wait_target, int *info, const stage_waits sw = stage_waits())
{
  // the C sub-tile is requested first (wave 0; clamped addresses, masked at the store): read at the end, each of
  // its four columns would be a memory round trip of its own on the tail of every task
  const int r = lane & 15, g = lane >> 4;
  double cv[4] = { 0.0, 0.0, 0.0, 0.0 };
  double *C = base + t.c_off;
  if (live && wave == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double *cp = C + min(r, t.mv - 1) + (int64_t)min(g + 4 * q, t.nv - 1) * t.ldc;
      cv[q] = PUB ? gload<true>(cp) : *(const volatile double *)cp; // in-launch hand-off: the last writer may be another workgroup of this launch
    }
  }
  if (PUB && wait) (void)wait_progress(wait, wait_target, 0, info);
  d4 acc = { 0.0, 0.0, 0.0, 0.0 };
  const int nsrc = live ? t.src_end - t.src_begin : 0;
  int st_done = (PUB && sw.n > 0 && nsrc > 0) ? stages_holding(sw, lane) : 0;
  if (nsrc >= 4) {
    // many sources (a target high in the tree collects one per descendant): the waves take whole sources
    // round-robin, so four descriptor -> operand load chains are in flight instead of one
    for (int s = t.src_begin + wave; s < t.src_end; s += 4) {
      const chol_upd_src sd = srcs[s];
      if (PUB && sd.stage > st_done) st_done = wait_stage(sw, sd.stage, lane);
      const int r0 = sd.range & 255, r1 = sd.range ? (sd.range >> 8) & 255 : t.mv, c0 = (sd.range >> 16) & 255, c1 = sd.range ? (sd.range >> 24) & 255 : t.nv;
      acc = rank_k_16x16<PUB>(acc, base + sd.a_off + t.ar, sd.lda, r1, base + sd.b_off + t.br, sd.ldb, c1, sd.k, lane, r0, c0);
    }
  } else {
    for (int s = t.src_begin; s < t.src_end; ++s) {
      const chol_upd_src sd = srcs[s];
      if (PUB && live && sd.stage > st_done) st_done = wait_stage(sw, sd.stage, lane);
      const int kc = ((((sd.k + 3) >> 2) + 3) >> 2) << 2; // K per wave, a multiple of 4
      const int k_lo = wave * kc;
      if (k_lo < sd.k) {
        const int kn = min(kc, sd.k - k_lo);
        const int r0 = sd.range & 255, r1 = sd.range ? (sd.range >> 8) & 255 : t.mv, c0 = (sd.range >> 16) & 255, c1 = sd.range ? (sd.range >> 24) & 255 : t.nv;
        acc = rank_k_16x16<PUB>(acc, base + sd.a_off + t.ar + (int64_t)k_lo * sd.lda, sd.lda, r1,
                           base + sd.b_off + t.br + (int64_t)k_lo * sd.ldb, sd.ldb, c1, kn, lane, r0, c0);
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) sAcc[wave - 1][q][lane] = acc[q];
  }
  lds_barrier(); // the group's partial sums are in LDS (uniform over the workgroup: one barrier per task)
  if (live && wave == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = g + 4 * q;
      const double v = ((acc[q] + sAcc[0][q][lane]) + sAcc[1][q][lane]) + sAcc[2][q][lane];
      if (r < t.mv && c < t.nv && (!t.lower || r >= c)) gstore<PUB>(&C[r + (int64_t)c * t.ldc], cv[q] - v);
    }
  }
}
// the same task by ONE wave (program launch: twelve tasks of a job at a time, one per wave -- no K split, no LDS reduction,
// no barrier; sources in program order, K in order: deterministic)
template <bool PUB>
__device__ __forceinline__ void update_task_wave(double *__restrict__ base, const chol_upd_task t, const chol_upd_src *__restrict__ srcs, int lane, const stage_waits sw = stage_waits())
{
  const int r = lane & 15, g = lane >> 4;
  double cv[4];
  double *C = base + t.c_off;
#pragma unroll
  for (int q = 0; q < 4; ++q) cv[q] = gload<PUB>(C + min(r, t.mv - 1) + (int64_t)min(g + 4 * q, t.nv - 1) * t.ldc);
  d4 acc = { 0.0, 0.0, 0.0, 0.0 };
  int st_done = (PUB && sw.n > 0) ? stages_holding(sw, lane) : 0;
  for (int s = t.src_begin; s < t.src_end; ++s) {
    const chol_upd_src sd = srcs[s];
    if (PUB && sd.stage > st_done) st_done = wait_stage(sw, sd.stage, lane);
    const int r0 = sd.range & 255, r1 = sd.range ? (sd.range >> 8) & 255 : t.mv, c0 = (sd.range >> 16) & 255, c1 = sd.range ? (sd.range >> 24) & 255 : t.nv;
    acc = rank_k_16x16<PUB>(acc, base + sd.a_off + t.ar, sd.lda, r1, base + sd.b_off + t.br, sd.ldb, c1, sd.k, lane, r0, c0);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = g + 4 * q;
    if (r < t.mv && c < t.nv && (!t.lower || r >= c)) gstore<PUB>(&C[r + (int64_t)c * t.ldc], cv[q] - acc[q]);
  }
}
__global__ __launch_bounds__(256) void k_update(double *__restrict__ base, const chol_upd_task *__restrict__ tasks,
                                                const chol_upd_src *__restrict__ srcs, int ntask, int per_xcd)
{
  __shared__ double sAcc[3][4][64];
  const int tid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per_xcd || tid >= ntask) return;
  update_task_body<false>(base, tasks[tid], srcs, sAcc, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), threadIdx.x & 63, true, nullptr, 0, nullptr);
}

// ------------------------------------------------------------------------------------------------
// UPDATE, macro-tile form for targets larger than one MFMA tile: one workgroup per 64x64 block of a
// target C tile.  The source panels are staged through LDS in 16-deep K chunks ([k][row] images, so
// the MFMA operand reads are 512 contiguous bytes per k-step and conflict free), double buffered
// with the next chunk's global loads in flight during the MFMAs; the four waves own the 2x2 grid of
// 32x32 quadrants (4 accumulators each), so every staged operand feeds two MFMAs: 8x the arithmetic
// intensity of k_update.  Same task / source lists, same program-order accumulation.
// ------------------------------------------------------------------------------------------------
// One LDS-DMA instruction (global_load_lds_dwordx4): lane l's 16 bytes at `g` land at lds + 16 l.  Written as asm so that the
// compiler does not see a vector-memory operation: through the builtin it waits vmcnt(0) ahead of the next LDS read, which
// serialises the ring; here the waits are the kernel's own counted s_waitcnt vmcnt(N) (cdna_hip_programming.md 5.7).  M0 (the LDS
// base of the transfer) is saved and restored inside the statement.
__device__ __forceinline__ void lds_dma16(const double *g, double *lds)
{
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) double *)lds);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}
#define MT 64
#ifndef MKB
#define MKB 16
#endif
#ifndef MT_STAGES
#define MT_STAGES 2 /* measured (scripts/mt_bench.hip, 8192^2 SYRK, TF/s at K = 144 / 512): 16-deep chunks x 2 stages 38.2 / 49.0, x 3 34.3 / 46.2, x 4 27.3 / 40.5; 8-deep x 3 35.9 / 44.5, x 4 35.5 / 44.8, x 6 31.3 / 40.3: workgroups per CU (LDS) beat prefetch depth; register-staged double buffering (round 1) 35 / 44 */
#endif
/* MT_STAGES: LDS ring of the full-tile path: stages of one 16-deep chunk of both operands */
#define MT_SRC_BATCH 32 /* source descriptors held in LDS at a time */
/* timing diagnostics of scripts/mt_bench.hip (wrong results): -DMT_NODMA leaves the operand transfers out, -DMT_NOBAR the ring's barriers */
#ifdef MT_NODMA
#define MT_DIAG_DMA(x_) ((void)0)
#else
#define MT_DIAG_DMA(x_) x_
#endif
#ifdef MT_NOBAR
#define MT_DIAG_BAR(x_) ((void)0)
#else
#define MT_DIAG_BAR(x_) x_
#endif
// One macro tile of TM x TN per workgroup of WR x WC waves, each wave a (TM / WR) x (TN / WC) block of 16x16 accumulators.
//   64 x 64, 2 x 2 waves (k_update_mt): 2 x 2 accumulators per wave, 8 flop per byte staged into LDS.
//   Measured against it (scripts/mt_bench.hip 8192^2 SYRK, TF/s at K = 144 / 432 / 512): 128 x 128 tiles, 4 x 2 waves of
//   2 x 4 accumulators, 16 flop per byte: 27 / 41 / 42 against 38 / 49 / 49 -- two workgroups of eight waves per CU (64 KB of LDS each)
//   hide less than four or five of four; 128 x 64 tiles (8 waves of 2 x 2 accumulators, three workgroups per CU, 25 % less staging
//   traffic per flop): 36 / 47 / 47 -- the staging traffic is not what bounds it; at K = 144 every shape is bound by the
//   read-modify-write of C, not by the MFMAs.  Workgroups per CU (LDS padded in the bench): 1 / 2 / 3 / 4 -> 27 / 40 / 46 / 48.6 TF/s at K = 432;
//   8-deep chunks x 2 stages (more workgroups, twice the barriers): 44.8; 32-deep x 2: 41.8; source descriptors by scalar loads instead of
//   the LDS copy (32 KB of LDS per workgroup exactly, a fifth workgroup per CU): 47.2
// one 16-deep chunk out of the LDS images sa ([k][TM rows]) / sb ([k][TN rows]) into the wave's accumulators; the operands of k-step
// kk + 1 are requested before the MFMAs of k-step kk are issued (the LDS round trip is off the MFMA chain)
template <int TM, int TN, int RM, int RN>
__device__ __forceinline__ void mt_chunk(d4 (&acc)[RM][RN], const double *sa, const double *sb, int g, int xo, int yo)
{
  double x[RM], y[RN];
#pragma unroll
  for (int i = 0; i < RM; ++i) x[i] = sa[g * TM + xo + 16 * i];
#pragma unroll
  for (int j = 0; j < RN; ++j) y[j] = sb[g * TN + yo + 16 * j];
#pragma unroll
  for (int kk = 0; kk < MKB / 4; ++kk) {
    double nx[RM], ny[RN];
#pragma unroll
    for (int i = 0; i < RM; ++i) nx[i] = 0.0;
#pragma unroll
    for (int j = 0; j < RN; ++j) ny[j] = 0.0;
    if (kk + 1 < MKB / 4) {
#pragma unroll
      for (int i = 0; i < RM; ++i) nx[i] = sa[(4 * (kk + 1) + g) * TM + xo + 16 * i];
#pragma unroll
      for (int j = 0; j < RN; ++j) ny[j] = sb[(4 * (kk + 1) + g) * TN + yo + 16 * j];
    }
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
      for (int j = 0; j < RN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(y[j], x[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < RM; ++i) x[i] = nx[i];
#pragma unroll
    for (int j = 0; j < RN; ++j) y[j] = ny[j];
  }
}
// Full tiles (the bulk of a large front) stage their operands by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave
// instruction = 128 consecutive doubles of the [k][rows] image, no registers) into a ring of MT_STAGES stages, MT_STAGES - 1 chunks in
// flight behind a counted vmcnt: a chunk's memory round trip is hidden behind the MFMAs of the chunks before it.
// The chunk sequence runs across the sources of the task (the ring is not drained between sources).  Edge tiles and the last
// partial chunk of a source (K not a multiple of 16) take the register-staged path (masked loads).
template <int TM, int TN, int WR, int WC>
__device__ __forceinline__ void update_mt_body(double *__restrict__ base, const chol_upd_task t, const chol_upd_src *__restrict__ srcs, int64_t arena_elems = 0)
{
  constexpr int NW = WR * WC, NT = 64 * NW, RM = TM / WR / 16, RN = TN / WC / 16;
  constexpr int QA = MKB * TM / 128 / NW, QB = MKB * TN / 128 / NW; // LDS-DMA instructions per wave, chunk and operand
  static_assert(QA * NW * 128 == MKB * TM && QB * NW * 128 == MKB * TN && NT % TM == 0 && NT % TN == 0, "tile shape");
  __shared__ double sA[MT_STAGES][MKB][TM];
  __shared__ double sB[MT_STAGES][MKB][TN];
  const int tt = threadIdx.x, lane = tt & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tt >> 6);
  const int r15 = lane & 15, g = lane >> 4;
  const int wr = wave % WR, wc = wave / WR;
  const int xo = (TM / WR) * wr + r15, yo = (TN / WC) * wc + r15;
  d4 acc[RM][RN];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j) acc[i][j] = (d4){ 0.0, 0.0, 0.0, 0.0 };
  // An edge tile (fewer than TM rows or TN columns: on the generated problems a quarter of the tiles, most of them a row or two short of
  // 64) takes the DMA path all the same when the rows it does not own are inside the arena (`arena_elems` doubles behind `base`): they
  // are other panels' data, their products land in accumulator rows / columns that the epilogue does not store
  bool full = t.mv == TM && t.nv == TN;
  if (!full && arena_elems > 0) {
    __shared__ int sOk;
    if (tt == 0) sOk = 1;
    lds_barrier(); // the initialisation has LANDED before any wave may clear the word (s_barrier alone is no LDS fence: gfx950 has back-off barriers)
    for (int s = t.src_begin + tt; s < t.src_end; s += NT) {
      const chol_upd_src sd = srcs[s];
      const int kf = (sd.k / MKB) * MKB;
      if (kf > 0 && (sd.a_off + t.ar + (TM - 1) + (int64_t)(kf - 1) * sd.lda >= arena_elems || sd.b_off + t.br + (TN - 1) + (int64_t)(kf - 1) * sd.ldb >= arena_elems)) sOk = 0;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    full = sOk != 0;
  }
  if (full) {
    // ---- LDS-DMA ring.  Chunk list = the full 16-deep chunks of every source, in order; (is, ik) = next chunk to issue.
    // Instruction q of an operand's chunk moves the doubles [128 q, 128 q + 128) of the [k][rows] image: lane l the pair at 128 q + 2 l.
    // The source descriptors are copied to LDS in batches first: inside the ring nothing may be read through the vector-memory
    // counter (a descriptor load and its vmcnt(0) would drain the ring every chunk).
    __shared__ chol_upd_src sS[MT_SRC_BATCH];
    for (int sb = t.src_begin; sb < t.src_end; sb += MT_SRC_BATCH) {
      const int ns = min(MT_SRC_BATCH, t.src_end - sb);
      __builtin_amdgcn_s_barrier(); // the previous batch is done with sS and with the ring
      if (tt < ns) sS[tt] = srcs[sb + tt];
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      int total = 0;
      for (int s = 0; s < ns; ++s) total += sS[s].k / MKB;
      // issue cursor: source `is`, `left` full chunks of it to go, the lane's addresses of the next chunk in registers (the descriptor in LDS is read
      // once per source, not once per chunk: its round trips sat in front of every chunk's MFMAs)
      int is = -1, left = 0, issued = 0;
      const double *pa[QA], *pb[QB];
      int64_t sta = 0, stb = 0;
#define MT_ISSUE()                                                                                                    \
      {                                                                                                               \
        while (left == 0) {                                                                                           \
          ++is;                                                                                                       \
          left = sS[is].k / MKB;                                                                                      \
          const int64_t ao_ = sS[is].a_off, bo_ = sS[is].b_off;                                                       \
          const int lda_ = sS[is].lda, ldb_ = sS[is].ldb;                                                             \
          _Pragma("unroll") for (int pp = 0; pp < QA; ++pp) {                                                         \
            const int e_ = 128 * (QA * wave + pp) + 2 * lane;                                                         \
            pa[pp] = base + ao_ + t.ar + e_ % TM + (int64_t)(e_ / TM) * lda_;                                         \
          }                                                                                                           \
          _Pragma("unroll") for (int pp = 0; pp < QB; ++pp) {                                                         \
            const int e_ = 128 * (QB * wave + pp) + 2 * lane;                                                         \
            pb[pp] = base + bo_ + t.br + e_ % TN + (int64_t)(e_ / TN) * ldb_;                                         \
          }                                                                                                           \
          sta = (int64_t)MKB * lda_; stb = (int64_t)MKB * ldb_;                                                       \
        }                                                                                                             \
        const int st_ = issued % MT_STAGES;                                                                           \
        _Pragma("unroll") for (int pp = 0; pp < QA; ++pp) { MT_DIAG_DMA(lds_dma16(pa[pp], &sA[st_][0][0] + 128 * (QA * wave + pp))); pa[pp] += sta; } \
        _Pragma("unroll") for (int pp = 0; pp < QB; ++pp) { MT_DIAG_DMA(lds_dma16(pb[pp], &sB[st_][0][0] + 128 * (QB * wave + pp))); pb[pp] += stb; } \
        --left; ++issued;                                                                                             \
      }
      for (int i = 0; i < MT_STAGES - 1 && issued < total; ++i) MT_ISSUE();
      for (int c = 0; c < total; ++c) {
        // this wave's DMA instructions of chunk c have landed when at most the later chunks' remain outstanding
        if (issued - c - 1 >= MT_STAGES - 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((QA + QB) * (MT_STAGES - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MT_DIAG_BAR(__builtin_amdgcn_s_barrier()); // every wave's part of chunk c is in LDS; every wave has left chunk c - 1
        if (issued < total) MT_ISSUE(); // into the stage chunk c - 1 occupied
        mt_chunk<TM, TN, RM, RN>(acc, &sA[c % MT_STAGES][0][0], &sB[c % MT_STAGES][0][0], g, xo, yo);
      }
#undef MT_ISSUE
    }
    __builtin_amdgcn_s_barrier(); // the tail path below re-uses stages 0 and 1
  }
  // ---- register-staged path: everything for edge tiles, the K tails (K mod 16 columns) of the sources for full tiles
  {
    constexpr int KA = NT / TM, KB = NT / TN; // k-columns the threads cover per pass of the A / B image
    const int arow = tt % TM, akq = tt / TM, brow = tt % TN, bkq = tt / TN;
    const bool sva = arow < t.mv, svb = brow < t.nv;
    int buf = 0;
    for (int s = t.src_begin; s < t.src_end; ++s) {
      const chol_upd_src sd = srcs[s];
      const int K = sd.k, kbeg = full ? (K / MKB) * MKB : 0;
      if (kbeg >= K) continue;
      const double *A = base + sd.a_off + t.ar + arow;
      const double *Bp = base + sd.b_off + t.br + brow;
      double ra[MKB / KA], rb[MKB / KB];
#pragma unroll
      for (int i = 0; i < MKB / KA; ++i) { const int k = kbeg + akq + KA * i; ra[i] = (sva && k < K) ? A[(int64_t)k * sd.lda] : 0.0; } // first chunk of this source
#pragma unroll
      for (int i = 0; i < MKB / KB; ++i) { const int k = kbeg + bkq + KB * i; rb[i] = (svb && k < K) ? Bp[(int64_t)k * sd.ldb] : 0.0; }
      for (int k0 = kbeg; k0 < K; k0 += MKB) {
#pragma unroll
        for (int i = 0; i < MKB / KA; ++i) sA[buf][akq + KA * i][arow] = ra[i];
#pragma unroll
        for (int i = 0; i < MKB / KB; ++i) sB[buf][bkq + KB * i][brow] = rb[i];
        if (k0 + MKB < K) { // next chunk's loads fly during this chunk's MFMAs
#pragma unroll
          for (int i = 0; i < MKB / KA; ++i) { const int k = k0 + MKB + akq + KA * i; ra[i] = (sva && k < K) ? A[(int64_t)k * sd.lda] : 0.0; }
#pragma unroll
          for (int i = 0; i < MKB / KB; ++i) { const int k = k0 + MKB + bkq + KB * i; rb[i] = (svb && k < K) ? Bp[(int64_t)k * sd.ldb] : 0.0; }
        }
        lds_barrier(); // chunk visible; the other buffer is free again (everyone is past its reads)
        mt_chunk<TM, TN, RM, RN>(acc, &sA[buf][0][0], &sB[buf][0][0], g, xo, yo);
        buf ^= 1;
      }
    }
  }
  // epilogue: all C values of the lane are requested before the first is used (clamped addresses, masked
  // stores); as read-modify-writes in a row each one would wait for its own memory round trip
  double cv[RM][RN][4];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = min((TM / WR) * wr + 16 * i + r15, t.mv - 1), c = min((TN / WC) * wc + 16 * j + g + 4 * q, t.nv - 1);
        cv[i][j][q] = *(const volatile double *)(base + t.c_off + r + (int64_t)c * t.ldc);
      }
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j) {
      const int r = (TM / WR) * wr + 16 * i + r15;
      double *C = base + t.c_off + r;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = (TN / WC) * wc + 16 * j + g + 4 * q;
        if (r < t.mv && c < t.nv && (!t.lower || t.ar + r >= t.br + c)) C[(int64_t)c * t.ldc] = cv[i][j][q] - acc[i][j][q]; // lower: a SYRK tile on the diagonal (ar, br: its origin in the target's own rows)
      }
    }
}
#ifdef MT_CLOCK /* timing diagnostic of scripts/mt_bench.hip: shader cycles (s_memtime) and 100 MHz ticks (s_memrealtime) one workgroup in the middle of the grid spends on its tile */
__device__ unsigned long long g_mt_clock[2];
#endif
__global__ __launch_bounds__(256) void k_update_mt(double *__restrict__ base, const chol_upd_task *__restrict__ tasks,
                                                   const chol_upd_src *__restrict__ srcs, int ntask, int per_xcd, int64_t arena_elems)
{
  const int tid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if ((blockIdx.x >> 3) >= per_xcd || tid >= ntask) return;
#ifdef MT_CLOCK
  const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  update_mt_body<MT, MT, 2, 2>(base, tasks[tid], srcs, arena_elems);
#ifdef MT_CLOCK
  if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) { g_mt_clock[0] = __builtin_amdgcn_s_memtime() - clk_t0; g_mt_clock[1] = __builtin_amdgcn_s_memrealtime() - clk_r0; }
#endif
}

// ================================================================================================
// Dense pivot kernels.  Diagonal blocks are TS = 16 wide (one fp64 MFMA tile).
//
// Triangular solves X = T L^-T against a 16x16 diagonal block use the explicit inverse of the block:
// X = T Linv^T is four accumulating MFMAs (solve16).  The register-resident POTRF gets Linv for free (the
// rows of the identity ride through its row-per-lane Cholesky in otherwise idle lanes); the workspace holds
// it per diagonal block as W[blk * 256 + k * 16 + c] = Linv(c, k), the layout solve16 reads.  The kernels
// that do not run that Cholesky (k_potrf_big, k_dinv for the BLAS-level TRSM) cut L into 4x4 blocks instead:
// the four 4x4 diagonal blocks are inverted side by side in the quads of a wave (linv4_quad, "Ydiag"), the
// solve is 7 dependent MFMAs (tile_solve), and Linv = tile_solve(I) (store_linv16).
//
// Register-resident design for pivots up to CHOL_RR_MAXN = 272 (17 tiles): the MI355X register
// file (512 KB per CU) is the only on-chip memory that holds a 259 x 259 fp64 lower triangle
// (269 KB; LDS has 160 KB), so the trailing matrix lives in VGPRs as 16x16 tiles in MFMA
// accumulator layout, spread round-robin over 11 "tile" waves of a 768-thread workgroup (3 waves per SIMD = 168 registers each); wave 0
// is the "factor" wave (Cholesky of the diagonal tile).  An accumulator tile is directly a valid
// "X" operand of the next MFMA (register q of lane l holds column (l >> 4) + 4 q = k-step q of the
// operand map), so neither the panel solve nor the trailing update T -= P_i P_j^T moves T; P goes
// through LDS.  Look-ahead: the owner of tile (k+1, k+1) updates and hands it to the factor wave
// first, so the factor wave works on step k+1 while the tile waves finish the updates of step k.
// ================================================================================================
#define TS 16
#define RR_MAXT CHOL_RR_MAXT
#define RR_NHEAVY CHOL_RR_NHEAVY /* tile waves that do not share a SIMD with the factor wave */
// In-kernel cycle stamps of the factor wave: diagnostic builds only (-DCHOL_STAMPS, scripts/stamp_potrf.hip)
#ifdef CHOL_STAMPS
__device__ unsigned long long g_stamps[16];
#define STAMP_DECL unsigned long long st_[8], acc_[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_[i]) :: "memory"); __builtin_amdgcn_sched_barrier(0); if ((i) > 0) acc_[i] += st_[i] - st_[(i) - 1]; } while (0)
__device__ unsigned long long g_trace[12][20][8]; // [wave][step][stamp]: absolute times, k_potrf_rr only
#define STAMPK(i) do { STAMP(i); if (lane == 0 && k < 20) g_trace[wave][k][i] = st_[i]; } while (0)
__device__ unsigned long long g_walk[12][20][8]; // [wave][step][point]: absolute times inside the tile waves' update walk
#define STAMPX(i) do { unsigned long long x_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x_) :: "memory"); __builtin_amdgcn_sched_barrier(0); if (lane == 0 && k < 20) g_walk[wave][k][i] = x_; } while (0)
#define STAMP_FLUSH do { if (lane == 0) for (int i_ = 0; i_ < 8; ++i_) g_stamps[i_] = acc_[i_]; } while (0)
#define STAMP_FLUSH2 do { if (lane == 0) for (int i_ = 0; i_ < 8; ++i_) g_stamps[8 + i_] = acc_[i_]; } while (0)
#ifndef STAMP_WAVE
#define STAMP_WAVE 0
#endif
#else
#define STAMP_WAVE 0
#define STAMP_DECL
#define STAMP(i)
#define STAMPK(i)
#define STAMPX(i)
#define STAMP_FLUSH
#define STAMP_FLUSH2
#endif

// -DCHOL_POLLS (scripts/stamp_potrf.hip): counts instead of clock stamps -- how often the factor wave found its look-ahead tiles late and how
// many polls that cost, how often it waited for the tile waves' counter; no s_memtime round trips, so the waves keep their pace
#ifdef CHOL_POLLS
__device__ unsigned long long g_polls[8];
#define POLL_DECL unsigned long long pc_[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }
#define POLL_ADD(i, v) pc_[i] += (v)
#define POLL_FLUSH do { if (lane == 0) for (int i_ = 0; i_ < 8; ++i_) g_polls[i_] = pc_[i_]; } while (0)
__device__ unsigned long long g_wpolls[12][8]; // tile waves: poll iterations (a 64-cycle sleep + an LDS round trip each) at fL, cRaw, the step's barrier, fP
#define POLL_WAIT(i, flag, target) do { while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < (target)) { pc_[i] += 1; __builtin_amdgcn_s_sleep(1); } asm volatile("" ::: "memory"); } while (0)
#define POLL_WFLUSH do { if (lane == 0) for (int i_ = 0; i_ < 8; ++i_) g_wpolls[wave][i_] = pc_[i_]; } while (0)
#else
#define POLL_WAIT(i, flag, target) lds_wait_ge(flag, target)
#define POLL_WFLUSH
#define POLL_DECL
#define POLL_ADD(i, v)
#define POLL_FLUSH
#endif

__device__ __forceinline__ double readlane_f64(double v, int l)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// broadcast of quad-lane I inside every quad (4 consecutive lanes), DPP quad_perm
template <int I> __device__ __forceinline__ double quad_bcast(double v)
{
  constexpr int ctrl = I | (I << 2) | (I << 4) | (I << 6);
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// 1/sqrt(d) to fp64 accuracy.  Measured on gfx950: the v_rsq_f64 seed is good to 5.2e-8, one Newton step
// leaves 4e-15 (36 ulp), two reach 1.4e-16.  A lone wave issues one DP instruction per ~11 cycles
// (scripts/dp_lat.hip), so the pivot chain is bound by its DP instruction COUNT: one third-order step
//   e = 1 - d y^2,  y <- y (1 + e/2 + 3 e^2/8)        (error ~ e^3 = 1e-22)
// costs 5 DP instructions against 7 for two Newton steps.
__device__ __forceinline__ double rsqrt_nr(double d)
{
  const double y = __builtin_amdgcn_rsq(d);
  const double e = fma(-(d * y), y, 1.0);
  const double q = e * fma(0.375, e, 0.5);
  return fma(y, q, y);
}

// acc += m[lane K of this lane's 16-lane row] * nt: the multiplier of a column step straight out of the lane that holds it (DPP
// row_newbcast, the one DPP form double-precision instructions take).  hipcc pads no hazards inside an asm statement: a VGPR written
// by the VALU needs two wait states before a DPP instruction reads it as its shuffled operand -- the first use of m in a column step
// carries them (m may have been produced by a compiler-inserted copy just ahead of the statement)
template <int K, bool FIRST> __device__ __forceinline__ void fmac_bcast(double &acc, double m, double nt)
{
  if (FIRST) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(m), "v"(nt), "n"(K));
  else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(m), "v"(nt), "n"(K));
}
// one column step of chol16_rows
template <int J> __device__ __forceinline__ void chol16_col(double (&a)[TS], double (&dd)[TS], double &myinv, int r15, int addr)
{
  const double d = readlane_f64(a[J], J);
  dd[J] = d;
  // the unscaled column J of the tile into EVERY 16-lane row (lane l reads lane l & 15 through the LDS crossbar, no memory): the passenger
  // rows take their multipliers from it like the tile's own rows.  Issued ahead of the rsqrt chain, which hides its round trip
  const int mlo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(a[J])), mhi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(a[J]));
  const double m = __hiloint2double(mhi, mlo);
  const double rv = rsqrt_nr(d); // the same value in every lane: stays in a vector register
  myinv = (r15 == J) ? rv : myinv;
  a[J] = a[J] * rv; // L(row, J); row J: d / sqrt(d)
  const double nt = -(a[J] * rv);
#define CHOL16_FM(K_) if constexpr (K_ > J && K_ < TS) fmac_bcast<K_, K_ == J + 1>(a[K_], m, nt);
  CHOL16_FM(1) CHOL16_FM(2) CHOL16_FM(3) CHOL16_FM(4) CHOL16_FM(5) CHOL16_FM(6) CHOL16_FM(7) CHOL16_FM(8)
  CHOL16_FM(9) CHOL16_FM(10) CHOL16_FM(11) CHOL16_FM(12) CHOL16_FM(13) CHOL16_FM(14) CHOL16_FM(15)
#undef CHOL16_FM
  __builtin_amdgcn_sched_barrier(0); // one column at a time
}
// 16x16 lower Cholesky, one row per lane (row = lane & 15 for the tile's rows; a 16-lane group may carry other rows -- the identity, a
// panel tile -- through the same column operations: they come out multiplied by L^-T).  a[c] = A(row, c) on entry (c <= row used),
// L(row, c) on exit.  myinv = 1 / L(row, row).  Returns the first non-positive pivot (1-based) or 0.
// Round 4: the column multipliers L(k, j) used to travel through scalar registers (two v_readlane per multiplier and column, 300 per tile);
// now every 16-lane row gets the tile's column by one ds_bpermute pair per column and the multiply-adds read their multiplier from lane
// k of their own row (DPP).  Same products, same sums, bit-identical results; scripts/chol16_bench.hip: 2396 -> 1800 ticks per tile.
// `mid` runs ahead of column CHOL16_HOOK_COL: the caller's LDS reads issued there complete under the rest of the factorisation.
#ifndef CHOL16_HOOK_COL
#define CHOL16_HOOK_COL 14
#endif
struct chol16_no_hook { __device__ __forceinline__ void operator()() const {} };
template <int J, class H> __device__ __forceinline__ void chol16_col_h(double (&a)[TS], double (&dd)[TS], double &myinv, int r15, int addr, H &mid)
{
  if constexpr (J == CHOL16_HOOK_COL) { mid(); __builtin_amdgcn_sched_barrier(0); }
  chol16_col<J>(a, dd, myinv, r15, addr);
}
template <class H = chol16_no_hook>
__device__ __forceinline__ int chol16_rows(double (&a)[TS], double &myinv, int r15, H &&mid = H())
{
  myinv = 0.0;
  double dd[TS]; // the pivots (wave uniform: SGPRs)
  const int addr = r15 << 2;
  chol16_col_h<0>(a, dd, myinv, r15, addr, mid); chol16_col_h<1>(a, dd, myinv, r15, addr, mid); chol16_col_h<2>(a, dd, myinv, r15, addr, mid); chol16_col_h<3>(a, dd, myinv, r15, addr, mid);
  chol16_col_h<4>(a, dd, myinv, r15, addr, mid); chol16_col_h<5>(a, dd, myinv, r15, addr, mid); chol16_col_h<6>(a, dd, myinv, r15, addr, mid); chol16_col_h<7>(a, dd, myinv, r15, addr, mid);
  chol16_col_h<8>(a, dd, myinv, r15, addr, mid); chol16_col_h<9>(a, dd, myinv, r15, addr, mid); chol16_col_h<10>(a, dd, myinv, r15, addr, mid); chol16_col_h<11>(a, dd, myinv, r15, addr, mid);
  chol16_col_h<12>(a, dd, myinv, r15, addr, mid); chol16_col_h<13>(a, dd, myinv, r15, addr, mid); chol16_col_h<14>(a, dd, myinv, r15, addr, mid); chol16_col_h<15>(a, dd, myinv, r15, addr, mid);
  // A pivot that is not positive (or NaN) turns its column, and through the rank-1 update every later column of every row, into
  // NaN: the LAST pivot tells whether any failed, and only then is the first one looked for (sixteen compare / select rounds
  // on the scalar unit otherwise sit on the pivot chain of every step)
  int bad = 0;
  if (!(dd[TS - 1] > 0.0)) {
#pragma unroll
    for (int j = 0; j < TS; ++j)
      if (!(dd[j] > 0.0) && bad == 0) bad = j + 1;
  }
  return bad;
}

// Inverses of the four 4x4 diagonal blocks, one block per quad.  Lane r15 = 4 b + i holds
// blk[k] = L(4b+i, 4b+k) and invd = 1 / L(4b+i, 4b+i).  On exit x[m] = Linv_bb(m, i): the lane owns
// column i of its block's inverse.
__device__ __forceinline__ void linv4_quad(const double (&blk)[4], double invd, double (&x)[4], int qi)
{
  const double inv0 = quad_bcast<0>(invd), inv1 = quad_bcast<1>(invd), inv2 = quad_bcast<2>(invd), inv3 = quad_bcast<3>(invd);
  const double l10 = quad_bcast<1>(blk[0]);
  const double l20 = quad_bcast<2>(blk[0]), l21 = quad_bcast<2>(blk[1]);
  const double l30 = quad_bcast<3>(blk[0]), l31 = quad_bcast<3>(blk[1]), l32 = quad_bcast<3>(blk[2]);
  x[0] = (qi == 0) ? inv0 : 0.0;
  x[1] = (qi == 1) ? inv1 : -(l10 * x[0]) * inv1;
  x[2] = (qi == 2) ? inv2 : -fma(l21, x[1], l20 * x[0]) * inv2;
  x[3] = (qi == 3) ? inv3 : -fma(l32, x[2], fma(l31, x[1], l30 * x[0])) * inv3;
}

// X = T L^-T for a 16x16 tile T in accumulator layout.  Lr[b] = L(r15, 4b + g), b = 0..2 (register b
// of the diagonal block in tile layout), yd = Ydiag(c = r15, k = g).
__device__ __forceinline__ d4 tile_solve(d4 t, const double (&Lr)[3], double yd)
{
  const d4 z = { 0.0, 0.0, 0.0, 0.0 };
  d4 x, s, u;
  s = __builtin_amdgcn_mfma_f64_16x16x4f64(yd, t[0], z, 0, 0, 0);
  x[0] = s[0];
  u = __builtin_amdgcn_mfma_f64_16x16x4f64(Lr[0], -x[0], t, 0, 0, 0);
  s = __builtin_amdgcn_mfma_f64_16x16x4f64(yd, u[1], z, 0, 0, 0);
  x[1] = s[1];
  u = __builtin_amdgcn_mfma_f64_16x16x4f64(Lr[1], -x[1], u, 0, 0, 0);
  s = __builtin_amdgcn_mfma_f64_16x16x4f64(yd, u[2], z, 0, 0, 0);
  x[2] = s[2];
  u = __builtin_amdgcn_mfma_f64_16x16x4f64(Lr[2], -x[2], u, 0, 0, 0);
  s = __builtin_amdgcn_mfma_f64_16x16x4f64(yd, u[3], z, 0, 0, 0);
  x[3] = s[3];
  return x;
}

// Linv(k,k) for the TRSM kernels, off every critical path: X = I L^-T = Linv^T by one tile_solve, stored
// as Wb[k * 16 + c] = Linv(c, k) -- the layout solve16() reads as its MFMA Y operand.
__device__ __forceinline__ void store_linv16(double *Wb, const double (&Lr)[3], double yd, int r15, int g)
{
  d4 id;
  int rr = r15;
  asm volatile("" : "+v"(rr)); // built where it is used: a hoisted identity tile costs 8 registers in the callers' loops
#pragma unroll
  for (int q = 0; q < 4; ++q) id[q] = (rr == g + 4 * q) ? 1.0 : 0.0;
  const d4 xi = tile_solve(id, Lr, yd); // xi(r, c) = Linv(c, r)
#pragma unroll
  for (int q = 0; q < 4; ++q) Wb[r15 * TS + g + 4 * q] = xi[q];
}
// X = T Linv^T with the explicit inverse: four accumulator-dependent MFMAs (256 cycles) instead of the
// seven operand-dependent ones of tile_solve (~700); w[st] = Wb[(4 st + g) * 16 + r15]
__device__ __forceinline__ d4 solve16(d4 t, const double (&w)[4])
{
  d4 x = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
  for (int st = 0; st < 4; ++st) x = __builtin_amdgcn_mfma_f64_16x16x4f64(w[st], t[st], x, 0, 0, 0);
  return x;
}

// the lane's own 4x4 diagonal-block row out of its 16-entry row: blk[k] = a[4 (r15 / 4) + k]
__device__ __forceinline__ void own_block_row(const double (&a)[TS], int r15, double (&blk)[4])
{
  const int b = r15 >> 2;
  // every stage is laundered through an empty asm statement: otherwise the select chain is turned into a
  // dynamically indexed load of a[], which puts the whole row array (and its stores in chol16_rows) in scratch
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    double v = a[k];
    asm volatile("" : "+v"(v));
    v = (b == 1) ? a[4 + k] : v;
    asm volatile("" : "+v"(v));
    v = (b == 2) ? a[8 + k] : v;
    asm volatile("" : "+v"(v));
    v = (b == 3) ? a[12 + k] : v;
    blk[k] = v;
  }
}

// Owner of a tile: chol_rr_owner (chol_plan.h, shared with the host, which builds the role tables with the schedule)
#define RR_HEAVY_ONLY CHOL_RR_HEAVY_ONLY
__device__ __forceinline__ void rr_owner(int idx, int ntl2, int &w, int &slot) { chol_rr_owner(idx, ntl2, &w, &slot); }
// position of a tile wave among the heavy ones (-1 for the two light waves)
__device__ __forceinline__ int rr_heavy_index(int w) { return (w & 3) == 3 ? -1 : w - (w >> 2); }
// number of indices in [0, cnt) that rr_owner() gives to tile wave w (hw = rr_heavy_index(w)): the wave's
// slots [0, rr_count) hold exactly those tiles
__device__ __forceinline__ int rr_count(int cnt, int ntl2, int w, int hw)
{
  if (ntl2 <= RR_HEAVY_ONLY) return hw >= 0 ? cnt / RR_NHEAVY + (cnt % RR_NHEAVY > hw) : 0;
  const int cyc = cnt / 51, rem = cnt % 51;
  if (hw >= 0) return cyc * 5 + (rem > hw) + (rem > 11 + hw) + (rem > 20 + hw) + (rem > 31 + hw) + (rem > 40 + hw);
  const int o = w == 3 ? 9 : 10;
  return cyc * 3 + (rem > o) + (rem > 20 + o) + (rem > 40 + o);
}
// two independent tile_solve chains interleaved: the 7 operand-dependent MFMAs of one hide in the latency of
// the other.  Each solved register goes to LDS (l0/l1, accumulator layout, + 64 per register) and to global
// memory (g0/g1, + 4 columns per register; nullptr = row past n) as soon as it exists, so it is not kept live.
__device__ __forceinline__ void tile_solve2(d4 u0, d4 u1, const double (&Lr)[3], double yd, double *l0, double *l1,
                                            double *g0, double *g1, int64_t lda4)
{
  const d4 z = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const d4 s0 = __builtin_amdgcn_mfma_f64_16x16x4f64(yd, u0[b], z, 0, 0, 0);
    const d4 s1 = __builtin_amdgcn_mfma_f64_16x16x4f64(yd, u1[b], z, 0, 0, 0);
    const double x0 = s0[b], x1 = s1[b];
    if (b < 3) {
      u0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Lr[b], -x0, u0, 0, 0, 0);
      u1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Lr[b], -x1, u1, 0, 0, 0);
    }
    l0[b * 64] = x0;
    l1[b * 64] = x1;
    if (g0) g0[b * lda4] = x0;
    if (g1) g1[b * lda4] = x1;
  }
}

__device__ __forceinline__ void tile_of_index(int idx, int T, int &ti, int &tj) { chol_rr_tile_of_index(idx, T, &ti, &tj); }

// ------------------------------------------------------------------------------------------------
// POTRF role (potrf_rr_body; k_potrf_rr, k_potrf_trsm): register-resident, one 768-thread workgroup per
// pivot block (n <= CHOL_RR_MAXN).
//
// Roles.  Wave 0 = factor wave, waves 1..11 = tile waves.  Columns 0 and 1 of the 16x16 tile grid are parked
// in LDS by the prologue; the tiles of columns >= 2 are dealt to the tile waves in reverse column-major
// order (rr_owner) and stay in registers (accumulator layout) while they still receive updates.
//
// Step k (block column k):
//   factor wave   a. Cholesky of the diagonal tile (k,k), one row per lane; lanes 16-31 carry the rows of
//                    the identity and come out as L(k,k)^-T.  Publishes both in LDS (sLW)     [flag fL]
//                 b. waits for the two look-ahead tiles of step k-1's trailing update       [flags fA, fD]
//                 c. solves the sub-diagonal tile (k+1,k) itself (4 MFMAs), publishes it      [flag fP]
//                 d. applies it to the diagonal tile (k+1,k+1) it already holds (4 MFMAs) and goes
//                    straight to step k+1; it re-uses an LDS buffer only after every tile wave has left
//                    the step that read it (cUpd, two steps back)
//   tile waves    1. wait for fL and for the raw tiles of column k (counter cRaw); the heavy waves solve
//                    the panel tiles (i,k), i >= k+2, out of LDS with the explicit inverse (two tiles
//                    interleaved, tile i -> heavy wave i mod 9)
//                 2. software barrier (LDS counter cSol) + fP
//                 3. trailing update of their live register tiles, next panel column first; a tile that
//                    just got its last update is parked: column k+1 -> sRaw (+cRaw, fA for the look-ahead
//                    tile), diagonal (k+2,k+2) -> sDg (fD)
//                 4. one light wave copies L(k,k), L(k,k)^-T and L(k+1,k) from LDS to global memory;
//                    cUpd += 1 (fused launch: after the wave's stores have completed; the last wave out
//                    publishes the column to the TRSM workgroups)
// All hand-offs are LDS words polled relaxed behind `s_waitcnt lgkmcnt(0)`; no s_barrier inside the loop
// (the factor wave would have to take part in it), nothing that drains vmcnt on the critical chain.
// ------------------------------------------------------------------------------------------------
#define RR_NW CHOL_RR_NW       /* tile waves */
#define RR_SLOTS CHOL_RR_SLOTS /* most tiles a tile wave owns: columns >= 2 of a 17 x 17 tile grid, 120 tiles (see rr_owner) */
#define RR_RSLOTS 11 /* ... of which live in registers; slot 11 (n > 256 only, columns 2-3: dead after step 2) lives in LDS */
#define RR_THREADS ((RR_NW + 1) * 64)
/* LDS image of the POTRF role, in doubles */
#define RR_OFF_SOL (RR_MAXT * TS * TS)
#define RR_OFF_DG (RR_OFF_SOL + 2 * RR_MAXT * TS * TS)
#define RR_OFF_LW (RR_OFF_DG + 2 * TS * TS)
#define RR_OFF_CONV (RR_OFF_LW + 2 * 4 * TS * (TS + 1))
#define RR_OFF_OV (RR_OFF_CONV + 2 * TS * (TS + 1))
#define RR_OFF_IJ (RR_OFF_OV + RR_NHEAVY * TS * TS)
#define RR_OFF_FLAG (RR_OFF_IJ + (RR_SLOTS * RR_NW + 16 + 3) / 4)
#define RR_OFF_KM (RR_OFF_FLAG + 5)                       /* per (slot, wave): first step whose update can be non-zero (skyline) */
#define RR_OFF_MASK (RR_OFF_KM + (RR_SLOTS * RR_NW + 7) / 8) /* per (step, wave): the slots that have work in the step, the panel tiles to solve */
#define RR_OFF_SKY (RR_OFF_MASK + (RR_MAXT * RR_NW + 1) / 2)  /* the block's tile-level skyline (24 bytes) */
#define RR_SMEM_DOUBLES (RR_OFF_SKY + 3)
#define RR_M_SOLVE0 CHOL_RR_M_SOLVE0 /* step mask: bit s < 12 = slot s; bits 12, 13 = the wave's first / second panel tile of the step is not structurally zero */
#define RR_M_SOLVE1 CHOL_RR_M_SOLVE1
// Note (measured): waves of a workgroup are dealt to the four SIMDs round-robin, so waves 0, 4 and 8
// share a SIMD, and fp64 MFMA runs on the same DP units as fp64 VALU: the tile waves' 64-cycle MFMAs
// on the factor wave's SIMD stretch its scalar chain (chol16 3.4k -> 4.9k cycles).  Leaving waves 4 and
// 8 without tiles fixes that (3.7k) but 9 tile waves need 17 register slots for n > 256 and the trailing
// update becomes the bottleneck on 3 SIMDs; 11 tile waves are faster end to end (79 vs 88 us at n = 259).

// LDS-only synchronisation.  Everything the waves of these kernels hand to each other goes through
// LDS, so the hand-offs must NOT use __syncthreads() / workgroup-scope fences: those also drain the
// vector-memory counter, i.e. every step would wait for the global stores of finished tiles and for
// the prefetch loads to land (measured: ~2k of the 3.2k cycles of a TRSM step).
__device__ __forceinline__ void lds_wait_ge(int *flag, int target)
{
  while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target)
    __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory"); // later LDS reads stay behind the poll (LDS executes a wave's accesses in order)
}
// the same, returning the value read
__device__ __forceinline__ int lds_wait_ge_v(int *flag, int target)
{
  int v;
  while ((v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) < target)
    __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
  return v;
}
// two adjacent flags (8-byte aligned pair) against the same target in one LDS read
__device__ __forceinline__ void lds_wait_both_ge(int *pair, int target)
{
  for (;;) {
    const long long v = __hip_atomic_load((long long *)pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const int lo = __builtin_amdgcn_readfirstlane((int)v), hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
    if (lo >= target && hi >= target) break;
    __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void lds_set(int *flag, int value, int lane)
{
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this wave's LDS writes have landed
  if (lane == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_inc(int *cnt, int lane)
{
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// tile (ti, tj) of the lower triangle of the n x n matrix A in accumulator layout; the last partial diagonal
// tile is padded with the identity, everything else past n and above the diagonal with zeros
template <bool PUB> __device__ __forceinline__ d4 load_tile(const double *A, int lda, int n, int ti, int tj, int r15, int g)
{
  const int row = ti * TS + r15;
  const bool rowok = row < n;
  const double *src = A + row + (int64_t)(tj * TS + g) * lda;
  d4 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = g + 4 * q, col = tj * TS + c;
    double e = (row == col) ? 1.0 : 0.0;
    if (rowok && col < n) e = (ti > tj || c <= r15) ? gload<PUB>(&src[(int64_t)(4 * q) * lda]) : 0.0;
    v[q] = e;
  }
  return v;
}
// ---- follower: external panel steps (program launch).  A pivot block that follows its children (or the previous column
// block of its own pivot) starts from ZERO accumulators and, before it reads its own tiles, applies the contributions
//   T(i, j) -= E_i E_j^T,   E = (the block's rows) x (16 columns of a source pivot block)
// as the source's TRSM strips publish them, column tile by column tile (chol_ext: where the rows are, how many columns,
// the per-column counters the strips raise once their stores have completed).  By the time the last source column is in,
// the diagonal block's update is complete and the block's own tiles are added on top by the prologue: no update launch,
// no kernel boundary and no wait for the slowest pivot of the level between a pivot and its parent.  Sources are taken
// in the order of the list (deterministic summation order).
// A follower's pivot block has at most CHOL_FOLLOW_MAXT = 10 column tiles (the schedule's blocks have 9): its register tiles are
// the slots [0, FOLLOW_SLOTS) of rr_owner's deal, an external panel is FOLLOW_LOADS doubles per thread.
#define FOLLOW_SLOTS 4
#define FOLLOW_LOADS ((CHOL_FOLLOW_MAXT * TS * TS + RR_THREADS - 1) / RR_THREADS)
template <int SLEEP = 0>
__device__ __forceinline__ void wait_list(const chol_wait *__restrict__ wl, int n, const int *ctr, const int *__restrict__ ctr_total, int epoch, int lane, int *info)
{ // SLEEP: poll interval in units of 64 cycles (0: WAIT_SLEEP, the interval of a blocked job)
  for (int b0 = 0; b0 < n; b0 += 64) {
    int c = 0, need = 0;
    const bool mine = b0 + lane < n;
    if (mine) { const chol_wait wt = wl[b0 + lane]; c = wt.ctr; need = wt.value + epoch * ctr_total[wt.ctr]; }
    for (int it = 0;; ++it) {
      const bool ok = !mine || __hip_atomic_load(&ctr[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need;
      if (!__ballot(!ok)) break;
      if (it >= (1 << 21)) { if (lane == 0) report_stall(info); break; } // x ~1 us per poll
#ifndef WAIT_SLEEP
#define WAIT_SLEEP 32
#endif
      __builtin_amdgcn_s_sleep(SLEEP ? SLEEP : WAIT_SLEEP); // ~1 us between polls: a blocked job is not on anybody's critical path by less than that
    }
  }
}
// one look at a wait list: true if every entry holds (n <= 64)
__device__ __forceinline__ bool wait_list_holds(const chol_wait *__restrict__ wl, int n, const int *ctr, const int *__restrict__ ctr_total, int epoch, int lane)
{
  bool ok = true;
  for (int b0 = 0; b0 < n; b0 += 64)
    if (b0 + lane < n) { const chol_wait wt = wl[b0 + lane]; ok = ok && __hip_atomic_load(&ctr[wt.ctr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= wt.value + epoch * ctr_total[wt.ctr]; }
  return !__ballot(!ok);
}
struct follow_args { const chol_ext *ext; int n_ext; const int *ctr; const int *ctr_total; int epoch; const chol_wait *wl; int n_wl; unsigned long long *stamp, *xstamp; };
// followed column tiles from item `from` on whose strips have all published: every lane polls one item's counter (one round trip)
__device__ __forceinline__ int ext_ready(const follow_args &f, int from, int lane)
{
  bool ok = true;
  if (from + lane < f.n_ext) {
    const int c = f.ext[from + lane].ctr;
    ok = __hip_atomic_load(&f.ctr[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - f.epoch * f.ctr_total[c] >= f.ext[from + lane].need;
  }
  const unsigned long long notok = __ballot(!ok);
  const int lead = notok ? __builtin_ctzll(notok) : 64;
  return min(from + lead, f.n_ext);
}
template <bool UPD, bool PUB, bool TR = false>
__device__ __forceinline__ void follow_external(const double *__restrict__ base, const follow_args f, int T, int n, double *sE, d4 (&tile)[11], d4 (&stage)[3],
                                                const int (&ijp)[12], int w, int lane, int tid, int *info, const double *A, int lda, int r15, int g, int *sReady)
{ // sReady: LDS word (0 at entry): leading items the factor wave has seen published
  const int lp = lane; // accumulator layout: register q of lane l at q * 64 + l
  int buf = 0;
  // Column tiles are consumed in the order of the list, TWO per round where two fit an LDS buffer (T <= 8) -- a column tile of a wide
  // follower is 16 KB handed over by other workgroups, and one memory round trip per tile is what bounds a follower that has fallen
  // behind its sources -- except the last two, which go one by one (the follower's start hangs on them).  The grouping is a function
  // of the list alone: every wave takes the same rounds (they poll for themselves), and a pair is the same sequence of MFMAs on the
  // same accumulators as two single rounds.  `ready` = leading items known to be published; the poll for more is issued before the
  // barrier and update of the round at hand and read after them, and the loads of the next round are issued before the update
  // whenever all of it is known to be ready.
  int ready = 0, ng = 0; // ng: items of the round starting at i whose loads are in flight (0 = not issued)
  double va[FOLLOW_LOADS], vb[FOLLOW_LOADS];
#ifndef FOLLOW_OWN_SLEEP
#define FOLLOW_OWN_SLEEP 4
#endif
#ifndef FOLLOW_OWN_LAST_MAX
#define FOLLOW_OWN_LAST_MAX 0 /* followers with at most this many items add their own tiles after the last one (3: lapl_3375 unchanged, lapl_400 45.8 -> 48.9 us) */
#endif
  static_assert(CHOL_FOLLOW_PAIR_MAXT == RR_MAXT, "a pair of followed column tiles shares one LDS buffer of RR_MAXT tiles");
  // where the follower's own tiles go in: before the last FOLLOW_OWN_BEFORE items (their round trip hides behind the wait for those) -- or,
  // for a follower with one source and a tail only (the next column block of a split pivot: its early jobs hang on the same strips as
  // its tail, they end after the tail has arrived), after the last item
  // (rounds and the own tiles' place: chol_follow_round / chol_follow_own_at in chol_plan.h -- functions of the list alone, checked on the host)
  const int own_at = f.n_ext <= FOLLOW_OWN_LAST_MAX ? f.n_ext : chol_follow_own_at(f.n_ext, T);
#define EXT_ROUND(I_) chol_follow_round(I_, f.n_ext, T)
#define EXT_LOAD(I_, V_)                                                                                             \
  {                                                                                                                  \
    const chol_ext xl_ = f.ext[I_];                                                                                  \
    const double *src_ = base + xl_.off;                                                                             \
    _Pragma("unroll") for (int it = 0; it < FOLLOW_LOADS; ++it) {                                                    \
      const int idx = tid + it * RR_THREADS;                                                                         \
      const int row = (idx >> 8) * TS + (idx & 15), col = (idx >> 4) & 15;                                           \
      V_[it] = (idx < T * TS * TS && row < n && col < xl_.ncol) ? gload<true>(&src_[row + (int64_t)col * xl_.ld]) : 0.0; \
    }                                                                                                                \
  }
#define EXT_LOAD_ROUND(I_)                                                                                           \
  {                                                                                                                  \
    EXT_LOAD(I_, va);                                                                                                \
    ng = EXT_ROUND(I_);                                                                                              \
    if (ng == 2) { EXT_LOAD((I_) + 1, vb); }                                                                         \
  }
#define EXT_UPDATE(acc_, ti_, tj_)                                                                                   \
  {                                                                                                                  \
    _Pragma("unroll") for (int st = 0; st < 4; ++st)                                                                 \
      acc_ = __builtin_amdgcn_mfma_f64_16x16x4f64(sB[(tj_) * (TS * TS) + st * 64 + lp], -sB[(ti_) * (TS * TS) + st * 64 + lp], acc_, 0, 0, 0); \
    if (G == 2) {                                                                                                    \
      _Pragma("unroll") for (int st = 0; st < 4; ++st)                                                               \
        acc_ = __builtin_amdgcn_mfma_f64_16x16x4f64(sB2[(tj_) * (TS * TS) + st * 64 + lp], -sB2[(ti_) * (TS * TS) + st * 64 + lp], acc_, 0, 0, 0); \
    }                                                                                                                \
  }
#define OWN_TILES()                                                                                                  \
  {                                                                                                                  \
 /* the follower's own tiles, before the last two followed column tiles (their round trip hides behind the wait for those) */ \
      if (!UPD) wait_list<FOLLOW_OWN_SLEEP>(f.wl, f.n_wl, f.ctr, f.ctr_total, f.epoch, lane, info); /* the factor wave polls: a follower waiting for its own tiles IS on the critical path */ \
      if (TR && !UPD && f.xstamp && lane == 0) f.xstamp[0] = __builtin_amdgcn_s_memrealtime(); \
      lds_barrier(); \
      if (UPD) { \
        int rx = r15, gx = g; /* opaque here: the addresses of these loads are invariant in the loop around them, and hoisting them spills */ \
        asm volatile("" : "+v"(rx), "+v"(gx)); \
    _Pragma("unroll")                                                                                            \
        for (int it = 0; it < 3; ++it) { \
          const int u = min(w + it * RR_NW, 2 * T - 2); \
          stage[it] += load_tile<PUB>(A, lda, n, u < T ? u : u - T + 1, u < T ? 0 : 1, rx, gx); \
        } \
        __builtin_amdgcn_sched_barrier(0); /* two batches of loads: all 28 values and their addresses at once do not fit the registers */ \
    _Pragma("unroll")                                                                                            \
        for (int s = 0; s < FOLLOW_SLOTS; ++s) \
          if (ijp[s] != 0xffff) tile[s] += load_tile<PUB>(A, lda, n, ijp[s] & 0xff, ijp[s] >> 8, rx, gx); \
        __builtin_amdgcn_sched_barrier(0); \
      } \
 \
  }
  for (int i = 0; i < f.n_ext;) {
    if (f.n_wl > 0 && i == own_at) OWN_TILES();
    if (ng == 0) {
      const int need = i + EXT_ROUND(i);
      if (ready < need) {
        // ONE wave of the workgroup polls the counters in global memory (the factor wave: it owns no tile) and passes what it has seen on
        // through an LDS word; the tile waves watch that word (dozens of followers are resident at a time: twelve polling waves each would
        // sit on the memory system the working jobs hand their data through).  The others load only after the word has moved.
        if (!UPD) {
          ready = ext_ready(f, i, lane);
          for (int it = 0; ready < need; ++it) { // bounded like wait_progress: a stall fails the factorisation through info
            if (it >= (1 << 24)) { if (lane == 0) report_stall(info); ready = f.n_ext; break; } // x ~0.15 us per poll
#ifndef EXT_SLEEP
#define EXT_SLEEP 4
#endif
            __builtin_amdgcn_s_sleep(EXT_SLEEP);
            ready = ext_ready(f, i, lane);
          }
          if (lane == 0) __hip_atomic_store(sReady, ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
          for (int it = 0;; ++it) {
            ready = __builtin_amdgcn_readfirstlane(__hip_atomic_load(sReady, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (ready >= need) break;
            if (it >= (1 << 25)) { if (lane == 0) report_stall(info); ready = f.n_ext; break; }
            __builtin_amdgcn_s_sleep(2);
          }
        }
      }
      EXT_LOAD_ROUND(i);
    }
    const int G = ng;
    if (TR && !UPD && f.xstamp && lane == 0 && i < 44) { f.xstamp[1] = f.n_ext; f.xstamp[2 + i] = __builtin_amdgcn_s_memrealtime(); }
    double *sB = sE + buf * (RR_MAXT * TS * TS), *sB2 = sB + T * TS * TS;
#pragma unroll
    for (int it = 0; it < FOLLOW_LOADS; ++it) {
      const int idx = tid + it * RR_THREADS;
      if (idx < T * TS * TS) { sB[idx] = va[it]; if (G == 2) sB2[idx] = vb[it]; }
    }
    const int j = i + G; // first item of the next round
    ng = 0;
    int pv = 0, pneed = 0;
    bool polled = false;
    if (j < f.n_ext && j + EXT_ROUND(j) <= ready) { EXT_LOAD_ROUND(j); }
    else if (j < f.n_ext && !UPD) { // ask now, look after the update (the factor wave; the tile waves look at its LDS word then)
      polled = true;
      if (j + lane < f.n_ext) { const int c = f.ext[j + lane].ctr; pv = __hip_atomic_load(&f.ctr[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); pneed = f.epoch * f.ctr_total[c] + f.ext[j + lane].need; }
    }
    lds_barrier();
    if (UPD) {
#pragma unroll
      for (int it = 0; it < 3; ++it) { // the tiles of columns 0 and 1 (parked in LDS by the prologue later on)
        const int u = w + it * RR_NW;
        if (u < 2 * T - 1) { const int tj = u < T ? 0 : 1, ti = u < T ? u : u - T + 1; EXT_UPDATE(stage[it], ti, tj); }
      }
#pragma unroll
      for (int s = 0; s < FOLLOW_SLOTS; ++s)
        if (ijp[s] != 0xffff) { const int ti = ijp[s] & 0xff, tj = ijp[s] >> 8; EXT_UPDATE(tile[s], ti, tj); }
    }
    buf ^= 1;
    if (polled) {
      const unsigned long long notok = __ballot(j + lane < f.n_ext && pv < pneed);
      const int r2 = min(j + (notok ? __builtin_ctzll(notok) : 64), f.n_ext);
      if (r2 > ready) { ready = r2; if (lane == 0) __hip_atomic_store(sReady, ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    } else if (UPD && j < f.n_ext && ready < j + EXT_ROUND(j)) {
      const int r2 = __builtin_amdgcn_readfirstlane(__hip_atomic_load(sReady, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      ready = r2 > ready ? r2 : ready;
    }
    i = j;
  }
  if (f.n_wl > 0 && own_at >= f.n_ext) OWN_TILES();
#undef OWN_TILES
#undef EXT_LOAD_ROUND
#undef EXT_ROUND
#undef EXT_UPDATE
#undef EXT_LOAD
}

template <bool PUB, bool FOLLOW = false, bool TR = false>
__device__ __forceinline__ void potrf_rr_body(double *__restrict__ base, double *__restrict__ ws, const chol_potrf_desc d, int *__restrict__ info,
                                              int *__restrict__ progress, int progress_base, double *smem, const unsigned char *__restrict__ sky, const follow_args fa = follow_args(), const int tid = threadIdx.x)
{ // sky: the descriptor's skyline in global memory (indexed per tile: a by-value copy would go to scratch)
 // tid: the thread index, passed in by k_program as a value the compiler cannot see through (nothing derived from it may be
  // hoisted out of the job loop: that is what would spill)
  // tiles in LDS are stored like accumulator registers: element (r, c) at c * 16 + r, so lane (r15, g)
  // register q sits at q * 64 + lp with lp = g * 16 + r15 (conflict free, and directly an MFMA operand)
  // the workgroup's LDS image (RR_SMEM_DOUBLES doubles, carved by the caller: the roles of the fused launch share it)
  double (*const sRaw)[TS * TS] = (double (*)[TS * TS])(smem);                         // [RR_MAXT] raw (fully updated, unsolved) tiles of the next panel column
  double (*const sSol)[RR_MAXT][TS * TS] = (double (*)[RR_MAXT][TS * TS])(smem + RR_OFF_SOL); // [2] solved panel P of step k (parity of k)
  double (*const sDg)[TS * TS] = (double (*)[TS * TS])(smem + RR_OFF_DG);              // [2] diagonal tiles on their way to the factor wave (parity of j)
  double (*const sLW)[4 * TS][TS + 1] = (double (*)[4 * TS][TS + 1])(smem + RR_OFF_LW); // [2] parity of k, one row per lane of the factor wave: rows 0-15 = L(k,k) [r][c], rows 16-31 = L(k,k)^-T [k][c] = Linv(c,k) (32-63: the duplicates lanes 32-63 carry; written so that the publication needs no exec mask)
  double (*const sConv)[TS + 1] = (double (*)[TS + 1])(smem + RR_OFF_CONV);            // [2 TS] factor wave: accumulator layout -> row per lane; rows 16-31 = identity
  double (*const sOv)[TS * TS] = (double (*)[TS * TS])(smem + RR_OFF_OV);              // [RR_NHEAVY] slot RR_RSLOTS of the heavy waves
  unsigned short *const sIJ = (unsigned short *)(smem + RR_OFF_IJ);                    // [RR_SLOTS * RR_NW + 16]
  int *const sFlag = (int *)(smem + RR_OFF_FLAG);                                      // [8] fL, fP, cSol, cUpd, cRaw, fA, fD
  unsigned char *const sKm = (unsigned char *)(smem + RR_OFF_KM);                      // [RR_SLOTS * RR_NW]
  int *const sMask = (int *)(smem + RR_OFF_MASK);                                      // [RR_MAXT * RR_NW] step k, tile wave w at k * RR_NW + w
  unsigned char *const sSky = (unsigned char *)(smem + RR_OFF_SKY);                    // [24]
  int *const fL = &sFlag[0], *const fP = &sFlag[1], *const cSol = &sFlag[2], *const cUpd = &sFlag[3];
  int *const cRaw = &sFlag[4]; // raw tiles parked in sRaw so far (column j contributes T - 1 - j)
  int *const fA = &sFlag[6];   // j: raw tile (j, j-1) is in sRaw[j]
  int *const fD = &sFlag[7];   // j: diagonal tile (j, j), updated through step j-2, is in sDg[j & 1]; (fA, fD) = one aligned 64-bit word
  int *const f00 = &sFlag[8];  // 1: the prologue has parked tile (0,0) in sDg[0] -- all the factor wave's first Cholesky waits for

  double *A = base + d.a_off;
  double *W = ws + d.dinv_off;
  const int n = d.n, lda = d.lda;
  const int T = (n + TS - 1) / TS;
  const int ntl = T * (T + 1) / 2;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r15 = lane & 15, g = lane >> 4;
  const int lp0 = g * TS + r15;

#define PSTAMP(i_) do { if (TR && fa.xstamp && fa.n_ext == 0 && tid == 64) fa.xstamp[24 + (i_)] = __builtin_amdgcn_s_memrealtime(); } while (0) /* diagnostic build: the prologue of a block, tile wave 0 */
  PSTAMP(0);
  // a block that follows nobody requests its tiles of columns 0 and 1 before anything else: their round trip runs under the arrival of the role tables
  d4 stage[3];
  const bool stage_early = !(FOLLOW && fa.n_ext > 0) && wave != 0;
  if (stage_early) {
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int u = min(wave - 1 + it * RR_NW, 2 * T - 2);
      stage[it] = load_tile<PUB>(A, lda, n, u < T ? u : u - T + 1, u < T ? 0 : 1, r15, g);
    }
  }
  // The role tables -- which tile lives in which (slot, wave), the per-(step, wave) work masks -- are a function of the block's size and skyline
  // alone: the schedule builds them once (chol_potrf_table) and they arrive behind the descriptor (d.tab); a descriptor without one (the BLAS-
  // and task-level entry points) has them built here
  const unsigned char *const tab = d.tab > 0 ? sky - __builtin_offsetof(chol_potrf_desc, sky) + d.tab : nullptr;
  if (tab) {
    static_assert((RR_MAXT * RR_NW) % 1 == 0 && RR_THREADS >= RR_MAXT * RR_NW + RR_SLOTS * RR_NW / 2 + RR_SLOTS * RR_NW / 4, "one pass copies the tables");
    if (tid < RR_MAXT * RR_NW) sMask[tid] = ((const int *)(tab + CHOL_RR_TAB_MASK))[tid];
    else if (tid < RR_MAXT * RR_NW + RR_SLOTS * RR_NW / 2) ((int *)sIJ)[tid - RR_MAXT * RR_NW] = ((const int *)(tab + CHOL_RR_TAB_IJ))[tid - RR_MAXT * RR_NW];
    else if (tid < RR_MAXT * RR_NW + RR_SLOTS * RR_NW / 2 + RR_SLOTS * RR_NW / 4) ((int *)sKm)[tid - RR_MAXT * RR_NW - RR_SLOTS * RR_NW / 2] = ((const int *)(tab + CHOL_RR_TAB_KM))[tid - RR_MAXT * RR_NW - RR_SLOTS * RR_NW / 2];
  } else {
    for (int t = tid; t < RR_SLOTS * RR_NW; t += RR_THREADS) { sIJ[t] = (unsigned short)0xffff; sKm[t] = 0; }
    for (int t = tid; t < RR_MAXT * RR_NW; t += RR_THREADS) sMask[t] = 0;
  }
  if (tid < 24) sSky[tid] = sky[tid];
  if (tid < 10) sFlag[tid] = 0; // (fA and fD go to 1 when the prologue has parked (1,0) and (1,1))
  if (tid < TS * TS) sConv[TS + (tid >> 4)][tid & 15] = (tid >> 4) == (tid & 15) ? 1.0 : 0.0;
  __syncthreads();
  PSTAMP(1);
  const int ntl2 = (T - 2) * (T - 1) / 2; // tiles of columns >= 2 (0 for T <= 2)
  if (!tab) {
  // Columns 0 and 1 never live in registers (column 0 receives no update, column 1 exactly one): the
  // prologue parks them in LDS.  The tiles of columns >= 2 are dealt in REVERSE column-major order (last
  // column first), so at step k a wave's live tiles (column > k) are its slots [0, rr_count), evenly spread
  // over the waves, and walking the slots downwards visits column k+1 -- next step's panel, the look-ahead
  // tiles first -- before the rest
  for (int t = tid + 2 * T - 1; t < ntl; t += RR_THREADS) {
    int ti, tj, ow, os;
    tile_of_index(t, T, ti, tj);
    rr_owner(ntl - 1 - t, ntl2, ow, os);
    sIJ[os * RR_NW + ow] = (unsigned short)(ti | (tj << 8));
    // skyline (leaf pivots): tile (i, j) receives P(i, k) P(j, k)^T, zero while k is left of either row's first tile
    const int kmin = max(sky[min(ti, 23)], sky[min(tj, 23)]);
    sKm[os * RR_NW + ow] = kmin | (tj < sky[min(ti, 23)] ? CHOL_RR_KM_ZERO : 0);
    // the steps in which the slot has work: its updates (steps kmin .. last) and its hand-over in step `last` -- an off-diagonal
    // tile is parked as a raw panel tile after step tj - 1, a diagonal tile goes to the factor wave one step earlier
    const int last = ti == tj ? tj - 2 : tj - 1;
    for (int k = min(kmin, last); k <= last; ++k) atomicOr(&sMask[k * RR_NW + ow], 1 << os);
  }
  // panel tile i of step k (i >= k + 2; (k+1, k) is the factor wave's) is solved by heavy wave i mod 9, unless it is left of the skyline
  for (int t = tid; t < T * T; t += RR_THREADS) {
    const int k = t / T, i = t % T;
    if (i >= k + 2 && sky[min(i, 23)] <= k) {
      const int hv = i % RR_NHEAVY;
      atomicOr(&sMask[k * RR_NW + hv + hv / 3], 1 << (i < k + 2 + RR_NHEAVY ? RR_M_SOLVE0 : RR_M_SOLVE1));
    }
  }
  __syncthreads();
  }
  PSTAMP(2);

  if (wave == 0) {
    // ================================================================== factor wave
    __builtin_amdgcn_s_setprio(3); // the block's chain runs on this wave: first at whatever it shares with the workgroup's other waves (lapl_3375: 176.2 -> 175.2 us, 4 of 4 A/B rounds)
    if (FOLLOW && fa.n_ext > 0) { // loads and barriers of the external panel steps (it owns no tile)
      d4 tdummy[RR_RSLOTS], sdummy[3];
      int idummy[RR_SLOTS];
      follow_external<false, PUB, TR>(base, fa, T, n, smem + RR_OFF_SOL, tdummy, sdummy, idummy, 0, lane, tid, info, A, lda, r15, g, &sFlag[5]);
      lds_barrier(); // the last panel has been read: the prologue may park column 1 in its place
      if (TR && fa.stamp && lane == 0) *fa.stamp = __builtin_amdgcn_s_memrealtime(); // diagnostic build (k_program<true>): the followed columns are in
    }
    __builtin_amdgcn_s_setprio(3);
    lds_wait_ge(f00, 1); // (0,0) is in sDg: the first Cholesky starts while the other waves' tiles are still on their way (the step's other inputs have flags of their own: fA, fD, cRaw, cUpd)
    d4 dk; // diagonal tile of the current step, accumulator layout
#pragma unroll
    for (int q = 0; q < 4; ++q) dk[q] = sDg[0][q * 64 + lp0];
    STAMP_DECL;
    POLL_DECL;
    for (int k = 0; k < T; ++k) {
      STAMPK(0);
      // ---- a. factor (k,k), one row per lane.  Lanes 16-31 carry the rows of the identity through the same
      //         column operations: they come out as the rows of L(k,k)^-T -- the explicit inverse every panel
      //         solve of this step (and the TRSM kernels later) multiplies with -- at no extra instruction
      double a[TS], unused;
#pragma unroll
      for (int q = 0; q < 4; ++q) sConv[r15][g + 4 * q] = dk[q];
#pragma unroll
      for (int c = 0; c < TS; ++c) a[c] = sConv[lane & 31][c];
      // sLW / sSol of this step's parity were last read in step k-2: every tile wave must have left it before they are written again.
      // The counter is read here, with the rows (by the end of the Cholesky it is a step old, and it is monotonic): a wait after the
      // Cholesky only where this value does not say so yet
      const int cupd_early = __hip_atomic_load(cUpd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      STAMPK(1);
      if (TR && fa.xstamp && lane == 0 && k < 24) fa.xstamp[72 + k] = __builtin_amdgcn_s_memrealtime(); // diagnostic build: the factor wave starts column k
      // ---- b. the two look-ahead tiles of step k-1's trailing update -- (k+1,k) raw and (k+1,k+1) -- reach LDS while this step's
      //         Cholesky runs (their owners do them first): they are requested HALF WAY through it, the flag pair ahead of the tiles in
      //         the same batch of LDS reads (LDS executes a wave's accesses in order), so that their round trip is over when the
      //         Cholesky ends; only a flag that was not up yet at that point costs a poll and a second read of the tiles
      const bool more = k + 1 < T;
      const int kn = more ? k + 1 : k;
      long long fl = 0;
      d4 raw, dn;
      const int bad = chol16_rows(a, unused, r15, [&]() {
        fl = __hip_atomic_load((long long *)fA, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // fA and fD: one 64-bit word
        asm volatile("" ::: "memory");
#pragma unroll
        for (int q = 0; q < 4; ++q) { raw[q] = sRaw[kn][q * 64 + lp0]; dn[q] = sDg[kn & 1][q * 64 + lp0]; }
      });
      STAMPK(2);
      if (bad && k * TS + bad <= n && lane == 0) {
        if (atomicCAS(&info[0], 0, d.col0 + k * TS + bad) == 0) info[1] = d.sep;
      }
      const int par = k & 1;
      if (__builtin_amdgcn_readfirstlane(cupd_early) < RR_NW * k) { POLL_ADD(2, 1); lds_wait_ge(cUpd, RR_NW * k); }
      // ---- c. publish L(k,k) and L(k,k)^-T (every lane writes its row: no exec mask) and, in the same batch, read the inverse back as
      //         the MFMA operand of the look-ahead solve: one LDS round trip for both
#pragma unroll
      for (int c = 0; c < TS; ++c) sLW[par][lane][c] = a[c]; // L: entries above the diagonal are finite junk nobody uses
      double wv[4];
#pragma unroll
      for (int st = 0; st < 4; ++st) wv[st] = sLW[par][TS + 4 * st + g][r15];
      lds_set(fL, k + 1, lane);
      STAMPK(3);
      if (!more) break;
      if (__builtin_amdgcn_readfirstlane((int)fl) < k + 1 || __builtin_amdgcn_readfirstlane((int)(fl >> 32)) < k + 1) {
        POLL_ADD(0, 1);
        lds_wait_both_ge(fA, k + 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) { raw[q] = sRaw[k + 1][q * 64 + lp0]; dn[q] = sDg[(k + 1) & 1][q * 64 + lp0]; }
      }
      STAMPK(4);
      // ---- d. solve (k+1, k); the tile goes to LDS for the tile waves, and -- the writes in flight -- into the diagonal tile
      //         (k+1,k+1) -= P P^T (both operands are the accumulator registers of P); fP once the writes have landed
      const d4 p = solve16(raw, wv);
#pragma unroll
      for (int q = 0; q < 4; ++q) sSol[par][k + 1][q * 64 + lp0] = p[q];
#pragma unroll
      for (int st = 0; st < 4; ++st) dn = __builtin_amdgcn_mfma_f64_16x16x4f64(p[st], -p[st], dn, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0); // the MFMAs are issued before the wait inside lds_set (it is no memory operation: the scheduler would sink them below it)
      lds_set(fP, k + 1, lane);
      STAMPK(5);
      dk = dn;
      STAMPK(6);
    }
    STAMP_FLUSH;
    POLL_FLUSH;
    __builtin_amdgcn_s_setprio(0);
  } else {
    // ================================================================== tile waves
    d4 tile[RR_RSLOTS];
    const int w = wave - 1;
    const int hw = rr_heavy_index(w);
    int ijp[RR_SLOTS]; // packed (i | j << 8) per slot, wave uniform
#pragma unroll
    for (int s = 0; s < RR_SLOTS; ++s) ijp[s] = __builtin_amdgcn_readfirstlane((int)sIJ[s * RR_NW + w]);
    int kmn[RR_SLOTS]; // per slot: the first step whose update of the tile can be non-zero (skyline of a leaf pivot; 0 otherwise)
#pragma unroll
    for (int s = 0; s < RR_SLOTS; ++s) kmn[s] = __builtin_amdgcn_readfirstlane((int)sKm[s * RR_NW + w]);
    unsigned zmask = 0; // slots whose tile is left of the skyline: zero in A, not loaded (wave uniform: a scalar branch per slot, no predicate in front of a load)
#pragma unroll
    for (int s = 0; s < RR_SLOTS; ++s) { zmask |= (kmn[s] & CHOL_RR_KM_ZERO) ? 1u << s : 0u; kmn[s] &= CHOL_RR_KM_ZERO - 1; }
    PSTAMP(5); // slot tables read

    // ---- prologue: columns 0 and 1 go straight to LDS -- (0,0), (1,1) -> sDg, (i,0) -> sRaw[i],
    //      (i,1), i >= 2 -> sSol[1][i] (free until the panel solve of step 1) -- then the register tiles
    //      Every load of the prologue is issued before the first wait: three tiles of columns 0 / 1 per wave at
    //      most (2 T - 1 <= 33 tiles over 11 waves), then the register tiles.
    if (FOLLOW && fa.n_ext > 0) {
      // A follower without waits of its own (no update job of an earlier phase writes its diagonal block: the parents of the
      // leaves) loads its own tiles first -- their round trip overlaps the wait for the first followed column -- and puts the
      // followed contributions on top.  One with waits starts from ZERO accumulators, consumes its children's columns as they
      // arrive, and adds its own tiles just before the last TWO followed column tiles (follow_external: there its waits must hold;
      // the load hides behind the wait for that column): it does not sit idle until its grandchildren's extend-add is through.
      // Which of the two is fixed by the schedule, not by timing: the summation order is the same in every run.
      const bool early = fa.n_wl == 0;
      const d4 zero4 = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
      for (int it = 0; it < 3; ++it) {
        const int u = min(w + it * RR_NW, 2 * T - 2);
        stage[it] = zero4;
        if (early) stage[it] = load_tile<PUB>(A, lda, n, u < T ? u : u - T + 1, u < T ? 0 : 1, r15, g);
      }
#pragma unroll
      for (int s = 0; s < FOLLOW_SLOTS; ++s) { // a follower's block has at most CHOL_FOLLOW_MAXT column tiles: FOLLOW_SLOTS register tiles per wave
        d4 v = zero4;
        if (early && ijp[s] != 0xffff) v = load_tile<PUB>(A, lda, n, ijp[s] & 0xff, ijp[s] >> 8, r15, g);
        tile[s] = v;
      }
      follow_external<true, PUB, TR>(base, fa, T, n, smem + RR_OFF_SOL, tile, stage, ijp, w, lane, tid, info, A, lda, r15, g, &sFlag[5]);
#pragma unroll
      for (int s = FOLLOW_SLOTS; s < RR_RSLOTS; ++s) tile[s] = zero4; // not live across the followed columns
      lds_barrier();
    } else {
    PSTAMP(6);
#pragma unroll
    for (int s = 0; s < RR_RSLOTS; ++s) {
      d4 v = { 0.0, 0.0, 0.0, 0.0 };
      if (ijp[s] != 0xffff && !(zmask & (1u << s))) v = load_tile<PUB>(A, lda, n, ijp[s] & 0xff, ijp[s] >> 8, r15, g);
      tile[s] = v;
    }
    }
    PSTAMP(3); // own tiles requested
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int u = w + it * RR_NW;
      if (u < 2 * T - 1) {
        const int tj = u < T ? 0 : 1, ti = u < T ? u : u - T + 1;
        double *park = ti == tj ? &sDg[ti][0] : tj == 0 ? &sRaw[ti][0] : &sSol[1][ti][0];
#pragma unroll
        for (int q = 0; q < 4; ++q) park[q * 64 + lp0] = stage[it][q];
        if (u == 0) lds_set(f00, 1, lane);
        if (ti == 1) lds_set(tj == 0 ? fA : fD, 1, lane);
        if (tj == 0 && ti > 0) lds_inc(cRaw, lane);
      }
    }
    if (ijp[RR_RSLOTS] != 0xffff) { // heavy waves only (rr_owner)
      const d4 v = load_tile<PUB>(A, lda, n, ijp[RR_RSLOTS] & 0xff, ijp[RR_RSLOTS] >> 8, r15, g);
#pragma unroll
      for (int q = 0; q < 4; ++q) sOv[hw][q * 64 + lp0] = v[q];
    }
    lds_inc(cUpd, lane);
    PSTAMP(4); // columns 0 and 1 parked
    STAMP_DECL;
    POLL_DECL;
    for (int k = 0; k < T; ++k) {
      STAMPK(0);
      int lp = lp0; // opaque once per step: keeps per-slot LDS addresses from being hoisted and spilled
      asm volatile("" : "+v"(lp));
      int k2 = k + 2; // opaque: otherwise "ti == k + 2" makes the flag value a per-slot constant, hoisted and spilled
      asm volatile("" : "+s"(k2));
      const int par = k & 1;
      const int mk = __builtin_amdgcn_readfirstlane(sMask[k * RR_NW + w]); // what this wave does in the step (arrives with the first poll)
      // ---- 1. panel solve out of LDS (needs L(k,k) and every raw tile of column k)
      POLL_WAIT(0, fL, k + 1);
      STAMPK(1);
      POLL_WAIT(1, cRaw, (k + 1) * (T - 1) - k * (k + 1) / 2);
      STAMPK(2);
      if (mk & ((1 << RR_M_SOLVE0) | (1 << RR_M_SOLVE1))) {
        // panel tile i goes to heavy wave i mod 9 (at most two each: T <= 17); the light waves keep the
        // factor wave's SIMD quiet.  X = T Linv^T: four accumulating MFMAs per tile, two tiles interleaved.  Tiles left of a
        // leaf pivot's skyline are zero and stay so: nobody solves, stores or (RR_UPDATE: kmin) reads them
        const int i0 = k + 2 + ((hw + RR_NHEAVY - ((k + 2) % RR_NHEAVY)) % RR_NHEAVY), i1 = i0 + RR_NHEAVY;
        double wv[4];
#pragma unroll
        for (int st = 0; st < 4; ++st) wv[st] = sLW[par][TS + 4 * st + g][r15];
        double *const col = A + (int64_t)(k * TS + g) * lda;
        if (((mk >> RR_M_SOLVE0) & 3) == 3) { // two tiles (pivot blocks wider than 11 tiles only), interleaved
          const int row0 = i0 * TS + r15, row1 = i1 * TS + r15;
          d4 raw0, raw1, x0 = { 0.0, 0.0, 0.0, 0.0 }, x1 = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
          for (int q = 0; q < 4; ++q) { raw0[q] = sRaw[i0][q * 64 + lp]; raw1[q] = sRaw[i1][q * 64 + lp]; }
#pragma unroll
          for (int st = 0; st < 4; ++st) {
            x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(wv[st], raw0[st], x0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(wv[st], raw1[st], x1, 0, 0, 0);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            sSol[par][i0][q * 64 + lp] = x0[q];
            sSol[par][i1][q * 64 + lp] = x1[q];
            if (row0 < n) gstore<PUB>(&col[row0 + (int64_t)(4 * q) * lda], x0[q]);
            if (row1 < n) gstore<PUB>(&col[row1 + (int64_t)(4 * q) * lda], x1[q]);
          }
        } else {
          const int i = (mk & (1 << RR_M_SOLVE0)) ? i0 : i1;
          const int row = i * TS + r15;
          d4 raw, x = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
          for (int q = 0; q < 4; ++q) raw[q] = sRaw[i][q * 64 + lp];
#pragma unroll
          for (int st = 0; st < 4; ++st) x = __builtin_amdgcn_mfma_f64_16x16x4f64(wv[st], raw[st], x, 0, 0, 0);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            sSol[par][i][q * 64 + lp] = x[q];
            if (row < n) gstore<PUB>(&col[row + (int64_t)(4 * q) * lda], x[q]);
          }
        }
      }
      // ---- 2. the whole panel is solved
      STAMPK(3);
      lds_inc(cSol, lane);
      POLL_WAIT(2, cSol, RR_NW * (k + 1));
      STAMPK(4);
      if (k + 1 < T) POLL_WAIT(3, fP, k + 1);
      STAMPK(5);
      // ---- 3. trailing update of the live slots [0, top], walked downwards: column k+1 first
      const double *const sS = &sSol[par][0][0];
      if (k == 0) { // column 1 out of LDS: its only update, (2,1) -- the look-ahead tile -- first
        for (int i = 2 + w; i < T; i += RR_NW) {
          d4 acc;
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] = sSol[1][i][q * 64 + lp];
          if (sSky[min(i, 23)] == 0) { // P(i, 0) is zero, and was not solved, otherwise
#pragma unroll
          for (int st = 0; st < 4; ++st)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sS[1 * (TS * TS) + st * 64 + lp], -sS[i * (TS * TS) + st * 64 + lp], acc, 0, 0, 0);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) sRaw[i][q * 64 + lp] = acc[q];
          lds_inc(cRaw, lane);
          if (i == 2) lds_set(fA, 2, lane);
        }
      }
      STAMPX(0);
      // one tile: acc -= P(ti) P(tj)^T, then park it if this was its last update; true if it was parked
#define RR_UPDATE(acc_, ti_, tj_, kmin_)                                                                           \
  {                                                                                                                \
    if (k >= (kmin_)) { /* left of the skyline P(ti, k) or P(tj, k) is zero: nothing to subtract */                \
    _Pragma("unroll") for (int st = 0; st < 4; ++st)                                                               \
      acc_ = __builtin_amdgcn_mfma_f64_16x16x4f64(sS[(tj_) * (TS * TS) + st * 64 + lp], -sS[(ti_) * (TS * TS) + st * 64 + lp], acc_, 0, 0, 0); \
    }                                                                                                              \
    if ((tj_) == k + 1) { /* last update: next panel column */                                                    \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) sRaw[ti_][q * 64 + lp] = acc_[q];                              \
      lds_inc(cRaw, lane);                                                                                         \
      if ((ti_) == k + 2) lds_set(fA, k2, lane); /* look-ahead: the factor wave solves it itself */                \
    } else if ((ti_) == (tj_) && (tj_) == k + 2) { /* diagonal tile, one step ahead */                            \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) sDg[(tj_) & 1][q * 64 + lp] = acc_[q];                         \
      lds_set(fD, k2, lane);                                                                                       \
    }                                                                                                              \
  }
      // the step mask names the slots with work: live tiles (column > k) that receive a non-zero update or are handed over
      if (mk & (1 << RR_RSLOTS)) { // the LDS-resident slot holds the wave's earliest columns: first in line
        const int ti = ijp[RR_RSLOTS] & 0xff, tj = ijp[RR_RSLOTS] >> 8;
        d4 acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = sOv[hw][q * 64 + lp];
        RR_UPDATE(acc, ti, tj, kmn[RR_RSLOTS]);
#pragma unroll
        for (int q = 0; q < 4; ++q) sOv[hw][q * 64 + lp] = acc[q];
        __builtin_amdgcn_sched_barrier(0);
      }
      STAMPX(1);
#pragma unroll
      for (int sg = (RR_RSLOTS - 1) / 4; sg >= 0; --sg) {
        STAMPX(2 + sg);
        if (mk & (0xf << (4 * sg))) {
#pragma unroll
          for (int s = (4 * sg + 3 < RR_RSLOTS ? 4 * sg + 3 : RR_RSLOTS - 1); s >= 4 * sg; --s) {
            if (mk & (1 << s)) { // ((k+1,k+1) went to the factor wave one step ago: not in the mask)
              const int ti = ijp[s] & 0xff, tj = ijp[s] >> 8;
              d4 acc = tile[s];
              RR_UPDATE(acc, ti, tj, kmn[s]);
              tile[s] = acc;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
#undef RR_UPDATE
      STAMPX(5);
      // ---- the light waves solve no panel tiles: one of them copies L(k,k) and Linv(k,k) from LDS to global
      //      memory after its (short) update chain, off every other wave's path (sLW of this parity stays
      //      valid until every tile wave has finished step k)
      if (w == ((k & 1) ? 7 : 3)) {
        const int row = k * TS + r15;
        double *dst = A + row + (int64_t)(k * TS + g) * lda;
        double *Wb = W + (int64_t)k * TS * TS;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const double lv = sLW[par][r15][g + 4 * b], wv = sLW[par][TS + g + 4 * b][r15];
          if (row < n && g + 4 * b <= r15) gstore<PUB>(&dst[(int64_t)(4 * b) * lda], lv);
          gstore<PUB>(&Wb[(g + 4 * b) * TS + r15], wv); // Wb[k * 16 + c] = Linv(c, k): the layout solve16() reads
        }
        if (k + 1 < T && row + TS < n) { // L(k+1, k), solved by the factor wave, is still in LDS
#pragma unroll
          for (int q = 0; q < 4; ++q) gstore<PUB>(&dst[TS + (int64_t)(4 * q) * lda], sSol[par][k + 1][q * 64 + lp]);
        }
      }
      // ---- 4.  Fused launch: the wave's stores of column k have landed before it counts itself out of the
      //      step, and the last wave out tells the TRSM workgroups that column k and Linv(k,k) can be read
      STAMPK(6);
      if (PUB) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        int old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(cUpd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old + 1 == RR_NW * (k + 2) && lane == 0) {
          __hip_atomic_store(progress, progress_base + k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (TR && fa.xstamp && k < 24) fa.xstamp[48 + k] = __builtin_amdgcn_s_memrealtime(); // diagnostic build: column k published
        }
      } else {
        lds_inc(cUpd, lane);
      }
    }
    if (w == STAMP_WAVE) { STAMP_FLUSH2; }
    POLL_WFLUSH;
  }
}

__global__ __launch_bounds__(RR_THREADS) void k_potrf_rr(double *__restrict__ base, double *__restrict__ ws,
                                                         const chol_potrf_desc *__restrict__ descs, int *__restrict__ info)
{
  __shared__ double smem[RR_SMEM_DOUBLES];
  potrf_rr_body<false>(base, ws, descs[blockIdx.x], info, nullptr, 0, smem, descs[blockIdx.x].sky);
}

// ------------------------------------------------------------------------------------------------
// POTRF for pivots larger than CHOL_RR_MAXN: same blocking, trailing matrix in global memory (L2).
// One workgroup of 256 threads per pivot.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_potrf_big(double *__restrict__ base, double *__restrict__ ws,
                                                   const chol_potrf_desc *__restrict__ descs, int *__restrict__ info)
{
  __shared__ double sL[TS][TS + 1];
  __shared__ double sYd[4][TS];
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
      double a[TS], blk[4], x[4], myinv;
      const int row = j0 + r15;
#pragma unroll
      for (int c = 0; c < TS; ++c) {
        double v = (r15 == c) ? 1.0 : 0.0;
        if (row < n && j0 + c < n) v = (c <= r15) ? A[row + (int64_t)(j0 + c) * lda] : 0.0;
        a[c] = v;
      }
      const int bad = chol16_rows(a, myinv, r15);
      if (bad && j0 + bad <= n && lane == 0) {
        if (atomicCAS(&info[0], 0, d.col0 + j0 + bad) == 0) info[1] = d.sep;
      }
      own_block_row(a, r15, blk);
      linv4_quad(blk, myinv, x, r15 & 3);
      if (lane < TS) {
        const int b4 = lane & ~3, qi = lane & 3;
#pragma unroll
        for (int c = 0; c < TS; ++c) {
          sL[lane][c] = (c <= lane) ? a[c] : 0.0;
          if (row < n && c <= lane) A[row + (int64_t)(j0 + c) * lda] = a[c];
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) sYd[qi][b4 + m] = x[m];
      }
    }
    __syncthreads();
    // panel: 16-row tiles below the diagonal tile
    const int nt = below > 0 ? (below + TS - 1) / TS : 0;
    {
      double Lr[3];
#pragma unroll
      for (int b = 0; b < 3; ++b) Lr[b] = sL[r15][g + 4 * b];
      const double yd = sYd[g][r15];
      if (wave == 3) store_linv16(W + (int64_t)k * TS * TS, Lr, yd, r15, g);
      for (int t = wave; t < nt; t += 4) {
        const int row = j0 + TS + t * TS + r15;
        d4 tv;
#pragma unroll
        for (int q = 0; q < 4; ++q) tv[q] = (row < n) ? A[row + (int64_t)(j0 + g + 4 * q) * lda] : 0.0;
        const d4 x = tile_solve(tv, Lr, yd);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (row < n) A[row + (int64_t)(j0 + g + 4 * q) * lda] = x[q];
      }
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
// Ydiag of the 16x16 diagonal blocks of an already factored L (BLAS-/task-level TRSM entry points,
// where L was not produced by a POTRF kernel of the same call chain).  One wave per block.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_dinv(const double *__restrict__ Lp, int n, int ldl, double *__restrict__ W)
{
  __shared__ double sYd[4][TS];
  const int j0 = blockIdx.x * TS, lane = threadIdx.x, r15 = lane & 15, g = lane >> 4;
  const int row = j0 + r15, b4 = r15 & ~3, qi = r15 & 3;
  double blk[4], x[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int col = j0 + b4 + k;
    double v = (k == qi) ? 1.0 : 0.0; // identity padding past n
    if (row < n && col < n) v = (k <= qi) ? Lp[row + (int64_t)col * ldl] : 0.0;
    blk[k] = v;
  }
  double diag = blk[0];
  diag = (qi == 1) ? blk[1] : diag;
  diag = (qi == 2) ? blk[2] : diag;
  diag = (qi == 3) ? blk[3] : diag;
  linv4_quad(blk, 1.0 / diag, x, qi);
  if (lane < TS) {
#pragma unroll
    for (int m = 0; m < 4; ++m) sYd[qi][b4 + m] = x[m];
  }
  __syncthreads();
  double Lr[3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const int c = g + 4 * b;
    double v = (r15 == c) ? 1.0 : 0.0;
    if (row < n && j0 + c < n) v = (c <= r15) ? Lp[row + (int64_t)(j0 + c) * ldl] : 0.0;
    Lr[b] = v;
  }
  store_linv16(W + (int64_t)blockIdx.x * TS * TS, Lr, sYd[g][r15], r15, g);
}

// ------------------------------------------------------------------------------------------------
// TRSM: B <- B L^-T for a strip of <= 16 rows (cblas_dtrsm Right/Lower/Trans/NonUnit alpha=1,
// blas.rg:99), n <= CHOL_RR_MAXN.  The strip's 16x16 column tiles live in registers (accumulator
// layout), tile J owned by wave J mod 4.  Right-looking: the owner solves X_J = T_J L(J,J)^-T
// (tile_solve), publishes X_J in LDS (double buffered, one barrier per step), and every wave
// applies T_J'' -= X_J L(J'', J)^T to its tiles J'' > J with the L tile streamed from global
// memory / L2 ahead of the barrier.
// ------------------------------------------------------------------------------------------------
#define TRSM_SLOTS 5 /* ceil(17 / 4) */
// one strip, four waves (wave = 0..3 of the strip's group); sX = the group's three LDS tiles
// BAND (program launch, a banded leaf pivot factored as ONE block of up to CHOL_RR_MAXN columns): tile (J2, J) of the block's L is zero for
// J2 - J > d.band (<= 4), so step J touches at most one tile per wave (J2 = wave mod 4 within (J, J + band]): the L prefetch ring
// holds ONE tile per step instead of one per slot -- what lets a strip of seventeen column tiles fit the registers.
template <bool PUB, int SLOTS, bool BAND = false, bool TR = false>
__device__ __forceinline__ void trsm_rr_body(double *__restrict__ base, const double *__restrict__ ws, const chol_trsm_desc d, double (*sX)[TS * TS],
                                             int wave, int lane, const int *__restrict__ progress, int progress_base, int *__restrict__ info,
                                             int *__restrict__ chan = nullptr, unsigned long long *xst = nullptr)
{ // chan (program launch): the strip's rows are followed by a POTRF workgroup -- counter chan[J] is raised once column tile J
  // of the strip has been stored (follow_external)
  const double *Lm = base + d.l_off;
  const double *W = ws + d.dinv_off;
  double *B = base + d.b_off;
  const int n = d.n, m = d.m, ldl = d.ldl, ldb = d.ldb;
  const int T = (n + TS - 1) / TS;
  const int r15 = lane & 15, g = lane >> 4;
  const int lp = g * TS + r15;
  const bool vrow = r15 < m;
  const int bw = BAND ? d.band : (1 << 20); // tiles below the diagonal that can be non-zero
#define LS(s_) (BAND ? 0 : (s_))

  d4 tile[SLOTS];
  // lane part of the address of L(J2 * 16 + r15, g): rows past n are clamped (their products only reach output columns >= n, which are
  // never stored).  Recomputed where it is used (two VALU instructions): kept as a per-slot array the band form's seventeen column
  // tiles pushed two of these 64-bit values into scratch, reloaded in every step of the leaf strips
  const int64_t gld = (int64_t)g * ldl;
#define VOFF(J2_) ((int64_t)min((J2_) * TS + r15, n - 1) + gld)
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int J = wave + 4 * s;
    d4 v = { 0.0, 0.0, 0.0, 0.0 };
    if (J < T) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = J * TS + g + 4 * q;
        v[q] = (vrow && col < n) ? gload<PUB>(&B[r15 + (int64_t)col * ldb]) : 0.0; // program launch: last written by an update job of the same launch
      }
    }
    tile[s] = v;
  }
  // L tiles (J2, J) of the updates of step J (uniform branch per slot, one scalar base per step)
#define LOAD_L(J_, buf_)                                                                                \
  if ((J_) < T) {                                                                                       \
    if (PUB) seen = wait_progress(progress, progress_base + (J_) + 1, seen, info);                      \
    const double *lb_ = Lm + (int64_t)((J_) * TS) * ldl;                                                \
    _Pragma("unroll") for (int s = 0; s < SLOTS; ++s) {                                            \
      const int J2_ = wave + 4 * s;                                                                     \
      if (J2_ > (J_) && J2_ < T && J2_ <= (J_) + bw) {                                                  \
        _Pragma("unroll") for (int st = 0; st < 4; ++st) buf_[LS(s)][st] = gload<PUB>(&lb_[VOFF(J2_) + (int64_t)(4 * st) * ldl]); \
      }                                                                                                 \
    }                                                                                                   \
  }
#define LOAD_W(J_, buf_)                                                                                \
  {                                                                                                     \
    if (PUB) seen = wait_progress(progress, progress_base + (J_) + 1, seen, info);                      \
    _Pragma("unroll") for (int st = 0; st < 4; ++st) buf_[st] = gload<PUB>(&W[(int64_t)(J_) * TS * TS + (4 * st + g) * TS + r15]); \
  }
#define PUBLISH_X(J_, x_)                                                                               \
  {                                                                                                     \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                     \
      const int col_ = (J_) * TS + g + 4 * q;                                                           \
      sX[(J_) % 3][q * 64 + lp] = x_[q];                                                                \
      if (vrow && col_ < n) gstore<PUB>(&B[r15 + (int64_t)col_ * ldb], x_[q]);                          \
    }                                                                                                   \
    if (PUB && chan && m > 0) { /* this wave stored the whole column tile: tell the follower once it has landed */ \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                  \
      if (lane == 0) __hip_atomic_fetch_add(&chan[J_], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  \
      if (TR && xst && lane == 0 && (J_) < 24) xst[48 + (J_)] = __builtin_amdgcn_s_memrealtime(); /* diagnostic build: column tile J_ published */ \
    }                                                                                                   \
  }
#define APPLY_X(JX_, s_)                                                                                \
  {                                                                                                     \
    d4 acc_ = tile[s_];                                                                                 \
    _Pragma("unroll") for (int st = 0; st < 4; ++st)                                                    \
      acc_ = __builtin_amdgcn_mfma_f64_16x16x4f64(lpre[(JX_) % 3][LS(s_)][st], -sX[(JX_) % 3][st * 64 + lp], acc_, 0, 0, 0); \
    tile[s_] = acc_;                                                                                    \
  }
  // operands are fetched two steps ahead; vmcnt retires in order, so the inverse the next solve waits
  // for is issued before the L tiles of the same step
  int seen = 0; // last progress value read (fused launch)
  // the inverse of column tile c is loaded by its owner (wave c mod 4) in step c - 2 and used in step c - 1: ONE buffer per wave (a ring
  // of three, as for the L tiles, was 16 more registers that the band form's strips spilled -- with a vmcnt(0) behind every prefetch)
  double lpre[3][BAND ? 1 : SLOTS][4], wpre[4];
#pragma unroll
  for (int st = 0; st < 4; ++st) wpre[st] = 0.0;
#pragma unroll
  for (int u = 0; u < 3; ++u) {
#pragma unroll
    for (int st = 0; st < 4; ++st) {
#pragma unroll
      for (int s = 0; s < (BAND ? 1 : SLOTS); ++s) lpre[u][s][st] = 0.0;
    }
  }
  if (wave == 0) LOAD_W(0, wpre);
  if (wave == 1 && 1 < T) LOAD_W(1, wpre);
  LOAD_L(0, lpre[0]);
  LOAD_L(1, lpre[1]);
  if (wave == 0) { // column tile 0 has no predecessors
    const d4 x = solve16(tile[0], wpre);
    PUBLISH_X(0, x);
  }
  STAMP_DECL;
#pragma unroll
  for (int J = 0; J < 4 * SLOTS; ++J) {
    if (J < T) {
      STAMP(0);
      const bool next_owner = (J + 1 < T) && (((J + 1) & 3) == wave);
      lds_barrier(); // X_J visible
      STAMP(1);
      if (next_owner) {
        // critical chain: bring tile J+1 up to date, solve it, publish it -- nothing else before the
        // next barrier; this wave's other updates with X_J are deferred to the next step
        const int s1 = (J + 1) >> 2;
        if (s1 < SLOTS) {
          APPLY_X(J, s1);
          const d4 x = solve16(tile[s1], wpre);
          tile[s1] = x;
          PUBLISH_X(J + 1, x);
        }
      } else {
        if (J >= 1 && (J & 3) == wave) { // owner of the previous look-ahead: catch up with X_{J-1}
#pragma unroll
          for (int s = 0; s < SLOTS; ++s) {
            const int J2 = wave + 4 * s;
            if (J2 > J && J2 < T && J2 <= J - 1 + bw) APPLY_X(J - 1, s);
          }
        }
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
          const int J2 = wave + 4 * s;
          if (J2 > J + 1 && J2 < T && J2 <= J + bw) APPLY_X(J, s);
        }
      }
      STAMP(2);
      // prefetch for step J+2: buffer (J+2) % 3 == (J-1) % 3 is free (deferred work uses X_J / lpre[J % 3])
      if ((J + 2 < T) && (((J + 2) & 3) == wave)) LOAD_W(J + 2, wpre);
      LOAD_L(J + 2, lpre[(J + 2) % 3]);
      STAMP(3);
    }
  }
  if (wave == 1 && blockIdx.x == 0) { STAMP_FLUSH; }
#undef LOAD_L
#undef LOAD_W
#undef PUBLISH_X
#undef APPLY_X
#undef LS
#undef VOFF
}

__global__ __launch_bounds__(256) void k_trsm_rr(double *__restrict__ base, const double *__restrict__ ws,
                                                 const chol_trsm_desc *__restrict__ descs)
{
  __shared__ double sX[3][TS * TS]; // [J mod 3][c * 16 + r]: solved column tiles, accumulator-register order
  trsm_rr_body<false, TRSM_SLOTS>(base, ws, descs[blockIdx.x], sX, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), threadIdx.x & 63, nullptr, 0, nullptr);
}

// ------------------------------------------------------------------------------------------------
// Fused POTRF + TRSM launch of one column-block step of a level.  Workgroups [0, n_potrf) factor one
// pivot block each (potrf_rr_body) and publish, column by column, how far L and Linv are in global
// memory; the others take three 16-row strips of one pivot block each (four waves per strip, trsm_rr_body)
// and follow their pivot's POTRF two columns behind instead of waiting for the whole level's POTRF
// launch to end: the TRSM launch (7-14 us, latency-bound) and one kernel boundary per step disappear
// from the critical path.  Data handed over inside the launch is written with agent-scope stores and
// read with agent-scope loads (device-coherent across the XCDs' L2s); the progress word is written
// after the workgroup's stores of that column have completed (s_waitcnt vmcnt(0)).
// ------------------------------------------------------------------------------------------------
#define FUSED_SLOTS ((CHOL_FUSE_MAXN / TS + 3) / 4) /* column tiles per wave of a strip: the registers of a 768-thread workgroup hold three */
__global__ __launch_bounds__(RR_THREADS) void k_potrf_trsm(double *__restrict__ base, double *__restrict__ ws,
                                                           const chol_potrf_desc *__restrict__ pdescs, int n_potrf,
                                                           const chol_trsm_desc *__restrict__ tdescs, int n_trsm,
                                                           const chol_upd_task *__restrict__ tasks, const chol_upd_src *__restrict__ srcs, int n_task, int n_upd_wg,
                                                           int *__restrict__ info, int *__restrict__ progress, int progress_base,
                                                           int *__restrict__ done, int done_target)
{
  __shared__ double smem[RR_SMEM_DOUBLES]; // one image, carved by the role of the workgroup
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int grp = wave >> 2;
  const int n_tg = (n_trsm + 2) / 3;
  if ((int)blockIdx.x < n_potrf) {
    potrf_rr_body<true>(base, ws, pdescs[blockIdx.x], info, progress + blockIdx.x, progress_base, smem, pdescs[blockIdx.x].sky);
  } else if ((int)blockIdx.x < n_potrf + n_tg) {
    double (*sX)[3][TS * TS] = (double (*)[3][TS * TS])smem;
    const int id = ((int)blockIdx.x - n_potrf) * 3 + grp;
    // every group runs the same number of barriers: the strips of a workgroup share one pivot block
    chol_trsm_desc d = tdescs[min(id, n_trsm - 1)];
    if (id >= n_trsm) d.m = 0;
    trsm_rr_body<true, FUSED_SLOTS>(base, ws, d, sX[grp], wave & 3, lane, progress + d.flag, progress_base, info);
    // the update workgroups of this launch (if any) read the solved strips: count this workgroup out once its stores have completed
    if (n_task > 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
      if (threadIdx.x == 0) __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    // update role: three tasks at a time (one per group of four waves), tasks dealt round-robin to the update
    // workgroups; the first one waits until every TRSM workgroup of the launch has counted itself out
    double (*sAcc)[3][3][4][64] = (double (*)[3][3][4][64])smem; // [parity of the round][group]
    const int u = (int)blockIdx.x - n_potrf - n_tg;
    int round = 0;
    for (int t0 = u * 3; t0 < n_task; t0 += n_upd_wg * 3, ++round) {
      const int tid = t0 + grp;
      const bool live = tid < n_task;
      update_task_body<true>(base, tasks[live ? tid : t0], srcs, sAcc[round & 1][grp], wave & 3, lane, live, round == 0 ? done : nullptr, done_target, info);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// PROGRAM launch: the whole factorisation of a small problem as ONE launch.  A grid of resident 768-thread workgroups draws
// jobs from a queue in a topological order (chol_build_program): POTRF of a pivot block, a group of three TRSM strips, a
// group of 16x16 update tasks of one target block -- the role bodies of the fused launch.  Jobs hand data to each other
// inside the launch through agent-scope stores / loads and monotonic counters (a job waits for its list of (counter, value)
// pairs, every lane polling one of them; it raises its own counters once every wave's stores have completed).  What this
// removes from the critical path of the level-by-level schedule: 13 kernel boundaries, the ramp of each launch, the wait for
// the slowest pivot of a level, and -- with followers (follow_external) -- the update launch and the prologue between a
// pivot and its parent / the next column block of a split pivot.
// Liveness: a workgroup only takes a job once it is running, jobs are taken in queue order, and a job waits for jobs that
// are ahead of it in the queue or for the strips of a source that the queue places within reach of the resident
// workgroups (checked on the host: chol_program_check); every spin is bounded and fails the factorisation through info.
// ------------------------------------------------------------------------------------------------
template <bool TR>
__global__ __launch_bounds__(RR_THREADS) void k_program(double *__restrict__ base, double *__restrict__ ws, const chol_job *__restrict__ jobs, int njobs,
                                                        const chol_wait *__restrict__ waits, const chol_potrf_desc *__restrict__ pdescs,
                                                        const chol_trsm_desc *__restrict__ tdescs, const chol_upd_task *__restrict__ tasks,
                                                        const chol_upd_src *__restrict__ srcs, const chol_ext *__restrict__ exts,
                                                        int *__restrict__ ctr, const int *__restrict__ ctr_total, int epoch, int *__restrict__ head, int head_base,
                                                        int *__restrict__ info, int *__restrict__ info_next, unsigned long long *__restrict__ trace)
{
  // the info words of the NEXT factorisation of this device object (the other of two slots) are cleared here: no memset node per launch
  if (info_next && blockIdx.x == 0 && threadIdx.x == 0) { info_next[0] = 0; info_next[1] = 0; } // trace (diagnostic runs only, else nullptr): per job the 100 MHz real-time clock when it was drawn, when its waits were over
  // and when it ended, and the workgroup that ran it
  __shared__ double smem[RR_SMEM_DOUBLES];
  __shared__ int s_job;
  const int wave_id = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // a scalar register for the whole launch; threadIdx.x itself (a vector register that
                                                                        // would stay live across every role body) is rebuilt per job from it and the lane count
  for (;;) {
#ifdef TID_PLAIN
    int tid = threadIdx.x;
#else
    int tid = wave_id * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#endif
    asm volatile("" : "+v"(tid)); // opaque per job: lane-dependent addresses of the role bodies are not hoisted out of this loop (and spilled)
    const int wave = wave_id;
    const int lane = tid & 63;
    const int grp = wave >> 2;
    __syncthreads(); // the previous job is through with the LDS image
    if (tid == 0) s_job = atomicAdd(head, 1) - head_base;
    __syncthreads();
    const int j = s_job;
    if (j >= njobs) break; // every workgroup draws exactly one job index past the end: head advances by njobs + gridDim.x per launch
    const chol_job jb = jobs[j];
    if (TR && trace && tid == 0) { trace[4 * j] = __builtin_amdgcn_s_memrealtime(); trace[4 * j + 3] = blockIdx.x; }
    // ONE wave polls (hundreds of blocked workgroups are resident at a time: twelve polling waves each would sit on the L2 the
    // working jobs hand their data through); the others park at the barrier
    // A job is usually drawn long before it may run: what does not depend on the wait is fetched ahead of it -- the first
    // round's task descriptor and its source descriptors (immutable; scalar loads, kept warm in the scalar cache)
    chol_upd_task tpre;
    if (jb.kind == 2) {
      const int tk = jb.first + min(jb.mode == 0 ? wave : grp, jb.n - 1);
      tpre = tasks[tk];
      int touch = 0;
      for (int sx = tpre.src_begin; sx < tpre.src_end; ++sx) touch += srcs[sx].k;
      asm volatile("" :: "s"(touch));
    }
    if (jb.n_pre > 0 && !(jb.kind == 0 && jb.n_ext > 0)) { // a follower looks at its waits itself (potrf_rr_body); an extend-add job's staged waits are its tasks' business
      if (wave == 0) wait_list(waits + jb.wait_first, jb.n_pre, ctr, ctr_total, epoch, lane, info);
      lds_barrier();
    }
    if (TR && trace && tid == 0) trace[4 * j + 1] = __builtin_amdgcn_s_memrealtime(); // a follower overwrites it when its followed columns are in
    if (jb.kind == 0) {
      const chol_potrf_desc pd = pdescs[jb.first];
      follow_args fa;
      fa.ext = exts + jb.ext_first; fa.n_ext = jb.n_ext; fa.ctr = ctr; fa.ctr_total = ctr_total; fa.epoch = epoch; fa.wl = waits + jb.wait_first; fa.n_wl = jb.n_wait; fa.stamp = TR && trace ? &trace[4 * j + 1] : nullptr; fa.xstamp = TR && trace ? &trace[4 * njobs + CHOL_TRACE_X * j] : nullptr;
      potrf_rr_body<true, true, TR>(base, ws, pd, info, ctr + pd.ctr, epoch * ctr_total[pd.ctr], smem, pdescs[jb.first].sky, fa, tid);
    } else if (jb.kind == 1) {
      double (*sX)[3][TS * TS] = (double (*)[3][TS * TS])smem;
      chol_trsm_desc d = tdescs[jb.first + min(grp, jb.n - 1)];
      if (grp >= jb.n) d.m = 0; // every group runs the same number of barriers: the strips of a job share one pivot block
      if (d.band > 0) // a banded leaf pivot factored as one block (up to CHOL_RR_MAXN columns: (CHOL_RR_MAXN / 16 + 3) / 4 column tiles per wave)
        trsm_rr_body<true, (CHOL_RR_MAXN / TS + 3) / 4, true, TR>(base, ws, d, sX[grp], wave & 3, lane, ctr + d.flag, epoch * ctr_total[d.flag], info, d.chan >= 0 ? ctr + d.chan : nullptr,
                                                                   TR && trace && grp == 0 ? &trace[4 * njobs + CHOL_TRACE_X * j] : nullptr);
      else
        trsm_rr_body<true, FUSED_SLOTS, false, TR>(base, ws, d, sX[grp], wave & 3, lane, ctr + d.flag, epoch * ctr_total[d.flag], info, d.chan >= 0 ? ctr + d.chan : nullptr,
                                                   TR && trace && grp == 0 ? &trace[4 * njobs + CHOL_TRACE_X * j] : nullptr);
    } else {
      stage_waits sw;
      sw.w = waits + jb.wait_first + jb.n_pre; sw.n = jb.n_wait - jb.n_pre; sw.ctr = ctr; sw.ctr_total = ctr_total; sw.epoch = epoch; sw.info = info;
      if (jb.mode == 0) {
        for (int t0 = jb.first; t0 < jb.first + jb.n; t0 += RR_NW + 1) { // light tasks: one per wave
          const int tk = t0 + wave;
          if (tk < jb.first + jb.n) update_task_wave<true>(base, t0 == jb.first ? tpre : tasks[tk], srcs, lane, sw);
        }
      } else { // heavy tasks: three at a time, four waves each (K or the sources split, fixed-order LDS reduction)
        double (*sAcc)[3][3][4][64] = (double (*)[3][3][4][64])smem; // [parity of the round][group]
        int round = 0;
        for (int t0 = jb.first; t0 < jb.first + jb.n; t0 += 3, ++round) {
          const int tk = t0 + grp;
          const bool live = tk < jb.first + jb.n;
          update_task_body<true>(base, t0 == jb.first ? tpre : tasks[live ? tk : t0], srcs, sAcc[round & 1][grp], wave & 3, lane, live, nullptr, 0, info, sw);
        }
      }
    }
    if (jb.sig[0] >= 0) { // the job's stores have completed (every wave's) before its counters move
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
      if (tid == 0) {
        __hip_atomic_fetch_add(&ctr[jb.sig[0]], jb.sig_add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (jb.sig[1] >= 0) __hip_atomic_fetch_add(&ctr[jb.sig[1]], jb.sig_add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (TR && trace && tid == 0) trace[4 * j + 2] = __builtin_amdgcn_s_memrealtime();
  }
}

// ------------------------------------------------------------------------------------------------
// TRSM for pivot blocks of at most CHOL_TRSM_W_MAXN columns: ONE WAVE per 16-row strip,
// the four strips of a workgroup share one pivot block (the schedule pads every block's strips to a
// multiple of four; descriptors with m = 0 are placeholders).
//   * The block's L tiles and the Linv(J,J) tiles go to LDS once, by LDS-DMA (global_load_lds, 1 KiB
//     per wave-instruction, no registers): every request of the workgroup is in flight before the
//     first wait, so the strip pays the memory latency (the pivot was written by another workgroup,
//     usually on another XCD: > 1 us) once instead of once per step.
//   * The strip's column tiles stay in registers.  Step J: X_J = T_0 Linv(J,J)^T (4 MFMAs), then
//     T_j -= X_J L(j, J)^T for j > J, X_J used as the MFMA operand straight out of its accumulator
//     registers; one straight-line instantiation per tile count.
//   * One wave issues an MFMA per 64 cycles at best, so the strip costs 4 T (T + 1) / 2 x 64 cycles plus the
//     staging: faster than the four-waves-per-strip kernel up to 64 columns, slower beyond (hence the limit).
// ------------------------------------------------------------------------------------------------
#define TW_MAXT ((CHOL_TRSM_W_MAXN + TS - 1) / TS)
// the solve of one strip of T tiles: tiles in registers, operands out of the staged LDS image (s0 = slot 0 + lane)
template <int T, int NT> __device__ __forceinline__ void trsm_w_solve(d4 (&tile)[NT], const double *s0, double *__restrict__ B, int n, int ldb,
                                                              bool vrow, int r15, int g)
{
#pragma unroll
  for (int J = 0; J < T; ++J) {
    const double *sd = s0 + (J * T - J * (J - 1) / 2) * (TS * TS); // slot (J, J)
    double wv[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) wv[st] = sd[st * 64];
    const d4 x = solve16(tile[J], wv);
    if (vrow) {
      int ro = r15;
      asm volatile("" : "+v"(ro)); // the lane's store address is rebuilt per column tile (two instructions): hoisted, it was the one 64-bit value
                                   // the nine-tile instance spilled, with a scratch reload in front of every tile's stores
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = J * TS + g + 4 * q;
        if (J + 1 < T || col < n) B[ro + (int64_t)col * ldb] = x[q];
      }
    }
    double nx[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      nx[st] = -x[st];
      asm volatile("" : "+v"(nx[st])); // negated once per step, not once per MFMA
    }
#pragma unroll
    for (int j = J + 1; j < T; ++j) {
      d4 acc = tile[j];
#pragma unroll
      for (int st = 0; st < 4; ++st) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sd[(j - J) * (TS * TS) + st * 64], nx[st], acc, 0, 0, 0);
      tile[j] = acc;
    }
  }
}
__global__ __launch_bounds__(256) void k_trsm_w(double *__restrict__ base, const double *__restrict__ ws,
                                                const chol_trsm_desc *__restrict__ descs, int ndesc)
{
  // slot(J2, J) = J T - J (J - 1) / 2 + (J2 - J), J2 >= J: tile L(J2, J) as the MFMA Y operand (element (c, k) at
  // (k / 4) * 64 + (k % 4) * 16 + c, i.e. accumulator-register order); the diagonal slots hold Linv(J,J) in the
  // layout solve16() reads (the workspace layout, copied verbatim)
  __shared__ double sT[TW_MAXT * (TW_MAXT + 1) / 2][TS * TS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int id0 = blockIdx.x * 4;
  const chol_trsm_desc d0 = descs[id0];
  const double *Lm = base + d0.l_off;
  const double *W = ws + d0.dinv_off;
  const int n = d0.n, ldl = d0.ldl;
  const int T = (n + TS - 1) / TS;
  const int r15 = lane & 15, g = lane >> 4, lp = lane;
  int64_t b_off = d0.b_off;
  int m = 0, ldb = d0.ldb;
  if (id0 + wave < ndesc) {
    const chol_trsm_desc d = descs[id0 + wave];
    b_off = d.b_off; m = d.m; ldb = d.ldb;
  }
  double *B = base + b_off;
  const bool vrow = r15 < m;
  STAMP_DECL;
  STAMP(0);

  // ---- the LDS-DMA pairs rows (r, r+1): with n odd the pair (n-1, n) is fetched from (n-2, n-1) (in bounds,
  //      finite) and row n-1 of the last row tile is patched afterwards from this register
  const bool needfix = (n & 1) && tid < TS * (T - 1);
  double fix = 0.0;
  if (needfix) fix = Lm[(n - 1) + (int64_t)tid * ldl]; // column tid = 16 J + c of row n-1
  // ---- stage: wave w takes the half-tiles (slot, half) with slot = (w >> 1) mod 2, half = w & 1
  {
    const int half = wave & 1;
    const int pi = half * 64 + lane;                            // element pair (2 pi, 2 pi + 1) of the tile image
    const int st = pi >> 5, gg = (pi >> 3) & 3, re = (pi & 7) * 2;
    int J = 0, J2 = wave >> 1, slot = wave >> 1;
    while (J < T && J2 >= T) { J2 = J2 - T + J + 1; ++J; }
    while (J < T) {
      const double *src;
      if (J2 == J) src = W + (int64_t)J * TS * TS + 2 * pi;
      else {
        const int row = J2 * TS + re;
        src = Lm + (row + 1 < n ? row : n - 2) + (int64_t)(J * TS + 4 * st + gg) * ldl;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)&sT[slot][half * 128], 16, 0, 0);
      J2 += 2; slot += 2;
      while (J < T && J2 >= T) { J2 = J2 - T + J + 1; ++J; }
    }
  }
  // ---- the strip's tiles (clamped, unconditional loads + select)
  d4 tile[TW_MAXT];
  {
    const int rb = min(r15, max(m - 1, 0));
#pragma unroll
    for (int J = 0; J < TW_MAXT; ++J) {
#pragma unroll
      for (int q = 0; q < 4; ++q) tile[J][q] = B[rb + (int64_t)min(J * TS + g + 4 * q, n - 1) * ldb];
    }
  }
  STAMP(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(2);
  __syncthreads(); // every wave's LDS-DMA has landed
  if (needfix) {
    const int J = tid >> 4, c = tid & 15, rr = (n - 1) & 15;
    sT[J * T - J * (J - 1) / 2 + (T - 1 - J)][(c >> 2) * 64 + (c & 3) * TS + rr] = fix;
  }
  __syncthreads();
  if (m <= 0) return;
  // the loads above are unconditional (clamped addresses) and only now masked: a load under its predicate gets
  // a branch and a vmcnt(0) of its own, 36 memory latencies in a row
#pragma unroll
  for (int J = 0; J < TW_MAXT; ++J) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double v = tile[J][q];
      asm volatile("" : "+v"(v));
      tile[J][q] = (vrow && J * TS + g + 4 * q < n) ? v : 0.0;
    }
  }
  STAMP(3);

  // ---- the solve, one straight-line instantiation per tile count (run-time guards per tile cost register
  //      copies at every join and keep the LDS operand reads from running ahead of the MFMAs)
  const double *const s0 = &sT[0][lp];
  switch (T) {
  case 1: trsm_w_solve<1>(tile, s0, B, n, ldb, vrow, r15, g); break;
  case 2: trsm_w_solve<2>(tile, s0, B, n, ldb, vrow, r15, g); break;
  case 3: trsm_w_solve<3>(tile, s0, B, n, ldb, vrow, r15, g); break;
  case 4: trsm_w_solve<4>(tile, s0, B, n, ldb, vrow, r15, g); break;
  default: break;
  }
  STAMP(4);
  if (blockIdx.x == 0 && wave == 0) { STAMP_FLUSH; }
}

// ------------------------------------------------------------------------------------------------
// TRSM, throughput form (level schedule, steps with thousands of strips: the wide fronts of the generated problems): the same
// one-wave-per-strip solve for pivot blocks up to CHOL_TRSM_WT_MAXN = 144 columns, TWELVE strips per 768-thread workgroup (168 registers a wave: the nine column tiles of a strip stay in registers) sharing
// one staged image of the block (45 L tiles + 9 inverses = 92 KB of LDS).  No barrier after the staging, no LDS traffic but operand
// reads: twelve independent MFMA chains per CU keep the matrix pipe busy, where the fused launch's four-waves-per-strip solve is a
// latency design (three strips per CU at a time, a barrier per column tile): 20 000 strips of a 60^3 front step take 6 800
// workgroups x 20 us there.  The POTRF of such a step is launched on its own ahead of this kernel.
// ------------------------------------------------------------------------------------------------
#define TT_MAXT ((CHOL_TRSM_WT_MAXN + TS - 1) / TS)
#define TT_WAVES CHOL_TRSM_WT_GROUP
__global__ __launch_bounds__(64 * TT_WAVES) void k_trsm_wt(double *__restrict__ base, const double *__restrict__ ws,
                                                           const chol_trsm_desc *__restrict__ descs, int ndesc)
{
  __shared__ double sT[TT_MAXT * (TT_MAXT + 1) / 2][TS * TS]; // slots as in k_trsm_w
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int id0 = blockIdx.x * TT_WAVES;
  const chol_trsm_desc d0 = descs[id0];
  const double *Lm = base + d0.l_off;
  const double *W = ws + d0.dinv_off;
  const int n = d0.n, ldl = d0.ldl;
  const int T = (n + TS - 1) / TS;
  const int r15 = lane & 15, g = lane >> 4, lp = lane;
  int64_t b_off = d0.b_off;
  int m = 0, ldb = d0.ldb;
  if (id0 + wave < ndesc) {
    const chol_trsm_desc d = descs[id0 + wave];
    b_off = d.b_off; m = d.m; ldb = d.ldb;
  }
  // the strip's base is wave uniform: kept in scalar registers (as a vector pair it was the one value the T = 9 instance spilled -- a scratch
  // reload in front of every column tile's store)
  b_off = ((int64_t)__builtin_amdgcn_readfirstlane((int)(b_off >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)b_off);
  m = __builtin_amdgcn_readfirstlane(m); ldb = __builtin_amdgcn_readfirstlane(ldb);
  double *B = base + b_off;
  const bool vrow = r15 < m;
  // the LDS-DMA pairs rows (r, r + 1): with n odd the pair (n - 1, n) is fetched from (n - 2, n - 1) and row n - 1 of the last row
  // tile is patched afterwards from this register (k_trsm_w)
  const bool needfix = (n & 1) && tid < TS * (T - 1);
  double fix = 0.0;
  if (needfix) fix = Lm[(n - 1) + (int64_t)tid * ldl];
  { // stage: half-tile h = 2 slot + half goes to wave h mod 16
    const int nslots = T * (T + 1) / 2;
    for (int h = wave; h < 2 * nslots; h += TT_WAVES) {
      const int slot = h >> 1, half = h & 1;
      int J = 0, rem = slot;
      while (rem >= T - J) { rem -= T - J; ++J; }
      const int J2 = J + rem;
      const int pi = half * 64 + lane; // element pair (2 pi, 2 pi + 1) of the tile image
      const int st = pi >> 5, gg = (pi >> 3) & 3, re = (pi & 7) * 2;
      const double *src;
      if (J2 == J) src = W + (int64_t)J * TS * TS + 2 * pi;
      else {
        const int row = J2 * TS + re;
        src = Lm + (row + 1 < n ? row : n - 2) + (int64_t)(J * TS + 4 * st + gg) * ldl;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)&sT[slot][half * 128], 16, 0, 0);
    }
  }
  d4 tile[TT_MAXT];
  {
    const int rb = min(r15, max(m - 1, 0));
#pragma unroll
    for (int J = 0; J < TT_MAXT; ++J) {
#pragma unroll
      for (int q = 0; q < 4; ++q) tile[J][q] = B[rb + (int64_t)min(J * TS + g + 4 * q, n - 1) * ldb];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads(); // every wave's LDS-DMA has landed
  if (needfix) {
    const int J = tid >> 4, c = tid & 15, rr = (n - 1) & 15;
    sT[J * T - J * (J - 1) / 2 + (T - 1 - J)][(c >> 2) * 64 + (c & 3) * TS + rr] = fix;
  }
  __syncthreads();
  if (m <= 0) return;
#pragma unroll
  for (int J = 0; J < TT_MAXT; ++J) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double v = tile[J][q];
      asm volatile("" : "+v"(v));
      tile[J][q] = (vrow && J * TS + g + 4 * q < n) ? v : 0.0;
    }
  }
  const double *const s0 = &sT[0][lp];
  switch (T) {
  case 1: trsm_w_solve<1>(tile, s0, B, n, ldb, vrow, r15, g); break;
  case 2: trsm_w_solve<2>(tile, s0, B, n, ldb, vrow, r15, g); break;
  case 3: trsm_w_solve<3>(tile, s0, B, n, ldb, vrow, r15, g); break;
  case 4: trsm_w_solve<4>(tile, s0, B, n, ldb, vrow, r15, g); break;
  case 5: trsm_w_solve<5>(tile, s0, B, n, ldb, vrow, r15, g); break;
  case 6: trsm_w_solve<6>(tile, s0, B, n, ldb, vrow, r15, g); break;
  case 7: trsm_w_solve<7>(tile, s0, B, n, ldb, vrow, r15, g); break;
  case 8: trsm_w_solve<8>(tile, s0, B, n, ldb, vrow, r15, g); break;
  case 9: trsm_w_solve<9>(tile, s0, B, n, ldb, vrow, r15, g); break;
  default: break;
  }
}

// ------------------------------------------------------------------------------------------------
// TRSM for pivots larger than CHOL_RR_MAXN: one independent wave per 16-row strip (4 strips per
// workgroup), left-looking from global memory: T_J = B_J - X_<J L(J,<J)^T, X_J = T_J L(J,J)^-T.
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
    double wv[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) wv[st] = W[(int64_t)J * TS * TS + (4 * st + g) * TS + r15];
    const d4 x = solve16(t, wv);
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
    const int rr = r - d.y_off; // the source's rows start at row y_off of the target separator (a stored row run of its block)
    if (rr >= 0 && rr < d.m) {
      valid = true;
      const double *A = base + d.a_off + rr;
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
// Driver-level solve (cholamd_solve) at scale.  The kernels above keep the reference's per-call shape (one
// workgroup per separator / per 256 target rows, dependent loads per column) and serve the BLAS-level entry
// points; on a 10^6-unknown factor they take seconds.  These stream every panel once instead:
//   * diagonal blocks: the 16x16 inverses of all diagonal blocks are recomputed into the solve workspace
//     (k_solve_dinv: the factorisation's workspace belongs to the last arena factored, not to this one), and
//     the triangular solves run in 64-column block steps with the in-block work done out of LDS
//   * off-diagonal blocks: one workgroup per row chunk of an (ancestor, separator) block, source-centric,
//     accumulating into y with hardware fp64 atomics (the only non-deterministic summation order in the library:
//     the solution agrees from run to run to rounding, not bit for bit)
// ------------------------------------------------------------------------------------------------
#define SNB 64
// The driver-level solve kernels are templates over the factor's element type TL (double, or float for the fp32 factor of
// the mixed-precision path): L is converted on load, vectors and arithmetic are fp64 either way.
template <class TL>
__global__ __launch_bounds__(64) void k_solve_dinv(const TL *__restrict__ base, const chol_trsv_desc *__restrict__ descs, double *__restrict__ W)
{ // grid (separators of the level, 16-column blocks of the widest one)
  __shared__ double sYd[4][TS];
  const chol_trsv_desc d = descs[blockIdx.x];
  const int j0 = blockIdx.y * TS;
  if (j0 >= d.n) return;
  const TL *Lp = base + d.a_off;
  const int n = d.n, ldl = d.lda;
  const int lane = threadIdx.x, r15 = lane & 15, g = lane >> 4;
  const int row = j0 + r15, b4 = r15 & ~3, qi = r15 & 3;
  double blk[4], x[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int col = j0 + b4 + k;
    double v = (k == qi) ? 1.0 : 0.0; // identity padding past n
    if (row < n && col < n) v = (k <= qi) ? (double)Lp[row + (int64_t)col * ldl] : 0.0;
    blk[k] = v;
  }
  double diag = blk[0];
  diag = (qi == 1) ? blk[1] : diag;
  diag = (qi == 2) ? blk[2] : diag;
  diag = (qi == 3) ? blk[3] : diag;
  linv4_quad(blk, 1.0 / diag, x, qi);
  if (lane < TS) {
#pragma unroll
    for (int m = 0; m < 4; ++m) sYd[qi][b4 + m] = x[m];
  }
  __syncthreads();
  double Lr[3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const int c = g + 4 * b;
    double v = (r15 == c) ? 1.0 : 0.0;
    if (row < n && j0 + c < n) v = (c <= r15) ? (double)Lp[row + (int64_t)(j0 + c) * ldl] : 0.0;
    Lr[b] = v;
  }
  store_linv16(W + d.dinv_off + (int64_t)blockIdx.y * TS * TS, Lr, sYd[g][r15], r15, g);
}

// y_s <- L_ss^-1 y_s (forward) or L_ss^-T y_s (backward), one workgroup per separator, 64-column block steps:
// the block's triangle and its four 16x16 inverses go to LDS, wave 0 solves the block there, then every thread
// folds the solved block into the rows below (forward) / the block gathers the solved rows below first (backward)
// Only the columns [col0, col0 + SSPAN) of every separator are handled per launch: the rows below that span are folded
// in (forward) / gathered (backward) by k_solve_panel over all CUs, so a wide separator's triangle is not streamed by
// one workgroup.
#define SSPAN 256
// sL: [r][c] of the current diagonal block; sxs: the span's part of the vector, in LDS from the first block to the last.  PUB: the solution is written with
// agent-scope stores (workgroups of the same launch read it: k_solve_step)
template <bool BWD, class TL, bool PUB, bool XAG = false>
__device__ __forceinline__ void trsv_body(const TL *__restrict__ base, const chol_trsv_desc &d, const double *__restrict__ Wall, double *__restrict__ y, int col0,
                                          double (*sL)[SNB + 1], double (*sW)[TS * TS], double *sxs)
{
  const TL *Lm = base + d.a_off;
  const double *W = Wall + d.dinv_off;
  double *x = y + d.x_off;
  const int lda = d.lda, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (d.n <= col0) return;
  const int n = min(d.n, col0 + SSPAN); // rows and columns of this launch's span
  const int nblk = (n - col0 + SNB - 1) / SNB;
  // The span is a chain of (at most) four 64-column blocks: in-block solve on wave 0, then the fold of the block into the span's other rows (forward:
  // the rows below, x[r] -= L(r, J) x_J; backward: the columns in front, x[c] -= L(J, c)^T x_J).  Nothing the chain reads from memory depends on x,
  // and a 256-thread workgroup alone on its CU owns 512 registers per lane: EVERY load of the span -- the four triangles, their 16x16 inverses and
  // the six fold passes -- is requested before the first block is solved, so the span costs one memory round trip instead of three per block
  // (100 spans per direction at 100^3 are the critical path of a solve: 62 / 71 us per span in profiles/r3, forward / backward, before this).
  constexpr int NB4 = SSPAN / SNB, NL = SNB * SNB / 256, NW = (SNB / TS) * TS * TS / 256, NF = NB4 * (NB4 - 1) / 2;
  static_assert(SSPAN == 256 && NB4 == 4, "one thread per span row, four blocks");
  TL lv[NB4][NL], fa[NF][16];
  double wv[NB4][NW];
  const int fi = tid >> 2, part = tid & 3; // fold: four threads per row (forward) / column (backward), sixteen elements each
#pragma unroll
  for (int bi = 0; bi < NB4; ++bi) {
    if (bi >= nblk) break;
    const int J0 = col0 + (BWD ? nblk - 1 - bi : bi) * SNB, jb = min(SNB, n - J0);
#pragma unroll
    for (int u = 0; u < NL; ++u) { // (clamped addresses: a partial block reads its last row / column again)
      const int e = tid + 256 * u, r = e & (SNB - 1), c = e >> 6;
      lv[bi][u] = Lm[(J0 + min(r, jb - 1)) + (int64_t)(J0 + min(c, jb - 1)) * lda];
    }
#pragma unroll
    for (int u = 0; u < NW; ++u) {
      const int e = tid + 256 * u, t = min(e >> 8, (jb - 1) / TS);
      wv[bi][u] = W[(int64_t)(J0 / TS + t) * TS * TS + (e & 255)];
    }
#pragma unroll
    for (int p = 0; p < NB4 - 1 - bi; ++p) { // the bi-th block of the chain has at most 3 - bi blocks of 64 rows / columns to fold into
      const int f = bi * (NB4 - 1) - bi * (bi - 1) / 2 + p;
      if (BWD) {
        const int c = col0 + 64 * p + fi; // a column in front of the block (every block in front of another one is full)
        if (col0 + 64 * p < J0) {
#pragma unroll
          for (int u = 0; u < 16; ++u) fa[f][u] = Lm[J0 + min(16 * part + u, jb - 1) + (int64_t)c * lda];
        }
      } else {
        const int r0 = J0 + jb + 64 * p;
        if (r0 < n) {
#pragma unroll
          for (int u = 0; u < 16; ++u) fa[f][u] = Lm[min(r0 + fi, n - 1) + (int64_t)(J0 + min(16 * part + u, jb - 1)) * lda];
        }
      }
    }
  }
  sxs[tid] = col0 + tid < n ? gload<XAG>(&x[col0 + tid]) : 0.0;
#pragma unroll
  for (int bi = 0; bi < NB4; ++bi) {
    if (bi >= nblk) break;
    const int J0 = col0 + (BWD ? nblk - 1 - bi : bi) * SNB, jb = min(SNB, n - J0);
    double *const sx = sxs + (J0 - col0); // the block's right-hand side, then its solution (the blocks solved before it have folded themselves in)
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int e = tid + 256 * u, r = e & (SNB - 1), c = e >> 6;
      sL[r][c] = (r < jb && c <= r) ? (double)lv[bi][u] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < NW; ++u) { const int e = tid + 256 * u; sW[e >> 8][e & 255] = wv[bi][u]; }
    __syncthreads();
    if (wave == 0) { // in-block solve with the 16x16 inverses, lane = row of the block.  The sub-block's right-hand side and solution go
                     // through sx (LDS executes a wave's accesses in order; every lane reads them back as broadcasts): sixteen shuffles of a value
                     // that the same loop updates were sixteen dependent LDS-crossbar round trips per sub-block and loop -- most of the span's time
      double v = sx[lane];
      const int nsub = (jb + TS - 1) / TS, g4 = lane >> 4, r16 = lane & 15;
      for (int su = 0; su < nsub; ++su) {
        const int j = BWD ? nsub - 1 - su : su;
        if (g4 == j) sx[lane] = v; // the sub-block's right-hand side, the earlier sub-blocks folded in
        // x_j = Linv_j v_j (forward) / Linv_j^T v_j (backward); sW[j][k * 16 + c] = Linv(c, k)
        double xj = 0.0;
#pragma unroll
        for (int k = 0; k < TS; ++k) xj += (BWD ? sW[j][r16 * TS + k] : sW[j][k * TS + r16]) * sx[j * TS + k];
        if (g4 == j) sx[lane] = xj;
        // fold x_j into the rest of the block: every lane forms its dot product (unconditional LDS reads: they batch behind one wait; inside a
        // per-lane-group branch each of the sixteen was a branch, two reads and a wait of its own), the lanes it concerns subtract it
        double fd = 0.0;
#pragma unroll
        for (int k = 0; k < TS; ++k) fd += (BWD ? sL[j * TS + k][lane] : sL[lane][j * TS + k]) * sx[j * TS + k];
        if (BWD ? g4 < j : g4 > j) v -= fd;
      }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NB4 - 1 - bi; ++p) {
      const int f = bi * (NB4 - 1) - bi * (bi - 1) / 2 + p;
      const int tgt = BWD ? 64 * p + fi : (J0 - col0) + jb + 64 * p + fi; // the target row / column inside the span
      const bool live = BWD ? col0 + 64 * p < J0 : J0 + jb + 64 * p < n;
      if (live) {
        double acc = 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += (16 * part + u < jb) ? (double)fa[f][u] * sx[16 * part + u] : 0.0;
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        if (part == 0 && col0 + tgt < n) sxs[tgt] -= acc;
      }
    }
    __syncthreads(); // the block's folds are in LDS (the next block's right-hand side), sL / sW may be rewritten
  }
  if (col0 + tid < n) gstore<PUB>(&x[col0 + tid], sxs[tid]);
}
template <bool BWD, class TL>
__global__ __launch_bounds__(256) void k_solve_trsv(const TL *__restrict__ base, const chol_trsv_desc *__restrict__ descs, const double *__restrict__ Wall,
                                                    double *__restrict__ y, int col0)
{
  __shared__ double sL[SNB][SNB + 1];
  __shared__ double sW[SNB / TS][TS * TS];
  __shared__ double sxs[SSPAN];
  trsv_body<BWD, TL, false>(base, descs[blockIdx.x], Wall, y, col0, sL, sW, sxs);
}
// The diagonal solve of one 256-column span out of an fp32 factor (round 4; the spans of the wide top separators are two thirds of a solve at 100^3: 101 per
// direction, 23 / 26 us each in k_solve_trsv).  One thread per row (forward) / column (backward) of the span; EVERY entry of the span's triangle the thread
// will need -- its row left of its own 16-block, or its column below it: up to 240 floats -- is requested before the first block is solved (a 256-thread
// workgroup alone on its CU owns 512 registers per lane), and so is its block's 16x16 inverse.  Then sixteen block steps with ONE barrier each: the sixteen
// lanes that hold block j's right-hand side form x_j = Linv_j r_j in registers (sixteen DPP multiply-adds: lane k of the row supplies r_k), publish it in
// LDS, and every thread on the far side of the block subtracts its sixteen products.  No triangle in LDS, no shuffles, no second barrier: the next block's
// lanes go on from their own registers.
// PUB: the solution is written with agent-scope stores -- workgroups of the same launch read it (k_solve_step); XAG: the right-hand side is read with
// agent-scope loads -- atomics of this launch have touched it (k_solve_leaf32)
template <bool BWD, bool PUB, bool XAG = false>
__device__ __forceinline__ void span32_body(const float *__restrict__ base, const chol_trsv_desc &d, const double *__restrict__ Wall, double *__restrict__ y, int col0, double *sx)
{
  const int lda = d.lda, tid = threadIdx.x, myb = tid >> 4, l16 = tid & 15;
  const int ns = min(d.n - col0, SSPAN), nb = (ns + TS - 1) / TS;
  const float *Lm = base + d.a_off + col0 + (int64_t)col0 * lda; // element (0, 0) of the span's diagonal block
  const double *Wb = Wall + d.dinv_off + (int64_t)(col0 / TS + min(myb, nb - 1)) * TS * TS; // Wb[k * 16 + c] = Linv(c, k) of the thread's own block
  double *x = y + d.x_off + col0;
  float lrow[(SSPAN / TS - 1) * TS], ltail[TS];
  double wl[TS];
  const bool live = tid < ns;
  const int nfull = ns / TS; // blocks of sixteen whole rows; block nfull (if nfull < nb) is the ragged one
#pragma unroll
  for (int k = 0; k < TS; ++k) wl[k] = BWD ? Wb[l16 * TS + k] : Wb[k * TS + l16]; // forward: Linv(l, k); backward: Linv(k, l)
  double r = live ? gload<XAG>(&x[tid]) : 0.0;
  if (!BWD) { // row tid: the columns of the blocks in front of its own
    const float *Lr = Lm + tid;
#pragma unroll
    for (int j = 0; j < SSPAN / TS - 1; ++j) {
      if (live && j < myb) {
#pragma unroll
        for (int k = 0; k < TS; ++k) lrow[j * TS + k] = Lr[(int64_t)(j * TS + k) * lda];
      } else {
#pragma unroll
        for (int k = 0; k < TS; ++k) lrow[j * TS + k] = 0.0f;
      }
    }
  } else { // column tid: the rows of the blocks below its own (slot j - 1 = block j); sixteen consecutive floats per block: 16-byte loads
    const float *Lc = Lm + (int64_t)min(tid, ns - 1) * lda;
#pragma unroll
    for (int j = 1; j < SSPAN / TS; ++j) {
      if (live && j > myb && j < nfull) {
#pragma unroll
        for (int k4 = 0; k4 < TS; k4 += 4) {
          const float4 v = *(const float4 *)&Lc[j * TS + k4];
          lrow[(j - 1) * TS + k4] = v.x; lrow[(j - 1) * TS + k4 + 1] = v.y; lrow[(j - 1) * TS + k4 + 2] = v.z; lrow[(j - 1) * TS + k4 + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int k = 0; k < TS; ++k) lrow[(j - 1) * TS + k] = 0.0f;
      }
    }
#pragma unroll
    for (int k = 0; k < TS; ++k) { // the ragged last block: its rows past the span's end do not exist (clamped address, zero value)
      const int row = nfull * TS + k;
      const float v = Lc[min(row, ns - 1)];
      ltail[k] = (live && nfull > myb && row < ns) ? v : 0.0f;
    }
  }
#pragma unroll
  for (int jb = 0; jb < SSPAN / TS; ++jb) {
    const int j = BWD ? SSPAN / TS - 1 - jb : jb; // compile time: the register indices below are constants
    if (j < nb) {
      if (myb == j) { // the block's own sixteen lanes: x_j = Linv_j r_j (backward: its transpose), lane k of the row supplies r_k
        double xj = 0.0;
#define SPAN_FM(K_) fmac_bcast<K_, K_ == 0>(xj, r, wl[K_]);
        SPAN_FM(0) SPAN_FM(1) SPAN_FM(2) SPAN_FM(3) SPAN_FM(4) SPAN_FM(5) SPAN_FM(6) SPAN_FM(7)
        SPAN_FM(8) SPAN_FM(9) SPAN_FM(10) SPAN_FM(11) SPAN_FM(12) SPAN_FM(13) SPAN_FM(14) SPAN_FM(15)
#undef SPAN_FM
        r = xj;
        sx[tid] = xj;
      }
      __syncthreads();
      if (BWD ? myb < j : myb > j) {
        double xv[TS], acc0 = 0.0, acc1 = 0.0; // the block's solution in eight 16-byte LDS reads, all in flight at once; two chains of eight multiply-adds
#pragma unroll
        for (int k = 0; k < TS; k += 2) {
          const double2 t = *(const double2 *)&sx[j * TS + k];
          xv[k] = t.x; xv[k + 1] = t.y;
        }
#pragma unroll
        for (int k = 0; k < TS; ++k) { // the value stays a float in its register until here: a conversion hoisted to the load would wait for it there
          float v = (BWD && j == nfull) ? ltail[k] : lrow[(BWD ? j - 1 : j) * TS + k];
          asm("" : "+v"(v));
          if (k & 1) acc1 += (double)v * xv[k]; else acc0 += (double)v * xv[k];
        }
        r -= acc0 + acc1;
      }
    }
  }
  if (live) gstore<PUB>(&x[tid], r);
}
template <bool BWD>
__global__ __launch_bounds__(256) void k_solve_span32(const float *__restrict__ base, const chol_trsv_desc *__restrict__ descs, const double *__restrict__ Wall,
                                                      double *__restrict__ y, int col0)
{
  __shared__ __attribute__((aligned(16))) double sx[SSPAN];
  const chol_trsv_desc d = descs[blockIdx.x];
  if (d.n <= col0) return;
  span32_body<BWD, false>(base, d, Wall, y, col0, sx);
}

// ---- column sums over a wave without the LDS crossbar (round 4) ----
// The backward sweep needs sum_i L(i, c) x(i) with the rows i along the lanes (the coalesced direction of a column-major panel): a reduction over the 64
// lanes per column.  Six __shfl_down stages per column are twelve ds_bpermute_b32 each, and the one LDS pipe of a CU bounded the whole backward sweep
// (2.4 TB/s on the leaf level of 100^3, 1.47 ms of an 11 ms solve).  gfx950 swaps half-waves and 16-lane rows BETWEEN two registers in one VALU
// instruction (v_permlane32_swap, v_permlane16_swap): two columns' partial sums fold into one register per stage, so sixteen columns cost 8 + 4 swaps-and-adds
// and four registers of four 16-lane rows, each finished by four DPP rotations -- 84 VALU instructions on four SIMDs in place of 192 bpermutes on one pipe.
typedef unsigned chol_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double fold32(double a, double b)
{ // lanes 0-31: a(l) + a(l + 32); lanes 32-63: b(l - 32) + b(l)
  const chol_u2 lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const chol_u2 hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ double fold16(double a, double b)
{ // 16-lane rows: (a.row0 + a.row1, b.row0 + b.row1, a.row2 + a.row3, b.row2 + b.row3)
  const chol_u2 lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const chol_u2 hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
template <int CTRL> __device__ __forceinline__ double dpp_mov64(double v)
{
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_sum16(double v)
{ // every lane: the sum over its 16-lane row
  v += dpp_mov64<0x128>(v); // row_ror:8
  v += dpp_mov64<0x124>(v); // row_ror:4
  v += dpp_mov64<0x4e>(v);  // quad_perm [2,3,0,1]
  v += dpp_mov64<0xb1>(v);  // quad_perm [1,0,3,2]
  return v;
}
// the sums over all 64 lanes of acc[0..15]: lane l with (l & 15) < 4 returns the sum of column wave_sum16_col(l); the other lanes return junk
__device__ __forceinline__ int wave_sum16_col(int lane) { return 4 * (lane & 3) + ((lane >> 5) & 1) + 2 * ((lane >> 4) & 1); }
__device__ __forceinline__ double wave_sum16(const double (&acc)[16], int lane)
{
  double t[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { // rows of t[i]: columns 4 i + (0, 2, 1, 3)
    const double s0 = fold32(acc[4 * i], acc[4 * i + 1]), s1 = fold32(acc[4 * i + 2], acc[4 * i + 3]);
    t[i] = row_sum16(fold16(s0, s1));
  }
  const int i = lane & 3;
  return i == 0 ? t[0] : i == 1 ? t[1] : i == 2 ? t[2] : t[3];
}
// y(c) -= sum over the chunk's rows of A(i, c) x(i) for the columns [0, ncols) of a column-major block: `A` points at (first row of the chunk, column 0),
// `mrows` rows of the chunk exist (<= 64 PER), xa[u] = x(row lane + 64 u) (0 past the end).  Wave `wave` of four takes the columns 16 (wave + 4 k) ..+15:
// 16 PER loads in flight per lane, one wave_sum16 and ONE atomic instruction (16 lanes) per sixteen columns.
template <int PER, class TL>
__device__ __forceinline__ void gather_columns(const TL *__restrict__ A, int lda, int mrows, int ncols, const double (&xa)[PER], double *__restrict__ yc, int lane, int wave)
{
  int ro[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) ro[u] = min(lane + 64 * u, mrows - 1);
  for (int c0 = 16 * wave; c0 < ncols; c0 += 64) {
    TL a[16][PER];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const TL *Ac = A + (int64_t)min(c0 + q, ncols - 1) * lda;
#pragma unroll
      for (int u = 0; u < PER; ++u) a[q][u] = Ac[ro[u]]; // unconditional (clamped rows repeat the last one, x = 0 there): a predicate here serialises the loads
    }
    double acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      acc[q] = (double)a[q][0] * xa[0];
#pragma unroll
      for (int u = 1; u < PER; ++u) acc[q] += (double)a[q][u] * xa[u];
    }
    const double sum = wave_sum16(acc, lane);
    const int c = c0 + wave_sum16_col(lane);
    if ((lane & 15) < 4 && c < ncols) unsafeAtomicAdd(&yc[c], -sum);
  }
}

// rows of a wide separator under a span that one workgroup of k_solve_panel folds / gathers.  The spans of the top separators are the solve's chain and
// at the root there is ONE separator: with 512-row chunks a span step of the 10^4-column root kept 10 CUs busy on average
#ifndef SPANEL_BW_ROWS
#define SPANEL_BW_ROWS 128
#endif
#define SPANEL_FW_ROWS 256
template <bool BWD, class TL>
__global__ __launch_bounds__(256) void k_solve_panel(const TL *__restrict__ base, const chol_trsv_desc *__restrict__ descs, double *__restrict__ y, int col0)
{
  __shared__ double sx[SSPAN];
  const chol_trsv_desc d = descs[blockIdx.x];
  const int r0 = col0 + SSPAN;
  const int rows = BWD ? SPANEL_BW_ROWS : SPANEL_FW_ROWS;
  const int n = d.band > 0 ? min(d.n, r0 + d.band) : d.n; // a banded (leaf) block: the rows from r0 + band on are zero under these columns
  if (n <= r0 + (int)blockIdx.y * rows) return;
  const TL *Lm = base + d.a_off;
  double *x = y + d.x_off;
  const int lda = d.lda, tid = threadIdx.x;
  if (!BWD) {
    sx[tid] = x[col0 + tid];
    __syncthreads();
    const int r = r0 + blockIdx.y * 256 + tid;
    const TL *A = Lm + min(r, n - 1) + (int64_t)col0 * lda;
    double acc = 0.0;
    for (int k = 0; k < SSPAN; k += 32) { // thirty-two loads in flight per thread
      TL a[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) a[u] = A[(int64_t)(k + u) * lda];
#pragma unroll
      for (int u = 0; u < 32; ++u) acc += (double)a[u] * sx[k + u];
    }
    if (r < n) x[r] -= acc;
  } else {
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int PER = SPANEL_BW_ROWS / 64;
    const int row0 = r0 + blockIdx.y * SPANEL_BW_ROWS;
    double xa[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int i = row0 + lane + 64 * u;
      xa[u] = i < n ? x[i] : 0.0;
    }
    gather_columns<PER>(Lm + row0 + (int64_t)col0 * lda, lda, min(n - row0, SPANEL_BW_ROWS), SSPAN, xa, x + col0, lane, wave);
  }
}

// One step of the span chain of the few wide separators at the top of the tree (fp32 factor) in ONE launch.  Launch by launch a step was: solve span k (one
// workgroup, ~10 us), then its whole panel (all CUs, ~12 us) -- but the next span needs only SSPAN rows of that panel.  Roles by blockIdx.y:
//   0                 the solve of span k (span32_body; publishes x_k with agent-scope stores, then the separator's flag = gen)
//   1 .. STEP_NB      the SSPAN rows / columns the NEXT span needs of span k's panel: their L entries are requested first, then the workgroup waits for the
//                     flag (the producer has the lowest block index of the launch and waits for nobody; the spin is bounded all the same: it poisons its
//                     part of the vector with NaN if it gives up), reads x_k with agent-scope loads and adds its part atomically
//   STEP_NB + 1 ..    the rest of the PREVIOUS span's panel (forward: rows behind span k under span k - 1; backward: the rows behind span k into the columns of
//                     span k - 1): everything it reads was final before the launch, it runs beside the solve
// so the chain per step is solve + flag + 1/16 of a SSPAN x SSPAN block instead of solve + launch + panel.
#define STEP_NB 16
#define STEP_MAX_SEPS 8    /* k_solve_step: a 512-register workgroup per role */
#define STEP_MAX_SEPS_W 128 /* k_solve_stepw (explicit span inverses): ordinary workgroups; = CHOL_STEPW_MAX_SEPS of chol_kernels.h (flags, scratch, inverses) */
static_assert(STEP_MAX_SEPS_W == CHOL_STEPW_MAX_SEPS, "flags and scratch of the step launches are sized by chol_kernels.h");
// A level of banded LEAVES (fp32 factor): the whole triangle of a leaf by ONE workgroup in ONE launch.  A leaf's factor stays inside the envelope of A, and
// with a band of at most SSPAN rows the panel under a span is a corner of the NEXT span's rows only: span by span the workgroup solves the span out of
// registers (span32_body) and folds it into / gathers from those <= 256 rows itself.  Launch by launch the 512 leaves of 100^3 took 6 span launches (512
// workgroups of one per CU: two rounds each) and 5 panel launches per sweep: 1.1 of the 7.1 ms of a solve.  Inside the launch the vector is written with
// agent-scope stores and read with agent-scope loads (the gather's atomics are performed in L2: a plain load could hit a line the CU cached before them).
template <bool BWD, class TL> struct leaf_span;
template <bool BWD> struct leaf_span<BWD, float> {
  static __device__ __forceinline__ void run(const float *base, const chol_trsv_desc &d, const double *Wall, double *y, int col0, double *sx) { span32_body<BWD, true, true>(base, d, Wall, y, col0, sx); }
};
template <bool BWD> struct leaf_span<BWD, double> { // (the fp64 factor: the 64-column block chain of k_solve_trsv)
  static __device__ __forceinline__ void run(const double *base, const chol_trsv_desc &d, const double *Wall, double *y, int col0, double *sx)
  {
    __shared__ double sL[SNB][SNB + 1];
    __shared__ double sW[SNB / TS][TS * TS];
    trsv_body<BWD, double, true, true>(base, d, Wall, y, col0, sL, sW, sx);
  }
};
template <bool BWD, class TL>
__global__ __launch_bounds__(256) void k_solve_leaf32(const TL *__restrict__ base, const chol_trsv_desc *__restrict__ descs, const double *__restrict__ Wall,
                                                      double *__restrict__ y)
{
  __shared__ __attribute__((aligned(16))) double sx[SSPAN];
  const chol_trsv_desc d = descs[blockIdx.x];
  const int n = d.n, lda = d.lda, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nspan = (n + SSPAN - 1) / SSPAN, band = d.band > 0 ? d.band : n;
  const TL *Lm = base + d.a_off;
  double *x = y + d.x_off;
  for (int i = 0; i < nspan; ++i) {
    const int sp = BWD ? nspan - 1 - i : i, col0 = sp * SSPAN, r0 = col0 + SSPAN;
    const int nr = min(n, r0 + band) - r0; // rows under the span inside the band (<= SSPAN; <= 0: none)
    if (BWD && nr > 0) { // x(span) -= L(rows under it, span)^T x(rows): the rows are the head of the span solved before this one
      double xa[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) xa[u] = lane + 64 * u < nr ? gload<true>(&x[r0 + lane + 64 * u]) : 0.0;
      gather_columns<4>(Lm + r0 + (int64_t)col0 * lda, lda, nr, SSPAN, xa, x + col0, lane, wave);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    leaf_span<BWD, TL>::run(base, d, Wall, y, col0, sx);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!BWD && nr > 0) { // x(rows under the span) -= L(rows, span) x(span)
      sx[tid] = gload<true>(&x[col0 + tid]);
      __syncthreads();
      const TL *A = Lm + r0 + min(tid, nr - 1) + (int64_t)col0 * lda;
      double acc = 0.0;
      for (int k = 0; k < SSPAN; k += 64) { // (the span's registers are free again: sixty-four loads in flight, four rounds)
        TL a[64];
#pragma unroll
        for (int u = 0; u < 64; ++u) a[u] = A[(int64_t)(k + u) * lda];
#pragma unroll
        for (int u = 0; u < 64; ++u) acc += (double)a[u] * sx[k + u];
      }
      if (tid < nr) gstore<true>(&x[r0 + tid], gload<true>(&x[r0 + tid]) - acc);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
}

template <bool BWD, class TL> struct step_span;
template <bool BWD> struct step_span<BWD, float> {
  static __device__ __forceinline__ void run(const float *base, const chol_trsv_desc &d, const double *Wall, double *y, int col0, double *sx) { span32_body<BWD, true>(base, d, Wall, y, col0, sx); }
};
template <bool BWD> struct step_span<BWD, double> { // the fp64 factor's span: the 64-column block chain of k_solve_trsv
  static __device__ __forceinline__ void run(const double *base, const chol_trsv_desc &d, const double *Wall, double *y, int col0, double *sx)
  {
    __shared__ double sL[SNB][SNB + 1];
    __shared__ double sW[SNB / TS][TS * TS];
    trsv_body<BWD, double, true>(base, d, Wall, y, col0, sL, sW, sx);
  }
};
template <bool BWD, class TL>
__global__ __launch_bounds__(256) void k_solve_step(const TL *__restrict__ base, const chol_trsv_desc *__restrict__ descs, const double *__restrict__ Wall,
                                                    double *__restrict__ y, int col0, int *__restrict__ flags, int gen)
{
  __shared__ __attribute__((aligned(16))) double sx[SSPAN];
  const chol_trsv_desc d = descs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lda = d.lda, n = d.n;
  const int role = blockIdx.y;
  if (n <= col0) return; // (every role of a separator without this span)
  const TL *Lm = base + d.a_off;
  double *x = y + d.x_off;
  if (role == 0) {
    step_span<BWD, TL>::run(base, d, Wall, y, col0, sx);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&flags[blockIdx.x], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (role <= STEP_NB) {
    const int part = role - 1;
    bool ok = true;
    if (!BWD) { // rows col0 + SSPAN + 16 part .. + 15 under the columns of span k: thread (row, sixteenth of the columns)
      const int r = tid & 15, kp = tid >> 4, row = col0 + SSPAN + TS * part + r;
      if (col0 + SSPAN + TS * part >= n) return;
      const TL *A = Lm + min(row, n - 1) + (int64_t)(col0 + TS * kp) * lda;
      TL a[TS];
#pragma unroll
      for (int u = 0; u < TS; ++u) a[u] = A[(int64_t)u * lda];
      if (tid < 64) {
        int v = 0, it = 0;
        for (; it < (1 << 22); ++it) {
          v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&flags[blockIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          if (v == gen) break;
          __builtin_amdgcn_s_sleep(4);
        }
        if (tid == 0) ((int *)sx)[0] = v == gen;
      }
      __syncthreads();
      ok = ((volatile int *)sx)[0] != 0;
      __syncthreads();
      double acc = 0.0;
#pragma unroll
      for (int u = 0; u < TS; ++u) acc += (double)a[u] * gload<true>(&x[col0 + TS * kp + u]);
      sx[kp * TS + r] = acc;
      __syncthreads();
      if (tid < TS) {
        double sum = 0.0;
#pragma unroll
        for (int q = 0; q < TS; ++q) sum += sx[q * TS + tid];
        if (!ok) sum = __builtin_nan("");
        if (row < n) unsafeAtomicAdd(&x[row], -sum);
      }
    } else { // the rows of span k into the columns col0 - SSPAN + 16 part .. + 15 of span k - 1: wave w the rows 64 w .. 64 w + 63
      if (col0 == 0) return;
      const int c0 = col0 - SSPAN + TS * part, ns = min(n - col0, SSPAN), row = 64 * wave + lane;
      const TL *A = Lm + col0 + min(row, ns - 1) + (int64_t)c0 * lda;
      TL a[TS];
#pragma unroll
      for (int q = 0; q < TS; ++q) a[q] = A[(int64_t)q * lda];
      if (tid < 64) {
        int v = 0, it = 0;
        for (; it < (1 << 22); ++it) {
          v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&flags[blockIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          if (v == gen) break;
          __builtin_amdgcn_s_sleep(4);
        }
        if (tid == 0) ((int *)sx)[0] = v == gen;
      }
      __syncthreads();
      ok = ((volatile int *)sx)[0] != 0;
      const double xr = row < ns ? gload<true>(&x[col0 + row]) : 0.0;
      double acc[TS];
#pragma unroll
      for (int q = 0; q < TS; ++q) acc[q] = (double)a[q] * xr;
      double sum = wave_sum16(acc, lane);
      if (!ok) sum = __builtin_nan("");
      if ((lane & 15) < 4) unsafeAtomicAdd(&x[c0 + wave_sum16_col(lane)], -sum);
    }
    return;
  }
  // the rest of the previous span's panel
  if (col0 == 0) return;
  const int chunk = role - STEP_NB - 1, pc0 = col0 - SSPAN, r0 = col0 + SSPAN;
  if (!BWD) {
    if (n <= r0 + chunk * SPANEL_FW_ROWS) return;
    sx[tid] = x[pc0 + tid];
    __syncthreads();
    const int r = r0 + chunk * SPANEL_FW_ROWS + tid;
    const TL *A = Lm + min(r, n - 1) + (int64_t)pc0 * lda;
    double acc = 0.0;
    for (int k = 0; k < SSPAN; k += 32) {
      TL a[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) a[u] = A[(int64_t)(k + u) * lda];
#pragma unroll
      for (int u = 0; u < 32; ++u) acc += (double)a[u] * sx[k + u];
    }
    if (r < n) unsafeAtomicAdd(&x[r], -acc); // (the rows of the next span also receive the middle role's sums)
  } else {
    constexpr int PER = SPANEL_BW_ROWS / 64;
    const int row0 = r0 + chunk * SPANEL_BW_ROWS;
    if (n <= row0) return;
    double xa[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int i = row0 + lane + 64 * u;
      xa[u] = i < n ? x[i] : 0.0;
    }
    gather_columns<PER>(Lm + row0 + (int64_t)pc0 * lda, lda, min(n - row0, SPANEL_BW_ROWS), SSPAN, xa, x + pc0, lane, wave);
  }
}

// ---- the span chain of the wide top separators with EXPLICIT inverses of the 256 x 256 diagonal spans (round 4) ----
// k_solve_step's chain per span is the span solve itself: 10-15 us of sixteen dependent 16-column block steps in one workgroup.  With inv(L_kk) at hand
// (k_solve_inv256, once per solve: the factor's 16x16 inverses doubled up block column by block column on the matrix cores) x_k = inv(L_kk) r_k is a
// 256 x 256 triangular matrix-vector product with NO chain: sixteen workgroups take sixteen rows (forward) / columns (backward) each, reading a right-hand
// side that was final before the launch.  Roles by blockIdx.y:
//   0 .. 15    part p of x_k into the separator's scratch vector xt (the right-hand side must stay until every part has read it), then flag p = gen
//   16 .. 31   wait for all sixteen flags; copy part p of xt into the vector; the SSPAN rows / columns the next span needs of span k's panel (as k_solve_step's
//              middle role, with x_k from xt)
//   32 ..      the rest of the previous span's panel (as k_solve_step)
// W256: [column][row] (column-major, 256 x 256 doubles per span), lower triangle in 16 x 16 blocks; blocks above the diagonal are never written or read.
template <class TL>
__global__ __launch_bounds__(64) void k_solve_inv256(const TL *__restrict__ base, const chol_trsv_desc *__restrict__ descs, const double *__restrict__ W16all,
                                                     double *__restrict__ W256, int nspan_max)
{ // one wave per (separator, span, block column j): X(i, j) = -inv(L_ii) sum_{k = j .. i-1} L(i, k) X(k, j), X(j, j) = inv(L_jj); the column's blocks stay in the
  // wave's registers in the fp64 MFMA result layout (register q of lane (n, g): row g + 4 q, column n), which is also the layout of the next product's second
  // operand with k = g + 4 s in instruction s
  const chol_trsv_desc d = descs[blockIdx.x];
  const int sp = blockIdx.y, j = blockIdx.z, col0 = sp * SSPAN;
  if (d.n <= col0) return;
  const int ns = min(d.n - col0, SSPAN), nb = (ns + TS - 1) / TS;
  if (j >= nb) return;
  const int lane = threadIdx.x, n16 = lane & 15, g = lane >> 4, lda = d.lda;
  const TL *Lm = base + d.a_off + col0 + (int64_t)col0 * lda;
  const double *W16 = W16all + d.dinv_off + (int64_t)(col0 / TS) * TS * TS; // W16[b * 256 + c * 16 + r] = inv(L_bb)(r, c)
  double *out = W256 + ((int64_t)blockIdx.x * nspan_max + sp) * (SSPAN * SSPAN) + (int64_t)(TS * j + n16) * SSPAN + g;
  d4 X[SSPAN / TS];
#pragma unroll
  for (int i = 0; i < SSPAN / TS; ++i) X[i] = (d4){ 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
  for (int i = 0; i < SSPAN / TS; ++i) {
    if (i == j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) X[i][q] = W16[i * TS * TS + n16 * TS + g + 4 * q];
    } else if (i > j && i < nb) {
      d4 S = { 0.0, 0.0, 0.0, 0.0 };
      const int row = min(TS * i + n16, ns - 1); // (first operand: lane = row of L(i, k))
#pragma unroll
      for (int k = 0; k < SSPAN / TS; ++k) {
        if (k >= j && k < i) {
#pragma unroll
          for (int s = 0; s < 4; ++s) S = __builtin_amdgcn_mfma_f64_16x16x4f64((double)Lm[row + (int64_t)(TS * k + g + 4 * s) * lda], X[k][s], S, 0, 0, 0);
        }
      }
      d4 D = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
      for (int s = 0; s < 4; ++s) D = __builtin_amdgcn_mfma_f64_16x16x4f64(W16[i * TS * TS + (g + 4 * s) * TS + n16], S[s], D, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) X[i][q] = -D[q];
    }
    if (i >= j && i < nb) {
#pragma unroll
      for (int q = 0; q < 4; ++q) out[TS * i + 4 * q] = X[i][q];
    }
  }
}
// every one of the sixteen flags equals gen (bounded; false: gave up)
__device__ __forceinline__ bool wait_parts16(const int *f, int gen, int tid, int *sflag)
{
  if (tid < 64) {
    bool ok = false;
    for (int it = 0; it < (1 << 22); ++it) {
      const int v = tid < STEP_NB ? __hip_atomic_load(&f[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : gen;
      if (__all(v == gen)) { ok = true; break; }
      __builtin_amdgcn_s_sleep(4);
    }
    if (tid == 0) *sflag = ok;
  }
  __syncthreads();
  const bool r = *(volatile int *)sflag != 0;
  __syncthreads();
  return r;
}
template <bool BWD, class TL>
__global__ __launch_bounds__(256) void k_solve_stepw(const TL *__restrict__ base, const chol_trsv_desc *__restrict__ descs, const double *__restrict__ W256,
                                                     int nspan_max, double *__restrict__ y, double *__restrict__ xtmp, int col0, int *__restrict__ flags, int gen)
{
  __shared__ __attribute__((aligned(16))) double sx[SSPAN];
  const chol_trsv_desc d = descs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lda = d.lda, n = d.n;
  const int role = blockIdx.y;
  if (n <= col0) return; // (every role of a separator without this span)
  const TL *Lm = base + d.a_off;
  double *x = y + d.x_off, *xt = xtmp + blockIdx.x * SSPAN;
  int *fl = flags + blockIdx.x * STEP_NB;
  const int ns = min(n - col0, SSPAN);
  if (role < STEP_NB) { // part `role` of x_k = inv(L_kk) r_k (forward: sixteen rows) / inv(L_kk)^T r_k (backward: sixteen columns)
    const int p = role, r = tid & 15, kp = tid >> 4;
    const double *W = W256 + ((int64_t)blockIdx.x * nspan_max + col0 / SSPAN) * (SSPAN * SSPAN);
    if (TS * p < ns) {
      double acc = 0.0;
      if (!BWD) { // thread (row 16 p + r, columns 16 kp .. + 15), the blocks left of the diagonal one and it
        if (kp <= p) {
          double w[TS], v[TS];
#pragma unroll
          for (int u = 0; u < TS; ++u) { w[u] = W[(int64_t)(TS * kp + u) * SSPAN + TS * p + r]; v[u] = x[col0 + min(TS * kp + u, ns - 1)]; }
#pragma unroll // (a ragged last block: the columns past the span are selected away, not multiplied by their zeros -- what lies behind the vector may be NaN)
          for (int u = 0; u < TS; ++u) acc += TS * kp + u < ns ? w[u] * v[u] : 0.0;
        }
        sx[kp * TS + r] = acc;
        __syncthreads();
        if (tid < TS) {
          double sum = 0.0;
#pragma unroll
          for (int q = 0; q < TS; ++q) sum += sx[q * TS + tid];
          if (TS * p + tid < ns) gstore<true>(&xt[TS * p + tid], sum);
        }
      } else { // thread (column 16 p + kp, rows 16 g + r of the blocks g >= p): sum over the sixteen lanes of a row
        const int c = TS * p + kp;
        if (c < ns) {
          const double *Wc = W + (int64_t)c * SSPAN;
          double w[SSPAN / TS], v[SSPAN / TS]; // unconditional loads (a predicate per load is a branch and a wait per load); what lies above the diagonal block
#pragma unroll                          // row or past the span is never written: selected away, not multiplied by zero
          for (int gq = 0; gq < SSPAN / TS; ++gq) { w[gq] = Wc[TS * gq + r]; v[gq] = x[col0 + min(TS * gq + r, ns - 1)]; }
#pragma unroll
          for (int gq = 0; gq < SSPAN / TS; ++gq) acc += (gq >= p && TS * gq + r < ns) ? w[gq] * v[gq] : 0.0;
        }
        acc = row_sum16(acc);
        if (r == 0 && c < ns) gstore<true>(&xt[c], acc);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&fl[p], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (role < 2 * STEP_NB) {
    const int part = role - STEP_NB;
    if (!BWD) { // rows col0 + SSPAN + 16 part .. + 15 under the columns of span k: thread (row, sixteenth of the columns)
      const int r = tid & 15, kp = tid >> 4, row = col0 + SSPAN + TS * part + r;
      const bool rows = col0 + SSPAN + TS * part < n;
      TL a[TS];
      if (rows) {
        const TL *A = Lm + min(row, n - 1) + (int64_t)(col0 + TS * kp) * lda;
#pragma unroll
        for (int u = 0; u < TS; ++u) a[u] = A[(int64_t)u * lda];
      }
      const bool ok = wait_parts16(fl, gen, tid, (int *)sx);
      if (tid < TS && TS * part + tid < ns) gstore<true>(&x[col0 + TS * part + tid], ok ? gload<true>(&xt[TS * part + tid]) : __builtin_nan(""));
      if (!rows) return;
      double acc = 0.0;
#pragma unroll
      for (int u = 0; u < TS; ++u) acc += (double)a[u] * gload<true>(&xt[TS * kp + u]);
      sx[kp * TS + r] = acc;
      __syncthreads();
      if (tid < TS) {
        double sum = 0.0;
#pragma unroll
        for (int q = 0; q < TS; ++q) sum += sx[q * TS + tid];
        if (!ok) sum = __builtin_nan("");
        if (row < n) unsafeAtomicAdd(&x[row], -sum);
      }
    } else { // the rows of span k into the columns col0 - SSPAN + 16 part .. + 15 of span k - 1: wave w the rows 64 w .. 64 w + 63
      const int c0 = col0 - SSPAN + TS * part, row = 64 * wave + lane;
      TL a[TS];
      if (col0 > 0) {
        const TL *A = Lm + col0 + min(row, ns - 1) + (int64_t)c0 * lda;
#pragma unroll
        for (int q = 0; q < TS; ++q) a[q] = A[(int64_t)q * lda];
      }
      const bool ok = wait_parts16(fl, gen, tid, (int *)sx);
      if (tid < TS && TS * part + tid < ns) gstore<true>(&x[col0 + TS * part + tid], ok ? gload<true>(&xt[TS * part + tid]) : __builtin_nan(""));
      if (col0 == 0) return;
      const double xr = row < ns ? gload<true>(&xt[row]) : 0.0;
      double acc[TS];
#pragma unroll
      for (int q = 0; q < TS; ++q) acc[q] = (double)a[q] * xr;
      double sum = wave_sum16(acc, lane);
      if (!ok) sum = __builtin_nan("");
      if ((lane & 15) < 4) unsafeAtomicAdd(&x[c0 + wave_sum16_col(lane)], -sum);
    }
    return;
  }
  // the rest of the previous span's panel, in (row chunk) x (half of the span's columns) pieces: the launch is as long as its longest workgroup
  if (col0 == 0) return;
  const int chunk = (role - 2 * STEP_NB) >> 1, kh = (role - 2 * STEP_NB) & 1, pc0 = col0 - SSPAN + (SSPAN / 2) * kh, r0 = col0 + SSPAN;
  if (!BWD) {
    if (n <= r0 + chunk * SPANEL_FW_ROWS) return;
    if (tid < SSPAN / 2) sx[tid] = x[pc0 + tid];
    __syncthreads();
    const int r = r0 + chunk * SPANEL_FW_ROWS + tid;
    const TL *A = Lm + min(r, n - 1) + (int64_t)pc0 * lda;
    double acc = 0.0;
    for (int k = 0; k < SSPAN / 2; k += 32) {
      TL a[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) a[u] = A[(int64_t)(k + u) * lda];
#pragma unroll
      for (int u = 0; u < 32; ++u) acc += (double)a[u] * sx[k + u];
    }
    if (r < n) unsafeAtomicAdd(&x[r], -acc);
  } else {
    constexpr int PER = SPANEL_BW_ROWS / 64;
    const int row0 = r0 + chunk * SPANEL_BW_ROWS;
    if (n <= row0) return;
    double xa[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int i = row0 + lane + 64 * u;
      xa[u] = i < n ? x[i] : 0.0;
    }
    gather_columns<PER>(Lm + row0 + (int64_t)pc0 * lda, lda, min(n - row0, SPANEL_BW_ROWS), SSPAN / 2, xa, x + pc0, lane, wave);
  }
}

// forward: y_anc[rows] -= A(rows, cols) y_s for one (row chunk, column chunk) of the block A = (anc, s); y_s staged through LDS.  items = (block, first
// row, first column) triples: the column chunks (CHOL_SOLVE_COLS) give the few tall blocks of the top levels enough workgroups to fill the chip
template <class TL>
__global__ __launch_bounds__(256) void k_solve_gemv_fwd(const TL *__restrict__ base, const chol_gemv_desc *__restrict__ blocks, const int *__restrict__ items,
                                                        double *__restrict__ y)
{
  __shared__ double sx[256];
  const chol_gemv_desc d = blocks[items[4 * blockIdx.x]];
  const int r = items[4 * blockIdx.x + 1] + threadIdx.x, c0 = max(items[4 * blockIdx.x + 2], items[4 * blockIdx.x + 3]), c1 = min(d.n, items[4 * blockIdx.x + 2] + CHOL_SOLVE_COLS);
  if (c0 >= c1) return; // (the rows of a leaf's panel are zero in front of their first entry of A: items[.. + 3])
  const TL *A = base + d.a_off + min(r, d.m - 1);
  const double *xs = y + d.y_off; // the separator's (already solved) part
  double acc = 0.0;
  for (int k0 = c0; k0 < c1; k0 += 256) {
    const int kb = min(256, c1 - k0);
    __syncthreads();
    if ((int)threadIdx.x < kb) sx[threadIdx.x] = xs[k0 + threadIdx.x];
    __syncthreads();
    const TL *Ak = A + (int64_t)k0 * d.lda;
    int k = 0;
    for (; k + 32 <= kb; k += 32) { // thirty-two loads in flight per thread: with sixteen, 32 waves x 256 B x (the ~60 % of lanes a leaf's blocks fill) is
      TL a[32];                     // under the ~90 KB a CU must keep in flight for its share of HBM
#pragma unroll
      for (int u = 0; u < 32; ++u) a[u] = Ak[(int64_t)(k + u) * d.lda];
#pragma unroll
      for (int u = 0; u < 32; ++u) acc += (double)a[u] * sx[k + u];
    }
    for (; k + 16 <= kb; k += 16) {
      TL a[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) a[u] = Ak[(int64_t)(k + u) * d.lda];
#pragma unroll
      for (int u = 0; u < 16; ++u) acc += (double)a[u] * sx[k + u];
    }
    for (; k + 8 <= kb; k += 8) {
      double a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = (double)Ak[(int64_t)(k + u) * d.lda];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += a[u] * sx[k + u];
    }
    for (; k < kb; ++k) acc += (double)Ak[(int64_t)k * d.lda] * sx[k];
  }
  if (r < d.m) unsafeAtomicAdd(&y[d.x_off + r], -acc);
}

// backward: y_s[c] -= sum over the rows of the separator's panel below its diagonal block of A(i, c) y_anc(i).  One workgroup = 64 columns of ONE separator
// and a run of its row-run descriptors (q0 .. q1: all of them, unless the level has too few separators to fill the chip): the sums stay in registers
// over every block the separator has into its ancestors and reach y with ONE atomic per column per workgroup.  (Per (block, 512-row chunk) the leaf
// level of 100^3 issued 7 M fp64 atomics -- several row chunks and ~8 ancestor blocks per column, device-scope read-modify-writes that bounded the level at
// 1.47 ms whatever the reduction cost: swapping the shuffles for the swap-based sums above without merging changed nothing, more chunks made it worse.)
template <class TL>
__global__ __launch_bounds__(256) void k_solve_gather_bwd(const TL *__restrict__ base, const chol_gemv_desc *__restrict__ blocks, const int *__restrict__ items,
                                                          double *__restrict__ y)
{
  const int q0 = items[4 * blockIdx.x], q1 = items[4 * blockIdx.x + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cw = items[4 * blockIdx.x + 2] + 16 * wave, n = blocks[q0].n;
  if (cw >= n) return; // no barrier in this kernel
  const int ncol = min(16, n - cw);
  constexpr int PER = 4;
  double acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0;
  for (int b = q0; b < q1; ++b) {
    const chol_gemv_desc d = blocks[b];
    if (d.c_lo >= cw + 16) continue; // the run's rows are zero in the wave's columns
    const TL *A = base + d.a_off + (int64_t)cw * d.lda;
    const double *xs = y + d.x_off;
    for (int r0 = 0; r0 < d.m; r0 += 64 * PER) {
      const int mrows = min(d.m - r0, 64 * PER);
      int ro[PER];
      double xa[PER];
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        ro[u] = r0 + min(lane + 64 * u, mrows - 1);
        xa[u] = (lane + 64 * u < mrows) ? xs[ro[u]] : 0.0;
      }
      TL a[16][PER];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const TL *Ac = A + (int64_t)min(q, ncol - 1) * d.lda;
#pragma unroll
        for (int u = 0; u < PER; ++u) a[q][u] = Ac[ro[u]]; // unconditional (clamped rows repeat the last one, x = 0 there): a predicate here serialises the loads
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
#pragma unroll
        for (int u = 0; u < PER; ++u) acc[q] += (double)a[q][u] * xa[u];
      }
    }
  }
  const double sum = wave_sum16(acc, lane);
  const int c = wave_sum16_col(lane);
  if ((lane & 15) < 4 && c < ncol) unsafeAtomicAdd(&y[blocks[q0].y_off + cw + c], -sum);
}

template <class TL> static int launch_solve_dinv_t(const TL *base, const chol_trsv_desc *descs, int n, int max_n, double *W, hipStream_t st)
{
  if (n <= 0 || max_n <= 0) return 0;
  hipLaunchKernelGGL(k_solve_dinv<TL>, dim3(n, (max_n + TS - 1) / TS), dim3(64), 0, st, base, descs, W);
  return (int)hipGetLastError();
}
#ifndef SOLVE_SPAN32
#define SOLVE_SPAN32 1 /* 0: the fp32 factor's spans through k_solve_trsv<., float> like the fp64 factor's (A/B) */
#endif
template <bool BWD> static void launch_span(const double *base, const chol_trsv_desc *descs, int n, const double *W, double *y, int col0, hipStream_t st)
{
  hipLaunchKernelGGL((k_solve_trsv<BWD, double>), dim3(n), dim3(256), 0, st, base, descs, W, y, col0);
}
template <bool BWD> static void launch_span(const float *base, const chol_trsv_desc *descs, int n, const double *W, double *y, int col0, hipStream_t st)
{
  if (SOLVE_SPAN32) hipLaunchKernelGGL((k_solve_span32<BWD>), dim3(n), dim3(256), 0, st, base, descs, W, y, col0);
  else hipLaunchKernelGGL((k_solve_trsv<BWD, float>), dim3(n), dim3(256), 0, st, base, descs, W, y, col0);
}
#ifndef SOLVE_STEP32
#define SOLVE_STEP32 1 /* 0: span and panel launch by launch at every level (A/B) */
#endif
template <class TL>
static bool launch_steps(const TL *base, const chol_trsv_desc *descs, int n, int max_n, const double *W, double *y, int backward, int *flags, int *gen, const double *W256,
                         double *xt, hipStream_t st)
{ // the few wide separators of a top level: one launch per span step (k_solve_stepw with the spans' explicit inverses, else k_solve_step)
  if (!SOLVE_STEP32 || !flags || n > (W256 ? STEP_MAX_SEPS_W : STEP_MAX_SEPS) || max_n <= SSPAN) return false;
  if (W256) {
    const int nspan = (max_n + SSPAN - 1) / SSPAN;
    for (int i = 0; i < nspan; i++) {
      const int sp = backward ? nspan - 1 - i : i, col0 = sp * SSPAN;
      const int behind = sp > 0 ? max_n - (col0 + SSPAN) : 0;
      const int rows = backward ? SPANEL_BW_ROWS : SPANEL_FW_ROWS;
      const dim3 grid(n, 2 * STEP_NB + (behind > 0 ? 2 * ((behind + rows - 1) / rows) : 0));
      *gen = *gen == 0x7fffffff ? 1 : *gen + 1;
      if (backward) hipLaunchKernelGGL((k_solve_stepw<true, TL>), grid, dim3(256), 0, st, base, descs, W256, nspan, y, xt, col0, flags, *gen);
      else hipLaunchKernelGGL((k_solve_stepw<false, TL>), grid, dim3(256), 0, st, base, descs, W256, nspan, y, xt, col0, flags, *gen);
    }
    return true;
  }
  const int nspan = (max_n + SSPAN - 1) / SSPAN;
  for (int i = 0; i < nspan; i++) {
    const int sp = backward ? nspan - 1 - i : i, col0 = sp * SSPAN;
    const int behind = sp > 0 ? max_n - (col0 + SSPAN) : 0; // rows behind span sp: the previous span's rest
    const int rows = backward ? SPANEL_BW_ROWS : SPANEL_FW_ROWS;
    const int nrest = behind > 0 ? (behind + rows - 1) / rows : 0;
    const bool mid = backward ? sp > 0 : max_n > col0 + SSPAN;
    const dim3 grid(n, nrest > 0 ? 1 + STEP_NB + nrest : mid ? 1 + STEP_NB : 1);
    *gen = *gen == 0x7fffffff ? 1 : *gen + 1;
    if (backward) hipLaunchKernelGGL((k_solve_step<true, TL>), grid, dim3(256), 0, st, base, descs, W, y, col0, flags, *gen);
    else hipLaunchKernelGGL((k_solve_step<false, TL>), grid, dim3(256), 0, st, base, descs, W, y, col0, flags, *gen);
  }
  return true;
}
#ifndef SOLVE_LEAF32
#define SOLVE_LEAF32 1 /* 0: banded leaves span by span, launch by launch (A/B) */
#endif
template <class TL> static bool launch_leaves(const TL *base, const chol_trsv_desc *descs, int n, const double *W, double *y, int backward, hipStream_t st)
{
  if (!SOLVE_LEAF32) return false;
  if (backward) hipLaunchKernelGGL((k_solve_leaf32<true, TL>), dim3(n), dim3(256), 0, st, base, descs, W, y);
  else hipLaunchKernelGGL((k_solve_leaf32<false, TL>), dim3(n), dim3(256), 0, st, base, descs, W, y);
  return true;
}
template <class TL>
static int launch_solve_trsv_t(const TL *base, const chol_trsv_desc *descs, int n, int max_n, int max_under, const double *W, double *y, int backward, int *flags, int *gen,
                               const double *W256, double *xt, hipStream_t st)
{ // all separators of a level; wide ones in spans of SSPAN columns: diagonal span by one workgroup each, the rows below by all CUs (max_under: the most
  // rows any separator of the level has to read under a span -- its band if it is a leaf; negative: every separator of the level is banded within one span)
  if (n <= 0) return 0;
  if (max_under < 0) {
    if (max_n > SSPAN && launch_leaves(base, descs, n, W, y, backward, st)) return (int)hipGetLastError();
    max_under = -max_under;
  }
  if (launch_steps(base, descs, n, max_n, W, y, backward, flags, gen, W256, xt, st)) return (int)hipGetLastError();
  const int nspan = (max_n + SSPAN - 1) / SSPAN;
  for (int i = 0; i < nspan; i++) {
    const int sp = backward ? nspan - 1 - i : i, col0 = sp * SSPAN;
    const int below = min(max_n - (col0 + SSPAN), max_under); // rows under the span in the widest separator
    if (backward) {
      if (below > 0) hipLaunchKernelGGL((k_solve_panel<true, TL>), dim3(n, (below + SPANEL_BW_ROWS - 1) / SPANEL_BW_ROWS), dim3(256), 0, st, base, descs, y, col0);
      launch_span<true>(base, descs, n, W, y, col0, st);
    } else {
      launch_span<false>(base, descs, n, W, y, col0, st);
      if (below > 0) hipLaunchKernelGGL((k_solve_panel<false, TL>), dim3(n, (below + 255) / 256), dim3(256), 0, st, base, descs, y, col0);
    }
  }
  return (int)hipGetLastError();
}
template <class TL> static int launch_solve_inv256_t(const TL *base, const chol_trsv_desc *descs, int n, int max_n, const double *W16, double *W256, hipStream_t st)
{
  if (n <= 0 || max_n <= SSPAN) return 0;
  const int nspan = (max_n + SSPAN - 1) / SSPAN;
  hipLaunchKernelGGL(k_solve_inv256<TL>, dim3(n, nspan, SSPAN / TS), dim3(64), 0, st, base, descs, W16, W256, nspan);
  return (int)hipGetLastError();
}
template <class TL> static int launch_solve_offdiag_t(const TL *base, const chol_gemv_desc *blocks, const int *items, int n_items, double *y, int backward, hipStream_t st)
{
  if (n_items <= 0) return 0;
  if (backward) hipLaunchKernelGGL(k_solve_gather_bwd<TL>, dim3(n_items), dim3(256), 0, st, base, blocks, items, y);
  else hipLaunchKernelGGL(k_solve_gemv_fwd<TL>, dim3(n_items), dim3(256), 0, st, base, blocks, items, y);
  return (int)hipGetLastError();
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
  hipLaunchKernelGGL(k_potrf_rr, dim3(n), dim3(RR_THREADS), 0, st, base, ws, descs, info);
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
int chol_launch_potrf_trsm(double *base, double *ws, const chol_potrf_desc *pdescs, int n_potrf, const chol_trsm_desc *tdescs, int n_trsm,
                           const chol_upd_task *tasks, const chol_upd_src *srcs, int n_task,
                           int *info, int *progress, int progress_base, int *done, int done_target, hipStream_t st)
{ // fused launch: POTRF workgroups first (lowest block indices), then groups of three strips, then update workgroups
  if (n_potrf <= 0) return 0;
  int n_upd_wg = (n_task + 2) / 3;
  if (n_upd_wg > 1024) n_upd_wg = 1024;
  hipLaunchKernelGGL(k_potrf_trsm, dim3(n_potrf + (n_trsm + 2) / 3 + n_upd_wg), dim3(RR_THREADS), 0, st, base, ws, pdescs, n_potrf, tdescs, n_trsm,
                     tasks, srcs, n_task, n_upd_wg, info, progress, progress_base, done, done_target);
  return (int)hipGetLastError();
}
int chol_launch_program(double *base, double *ws, const chol_job *jobs, int njobs, const chol_wait *waits, const chol_potrf_desc *pdescs, const chol_trsm_desc *tdescs,
                        const chol_upd_task *tasks, const chol_upd_src *srcs, const chol_ext *exts, int *ctr, const int *ctr_total, int epoch, int *head, int head_base,
                        int grid, int *info, int *info_next, unsigned long long *trace, hipStream_t st)
{
  if (njobs <= 0) return 0;
  // the product kernel carries no stamp code (k_program<false>); the per-job stamps of cholamd_device_program_trace run the diagnostic instance
  if (trace) hipLaunchKernelGGL(k_program<true>, dim3(grid), dim3(RR_THREADS), 0, st, base, ws, jobs, njobs, waits, pdescs, tdescs, tasks, srcs, exts, ctr, ctr_total, epoch, head, head_base, info, info_next, trace);
  else hipLaunchKernelGGL(k_program<false>, dim3(grid), dim3(RR_THREADS), 0, st, base, ws, jobs, njobs, waits, pdescs, tdescs, tasks, srcs, exts, ctr, ctr_total, epoch, head, head_base, info, info_next, trace);
  return (int)hipGetLastError();
}
int chol_launch_trsm_w(double *base, const double *ws, const chol_trsm_desc *descs, int n, hipStream_t st)
{ // strips of pivot blocks up to CHOL_TRSM_W_MAXN columns, one wave each; every aligned group of four descriptors shares one block
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_trsm_w, dim3((n + 3) / 4), dim3(256), 0, st, base, ws, descs, n);
  return (int)hipGetLastError();
}
int chol_launch_trsm_wt(double *base, const double *ws, const chol_trsm_desc *descs, int n, hipStream_t st)
{ // strips of pivot blocks up to CHOL_TRSM_WT_MAXN columns, one wave each; every aligned group of CHOL_TRSM_WT_GROUP descriptors shares one block
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_trsm_wt, dim3((n + TT_WAVES - 1) / TT_WAVES), dim3(64 * TT_WAVES), 0, st, base, ws, descs, n);
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
  const int per_xcd = (ntask + 7) / 8;
  hipLaunchKernelGGL(k_update, dim3(per_xcd * 8), dim3(256), 0, st, base, tasks, srcs, ntask, per_xcd);
  return (int)hipGetLastError();
}
int chol_launch_update_mt(double *base, const chol_upd_task *tasks, const chol_upd_src *srcs, int ntask, int64_t arena_elems, hipStream_t st)
{ // arena_elems: doubles in the arena behind `base` (0: unknown -- edge tiles then stage their operands through registers)
  if (ntask <= 0) return 0;
  const int per_xcd = (ntask + 7) / 8;
  hipLaunchKernelGGL(k_update_mt, dim3(per_xcd * 8), dim3(256), 0, st, base, tasks, srcs, ntask, per_xcd, arena_elems);
  return (int)hipGetLastError();
}
int chol_launch_permute(const double *in, const int *perm, double *out, int n, int inverse, hipStream_t st)
{
  if (n <= 0) return 0;
  if (inverse) hipLaunchKernelGGL(k_permute_out, dim3((n + 255) / 256), dim3(256), 0, st, in, perm, out, n);
  else hipLaunchKernelGGL(k_permute_in, dim3((n + 255) / 256), dim3(256), 0, st, in, perm, out, n);
  return (int)hipGetLastError();
}
int chol_launch_solve_dinv(const double *base, const chol_trsv_desc *descs, int n, int max_n, double *W, hipStream_t st) { return launch_solve_dinv_t(base, descs, n, max_n, W, st); }
int chol_launch_solve_trsv(const double *base, const chol_trsv_desc *descs, int n, int max_n, int max_under, const double *W, double *y, int backward, int *flags, int *gen, const double *W256, double *xt, hipStream_t st) { return launch_solve_trsv_t(base, descs, n, max_n, max_under, W, y, backward, flags, gen, W256, xt, st); }
int chol_launch_solve_inv256(const double *base, const chol_trsv_desc *descs, int n, int max_n, const double *W16, double *W256, hipStream_t st) { return launch_solve_inv256_t(base, descs, n, max_n, W16, W256, st); }
int chol32_launch_solve_inv256(const float *base, const chol_trsv_desc *descs, int n, int max_n, const double *W16, double *W256, hipStream_t st) { return launch_solve_inv256_t(base, descs, n, max_n, W16, W256, st); }
int chol_launch_solve_offdiag(const double *base, const chol_gemv_desc *blocks, const int *items, int n_items, double *y, int backward, hipStream_t st) { return launch_solve_offdiag_t(base, blocks, items, n_items, y, backward, st); }
int chol32_launch_solve_dinv(const float *base, const chol_trsv_desc *descs, int n, int max_n, double *W, hipStream_t st) { return launch_solve_dinv_t(base, descs, n, max_n, W, st); }
int chol32_launch_solve_trsv(const float *base, const chol_trsv_desc *descs, int n, int max_n, int max_under, const double *W, double *y, int backward, int *flags, int *gen, const double *W256, double *xt, hipStream_t st) { return launch_solve_trsv_t(base, descs, n, max_n, max_under, W, y, backward, flags, gen, W256, xt, st); }
int chol32_launch_solve_offdiag(const float *base, const chol_gemv_desc *blocks, const int *items, int n_items, double *y, int backward, hipStream_t st) { return launch_solve_offdiag_t(base, blocks, items, n_items, y, backward, st); }
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
