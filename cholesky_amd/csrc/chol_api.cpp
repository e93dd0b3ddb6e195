// C ABI glue of libcholamd: device objects, the level schedule, the task-level (fused_*) and the
// BLAS-level entry points.  Compiled with hipcc; everything exported is extern "C" (include/cholamd.h).
// There is no CPU fallback: every compute entry point needs a HIP device and fails loudly otherwise.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "chol_kernels.h"
#include "chol_plan.h"
#include "cholamd.h"

#define HIPCHK(call)                                                                                  \
  do {                                                                                                \
    hipError_t e_ = (call);                                                                           \
    if (e_ != hipSuccess) {                                                                           \
      chol_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);     \
      return (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? CHOLAMD_ERR_NO_DEVICE : CHOLAMD_ERR_HIP; \
    }                                                                                                 \
  } while (0)

struct level_dev {
  int n_potrf = 0, n_trsm = 0, n_task = 0, n_src = 0;
  std::vector<chol_phase> phase; // launches of the level in order (big pivots are factored in column blocks)
  chol_potrf_desc *potrf = nullptr;
  chol_trsm_desc *trsm = nullptr;
  chol_upd_task *task = nullptr, *task_mt = nullptr;
  chol_upd_src *src = nullptr;
  std::vector<chol_bcast> bcast;  // distributed top levels: the column blocks a phase of kind 6 broadcasts
};
struct solve_dev {
  int n_trsv = 0, n_grp = 0, n_fw = 0, n_bw = 0;
  chol_trsv_desc *trsv = nullptr;
  chol_gemv_desc *fw = nullptr, *bw = nullptr;
  int *grp_start = nullptr, *grp_rows = nullptr, *bw_start = nullptr;
  int n_ifw = 0, n_ibw = 0, max_n = 0, max_under = 0;
  int64_t w256_off = -1; // this level's explicit span inverses in cholamd_device::w256 (levels of at most 8 separators wider than a span); -1: none
  int *ifw = nullptr, *ibw = nullptr;
};
struct timed_launch { hipEvent_t a, b; int kind; };

struct cholamd_device {
  struct vmm_arena { void *va; size_t total; std::vector<hipMemGenericAllocationHandle_t> handles; };
  std::vector<vmm_arena> vmm; // sharded arenas of this rank (cholamd_device_alloc_arena)

  const cholamd_plan *plan = nullptr;
  int dev = 0, rank = 0, world = 1;
  std::vector<level_dev> lv;
  std::vector<solve_dev> sv;
  bool solve_ready = false;
  int solve_rank = 0, solve_world = 1; // the partition the solve lists were built for
  int *zr_sub = nullptr; int n_zr_sub = 0; // distributed solve: (offset, length) ranges of the permuted vector this rank starts from zero in (other ranks' subtrees; the shared top on ranks other than 0)
  double *ws = nullptr;
  double *ws_solve = nullptr; // 16x16 inverses of the diagonal blocks of the arena being solved with
  int *step_flags = nullptr;  // flags of the step launches of the wide top separators' span chains (k_solve_step): one per separator of a level, 64 ints
  int step_gen = 0;           // ... and the number of the last such launch (a flag equal to it: the launch's span is solved)
  double *w256 = nullptr;     // explicit inverses of the 256-column diagonal spans of the wide top separators (k_solve_inv256, recomputed with the 16x16 inverses)
  double *step_xt = nullptr;  // 8 x 256 doubles: a step launch's x_k before it replaces the right-hand side
  bool keep_inverses = false; // the correction solves of a refinement: same factor as the solve before them, its diagonal inverses (16x16, spans) are kept
  int *info = nullptr;      // [0] first failing column, [1] separator; two slots of two ints: the program launch alternates between them (each launch clears the other
                            // one for the next: no memset node per factorisation), every other path uses slot 0
  int *info_last = nullptr; // the slot of the most recent factorisation (cholamd_factor_info)
  int info_parity = 0; bool info_foreign = false; // program path: slot of the next launch; slot 0 was used by another path since
  int *progress = nullptr;  // fused launches: columns published per pivot block (epoch * 64 + columns); [nsep + 1] = TRSM workgroups finished
  int epoch = 0, done_total = 0;
  int64_t *a_dst = nullptr; double *a_val = nullptr; int *perm = nullptr; double *ytmp = nullptr;
  // distributed top levels: the entries of A in the column blocks of the shared top THIS rank owns ([0]: by the fp64 schedule's blocks, [1]: the fp32 one's)
  int64_t *top_dst[2] = { nullptr, nullptr }; double *top_val[2] = { nullptr, nullptr }; int64_t top_n[2] = { 0, 0 }; int top_gen[2] = { -1, -1 };
  bool timing = false;
  std::vector<timed_launch> tl;
  std::vector<hipEvent_t> pool;
  // the one-launch program of the whole factorisation (single GPU, small problems: chol_build_program / k_program)
  level_dev prog;
  chol_job *jobs = nullptr; chol_wait *pwaits = nullptr; chol_ext *exts = nullptr;
  int *pctr = nullptr, *pctr_total = nullptr; // counters (last one: the queue head) and what one factorisation adds to each
  int n_job = 0, n_pctr = 0, prog_grid = 0, prog_epoch = 0, prog_epoch_limit = 0;
  bool prog_ready = false;
  unsigned long long *trace = nullptr; // diagnostic: per-job clock stamps of the next program launches (cholamd_device_program_trace)
  std::vector<chol_job> jobs_host;
  // mixed precision (fp32 factor + fp64 refinement): work lists of the fp32 kernels, their workspace, A as a device CSR
  std::vector<level_dev> lv32;
  float *ws32 = nullptr;
  int64_t *csr_ptr = nullptr; int *csr_col = nullptr; double *csr_val = nullptr;
  double *rvec = nullptr, *dxvec = nullptr, *partial = nullptr;
  // extend-add exchange under the distributed top levels: staging of the copies received for the owned column blocks, descriptors of the sum
  void *xstage = nullptr; size_t xstage_bytes = 0; void *xdesc = nullptr; int xdesc_gen = -1, xdesc_elem = 0, sched_gen = 0;
  // switches, read from the environment once at cholamd_device_create (cholamd_device_set_option changes them later)
  chol_sched_opts opt;
  bool solve_reference_shape = false; // cholamd_solve with the per-call (deterministic) kernels of the BLAS-level entry points
};

static int no_device_error()
{
  chol_set_error("no usable HIP device (libcholamd has no CPU fallback)");
  return CHOLAMD_ERR_NO_DEVICE;
}

extern "C" int cholamd_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

template <class T> static int upload_vec(T **dptr, const T *h, size_t n)
{
  *dptr = nullptr;
  if (n == 0) return 0;
  HIPCHK(hipMalloc((void **)dptr, n * sizeof(T)));
  HIPCHK(hipMemcpy(*dptr, h, n * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

static void free_level(level_dev &l)
{
  (void)hipFree(l.potrf); (void)hipFree(l.trsm); (void)hipFree(l.task); (void)hipFree(l.task_mt); (void)hipFree(l.src);
  l = level_dev();
}
static void free_levels(cholamd_device *d)
{
  for (auto &l : d->lv) free_level(l);
  d->lv.clear();
  for (auto &l : d->lv32) free_level(l);
  d->lv32.clear();
  free_level(d->prog);
  (void)hipFree(d->jobs); (void)hipFree(d->pwaits); (void)hipFree(d->exts); (void)hipFree(d->pctr); (void)hipFree(d->pctr_total);
  d->jobs = nullptr; d->pwaits = nullptr; d->exts = nullptr; d->pctr = nullptr; d->pctr_total = nullptr;
  d->prog_ready = false;
}
static void free_solve_lists(cholamd_device *d)
{
  for (auto &s : d->sv) { (void)hipFree(s.trsv); (void)hipFree(s.fw); (void)hipFree(s.bw); (void)hipFree(s.grp_start); (void)hipFree(s.grp_rows); (void)hipFree(s.bw_start); (void)hipFree(s.ifw); (void)hipFree(s.ibw); }
  d->sv.clear();
  (void)hipFree(d->w256); d->w256 = nullptr;
  (void)hipFree(d->zr_sub); d->zr_sub = nullptr; d->n_zr_sub = 0;
  d->solve_ready = false;
}
static int upload_level(level_dev &l, const chol_level_work &w, bool with_tables = true)
{
  l.n_potrf = w.n_potrf; l.n_trsm = w.n_trsm; l.n_task = w.n_task; l.n_src = w.n_src;
  l.phase.assign(w.phase, w.phase + w.n_phase);
  l.bcast.assign(w.bcast, w.bcast + w.n_bcast);
  // the POTRF descriptors travel with their role tables (chol_potrf_table) behind them in one buffer: desc.tab = byte offset from the descriptor itself
  int rc = 0;
  if (w.n_potrf > 0 && with_tables && !std::getenv("CHOLAMD_NO_ROLE_TABLES")) { // the switch: A/B and the parity test against the in-kernel construction
    const size_t dbytes = ((size_t)w.n_potrf * sizeof(chol_potrf_desc) + 15) / 16 * 16;
    std::vector<unsigned char> buf(dbytes + (size_t)w.n_potrf * CHOL_RR_TAB_BYTES);
    for (int i = 0; i < w.n_potrf; i++) {
      chol_potrf_desc pd = w.potrf[i];
      if (pd.n <= CHOL_RR_MAXN) {
        const size_t at = dbytes + (size_t)i * CHOL_RR_TAB_BYTES;
        pd.tab = (int)(at - (size_t)i * sizeof(chol_potrf_desc)); // relative to the descriptor itself: launches take sub-ranges of the array
        chol_potrf_table(pd.n, pd.sky, buf.data() + at);
      }
      std::memcpy(buf.data() + (size_t)i * sizeof(chol_potrf_desc), &pd, sizeof pd);
    }
    HIPCHK(hipMalloc((void **)&l.potrf, buf.size()));
    HIPCHK(hipMemcpy(l.potrf, buf.data(), buf.size(), hipMemcpyHostToDevice));
  } else rc = upload_vec(&l.potrf, w.potrf, (size_t)w.n_potrf);
  if (!rc) rc = upload_vec(&l.trsm, w.trsm, (size_t)w.n_trsm);
  if (!rc) rc = upload_vec(&l.task, w.task, (size_t)w.n_task);
  if (!rc) rc = upload_vec(&l.task_mt, w.task_mt, (size_t)w.n_task_mt);
  if (!rc) rc = upload_vec(&l.src, w.src, (size_t)w.n_src);
  return rc;
}

static int build_levels(cholamd_device *d)
{
  free_levels(d);
  d->sched_gen++;
  const int L = d->plan->levels;
  d->lv.resize(L);
  for (int lvl = 0; lvl < L; lvl++) {
    chol_level_work w;
    int rc = chol_build_level_work(d->plan, &d->opt, lvl, d->rank, d->world, &w);
    if (rc) return rc;
    rc = upload_level(d->lv[lvl], w);
    chol_level_work_free(&w);
    if (rc) return rc;
  }
  // single GPU, small problem: the same factorisation as one program launch (the per-level lists above stay for level
  // ranges, the partitioned run and the per-launch timing)
  if (d->world == 1 && d->opt.program) {
    chol_level_work w;
    chol_program g;
    // Liveness is a property of the job list: a follower is queued ahead of some strips it follows, so progress needs a minimum of
    // co-resident workgroups.  The list is simulated here, for every plan and option set, with FOUR resident workgroups and counters
    // raised only on job completion (stricter than the device): a program that cannot make progress that way is not used -- the
    // level-by-level lists below are.  (A GPU shared with other kernels or streams only delays jobs: every wait is bounded by ~2 s.)
    const bool built = chol_build_program(d->plan, &d->opt, &w, &g) == 0;
    const bool live = built && chol_program_check_built(d->plan, &d->opt, 4, &w, &g) == 0;
    if (built && !live) { chol_level_work_free(&w); chol_program_free(&g); }
    if (live) {
      int rc = upload_level(d->prog, w);
      std::vector<int> tot(g.ctr_total, g.ctr_total + g.n_ctr);
      int ncu = 256;
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, d->dev) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
      d->prog_grid = g.n_job < ncu ? g.n_job : ncu;
      tot.push_back(g.n_job + d->prog_grid); // the queue head: every workgroup draws one index past the end
      int mx = 1;
      for (int t : tot) mx = t > mx ? t : mx;
      d->prog_epoch_limit = (1 << 30) / mx;
      d->n_job = g.n_job; d->n_pctr = (int)tot.size(); d->prog_epoch = 0;
      d->jobs_host.assign(g.job, g.job + g.n_job);
      if (!rc) rc = upload_vec(&d->jobs, g.job, (size_t)g.n_job);
      const chol_wait no_wait = { 0, 0 };
      chol_ext no_ext;
      std::memset(&no_ext, 0, sizeof no_ext);
      if (!rc) rc = g.n_wait > 0 ? upload_vec(&d->pwaits, g.wait, (size_t)g.n_wait) : upload_vec(&d->pwaits, &no_wait, (size_t)1);
      if (!rc) rc = g.n_ext > 0 ? upload_vec(&d->exts, g.ext, (size_t)g.n_ext) : upload_vec(&d->exts, &no_ext, (size_t)1);
      if (!rc) rc = upload_vec(&d->pctr_total, tot.data(), tot.size());
      if (!rc) {
        HIPCHK(hipMalloc((void **)&d->pctr, tot.size() * sizeof(int)));
        HIPCHK(hipMemset(d->pctr, 0, tot.size() * sizeof(int)));
      }
      chol_level_work_free(&w);
      chol_program_free(&g);
      if (rc) return rc;
      d->prog_ready = d->n_job > 0;
    }
  }
  return 0;
}

extern "C" int cholamd_device_create(const cholamd_plan *plan, int device_id, cholamd_device **out)
{
  *out = nullptr;
  if (!plan) { chol_set_error("null plan"); return CHOLAMD_ERR_ARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return no_device_error();
  if (device_id < 0 || device_id >= ndev) { chol_set_error("device %d out of range (%d devices)", device_id, ndev); return CHOLAMD_ERR_NO_DEVICE; }
  HIPCHK(hipSetDevice(device_id));
  cholamd_device *d = new cholamd_device();
  d->plan = plan; d->dev = device_id;
  chol_sched_opts_from_env(&d->opt);
  { const char *e = getenv("CHOLAMD_SOLVE_REFERENCE_SHAPE"); d->solve_reference_shape = e && *e && atoi(e) != 0; }
  int rc = build_levels(d);
  if (!rc) {
    hipError_t e = hipMalloc((void **)&d->ws, (size_t)(plan->ws_doubles > 0 ? plan->ws_doubles : 1) * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&d->info, 4 * sizeof(int));
    if (e == hipSuccess) e = hipMemset(d->info, 0, 4 * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&d->progress, (size_t)(plan->nsep + 2) * sizeof(int));
    if (e == hipSuccess) e = hipMemset(d->progress, 0, (size_t)(plan->nsep + 2) * sizeof(int));
    if (e != hipSuccess) { chol_set_error("hipMalloc: %s", hipGetErrorString(e)); rc = CHOLAMD_ERR_HIP; }
  }
  if (!rc) rc = upload_vec(&d->a_dst, plan->a_dst, (size_t)plan->nnz_a);
  if (!rc) rc = upload_vec(&d->a_val, plan->a_val, (size_t)plan->nnz_a);
  if (!rc) rc = upload_vec(&d->perm, plan->perm, (size_t)plan->n);
  if (rc) { cholamd_device_destroy(d); return rc; }
  *out = d;
  return 0;
}

extern "C" void cholamd_device_destroy(cholamd_device *d)
{
  if (!d) return;
  (void)hipSetDevice(d->dev);
  free_levels(d);
  for (auto &a : d->vmm) { // sharded arenas the caller did not free
    (void)hipDeviceSynchronize();
    (void)hipMemUnmap(a.va, a.total);
    for (auto h : a.handles) (void)hipMemRelease(h);
    (void)hipMemAddressFree(a.va, a.total);
  }
  free_solve_lists(d);
  (void)hipFree(d->ws32); (void)hipFree(d->csr_ptr); (void)hipFree(d->csr_col); (void)hipFree(d->csr_val); (void)hipFree(d->rvec); (void)hipFree(d->dxvec); (void)hipFree(d->partial);
  (void)hipFree(d->xstage); (void)hipFree(d->xdesc);
  (void)hipFree(d->ws); (void)hipFree(d->ws_solve); (void)hipFree(d->step_flags); (void)hipFree(d->w256); (void)hipFree(d->step_xt); (void)hipFree(d->info); (void)hipFree(d->progress); (void)hipFree(d->a_dst); (void)hipFree(d->a_val); (void)hipFree(d->perm); (void)hipFree(d->ytmp);
  for (int q = 0; q < 2; q++) { (void)hipFree(d->top_dst[q]); (void)hipFree(d->top_val[q]); }
  for (auto &t : d->tl) { d->pool.push_back(t.a); d->pool.push_back(t.b); }
  for (auto e : d->pool) (void)hipEventDestroy(e);
  delete d;
}

extern "C" int cholamd_device_set_partition(cholamd_device *d, int rank, int world)
{
  HIPCHK(hipSetDevice(d->dev));
  if (world < 1 || (world & (world - 1)) || rank < 0 || rank >= world || chol_split_level(world) > d->plan->levels - 1) {
    chol_set_error("bad partition: rank %d of %d for a %d-level tree", rank, world, d->plan->levels);
    return CHOLAMD_ERR_ARG;
  }
  d->rank = rank; d->world = world;
  free_solve_lists(d); // rebuilt for the new partition at the next solve
  return build_levels(d);
}

extern "C" int cholamd_device_bcast_phases(const cholamd_device *d)
{ // broadcast phases (kind 6: top levels distributed by column blocks, option dist_top) in the device's per-level lists; > 0: the
  // factorisation needs a communicator (cholamd_factor_sharded / cholamd_factor_multi)
  int n = 0;
  for (const level_dev &l : d->lv) for (const chol_phase &ph : l.phase) n += ph.kind == 6;
  return n;
}
extern "C" int cholamd_device_set_option(cholamd_device *d, const char *name, int value)
{
  HIPCHK(hipSetDevice(d->dev));
  const std::string n(name ? name : "");
  bool rebuild = true;
  if (n == "split_min") d->opt.split_min = value;
  else if (n == "split_nb") d->opt.split_nb = value;
  else if (n == "fuse") d->opt.fuse = value != 0;
  else if (n == "fuse_update_max") d->opt.fuse_update_max = value;
  else if (n == "mt_min_tiles") d->opt.mt_min_tiles = value;
  else if (n == "cells") d->opt.cells = value != 0;
  else if (n == "program") d->opt.program = value != 0;
  else if (n == "follow") d->opt.follow = value != 0;
  else if (n == "super_blocks") d->opt.super_blocks = value;
  else if (n == "follow_tail") d->opt.follow_tail = value;
  else if (n == "follow_tail_split") d->opt.follow_tail_split = value;
  else if (n == "staged") d->opt.staged = value != 0;
  else if (n == "fine_upd") d->opt.fine_upd = value != 0;
  else if (n == "skyline") d->opt.skyline = value != 0;
  else if (n == "merge_targets") d->opt.merge_targets = value != 0;
  else if (n == "leaf_envelope") d->opt.leaf_envelope = value != 0;
  else if (n == "trsm_wt_min") d->opt.trsm_wt_min = value < 0 ? 0 : value;
  else if (n == "stage_chunk") d->opt.stage_chunk = value < 0 ? 0 : value;
  else if (n == "dist_top") d->opt.dist_top = value;
  else if (n == "solve_reference_shape") { d->solve_reference_shape = value != 0; rebuild = false; }
  else { chol_set_error("unknown option '%s'", n.c_str()); return CHOLAMD_ERR_ARG; }
  return rebuild ? build_levels(d) : 0;
}

extern "C" int cholamd_device_alloc(cholamd_device *d, int64_t doubles, double **dptr)
{
  HIPCHK(hipSetDevice(d->dev));
  HIPCHK(hipMalloc((void **)dptr, (size_t)doubles * sizeof(double)));
  return 0;
}
extern "C" int cholamd_device_free(cholamd_device *d, double *dptr)
{
  HIPCHK(hipSetDevice(d->dev));
  HIPCHK(hipFree(dptr));
  return 0;
}
extern "C" int cholamd_device_upload(cholamd_device *d, double *d_dst, const double *h_src, int64_t doubles, void *stream)
{
  HIPCHK(hipSetDevice(d->dev));
  HIPCHK(hipMemcpyAsync(d_dst, h_src, (size_t)doubles * sizeof(double), hipMemcpyHostToDevice, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}
extern "C" int cholamd_device_download(cholamd_device *d, double *h_dst, const double *d_src, int64_t doubles, void *stream)
{
  HIPCHK(hipSetDevice(d->dev));
  HIPCHK(hipMemcpyAsync(h_dst, d_src, (size_t)doubles * sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}
extern "C" int cholamd_device_sync(cholamd_device *d, void *stream)
{
  HIPCHK(hipSetDevice(d->dev));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

// ---- per-rank arenas (multi-GPU) -----------------------------------------------------------------
// The ranges of the arena a rank works on: the panels of its subtrees (one contiguous label range per tree level below the cut) and the shared
// top (the tail).  Rank 0 -- which gathers the factor and solves -- and a single-GPU device own everything.
static void owned_ranges(const cholamd_device *d, std::vector<std::pair<int64_t, int64_t>> &out)
{
  const cholamd_plan *p = d->plan;
  out.clear();
  if (d->world <= 1 || d->rank == 0) { out.push_back({ 0, p->arena }); return; }
  for (int s = 1; s <= p->nsep; s++) {
    const int o = chol_owner_of(p, s, d->world);
    if (o >= 0 && o != d->rank) continue;
    const int64_t lo = p->panel_off[s], hi = s < p->nsep ? p->panel_off[s + 1] : p->arena;
    if (!out.empty() && out.back().second == lo) out.back().second = hi; else out.push_back({ lo, hi });
  }
}
// zero what the rank owns (the other ranks' panels are never read on this rank; in a sharded arena they alias one scratch chunk).  A plain
// allocation is cleared whole: whole-arena consumers on a rank other than 0 (writers, arena_to_dense, the debug replay) then read zeros
// where the rank stores nothing
static int clear_owned(cholamd_device *d, void *arena, size_t elem, hipStream_t st)
{
  bool sharded = false;
  for (const auto &v : d->vmm) if (v.va == arena) sharded = true;
  if (!sharded) { HIPCHK(hipMemsetAsync(arena, 0, (size_t)d->plan->arena * elem, st)); return 0; }
  std::vector<std::pair<int64_t, int64_t>> r;
  owned_ranges(d, r);
  for (auto &x : r) HIPCHK(hipMemsetAsync((char *)arena + (size_t)x.first * elem, 0, (size_t)(x.second - x.first) * elem, st));
  return 0;
}
#define CHOL_VMM_CHUNK ((size_t)2 << 20)   /* owned ranges are backed in 2 MB steps */
#define CHOL_VMM_SCRATCH ((size_t)64 << 20) /* the chunk every range of another rank's panels aliases */
// An arena for this rank of cholamd_plan_arena_doubles() elements of elem_bytes (8: fp64, 4: the fp32 factor) whose address range is complete --
// every work descriptor keeps its offset -- but of which only the rank's own panels and the shared top are backed by memory of their own
// (hipMemAddressReserve / hipMemCreate / hipMemMap): the ranges of the other ranks' panels all alias ONE 64 MB scratch chunk (this rank never
// reads them; the fill's scatter of A and the macro-tile kernels' edge reads may touch them).  Rank 0 and single-GPU devices get a plain
// allocation.  *backed_bytes: the device memory the arena really takes.
extern "C" int cholamd_device_alloc_arena(cholamd_device *d, int elem_bytes, void **dptr, int64_t *backed_bytes)
{
  HIPCHK(hipSetDevice(d->dev));
  if (elem_bytes != 4 && elem_bytes != 8) { chol_set_error("alloc_arena: element size %d", elem_bytes); return CHOLAMD_ERR_ARG; }
  const size_t full = (size_t)d->plan->arena * (size_t)elem_bytes;
  if (d->world <= 1 || d->rank == 0) {
    HIPCHK(hipMalloc(dptr, full > 0 ? full : 1));
    if (backed_bytes) *backed_bytes = (int64_t)full;
    return 0;
  }
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = d->dev;
  size_t gran = 0;
  HIPCHK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  const size_t G = gran > CHOL_VMM_CHUNK ? gran : CHOL_VMM_CHUNK, total = (full + G - 1) / G * G, scratch = (CHOL_VMM_SCRATCH + G - 1) / G * G;
  std::vector<std::pair<int64_t, int64_t>> own;
  owned_ranges(d, own);
  std::vector<std::pair<size_t, size_t>> back; // owned byte ranges, widened to the chunk size, merged
  for (auto &x : own) {
    const size_t lo = (size_t)x.first * elem_bytes / G * G, hi = std::min(total, ((size_t)x.second * elem_bytes + G - 1) / G * G);
    if (!back.empty() && lo <= back.back().second) back.back().second = std::max(back.back().second, hi); else back.push_back({ lo, hi });
  }
  cholamd_device::vmm_arena A; A.va = nullptr; A.total = total;
  HIPCHK(hipMemAddressReserve(&A.va, total, G, nullptr, 0));
  auto fail = [&](hipError_t e, const char *what) {
    (void)hipMemUnmap(A.va, total);
    for (auto h : A.handles) (void)hipMemRelease(h);
    (void)hipMemAddressFree(A.va, total);
    chol_set_error("alloc_arena: %s: %s", what, hipGetErrorString(e));
    return CHOLAMD_ERR_HIP;
  };
  hipMemGenericAllocationHandle_t hs;
  hipError_t e = hipMemCreate(&hs, scratch, &prop, 0);
  if (e != hipSuccess) { (void)hipMemAddressFree(A.va, total); chol_set_error("alloc_arena: hipMemCreate: %s", hipGetErrorString(e)); return CHOLAMD_ERR_HIP; }
  A.handles.push_back(hs);
  size_t backed = scratch, pos = 0;
  auto alias = [&](size_t lo, size_t hi) { // [lo, hi) onto the scratch chunk, piece by piece
    for (size_t o = lo; o < hi;) {
      const size_t n = std::min(scratch, hi - o);
      hipError_t e2 = hipMemMap((char *)A.va + o, n, 0, hs, 0);
      if (e2 != hipSuccess) return e2;
      o += n;
    }
    return hipSuccess;
  };
  for (auto &b : back) {
    if ((e = alias(pos, b.first)) != hipSuccess) return fail(e, "hipMemMap (scratch)");
    hipMemGenericAllocationHandle_t h;
    if ((e = hipMemCreate(&h, b.second - b.first, &prop, 0)) != hipSuccess) return fail(e, "hipMemCreate");
    A.handles.push_back(h);
    if ((e = hipMemMap((char *)A.va + b.first, b.second - b.first, 0, h, 0)) != hipSuccess) return fail(e, "hipMemMap");
    backed += b.second - b.first;
    pos = b.second;
  }
  if ((e = alias(pos, total)) != hipSuccess) return fail(e, "hipMemMap (scratch)");
  // access for the rank's own device AND for every peer that can reach it: the local communicator's ordered sum, its peer copies and
  // the gather to rank 0 read other ranks' arenas from their owners' devices, and hipDeviceEnablePeerAccess does not cover memory
  // mapped with hipMemMap
  std::vector<hipMemAccessDesc> acc;
  int ndev = 0;
  (void)hipGetDeviceCount(&ndev);
  for (int q = 0; q < ndev; q++) {
    int can = q == d->dev;
    if (!can && hipDeviceCanAccessPeer(&can, q, d->dev) != hipSuccess) can = 0;
    if (!can) continue;
    hipMemAccessDesc a1 = {};
    a1.location.type = hipMemLocationTypeDevice; a1.location.id = q; a1.flags = hipMemAccessFlagsProtReadWrite;
    acc.push_back(a1);
  }
  if ((e = hipMemSetAccess(A.va, total, acc.data(), acc.size())) != hipSuccess) return fail(e, "hipMemSetAccess");
  d->vmm.push_back(A);
  *dptr = A.va;
  if (backed_bytes) *backed_bytes = (int64_t)backed;
  return 0;
}
extern "C" int cholamd_device_free_arena(cholamd_device *d, void *dptr)
{
  HIPCHK(hipSetDevice(d->dev));
  for (size_t i = 0; i < d->vmm.size(); i++)
    if (d->vmm[i].va == dptr) {
      HIPCHK(hipDeviceSynchronize());
      (void)hipMemUnmap(dptr, d->vmm[i].total);
      for (auto h : d->vmm[i].handles) (void)hipMemRelease(h);
      (void)hipMemAddressFree(dptr, d->vmm[i].total);
      d->vmm.erase(d->vmm.begin() + i);
      return 0;
    }
  HIPCHK(hipFree(dptr));
  return 0;
}

// Multi-GPU: which entries of A a rank's fill scatters.  Everything under the cut on every rank (the panels of the other ranks' subtrees are
// never read).  The shared top of the tree (the tail of the arena) must start from A on exactly ONE rank per element, so that the sum over
// the ranks after the local levels is A_top - all contributions: with replicated top levels (one all-reduce of the tail) that is rank 0; with
// the top levels distributed by column blocks it is the block's OWNER -- the extend-add exchange then carries a block only from the ranks whose
// subtrees reach it (chol_top_contributors), and rank 0 sends no more than any other rank.  `*below` = leading entries of (a_dst, a_val) to
// scatter; (*tdst, *tval, *ntop) = further entries (device arrays).  f32: the fp32 schedule's column blocks.
static void exchange_pieces(const cholamd_device *d, const std::vector<level_dev> &lv, std::vector<struct xpiece> &out);
static int top_entries(cholamd_device *d, int f32, int64_t *below, const int64_t **tdst, const double **tval, int64_t *ntop);
extern "C" int cholamd_device_fill(cholamd_device *d, double *d_arena, void *stream)
{
  HIPCHK(hipSetDevice(d->dev));
  hipStream_t st = (hipStream_t)stream;
  { int rc = clear_owned(d, d_arena, sizeof(double), st); if (rc) return rc; }
  int64_t below = 0, ntop = 0; const int64_t *tdst = nullptr; const double *tval = nullptr;
  { int rc = top_entries(d, 0, &below, &tdst, &tval, &ntop); if (rc) return rc; }
  HIPCHK((hipError_t)chol_launch_scatter(d_arena, d->a_dst, d->a_val, below, st));
  if (ntop > 0) HIPCHK((hipError_t)chol_launch_scatter(d_arena, tdst, tval, ntop, st));
  return 0;
}

// ---- timing helpers -------------------------------------------------------------------------
static hipEvent_t get_event(cholamd_device *d)
{
  if (!d->pool.empty()) { hipEvent_t e = d->pool.back(); d->pool.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
struct scoped_timer {
  cholamd_device *d; hipStream_t st; int kind; hipEvent_t a{}, b{}; bool on;
  scoped_timer(cholamd_device *d_, hipStream_t st_, int kind_, bool active) : d(d_), st(st_), kind(kind_), on(d_->timing && active)
  {
    if (on) { a = get_event(d); b = get_event(d); (void)hipEventRecord(a, st); }
  }
  ~scoped_timer()
  {
    if (on) { (void)hipEventRecord(b, st); d->tl.push_back({ a, b, kind }); }
  }
};

extern "C" int cholamd_device_set_timing(cholamd_device *d, int on)
{
  d->timing = on != 0;
  for (auto &t : d->tl) { d->pool.push_back(t.a); d->pool.push_back(t.b); }
  d->tl.clear();
  return 0;
}
// kinds: 0 potrf (+ fused trsm), 1 trsm, 2 update, 3 program launch, 4 extend-add exchange (RCCL), 5 broadcasts of the distributed top levels
#define CHOL_TIMING_KINDS 8
#define CHOL_TK_EXCHANGE 4
#define CHOL_TK_BCAST 5
extern "C" int cholamd_device_get_timing_ex(cholamd_device *d, float ms_by_kind[CHOL_TIMING_KINDS], int launches_by_kind[CHOL_TIMING_KINDS])
{
  HIPCHK(hipSetDevice(d->dev));
  for (int k = 0; k < CHOL_TIMING_KINDS; k++) { ms_by_kind[k] = 0.f; launches_by_kind[k] = 0; }
  for (auto &t : d->tl) {
    HIPCHK(hipEventSynchronize(t.b));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, t.a, t.b));
    ms_by_kind[t.kind] += ms; launches_by_kind[t.kind]++;
    d->pool.push_back(t.a); d->pool.push_back(t.b);
  }
  d->tl.clear();
  return 0;
}
extern "C" int cholamd_device_get_timing(cholamd_device *d, float ms_by_kind[4], int launches_by_kind[4])
{ // the four compute kinds (the exchange kinds of a sharded run: cholamd_device_get_timing_ex)
  float ms[CHOL_TIMING_KINDS]; int n[CHOL_TIMING_KINDS];
  const int rc = cholamd_device_get_timing_ex(d, ms, n);
  if (rc) return rc;
  for (int k = 0; k < 4; k++) { ms_by_kind[k] = ms[k]; launches_by_kind[k] = n[k]; }
  return 0;
}

extern "C" int cholamd_device_event_overhead(cholamd_device *d, void *stream, float *ms_out)
{ // what a (record, record) pair with NOTHING between reads on this stream: the share of every timed launch above that is
  // the event commands themselves, not the kernel (bench.py subtracts it)
  HIPCHK(hipSetDevice(d->dev));
  hipStream_t st = (hipStream_t)stream;
  const int reps = 64;
  std::vector<hipEvent_t> ev(2 * reps);
  for (auto &e : ev) e = get_event(d);
  for (int i = 0; i < reps; i++) { HIPCHK(hipEventRecord(ev[2 * i], st)); HIPCHK(hipEventRecord(ev[2 * i + 1], st)); }
  HIPCHK(hipStreamSynchronize(st));
  double acc = 0.0;
  for (int i = 0; i < reps; i++) { float ms = 0.f; HIPCHK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); acc += ms; }
  for (auto e : ev) d->pool.push_back(e);
  *ms_out = (float)(acc / reps);
  return 0;
}

// ---- the hot path ---------------------------------------------------------------------------
static int launch_phase(cholamd_device *d, const level_dev &l, const chol_phase &ph, double *d_arena, hipStream_t st)
{
  if (ph.kind == 5) {
    // the progress words (epoch * 64 + columns) and the count of finished TRSM workgroups are monotonic across launches:
    // both start over, in stream order, long before either can wrap
    if (d->epoch >= (1 << 24) || d->done_total >= (1 << 30)) { HIPCHK(hipMemsetAsync(d->progress, 0, (size_t)(d->plan->nsep + 2) * sizeof(int), st)); d->epoch = 0; d->done_total = 0; }
    d->epoch++;
    if (ph.n3 > 0) d->done_total += (ph.n2 + 2) / 3; // TRSM workgroups of this launch count themselves out only when update tasks ride along
    HIPCHK((hipError_t)chol_launch_potrf_trsm(d_arena, d->ws, l.potrf + ph.first, ph.n, l.trsm + ph.first2, ph.n2, l.task + ph.first3, l.src, ph.n3,
                                              d->info, d->progress, d->epoch * 64, d->progress + d->plan->nsep + 1, d->done_total, st));
  } else if (ph.kind == 0) HIPCHK((hipError_t)chol_launch_potrf(d_arena, d->ws, l.potrf + ph.first, ph.n, d->info, st));
  else if (ph.kind == 1) HIPCHK((hipError_t)chol_launch_trsm(d_arena, d->ws, l.trsm + ph.first, ph.n, st));
  else if (ph.kind == 4) HIPCHK((hipError_t)chol_launch_trsm_w(d_arena, d->ws, l.trsm + ph.first, ph.n, st));
  else if (ph.kind == 7) HIPCHK((hipError_t)chol_launch_trsm_wt(d_arena, d->ws, l.trsm + ph.first, ph.n, st));
  else if (ph.kind == 2) HIPCHK((hipError_t)chol_launch_update(d_arena, l.task + ph.first, l.src, ph.n, st));
  else if (ph.kind == 3) HIPCHK((hipError_t)chol_launch_update_mt(d_arena, l.task_mt + ph.first, l.src, ph.n, (int64_t)d->plan->arena, st));
  return 0;
}
struct cholamd_comm;
static int bcast_rank(cholamd_device *d, const level_dev &l, const chol_phase &ph, double *d_arena, cholamd_comm *c, hipStream_t st);
// levels [level_lo, level_hi] of one rank; a broadcast phase (kind 6: distributed top levels) goes through `c`
static int factor_levels_comm(cholamd_device *d, double *d_arena, int level_hi, int level_lo, cholamd_comm *c, hipStream_t st)
{
  HIPCHK(hipSetDevice(d->dev));
  const int L = d->plan->levels;
  if (level_hi >= L) level_hi = L - 1;
  if (level_lo < 0) level_lo = 0;
  if (level_hi == L - 1) HIPCHK(hipMemsetAsync(d->info, 0, 2 * sizeof(int), st));
  d->info_last = d->info; d->info_foreign = true; // slot 0: the level-by-level paths
  for (int lvl = level_hi; lvl >= level_lo; lvl--) { // mmat.rg:1227
    const level_dev &l = d->lv[lvl];
    for (const chol_phase &ph : l.phase) {
      if (ph.kind == 6) {
        scoped_timer t(d, st, CHOL_TK_BCAST, ph.n > 0);
        int rc = bcast_rank(d, l, ph, d_arena, c, st);
        if (rc) return rc;
        continue;
      }
      scoped_timer t(d, st, ph.kind == 3 ? 2 : (ph.kind == 4 || ph.kind == 7) ? 1 : ph.kind == 5 ? 0 : ph.kind, ph.n > 0);
      int rc = launch_phase(d, l, ph, d_arena, st);
      if (rc) return rc;
    }
  }
  return 0;
}
extern "C" int cholamd_factor_levels(cholamd_device *d, double *d_arena, int level_hi, int level_lo, void *stream)
{
  return factor_levels_comm(d, d_arena, level_hi, level_lo, nullptr, (hipStream_t)stream);
}
extern "C" int cholamd_factor(cholamd_device *d, double *d_arena, void *stream)
{
  if (!d->prog_ready) return cholamd_factor_levels(d, d_arena, d->plan->levels - 1, 0, stream);
  HIPCHK(hipSetDevice(d->dev));
  hipStream_t st = (hipStream_t)stream;
  if (d->info_foreign) { HIPCHK(hipMemsetAsync(d->info, 0, 4 * sizeof(int), st)); d->info_foreign = false; } // another path wrote slot 0 since
  int *const info_cur = d->info + 2 * d->info_parity, *const info_next = d->info + 2 * (d->info_parity ^ 1);
  d->info_parity ^= 1; d->info_last = info_cur;
  // counters are monotonic across factorisations (no reset between them); they start over, in stream order, long before
  // any of them can wrap
  if (d->prog_epoch >= d->prog_epoch_limit) { HIPCHK(hipMemsetAsync(d->pctr, 0, (size_t)d->n_pctr * sizeof(int), st)); d->prog_epoch = 0; }
  const level_dev &l = d->prog;
  {
    scoped_timer t(d, st, 3, true);
    HIPCHK((hipError_t)chol_launch_program(d_arena, d->ws, d->jobs, d->n_job, d->pwaits, l.potrf, l.trsm, l.task, l.src, d->exts, d->pctr, d->pctr_total, d->prog_epoch,
                                           d->pctr + d->n_pctr - 1, d->prog_epoch * (d->n_job + d->prog_grid), d->prog_grid, info_cur, info_next, d->trace, st));
  }
  d->prog_epoch++;
  return 0;
}
extern "C" int cholamd_device_program_trace(cholamd_device *d, double *d_arena, void *stream, int64_t cap, int64_t *out, int *njobs_out)
{ // diagnostic: one program launch on d_arena with per-job stamps; out[5 j ..] = kind, drawn, waits over, ended (10 ns ticks since
  // the first job was drawn), workgroup
  HIPCHK(hipSetDevice(d->dev));
  if (!d->prog_ready) { chol_set_error("no program launch for this problem / these options"); return CHOLAMD_ERR_ARG; }
  *njobs_out = d->n_job;
  if (cap < (int64_t)5 * d->n_job) return 0;
  // 4 stamps per job, then CHOL_TRACE_X per job (chol_kernels.h)
  const size_t nst = (size_t)(4 + CHOL_TRACE_X) * d->n_job;
  HIPCHK(hipMalloc((void **)&d->trace, nst * sizeof(unsigned long long)));
  HIPCHK(hipMemset(d->trace, 0, nst * sizeof(unsigned long long)));
  int rc = cholamd_factor(d, d_arena, stream);
  std::vector<unsigned long long> h(nst);
  if (!rc) {
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    HIPCHK(hipMemcpy(h.data(), d->trace, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  }
  (void)hipFree(d->trace);
  d->trace = nullptr;
  if (rc) return rc;
  unsigned long long t0 = ~0ull;
  for (int j = 0; j < d->n_job; j++) if (h[4 * j] && h[4 * j] < t0) t0 = h[4 * j];
  for (int j = 0; j < d->n_job; j++) {
    out[5 * j] = d->jobs_host[j].kind + (d->jobs_host[j].kind == 0 && d->jobs_host[j].n_ext > 0 ? 10 : 0);
    for (int q = 0; q < 3; q++) out[5 * j + 1 + q] = (int64_t)(h[4 * j + q] - t0);
    out[5 * j + 4] = (int64_t)h[4 * j + 3];
  }
  if (const char *path = getenv("CHOLAMD_TRACE_FOLLOW")) { // diagnostic: the followers' per-item stamps as text (us since the first job was drawn)
    if (FILE *fp = fopen(path, "w")) {
      for (int j = 0; j < d->n_job; j++) {
        const unsigned long long *x = &h[(size_t)4 * d->n_job + (size_t)CHOL_TRACE_X * j];
        const int kind = d->jobs_host[j].kind;
        bool any = false;
        for (int i = 0; i < CHOL_TRACE_X; i++) any = any || x[i] != 0;
        if (kind == 2 || !any) continue;
        fprintf(fp, "job %d kind %d", j, kind);
        if (kind == 0 && x[1]) {
          fprintf(fp, " items %llu own-tiles-wait-over %.1f rounds:", x[1], x[0] ? (double)(x[0] - t0) * 0.01 : -1.0);
          for (unsigned long long i = 0; i < x[1] && i < 44; i++) if (x[2 + i]) fprintf(fp, " %llu:%.1f", i, (double)(x[2 + i] - t0) * 0.01);
        }
        if (kind == 0 && x[24]) { fprintf(fp, " | prologue (entry, LDS init, tables, slot tables read, columns 0-1 requested, tiles requested, columns 0-1 parked):"); const int ord[7] = { 0, 1, 2, 5, 6, 3, 4 }; for (int k = 0; k < 7; k++) fprintf(fp, " %.1f", x[24 + ord[k]] ? (double)(x[24 + ord[k]] - t0) * 0.01 : -1.0); }
        fprintf(fp, kind == 0 ? " | column started:" : " | POTRF column seen:");
        for (int k = 0; k < 24; k++) if (x[72 + k]) fprintf(fp, " %d:%.1f", k, (double)(x[72 + k] - t0) * 0.01);
        fprintf(fp, kind == 0 ? " | column published:" : " | column tile on the channel:");
        for (int k = 0; k < 24; k++) if (x[48 + k]) fprintf(fp, " %d:%.1f", k, (double)(x[48 + k] - t0) * 0.01);
        fprintf(fp, "\n");
      }
      fclose(fp);
    }
  }
  return 0;
}
extern "C" int cholamd_factor_info(cholamd_device *d, int *sep_out)
{
  HIPCHK(hipSetDevice(d->dev));
  int h[2] = { 0, 0 };
  HIPCHK(hipMemcpy(h, d->info_last ? d->info_last : d->info, sizeof h, hipMemcpyDeviceToHost));
  if (sep_out) *sep_out = h[1];
  if (h[0] == CHOLAMD_ERR_STALL)
    chol_set_error("factorisation stalled: a workgroup of a fused launch waited ~50 ms for a progress word of the same launch and gave up; the factor is not valid");
  else if (h[0] < 0) chol_set_error("factorisation failed with internal code %d", h[0]);
  return h[0];
}

// ---- solve ----------------------------------------------------------------------------------
__global__ void k_zero_ranges(double *y, const int *ranges, int nr)
{ // ranges[2 i], ranges[2 i + 1] = offset, length; one block row per range
  const int off = ranges[2 * blockIdx.y], len = ranges[2 * blockIdx.y + 1];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += gridDim.x * blockDim.x) y[off + i] = 0.0;
}
// the solve lists of the whole tree (rank 0 of 1: cholamd_solve / cholamd_solve_refine, also on a partitioned device whose arena holds the gathered
// factor) or of one rank's share (the distributed solve); one set is kept, rebuilt when the other is asked for
static int build_solve(cholamd_device *d, int rank = 0, int world = 1)
{
  const int L = d->plan->levels;
  if (d->solve_ready && d->solve_rank == rank && d->solve_world == world) return 0;
  free_solve_lists(d);
  d->solve_rank = rank; d->solve_world = world;
  d->sv.resize(L);
  if (world > 1) { // what a rank of the distributed solve starts from zero: every position it does not own, the shared top unless it is rank 0
    const cholamd_plan *p = d->plan;
    std::vector<int> zr;
    for (int s = 1; s <= p->nsep; s++) {
      const int o = chol_owner_of(p, s, world);
      if ((o >= 0 && o != rank) || (o < 0 && rank != 0)) {
        if (!zr.empty() && zr[zr.size() - 2] + zr.back() == p->sep_off[s]) zr.back() += p->sep_size[s];
        else { zr.push_back(p->sep_off[s]); zr.push_back(p->sep_size[s]); }
      }
    }
    d->n_zr_sub = (int)zr.size() / 2;
    if (!zr.empty()) { int rc = upload_vec(&d->zr_sub, zr.data(), zr.size()); if (rc) return rc; }
  }
  for (int lvl = 0; lvl < L; lvl++) {
    chol_solve_level w;
    int rc = chol_build_solve_level_part(d->plan, lvl, rank, world, &w);
    if (rc) return rc;
    solve_dev &s = d->sv[lvl];
    s.n_trsv = w.n_trsv; s.n_grp = w.n_grp; s.n_fw = w.n_fw; s.n_bw = w.n_bw;
    rc = upload_vec(&s.trsv, w.trsv, (size_t)w.n_trsv);
    if (!rc) rc = upload_vec(&s.fw, w.fw, (size_t)w.n_fw);
    if (!rc) rc = upload_vec(&s.bw, w.bw, (size_t)w.n_bw);
    if (!rc) rc = upload_vec(&s.grp_start, w.grp_start, (size_t)w.n_grp + 1);
    if (!rc) rc = upload_vec(&s.grp_rows, w.grp_rows, (size_t)2 * w.n_grp);
    if (!rc) rc = upload_vec(&s.bw_start, w.bw_start, (size_t)w.n_trsv + 1);
    s.n_ifw = w.n_ifw; s.n_ibw = w.n_ibw; s.max_n = w.max_n; s.max_under = w.banded && w.max_rows_under_span > 0 ? -w.max_rows_under_span : w.max_rows_under_span;
    if (!rc) rc = upload_vec(&s.ifw, w.ifw, (size_t)4 * w.n_ifw);
    if (!rc) rc = upload_vec(&s.ibw, w.ibw, (size_t)4 * w.n_ibw);
    chol_solve_level_free(&w);
    if (rc) return rc;
  }
  if (!d->ytmp) HIPCHK(hipMalloc((void **)&d->ytmp, (size_t)d->plan->n * sizeof(double)));
  if (!d->ws_solve) HIPCHK(hipMalloc((void **)&d->ws_solve, (size_t)(d->plan->ws_doubles > 0 ? d->plan->ws_doubles : 1) * sizeof(double)));
  if (!d->step_flags) { HIPCHK(hipMalloc((void **)&d->step_flags, CHOL_STEPW_MAX_SEPS * 16 * sizeof(int))); HIPCHK(hipMemset(d->step_flags, 0, CHOL_STEPW_MAX_SEPS * 16 * sizeof(int))); }
  if (!d->step_xt) { HIPCHK(hipMalloc((void **)&d->step_xt, CHOL_STEPW_MAX_SEPS * 256 * sizeof(double))); HIPCHK(hipMemset(d->step_xt, 0, CHOL_STEPW_MAX_SEPS * 256 * sizeof(double))); }
  int64_t w256 = 0; // explicit inverses of the diagonal spans where the span chain is the solve's critical path: the levels of at most 8 separators
  if (!std::getenv("CHOLAMD_SOLVE_NO_INV256"))
    for (int lvl = 0; lvl < L; lvl++) {
      solve_dev &s = d->sv[lvl];
      if (s.n_trsv < 1 || s.n_trsv > CHOL_STEPW_MAX_SEPS || s.max_n <= 256 || s.max_under < 0) continue; // (a level of banded leaves: k_solve_leaf32 for an fp32 factor)
      s.w256_off = w256;
      w256 += (int64_t)s.n_trsv * ((s.max_n + 255) / 256) * 65536;
    }
  if (w256 > 0) HIPCHK(hipMalloc((void **)&d->w256, (size_t)w256 * sizeof(double)));
  d->solve_ready = true;
  return 0;
}

// first position of the shared top of the tree in the permuted vector (the separators above the cut are the last labels: a contiguous tail, as in the arena)
static int64_t top_vec_offset(const cholamd_device *d) { return d->solve_world > 1 ? d->plan->sep_off[d->plan->nsep - (d->solve_world - 1) + 1] : d->plan->n; }
// the streamed solve (every panel read once) with a factor of element type TL; vectors and arithmetic are fp64
struct keep_inverses_scope { // set on the devices of a refinement after its first solve, cleared on every way out
  cholamd_device *const *devs; int n;
  keep_inverses_scope(cholamd_device *const *devs_, int n_) : devs(devs_), n(n_) { for (int g = 0; g < n; g++) devs[g]->keep_inverses = true; }
  ~keep_inverses_scope() { for (int g = 0; g < n; g++) devs[g]->keep_inverses = false; }
};
static int lsolve_dinv(const double *a, const chol_trsv_desc *t, int n, int mx, double *W, hipStream_t st) { return chol_launch_solve_dinv(a, t, n, mx, W, st); }
static int lsolve_dinv(const float *a, const chol_trsv_desc *t, int n, int mx, double *W, hipStream_t st) { return chol32_launch_solve_dinv(a, t, n, mx, W, st); }
static int lsolve_trsv(cholamd_device *d, const double *a, const chol_trsv_desc *t, int n, int mx, int mu, const double *W, double *y, int bw, const double *W256, hipStream_t st) { return chol_launch_solve_trsv(a, t, n, mx, mu, W, y, bw, d->step_flags, &d->step_gen, W256, d->step_xt, st); }
static int lsolve_trsv(cholamd_device *d, const float *a, const chol_trsv_desc *t, int n, int mx, int mu, const double *W, double *y, int bw, const double *W256, hipStream_t st) { return chol32_launch_solve_trsv(a, t, n, mx, mu, W, y, bw, d->step_flags, &d->step_gen, W256, d->step_xt, st); }
static int lsolve_inv256(const double *a, const chol_trsv_desc *t, int n, int mx, const double *W16, double *W256, hipStream_t st) { return chol_launch_solve_inv256(a, t, n, mx, W16, W256, st); }
static int lsolve_inv256(const float *a, const chol_trsv_desc *t, int n, int mx, const double *W16, double *W256, hipStream_t st) { return chol32_launch_solve_inv256(a, t, n, mx, W16, W256, st); }
static int lsolve_off(const double *a, const chol_gemv_desc *g, const int *it, int n, double *y, int bw, hipStream_t st) { return chol_launch_solve_offdiag(a, g, it, n, y, bw, st); }
static int lsolve_off(const float *a, const chol_gemv_desc *g, const int *it, int n, double *y, int bw, hipStream_t st) { return chol32_launch_solve_offdiag(a, g, it, n, y, bw, st); }
// The streamed solve in three phases, so that a partitioned device can put the two vector reductions of the distributed solve between them:
//   phase 0: y = P b (a rank of a partition: zero outside what it owns), the 16x16 inverses, forward sweep of the levels under the cut
//            -> [sum of the shared top's part of y over the ranks]
//   phase 1: forward and backward sweep of the levels above the cut (every rank: it holds the whole factored top), backward sweep under the cut;
//            ranks other than 0 then clear the top's part again  -> [sum of y over the ranks: every rank has the whole permuted solution]
//   phase 2: x = P^T y
// A device that is not partitioned runs them back to back: the cut is at level 0.
template <class TL> static int solve_phase(cholamd_device *d, const TL *d_arena, const double *d_b, double *d_x, int phase, hipStream_t st)
{
  const int L = d->plan->levels, n = d->plan->n, cut = d->solve_world > 1 ? chol_split_level(d->solve_world) : 0;
  double *y = d->ytmp;
  if (phase == 0) {
    HIPCHK((hipError_t)chol_launch_permute(d_b, d->perm, y, n, 0, st));
    if (d->n_zr_sub > 0) { hipLaunchKernelGGL(k_zero_ranges, dim3(16, d->n_zr_sub), dim3(256), 0, st, y, d->zr_sub, d->n_zr_sub); HIPCHK(hipGetLastError()); }
    // the 16x16 inverses of this arena's diagonal blocks (the factorisation's workspace belongs to the last arena factored)
    for (int lvl = 0; lvl < L && !d->keep_inverses; lvl++) {
      const solve_dev &s = d->sv[lvl];
      HIPCHK((hipError_t)lsolve_dinv(d_arena, s.trsv, s.n_trsv, s.max_n, d->ws_solve, st));
      if (s.w256_off >= 0 && d->w256) HIPCHK((hipError_t)lsolve_inv256(d_arena, s.trsv, s.n_trsv, s.max_n, d->ws_solve, d->w256 + s.w256_off, st));
    }
  }
  if (phase <= 1) {
    const int hi = phase == 0 ? L - 1 : cut - 1, lo = phase == 0 ? cut : 0;
    for (int lvl = hi; lvl >= lo; lvl--) { // forward, mmat.rg:1395-1435: TRSV per separator, then its panel into the ancestors
      const solve_dev &s = d->sv[lvl];
      HIPCHK((hipError_t)lsolve_trsv(d, d_arena, s.trsv, s.n_trsv, s.max_n, s.max_under, d->ws_solve, y, 0, s.w256_off >= 0 && d->w256 ? d->w256 + s.w256_off : nullptr, st));
      HIPCHK((hipError_t)lsolve_off(d_arena, s.bw, s.ifw, s.n_ifw, y, 0, st));
    }
  }
  if (phase == 1) {
    for (int lvl = 0; lvl < L; lvl++) { // backward, mmat.rg:1438-1479: gather from the ancestors, then TRSV^T
      const solve_dev &s = d->sv[lvl];
      HIPCHK((hipError_t)lsolve_off(d_arena, s.bw, s.ibw, s.n_ibw, y, 1, st));
      HIPCHK((hipError_t)lsolve_trsv(d, d_arena, s.trsv, s.n_trsv, s.max_n, s.max_under, d->ws_solve, y, 1, s.w256_off >= 0 && d->w256 ? d->w256 + s.w256_off : nullptr, st));
    }
    if (d->solve_world > 1 && d->solve_rank != 0) { // the top's part of the solution is counted once in the sum that follows: rank 0's
      const int64_t t0 = top_vec_offset(d);
      HIPCHK(hipMemsetAsync(y + t0, 0, (size_t)(n - t0) * sizeof(double), st));
    }
  }
  if (phase == 2) HIPCHK((hipError_t)chol_launch_permute(y, d->perm, d_x, n, 1, st));
  return 0;
}
template <class TL> static int solve_streamed(cholamd_device *d, const TL *d_arena, const double *d_b, double *d_x, hipStream_t st)
{
  { int rc = build_solve(d); if (rc) return rc; } // the whole tree: the arena holds the complete factor (single GPU, or gathered on this rank)
  for (int ph = 0; ph < 3; ph++) { int rc = solve_phase(d, d_arena, d_b, d_x, ph, st); if (rc) return rc; }
  return 0;
}
extern "C" int cholamd_solve(cholamd_device *d, const double *d_arena, const double *d_b, double *d_x, void *stream)
{
  HIPCHK(hipSetDevice(d->dev));
  { int rc = build_solve(d); if (rc) return rc; }
  hipStream_t st = (hipStream_t)stream;
  if (!d->solve_reference_shape) return solve_streamed(d, d_arena, d_b, d_x, st);
  // the per-call kernels the BLAS-level entry points use (deterministic, slow at scale)
  const int L = d->plan->levels, n = d->plan->n;
  double *y = d->ytmp;
  HIPCHK((hipError_t)chol_launch_permute(d_b, d->perm, y, n, 0, st));
  for (int lvl = L - 1; lvl >= 0; lvl--) { // forward, mmat.rg:1395-1435
    const solve_dev &s = d->sv[lvl];
    HIPCHK((hipError_t)chol_launch_trsv_fwd(d_arena, s.trsv, s.n_trsv, y, st));
    HIPCHK((hipError_t)chol_launch_gemv_fwd(d_arena, s.fw, s.grp_start, s.grp_rows, s.n_grp, y, st));
  }
  for (int lvl = 0; lvl < L; lvl++) { // backward, mmat.rg:1438-1479
    const solve_dev &s = d->sv[lvl];
    HIPCHK((hipError_t)chol_launch_bwd(d_arena, s.trsv, s.bw, s.bw_start, s.n_trsv, y, st));
  }
  HIPCHK((hipError_t)chol_launch_permute(y, d->perm, d_x, n, 1, st));
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Mixed precision (BASELINE config 5): fp32 factor, fp64 iterative refinement of the solve.
// The fp32 arena has the element layout of the fp64 one (cholamd_plan_arena_doubles() floats).
// ---------------------------------------------------------------------------------------------
static int ensure_f32(cholamd_device *d)
{
  if (!d->lv32.empty()) return 0;
  const int L = d->plan->levels;
  chol_sched_opts o = d->opt;
  o.split_min = CHOL32_MAXN; o.split_nb = CHOL32_MAXN; o.fuse = 0; o.fuse_update_max = 0; o.trsm_group = CHOL32_TRSM_GROUP; // pivot blocks the LDS-resident fp32 POTRF takes, one launch per phase
  d->lv32.resize(L);
  for (int lvl = 0; lvl < L; lvl++) {
    chol_level_work w;
    int rc = chol_build_level_work(d->plan, &o, lvl, d->rank, d->world, &w);
    if (!rc) rc = upload_level(d->lv32[lvl], w, false); // the fp32 kernels have no role tables
    chol_level_work_free(&w);
    if (rc) { for (auto &l : d->lv32) free_level(l); d->lv32.clear(); return rc; }
  }
  if (!d->ws32) HIPCHK(hipMalloc((void **)&d->ws32, (size_t)(d->plan->ws_doubles > 0 ? d->plan->ws_doubles : 1) * sizeof(float)));
  return 0;
}
extern "C" int cholamd_device_fill_f32(cholamd_device *d, float *d_arena32, void *stream)
{
  HIPCHK(hipSetDevice(d->dev));
  hipStream_t st = (hipStream_t)stream;
  { int rc = clear_owned(d, d_arena32, sizeof(float), st); if (rc) return rc; }
  { int rc = ensure_f32(d); if (rc) return rc; } // the fp32 schedule's column blocks decide which entries of the shared top are this rank's
  int64_t below = 0, ntop = 0; const int64_t *tdst = nullptr; const double *tval = nullptr;
  { int rc = top_entries(d, 1, &below, &tdst, &tval, &ntop); if (rc) return rc; }
  HIPCHK((hipError_t)chol32_launch_scatter(d_arena32, d->a_dst, d->a_val, below, st));
  if (ntop > 0) HIPCHK((hipError_t)chol32_launch_scatter(d_arena32, tdst, tval, ntop, st));
  return 0;
}
static int launch_phase_f32(cholamd_device *d, const level_dev &l, const chol_phase &ph, float *d_arena32, hipStream_t st)
{
  if (ph.kind == 0) HIPCHK((hipError_t)chol32_launch_potrf(d_arena32, d->ws32, l.potrf + ph.first, ph.n, d->info, st));
  else if (ph.kind == 7) HIPCHK((hipError_t)chol32_launch_trsm_wt(d_arena32, d->ws32, l.trsm + ph.first, ph.n, st));
  else if (ph.kind == 1 || ph.kind == 4) HIPCHK((hipError_t)chol32_launch_trsm(d_arena32, d->ws32, l.trsm + ph.first, ph.n, st));
  else if (ph.kind == 2) HIPCHK((hipError_t)chol32_launch_update(d_arena32, l.task + ph.first, l.src, ph.n, st));
  else if (ph.kind == 3) HIPCHK((hipError_t)chol32_launch_update_mt(d_arena32, l.task_mt + ph.first, l.src, ph.n, (int64_t)d->plan->arena, st));
  else { chol_set_error("internal: phase kind %d in the fp32 schedule", ph.kind); return CHOLAMD_ERR_ARG; }
  return 0;
}
template <class T> static int bcast_rank_t(cholamd_device *d, const level_dev &l, const chol_phase &ph, T *d_arena, cholamd_comm *c, hipStream_t st);
static int factor_levels_f32_comm(cholamd_device *d, float *d_arena32, int level_hi, int level_lo, cholamd_comm *c, hipStream_t st);
extern "C" int cholamd_factor_levels_f32(cholamd_device *d, float *d_arena32, int level_hi, int level_lo, void *stream)
{
  return factor_levels_f32_comm(d, d_arena32, level_hi, level_lo, nullptr, (hipStream_t)stream);
}
static int factor_levels_f32_comm(cholamd_device *d, float *d_arena32, int level_hi, int level_lo, cholamd_comm *c, hipStream_t st)
{
  HIPCHK(hipSetDevice(d->dev));
  int rc = ensure_f32(d);
  if (rc) return rc;
  const int L = d->plan->levels;
  if (level_hi >= L) level_hi = L - 1;
  if (level_lo < 0) level_lo = 0;
  if (level_hi == L - 1) HIPCHK(hipMemsetAsync(d->info, 0, 2 * sizeof(int), st));
  d->info_last = d->info; d->info_foreign = true; // slot 0: the level-by-level paths
  for (int lvl = level_hi; lvl >= level_lo; lvl--) {
    const level_dev &l = d->lv32[lvl];
    for (const chol_phase &ph : l.phase) {
      if (ph.kind == 6) { // distributed top levels: the step's column blocks travel from their owners to every rank
        scoped_timer t(d, st, CHOL_TK_BCAST, ph.n > 0);
        if ((rc = bcast_rank_t<float>(d, l, ph, d_arena32, c, st))) return rc;
        continue;
      }
      scoped_timer t(d, st, ph.kind == 3 ? 2 : (ph.kind == 4 || ph.kind == 7) ? 1 : ph.kind, ph.n > 0);
      if ((rc = launch_phase_f32(d, l, ph, d_arena32, st))) return rc;
    }
  }
  return 0;
}
extern "C" int cholamd_factor_f32(cholamd_device *d, float *d_arena32, void *stream)
{
  return cholamd_factor_levels_f32(d, d_arena32, d->plan->levels - 1, 0, stream);
}
extern "C" int cholamd_solve_f32(cholamd_device *d, const float *d_arena32, const double *d_b, double *d_x, void *stream)
{
  HIPCHK(hipSetDevice(d->dev));
  { int rc = build_solve(d); if (rc) return rc; }
  return solve_streamed(d, d_arena32, d_b, d_x, (hipStream_t)stream);
}
static int ensure_refine(cholamd_device *d)
{
  if (d->csr_ptr) return 0;
  const cholamd_plan *p = d->plan;
  const int n = p->n;
  int rc = upload_vec(&d->csr_ptr, p->csr_ptr, (size_t)n + 1);
  if (!rc) rc = upload_vec(&d->csr_col, p->csr_col, (size_t)(p->csr_ptr[n] > 0 ? p->csr_ptr[n] : 1));
  if (!rc) rc = upload_vec(&d->csr_val, p->csr_val, (size_t)(p->csr_ptr[n] > 0 ? p->csr_ptr[n] : 1));
  if (rc) return rc;
  HIPCHK(hipMalloc((void **)&d->rvec, (size_t)n * sizeof(double)));
  HIPCHK(hipMalloc((void **)&d->dxvec, (size_t)n * sizeof(double)));
  HIPCHK(hipMalloc((void **)&d->partial, (size_t)2 * ((n + 255) / 256) * sizeof(double)));
  return 0;
}
// r = b - A x on the device (fp64, A = the matrix file's entries, both triangles), ||r|| / ||b|| back on the host
static int residual_norm(cholamd_device *d, const double *d_b, const double *d_x, double *d_r, double *relres, hipStream_t st)
{
  const int n = d->plan->n, nb = (n + 255) / 256;
  HIPCHK((hipError_t)chol_launch_residual(d->csr_ptr, d->csr_col, d->csr_val, d_b, d_x, d_r, n, d->partial, st));
  std::vector<double> h((size_t)2 * nb);
  HIPCHK(hipMemcpyAsync(h.data(), d->partial, h.size() * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  double r2 = 0.0, b2 = 0.0;
  for (int i = 0; i < nb; i++) { r2 += h[2 * i]; b2 += h[2 * i + 1]; }
  *relres = b2 > 0.0 ? std::sqrt(r2 / b2) : std::sqrt(r2);
  return 0;
}
extern "C" int cholamd_residual(cholamd_device *d, const double *d_b, const double *d_x, double *d_r, double *relres_out, void *stream)
{
  HIPCHK(hipSetDevice(d->dev));
  int rc = ensure_refine(d);
  if (rc) return rc;
  return residual_norm(d, d_b, d_x, d_r, relres_out, (hipStream_t)stream);
}
extern "C" int cholamd_solve_refine(cholamd_device *d, const float *d_arena32, const double *d_b, double *d_x, int max_iter, double tol,
                                    int *iters_out, double *relres_out, void *stream)
{
  HIPCHK(hipSetDevice(d->dev));
  int rc = ensure_refine(d);
  if (!rc) rc = build_solve(d);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int n = d->plan->n;
  if (max_iter < 0) max_iter = 0;
  // x0 = M^-1 b with M = L32 L32^T; then x += M^-1 (b - A x) until ||b - A x|| <= tol ||b||
  rc = solve_streamed(d, d_arena32, d_b, d_x, st);
  if (rc) return rc;
  keep_inverses_scope keep(&d, 1);
  double rel = 0.0;
  int it = 0;
  for (;; ++it) {
    if ((rc = residual_norm(d, d_b, d_x, d->rvec, &rel, st))) return rc;
    if (!(rel > tol) || it >= max_iter) break; // also leaves on NaN
    if ((rc = solve_streamed(d, d_arena32, d->rvec, d->dxvec, st))) return rc;
    HIPCHK((hipError_t)chol_launch_axpy1(d_x, d->dxvec, n, st));
  }
  if (iters_out) *iters_out = it;
  if (relres_out) *relres_out = rel;
  if (rel != rel) { chol_set_error("iterative refinement produced NaN (fp32 factorisation broke down)"); return CHOLAMD_ERR_ARG; }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// L-A / L-B: stand-alone batched launches with pointer-valued descriptors (base = nullptr)
// ---------------------------------------------------------------------------------------------
static inline int64_t poff(const void *p) { return (int64_t)((uintptr_t)p / sizeof(double)); }

struct scratch { // per-call device scratch, freed at scope exit after the stream has been synchronised
  std::vector<void *> ptrs;
  ~scratch() { for (void *p : ptrs) (void)hipFree(p); }
  template <class T> int put(T **dptr, const T *h, size_t n, hipStream_t st)
  {
    *dptr = nullptr;
    if (n == 0) return 0;
    HIPCHK(hipMalloc((void **)dptr, n * sizeof(T)));
    ptrs.push_back(*dptr);
    HIPCHK(hipMemcpyAsync(*dptr, h, n * sizeof(T), hipMemcpyHostToDevice, st));
    return 0;
  }
  template <class T> int get(T **dptr, size_t n)
  {
    HIPCHK(hipMalloc((void **)dptr, (n ? n : 1) * sizeof(T)));
    ptrs.push_back(*dptr);
    return 0;
  }
};

static int check_ptr(const void *p, const char *what)
{
  if (!p || ((uintptr_t)p & 7)) { chol_set_error("%s must be a non-null, 8-byte aligned pointer", what); return CHOLAMD_ERR_ARG; }
  return 0;
}
static int have_device()
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return no_device_error();
  return 0;
}

static void add_update_tasks(std::vector<chol_upd_task> &tasks, int src_begin, int src_end, double *c, int ldc, int m, int n, bool syrk)
{
  const int tr = (m + 15) / 16, tc = (n + 15) / 16;
  for (int a = 0; a < tr; a++)
    for (int b = 0; b < tc; b++) {
      if (syrk && b > a) continue;
      chol_upd_task t;
      std::memset(&t, 0, sizeof t);
      t.c_off = poff(c) + a * 16 + (int64_t)b * 16 * ldc;
      t.ldc = ldc;
      t.mv = (short)(m - a * 16 < 16 ? m - a * 16 : 16);
      t.nv = (short)(n - b * 16 < 16 ? n - b * 16 : 16);
      t.lower = (syrk && a == b);
      t.src_begin = src_begin; t.src_end = src_end; t.ar = a * 16; t.br = b * 16;
      tasks.push_back(t);
    }
}

static int run_updates(std::vector<chol_upd_task> &tasks, std::vector<chol_upd_src> &srcs, hipStream_t st)
{
  if (tasks.empty()) return 0;
  scratch sc;
  chol_upd_task *dt; chol_upd_src *ds;
  int rc = sc.put(&dt, tasks.data(), tasks.size(), st);
  if (!rc) rc = sc.put(&ds, srcs.data(), srcs.size(), st);
  if (rc) return rc;
  HIPCHK((hipError_t)chol_launch_update(nullptr, dt, ds, (int)tasks.size(), st));
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

// B tiles (rows m_i of width n) <- B L^-T for one pivot L (n x n, ld ldl)
static int run_trsm(const double *Lp, int n, int ldl, const std::vector<chol_trsm_desc> &rows, hipStream_t st)
{
  if (rows.empty() || n == 0) return 0;
  scratch sc;
  double *W; chol_trsm_desc *dd;
  const size_t nb = (size_t)(n + CHOL_NB - 1) / CHOL_NB;
  int rc = sc.get(&W, nb * CHOL_NB * CHOL_NB);
  if (rc) return rc;
  HIPCHK((hipError_t)chol_launch_dinv(Lp, n, ldl, W, st));
  std::vector<chol_trsm_desc> v;
  for (const auto &r : rows)
    for (int r0 = 0; r0 < r.m; r0 += CHOL_TRSM_ROWS) {
      chol_trsm_desc t = r;
      t.l_off = poff(Lp); t.dinv_off = poff(W); t.b_off = r.b_off + r0;
      t.m = r.m - r0 < CHOL_TRSM_ROWS ? r.m - r0 : CHOL_TRSM_ROWS;
      v.push_back(t);
    }
  rc = sc.put(&dd, v.data(), v.size(), st);
  if (rc) return rc;
  if (n <= CHOL_TRSM_W_MAXN) HIPCHK((hipError_t)chol_launch_trsm_w(nullptr, nullptr, dd, (int)v.size(), st));
  else if (n <= CHOL_RR_MAXN) HIPCHK((hipError_t)chol_launch_trsm(nullptr, nullptr, dd, (int)v.size(), st));
  else HIPCHK((hipError_t)chol_launch_trsm_big(nullptr, nullptr, dd, (int)v.size(), st));
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

static int run_potrf(const std::vector<chol_potrf_desc> &v, hipStream_t st, int *info_out, int *sep_out)
{
  if (info_out) *info_out = 0;
  if (v.empty()) return 0;
  scratch sc;
  size_t wsz = 0;
  std::vector<chol_potrf_desc> dv;
  for (const auto &p : v) if (p.n <= CHOL_RR_MAXN) dv.push_back(p);
  const int n_small = (int)dv.size();
  for (const auto &p : v) if (p.n > CHOL_RR_MAXN) dv.push_back(p);
  for (auto &p : dv) { p.dinv_off = (int64_t)wsz; wsz += (size_t)((p.n + CHOL_NB - 1) / CHOL_NB) * CHOL_NB * CHOL_NB; }
  double *W; int *info; chol_potrf_desc *dd;
  int rc = sc.get(&W, wsz);
  if (!rc) rc = sc.get(&info, (size_t)2);
  if (rc) return rc;
  HIPCHK(hipMemsetAsync(info, 0, 2 * sizeof(int), st));
  rc = sc.put(&dd, dv.data(), dv.size(), st);
  if (rc) return rc;
  HIPCHK((hipError_t)chol_launch_potrf(nullptr, W, dd, n_small, info, st));
  HIPCHK((hipError_t)chol_launch_potrf_big(nullptr, W, dd + n_small, (int)dv.size() - n_small, info, st));
  int h[2] = { 0, 0 };
  HIPCHK(hipMemcpyAsync(h, info, sizeof h, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (info_out) *info_out = h[0];
  if (sep_out) *sep_out = h[1];
  return 0;
}

// ---- L-B: fused leaf tasks -------------------------------------------------------------------
static double *tile_ptr(const cholamd_region *r, const cholamd_filled *f)
{ // get_raw_ptr_2d, blas.rg:35-43; a row-compacted block instance stores only the 16-row tiles its filled tiles touch (tile_row)
  const int row = f->lo_x - r->lo_x;
  const int64_t srow = r->tile_row ? (int64_t)r->tile_row[row / CHOL_NB] * CHOL_NB + row % CHOL_NB : row;
  return r->ptr + srow + (int64_t)(f->lo_y - r->lo_y) * r->ld;
}
static int region_ok(const cholamd_region *r, const char *name)
{
  if (!r) { chol_set_error("region %s is null", name); return CHOLAMD_ERR_ARG; }
  return check_ptr(r->ptr, name);
}
static int tile_inside(const cholamd_region *r, const cholamd_filled *f)
{
  if (f->lo_x < r->lo_x || f->lo_y < r->lo_y || f->hi_x > r->hi_x || f->hi_y > r->hi_y) {
    chol_set_error("tile (%d,%d,%d) lies outside its region", f->sep_x, f->sep_y, f->cluster);
    return CHOLAMD_ERR_ARG;
  }
  if (r->tile_row) // every row of the tile must have storage, consecutively
    for (int t = (f->lo_x - r->lo_x) / CHOL_NB; t <= (f->hi_x - r->lo_x) / CHOL_NB; t++)
      if (r->tile_row[t] < 0 || (t > (f->lo_x - r->lo_x) / CHOL_NB && r->tile_row[t] != r->tile_row[t - 1] + 1)) {
        chol_set_error("tile (%d,%d,%d) touches rows its region does not store", f->sep_x, f->sep_y, f->cluster);
        return CHOLAMD_ERR_ARG;
      }
  return 0;
}

extern "C" int cholamd_plan_region(const cholamd_plan *p, double *d_arena, int r, int c, cholamd_region *out)
{
  const chol_block *B = chol_plan_block(p, r, c);
  if (!B || !out) { chol_set_error("no block (%d, %d)", r, c); return CHOLAMD_ERR_ARG; }
  cholamd_region rg = { d_arena + B->off, B->ld, B->lo_x, B->lo_y, B->hi_x, B->hi_y, B->tmap };
  *out = rg;
  return 0;
}

extern "C" int cholamd_fused_dpotrf(const cholamd_region *rA, const cholamd_filled *fa, int nA, int level, int interval, int debug, void *stream)
{
  int rc = have_device();
  if (!rc) rc = region_ok(rA, "rA");
  if (rc) return rc;
  std::vector<chol_potrf_desc> v;
  for (int i = 0; i < nA; i++) {
    if ((rc = tile_inside(rA, &fa[i]))) return rc;
    const int m = fa[i].hi_x - fa[i].lo_x + 1;
    if (debug)
      printf("POTRF: {'A': (%d, %d, %d), 'A_Lo': (%d, %d), 'A_Hi': (%d, %d), 'SizeA': (%d, %d), 'Block': (%d, %d), 'Level': %d, 'Interval': %d}\n",
             fa[i].sep_x, fa[i].sep_y, fa[i].cluster, fa[i].lo_x, fa[i].lo_y, fa[i].hi_x, fa[i].hi_y, m, fa[i].hi_y - fa[i].lo_y + 1,
             fa[i].sep_x, fa[i].sep_y, level, interval);
    if (m == 0) continue; // blas.rg:68
    chol_potrf_desc p = { poff(tile_ptr(rA, &fa[i])), 0, m, rA->ld, fa[i].sep_x, 0, 0, 0, { 0 } };
    v.push_back(p);
  }
  int info = 0;
  rc = run_potrf(v, (hipStream_t)stream, &info, nullptr);
  return rc ? rc : info;
}

extern "C" int cholamd_fused_dtrsm(const cholamd_region *rA, const cholamd_region *rB, const cholamd_filled *fa, int nA,
                                   const cholamd_filled *fb, int nB, int level, int interval, int debug, void *stream)
{
  int rc = have_device();
  if (!rc) rc = region_ok(rA, "rA");
  if (!rc) rc = region_ok(rB, "rB");
  if (rc) return rc;
  for (int i = 0; i < nA; i++) {
    if ((rc = tile_inside(rA, &fa[i]))) return rc;
    const double *Lp = tile_ptr(rA, &fa[i]);
    std::vector<chol_trsm_desc> rows;
    int n = 0;
    for (int j = 0; j < nB; j++) {
      if ((rc = tile_inside(rB, &fb[j]))) return rc;
      const int m = fb[j].hi_x - fb[j].lo_x + 1;
      n = fb[j].hi_y - fb[j].lo_y + 1;
      if (debug)
        printf("TRSM: {'A': (%d, %d, %d), 'A_Lo': (%d, %d), 'A_Hi': (%d, %d), 'SizeA': (%d, %d), 'B': (%d, %d, %d), 'B_Lo': (%d, %d), 'B_Hi': (%d, %d), 'SizeB': (%d, %d), 'Block': (%d, %d), 'Level': %d, 'Interval': %d}\n",
               fa[i].sep_x, fa[i].sep_y, fa[i].cluster, fa[i].lo_x, fa[i].lo_y, fa[i].hi_x, fa[i].hi_y, fa[i].hi_x - fa[i].lo_x + 1, fa[i].hi_y - fa[i].lo_y + 1,
               fb[j].sep_x, fb[j].sep_y, fb[j].cluster, fb[j].lo_x, fb[j].lo_y, fb[j].hi_x, fb[j].hi_y, m, n, fb[j].sep_x, fb[j].sep_y, level, interval);
      if (n != fa[i].hi_x - fa[i].lo_x + 1) { chol_set_error("TRSM: B tile width %d != pivot size %d", n, fa[i].hi_x - fa[i].lo_x + 1); return CHOLAMD_ERR_ARG; }
      chol_trsm_desc t = { 0, 0, poff(tile_ptr(rB, &fb[j])), n, rA->ld, m, rB->ld };
      rows.push_back(t);
    }
    if ((rc = run_trsm(Lp, n, rA->ld, rows, (hipStream_t)stream))) return rc;
  }
  return 0;
}

static int fused_update(const cholamd_region *rA, const cholamd_region *rB, const cholamd_region *rC,
                        const cholamd_filled *fa, int nA, const cholamd_filled *fb, int nB, const cholamd_filled *fc, int nC,
                        int ccs, int level, int interval, int debug, void *stream, bool is_syrk)
{
  int rc = have_device();
  if (!rc) rc = region_ok(rA, "rA");
  if (!rc) rc = region_ok(rB, "rB");
  if (!rc) rc = region_ok(rC, "rC");
  if (rc) return rc;
  std::vector<chol_upd_task> tasks;
  std::vector<chol_upd_src> srcs;
  for (int i = 0; i < nA; i++) {
    const cholamd_filled &a = fa[i];
    if ((rc = tile_inside(rA, &a))) return rc;
    const int row = a.cluster, sAx = a.hi_x - a.lo_x + 1, sAy = a.hi_y - a.lo_y + 1;
    for (int j = 0; j < nB; j++) {
      const cholamd_filled &b = fb[j];
      if ((rc = tile_inside(rB, &b))) return rc;
      const int col = b.cluster, sBx = b.hi_x - b.lo_x + 1;
      const int cz = row * ccs + col;
      const cholamd_filled *c = nullptr;
      for (int k = 0; k < nC; k++) // the reference's linear search, blas.rg:385-392 / 468-475
        if (fc[k].sep_x == a.sep_x && fc[k].sep_y == b.sep_x && fc[k].cluster == cz) { c = &fc[k]; break; }
      if (!c) continue;
      if ((rc = tile_inside(rC, c))) return rc;
      const int sCx = c->hi_x - c->lo_x + 1, sCy = c->hi_y - c->lo_y + 1;
      if (sCx <= 0 || sCy <= 0) continue;
      if (is_syrk && col > row) continue;
      if (debug)
        printf("GEMM: {'A': (%d, %d, %d), 'A_Lo': (%d, %d), 'A_Hi': (%d, %d), 'sizeA': (%d, %d), 'B': (%d, %d, %d), 'B_Lo': (%d, %d), 'B_Hi': (%d, %d), 'sizeB': (%d, %d), 'C': (%d, %d, %d), 'C_Lo': (%d, %d), 'C_Hi': (%d, %d), 'sizeC': (%d, %d), 'Block': (%d, %d), 'Level': %d, 'Interval': %d}\n",
               a.sep_x, a.sep_y, a.cluster, a.lo_x, a.lo_y, a.hi_x, a.hi_y, sAx, sAy, b.sep_x, b.sep_y, b.cluster, b.lo_x, b.lo_y, b.hi_x, b.hi_y, sBx, b.hi_y - b.lo_y + 1,
               c->sep_x, c->sep_y, c->cluster, c->lo_x, c->lo_y, c->hi_x, c->hi_y, sCx, sCy, c->sep_x, c->sep_y, level, interval);
      const bool syrk = is_syrk && col == row;
      chol_upd_src s = { poff(tile_ptr(rA, &a)), poff(tile_ptr(rB, &b)), rA->ld, rB->ld, sAy, 0, 0, 0 };
      srcs.push_back(s);
      // distinct (a, b) pairs address distinct C tiles inside one fused task, so one source per task group
      add_update_tasks(tasks, (int)srcs.size() - 1, (int)srcs.size(), tile_ptr(rC, c), rC->ld, syrk ? sCx : sAx, syrk ? sCx : sBx, syrk);
    }
  }
  return run_updates(tasks, srcs, (hipStream_t)stream);
}

extern "C" int cholamd_fused_dsyrk(const cholamd_region *rA, const cholamd_region *rB, const cholamd_region *rC,
                                   const cholamd_filled *fa, int nA, const cholamd_filled *fb, int nB, const cholamd_filled *fc, int nC,
                                   int ccs, int level, int interval, int debug, void *stream)
{
  return fused_update(rA, rB, rC, fa, nA, fb, nB, fc, nC, ccs, level, interval, debug, stream, true);
}
extern "C" int cholamd_fused_dgemm(const cholamd_region *rA, const cholamd_region *rB, const cholamd_region *rC,
                                   const cholamd_filled *fa, int nA, const cholamd_filled *fb, int nB, const cholamd_filled *fc, int nC,
                                   int ccs, int level, int interval, int debug, void *stream)
{
  return fused_update(rA, rB, rC, fa, nA, fb, nB, fc, nC, ccs, level, interval, debug, stream, false);
}

// ---- the reference's debug mode: the main loop of mmat.rg:1227-1355 literally, one fused task at a time, with write_blocks dumps
static int debug_dump(cholamd_device *d, const double *d_arena, std::vector<double> &host, const char *dir, int level, const char *op,
                      int ax, int ay, int bx, int by, int cx, int cy, int full)
{
  const cholamd_plan *p = d->plan;
  HIPCHK(hipMemcpy(host.data(), d_arena, (size_t)p->arena * sizeof(double), hipMemcpyDeviceToHost));
  char stem[1100], file[1200], header[200];
  if (!std::strcmp(op, "POTRF")) { snprintf(stem, sizeof stem, "%s/potrf_lvl%d_a%d%d", dir, level, ax, ay); snprintf(header, sizeof header, "Level: %d POTRF A=(%d, %d)", level, ax, ay); }
  else if (!std::strcmp(op, "TRSM")) { snprintf(stem, sizeof stem, "%s/trsm_lvl%d_a%d%d_b%d%d", dir, level, ax, ay, bx, by); snprintf(header, sizeof header, "Level: %d TRSM A=(%d, %d) B=(%d, %d)", level, ax, ay, bx, by); }
  else { snprintf(stem, sizeof stem, "%s/gemm_lvl%d_a%d%d_b%d%d_c%d%d", dir, level, ax, ay, bx, by, cx, cy); snprintf(header, sizeof header, "Level: %d GEMM A=(%d, %d) B=(%d, %d) C=(%d, %d)", level, ax, ay, bx, by, cx, cy); }
  snprintf(file, sizeof file, "%s.mtx", stem);
  printf("filename: %s\n", file);
  printf("saving matrix to: %s\n\n", file);
  int rc = cholamd_plan_write_matrix(p, host.data(), file, full);
  if (rc) return rc;
  snprintf(file, sizeof file, "%s.txt", stem);
  printf("filename: %s\n", file);
  return cholamd_plan_write_blocks_txt(p, host.data(), file, header);
}
extern "C" int cholamd_factor_debug(cholamd_device *d, double *d_arena, const char *dir, int full_precision, void *stream)
{
  HIPCHK(hipSetDevice(d->dev));
  const cholamd_plan *p = d->plan;
  const int L = p->levels, ns = p->nsep;
  std::vector<double> host((size_t)p->arena);
  auto region_of = [&](int r, int c) {
    const chol_block *B = chol_plan_block(p, r, c);
    cholamd_region rg = { d_arena + B->off, B->ld, B->lo_x, B->lo_y, B->hi_x, B->hi_y, B->tmap };
    return rg;
  };
  for (int lvl = L - 1; lvl >= 0; lvl--) {
    const int lbl = L - 1 - lvl;
    // filled tiles of the level's snapshot, grouped by block (the snapshot is ordered by (row label, col label, cluster))
    std::vector<std::vector<cholamd_filled>> fl((size_t)(ns + 1) * (ns + 1));
    for (int64_t i = 0; i < p->snap_n[lbl]; i++) fl[(size_t)p->snap[lbl][i].sep_x * (ns + 1) + p->snap[lbl][i].sep_y].push_back(p->snap[lbl][i]);
    auto F = [&](int r, int c) -> std::vector<cholamd_filled> & { return fl[(size_t)r * (ns + 1) + c]; };
    const int h0 = 1 << lvl, h1 = (1 << (lvl + 1)) - 1;
    for (int h = h0; h <= h1; h++) { // POTRF sweep, mmat.rg:1240-1257
      const int s = p->tree[h];
      cholamd_region rA = region_of(s, s);
      int rc = cholamd_fused_dpotrf(&rA, F(s, s).data(), (int)F(s, s).size(), lvl, lbl, 1, stream);
      if (rc < 0) return rc;
      if (rc > 0) { int h2[2] = { rc, s }; HIPCHK(hipMemcpy(d->info, h2, sizeof h2, hipMemcpyHostToDevice)); d->info_last = d->info; d->info_foreign = true; }
      if ((rc = debug_dump(d, d_arena, host, dir, lvl, "POTRF", s, s, 0, 0, 0, 0, full_precision))) return rc;
    }
    for (int h = h0; h <= h1; h++) { // TRSM sweep, mmat.rg:1259-1291
      const int s = p->tree[h];
      for (int hp = h / 2; hp >= 1; hp /= 2) {
        const int par = p->tree[hp];
        cholamd_region rA = region_of(s, s), rB = region_of(par, s);
        int rc = cholamd_fused_dtrsm(&rA, &rB, F(s, s).data(), (int)F(s, s).size(), F(par, s).data(), (int)F(par, s).size(), lvl, lbl, 1, stream);
        if (rc) return rc;
        if ((rc = debug_dump(d, d_arena, host, dir, lvl, "TRSM", s, s, par, s, 0, 0, full_precision))) return rc;
      }
    }
    for (int h = h0; h <= h1; h++) { // SYRK / GEMM sweep, mmat.rg:1293-1347
      const int s = p->tree[h];
      for (int hp = h / 2; hp >= 1; hp /= 2) {
        const int par = p->tree[hp];
        const int ccs = cholamd_plan_ntiles_at_level(p, par, lvl);
        for (int hg = hp; hg >= 1; hg /= 2) {
          const int gp = p->tree[hg];
          cholamd_region rA = region_of(gp, s), rB = region_of(par, s), rC = region_of(gp, par);
          int rc = (gp == par ? cholamd_fused_dsyrk : cholamd_fused_dgemm)(&rA, &rB, &rC, F(gp, s).data(), (int)F(gp, s).size(), F(par, s).data(), (int)F(par, s).size(),
                                                                             F(gp, par).data(), (int)F(gp, par).size(), ccs, lvl, lbl, 1, stream);
          if (rc) return rc;
          if ((rc = debug_dump(d, d_arena, host, dir, lvl, "GEMM", gp, s, par, s, gp, par, full_precision))) return rc;
        }
      }
    }
  }
  return 0;
}

// ---- L-A: device-pointer BLAS ----------------------------------------------------------------
extern "C" int cholamd_dpotrf_dev(int n, double *d_a, int lda, int *d_info, void *stream)
{
  int rc = have_device();
  if (rc) return rc;
  if (n < 0 || lda < (n > 1 ? n : 1)) { chol_set_error("dpotrf: bad n/lda"); return CHOLAMD_ERR_ARG; }
  if (n == 0) return 0;
  if ((rc = check_ptr(d_a, "a"))) return rc;
  std::vector<chol_potrf_desc> v(1);
  v[0] = { poff(d_a), 0, n, lda, 0, 0 };
  int info = 0;
  rc = run_potrf(v, (hipStream_t)stream, &info, nullptr);
  if (rc) return rc;
  if (d_info) HIPCHK(hipMemcpy(d_info, &info, sizeof(int), hipMemcpyHostToDevice));
  return info;
}
extern "C" int cholamd_dtrsm_dev(int m, int n, const double *d_a, int lda, double *d_b, int ldb, void *stream)
{
  int rc = have_device();
  if (rc) return rc;
  if (m < 0 || n < 0 || lda < (n > 1 ? n : 1) || ldb < (m > 1 ? m : 1)) { chol_set_error("dtrsm: bad sizes"); return CHOLAMD_ERR_ARG; }
  if (m == 0 || n == 0) return 0;
  if ((rc = check_ptr(d_a, "a")) || (rc = check_ptr(d_b, "b"))) return rc;
  std::vector<chol_trsm_desc> rows(1);
  rows[0] = { 0, 0, poff(d_b), n, lda, m, ldb };
  return run_trsm(d_a, n, lda, rows, (hipStream_t)stream);
}
extern "C" int cholamd_dgemm_dev(int m, int n, int k, const double *d_a, int lda, const double *d_b, int ldb, double *d_c, int ldc, void *stream)
{
  int rc = have_device();
  if (rc) return rc;
  if (m < 0 || n < 0 || k < 0 || lda < (m > 1 ? m : 1) || ldb < (n > 1 ? n : 1) || ldc < (m > 1 ? m : 1)) { chol_set_error("dgemm: bad sizes"); return CHOLAMD_ERR_ARG; }
  if (m == 0 || n == 0 || k == 0) return 0;
  if ((rc = check_ptr(d_a, "a")) || (rc = check_ptr(d_b, "b")) || (rc = check_ptr(d_c, "c"))) return rc;
  std::vector<chol_upd_task> tasks;
  std::vector<chol_upd_src> srcs(1);
  srcs[0] = { poff(d_a), poff(d_b), lda, ldb, k, 0 };
  add_update_tasks(tasks, 0, 1, d_c, ldc, m, n, false);
  return run_updates(tasks, srcs, (hipStream_t)stream);
}
extern "C" int cholamd_dsyrk_dev(int n, int k, const double *d_a, int lda, double *d_c, int ldc, void *stream)
{
  int rc = have_device();
  if (rc) return rc;
  if (n < 0 || k < 0 || lda < (n > 1 ? n : 1) || ldc < (n > 1 ? n : 1)) { chol_set_error("dsyrk: bad sizes"); return CHOLAMD_ERR_ARG; }
  if (n == 0 || k == 0) return 0;
  if ((rc = check_ptr(d_a, "a")) || (rc = check_ptr(d_c, "c"))) return rc;
  std::vector<chol_upd_task> tasks;
  std::vector<chol_upd_src> srcs(1);
  srcs[0] = { poff(d_a), poff(d_a), lda, lda, k, 0 };
  add_update_tasks(tasks, 0, 1, d_c, ldc, n, n, true);
  return run_updates(tasks, srcs, (hipStream_t)stream);
}
extern "C" int cholamd_dtrsv_dev(int trans, int n, const double *d_a, int lda, double *d_x, void *stream)
{
  int rc = have_device();
  if (rc) return rc;
  if (n < 0 || lda < (n > 1 ? n : 1) || (trans != CholamdNoTrans && trans != CholamdTrans)) { chol_set_error("dtrsv: bad arguments"); return CHOLAMD_ERR_ARG; }
  if (n == 0) return 0;
  if ((rc = check_ptr(d_a, "a")) || (rc = check_ptr(d_x, "x"))) return rc;
  hipStream_t st = (hipStream_t)stream;
  scratch sc;
  chol_trsv_desc t = { poff(d_a), n, lda, 0, 0 };
  chol_trsv_desc *dt;
  if ((rc = sc.put(&dt, &t, 1, st))) return rc;
  if (trans == CholamdNoTrans) {
    HIPCHK((hipError_t)chol_launch_trsv_fwd(nullptr, dt, 1, d_x, st));
  } else {
    int zero2[2] = { 0, 0 };
    int *ds;
    if ((rc = sc.put(&ds, zero2, 2, st))) return rc;
    HIPCHK((hipError_t)chol_launch_bwd(nullptr, dt, nullptr, ds, 1, d_x, st));
  }
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}
extern "C" int cholamd_dgemv_dev(int trans, int m, int n, const double *d_a, int lda, const double *d_x, double *d_y, void *stream)
{
  int rc = have_device();
  if (rc) return rc;
  if (m < 0 || n < 0 || lda < (m > 1 ? m : 1) || (trans != CholamdNoTrans && trans != CholamdTrans)) { chol_set_error("dgemv: bad arguments"); return CHOLAMD_ERR_ARG; }
  if (m == 0 || n == 0) return 0;
  if ((rc = check_ptr(d_a, "a")) || (rc = check_ptr(d_x, "x")) || (rc = check_ptr(d_y, "y"))) return rc;
  hipStream_t st = (hipStream_t)stream;
  scratch sc;
  if (trans == CholamdNoTrans) { // y(m) -= A x(n): the forward kernel with pointer-valued offsets
    std::vector<chol_gemv_desc> g;
    std::vector<int> gs, gr;
    for (int row0 = 0; row0 < m; row0 += 256) {
      gs.push_back((int)g.size());
      gr.push_back(row0); gr.push_back(0);
      chol_gemv_desc d = { poff(d_a), m, n, lda, 0, 0 };
      g.push_back(d);
    }
    gs.push_back((int)g.size());
    // vectors are addressed relative to y: x_off = x - y in doubles
    for (auto &d : g) d.x_off = (int)(poff(d_x) - poff(d_y));
    chol_gemv_desc *dg; int *dgs, *dgr;
    if ((rc = sc.put(&dg, g.data(), g.size(), st)) || (rc = sc.put(&dgs, gs.data(), gs.size(), st)) || (rc = sc.put(&dgr, gr.data(), gr.size(), st))) return rc;
    if (poff(d_x) - poff(d_y) != (int64_t)(int)(poff(d_x) - poff(d_y))) { chol_set_error("dgemv: x and y too far apart"); return CHOLAMD_ERR_ARG; }
    HIPCHK((hipError_t)chol_launch_gemv_fwd(nullptr, dg, dgs, dgr, (int)gs.size() - 1, d_y, st));
  } else { // y(n) -= A^T x(m): the backward kernel with an empty triangle (n = 0) does only the gather
    chol_trsv_desc t = { 0, 0, 1, 0, 0 };
    chol_gemv_desc d = { poff(d_a), m, n, lda, (int)(poff(d_x) - poff(d_y)), 0 };
    if (poff(d_x) - poff(d_y) != (int64_t)d.x_off) { chol_set_error("dgemv: x and y too far apart"); return CHOLAMD_ERR_ARG; }
    int start[2] = { 0, 1 };
    chol_trsv_desc *dt; chol_gemv_desc *dg; int *ds;
    if ((rc = sc.put(&dt, &t, 1, st)) || (rc = sc.put(&dg, &d, 1, st)) || (rc = sc.put(&ds, start, 2, st))) return rc;
    HIPCHK((hipError_t)chol_launch_bwd(nullptr, dt, dg, ds, 1, d_y, st));
  }
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

// ---- L-A: host-pointer CBLAS / LAPACKE replacements -------------------------------------------
static thread_local int g_blas_status = 0;
extern "C" int cholamd_blas_status(void) { return g_blas_status; }
extern "C" void cholamd_openblas_set_num_threads(int) {} // mmat.rg:1057: parallelism is the GPU's

struct host_mat { // a host matrix mirrored on the device for the duration of one call
  double *d = nullptr; double *h = nullptr; int rows = 0, cols = 0, ld = 0; bool writeback = false;
  int up(const double *hp, int r, int c, int ldh, bool wb)
  {
    h = const_cast<double *>(hp); rows = r; cols = c; ld = ldh; writeback = wb;
    if (r == 0 || c == 0) return 0;
    HIPCHK(hipMalloc((void **)&d, (size_t)ld * cols * sizeof(double)));
    HIPCHK(hipMemcpy(d, h, ((size_t)ld * (cols - 1) + rows) * sizeof(double), hipMemcpyHostToDevice));
    return 0;
  }
  int down()
  {
    if (d && writeback) HIPCHK(hipMemcpy2D(h, (size_t)ld * sizeof(double), d, (size_t)ld * sizeof(double), (size_t)rows * sizeof(double), cols, hipMemcpyDeviceToHost));
    return 0;
  }
  ~host_mat() { if (d) (void)hipFree(d); }
};

extern "C" int cholamd_LAPACKE_dpotrf(int layout, char uplo, int n, double *a, int lda)
{
  g_blas_status = 0;
  if (layout != CholamdColMajor || (uplo != 'L' && uplo != 'l')) { chol_set_error("LAPACKE_dpotrf: only ColMajor/'L' (blas.rg:71)"); return g_blas_status = CHOLAMD_ERR_ARG; }
  int rc = have_device();
  if (rc) return g_blas_status = rc;
  if (n == 0) return 0;
  host_mat A;
  if ((rc = A.up(a, n, n, lda, true))) return g_blas_status = rc;
  int info = cholamd_dpotrf_dev(n, A.d, lda, nullptr, nullptr);
  if (info < 0) return g_blas_status = info;
  if ((rc = A.down())) return g_blas_status = rc;
  return info;
}
extern "C" void cholamd_cblas_dtrsm(int layout, int side, int uplo, int transa, int diag, int m, int n, double alpha,
                                    const double *a, int lda, double *b, int ldb)
{
  g_blas_status = 0;
  if (layout != CholamdColMajor || side != CholamdRight || uplo != CholamdLower || transa != CholamdTrans || diag != CholamdNonUnit || alpha != 1.0) {
    chol_set_error("cblas_dtrsm: only ColMajor/Right/Lower/Trans/NonUnit/alpha=1 (blas.rg:99)");
    g_blas_status = CHOLAMD_ERR_ARG; return;
  }
  int rc = have_device();
  if (rc) { g_blas_status = rc; return; }
  host_mat A, B;
  if ((rc = A.up(a, n, n, lda, false)) || (rc = B.up(b, m, n, ldb, true))) { g_blas_status = rc; return; }
  if ((rc = cholamd_dtrsm_dev(m, n, A.d, lda, B.d, ldb, nullptr)) || (rc = B.down())) g_blas_status = rc;
}
extern "C" void cholamd_cblas_dgemm(int layout, int transa, int transb, int m, int n, int k, double alpha, const double *a, int lda,
                                    const double *b, int ldb, double beta, double *c, int ldc)
{
  g_blas_status = 0;
  if (layout != CholamdColMajor || transa != CholamdNoTrans || transb != CholamdTrans || alpha != -1.0 || beta != 1.0) {
    chol_set_error("cblas_dgemm: only ColMajor/NoTrans/Trans/alpha=-1/beta=1 (blas.rg:139)");
    g_blas_status = CHOLAMD_ERR_ARG; return;
  }
  int rc = have_device();
  if (rc) { g_blas_status = rc; return; }
  host_mat A, B, C;
  if ((rc = A.up(a, m, k, lda, false)) || (rc = B.up(b, n, k, ldb, false)) || (rc = C.up(c, m, n, ldc, true))) { g_blas_status = rc; return; }
  if ((rc = cholamd_dgemm_dev(m, n, k, A.d, lda, B.d, ldb, C.d, ldc, nullptr)) || (rc = C.down())) g_blas_status = rc;
}
extern "C" void cholamd_cblas_dsyrk(int layout, int uplo, int trans, int n, int k, double alpha, const double *a, int lda,
                                    double beta, double *c, int ldc)
{
  g_blas_status = 0;
  if (layout != CholamdColMajor || uplo != CholamdLower || trans != CholamdNoTrans || alpha != -1.0 || beta != 1.0) {
    chol_set_error("cblas_dsyrk: only ColMajor/Lower/NoTrans/alpha=-1/beta=1 (blas.rg:187)");
    g_blas_status = CHOLAMD_ERR_ARG; return;
  }
  int rc = have_device();
  if (rc) { g_blas_status = rc; return; }
  host_mat A, C;
  if ((rc = A.up(a, n, k, lda, false)) || (rc = C.up(c, n, n, ldc, true))) { g_blas_status = rc; return; }
  if ((rc = cholamd_dsyrk_dev(n, k, A.d, lda, C.d, ldc, nullptr)) || (rc = C.down())) g_blas_status = rc;
}
extern "C" void cholamd_cblas_dtrsv(int layout, int uplo, int transa, int diag, int n, const double *a, int lda, double *x, int incx)
{
  g_blas_status = 0;
  if (layout != CholamdColMajor || uplo != CholamdLower || diag != CholamdNonUnit || incx != 1 || (transa != CholamdNoTrans && transa != CholamdTrans)) {
    chol_set_error("cblas_dtrsv: only ColMajor/Lower/NonUnit/incx=1 (blas.rg:226)");
    g_blas_status = CHOLAMD_ERR_ARG; return;
  }
  int rc = have_device();
  if (rc) { g_blas_status = rc; return; }
  host_mat A, X;
  if ((rc = A.up(a, n, n, lda, false)) || (rc = X.up(x, n, 1, n > 1 ? n : 1, true))) { g_blas_status = rc; return; }
  if ((rc = cholamd_dtrsv_dev(transa, n, A.d, lda, X.d, nullptr)) || (rc = X.down())) g_blas_status = rc;
}
extern "C" void cholamd_cblas_dgemv(int layout, int trans, int m, int n, double alpha, const double *a, int lda, const double *x, int incx,
                                    double beta, double *y, int incy)
{
  g_blas_status = 0;
  if (layout != CholamdColMajor || alpha != -1.0 || beta != 1.0 || incx != 1 || incy != 1 || (trans != CholamdNoTrans && trans != CholamdTrans)) {
    chol_set_error("cblas_dgemv: only ColMajor/alpha=-1/beta=1/inc=1 (blas.rg:263)");
    g_blas_status = CHOLAMD_ERR_ARG; return;
  }
  int rc = have_device();
  if (rc) { g_blas_status = rc; return; }
  const int lx = trans == CholamdNoTrans ? n : m, ly = trans == CholamdNoTrans ? m : n;
  if (m == 0 || n == 0) return;
  // x and y share one device buffer so that 32-bit relative offsets always suffice
  double *xy = nullptr;
  if (hipMalloc((void **)&xy, (size_t)(lx + ly) * sizeof(double)) != hipSuccess) { g_blas_status = CHOLAMD_ERR_HIP; return; }
  host_mat A;
  rc = A.up(a, m, n, lda, false);
  if (!rc && hipMemcpy(xy, x, (size_t)lx * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = CHOLAMD_ERR_HIP;
  if (!rc && hipMemcpy(xy + lx, y, (size_t)ly * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = CHOLAMD_ERR_HIP;
  if (!rc) rc = cholamd_dgemv_dev(trans, m, n, A.d, lda, xy, xy + lx, nullptr);
  if (!rc && hipMemcpy(y, xy + lx, (size_t)ly * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = CHOLAMD_ERR_HIP;
  (void)hipFree(xy);
  g_blas_status = rc;
}

// ---------------------------------------------------------------------------------------------
// Multi-GPU: subtree sharding + ONE extend-add exchange over RCCL.
// The separator tree is cut at level d = log2(world): rank g owns the subtrees under its level-d separator and
// factors them locally; contributions to the shared top of the tree accumulate in the rank's own copy of the
// arena tail (the top panels are the contiguous tail of the arena: chol_plan.h); one ncclAllReduce(sum) over
// that tail is the exchange; the top levels follow on every rank.
// ---------------------------------------------------------------------------------------------
// A LOCAL communicator (cholamd_comm_create_local) joins rank objects of ONE process without RCCL: the exchange is a device-side
// ordered sum of the tails and peer copies, events order the ranks' streams.  The ranks may share a device (how the one-GPU test
// box runs world 2 ... 8); usable through cholamd_factor_multi only, which sees every rank's arena.
#define CHOL_LOCAL_MAX 64
struct local_group {
  int n = 0, refs = 0;
  std::vector<int> dev;
  std::vector<hipEvent_t> ev; // [g]: rank g's stream has reached the exchange; [n + g]: rank g's part of the exchange is enqueued
};
struct cholamd_comm {
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0;
  bool owned = true;
  local_group *local = nullptr;
};
#define NCCLCHK(call)                                                                                 \
  do {                                                                                                \
    ncclResult_t r_ = (call);                                                                         \
    if (r_ != ncclSuccess) {                                                                          \
      chol_set_error("%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_), __FILE__, __LINE__);     \
      return CHOLAMD_ERR_COMM;                                                                        \
    }                                                                                                 \
  } while (0)

extern "C" int cholamd_comm_unique_id(char id[CHOLAMD_UNIQUE_ID_BYTES])
{
  static_assert(sizeof(ncclUniqueId) <= CHOLAMD_UNIQUE_ID_BYTES, "unique id size");
  ncclUniqueId u;
  NCCLCHK(ncclGetUniqueId(&u));
  std::memset(id, 0, CHOLAMD_UNIQUE_ID_BYTES);
  std::memcpy(id, &u, sizeof u);
  return 0;
}
extern "C" int cholamd_comm_create(cholamd_device *d, int world, int rank, const char id[CHOLAMD_UNIQUE_ID_BYTES], cholamd_comm **out)
{
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) { chol_set_error("bad communicator rank %d of %d", rank, world); return CHOLAMD_ERR_ARG; }
  HIPCHK(hipSetDevice(d->dev));
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  cholamd_comm *c = new cholamd_comm();
  c->world = world; c->rank = rank;
  ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank);
  if (r != ncclSuccess) { chol_set_error("ncclCommInitRank failed: %s", ncclGetErrorString(r)); delete c; return CHOLAMD_ERR_COMM; }
  *out = c;
  return 0;
}
extern "C" int cholamd_comm_create_all(cholamd_device *const *devs, int n, cholamd_comm **out)
{ // one process driving n devices (cholamd_mmat --gpus n)
  if (n < 1) { chol_set_error("no devices"); return CHOLAMD_ERR_ARG; }
  std::vector<int> ids(n);
  std::vector<ncclComm_t> comms(n);
  for (int i = 0; i < n; i++) ids[i] = devs[i]->dev;
  NCCLCHK(ncclCommInitAll(comms.data(), n, ids.data()));
  for (int i = 0; i < n; i++) { out[i] = new cholamd_comm(); out[i]->comm = comms[i]; out[i]->world = n; out[i]->rank = i; }
  return 0;
}
extern "C" int cholamd_comm_create_local(cholamd_device *const *devs, int n, cholamd_comm **out)
{ // one process, n rank objects, no RCCL: device-side sums and peer copies (the ranks may share a device)
  if (n < 1 || n > CHOL_LOCAL_MAX) { chol_set_error("local communicator of %d ranks (1 .. %d)", n, CHOL_LOCAL_MAX); return CHOLAMD_ERR_ARG; }
  local_group *G = new local_group();
  G->n = n; G->refs = n; G->dev.resize(n); G->ev.resize(2 * n);
  for (int i = 0; i < n; i++) G->dev[i] = devs[i]->dev;
  for (int i = 0; i < n; i++) {
    HIPCHK(hipSetDevice(G->dev[i]));
    HIPCHK(hipEventCreateWithFlags(&G->ev[i], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&G->ev[n + i], hipEventDisableTiming));
    for (int j = 0; j < n; j++) if (G->dev[j] != G->dev[i]) {
      hipError_t e = hipDeviceEnablePeerAccess(G->dev[j], 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { chol_set_error("no peer access from device %d to device %d: %s", G->dev[i], G->dev[j], hipGetErrorString(e)); return CHOLAMD_ERR_COMM; }
      (void)hipGetLastError();
    }
  }
  for (int i = 0; i < n; i++) { out[i] = new cholamd_comm(); out[i]->world = n; out[i]->rank = i; out[i]->owned = false; out[i]->local = G; }
  return 0;
}
extern "C" int cholamd_comm_adopt(void *nccl_comm, int world, int rank, cholamd_comm **out)
{ // an ncclComm_t the caller made (and keeps): e.g. the one a Legion mapper or torch's process group holds
  if (!nccl_comm) { chol_set_error("null ncclComm_t"); return CHOLAMD_ERR_ARG; }
  cholamd_comm *c = new cholamd_comm();
  c->comm = (ncclComm_t)nccl_comm; c->world = world; c->rank = rank; c->owned = false;
  *out = c;
  return 0;
}
extern "C" int cholamd_comm_count(const cholamd_comm *c, int *ranks_out)
{ // ncclCommCount of the RCCL communicator behind the handle (a local communicator: its rank objects)
  if (!c) { chol_set_error("null communicator"); return CHOLAMD_ERR_ARG; }
  if (!c->comm) { *ranks_out = c->local ? c->local->n : 0; return 0; }
  int n = 0;
  NCCLCHK(ncclCommCount(c->comm, &n));
  *ranks_out = n;
  return 0;
}
extern "C" void cholamd_comm_destroy(cholamd_comm *c)
{
  if (!c) return;
  if (c->owned && c->comm) (void)ncclCommDestroy(c->comm);
  if (c->local && --c->local->refs == 0) {
    for (size_t i = 0; i < c->local->ev.size(); i++) { (void)hipSetDevice(c->local->dev[i % c->local->n]); (void)hipEventDestroy(c->local->ev[i]); }
    delete c->local;
  }
  delete c;
}
static int tail_of(const cholamd_device *d, int64_t *tail, int64_t *count)
{
  const cholamd_plan *p = d->plan;
  *tail = d->world > 1 ? p->panel_off[p->nsep - (d->world - 1) + 1] : p->arena;
  *count = p->arena - *tail;
  return 0;
}
extern "C" int64_t cholamd_device_tail_offset(const cholamd_device *d)
{
  int64_t t, c;
  tail_of(d, &t, &c);
  return t;
}
extern "C" int cholamd_comm_allreduce(cholamd_comm *c, double *d_buf, int64_t count, void *stream)
{ // in-place fp64 sum over the communicator's ranks, asynchronous on `stream` (of the current device)
  if (!c || !c->comm) { chol_set_error(c && c->local ? "a local communicator exchanges through cholamd_factor_multi only" : "null communicator"); return CHOLAMD_ERR_ARG; }
  if (count <= 0) return 0;
  NCCLCHK(ncclAllReduce(d_buf, d_buf, (size_t)count, ncclDouble, ncclSum, c->comm, (hipStream_t)stream));
  return 0;
}
static int comm_matches(const cholamd_device *d, const cholamd_comm *c)
{
  if (c && c->world == d->world && c->rank == d->rank) return 0;
  chol_set_error("communicator (rank %d of %d) does not match the device partition (rank %d of %d)", c ? c->rank : -1, c ? c->world : -1, d->rank, d->world);
  return CHOLAMD_ERR_ARG;
}
// ---- the extend-add exchange ------------------------------------------------------------------
// Replicated top levels: every rank needs the whole summed tail -> ONE in-place all-reduce.
// Top levels distributed by column blocks (option dist_top): after the exchange a rank works only on the column blocks it OWNS (it
// factors and solves them, applies the updates into them); every other block reaches it factored, by broadcast.  So the sum of a
// block is needed on its owner alone: every rank SENDS its partial copy of each block it does not own straight to the owner
// (grouped ncclSend / ncclRecv: point-to-point, all seven xGMI links of a GPU at once instead of a ring's one per direction), the
// owner adds the world - 1 copies it received to its own in RANK ORDER (deterministic: the same sum in every run).  A rank receives
// (world - 1) x (its owned blocks) and sends the blocks it does not own once: (world - 1) / world of the tail each way, against
// 2 (world - 1) / world of it each way for the ring all-reduce -- and nothing is added twice on the way.
// PATH-AWARE (round 4): a rank's subtree reaches only the top separators on its own path to the root (the reference touches a C tile only
// where the fill marks it, blas.rg:385-395, and the fill of a top block comes from the subtrees under it): its copy of every other
// top panel is zero -- unless it is rank 0, whose arena starts from A there.  So a rank SENDS only the blocks of the separators on its
// path (rank 0: all) and an owner RECEIVES only from the ranks under the block's separator and from rank 0 (chol_top_contributors);
// both ends compute the same lists from the plan, nothing about them travels.
struct xpiece { int64_t off, count, stage; int owner; unsigned contrib; };
// the column blocks of the levels above the cut, with their owners (the broadcast lists of the distributed top levels); empty: replicated
static void exchange_pieces(const cholamd_device *d, const std::vector<level_dev> &lv, std::vector<xpiece> &out)
{
  out.clear();
  const int split = chol_split_level(d->world);
  for (int lvl = split - 1; lvl >= 0 && lvl < (int)lv.size(); lvl--)
    for (const chol_bcast &b : lv[lvl].bcast) out.push_back({ b.off, b.count, 0, b.owner, chol_top_contributors(b.heap, d->world) });
  int64_t st = 0; // staging slots of the pieces this rank owns: one copy per contributing rank other than itself, senders in rank order
  for (xpiece &x : out) if (x.owner == d->rank) { x.stage = st; st += x.count * __builtin_popcount(x.contrib & ~(1u << d->rank)); }
}
static int top_entries(cholamd_device *d, int f32, int64_t *below, const int64_t **tdst, const double **tval, int64_t *ntop)
{
  const cholamd_plan *p = d->plan;
  *tdst = nullptr; *tval = nullptr; *ntop = 0;
  if (d->world <= 1) { *below = p->nnz_a; return 0; }
  const int64_t tail = p->panel_off[p->nsep - (d->world - 1) + 1];
  int64_t lo = 0, hi = p->nnz_a;
  while (lo < hi) { int64_t mid = (lo + hi) / 2; if (p->a_dst[mid] < tail) lo = mid + 1; else hi = mid; }
  *below = lo;
  if (d->top_gen[f32] != d->sched_gen) { // once per schedule
    (void)hipFree(d->top_dst[f32]); (void)hipFree(d->top_val[f32]); d->top_dst[f32] = nullptr; d->top_val[f32] = nullptr; d->top_n[f32] = 0;
    std::vector<xpiece> px;
    exchange_pieces(d, f32 ? d->lv32 : d->lv, px);
    if (px.empty()) d->top_n[f32] = d->rank == 0 ? -1 : 0; // replicated top levels: rank 0 scatters the whole tail
    else {
      std::vector<xpiece> mine;
      for (const xpiece &x : px) if (x.owner == d->rank) mine.push_back(x);
      std::sort(mine.begin(), mine.end(), [](const xpiece &a, const xpiece &b) { return a.off < b.off; });
      std::vector<int64_t> td; std::vector<double> tv;
      size_t q = 0;
      for (int64_t e = lo; e < p->nnz_a; e++) { // a_dst ascends: one merge pass over the owned blocks
        while (q < mine.size() && mine[q].off + mine[q].count <= p->a_dst[e]) q++;
        if (q < mine.size() && p->a_dst[e] >= mine[q].off) { td.push_back(p->a_dst[e]); tv.push_back(p->a_val[e]); }
      }
      if (!td.empty()) {
        int rc = upload_vec(&d->top_dst[f32], td.data(), td.size());
        if (!rc) rc = upload_vec(&d->top_val[f32], tv.data(), tv.size());
        if (rc) return rc;
      }
      d->top_n[f32] = (int64_t)td.size();
    }
    d->top_gen[f32] = d->sched_gen;
  }
  if (d->top_n[f32] < 0) *below = p->nnz_a;
  else { *tdst = d->top_dst[f32]; *tval = d->top_val[f32]; *ntop = d->top_n[f32]; }
  return 0;
}
struct sum_desc { int64_t off, count, stage; unsigned contrib, pad; };
template <class T> __global__ void k_sum_owned(T *arena, const T *stage, const sum_desc *desc, int world, int rank)
{ // arena[off + i] = sum over the ranks in rank order; rank r's copy: this rank's own arena for r == rank, else -- if r contributes -- the next staging slot
  const sum_desc d = desc[blockIdx.y];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.count; i += (int64_t)gridDim.x * blockDim.x) {
    T s = 0;
    int slot = 0;
    for (int r = 0; r < world; r++) {
      if (r == rank) s += arena[d.off + i];
      else if (d.contrib & (1u << r)) { s += stage[d.stage + (int64_t)slot * d.count + i]; slot++; }
    }
    arena[d.off + i] = s;
  }
}
// The rank-ordered sum of what exchange_owned() received: it must reach the stream AFTER the sends and receives, and those reach it only
// when the OUTERMOST RCCL group ends -- a caller that groups the exchanges of several ranks (factor_multi_t) posts them all with
// `defer_sum` and calls this once its group has ended.
template <class T> static int exchange_owned_sum(cholamd_device *d, T *arena, const std::vector<xpiece> &px, hipStream_t st)
{
  int64_t mx = 0; unsigned mine = 0;
  for (const xpiece &x : px) if (x.owner == d->rank) { mx = x.count > mx ? x.count : mx; mine++; }
  if (!mine) return 0;
  HIPCHK(hipSetDevice(d->dev));
  const int64_t bx = (mx + 255) / 256;
  hipLaunchKernelGGL(k_sum_owned<T>, dim3((unsigned)(bx < 1024 ? bx : 1024), mine), dim3(256), 0, st, arena, (const T *)d->xstage, (const sum_desc *)d->xdesc, d->world, d->rank);
  HIPCHK(hipGetLastError());
  return 0;
}
template <class T> static int exchange_owned(cholamd_device *d, T *arena, const std::vector<xpiece> &px, cholamd_comm *c, hipStream_t st, bool defer_sum)
{
  const ncclDataType_t ty = sizeof(T) == 8 ? ncclDouble : ncclFloat;
  int64_t need = 0; std::vector<sum_desc> mine;
  for (const xpiece &x : px) if (x.owner == d->rank) { need += x.count * __builtin_popcount(x.contrib & ~(1u << d->rank)); mine.push_back({ x.off, x.count, x.stage, x.contrib, 0 }); }
  if ((size_t)need * sizeof(T) > d->xstage_bytes) {
    HIPCHK(hipStreamSynchronize(st));
    (void)hipFree(d->xstage); d->xstage = nullptr; d->xstage_bytes = 0;
    HIPCHK(hipMalloc(&d->xstage, (size_t)need * sizeof(T)));
    d->xstage_bytes = (size_t)need * sizeof(T);
  }
  if (d->xdesc_gen != d->sched_gen || d->xdesc_elem != (int)sizeof(T)) { // the descriptors of the sum kernel, once per schedule
    HIPCHK(hipStreamSynchronize(st));
    (void)hipFree(d->xdesc); d->xdesc = nullptr;
    if (!mine.empty()) { int rc = upload_vec((sum_desc **)&d->xdesc, mine.data(), mine.size()); if (rc) return rc; }
    d->xdesc_gen = d->sched_gen; d->xdesc_elem = (int)sizeof(T);
  }
  T *stage = (T *)d->xstage;
  NCCLCHK(ncclGroupStart());
  for (const xpiece &x : px) {
    ncclResult_t r = ncclSuccess;
    if (x.owner == d->rank) {
      for (int q = 0, slot = 0; q < d->world && r == ncclSuccess; q++) {
        if (q == d->rank || !(x.contrib & (1u << q))) continue;
        r = ncclRecv(stage + x.stage + (int64_t)slot * x.count, (size_t)x.count, ty, q, c->comm, st);
        slot++;
      }
    } else if (x.contrib & (1u << d->rank)) r = ncclSend(arena + x.off, (size_t)x.count, ty, x.owner, c->comm, st);
    if (r != ncclSuccess) { (void)ncclGroupEnd(); chol_set_error("ncclSend/ncclRecv failed: %s", ncclGetErrorString(r)); return CHOLAMD_ERR_COMM; }
  }
  NCCLCHK(ncclGroupEnd());
  return defer_sum ? 0 : exchange_owned_sum<T>(d, arena, px, st);
}
// defer_sum: the caller holds an RCCL group open around this call and runs exchange_sum_t() after closing it
template <class T> static int exchange_tail_t(cholamd_device *d, T *arena, const std::vector<level_dev> &lv, cholamd_comm *c, hipStream_t st, bool defer_sum = false)
{
  HIPCHK(hipSetDevice(d->dev));
  if (comm_matches(d, c)) return CHOLAMD_ERR_ARG;
  if (!c->comm) { chol_set_error("a local communicator exchanges through cholamd_factor_multi only"); return CHOLAMD_ERR_ARG; }
  std::vector<xpiece> px;
  exchange_pieces(d, lv, px);
  if (!px.empty()) return exchange_owned<T>(d, arena, px, c, st, defer_sum);
  int64_t tail, count;
  tail_of(d, &tail, &count);
  if (count <= 0) return 0;
  NCCLCHK(ncclAllReduce(arena + tail, arena + tail, (size_t)count, sizeof(T) == 8 ? ncclDouble : ncclFloat, ncclSum, c->comm, st));
  return 0;
}
template <class T> static int exchange_sum_t(cholamd_device *d, T *arena, const std::vector<level_dev> &lv, hipStream_t st)
{ // second half of exchange_tail_t(..., defer_sum = true)
  std::vector<xpiece> px;
  exchange_pieces(d, lv, px);
  return px.empty() ? 0 : exchange_owned_sum<T>(d, arena, px, st);
}
extern "C" int cholamd_exchange_tail(cholamd_device *d, double *d_arena, cholamd_comm *c, void *stream)
{
  return exchange_tail_t<double>(d, d_arena, d->lv, c, (hipStream_t)stream);
}
extern "C" int cholamd_exchange_volume(const cholamd_device *d, int64_t out[4])
{ // elements this rank receives / sends in the exchange of its partition, elements of the tail, pieces (0: one all-reduce of the tail:
  // a ring moves 2 (world - 1) / world of it each way)
  std::vector<xpiece> px;
  exchange_pieces(d, d->lv, px);
  int64_t tail, count;
  tail_of(d, &tail, &count);
  out[0] = out[1] = 0; out[2] = count; out[3] = (int64_t)px.size();
  for (const xpiece &x : px) { if (x.owner == d->rank) out[0] += x.count * __builtin_popcount(x.contrib & ~(1u << d->rank)); else if (x.contrib & (1u << d->rank)) out[1] += x.count; }
  if (px.empty() && d->world > 1) out[0] = out[1] = 2 * count * (d->world - 1) / d->world;
  return 0;
}
// one rank's broadcasts of a phase of kind 6 (distributed top levels): every column block of the step travels from its owner to all
// ranks, in place at the same arena offset; one RCCL group
template <class T> static int bcast_rank_t(cholamd_device *d, const level_dev &l, const chol_phase &ph, T *d_arena, cholamd_comm *c, hipStream_t st)
{
  if (!c) { chol_set_error("the distributed top levels (option dist_top) need a communicator: cholamd_factor_sharded / cholamd_factor_multi"); return CHOLAMD_ERR_ARG; }
  if (comm_matches(d, c)) return CHOLAMD_ERR_ARG;
  if (!c->comm) { chol_set_error("a local communicator exchanges through cholamd_factor_multi only"); return CHOLAMD_ERR_ARG; }
  NCCLCHK(ncclGroupStart());
  for (int i = ph.first; i < ph.first + ph.n; i++) {
    const chol_bcast &b = l.bcast[i];
    ncclResult_t r = ncclBroadcast(d_arena + b.off, d_arena + b.off, (size_t)b.count, sizeof(T) == 8 ? ncclDouble : ncclFloat, b.owner, c->comm, st);
    if (r != ncclSuccess) { (void)ncclGroupEnd(); chol_set_error("ncclBroadcast failed: %s", ncclGetErrorString(r)); return CHOLAMD_ERR_COMM; }
  }
  NCCLCHK(ncclGroupEnd());
  return 0;
}
static int bcast_rank(cholamd_device *d, const level_dev &l, const chol_phase &ph, double *d_arena, cholamd_comm *c, hipStream_t st)
{
  return bcast_rank_t<double>(d, l, ph, d_arena, c, st);
}
static int factor_levels_f32_comm(cholamd_device *d, float *d_arena32, int level_hi, int level_lo, cholamd_comm *c, hipStream_t st);
extern "C" int cholamd_factor_sharded(cholamd_device *d, double *d_arena, cholamd_comm *c, void *stream)
{
  const int L = d->plan->levels, split = chol_split_level(d->world);
  if (d->world == 1) return cholamd_factor(d, d_arena, stream);
  int rc = cholamd_factor_levels(d, d_arena, L - 1, split, stream);
  if (!rc) {
    scoped_timer t(d, (hipStream_t)stream, CHOL_TK_EXCHANGE, true);
    rc = cholamd_exchange_tail(d, d_arena, c, stream);
  }
  if (!rc) rc = factor_levels_comm(d, d_arena, split - 1, 0, c, (hipStream_t)stream);
  return rc;
}
// the same with the fp32 factor (BASELINE config 5: mixed precision x multi-GPU): fp32 arena of the fp64 arena's element layout, the
// fp32 schedule partitioned like the fp64 one, the exchange and the broadcasts on floats
extern "C" int cholamd_factor_sharded_f32(cholamd_device *d, float *d_arena32, cholamd_comm *c, void *stream)
{
  const int L = d->plan->levels, split = chol_split_level(d->world);
  if (d->world == 1) return cholamd_factor_f32(d, d_arena32, stream);
  int rc = ensure_f32(d);
  if (!rc) rc = factor_levels_f32_comm(d, d_arena32, L - 1, split, nullptr, (hipStream_t)stream);
  if (!rc) {
    scoped_timer t(d, (hipStream_t)stream, CHOL_TK_EXCHANGE, true);
    rc = exchange_tail_t<float>(d, d_arena32, d->lv32, c, (hipStream_t)stream);
  }
  if (!rc) rc = factor_levels_f32_comm(d, d_arena32, split - 1, 0, c, (hipStream_t)stream);
  return rc;
}

// ---- one process driving n ranks ----
struct ptr_pack { void *p[CHOL_LOCAL_MAX]; };
template <class T> __global__ void k_sum_ranks(ptr_pack P, int n, T *dst, int64_t count)
{ // dst[i] = sum over the n copies in the order of the pack (rank order: run-to-run identical); dst may be one of them
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    T s = ((const T *)P.p[0])[i];
    for (int r = 1; r < n; r++) s += ((const T *)P.p[r])[i];
    dst[i] = s;
  }
}
static hipStream_t stream_of(void *const *streams, int g) { return streams ? (hipStream_t)streams[g] : nullptr; }
template <class T> static int local_exchange(cholamd_device *const *devs, T *const *arenas, const std::vector<level_dev> &lv0, local_group *G, int n, void *const *streams)
{
  std::vector<xpiece> px;
  exchange_pieces(devs[0], lv0, px);
  int64_t tail, count;
  tail_of(devs[0], &tail, &count);
  if (count <= 0) return 0;
  for (int g = 0; g < n; g++) { HIPCHK(hipSetDevice(devs[g]->dev)); HIPCHK(hipEventRecord(G->ev[g], stream_of(streams, g))); } // every rank's subtree levels are in its stream
  if (px.empty()) { // replicated top: the ordered sum on rank 0, peer copies to the others
    HIPCHK(hipSetDevice(devs[0]->dev));
    ptr_pack P;
    for (int g = 0; g < n; g++) { P.p[g] = arenas[g] + tail; if (g) HIPCHK(hipStreamWaitEvent(stream_of(streams, 0), G->ev[g], 0)); }
    const int64_t blocks = (count + 255) / 256;
    hipLaunchKernelGGL(k_sum_ranks<T>, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream_of(streams, 0), P, n, arenas[0] + tail, count);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(G->ev[n], stream_of(streams, 0)));
    for (int g = 1; g < n; g++) {
      HIPCHK(hipSetDevice(devs[g]->dev));
      HIPCHK(hipStreamWaitEvent(stream_of(streams, g), G->ev[n], 0));
      HIPCHK(hipMemcpyPeerAsync(arenas[g] + tail, devs[g]->dev, arenas[0] + tail, devs[0]->dev, (size_t)count * sizeof(T), stream_of(streams, g)));
      HIPCHK(hipEventRecord(G->ev[n + g], stream_of(streams, g)));
    }
    HIPCHK(hipSetDevice(devs[0]->dev));
    for (int g = 1; g < n; g++) HIPCHK(hipStreamWaitEvent(stream_of(streams, 0), G->ev[n + g], 0)); // rank 0 goes on writing its tail
    return 0;
  }
  // distributed top: every column block is summed (rank order) into its OWNER's arena by the owner's stream; the other ranks' copies of
  // it are read, not written.  Nobody goes on before every owner has read what it needs (the broadcasts overwrite those copies)
  for (int o = 0; o < n; o++) {
    HIPCHK(hipSetDevice(devs[o]->dev));
    for (int g = 0; g < n; g++) if (g != o) HIPCHK(hipStreamWaitEvent(stream_of(streams, o), G->ev[g], 0));
    for (const xpiece &x : px) {
      if (x.owner != o) continue;
      ptr_pack P; // the copies that can be non-zero (the same set the RCCL path sends) and the owner's own, in rank order
      int np = 0;
      for (int g = 0; g < n; g++) if (g == o || (x.contrib & (1u << g))) P.p[np++] = arenas[g] + x.off;
      const int64_t blocks = (x.count + 255) / 256;
      hipLaunchKernelGGL(k_sum_ranks<T>, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, stream_of(streams, o), P, np, arenas[o] + x.off, x.count);
      HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(G->ev[n + o], stream_of(streams, o)));
  }
  for (int g = 0; g < n; g++) {
    HIPCHK(hipSetDevice(devs[g]->dev));
    for (int o = 0; o < n; o++) if (o != g) HIPCHK(hipStreamWaitEvent(stream_of(streams, g), G->ev[n + o], 0));
  }
  return 0;
}
template <class T> static int local_bcast(cholamd_device *const *devs, T *const *arenas, local_group *G, int n, void *const *streams, const level_dev &l, const chol_phase &ph)
{
  std::vector<char> owner_ready(n, 0);
  for (int i = ph.first; i < ph.first + ph.n; i++) {
    const int o = l.bcast[i].owner;
    if (owner_ready[o]) continue;
    owner_ready[o] = 1;
    HIPCHK(hipSetDevice(devs[o]->dev));
    HIPCHK(hipEventRecord(G->ev[o], stream_of(streams, o))); // the owner's POTRF + TRSM of the step are in its stream
  }
  for (int g = 0; g < n; g++) {
    HIPCHK(hipSetDevice(devs[g]->dev));
    for (int o = 0; o < n; o++) if (owner_ready[o] && o != g) HIPCHK(hipStreamWaitEvent(stream_of(streams, g), G->ev[o], 0));
    for (int i = ph.first; i < ph.first + ph.n; i++) {
      const chol_bcast &b = l.bcast[i];
      if (b.owner == g) continue; // the owner does not touch the block again: the copies may read it while its stream goes on
      HIPCHK(hipMemcpyPeerAsync(arenas[g] + b.off, devs[g]->dev, arenas[b.owner] + b.off, devs[b.owner]->dev, (size_t)b.count * sizeof(T), stream_of(streams, g)));
    }
  }
  return 0;
}
static int launch_phase_f32(cholamd_device *d, const level_dev &l, const chol_phase &ph, float *d_arena32, hipStream_t st);
// T = double: the fp64 schedule (devs[g]->lv); T = float: the fp32 one (lv32)
template <class T> static int factor_multi_t(cholamd_device *const *devs, T *const *arenas, cholamd_comm *const *comms, int n, void *const *streams)
{ // one process, n ranks: everything is asynchronous on each rank's stream
  constexpr bool F32 = sizeof(T) == 4;
  if (n < 1) { chol_set_error("no devices"); return CHOLAMD_ERR_ARG; }
  const int L = devs[0]->plan->levels, split = chol_split_level(n);
  local_group *G = comms[0] ? comms[0]->local : nullptr;
  for (int g = 0; g < n; g++) {
    if (devs[g]->world != n || devs[g]->rank != g) { chol_set_error("device %d is not partitioned as rank %d of %d", g, g, n); return CHOLAMD_ERR_ARG; }
    if (comm_matches(devs[g], comms[g])) return CHOLAMD_ERR_ARG;
    if (comms[g]->local != G || (!G && !comms[g]->comm)) { chol_set_error("the %d communicators are not of one kind", n); return CHOLAMD_ERR_ARG; }
    int rc;
    if (F32) { rc = ensure_f32(devs[g]); if (!rc) rc = factor_levels_f32_comm(devs[g], (float *)arenas[g], L - 1, split, nullptr, stream_of(streams, g)); }
    else rc = cholamd_factor_levels(devs[g], (double *)arenas[g], L - 1, split, stream_of(streams, g));
    if (rc) return rc;
  }
  auto levels_of = [&](int g) -> const std::vector<level_dev> & { return F32 ? devs[g]->lv32 : devs[g]->lv; };
  if (G) { int rc = local_exchange<T>(devs, arenas, levels_of(0), G, n, streams); if (rc) return rc; }
  else {
    NCCLCHK(ncclGroupStart());
    for (int g = 0; g < n; g++) {
      int rc = exchange_tail_t<T>(devs[g], arenas[g], levels_of(g), comms[g], stream_of(streams, g), true);
      if (rc) { (void)ncclGroupEnd(); return rc; }
    }
    NCCLCHK(ncclGroupEnd());
    // the owners' sums go behind the sends and receives, which entered the streams only now (the end of the outermost group)
    for (int g = 0; g < n; g++) { int rc = exchange_sum_t<T>(devs[g], arenas[g], levels_of(g), stream_of(streams, g)); if (rc) return rc; }
  }
  // top levels: every rank runs its phases up to its next broadcast phase; the broadcasts of all ranks form one group
  for (int lvl = split - 1; lvl >= 0; lvl--) {
    std::vector<size_t> cur(n, 0);
    for (;;) {
      int at_bcast = 0;
      for (int g = 0; g < n; g++) {
        const level_dev &l = levels_of(g)[lvl];
        HIPCHK(hipSetDevice(devs[g]->dev));
        while (cur[g] < l.phase.size() && l.phase[cur[g]].kind != 6) {
          int rc = F32 ? launch_phase_f32(devs[g], l, l.phase[cur[g]], (float *)arenas[g], stream_of(streams, g)) : launch_phase(devs[g], l, l.phase[cur[g]], (double *)arenas[g], stream_of(streams, g));
          if (rc) return rc;
          cur[g]++;
        }
        if (cur[g] < l.phase.size()) at_bcast++;
      }
      if (at_bcast == 0) break;
      if (at_bcast != n) { chol_set_error("internal: the ranks disagree on the broadcast sequence of level %d", lvl); return CHOLAMD_ERR_ARG; }
      if (G) { int rc = local_bcast<T>(devs, arenas, G, n, streams, levels_of(0)[lvl], levels_of(0)[lvl].phase[cur[0]]); if (rc) return rc; }
      else {
        NCCLCHK(ncclGroupStart());
        for (int g = 0; g < n; g++) {
          const level_dev &l = levels_of(g)[lvl];
          const chol_phase &ph = l.phase[cur[g]];
          for (int i = ph.first; i < ph.first + ph.n; i++) {
            const chol_bcast &b = l.bcast[i];
            ncclResult_t r = ncclBroadcast(arenas[g] + b.off, arenas[g] + b.off, (size_t)b.count, F32 ? ncclFloat : ncclDouble, b.owner, comms[g]->comm, stream_of(streams, g));
            if (r != ncclSuccess) { (void)ncclGroupEnd(); chol_set_error("ncclBroadcast failed: %s", ncclGetErrorString(r)); return CHOLAMD_ERR_COMM; }
          }
        }
        NCCLCHK(ncclGroupEnd());
      }
      for (int g = 0; g < n; g++) cur[g]++;
    }
  }
  return 0;
}
extern "C" int cholamd_factor_multi(cholamd_device *const *devs, double *const *arenas, cholamd_comm *const *comms, int n, void *const *streams)
{
  if (n == 1) return cholamd_factor(devs[0], arenas[0], streams ? streams[0] : nullptr);
  return factor_multi_t<double>(devs, arenas, comms, n, streams);
}
extern "C" int cholamd_factor_multi_f32(cholamd_device *const *devs, float *const *arenas32, cholamd_comm *const *comms, int n, void *const *streams)
{
  if (n == 1) return cholamd_factor_f32(devs[0], arenas32[0], streams ? streams[0] : nullptr);
  return factor_multi_t<float>(devs, arenas32, comms, n, streams);
}
// one rank's part of the gather: the panels of the subtrees it owns travel to rank 0 (grouped ncclSend / ncclRecv), whose arena then
// holds the complete factor (for the solve / refinement and the writers).  elem_bytes = 8 (fp64 arena) or 4 (fp32)
extern "C" int cholamd_gather_to_root(cholamd_device *d, void *d_arena, int elem_bytes, cholamd_comm *c, void *stream)
{
  HIPCHK(hipSetDevice(d->dev));
  if (d->world == 1) return 0;
  if (comm_matches(d, c)) return CHOLAMD_ERR_ARG;
  if (!c->comm) { chol_set_error("a local communicator gathers through cholamd_gather_factor only"); return CHOLAMD_ERR_ARG; }
  if (elem_bytes != 4 && elem_bytes != 8) { chol_set_error("element size %d", elem_bytes); return CHOLAMD_ERR_ARG; }
  const cholamd_plan *p = d->plan;
  NCCLCHK(ncclGroupStart());
  for (int s = 1; s <= p->nsep; s++) {
    const int g = chol_owner_of(p, s, d->world);
    if (g <= 0 || (d->rank != 0 && d->rank != g)) continue;
    const int64_t off = p->panel_off[s], len = (s < p->nsep ? p->panel_off[s + 1] : p->arena) - off;
    char *ptr = (char *)d_arena + (size_t)off * elem_bytes;
    ncclResult_t r = d->rank == 0 ? ncclRecv(ptr, (size_t)len, elem_bytes == 8 ? ncclDouble : ncclFloat, g, c->comm, (hipStream_t)stream)
                                  : ncclSend(ptr, (size_t)len, elem_bytes == 8 ? ncclDouble : ncclFloat, 0, c->comm, (hipStream_t)stream);
    if (r != ncclSuccess) { (void)ncclGroupEnd(); chol_set_error("gather: %s", ncclGetErrorString(r)); return CHOLAMD_ERR_COMM; }
  }
  NCCLCHK(ncclGroupEnd());
  return 0;
}
static int gather_factor_bytes(cholamd_device *const *devs, void *const *arenas, int n, void *const *streams, int elem_bytes);
extern "C" int cholamd_gather_factor(cholamd_device *const *devs, double *const *arenas, int n, void *const *streams)
{
  return gather_factor_bytes(devs, (void *const *)arenas, n, streams, 8);
}
extern "C" int cholamd_gather_factor_f32(cholamd_device *const *devs, float *const *arenas32, int n, void *const *streams)
{
  return gather_factor_bytes(devs, (void *const *)arenas32, n, streams, 4);
}
static int gather_factor_bytes(cholamd_device *const *devs, void *const *arenas, int n, void *const *streams, int elem_bytes)
{ // the panels of the subtrees ranks 1..n-1 own -> device 0's arena (peer copies), so that it holds the complete factor
  const cholamd_plan *p = devs[0]->plan;
  for (int g = 1; g < n; g++) {
    HIPCHK(hipSetDevice(devs[g]->dev));
    HIPCHK(hipStreamSynchronize(streams ? (hipStream_t)streams[g] : nullptr));
  }
  HIPCHK(hipSetDevice(devs[0]->dev));
  for (int s = 1; s <= p->nsep; s++) {
    const int g = chol_owner_of(p, s, n);
    if (g <= 0) continue;
    const int64_t off = p->panel_off[s];
    const int64_t len = (s < p->nsep ? p->panel_off[s + 1] : p->arena) - off;
    HIPCHK(hipMemcpyPeerAsync((char *)arenas[0] + (size_t)off * elem_bytes, devs[0]->dev, (const char *)arenas[g] + (size_t)off * elem_bytes, devs[g]->dev, (size_t)len * elem_bytes, streams ? (hipStream_t)streams[0] : nullptr));
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Distributed solve (mmat.rg:1394-1479 sharded like the factorisation).  After cholamd_factor_sharded / _multi a rank's arena holds
// the panels of its own subtrees and the WHOLE factored top (replicated, or broadcast block by block under dist_top).  So: every rank sweeps its own
// subtrees forward (the contributions to the top accumulate in its copy of the top's part of y, which starts from b on rank 0 and from zero
// elsewhere), ONE sum of that part over the ranks, every rank solves the top forward and backward redundantly, sweeps its subtrees backward, and one
// sum of the permuted vector (every position is non-zero on exactly one rank) gives every rank the solution.  Only vectors travel: (world - 1) top
// separators' worth of doubles, then n doubles -- against the 17-33 GB of factor that cholamd_gather_to_root moved at 100^3.  The iterative
// refinement runs the same loop on every rank (residual in fp64 against A, replicated; the corrections solved as above): identical iterates.
// ---------------------------------------------------------------------------------------------
template <class TL> static int solve_sharded_t(cholamd_device *d, const TL *d_arena, const double *d_b, double *d_x, cholamd_comm *c, hipStream_t st)
{
  HIPCHK(hipSetDevice(d->dev));
  if (d->world == 1) return solve_streamed(d, d_arena, d_b, d_x, st);
  int rc = build_solve(d, d->rank, d->world);
  if (rc) return rc;
  if (comm_matches(d, c)) return CHOLAMD_ERR_ARG;
  if (!c->comm) { chol_set_error("a local communicator solves through cholamd_solve_multi only"); return CHOLAMD_ERR_ARG; }
  const int64_t t0 = top_vec_offset(d), n = d->plan->n;
  if ((rc = solve_phase(d, d_arena, d_b, d_x, 0, st))) return rc;
  NCCLCHK(ncclAllReduce(d->ytmp + t0, d->ytmp + t0, (size_t)(n - t0), ncclDouble, ncclSum, c->comm, st));
  if ((rc = solve_phase(d, d_arena, d_b, d_x, 1, st))) return rc;
  NCCLCHK(ncclAllReduce(d->ytmp, d->ytmp, (size_t)n, ncclDouble, ncclSum, c->comm, st));
  return solve_phase(d, d_arena, d_b, d_x, 2, st);
}
extern "C" int cholamd_solve_sharded(cholamd_device *d, const double *d_arena, const double *d_b, double *d_x, cholamd_comm *c, void *stream)
{
  return solve_sharded_t<double>(d, d_arena, d_b, d_x, c, (hipStream_t)stream);
}
extern "C" int cholamd_solve_sharded_f32(cholamd_device *d, const float *d_arena32, const double *d_b, double *d_x, cholamd_comm *c, void *stream)
{
  return solve_sharded_t<float>(d, d_arena32, d_b, d_x, c, (hipStream_t)stream);
}
extern "C" int cholamd_solve_refine_sharded(cholamd_device *d, const float *d_arena32, const double *d_b, double *d_x, int max_iter, double tol,
                                            int *iters_out, double *relres_out, cholamd_comm *c, void *stream)
{ // every rank calls it with the same b; every rank ends with the same x, iteration count and residual
  HIPCHK(hipSetDevice(d->dev));
  int rc = ensure_refine(d);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int n = d->plan->n;
  if (max_iter < 0) max_iter = 0;
  if ((rc = solve_sharded_t<float>(d, d_arena32, d_b, d_x, c, st))) return rc;
  keep_inverses_scope keep(&d, 1);
  double rel = 0.0;
  int it = 0;
  for (;; ++it) {
    if ((rc = residual_norm(d, d_b, d_x, d->rvec, &rel, st))) return rc;
    if (!(rel > tol) || it >= max_iter) break; // also leaves on NaN (the same decision on every rank: the iterates are identical)
    if ((rc = solve_sharded_t<float>(d, d_arena32, d->rvec, d->dxvec, c, st))) return rc;
    HIPCHK((hipError_t)chol_launch_axpy1(d_x, d->dxvec, n, st));
  }
  if (iters_out) *iters_out = it;
  if (relres_out) *relres_out = rel;
  if (rel != rel) { chol_set_error("iterative refinement produced NaN (fp32 factorisation broke down)"); return CHOLAMD_ERR_ARG; }
  return 0;
}
// one process driving n ranks: the two sums as an ordered device-side sum on rank 0's stream and peer copies back (local communicator), or grouped
// ncclAllReduce calls (cholamd_comm_create_all)
static int multi_sum_vec(cholamd_device *const *devs, cholamd_comm *const *comms, int n, void *const *streams, int64_t off, int64_t count)
{ // devs[g]->ytmp[off, off + count) <- sum over the ranks, on every rank
  if (count <= 0) return 0;
  local_group *G = comms[0]->local;
  if (!G) {
    NCCLCHK(ncclGroupStart());
    for (int g = 0; g < n; g++) {
      ncclResult_t r = ncclAllReduce(devs[g]->ytmp + off, devs[g]->ytmp + off, (size_t)count, ncclDouble, ncclSum, comms[g]->comm, stream_of(streams, g));
      if (r != ncclSuccess) { (void)ncclGroupEnd(); chol_set_error("ncclAllReduce failed: %s", ncclGetErrorString(r)); return CHOLAMD_ERR_COMM; }
    }
    NCCLCHK(ncclGroupEnd());
    return 0;
  }
  for (int g = 0; g < n; g++) { HIPCHK(hipSetDevice(devs[g]->dev)); HIPCHK(hipEventRecord(G->ev[g], stream_of(streams, g))); }
  HIPCHK(hipSetDevice(devs[0]->dev));
  ptr_pack P;
  for (int g = 0; g < n; g++) { P.p[g] = devs[g]->ytmp + off; if (g) HIPCHK(hipStreamWaitEvent(stream_of(streams, 0), G->ev[g], 0)); }
  const int64_t blocks = (count + 255) / 256;
  hipLaunchKernelGGL(k_sum_ranks<double>, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, stream_of(streams, 0), P, n, devs[0]->ytmp + off, count);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(G->ev[n], stream_of(streams, 0)));
  for (int g = 1; g < n; g++) {
    HIPCHK(hipSetDevice(devs[g]->dev));
    HIPCHK(hipStreamWaitEvent(stream_of(streams, g), G->ev[n], 0));
    HIPCHK(hipMemcpyPeerAsync(devs[g]->ytmp + off, devs[g]->dev, devs[0]->ytmp + off, devs[0]->dev, (size_t)count * sizeof(double), stream_of(streams, g)));
    HIPCHK(hipEventRecord(G->ev[n + g], stream_of(streams, g)));
  }
  HIPCHK(hipSetDevice(devs[0]->dev));
  for (int g = 1; g < n; g++) HIPCHK(hipStreamWaitEvent(stream_of(streams, 0), G->ev[n + g], 0)); // rank 0 goes on writing its vector
  return 0;
}
template <class TL> static int solve_multi_t(cholamd_device *const *devs, const TL *const *arenas, const double *const *bs, double *const *xs, cholamd_comm *const *comms,
                                             int n, void *const *streams)
{
  if (n < 1) { chol_set_error("no devices"); return CHOLAMD_ERR_ARG; }
  if (n == 1) { HIPCHK(hipSetDevice(devs[0]->dev)); return solve_streamed(devs[0], arenas[0], bs[0], xs[0], stream_of(streams, 0)); }
  local_group *G = comms[0] ? comms[0]->local : nullptr;
  for (int g = 0; g < n; g++) {
    if (devs[g]->world != n || devs[g]->rank != g) { chol_set_error("device %d is not partitioned as rank %d of %d", g, g, n); return CHOLAMD_ERR_ARG; }
    if (comm_matches(devs[g], comms[g])) return CHOLAMD_ERR_ARG;
    if (comms[g]->local != G || (!G && !comms[g]->comm)) { chol_set_error("the %d communicators are not of one kind", n); return CHOLAMD_ERR_ARG; }
    HIPCHK(hipSetDevice(devs[g]->dev));
    int rc = build_solve(devs[g], g, n);
    if (rc) return rc;
  }
  const int64_t t0 = top_vec_offset(devs[0]), nn = devs[0]->plan->n;
  for (int ph = 0; ph < 3; ph++) {
    for (int g = 0; g < n; g++) {
      HIPCHK(hipSetDevice(devs[g]->dev));
      int rc = solve_phase(devs[g], arenas[g], bs[g], xs[g], ph, stream_of(streams, g));
      if (rc) return rc;
    }
    if (ph < 2) { int rc = multi_sum_vec(devs, comms, n, streams, ph == 0 ? t0 : 0, ph == 0 ? nn - t0 : nn); if (rc) return rc; }
  }
  return 0;
}
extern "C" int cholamd_solve_multi(cholamd_device *const *devs, const double *const *arenas, const double *const *bs, double *const *xs, cholamd_comm *const *comms,
                                   int n, void *const *streams)
{
  return solve_multi_t<double>(devs, arenas, bs, xs, comms, n, streams);
}
extern "C" int cholamd_solve_refine_multi(cholamd_device *const *devs, const float *const *arenas32, const double *const *bs, double *const *xs, int max_iter, double tol,
                                          int *iters_out, double *relres_out, cholamd_comm *const *comms, int n, void *const *streams)
{
  if (n < 1) { chol_set_error("no devices"); return CHOLAMD_ERR_ARG; }
  if (max_iter < 0) max_iter = 0;
  for (int g = 0; g < n; g++) { HIPCHK(hipSetDevice(devs[g]->dev)); int rc = ensure_refine(devs[g]); if (rc) return rc; }
  int rc = solve_multi_t<float>(devs, arenas32, bs, xs, comms, n, streams);
  if (rc) return rc;
  std::vector<const double *> rv(n);
  std::vector<double *> dx(n);
  for (int g = 0; g < n; g++) { rv[g] = devs[g]->rvec; dx[g] = devs[g]->dxvec; }
  keep_inverses_scope keep(devs, n);
  double rel = 0.0;
  int it = 0;
  for (;; ++it) {
    for (int g = 0; g < n; g++) { // every rank forms the residual of its own (identical) iterate
      double rg = 0.0;
      HIPCHK(hipSetDevice(devs[g]->dev));
      if ((rc = residual_norm(devs[g], bs[g], xs[g], devs[g]->rvec, &rg, stream_of(streams, g)))) return rc;
      if (g == 0) rel = rg;
    }
    if (!(rel > tol) || it >= max_iter) break;
    if ((rc = solve_multi_t<float>(devs, arenas32, rv.data(), dx.data(), comms, n, streams))) return rc;
    for (int g = 0; g < n; g++) { HIPCHK(hipSetDevice(devs[g]->dev)); HIPCHK((hipError_t)chol_launch_axpy1(xs[g], devs[g]->dxvec, devs[g]->plan->n, stream_of(streams, g))); }
  }
  if (iters_out) *iters_out = it;
  if (relres_out) *relres_out = rel;
  if (rel != rel) { chol_set_error("iterative refinement produced NaN (fp32 factorisation broke down)"); return CHOLAMD_ERR_ARG; }
  return 0;
}
