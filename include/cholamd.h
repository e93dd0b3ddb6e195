/*
 * cholamd.h -- C ABI of the MI355X-native supernodal Cholesky hot path.
 *
 * This is the drop-in boundary for ONE path of syamajala/cholesky: the per-supernode
 * POTRF / TRSM / SYRK / GEMM frontal updates driven by the nested-dissection separator tree
 * (reference: mmat.rg:1227-1355 -> blas.rg:292-504 -> libcblas/liblapacke).  Plain pointers and
 * sizes only; no torch / HIP types appear in any signature (streams are passed as void*).
 *
 * Three nested levels are exported, each replacing a reference interface (file:line cited at each
 * declaration, paths relative to the reference tree):
 *
 *   L-A  BLAS level   -- what Terra links today (blas.rg:18-22, mmat.rg:29-30): the six CBLAS /
 *                        LAPACKE entry points with the exact parameter combinations the reference
 *                        uses.  Host pointers in, host pointers out (like the CPU library they
 *                        replace); `_dev` variants take device pointers + stream.
 *   L-B  task level   -- the four `fused_*` leaf tasks (blas.rg:292-504): one call = one task =
 *                        one batched HIP launch over the task's list of filled tiles.
 *   L-C  driver level -- what `main` does around them (mmat.rg:1097-1362): ingest, symbolic
 *                        analysis, A scatter, the level schedule, factor/solution writers, solve.
 *
 * All functions return 0 on success unless stated otherwise.  Errors are never silent: there is no
 * CPU fallback anywhere in this library -- if no HIP device is usable, every compute entry point
 * returns CHOLAMD_ERR_NO_DEVICE.
 */
#ifndef CHOLAMD_H
#define CHOLAMD_H

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ----------------------------------------------------------------------------------------- */
/* error codes                                                                                 */
/* ----------------------------------------------------------------------------------------- */
#define CHOLAMD_OK 0
#define CHOLAMD_ERR_IO (-1)          /* fopen/parse failure (the reference leaves these unchecked, mnd.c:33,84,161,209) */
#define CHOLAMD_ERR_FORMAT (-2)      /* file violates the on-disk contract */
#define CHOLAMD_ERR_INVARIANT (-3)   /* ordering/cluster invariant the reference relies on is violated */
#define CHOLAMD_ERR_ARG (-4)         /* unsupported parameter combination */
#define CHOLAMD_ERR_NO_DEVICE (-5)   /* no usable HIP device: there is NO CPU fallback */
#define CHOLAMD_ERR_HIP (-6)         /* a HIP runtime call failed (see cholamd_last_error) */
#define CHOLAMD_ERR_NOMEM (-7)
#define CHOLAMD_ERR_COMM (-9)        /* an RCCL call failed (see cholamd_last_error) */
#define CHOLAMD_ERR_STALL (-8)       /* cholamd_factor_info only: a workgroup of a fused launch gave up waiting (~50 ms) for a progress
                                      * word of the same launch; the factor in the arena is NOT valid */
/* > 0: LAPACK-style info of the first failing pivot (see cholamd_factor_info) */

const char *cholamd_last_error(void);
const char *cholamd_version(void);

/* ----------------------------------------------------------------------------------------- */
/* Matrix-Market ingest: the four mmio.c entry points the reference calls                       */
/*   mm_read_banner        mmio.c:96-179   (called mmat.rg:81)                                  */
/*   mm_read_mtx_crd_size  mmio.c:189-217  (called mmat.rg:91)                                  */
/*   mm_write_banner       mmio.c:406-415  (called mmat.rg:128)                                 */
/*   mm_write_mtx_crd_size mmio.c:181-187  (called mmat.rg:129)                                 */
/* Same names, same argument meaning, same return codes (mmio.h:73-79), so libmmio.so's users    */
/* can link this library instead.  Like the reference, mm_read_banner does NOT run mm_is_valid,  */
/* so the fixtures' "real hermitian" banner is accepted.  One deliberate difference:             */
/* mm_write_mtx_crd_size returns 0 when the size line was written; the reference compares        */
/* fprintf's character count with 3 (mmio.c:181-187) and so reports MM_COULD_NOT_WRITE_FILE for   */
/* every successful write -- a return value its only caller ignores (mmat.rg:129).                */
/* ----------------------------------------------------------------------------------------- */
typedef char MM_typecode[4];
#define MM_COULD_NOT_READ_FILE 11
#define MM_PREMATURE_EOF 12
#define MM_NOT_MTX 13
#define MM_NO_HEADER 14
#define MM_UNSUPPORTED_TYPE 15
#define MM_LINE_TOO_LONG 16
#define MM_COULD_NOT_WRITE_FILE 17

int mm_read_banner(FILE *f, MM_typecode *matcode);
int mm_read_mtx_crd_size(FILE *f, int *M, int *N, int *nz);
int mm_write_banner(FILE *f, MM_typecode matcode);
int mm_write_mtx_crd_size(FILE *f, int M, int N, int nz);
char *mm_typecode_to_str(MM_typecode matcode); /* mmio.c:488-511; caller frees */

/* ----------------------------------------------------------------------------------------- */
/* Separator / cluster / matrix / vector readers: plain-array equivalents of mnd.h:28-66        */
/* (the reference versions write into Legion accessors; the file formats are the contract).     */
/* ----------------------------------------------------------------------------------------- */
typedef struct cholamd_sepinfo { int levels; int num_separators; } cholamd_sepinfo; /* SepInfo, mnd.h:23-26 */

/* read_separators (mnd.c:22-69).  idx_out[pos] = original dof at permuted position pos,
 * sep_out[pos] = 1-based separator label; both of length dim. */
int cholamd_read_separators(const char *file, int dim, int *idx_out, int *sep_out, cholamd_sepinfo *info);

/* read_clusters (mnd.c:71-150).  Emits the flat (idx, interval, sep) triples in file order, at
 * most cap of them; *count_out = number available.  Returns max_int_size (>= 0) like the
 * reference, or a negative error code. */
int cholamd_read_clusters(const char *file, int *idx_out, int *interval_out, int *sep_out, int64_t cap, int64_t *count_out);

/* read_matrix (mnd.c:152-199): nz coordinate entries, 0-based, as stored in the file (the
 * reference hashes them; here they come back as plain COO arrays). */
int cholamd_read_matrix(const char *file, int nz, int *row_out, int *col_out, double *val_out);

/* read_vector (mnd.c:201-229): skips three header lines blindly, then n values. */
int cholamd_read_vector(const char *file, int n, double *out);

/* ----------------------------------------------------------------------------------------- */
/* L-C: plan (host) = ingest + symbolic analysis of mmat.rg:1097-1209, re-expressed as plain C  */
/* ----------------------------------------------------------------------------------------- */
typedef struct cholamd_plan cholamd_plan;

/* Filled tile descriptor: fspace Filled, blas.rg:55-61 (filled == 0 means FILLED, sic). */
typedef struct cholamd_filled {
  int filled;
  int sep_x, sep_y;       /* block colour: (row separator, col separator), 1-based labels */
  int interval;           /* interval label of the snapshot */
  int cluster;            /* tile id z = row * ncols + col */
  int lo_x, lo_y, hi_x, hi_y; /* rect2d bounds, inclusive, permuted matrix coordinates */
} cholamd_filled;

/* One reference BLAS call of the level schedule, in the reference's program order. */
typedef struct cholamd_op {
  int op;                 /* 0 POTRF, 1 TRSM, 2 SYRK, 3 GEMM */
  int level;
  int m, n, k;
  int a_sx, a_sy, a_z;    /* A tile colour */
  int b_sx, b_sy, b_z;    /* B tile colour (0 if unused) */
  int c_sx, c_sy, c_z;    /* C tile colour (0 if unused) */
} cholamd_op;

int cholamd_plan_create(const char *matrix_file, const char *separator_file, const char *clusters_file, cholamd_plan **out);
/* Same, from arrays already in memory (generated problems).  perm[pos] = dof; sep_sizes by label
 * 1..nsep; clusters as (idx, interval, sep) triples like cholamd_read_clusters emits; A as COO
 * lower triangle (row >= col), 0-based original coordinates. */
int cholamd_plan_create_from_arrays(int n, int levels, const int *perm, const int *sep_sizes,
                                    const int *cl_idx, const int *cl_interval, const int *cl_sep, int64_t cl_count,
                                    int64_t nz, const int *a_row, const int *a_col, const double *a_val,
                                    const char *banner, cholamd_plan **out);
void cholamd_plan_destroy(cholamd_plan *p);

int cholamd_plan_n(const cholamd_plan *p);
int cholamd_plan_nz(const cholamd_plan *p);
int cholamd_plan_levels(const cholamd_plan *p);
int cholamd_plan_num_separators(const cholamd_plan *p);
int cholamd_plan_max_int_size(const cholamd_plan *p);
int cholamd_plan_num_blocks(const cholamd_plan *p);
int64_t cholamd_plan_arena_doubles(const cholamd_plan *p);   /* size of the panel arena */
/* Row compaction.  panel(s) stores the rows of s and of its parent in full; of every higher ancestor only the 16-row tiles (rows 16 t ..
 * 16 t + 15 of the ancestor) that a filled tile of block (ancestor, s) touches when s is eliminated -- the rest of the reference's dense
 * block instance stays zero and is never touched (blas.rg:385-395).  CHOLAMD_COMPACT=0 in the environment at plan creation stores every
 * row (the round-1/2 layout).  arena_dense_doubles: the size without compaction; block_tile_map: out[t] = stored position of tile t of
 * block (r, c) in units of 16 rows from the block's first stored row, -1 = not stored; returns the number of tiles. */
int64_t cholamd_plan_arena_dense_doubles(const cholamd_plan *p);
int cholamd_plan_block_tile_map(const cholamd_plan *p, int r, int c, int *out);
int64_t cholamd_plan_dropped_entries(const cholamd_plan *p); /* entries of A outside every allocated block */
const char *cholamd_plan_banner(const cholamd_plan *p);
void cholamd_plan_perm(const cholamd_plan *p, int *out);          /* n ints */
void cholamd_plan_sep_sizes(const cholamd_plan *p, int *out);     /* nsep ints, by label */
void cholamd_plan_sep_offsets(const cholamd_plan *p, int *out);   /* nsep ints, by label */
void cholamd_plan_tree(const cholamd_plan *p, int *out);          /* nsep ints: heap index-1 -> label (mmat.rg:834-849) */
/* per allocated block, ordered by (row label, col label): r, c, lo_x, lo_y, hi_x, hi_y, ld, and
 * the block's offset (in doubles) inside the arena split into two ints (lo32, hi32): 9 ints */
void cholamd_plan_blocks(const cholamd_plan *p, int *out);
/* snapshot `interval_lbl` of compute_filled_clusters (mmat.rg:1000-1016), filled tiles only,
 * ordered by (row label, col label, cluster) */
int64_t cholamd_plan_snapshot_count(const cholamd_plan *p, int interval_lbl);
void cholamd_plan_snapshot(const cholamd_plan *p, int interval_lbl, cholamd_filled *out);
/* the reference's BLAS call list (program order of mmat.rg:1227-1355) and its work */
int64_t cholamd_plan_num_ops(const cholamd_plan *p);
void cholamd_plan_ops(const cholamd_plan *p, cholamd_op *out);
void cholamd_plan_counts(const cholamd_plan *p, int level /* -1 = all */, int64_t calls[4], double flops[4]);
double cholamd_plan_flops(const cholamd_plan *p);           /* F_ref, SURVEY 8d */
int64_t cholamd_plan_nnz_a(const cholamd_plan *p);          /* nnz(tril A) kept */
int64_t cholamd_plan_nnz_l(const cholamd_plan *p);          /* exact symbolic nnz(L) of P A P^T (scalar elimination tree) */
int64_t cholamd_plan_nnz_tiles(const cholamd_plan *p);      /* area of the filled tiles the schedule stores (>= nnz(L)) */
double cholamd_plan_fmin(const cholamd_plan *p);            /* F_min = sum_j colcount_j^2 */
int64_t cholamd_plan_alg_bytes(const cholamd_plan *p);      /* B_alg = 8 (nnz(tril A) + nnz(L)) */

/* fill_block for every block (mmat.rg:529-633, 1216-1224): zero the arena and scatter A. */
int cholamd_plan_fill_host(const cholamd_plan *p, double *arena);
/* Multi-GPU view of the same: with `world` ranks the shared top of the tree (the tail of
 * the arena starting at *tail_offset_out) receives A's entries on rank 0 only, so that the sum over
 * ranks of the tails after the local levels equals A_top minus every contribution. */
int cholamd_plan_fill_host_part(const cholamd_plan *p, double *arena, int rank, int world, int64_t *tail_offset_out);
/* sizes of the device work lists of one tree level for (rank, world): potrf descriptors, trsm
 * strips, update tasks, update sources */
int cholamd_plan_level_work_counts(const cholamd_plan *p, int level, int rank, int world, int out[4]);
/* the volumes of cholamd_plan_level_work_volume (single GPU) under the level schedule's merging switch (option merge_targets) and macro-tile
 * threshold (mt_min_tiles, < 0: default), plus the list lengths: out[6..8] = 16x16 tasks, 64x64 macro-tile tasks, TRSM strips */
int cholamd_plan_level_work_volume_opts(const cholamd_plan *p, int level, int merge_targets, int mt_min_tiles, int64_t out[9]);
/* how full the 64 x 64 macro-tile update tasks of a level are (single GPU): out = { tasks, tasks with all 64 x 64 elements valid,
 * sum of valid elements x depth, sum of tile elements x depth } */
int cholamd_plan_level_mt_fill(const cholamd_plan *p, int level, int64_t out[4]);
/* volumes of the same lists with the top levels replicated (dist_top = 0) or distributed by column blocks (1; 2 = automatic):
 * POTRF columns, TRSM elements, update volume (target elements x source depth), broadcast entries, broadcast doubles, checksum
 * of the broadcast list (the same on every rank) */
int cholamd_plan_level_work_volume(const cholamd_plan *p, int level, int rank, int world, int dist_top, int64_t out[6]);
/* volume of the extend-add exchange of (rank, world) under dist_top 0 / 1 / 2 (auto), in arena elements: received, sent, the tail,
 * column-block pieces (0 pieces: the all-reduce of the replicated top levels) -- the host-side count behind cholamd_exchange_volume.
 * Path-aware since round 4: only the blocks of the top separators on the sender's own root path travel */
int cholamd_plan_exchange_volume(const cholamd_plan *p, int rank, int world, int dist_top, int64_t out[4]);
/* the column-block pieces of that exchange (the broadcast lists of the levels above the cut): out[i] = { arena offset, elements, owner rank, heap index
 * of the top separator }, at most `max` of them; returns their number (< 0: error).  A rank sends piece i to its owner iff it does not own it and
 * its subtree hangs under that separator: the reference touches a C tile only where the fill marks it (blas.rg:385-395), and the fill of a top
 * block comes from A (scattered on the block's owner by cholamd_device_fill) and from the subtrees under it */
int cholamd_plan_exchange_pieces(const cholamd_plan *p, int world, int dist_top, int max, int64_t (*out)[4]);
/* host-side self-check of the one-launch program cholamd_factor() runs for small problems on one GPU (chol_build_program):
 * simulated with `workers` resident workgroups and every counter raised only on job completion, no job may starve; counters
 * total up; pivot blocks and TRSM rows equal those of the per-level lists.  0 = consistent, otherwise cholamd_last_error()
 * says what is wrong (also when the problem does not qualify for the program launch).  follow: with / without followers. */
int cholamd_plan_program_check(const cholamd_plan *p, int follow, int workers);
/* the same with followers under other values of the options "follow_tail", "split_min", "split_nb" (negative: the default); the other
 * switches as the environment sets them at the time of the call */
int cholamd_plan_program_check_opts(const cholamd_plan *p, int follow_tail, int split_min, int split_nb, int workers);
int cholamd_plan_program_counts(const cholamd_plan *p, int follow, int out[6]);
/* the rounds (of one or two followed column tiles) in which a following POTRF job of `tile_columns` column tiles consumes a list of n_ext items --
 * the kernel's own grouping, a function of the list alone; and the following jobs of a plan's program: (job, items, tile columns, waits) each */
int cholamd_follow_rounds(int n_ext, int tile_columns, int cap, int *rounds_out, int *own_at_out);
int64_t cholamd_plan_program_followers(const cholamd_plan *p, int64_t cap, int *out);
int64_t cholamd_plan_program_jobs(const cholamd_plan *p, int follow, int64_t cap, int *out); /* diagnostic dump of the job queue (scripts/prog_trace.py) */ /* jobs, following POTRF jobs, update tasks, strips, counters, followed panels */
/* dense N x N col-major image of an arena (zeros outside allocated blocks) and back */
int cholamd_plan_arena_to_dense(const cholamd_plan *p, const double *arena, double *dense);
/* write_matrix (mmat.rg:102-147): banner, "M N nnz", "row col %0.8g" per non-zero, block by
 * block; full_precision != 0 writes %.17g instead (needed for the 1e-10 gate). */
int cholamd_plan_write_matrix(const cholamd_plan *p, const double *arena, const char *file, int full_precision);
/* write_solution (mmat.rg:785-798): n lines "%0.8g" (or %.17g), original dof order, no header */
int cholamd_write_solution(const char *file, const double *x, int n, int full_precision);
/* the reference's -d structured log lines: Block lines + the POTRF / TRSM / GEMM lines of the whole op list (blas.rg:308,340,405),
 * from the symbolic phase alone (no device) */
int cholamd_plan_write_debug_log(const cholamd_plan *p, FILE *f);
/* what the reference's symbolic phase prints with -d: Block (mmat.rg:331), per level Cluster (mmat.rg:396,432) and Fill
 * (mmat.rg:1010) lines; the op lines are printed by the fused tasks as they run (cholamd_factor_debug) */
int cholamd_plan_write_debug_header(const cholamd_plan *p, FILE *f);
/* write_blocks' text dump (mmat.rg:183-217): `header`, then every allocated block with its values as "%0.2f, " */
int cholamd_plan_write_blocks_txt(const cholamd_plan *p, const double *arena, const char *file, const char *header);
int cholamd_plan_ntiles_at_level(const cholamd_plan *p, int sep, int level);

/* ----------------------------------------------------------------------------------------- */
/* Problem generator (absent from the reference, whose orderings come from an      */
/* external tool): d-dimensional 5-/7-point Laplacian on an nx x ny x nz grid (nz = 1 for 2-D),   */
/* natural index x + nx*y + nx*ny*z, diagonal 2*dim, off-diagonal -1, lower triangle -- exactly    */
/* the fixtures' matrices -- with a geometric nested-dissection ordering of `levels` tree levels  */
/* (split the longest side at its middle plane) and cluster lists that satisfy the invariants of  */
/* SURVEY A.3 (tiles of about `tile` dofs at interval 0, halved per interval, ONE tile at the      */
/* interval in force when a separator is eliminated).  Outputs are in the reference's on-disk      */
/* formats (Appendix A) so that the reference, the oracle and this library read the same files.    */
/* ----------------------------------------------------------------------------------------- */
typedef struct cholamd_problem cholamd_problem;
int cholamd_generate_laplacian(int nx, int ny, int nz, int levels, int tile, cholamd_problem **out);
void cholamd_problem_destroy(cholamd_problem *g);
int cholamd_problem_n(const cholamd_problem *g);
int cholamd_problem_nz(const cholamd_problem *g);
/* writes <prefix>.mtx, <prefix>_ord_<levels>.txt, <prefix>_clust_<levels>.txt and B_<n>x1.mtx-style
 * rhs (b_i = 1 + (7919 i mod 10)) as <prefix>_B.mtx */
int cholamd_problem_write(const cholamd_problem *g, const char *prefix);
/* plan straight from memory (no files) */
int cholamd_plan_create_from_problem(const cholamd_problem *g, cholamd_plan **out);
void cholamd_problem_rhs(const cholamd_problem *g, double *b);

/* ----------------------------------------------------------------------------------------- */
/* L-C: device side.  The arena (all per-separator panels, col-major, contiguous) lives in HBM. */
/* ----------------------------------------------------------------------------------------- */
typedef struct cholamd_device cholamd_device;

int cholamd_device_count(void);
/* Uploads the level schedule (op descriptors) of `plan` to device `device_id`. */
int cholamd_device_create(const cholamd_plan *plan, int device_id, cholamd_device **out);
void cholamd_device_destroy(cholamd_device *d);
/* device memory owned by the library (hipMalloc); callers may instead pass their own buffers
 * (e.g. torch tensors) of cholamd_plan_arena_doubles() doubles to the calls below */
int cholamd_device_alloc(cholamd_device *d, int64_t doubles, double **dptr);
int cholamd_device_free(cholamd_device *d, double *dptr);
int cholamd_device_upload(cholamd_device *d, double *d_dst, const double *h_src, int64_t doubles, void *stream);
int cholamd_device_download(cholamd_device *d, double *h_dst, const double *d_src, int64_t doubles, void *stream);
int cholamd_device_sync(cholamd_device *d, void *stream);
/* A scatter on the device (fill_block, mmat.rg:1216-1224): zero d_arena, scatter tril(A).  A partitioned device (cholamd_device_set_partition)
 * scatters every entry under the cut and, of the shared top of the tree, the entries exactly one rank must start from: all of them on rank 0 when the
 * top levels are replicated, those of the column blocks the rank OWNS when they are distributed (option dist_top) -- fill after set_partition /
 * set_option, with the schedule the factorisation will use. */
int cholamd_device_fill(cholamd_device *d, double *d_arena, void *stream);
/* The hot path: the whole level loop of mmat.rg:1227-1355 on d_arena, asynchronously on stream.  Small problems (every
 * pivot block <= 192 columns, no macro-tile phase: the reference's fixtures) run as ONE launch of resident workgroups that
 * draw POTRF / TRSM / update jobs from a queue and hand data to each other through counters (option "program"); otherwise,
 * per tree level and column-block step of its pivots: one fused POTRF+TRSM launch and the update launch(es).  level_lo/level_hi
 * restrict the loop to tree levels [level_lo, level_hi] (inclusive; pass 0, levels-1 for everything) -- used by
 * the multi-GPU driver to run subtree levels and top levels separately.
 * A device object carries one workspace (diagonal-block inverses, progress words of the fused launches, info):
 * it serves one factorisation or solve at a time; use one object per concurrent stream of work.  The arena is
 * the caller's: any number of arenas can be factored one after the other with the same object.
 * Co-residency of the one-launch form: its workgroups wait for each other inside the launch.  The job list is checked when the
 * device object is built (simulated with four resident workgroups, counters raised on job completion: a list that cannot make
 * progress that way is not used, the per-level launches are); on a GPU shared with other kernels a job may wait for a producer
 * that is not resident yet -- every in-launch wait gives up after about two seconds of polling and the factorisation then FAILS
 * through info = CHOLAMD_ERR_STALL (cholamd_factor_info), it never hangs.  A stalled factorisation has written part of the arena:
 * refill (cholamd_device_fill) before factoring again; option "program" = 0 selects the per-level launches, which need no
 * co-residency beyond one launch. */
int cholamd_factor(cholamd_device *d, double *d_arena, void *stream);
int cholamd_factor_levels(cholamd_device *d, double *d_arena, int level_hi, int level_lo, void *stream);
/* Restrict the schedule to the subtrees owned by `rank` of `world` (a power of two <= 2^(levels-1)):
 * separators below the split level that are not in this rank's subtrees are skipped.  world == 1
 * restores the full schedule. */
int cholamd_device_set_partition(cholamd_device *d, int rank, int world);
/* diagnostic: one program launch with per-job clock stamps.  out (cap >= 5 * jobs int64) receives per job, in queue order:
 * kind (0 POTRF, 10 POTRF that follows, 1 TRSM group, 2 update group), then the 100 MHz real-time clock (10 ns ticks since the
 * first job was drawn) when the job was drawn, when its waits were over and when it ended, and the workgroup that ran it.
 * *njobs_out = number of jobs (call with cap = 0 to size the buffer). */
int cholamd_device_program_trace(cholamd_device *d, double *d_arena, void *stream, int64_t cap, int64_t *out, int *njobs_out);
/* Debug mode of the reference (mmat.rg -d <dir>): the level loop of mmat.rg:1227-1355 LITERALLY -- one fused task
 * at a time through the task-level entry points below, serialised -- printing the tasks' POTRF / TRSM / GEMM lines on stdout
 * and, after every task, dumping the whole matrix as write_blocks does (mmat.rg:174-218): <dir>/<gen_filename>.mtx
 * (write_matrix format) and .txt (block dump), with gen_filename's names (mmat.rg:149-172: potrf_lvlL_aXY,
 * trsm_lvlL_aXY_bXY, gemm_lvlL_aXY_bXY_cXY, the block colours printed with %d%d as the reference does) and its
 * "filename: ..." lines.  Slow by design; the factor it leaves in d_arena equals cholamd_factor's to rounding. */
int cholamd_factor_debug(cholamd_device *d, double *d_arena, const char *dir, int full_precision, void *stream);
/* LAPACK-style info after the stream has been synchronised: 0 ok; k > 0 = leading minor k of the
 * pivot of separator *sep_out is not positive definite (the reference ignores this, blas.rg:71);
 * < 0 = the factorisation itself failed (CHOLAMD_ERR_STALL, or a HIP error code of this call) and
 * cholamd_last_error() says why -- callers must not use the factor in either non-zero case. */
int cholamd_factor_info(cholamd_device *d, int *sep_out);
/* Schedule / kernel-selection switches of this device object.  They default to the values the environment gave when
 * the object was created (CHOLAMD_SPLIT_MIN, CHOLAMD_SPLIT_NB, CHOLAMD_NO_FUSE, CHOLAMD_FUSE_UPDATE_MAX,
 * CHOLAMD_MT_MIN_TILES, CHOLAMD_NO_CELLS, CHOLAMD_SOLVE_REFERENCE_SHAPE: read once, there); names: "split_min",
 * "split_nb", "fuse", "fuse_update_max", "mt_min_tiles", "cells", "solve_reference_shape", "program" (the whole factorisation
 * of a small problem as one launch; CHOLAMD_NO_PROGRAM), "follow" (its pivot blocks follow their children's TRSM strips;
 * CHOLAMD_NO_FOLLOW), "follow_tail" (followers of more than four tile columns take the last follow_tail column tiles of each source
 * themselves, update jobs bring the rest; 0 = they take everything; CHOLAMD_FOLLOW_TAIL; "follow_tail_split": the same for the next
 * column block of a split pivot), "staged" (the extend-add jobs of the
 * program launch take their sources pivot block by pivot block as those are solved instead of waiting for all of them;
 * CHOLAMD_NO_STAGED), "fine_upd" (followed strips wait for the update jobs into their own rows' block only; CHOLAMD_NO_FINE_UPD),
 * (the program launch factors pivots up to 176 columns whole while "split_min" / "split_nb" are at their defaults: CHOL_PROG_SPLIT_MIN),
 * "trsm_wt_min" (level schedule: a column-block step with at least this many TRSM strips launches its POTRFs on their own and solves
 * the strips with the throughput kernel, one wave per strip; 0 = never; CHOLAMD_TRSM_WT_MIN),
 * "skyline" (program launch: the tile-level skyline of the leaf pivots -- the envelope of A inside the block -- is used: tile updates and
 * panel tiles left of it are skipped, banded leaves up to 272 columns are factored as one block; CHOLAMD_NO_SKYLINE), "stage_chunk" (such a
 * leaf's columns reach the extend-add jobs in chunks of this many column tiles; 0 = the whole block at once, the default;
 * CHOLAMD_STAGE_CHUNK),
 * "leaf_envelope" (level schedule: a leaf's factor stays inside the envelope of A, so the strips, trailing updates and extend-add sources of the leaves
 * leave out what is structurally zero -- identical factors, 16-35 % less time on the generated grids; CHOLAMD_NO_LEAF_ENVELOPE),
 * "super_blocks", "dist_top" (0 / 1 / 2 = automatic: top levels of a partitioned run distributed by column
 * blocks, see Multi-GPU below; CHOLAMD_DIST_TOP).  Rebuilds the work lists. */
int cholamd_device_set_option(cholamd_device *d, const char *name, int value);
/* number of broadcast phases in the partitioned schedule (> 0: the top levels are distributed by column blocks, option "dist_top",
 * and cholamd_factor_levels over them needs a communicator: use cholamd_factor_sharded / cholamd_factor_multi) */
int cholamd_device_bcast_phases(const cholamd_device *d);
/* Solve phase, mmat.rg:1364-1495: b and x in ORIGINAL dof order (device pointers, n doubles).
 * The off-diagonal blocks accumulate into the vector with hardware fp64 atomics, so x agrees from run to run to
 * rounding (~1e-16 relative), NOT bit for bit; option "solve_reference_shape" selects the deterministic per-call
 * kernels of the BLAS-level entry points instead (slow beyond ~10^5 unknowns).
 * What the streamed solve does NOT read: the structural zeros of the leaf panels (cholamd_plan_solve_skips; CHOLAMD_SOLVE_NO_BAND=1 in the environment
 * when the device object builds its solve lists reads everything).  The levels of at most 128 separators wider than 256 columns are solved with EXPLICIT
 * inverses of their 256-column diagonal spans (formed from the factor at the start of a solve -- of a refinement: its corrections reuse them -- in
 * 512 KiB of device memory per span: 0.4 GB for a 100^3 grid; CHOLAMD_SOLVE_NO_INV256=1 keeps the substitution form).  Their span steps hand data
 * between workgroups INSIDE a launch behind flags; the waits are bounded like the program launch's, and one that gives up poisons its part of the
 * vector with NaN (a residual or refinement then reports it) instead of hanging the device. */
int cholamd_solve(cholamd_device *d, const double *d_arena, const double *d_b, double *d_x, void *stream);
/* ---- mixed precision (BASELINE config 5; not in the reference, whose arithmetic is fp64 CBLAS throughout): fp32 factor +
 * fp64 iterative refinement.  The fp32 arena has the fp64 arena's element layout: cholamd_plan_arena_doubles() FLOATS.
 * The fp32 schedule factors pivots in blocks of at most 128 columns out of LDS (v_mfma_f32 kernels, one launch per phase);
 * info as for the fp64 path (cholamd_factor_info).  cholamd_solve_f32 is one solve with the fp32 factor (vectors and
 * arithmetic fp64, L converted on load); cholamd_solve_refine iterates x += (L32 L32^T)^-1 (b - A x) with the residual
 * in fp64 against the matrix file's A until ||b - A x|| / ||b|| <= tol or max_iter corrections have been applied, and
 * returns the corrections used and the final relative residual.  cholamd_residual computes that residual for any x
 * (d_r may be NULL); both synchronise the stream. */
int cholamd_device_fill_f32(cholamd_device *d, float *d_arena32, void *stream);
int cholamd_factor_f32(cholamd_device *d, float *d_arena32, void *stream);
int cholamd_factor_levels_f32(cholamd_device *d, float *d_arena32, int level_hi, int level_lo, void *stream);
int cholamd_solve_f32(cholamd_device *d, const float *d_arena32, const double *d_b, double *d_x, void *stream);
int cholamd_solve_refine(cholamd_device *d, const float *d_arena32, const double *d_b, double *d_x, int max_iter, double tol,
                         int *iters_out, double *relres_out, void *stream);
int cholamd_residual(cholamd_device *d, const double *d_b, const double *d_x, double *d_r, double *relres_out, void *stream);
/* average device time (ms) of the three kernel families of the last cholamd_factor call measured
 * with HIP events on its stream; valid after cholamd_device_sync.  Enable with set_timing(1). */
int cholamd_device_set_timing(cholamd_device *d, int on);
int cholamd_device_get_timing(cholamd_device *d, float ms_by_kind[4], int launches_by_kind[4]);
/* the same with the kinds of a sharded run: [0..3] as above (POTRF (+ fused TRSM), TRSM, update, program launch), [4] the RCCL
 * extend-add exchange, [5] the grouped ncclBroadcasts of the distributed top levels, [6..7] unused */
int cholamd_device_get_timing_ex(cholamd_device *d, float ms_by_kind[8], int launches_by_kind[8]);
/* mean elapsed time (ms) of an event pair with nothing between them on `stream`: the part of every timed launch
 * above that is the event commands, not the kernel */
int cholamd_device_event_overhead(cholamd_device *d, void *stream, float *ms_out);

/* ----------------------------------------------------------------------------------------- */
/* Multi-GPU (not in the reference, whose Legion runtime would move instances          */
/* implicitly): one rank per GPU, the separator tree cut at level log2(world).  Rank g factors    */
/* the subtrees under its level-d separator (cholamd_device_set_partition), its contributions to  */
/* the shared top of the tree accumulate in its own copy of the arena TAIL (the top panels are    */
/* contiguous: offset cholamd_device_tail_offset), ONE RCCL all-reduce (sum, fp64) over that tail  */
/* is the extend-add exchange, then the top levels: REPLICATED on every rank (small roots: they   */
/* are a latency chain) or, option "dist_top" (automatic from a root separator of 1024 columns),  */
/* DISTRIBUTED by column blocks: block b of a top separator belongs to rank (b + heap index) mod  */
/* world; its owner factors it and solves the rows below it, ncclBroadcast carries the block to    */
/* every rank, every rank applies the trailing update and the extend-add to the column blocks IT    */
/* owns (one owner per element, sources in program order: deterministic).  libcholamd links RCCL;   */
/* the communicator is an ncclComm_t made here or adopted from the caller.                         */
/* ----------------------------------------------------------------------------------------- */
typedef struct cholamd_comm cholamd_comm;
#define CHOLAMD_UNIQUE_ID_BYTES 128
int cholamd_comm_unique_id(char id[CHOLAMD_UNIQUE_ID_BYTES]);   /* ncclGetUniqueId: rank 0 calls it and hands the bytes to every rank */
int cholamd_comm_create(cholamd_device *d, int world, int rank, const char id[CHOLAMD_UNIQUE_ID_BYTES], cholamd_comm **out); /* ncclCommInitRank on d's GPU (collective) */
int cholamd_comm_create_all(cholamd_device *const *devs, int n, cholamd_comm **out /* n handles */); /* one process, n GPUs: ncclCommInitAll */
/* one process, n rank objects, NO RCCL: device-side ordered sum of the tails + peer copies, events between the ranks' streams.
 * The ranks may share a device (the one-GPU test box runs world 2 ... 8 this way).  For cholamd_factor_multi only. */
int cholamd_comm_create_local(cholamd_device *const *devs, int n, cholamd_comm **out /* n handles */);
int cholamd_comm_adopt(void *nccl_comm /* ncclComm_t */, int world, int rank, cholamd_comm **out);  /* the caller keeps ownership of the ncclComm_t */
int cholamd_comm_count(const cholamd_comm *c, int *ranks_out); /* ncclCommCount: the ranks the communicator really joins */
void cholamd_comm_destroy(cholamd_comm *c);
int cholamd_comm_allreduce(cholamd_comm *c, double *d_buf, int64_t count, void *stream); /* in-place fp64 sum (ncclAllReduce), asynchronous on `stream` */
/* Per-rank arenas (multi-GPU).  An arena of cholamd_plan_arena_doubles() elements of elem_bytes (8: fp64, 4: the fp32 factor) for this rank: its
 * address range is complete, so every entry point takes it like a plain allocation, but only the rank's own panels (the subtrees cholamd_device_set_partition
 * gave it) and the shared top of the tree are backed by memory of their own (hipMemAddressReserve / hipMemCreate / hipMemMap); the ranges of the other
 * ranks' panels -- never read on this rank -- all alias one 64 MB scratch chunk.  Rank 0 (it gathers the factor and solves) and single-GPU devices get a
 * plain hipMalloc.  Call after cholamd_device_set_partition; *backed_bytes = device memory the arena takes; free with cholamd_device_free_arena (any
 * arena of cholamd_device_alloc / _alloc_arena). */
int cholamd_device_alloc_arena(cholamd_device *d, int elem_bytes, void **dptr, int64_t *backed_bytes);
int cholamd_device_free_arena(cholamd_device *d, void *dptr);
int64_t cholamd_device_tail_offset(const cholamd_device *d);    /* first double of the shared top of the tree in the arena (arena size if world == 1) */
/* the extend-add exchange alone, asynchronous on `stream`.  Replicated top levels: in-place ncclAllReduce(sum) of d_arena[tail .. arena).
 * Top levels distributed by column blocks (option "dist_top"): OWNER-DIRECTED -- a rank works only on the column blocks it owns after the
 * exchange, so every rank sends its partial copy of each block it does not own straight to the owner (grouped ncclSend / ncclRecv:
 * point-to-point over all xGMI links at once) and the owner adds the world - 1 copies to its own in rank order (deterministic).  Each
 * rank receives (world - 1) x its owned blocks and sends the others once: (world - 1) / world of the tail each way, half a ring
 * all-reduce's volume; blocks a rank does not own hold partial sums afterwards and are overwritten by the owners' broadcasts. */
int cholamd_exchange_tail(cholamd_device *d, double *d_arena, cholamd_comm *c, void *stream);
/* elements this rank receives / sends in that exchange, elements of the tail, column-block pieces (0 pieces: the all-reduce) */
int cholamd_exchange_volume(const cholamd_device *d, int64_t out[4]);
/* one rank's part of a sharded factorisation: local levels, exchange, top levels; asynchronous on `stream`.
 * The arena must have been filled by cholamd_device_fill AFTER cholamd_device_set_partition (rank-aware fill). */
int cholamd_factor_sharded(cholamd_device *d, double *d_arena, cholamd_comm *c, void *stream);
/* the same with the fp32 factor (mixed precision x multi-GPU, BASELINE config 5): fp32 arena (cholamd_device_fill_f32 after
 * set_partition), the fp32 schedule partitioned like the fp64 one, exchange and broadcasts on floats */
int cholamd_factor_sharded_f32(cholamd_device *d, float *d_arena32, cholamd_comm *c, void *stream);
/* the panels of the subtrees this rank owns travel to rank 0 (grouped ncclSend / ncclRecv), whose arena then holds the complete
 * factor -- for cholamd_solve / cholamd_solve_refine on rank 0 and the writers.  elem_bytes: 8 (fp64 arena) or 4 (fp32 arena) */
int cholamd_gather_to_root(cholamd_device *d, void *d_arena, int elem_bytes, cholamd_comm *c, void *stream);
/* the same for one process driving n GPUs (devs[g] partitioned as rank g of n): the n all-reduces form one RCCL group */
int cholamd_factor_multi(cholamd_device *const *devs, double *const *arenas, cholamd_comm *const *comms, int n, void *const *streams /* or NULL */);
/* after cholamd_factor_multi: peer-copies the panels of the subtrees owned by ranks 1..n-1 into arenas[0], which then
 * holds the complete factor (for cholamd_solve and the writers) */
int cholamd_gather_factor(cholamd_device *const *devs, double *const *arenas, int n, void *const *streams /* or NULL */);
/* both for the fp32 factor */
int cholamd_factor_multi_f32(cholamd_device *const *devs, float *const *arenas32, cholamd_comm *const *comms, int n, void *const *streams /* or NULL */);
int cholamd_gather_factor_f32(cholamd_device *const *devs, float *const *arenas32, int n, void *const *streams /* or NULL */);
/* Distributed solve (mmat.rg:1394-1479 sharded like the factorisation): every rank sweeps the separators of its own subtrees, the shared top of the
 * tree -- which every rank holds completely after cholamd_factor_sharded / _multi -- is solved redundantly, and only VECTORS travel: one sum over the
 * ranks of the top's part of the right-hand side after the forward sweep under the cut, one sum of the solution at the end (RCCL all-reduces;
 * device-side ordered sums + peer copies over a local communicator).  Every rank passes the same b and ends with the same x.  _f32: an fp32 factor
 * (solution to fp32-factor accuracy); cholamd_solve_refine_sharded: the fp64 iterative refinement of cholamd_solve_refine, every rank running the same
 * loop on identical iterates (residual against A on every rank, corrections solved as above).  No cholamd_gather_to_root is needed for either. */
/* host-side view of rank's share of the solve lists of one tree level: out = { separators, (ancestor, separator) row runs, forward row chunks, backward row
 * chunks, columns solved }; below the cut the ranks' shares tile the undivided lists, above it every rank holds them whole */
int cholamd_plan_solve_counts(const cholamd_plan *p, int level, int rank, int world, int64_t out[5]);
/* what the solve does not read of a level's panels (the structural zeros of the LEAVES: nothing reaches a leaf from below, so its factor stays inside
 * the envelope of A).  seps: 3 ints per separator of the level's list -- first position in the permuted vector, columns, band (L(i, j) = 0 inside the
 * diagonal block for i - j > band; 0: read whole); runs: 5 ints per (ancestor, separator) row run -- first position of the run's rows in the permuted
 * vector, rows, first position of the separator's columns, columns, c_lo (the run's rows are zero in the separator's columns [0, c_lo)).  Counts as in
 * cholamd_plan_solve_counts(p, level, 0, 1, .): out[0] separators, out[1] runs. */
int cholamd_plan_solve_skips(const cholamd_plan *p, int level, int *seps, int *runs);
int cholamd_solve_sharded(cholamd_device *d, const double *d_arena, const double *d_b, double *d_x, cholamd_comm *c, void *stream);
int cholamd_solve_sharded_f32(cholamd_device *d, const float *d_arena32, const double *d_b, double *d_x, cholamd_comm *c, void *stream);
int cholamd_solve_refine_sharded(cholamd_device *d, const float *d_arena32, const double *d_b, double *d_x, int max_iter, double tol,
                                 int *iters_out, double *relres_out, cholamd_comm *c, void *stream);
/* the same driven by one process for n rank objects (communicators of cholamd_comm_create_all or cholamd_comm_create_local); bs[g], xs[g]: rank g's
 * device copies of b and x */
int cholamd_solve_multi(cholamd_device *const *devs, const double *const *arenas, const double *const *bs, double *const *xs, cholamd_comm *const *comms,
                        int n, void *const *streams);
int cholamd_solve_refine_multi(cholamd_device *const *devs, const float *const *arenas32, const double *const *bs, double *const *xs, int max_iter, double tol,
                               int *iters_out, double *relres_out, cholamd_comm *const *comms, int n, void *const *streams);

/* ----------------------------------------------------------------------------------------- */
/* L-B: task level -- the four fused leaf tasks of blas.rg.  A region is a block instance:     */
/* device pointer of element (lo_x, lo_y), leading dimension, and its bounds in permuted         */
/* matrix coordinates (what get_raw_ptr_2d, blas.rg:35-43, resolves from Legion).                */
/* filled lists are HOST arrays, exactly the `Filled` records the reference tasks iterate.       */
/* ----------------------------------------------------------------------------------------- */
typedef struct cholamd_region {
  double *ptr;
  int ld;
  int lo_x, lo_y, hi_x, hi_y;
  const int *tile_row; /* NULL: every row lo_x..hi_x is stored, row x at ptr + (x - lo_x).  Else (row-compacted block instance of the arena):
                        * 16-row tile t of the block (rows lo_x + 16 t ..) is stored at rows 16 * tile_row[t] .. from ptr, or not at all (-1: no
                        * filled tile touches it -- the reference never reads or writes those rows, blas.rg:385-395) */
} cholamd_region;
/* the block instance (r, c) of an arena laid out by `plan` (tile_row points into the plan) */
int cholamd_plan_region(const cholamd_plan *plan, double *d_arena, int r, int c, cholamd_region *out);

/* fused_dpotrf, blas.rg:292-315.  Returns LAPACK info of the first failing tile (the reference discards it). */
int cholamd_fused_dpotrf(const cholamd_region *rA, const cholamd_filled *filled_rA, int nA, int level, int interval, int debug, void *stream);
/* fused_dtrsm, blas.rg:317-351 */
int cholamd_fused_dtrsm(const cholamd_region *rA, const cholamd_region *rB, const cholamd_filled *filled_rA, int nA,
                        const cholamd_filled *filled_rB, int nB, int level, int interval, int debug, void *stream);
/* fused_dsyrk, blas.rg:353-436 */
int cholamd_fused_dsyrk(const cholamd_region *rA, const cholamd_region *rB, const cholamd_region *rC,
                        const cholamd_filled *filled_rA, int nA, const cholamd_filled *filled_rB, int nB,
                        const cholamd_filled *filled_rC, int nC, int col_cluster_size, int level, int interval, int debug, void *stream);
/* fused_dgemm, blas.rg:438-504 */
int cholamd_fused_dgemm(const cholamd_region *rA, const cholamd_region *rB, const cholamd_region *rC,
                        const cholamd_filled *filled_rA, int nA, const cholamd_filled *filled_rB, int nB,
                        const cholamd_filled *filled_rC, int nC, int col_cluster_size, int level, int interval, int debug, void *stream);

/* ----------------------------------------------------------------------------------------- */
/* L-A: BLAS level -- the C symbols Terra binds (blas.rg:71, 99, 139, 187, 226, 263;           */
/* mmat.rg:1057).  Enum values are CBLAS's.  Only the parameter combinations the reference      */
/* issues are implemented; anything else sets cholamd_last_error and returns/raises ERR_ARG.    */
/* `_dev` variants: device pointers, asynchronous on `stream`.                                   */
/* ----------------------------------------------------------------------------------------- */
enum { CholamdColMajor = 102, CholamdNoTrans = 111, CholamdTrans = 112, CholamdUpper = 121, CholamdLower = 122,
       CholamdNonUnit = 131, CholamdLeft = 141, CholamdRight = 142 };

int cholamd_LAPACKE_dpotrf(int matrix_layout, char uplo, int n, double *a, int lda);                      /* blas.rg:71 */
void cholamd_cblas_dtrsm(int layout, int side, int uplo, int transa, int diag, int m, int n, double alpha,
                         const double *a, int lda, double *b, int ldb);                                   /* blas.rg:99 */
void cholamd_cblas_dgemm(int layout, int transa, int transb, int m, int n, int k, double alpha, const double *a, int lda,
                         const double *b, int ldb, double beta, double *c, int ldc);                      /* blas.rg:139 */
void cholamd_cblas_dsyrk(int layout, int uplo, int trans, int n, int k, double alpha, const double *a, int lda,
                         double beta, double *c, int ldc);                                                /* blas.rg:187 */
void cholamd_cblas_dtrsv(int layout, int uplo, int transa, int diag, int n, const double *a, int lda, double *x, int incx); /* blas.rg:226 */
void cholamd_cblas_dgemv(int layout, int trans, int m, int n, double alpha, const double *a, int lda, const double *x, int incx,
                         double beta, double *y, int incy);                                               /* blas.rg:263 */
void cholamd_openblas_set_num_threads(int num_threads);                                                  /* mmat.rg:1057: accepted, no effect */
int cholamd_blas_status(void); /* 0, or the error code of the last L-A call on this thread */

int cholamd_dpotrf_dev(int n, double *d_a, int lda, int *d_info, void *stream);
int cholamd_dtrsm_dev(int m, int n, const double *d_a, int lda, double *d_b, int ldb, void *stream);
int cholamd_dgemm_dev(int m, int n, int k, const double *d_a, int lda, const double *d_b, int ldb, double *d_c, int ldc, void *stream);
int cholamd_dsyrk_dev(int n, int k, const double *d_a, int lda, double *d_c, int ldc, void *stream);
int cholamd_dtrsv_dev(int trans, int n, const double *d_a, int lda, double *d_x, void *stream);
int cholamd_dgemv_dev(int trans, int m, int n, const double *d_a, int lda, const double *d_x, double *d_y, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CHOLAMD_H */
