/* Internal host-side data model of libcholamd (not installed; the public ABI is include/cholamd.h).
 *
 * Storage layout (the MI355X-first replacement of the mapper's per-block instances,
 * cholesky.cc:65-73): ONE contiguous fp64 arena holding one column-major PANEL per separator s.
 * Panel(s) has the separator's own columns (n_s of them) and the rows of s followed by the rows
 * of every ancestor of s in increasing permuted position (parent, grand-parent, ..., root), so
 * block (anc, s) of the reference is a row slice of panel(s) with the panel's leading dimension.
 * ROW COMPACTION: the slice of the separator itself and of its parent hold every row; the slice of a
 * higher ancestor holds only the 16-row tiles (rows 16 t .. 16 t + 15 of the ancestor) that a filled tile
 * of the block touches at the time the separator is eliminated -- the others stay zero in the reference's
 * dense block and are never read or written (blas.rg:385-395), here they have no storage
 * (chol_block.tmap / chol_block_row; lapl_3375: 13.9 -> 8.9 MB, gen 60^3: 15.2 -> 5.1 GB).
 * Panels are laid out by ascending label, hence the panels of the top
 * of the tree (root last) form a contiguous tail of the arena -- the extend-add exchange buffer of
 * the multi-GPU driver.
 */
#ifndef CHOL_PLAN_H
#define CHOL_PLAN_H

#include <stdint.h>
#include "cholamd.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CHOL_NB 16        /* diagonal-block width of the POTRF/TRSM kernels = one fp64 MFMA tile */
#define CHOL_RR_MAXN 272  /* largest pivot the register-resident kernels take (17 tiles) */
#ifndef CHOL_FOLLOW_ALL_MAXT
#define CHOL_FOLLOW_ALL_MAXT 4
#endif
// /* followers of at most this many column tiles take every column of their sources themselves */
#define CHOL_FOLLOW_TAIL 3     /* default of option follow_tail */
#define CHOL_FOLLOW_MAXT 10 /* most column tiles of a pivot block that follows its sources inside the program launch (chol_kernels.hip, follow_external) */
#define CHOL32_TRSM_GROUP 16 /* strips per workgroup of the fp32 throughput TRSM (k32_trsm_wt): the fp32 schedule pads every pivot block's strips to it */
#define CHOL32_MAXN 128   /* widest pivot block of the fp32 path: its lower triangle is factored out of LDS (chol_kernels_f32.hip) */

typedef struct {
  int n_int;    /* number of intervals */
  int *len;     /* boundary-list length per interval */
  int **raw;    /* boundary lists as in the file (interval t>0 indexes interval t-1) */
  int **start;  /* resolved to dof offsets inside the separator: start[t][i], i < len[t] */
} chol_clusters;

typedef struct {
  int r, c;                       /* labels */
  int lo_x, lo_y, hi_x, hi_y;     /* permuted coords, inclusive */
  int rows, cols, ld;
  int64_t off;                    /* arena offset (doubles) of the block's first stored row in column lo_y */
  int *tmap;                      /* row compaction: tmap[t] = position of the block's 16-row tile t among the tiles the panel stores, -1 = not
                                   * stored (structurally zero); NULL: every row is stored (diagonal and parent blocks, CHOLAMD_COMPACT=0) */
  int crows;                      /* rows the panel stores for this block */
} chol_block;
/* arena offset of row `row` (relative to the block) in the block's first column; rows of one filled tile, or of adjacent filled tiles, are
 * consecutive.  -1: the row has no storage */
static inline int64_t chol_block_row(const chol_block *B, int row)
{
  if (!B->tmap) return B->off + row;
  const int t = B->tmap[row / CHOL_NB];
  return t < 0 ? -1 : B->off + (int64_t)t * CHOL_NB + row % CHOL_NB;
}

/* device-side work descriptors (shared with the kernels; plain ints / int64 offsets in doubles
 * relative to a base pointer passed at launch) */
typedef struct {
  int64_t a_off;      /* diagonal block */
  int64_t dinv_off;   /* workspace offset of this separator's inverted diagonal NBxNB blocks */
  int n, lda;
  int sep;            /* label (for info reporting) */
  int col0;           /* first column of this diagonal block inside its pivot (blocked big pivots) */
  int ctr;            /* program launch: the block's progress counter (columns published) */
  int tab;            /* > 0: byte offset, from this descriptor's own address, of the block's role table (chol_potrf_table: which
                       * tile lives in which wave's register slot, the per-step work masks) -- built with the schedule instead of by every workgroup's
                       * prologue; 0: the kernel builds it */
  unsigned char sky[24]; /* tile-level skyline of the block (leaf pivots; all zero otherwise): tile (i, j) of the block is structurally zero for
                          * j < sky[i] -- its trailing updates are skipped (chol_kernels.hip, potrf_rr_body) */
} chol_potrf_desc;

/* ---- the POTRF role's deal of tiles to waves and its per-step work masks (potrf_rr_body in chol_kernels.hip), shared by the kernel and
 * by the host, which builds them with the schedule (chol_potrf_table) */
#define CHOL_RR_MAXT 17   /* tile columns of the widest register-resident pivot block */
#define CHOL_RR_NW 11     /* tile waves */
#define CHOL_RR_NHEAVY 9  /* tile waves that do not share a SIMD with the factor wave */
#define CHOL_RR_SLOTS 12  /* most tiles a tile wave owns */
#ifndef CHOL_RR_HEAVY_ONLY
#define CHOL_RR_HEAVY_ONLY 36 /* pivot blocks with at most this many register tiles (T <= 10) leave the factor wave's SIMD to the factor wave */
#endif
#define CHOL_RR_M_SOLVE0 12 /* step mask: bit s < 12 = slot s; bits 12, 13 = the wave's first / second panel tile of the step is not structurally zero */
#define CHOL_RR_M_SOLVE1 13
#define CHOL_RR_KM_ZERO 0x80 /* km: the tile is left of the block's skyline (zero in A: the prologue does not load it); the low bits: first step with a non-zero update */
#define CHOL_RR_TAB_MASK 0                                              /* int [CHOL_RR_MAXT][CHOL_RR_NW] */
#define CHOL_RR_TAB_IJ (CHOL_RR_MAXT * CHOL_RR_NW * 4)                   /* unsigned short [CHOL_RR_SLOTS][CHOL_RR_NW]: i | j << 8, 0xffff = empty */
#define CHOL_RR_TAB_KM (CHOL_RR_TAB_IJ + CHOL_RR_SLOTS * CHOL_RR_NW * 2) /* unsigned char [CHOL_RR_SLOTS][CHOL_RR_NW] */
#define CHOL_RR_TAB_BYTES ((CHOL_RR_TAB_KM + CHOL_RR_SLOTS * CHOL_RR_NW + 15) / 16 * 16)
#ifdef __HIPCC__
#define CHOL_HD __host__ __device__ __forceinline__
#else
#define CHOL_HD static inline
#endif
/* tile index -> (i, j) of the column-major enumeration of the lower triangle of a T x T tile grid */
CHOL_HD void chol_rr_tile_of_index(int idx, int T, int *ti, int *tj)
{
  int j = 0;
  while (j < T && idx >= T - j) { idx -= T - j; ++j; }
  if (j >= T) { *ti = -1; *tj = 1 << 20; } else { *ti = j + idx; *tj = j; }
}
/* Owner of tile `idx` of the reverse column-major enumeration of the tiles of columns >= 2.  Hardware waves 0, 4, 8 share a SIMD and fp64 MFMA
 * shares the DP units with the factor wave's scalar chain, so the two tile waves on that SIMD (tile-wave indices 3 and 7, "light") get 3 tiles
 * for every 5 of the nine others: tiles are dealt in rounds of 11, 9, 11, 9, 11 (= 51 per cycle; 153 tiles = 3 cycles -> 15 per heavy wave, 9
 * per light wave).  Slots grow with idx for every wave, so a wave's active tiles (column > k) are a suffix of its slots. */
CHOL_HD void chol_rr_owner(int idx, int ntl2, int *w, int *slot)
{
  if (ntl2 <= CHOL_RR_HEAVY_ONLY) { const int off = idx % CHOL_RR_NHEAVY; *w = off + off / 3; *slot = idx / CHOL_RR_NHEAVY; return; }
  const int cyc = idx / 51, pos = idx % 51;
  const int round = pos < 11 ? 0 : pos < 20 ? 1 : pos < 31 ? 2 : pos < 40 ? 3 : 4;
  const int off = pos - (round == 0 ? 0 : round == 1 ? 11 : round == 2 ? 20 : round == 3 ? 31 : 40);
  if (off < CHOL_RR_NHEAVY) { /* heavy waves in order: tile-wave indices 0,1,2,4,5,6,8,9,10 */
    *w = off + off / 3;
    *slot = cyc * 5 + round;
  } else {
    *w = off == CHOL_RR_NHEAVY ? 3 : 7;
    *slot = cyc * 3 + round / 2;
  }
}
/* the role table of an n-column block with tile skyline sky[24] (CHOL_RR_TAB_BYTES bytes; chol_schedule.c) */
void chol_potrf_table(int n, const unsigned char *sky, unsigned char *out);

typedef struct {
  int64_t l_off;      /* pivot block (n x n, ld ldl) */
  int64_t dinv_off;
  int64_t b_off;      /* first row of this row chunk inside the panel */
  int n, ldl, m, ldb; /* m rows (<= CHOL_TRSM_ROWS) */
  int flag;           /* fused POTRF+TRSM launch: index of the pivot block's POTRF descriptor in the same launch; program launch: its progress counter */
  int chan;           /* program launch: first counter of the follow channel the strip reports its column tiles on, or -1 */
  int band, pad;      /* program launch, banded leaf pivot wider than CHOL_FUSE_MAXN factored as ONE block: tile (J2, J) of its L is zero for
                       * J2 - J > band (<= 4): the strip keeps one L tile per step and wave (0: dense pivot block) */
} chol_trsm_desc;

#define CHOL_TRSM_W_MAXN 64 /* widest pivot block the one-wave-per-strip TRSM kernel takes (4 tiles); measured against k_trsm_rr: 6.6 vs 8.0 us at 32, 8.4 vs 9.6 at 64, 12.6 vs 11.1 at 128 */
#define CHOL_TRSM_WT_MAXN 144 /* widest pivot block of the throughput TRSM kernel (k_trsm_wt: one wave per strip, 12 strips per workgroup sharing the block's LDS image) */
#define CHOL_TRSM_WT_GROUP 12
#define CHOL_TRSM_WT_MIN 2048 /* strips (estimated from the panel heights) from which a step of the level schedule solves them with k_trsm_wt after a POTRF launch of its own */
#define CHOL_FUSE_MAXN 192 /* widest pivot block of a fused POTRF+TRSM launch (12 column tiles: three per wave of a strip) */
#define CHOL_FUSE_UPDATE_MAX 0 /* most 16x16 update tasks a fused launch carries (chol_schedule.c); measured on lapl_3375: 0/64/256 within noise (227 us), 900 -> 239, all -> 248: the role exists, a launch of its own is as fast */
#define CHOL_TRSM_ROWS 16

typedef struct {
  int64_t a_off, b_off; /* row 0 of the task's output sub-tile in the A rows / B rows of the source */
  int lda, ldb, k;
  int range;            /* 0: the source covers the whole sub-tile; else r0 | r1 << 8 | c0 << 16 | c1 << 24: it covers rows
                         * [r0, r1) x columns [c0, c1) of it only (grid-cell tasks: a 16x16 cell of the target block cuts
                         * through the cluster tiles of the reference) */
  int stage, pad;       /* program launch, extend-add jobs: the source may be read once the first `stage` staged waits of the job hold (0: no
                         * staged wait); a task's sources are sorted by stage */
} chol_upd_src;

typedef struct {
  int64_t c_off;      /* top-left element of this 16x16 (or smaller) output sub-tile */
  int ldc;
  short mv, nv;       /* valid rows / cols (<= 16) */
  int lower;          /* 1: diagonal sub-tile of a SYRK target, store only row >= col */
  int src_begin, src_end;
  int ar, br;         /* row offset of this sub-tile inside the A tile rows / B tile rows */
  int blk;            /* target block index (program launch: tasks of one block form the update jobs) */
} chol_upd_task;

/* ---- program launch (chol_build_program, k_program): the whole factorisation as ONE launch of resident workgroups that
 * draw jobs from a queue in a topological order and hand data to each other through counters in global memory.
 * Counters are monotonic across factorisations: a wait for `value` on counter c means  ctr[c] - epoch * total[c] >= value,
 * total[c] = what one factorisation adds to c. */
typedef struct { int ctr, value; } chol_wait;
typedef struct {
  int kind;            /* 0 POTRF of one pivot block, 1 TRSM group (<= 3 strips of one pivot block), 2 update group (16x16 tasks of one target block) */
  int first, n;        /* POTRF: descriptor index; TRSM: strips [first, first + n); update: tasks [first, first + n) */
  int wait_first, n_wait; /* chol_wait entries that must hold before the job touches its data */
  int sig[2], sig_add; /* counters raised by sig_add once the job's stores have completed (-1: none); POTRF jobs publish
                        * their column progress themselves */
  int ext_first, n_ext;/* POTRF: the external panels it follows (chol_ext), none = plain POTRF */
  int mode, n_pre;     /* n_pre: the leading waits that gate the whole job; the others are STAGED (update jobs of the extend-add: one per source
                        * pivot block in expected order of completion, chol_upd_src.stage) and are looked at by the tasks.  mode: update: 0 = one wave per task (light tasks, twelve at a time), 1 = four waves per task splitting K / the sources
                        * (heavy tasks, three at a time) */
} chol_job;
typedef struct {       /* one followed column tile: (rows of the follower's block) x (16 columns of a source pivot block).  A follower's items are
                        * queued in the order their columns are EXPECTED to arrive (position of the column in its source's pivot chain: the
                        * columns of two children that run side by side alternate) -- a fixed order: the sums are the same in every run */
  int64_t off;         /* first of the follower's rows in the first column of the tile */
  int ld, ncol;        /* leading dimension of the source panel, columns of the tile (<= 16) */
  int ctr, need;       /* counter raised by every one of the `need` strips covering the rows once it has stored this column tile */
} chol_ext;
/* How a follower consumes its list of followed column tiles (follow_external): in ROUNDS of one or two items.  The grouping is a pure
 * function of (position in the list, length of the list, column tiles of the follower) -- every wave of the workgroup walks the list for
 * itself and must count the same barriers (a grouping that depended on what had been published at the time dead-locked once: the waves
 * disagreed).  Shared by the kernel and the host-side check (cholamd_follow_rounds, tests/test_host.py). */
#define CHOL_FOLLOW_PAIR_MAXT 17   /* two column tiles share an LDS buffer of this many tiles */
#define CHOL_FOLLOW_SINGLE_TAIL 0  /* last items of a list that go one by one (0 / 1 / 2 measured: 189.5 / 190.4 / 190.2 us on lapl_3375) */
#define CHOL_FOLLOW_OWN_BEFORE 2   /* the follower's own tiles go in at the last round that starts at or before item n - CHOL_FOLLOW_OWN_BEFORE */
#if defined(__HIPCC__)
#define CHOL_HD __host__ __device__ __forceinline__
#else
#define CHOL_HD static inline
#endif
CHOL_HD int chol_follow_round(int i, int n_ext, int T)
{ /* items of the round that starts at item i */
  const int pair_lim = 2 * T <= CHOL_FOLLOW_PAIR_MAXT ? n_ext - CHOL_FOLLOW_SINGLE_TAIL : 0; /* items [i, i + 1] form a round while i + 1 < pair_lim */
  return i + 1 < pair_lim ? 2 : 1;
}
CHOL_HD int chol_follow_own_at(int n_ext, int T)
{ /* first item of the round in front of which the follower's own tiles are added */
  int own_at = 0;
  for (int i = 0; i < n_ext; i += chol_follow_round(i, n_ext, T))
    if (i <= n_ext - CHOL_FOLLOW_OWN_BEFORE) own_at = i;
  return own_at;
}

typedef struct {
  int n_job; chol_job *job;
  int n_wait; chol_wait *wait;
  int n_ext; chol_ext *ext;
  int n_ctr; int *ctr_total;
  int follow;          /* followers in use */
} chol_program;

/* one batched launch: descriptors [first, first + n) of the level's array of that kind */
typedef struct {
  int kind;  /* 0 potrf, 1 trsm, 2 update (16x16 tasks), 3 update (64x64 macro-tile tasks), 4 trsm (every strip's pivot block <= CHOL_TRSM_W_MAXN wide), 7 trsm (throughput form, k_trsm_wt),
              * 5 fused potrf (first, n) + trsm (first2, n2) + 16x16 update tasks (first3, n3) */
  int first, n, first2, n2, first3, n3;
} chol_phase;
/* kind 6 (distributed top levels): broadcast of the column blocks [first, first + n) of the level's bcast list: every rank receives
 * `count` doubles at arena offset `off` from rank `owner` */
typedef struct { int64_t off, count; int owner, heap; } chol_bcast; /* heap: heap index of the top separator the block belongs to */
/* ranks whose copy of a column block of top separator `heap` (tree level < log2(world)) can be non-zero before the extend-add exchange: the ranks
 * whose subtrees hang under the separator -- their updates are the only ones that reach it (A's entries of the block start on its owner:
 * cholamd_device_fill, which receives nothing from itself).  Bit r = rank r. */
static inline unsigned chol_top_contributors(int heap, int world)
{
  int d = 0, l = 0;
  while ((1 << d) < world) d++;
  while ((heap >> (l + 1)) > 0) l++;
  const int lo = (heap << (d - l)) - (1 << d), hi = ((heap + 1) << (d - l)) - (1 << d);
  unsigned m = 0u;
  for (int g = lo; g < hi; g++) m |= 1u << g;
  return m;
}

/* schedule switches of one device object (chol_schedule.c): read from the environment once, at cholamd_device_create */
#define CHOL_DIST_MIN 1024      /* dist_top = 2 (auto): the top levels are distributed when the root separator has this many columns */
#define CHOL_MT_MIN_TILES 8192 /* a phase whose 16x16 sub-tile count reaches this goes to 64x64 macro tiles for its larger targets */
typedef struct {
  int split_min, split_nb;  /* pivots wider than split_min are factored in column blocks of at most split_nb columns */
  int fuse;                 /* POTRF + TRSM of a column-block step in one launch */
  int fuse_update_max;      /* most 16x16 update tasks such a launch carries */
  int mt_min_tiles;
  int trsm_group;           /* unfused schedule: every pivot block's strips padded to this many, TRSM phases of kind 7 (the fp32 schedule: CHOL32_TRSM_GROUP); 0: off */
  int trsm_wt_min;          /* level schedule: steps with at least this many TRSM strips take the throughput TRSM (0: never) */
  int cells;                /* extend-add of small phases by 16x16 grid cells of the target blocks */
  int program;              /* small problems: the whole factorisation as one launch (chol_build_program) */
  int follow;               /* ... in which pivot blocks follow their children's / predecessor's TRSM strips */
  int dist_top;             /* 0 / 1 / 2 = auto (CHOL_DIST_MIN); world > 1: the levels above the cut are distributed over the ranks by column blocks (owner factors and
                             * solves a block, broadcasts it, every rank updates the column blocks it owns) instead of replicated */
  int skyline;              /* program launch: the tile-level skyline of the leaf pivots is used (zero early parts, zero tile updates skipped,
                             * banded leaves up to CHOL_RR_MAXN columns factored as one block) */
  int stage_chunk;          /* program launch: a banded leaf factored as one block hands its columns to the extend-add jobs in chunks of this many
                             * column tiles (every strip of the block then publishes its column tiles); 0: the whole block at once */
  int fine_upd;             /* program launch: followed strips wait for the update jobs into THEIR rows' block, not for all into the panel */
  int staged;               /* program launch: the extend-add jobs take their sources as the source pivots finish (staged waits) */
  int follow_tail_split;    /* the same tail for the next column block of a split pivot (one source: the strips of its own rows) */
  int follow_tail;          /* a follower wider than CHOL_FOLLOW_ALL_MAXT column tiles takes only the LAST follow_tail column tiles of each source
                             * itself; the columns before them reach its diagonal block through update jobs on other CUs (0: follows everything) */
  int merge_targets;        /* level schedule: extend-add targets (and the row runs of a panel) that are neighbours in storage are merged, so that the
                             * 64x64 macro tiles and the 16-row strips fill up (bit-identical sums) */
  int leaf_envelope;        /* level schedule: a LEAF's factor stays inside the envelope of A (nothing reaches a leaf from below), so its TRSM strips, trailing
                             * updates and extend-add sources leave out what is structurally zero: rows beyond the band of a column block, ancestor rows
                             * in front of their first entry, the columns of a source in front of its rows' first entries (identical results) */
  int super_blocks;         /* column blocks per super-block of a wide pivot: the trailing matrix beyond a super-block gets one update of
                             * rank super_blocks * (block width) instead of one per block */
} chol_sched_opts;
#define CHOL_SUPER_BLOCKS 3
#define CHOL_STAGE_CHUNK 0
void chol_sched_opts_default(chol_sched_opts *o);
void chol_sched_opts_from_env(chol_sched_opts *o);

#define CHOL_SPLIT_MIN 144 /* pivots wider than this are factored in column blocks (chol_schedule.c); measured on lapl_3375: 258 us unsplit, 240 us at 144/144 */
#ifndef CHOL_PROG_SPLIT_MIN
#define CHOL_PROG_SPLIT_MIN 176
#endif
/* CHOL_PROG_SPLIT_MIN: the program launch factors pivots up to this width whole when split_min is at its default (lapl_3375: its 161 / 163 / 174-column leaves
                                * as one 11-tile block, 177.9 -> 176.4 us; the level-by-level schedule is better off at 144: 215 against 220 us) */
#define CHOL_SPLIT_NB 144  /* ... of at most this many columns */

typedef struct {
  int level;
  int n_phase; chol_phase *phase;          /* launches of this level, in order */
  int n_potrf; chol_potrf_desc *potrf;
  int n_trsm; chol_trsm_desc *trsm;
  int n_task; chol_upd_task *task;         /* 16x16 sub-tile tasks (k_update) */
  int n_task_mt; chol_upd_task *task_mt;   /* 64x64 macro-tile tasks (k_update_mt): targets larger than 16x16 */
  int n_src; chol_upd_src *src;            /* task.src_begin/src_end index this array */
  int n_bcast; chol_bcast *bcast;          /* distributed top levels: the column blocks exchanged after each step */
} chol_level_work;

/* solve-phase descriptors */
typedef struct {
  int64_t a_off; int n, lda; int x_off; int sep;
  int64_t dinv_off; /* the separator's 16x16 diagonal-block inverses in the solve workspace */
  int band;         /* > 0: L(i, j) = 0 inside the separator's diagonal block for i - j > band (a LEAF's factor stays inside the envelope of A: the solve does not
                     * read the rows of a span's panel beyond it); 0: dense */
} chol_trsv_desc;
typedef struct {
  int64_t a_off; int m, n, lda; int x_off, y_off; /* y(m) -= A x(n)  or  y(n) -= A^T x(m); one descriptor per stored row run of a block.  Forward sources of the
                                                   * target-centric level solve (chol_solve_level.fw): y_off = the run's first row inside the target separator */
  int c_lo;                                       /* the run's rows are zero in the columns [0, c_lo) (rows of a LEAF's panel start at their first entry of A); 0: unknown */
} chol_gemv_desc;

struct cholamd_plan {
  int n, nz_file, levels, nsep, max_int_size;
  char banner[160];
  MM_typecode typecode;
  int *perm;        /* dof at permuted position */
  int *iperm;       /* permuted position of dof */
  int *sep_of_pos;  /* label per permuted position */
  int *sep_size, *sep_off;        /* index 1..nsep */
  int *tree;                      /* heap index 1..nsep -> label */
  int *heap_of, *level_of;        /* label -> heap index / tree level */
  chol_clusters *cl;              /* index 1..nsep */
  int nblk; chol_block *blk;
  int *blk_index;                 /* (nsep+1)^2 -> index into blk or -1 */
  int64_t *panel_off; int *panel_ld, *panel_rows; /* per label (panel_rows: rows stored) */
  int compact;                    /* row compaction of the ancestor blocks in force */
  int64_t arena_dense;            /* what the arena would be with every ancestor row stored */
  int64_t arena;                  /* doubles */
  int64_t ws_doubles;             /* workspace (inverted diagonal blocks) */
  int64_t *dinv_off;              /* per label */
  /* tril(A) mapped into the arena */
  int64_t nnz_a, dropped; int64_t *a_dst; double *a_val;
  /* A, both triangles, CSR over ORIGINAL dof indices (residual of the iterative refinement) */
  int64_t *csr_ptr; int *csr_col; double *csr_val;
  /* fill snapshots per interval label */
  int64_t *snap_n; cholamd_filled **snap;
  /* reference-order BLAS call list */
  int64_t nops, cap_ops; cholamd_op *ops;
  int64_t calls[16][4]; double flops[16][4];
  int64_t nnz_l;      /* exact scalar-symbolic nnz(L) of P A P^T */
  int64_t nnz_tiles;  /* area of the filled tiles the schedule touches (>= nnz_l) */
  double fmin;        /* sum_j colcount_j^2, the lower bound F_min of SURVEY 8d */
};

/* chol_symbolic.c */
int chol_plan_finish(struct cholamd_plan *p, int nz, const int *a_row, const int *a_col, const double *a_val);
const chol_block *chol_plan_block(const struct cholamd_plan *p, int r, int c);
int chol_ntiles(const struct cholamd_plan *p, int sep, int interval);
/* Build the device work lists of one tree level for (rank, world); caller frees with chol_level_work_free */
/* the whole factorisation as one program (single GPU); descriptors in `w`, jobs in `prog`; CHOLAMD_ERR_ARG (with the reason in
 * cholamd_last_error) if the problem does not qualify (a pivot block wider than CHOL_FUSE_MAXN, macro-tile phases) */
int chol_build_program(const struct cholamd_plan *p, const chol_sched_opts *opts, chol_level_work *w, chol_program *prog);
void chol_program_free(chol_program *prog);
/* host-side self-check: every wait is satisfiable by the signals of jobs the queue order lets run (simulated with `workers`
 * resident workgroups), counters total up, the jobs cover the work of the per-level lists */
int chol_program_check(const struct cholamd_plan *p, const chol_sched_opts *opts, int workers);
int chol_program_check_built(const struct cholamd_plan *p, const chol_sched_opts *opts, int workers, const chol_level_work *w, const chol_program *g);
int chol_build_level_work(const struct cholamd_plan *p, const chol_sched_opts *opts /* NULL: defaults */, int level, int rank, int world, chol_level_work *out);
void chol_level_work_free(chol_level_work *w);
int chol_owner_of(const struct cholamd_plan *p, int label, int world); /* -1: shared top of the tree */
int chol_split_level(int world);
/* solve lists of one tree level; caller frees with chol_solve_level_free */
typedef struct {
  int n_trsv; chol_trsv_desc *trsv;                 /* one per separator of the level (forward and backward) */
  int n_fw; chol_gemv_desc *fw;                     /* forward sources, grouped by target row chunk */
  int n_grp; int *grp_start; int *grp_rows;         /* grp_start[n_grp+1]; grp_rows = (row0, y_off) pairs */
  int n_bw; chol_gemv_desc *bw; int *bw_start;      /* backward sources per separator: bw_start[n_trsv+1] */
  /* driver-level solve (cholamd_solve): the (ancestor, separator) blocks `bw` cut into row chunks, one workgroup each:
   * forward: (block index, first row, first column, first non-zero column of these rows) quadruples, chunks of CHOL_SOLVE_FW_ROWS rows x CHOL_SOLVE_COLS columns (the tall blocks of the top
   * levels are few: without the column cut 40 workgroups carried a level); backward: (first run, end run, first column, 0) quadruples -- 64 columns of
   * one separator over a range of its row runs `bw` (runs are cut at CHOL_SOLVE_BW_ROWS rows; the range is all of the separator's runs unless the level
   * then has fewer than CHOL_SOLVE_BW_ITEMS workgroups) */
  int n_ifw; int *ifw;
  int n_ibw; int *ibw;
  int max_n;                                        /* widest separator of the level */
  int max_rows_under_span;                          /* most rows any separator of the level has to read under a 256-column span (band / dense) */
  int banded;                                       /* every separator of the level is at most one span wide or a leaf with a band of at most one span: its whole
                                                     * triangle can be solved by one workgroup in one launch (k_solve_leaf32) */
} chol_solve_level;
#define CHOL_SOLVE_FW_ROWS 256
#define CHOL_SOLVE_BW_ROWS 1024
#define CHOL_SOLVE_BW_COLS 64
#define CHOL_SOLVE_BW_ITEMS 4096
#define CHOL_SOLVE_COLS 1024
int chol_build_solve_level(const struct cholamd_plan *p, int level, chol_solve_level *out);
int chol_build_solve_level_part(const struct cholamd_plan *p, int level, int rank, int world, chol_solve_level *out);
void chol_solve_level_free(chol_solve_level *w);
void chol_set_error(const char *fmt, ...);

#ifdef __cplusplus
}
#endif
#endif
