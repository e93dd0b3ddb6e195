/* Device work lists: turns the reference-order tile schedule of one tree level into the three
 * batched launches of the HIP path.
 *
 *   POTRF  one descriptor per separator of the level (pivot = one tile).
 *   TRSM   the filled row tiles of all ancestor blocks (par, s) are contiguous row ranges of
 *          panel(s); adjacent tiles are merged into row runs and cut into chunks of at most
 *          CHOL_TRSM_ROWS rows (rows of a TRSM are independent, so this regrouping does not
 *          change any result bit).
 *   UPDATE target-centric: every filled C tile (gp, par, i, j) of the level owns the list of its
 *          sources (s, A tile (gp,s,i), B tile (par,s,j)) in the reference's program order and is
 *          cut into 16x16 output sub-tiles = one wavefront task each.  One owner per output
 *          element => no atomics, deterministic, and the accumulation order over descendants is
 *          the reference's (mmat.rg:1293-1347; "reads writes(rC)" serialises them, blas.rg:365).
 *
 * Multi-GPU: with `world` ranks the tree is cut at level d = log2(world); separators at levels
 * >= d belong to the rank owning their level-d ancestor, levels < d are shared (every rank runs
 * them after the extend-add exchange).  SURVEY 8e.
 */
#define _GNU_SOURCE
#include <stdlib.h>
#include <string.h>

#include "chol_plan.h"
#include "cholamd.h"

typedef struct cholamd_plan plan_t;
#define BIDX(p, r, c) ((p)->blk_index[(size_t)(r) * ((p)->nsep + 1) + (c)])

int chol_split_level(int world)
{
  int d = 0;
  while ((1 << d) < world) d++;
  return d;
}

int chol_owner_of(const plan_t *p, int label, int world)
{
  int d = chol_split_level(world), lvl = p->level_of[label];
  if (lvl < d) return -1;
  return (p->heap_of[label] >> (lvl - d)) - (1 << d);
}

typedef struct {
  int64_t key;   /* (C block index, tile id) */
  int64_t seq;   /* reference program order */
  int64_t c_off, a_off, b_off;
  int ldc, lda, ldb, m, n, k, syrk;
  int bc, crow, ccol; /* the target block and the tile's first row / column inside it */
  int src_sep;        /* the separator whose panel the contribution comes from */
} upd_tuple;

/* a target of the extend-add with its sources tu[i, e): one cluster-tile pair, or several merged ones */
typedef struct { int i, e, bc, crow, ccol, m, n, syrk; int64_t c_off; int ldc, dead; } tgt_group;
static int cmp_group_rows(const void *x, const void *y)
{
  const tgt_group *a = x, *b = y;
  if (a->bc != b->bc) return a->bc < b->bc ? -1 : 1;
  if (a->ccol != b->ccol) return a->ccol < b->ccol ? -1 : 1;
  if (a->n != b->n) return a->n < b->n ? -1 : 1;
  return a->crow < b->crow ? -1 : (a->crow > b->crow);
}
static int cmp_group_first(const void *x, const void *y)
{
  const tgt_group *a = x, *b = y;
  return a->i < b->i ? -1 : (a->i > b->i);
}
/* by the separator of the first source, then as cmp_group_first (the leaf level: the targets fed by one leaf's panel next to each other in the task list) */
static int cmp_group_source(const void *x, const void *y, void *tu_)
{
  const tgt_group *a = x, *b = y;
  const upd_tuple *cmp_group_tu = tu_;
  const int sa = cmp_group_tu[a->i].src_sep, sb = cmp_group_tu[b->i].src_sep;
  if (sa != sb) return sa < sb ? -1 : 1;
  return a->i < b->i ? -1 : (a->i > b->i);
}
/* the sources of b are those of a, shifted by dr rows in the A operand and dc rows in the B operand */
static int same_sources(const upd_tuple *tu, const tgt_group *a, const tgt_group *b, int dr, int dc);

static int cmp_tuple(const void *x, const void *y)
{
  const upd_tuple *a = x, *b = y;
  if (a->key != b->key) return a->key < b->key ? -1 : 1;
  return a->seq < b->seq ? -1 : (a->seq > b->seq);
}

static int same_sources(const upd_tuple *tu, const tgt_group *a, const tgt_group *b, int dr, int dc)
{
  if (a->e - a->i != b->e - b->i) return 0;
  for (int q = 0; q < a->e - a->i; q++) {
    const upd_tuple *x = &tu[a->i + q], *y = &tu[b->i + q];
    if (x->src_sep != y->src_sep || x->lda != y->lda || x->ldb != y->ldb || x->k != y->k || y->a_off != x->a_off + dr || y->b_off != x->b_off + dc) return 0;
  }
  return 1;
}

/* index of the filled tiles of snapshot lbl per block: first[b], count[b] */
static void index_snapshot(const plan_t *p, int lbl, int64_t *first, int *count)
{
  for (int b = 0; b < p->nblk; b++) { first[b] = 0; count[b] = 0; }
  const cholamd_filled *v = p->snap[lbl];
  for (int64_t i = 0; i < p->snap_n[lbl]; i++) {
    int b = BIDX(p, v[i].sep_x, v[i].sep_y);
    if (count[b] == 0) first[b] = i;
    count[b]++;
  }
}

/* growable arrays of the level work */
typedef struct { int64_t c_off; int ldc, m, n, syrk, src_begin, src_end, blk, ar0, br0; } upd_target;
typedef struct { chol_level_work *w; const chol_sched_opts *o; int force_fine; int cur_blk; const int *follow_lim;
  /* distributed top level: targets are cut at the column blocks of the target separator and kept only where this rank owns the block */
  int dist_world, dist_rank; const struct cholamd_plan *dist_plan; int tgt_sep, tgt_col0; int cap_b;
  const int *sw_of; /* program launch, staged waits: columns per stage of a separator's sources (its column-block width, or the chunk of a banded leaf) */
  int cur_band; const int *bw_of; /* program launch: band of the pivot block whose strips are being pushed (0: dense); column-block width per
                                   * separator where it differs from pivot_block_width() (banded leaves factored as one block), or NULL */
  int cap_p, cap_t, cap_k, cap_km, cap_s, cap_ph; upd_target *pend; int n_pend, cap_pend; } builder;
/* Blocking of a pivot at the schedule level.  The POTRF kernel takes pivots up to CHOL_RR_MAXN whole, but one
 * workgroup's MFMA throughput bounds the early steps of a large one (the trailing update of step 0 of a
 * 17 x 17 tile grid is 120 tile updates on one CU); a pivot wider than CHOL_SPLIT_MIN is therefore factored
 * in equal column blocks of at most CHOL_SPLIT_NB columns (a multiple of 16), with the TRSM of the rows below
 * and the rank-nb update of the trailing columns spread over the whole chip between the blocks. */
static int env_int(const char *name, int dflt) { const char *e = getenv(name); return e && *e ? atoi(e) : dflt; }
/* schedule switches: defaults, overridden once per device object from the environment (chol_sched_opts_from_env,
 * called by cholamd_device_create) or by cholamd_device_set_option */
void chol_sched_opts_default(chol_sched_opts *o)
{
  o->split_min = CHOL_SPLIT_MIN; o->split_nb = CHOL_SPLIT_NB; o->fuse = 1; o->fuse_update_max = CHOL_FUSE_UPDATE_MAX;
  o->mt_min_tiles = CHOL_MT_MIN_TILES; o->trsm_wt_min = CHOL_TRSM_WT_MIN; o->trsm_group = 0; o->cells = 1; o->program = 1; o->follow = 1; o->super_blocks = CHOL_SUPER_BLOCKS; o->dist_top = 2; o->follow_tail = CHOL_FOLLOW_TAIL; o->follow_tail_split = CHOL_FOLLOW_TAIL; o->staged = 1; o->fine_upd = 1; o->skyline = 1; o->stage_chunk = CHOL_STAGE_CHUNK; o->merge_targets = 1; o->leaf_envelope = 1;
}
void chol_sched_opts_from_env(chol_sched_opts *o)
{
  chol_sched_opts_default(o);
  o->split_min = env_int("CHOLAMD_SPLIT_MIN", o->split_min);
  o->split_nb = env_int("CHOLAMD_SPLIT_NB", o->split_nb);
  o->fuse = !env_int("CHOLAMD_NO_FUSE", 0);
  o->fuse_update_max = env_int("CHOLAMD_FUSE_UPDATE_MAX", o->fuse_update_max);
  o->mt_min_tiles = env_int("CHOLAMD_MT_MIN_TILES", o->mt_min_tiles);
  o->trsm_wt_min = env_int("CHOLAMD_TRSM_WT_MIN", o->trsm_wt_min);
  o->cells = !env_int("CHOLAMD_NO_CELLS", 0);
  o->program = !env_int("CHOLAMD_NO_PROGRAM", 0);
  o->follow = !env_int("CHOLAMD_NO_FOLLOW", 0);
  o->dist_top = env_int("CHOLAMD_DIST_TOP", o->dist_top);
  o->super_blocks = env_int("CHOLAMD_SUPER_BLOCKS", o->super_blocks);
  o->follow_tail = env_int("CHOLAMD_FOLLOW_TAIL", o->follow_tail);
  o->follow_tail_split = env_int("CHOLAMD_FOLLOW_TAIL_SPLIT", o->follow_tail_split);
  o->staged = !env_int("CHOLAMD_NO_STAGED", 0);
  o->fine_upd = !env_int("CHOLAMD_NO_FINE_UPD", 0);
  o->skyline = !env_int("CHOLAMD_NO_SKYLINE", 0);
  o->merge_targets = !env_int("CHOLAMD_NO_MERGE_TARGETS", 0);
  o->leaf_envelope = !env_int("CHOLAMD_NO_LEAF_ENVELOPE", 0);
  o->stage_chunk = env_int("CHOLAMD_STAGE_CHUNK", o->stage_chunk);
}
static int split_nb(const chol_sched_opts *o) { int v = o->split_nb; if (v > CHOL_RR_MAXN) v = CHOL_RR_MAXN; v = (v + 15) / 16 * 16; if (v < 16) v = 16; return v; }
static int pivot_blocks(const chol_sched_opts *o, int n) { return n > o->split_min || n > CHOL_RR_MAXN ? (n + split_nb(o) - 1) / split_nb(o) : 1; }
static int pivot_block_width(const chol_sched_opts *o, int n) { const int nb = pivot_blocks(o, n); return nb == 1 ? n : ((n + nb - 1) / nb + 15) / 16 * 16; }

/* a phase whose 16x16 sub-tile count reaches this goes to 64x64 macro tiles for its larger targets:
 * below it the 16x16 split-K workgroups are what fills the 256 CUs, above it their 4x operand
 * re-reads are what costs */

static void push_phase_f(builder *B, int kind, int first, int n, int force);
static void push_phase(builder *B, int kind, int first, int n) { push_phase_f(B, kind, first, n, 0); }
/* force: an empty launch keeps its place in the sequence (distributed top levels: every rank walks the same phase sequence,
 * the broadcasts in it are collective) */
static void push_phase_f(builder *B, int kind, int first, int n, int force)
{
  if (n <= 0 && !force) return;
  if (n < 0) n = 0;
  chol_level_work *w = B->w;
  if (w->n_phase == B->cap_ph) { B->cap_ph = B->cap_ph ? 2 * B->cap_ph : 16; w->phase = realloc(w->phase, B->cap_ph * sizeof(chol_phase)); }
  chol_phase ph = { kind, first, n, 0, 0, 0, 0 };
  w->phase[w->n_phase++] = ph;
}
/* The POTRF role's tables for one block, exactly what potrf_rr_body's prologue would build (and builds, for descriptors without one): the tile
 * of every (slot, wave), the first step with a non-zero update per slot (skyline), and per (step, wave) the slots that have work and the panel
 * tiles to solve. */
void chol_potrf_table(int n, const unsigned char *sky, unsigned char *out)
{
  int *mask = (int *)(out + CHOL_RR_TAB_MASK);
  unsigned short *ij = (unsigned short *)(out + CHOL_RR_TAB_IJ);
  unsigned char *km = out + CHOL_RR_TAB_KM;
  memset(out, 0, CHOL_RR_TAB_BYTES);
  for (int t = 0; t < CHOL_RR_SLOTS * CHOL_RR_NW; t++) ij[t] = 0xffff;
  const int T = (n + CHOL_NB - 1) / CHOL_NB, ntl = T * (T + 1) / 2, ntl2 = (T - 2) * (T - 1) / 2;
  for (int t = 2 * T - 1; t < ntl; t++) {
    int ti, tj, ow, os;
    chol_rr_tile_of_index(t, T, &ti, &tj);
    chol_rr_owner(ntl - 1 - t, ntl2, &ow, &os);
    ij[os * CHOL_RR_NW + ow] = (unsigned short)(ti | (tj << 8));
    const int si = sky[ti < 23 ? ti : 23], sj = sky[tj < 23 ? tj : 23], kmin = si > sj ? si : sj;
    km[os * CHOL_RR_NW + ow] = (unsigned char)(kmin | (tj < si ? CHOL_RR_KM_ZERO : 0));
    const int last = ti == tj ? tj - 2 : tj - 1;
    for (int k = kmin < last ? kmin : last; k <= last; k++) mask[k * CHOL_RR_NW + ow] |= 1 << os;
  }
  for (int k = 0; k < T; k++)
    for (int i = k + 2; i < T; i++)
      if (sky[i < 23 ? i : 23] <= k) {
        const int hv = i % CHOL_RR_NHEAVY;
        mask[k * CHOL_RR_NW + hv + hv / 3] |= 1 << (i < k + 2 + CHOL_RR_NHEAVY ? CHOL_RR_M_SOLVE0 : CHOL_RR_M_SOLVE1);
      }
}
static void push_potrf(builder *B, chol_potrf_desc d)
{
  chol_level_work *w = B->w;
  if (w->n_potrf == B->cap_p) { B->cap_p = B->cap_p ? 2 * B->cap_p : 32; w->potrf = realloc(w->potrf, B->cap_p * sizeof(chol_potrf_desc)); }
  w->potrf[w->n_potrf++] = d;
}
/* a run of `m` consecutive panel rows -> strips of CHOL_TRSM_ROWS */
static void push_trsm_run(builder *B, int64_t l_off, int64_t dinv_off, int64_t b_off, int n, int ld, int m, int flag)
{
  chol_level_work *w = B->w;
  for (int r0 = 0; r0 < m; r0 += CHOL_TRSM_ROWS) {
    if (w->n_trsm == B->cap_t) { B->cap_t = B->cap_t ? 2 * B->cap_t : 64; w->trsm = realloc(w->trsm, B->cap_t * sizeof(chol_trsm_desc)); }
    const int mm = m - r0 < CHOL_TRSM_ROWS ? m - r0 : CHOL_TRSM_ROWS;
    chol_trsm_desc td = { l_off, dinv_off, b_off + r0, n, ld, mm, ld, flag, -1, B->cur_band, 0 };
    w->trsm[w->n_trsm++] = td;
  }
}
/* k_trsm_w (four strips) and the fused POTRF+TRSM launch (three) give the strips of a workgroup one pivot block:
 * after the strips of a block, placeholders (m = 0) fill the group, counted from the first strip of the phase */
static void pad_trsm_group(builder *B, int phase_first, int group, int64_t l_off, int64_t dinv_off, int64_t b_off, int n, int ld, int flag)
{
  chol_level_work *w = B->w;
  while ((w->n_trsm - phase_first) % group != 0) {
    if (w->n_trsm == B->cap_t) { B->cap_t = B->cap_t ? 2 * B->cap_t : 64; w->trsm = realloc(w->trsm, B->cap_t * sizeof(chol_trsm_desc)); }
    chol_trsm_desc td = { l_off, dinv_off, b_off, n, ld, 0, ld, flag, -1, B->cur_band, 0 };
    w->trsm[w->n_trsm++] = td;
  }
}
static int push_src(builder *B, chol_upd_src sd)
{
  chol_level_work *w = B->w;
  if (w->n_src == B->cap_s) { B->cap_s = B->cap_s ? 2 * B->cap_s : 256; w->src = realloc(w->src, B->cap_s * sizeof(chol_upd_src)); }
  w->src[w->n_src] = sd;
  return w->n_src++;
}
/* tasks of one m x n target whose sources are [src_begin, src_end): 16x16 sub-tiles for small targets
 * (k_update: the four waves split K), 64x64 macro tiles otherwise (k_update_mt: LDS-staged panels) */
#ifndef MT_ORDER
#define MT_ORDER 8
#endif
static void emit_tasks(builder *B, int macro, int64_t c_off, int ldc, int m, int n, int syrk, int src_begin, int src_end, int blk, int ar0, int br0)
{
  chol_level_work *w = B->w;
  const int ts = macro ? 64 : 16;
  const int tr = (m + ts - 1) / ts, tc = (n + ts - 1) / ts;
  /* macro tiles in MT_ORDER x MT_ORDER blocks of the tile grid: the tasks an XCD works on at one time (a contiguous stretch of the list, k_update_mt)
   * then share MT_ORDER row panels and MT_ORDER column panels through its L2 instead of one row panel and sixty-four column panels */
  const int blk_o = macro ? MT_ORDER : (tr > tc ? tr : tc) + 1;
  for (int a0 = 0; a0 < tr; a0 += blk_o)
  for (int b0 = 0; b0 < tc; b0 += blk_o)
  for (int a = a0; a < tr && a < a0 + blk_o; a++)
    for (int b = b0; b < tc && b < b0 + blk_o; b++) {
      if (syrk && b > a) continue;
      chol_upd_task *t;
      if (macro) {
        if (w->n_task_mt == B->cap_km) { B->cap_km = B->cap_km ? 2 * B->cap_km : 256; w->task_mt = realloc(w->task_mt, B->cap_km * sizeof(chol_upd_task)); }
        t = &w->task_mt[w->n_task_mt++];
      } else {
        if (w->n_task == B->cap_k) { B->cap_k = B->cap_k ? 2 * B->cap_k : 256; w->task = realloc(w->task, B->cap_k * sizeof(chol_upd_task)); }
        t = &w->task[w->n_task++];
      }
      memset(t, 0, sizeof *t);
      t->c_off = c_off + a * ts + (int64_t)b * ts * ldc;
      t->ldc = ldc;
      t->mv = (short)(m - a * ts < ts ? m - a * ts : ts);
      t->nv = (short)(n - b * ts < ts ? n - b * ts : ts);
      t->lower = (syrk && a == b);
      t->src_begin = src_begin; t->src_end = src_end;
      t->ar = ar0 + a * ts; t->br = br0 + b * ts;
      t->blk = blk;
    }
}

/* targets of the current update phase; flush_targets() chooses the tile shape once the phase is complete */
static void push_target(builder *B, int64_t c_off, int ldc, int m, int n, int syrk, int src_begin, int src_end, int ar0, int br0)
{
  if (m <= 0 || n <= 0) return;
  if (B->n_pend == B->cap_pend) { B->cap_pend = B->cap_pend ? 2 * B->cap_pend : 256; B->pend = realloc(B->pend, B->cap_pend * sizeof(upd_target)); }
  upd_target t = { c_off, ldc, m, n, syrk, src_begin, src_end, B->cur_blk, ar0, br0 };
  B->pend[B->n_pend++] = t;
}
static int pivot_block_width(const chol_sched_opts *o, int n);
/* owner of column block `blk` of top separator `sep` among `world` ranks (cyclic, shifted by the separator's heap index) */
static int dist_owner(const struct cholamd_plan *p, int sep, int blk, int world) { return (blk + p->heap_of[sep]) % world; }
static void push_tasks(builder *B, int64_t c_off, int ldc, int m, int n, int syrk, int src_begin, int src_end)
{
  if (B->dist_world <= 1) { push_target(B, c_off, ldc, m, n, syrk, src_begin, src_end, 0, 0); return; }
  /* cut the target's columns at the column blocks of the target separator; keep the pieces this rank owns.  Columns j of the
   * target are columns tgt_col0 + j of the separator's pivot; a piece of a SYRK target is its own triangle plus the rows below */
  const int bw = pivot_block_width(B->o, B->dist_plan->sep_size[B->tgt_sep]);
  for (int j0 = 0; j0 < n;) {
    const int blk = (B->tgt_col0 + j0) / bw;
    int j1 = (blk + 1) * bw - B->tgt_col0;
    if (j1 > n) j1 = n;
    if (dist_owner(B->dist_plan, B->tgt_sep, blk, B->dist_world) == B->dist_rank) {
      if (syrk) {
        push_target(B, c_off + j0 + (int64_t)j0 * ldc, ldc, j1 - j0, j1 - j0, 1, src_begin, src_end, j0, j0);
        push_target(B, c_off + j1 + (int64_t)j0 * ldc, ldc, m - j1, j1 - j0, 0, src_begin, src_end, j1, j0);
      } else push_target(B, c_off + (int64_t)j0 * ldc, ldc, m, j1 - j0, 0, src_begin, src_end, 0, j0);
    }
    j0 = j1;
  }
}
static void flush_targets(builder *B)
{
  int64_t fine = 0;
  for (int i = 0; i < B->n_pend; i++) {
    const int64_t tr = (B->pend[i].m + 15) / 16, tc = (B->pend[i].n + 15) / 16;
    fine += B->pend[i].syrk ? tr * (tr + 1) / 2 : tr * tc;
  }
  const int big = fine >= B->o->mt_min_tiles && !B->force_fine;
  for (int i = 0; i < B->n_pend; i++) {
    const upd_target *t = &B->pend[i];
    emit_tasks(B, big && (t->m > 16 || t->n > 16), t->c_off, t->ldc, t->m, t->n, t->syrk, t->src_begin, t->src_end, t->blk, t->ar0, t->br0);
  }
  B->n_pend = 0;
}

/* Extend-add targets of a level as 16x16 GRID CELLS of the target blocks instead of the reference's cluster tiles.
 * The fixtures' clusters are a few rows high (most C tiles of lapl_3375 are 1-2 rows by 1-16 columns): one task per
 * 16x16 piece of a cluster tile executes 5x the useful flops and, worse, is 6 000 latency-bound tasks.  A cell collects
 * from every source the part of its contribution that falls inside the cell (rows [r0, r1) x columns [c0, c1) of it,
 * chol_upd_src.range); sources stay in program order, every element keeps exactly one owner.  Diagonal blocks: cells
 * above the diagonal are skipped, diagonal cells store row >= column only (the reference skips col > row cluster
 * pairs and runs SYRK on col == row, blas.rg:396-431: the same elements).  Returns the number of tasks. */
typedef struct { int64_t key, seq, c_off, a_off, b_off; int ldc, lda, ldb, k, mv, nv, lower, range, src_sep; } cell_piece;
static int cmp_piece(const void *x, const void *y)
{
  const cell_piece *a = x, *b = y;
  if (a->key != b->key) return a->key < b->key ? -1 : 1;
  return a->seq < b->seq ? -1 : (a->seq > b->seq);
}
static int emit_cell_tasks(builder *B, const plan_t *p, const upd_tuple *tu, int ntu)
{
  chol_level_work *w = B->w;
  int cap = 4 * ntu + 16, np = 0;
  cell_piece *pc = malloc((size_t)cap * sizeof(cell_piece));
  for (int i = 0; i < ntu; i++) {
    const upd_tuple *u = &tu[i];
    const chol_block *Bc = &p->blk[u->bc];
    const int diag = Bc->r == Bc->c;
    for (int I = u->crow / 16; I <= (u->crow + u->m - 1) / 16; I++)
      for (int J = u->ccol / 16; J <= (u->ccol + u->n - 1) / 16; J++) {
        if (diag && J > I) continue;
        /* program launch with followers: the leading lim x lim part of a parent's diagonal block receives its children's
         * contributions inside the parent's POTRF job (follow_external), not from update tasks */
        if (B->follow_lim && B->follow_lim[u->bc] > 0 && 16 * I < B->follow_lim[u->bc] && 16 * J < B->follow_lim[u->bc]) continue;
        if (B->dist_world > 1 && dist_owner(p, Bc->c, (16 * J) / pivot_block_width(B->o, p->sep_size[Bc->c]), B->dist_world) != B->dist_rank) continue; /* the cell's column block is another rank's */
        if (np == cap) { cap *= 2; pc = realloc(pc, (size_t)cap * sizeof(cell_piece)); }
        cell_piece *q = &pc[np++];
        const int r0 = (u->crow > 16 * I ? u->crow : 16 * I) - 16 * I, r1 = (u->crow + u->m < 16 * I + 16 ? u->crow + u->m : 16 * I + 16) - 16 * I;
        const int c0 = (u->ccol > 16 * J ? u->ccol : 16 * J) - 16 * J, c1 = (u->ccol + u->n < 16 * J + 16 ? u->ccol + u->n : 16 * J + 16) - 16 * J;
        q->key = ((int64_t)u->bc << 40) | ((int64_t)I << 20) | (int64_t)J;
        q->seq = u->seq;
        q->c_off = chol_block_row(Bc, 16 * I) + (int64_t)(16 * J) * Bc->ld; q->ldc = Bc->ld;
        q->mv = Bc->rows - 16 * I < 16 ? Bc->rows - 16 * I : 16;
        q->nv = Bc->cols - 16 * J < 16 ? Bc->cols - 16 * J : 16;
        q->lower = diag && I == J;
        /* operand rows as seen from row 0 / column 0 of the cell (only [r0, r1) / [c0, c1) are read) */
        q->a_off = u->a_off + (16 * I - u->crow); q->lda = u->lda;
        q->b_off = u->b_off + (16 * J - u->ccol); q->ldb = u->ldb;
        q->k = u->k;
        q->src_sep = u->src_sep;
        q->range = r0 | (r1 << 8) | (c0 << 16) | (c1 << 24);
      }
  }
  qsort(pc, np, sizeof(cell_piece), cmp_piece);
  int ntask = 0;
  for (int i = 0; i < np;) {
    int e = i + 1;
    while (e < np && pc[e].key == pc[i].key) e++;
    const int sb = w->n_src;
    for (int q = i; q < e;) {
      /* the pieces one source panel sends into this cell (one per cluster-tile pair, consecutive in program order)
       * share their operand rows: one source entry over the bounding rows x columns.  The rows in between belong to
       * unfilled tiles of the panel: structural zeros of L, still zero in the arena */
      int r0 = pc[q].range & 255, r1 = (pc[q].range >> 8) & 255, c0 = (pc[q].range >> 16) & 255, c1 = (pc[q].range >> 24) & 255, f = q + 1;
      while (f < e && pc[f].a_off == pc[q].a_off && pc[f].b_off == pc[q].b_off && pc[f].k == pc[q].k) {
        const int a0 = pc[f].range & 255, a1 = (pc[f].range >> 8) & 255, b0 = (pc[f].range >> 16) & 255, b1 = (pc[f].range >> 24) & 255;
        if (a0 < r0) r0 = a0;
        if (a1 > r1) r1 = a1;
        if (b0 < c0) c0 = b0;
        if (b1 > c1) c1 = b1;
        f++;
      }
      const int rg_ = r0 | (r1 << 8) | (c0 << 16) | (c1 << 24);
      if (B->force_fine && B->o->staged) {
        /* program launch with staged waits: one source per pivot BLOCK of the source separator (its columns c0 .. c0 + nb of the
         * panel), tagged -(64 separator + block) - 1 for now (emit_update_jobs turns the tag into the stage): what is left to
         * do when the separator's last block has been solved is that block's columns, not the whole pivot's */
        const int bw_ = B->sw_of ? B->sw_of[pc[q].src_sep] : B->bw_of ? B->bw_of[pc[q].src_sep] : pivot_block_width(B->o, p->sep_size[pc[q].src_sep]);
        for (int cb = 0, st_ = 0; cb < pc[q].k; cb += bw_, st_++) {
          const int kb = pc[q].k - cb < bw_ ? pc[q].k - cb : bw_;
          chol_upd_src sd = { pc[q].a_off + (int64_t)cb * pc[q].lda, pc[q].b_off + (int64_t)cb * pc[q].ldb, pc[q].lda, pc[q].ldb, kb, rg_, -(64 * pc[q].src_sep + st_) - 1, 0 };
          push_src(B, sd);
        }
      } else {
        chol_upd_src sd = { pc[q].a_off, pc[q].b_off, pc[q].lda, pc[q].ldb, pc[q].k, rg_, 0, 0 };
        push_src(B, sd);
      }
      q = f;
    }
    if (w->n_task == B->cap_k) { B->cap_k = B->cap_k ? 2 * B->cap_k : 256; w->task = realloc(w->task, B->cap_k * sizeof(chol_upd_task)); }
    chol_upd_task *t = &w->task[w->n_task++];
    memset(t, 0, sizeof *t);
    t->c_off = pc[i].c_off; t->ldc = pc[i].ldc;
    t->mv = (short)pc[i].mv; t->nv = (short)pc[i].nv;
    t->lower = pc[i].lower;
    t->src_begin = sb; t->src_end = w->n_src;
    t->blk = (int)(pc[i].key >> 40);
    ntask++;
    i = e;
  }
  free(pc);
  return ntask;
}
/* small update phases go cell by cell, large ones (macro tiles pay) tile by tile */
static int tuples_are_small(const chol_sched_opts *o, const upd_tuple *tu, int ntu)
{
  const int mt_min = o->mt_min_tiles;
  if (!o->cells) return 0;
  int64_t fine = 0;
  for (int i = 0; i < ntu; i++)
    if (i == 0 || tu[i].key != tu[i - 1].key) fine += (int64_t)((tu[i].m + 15) / 16) * ((tu[i].n + 15) / 16);
  return fine < mt_min;
}

/* filled row runs of the ancestor blocks of panel(s): (arena offset of the run's first row in column 0
 * of the panel, rows); tiles come in increasing row order and adjacent ones are merged */
typedef struct { int64_t off; int m; int64_t pos; } row_run; /* pos: the run's first row in the panel's uncompacted row numbering (adjacency is decided there) */
/* which: 0 = every ancestor, 1 = the parent only, 2 = every ancestor but the parent */
static int ancestor_runs_of(const plan_t *p, int h, int which, int by_storage, const cholamd_filled *snap, const int64_t *first, const int *count, row_run **out)
{
  const int s = p->tree[h];
  int cap = 16, n = 0;
  row_run *r = malloc(cap * sizeof(row_run));
  int64_t base = p->sep_size[s]; /* rows in front of the block in the uncompacted panel */
  for (int hp = h / 2; hp >= 1; base += p->sep_size[p->tree[hp]], hp /= 2) {
    if ((which == 1 && hp != h / 2) || (which == 2 && hp == h / 2)) continue;
    const int b = BIDX(p, p->tree[hp], s);
    const chol_block *Bk = &p->blk[b];
    for (int q = 0; q < count[b]; q++) {
      const cholamd_filled *f = &snap[first[b] + q];
      const int64_t off = chol_block_row(Bk, f->lo_x - Bk->lo_x), pos = base + (f->lo_x - Bk->lo_x);
      const int m = f->hi_x - f->lo_x + 1;
      /* adjacent rows (of adjacent kept tiles) are adjacent in storage; with `by_storage` also runs that only storage makes neighbours
       * (the unkept tiles between them have no rows): the rows of a panel are independent in the TRSM and in the panel's own trailing
       * update, so longer runs mean fuller strips and macro tiles and the same numbers */
      if (n > 0 && (by_storage || r[n - 1].pos + r[n - 1].m == pos) && r[n - 1].off + r[n - 1].m == off) { r[n - 1].m += m; continue; }
      if (n == cap) { cap *= 2; r = realloc(r, cap * sizeof(row_run)); }
      r[n].off = off; r[n].m = m; r[n].pos = pos; n++;
    }
  }
  *out = r;
  return n;
}
static int ancestor_runs(const plan_t *p, int h, int by_storage, const cholamd_filled *snap, const int64_t *first, const int *count, row_run **out)
{
  return ancestor_runs_of(p, h, 0, by_storage, snap, first, count, out);
}

static int **leaf_row_first(const plan_t *p);
/* smallest first-entry column over the panel rows [pr, pr + m) of a leaf */
static int rows_first(const int *first, int pr, int m, int n)
{
  int f = n;
  for (int r = 0; r < m; r++) if (first[pr + r] < f) f = first[pr + r];
  return f;
}
int chol_build_level_work(const plan_t *p, const chol_sched_opts *opts, int level, int rank, int world, chol_level_work *w)
{
  chol_sched_opts dflt;
  if (!opts) { chol_sched_opts_default(&dflt); opts = &dflt; }
  memset(w, 0, sizeof *w);
  w->level = level;
  builder Bd; memset(&Bd, 0, sizeof Bd); Bd.w = w; Bd.o = opts;
  builder *B = &Bd;
  const int L = p->levels, lbl = L - 1 - level;
  const int d = chol_split_level(world);
  if (world < 1 || (1 << d) != world || d > L - 1 || rank < 0 || rank >= world) { chol_set_error("bad partition rank %d of %d", rank, world); return CHOLAMD_ERR_ARG; }
  const int h0 = 1 << level, h1 = (1 << (level + 1)) - 1;
  int64_t *first = malloc(p->nblk * sizeof(int64_t));
  int *count = malloc(p->nblk * sizeof(int));
  index_snapshot(p, lbl, first, count);
  const cholamd_filled *snap = p->snap[lbl];

  /* the separators of this level this rank works on */
  int *hs = malloc((h1 - h0 + 1) * sizeof(int)), nh = 0, steps = 0;
  for (int h = h0; h <= h1; h++) {
    const int s = p->tree[h];
    if (level >= d && chol_owner_of(p, s, world) != rank) continue;
    if (count[BIDX(p, s, s)] == 0 || p->sep_size[s] == 0) continue;
    hs[nh++] = h;
    if (pivot_blocks(opts, p->sep_size[s]) > steps) steps = pivot_blocks(opts, p->sep_size[s]);
  }
  if (steps == 0) steps = 1;
  /* leaves: first entry of every panel row (leaf_envelope) */
  int **rfirst = (level == L - 1 && L > 1 && opts->leaf_envelope && nh > 0) ? leaf_row_first(p) : NULL;
  /* ... their ancestor row runs in 64-row pieces where the level is throughput work (the generated grids); on a small problem (lapl_3375: 16 leaves, fused
   * latency-bound launches) more runs are more padded strip groups: 210 -> 227 us */
  int64_t level_rows = 0;
  for (int q = 0; q < nh; q++) level_rows += p->panel_rows[p->tree[hs[q]]];
  const int fine_pieces = level_rows >= 65536;
  /* distributed top level (world > 1, option dist_top): column block `st` of separator s is factored and solved by its owner,
   * broadcast, and every rank applies the updates into the column blocks IT owns (push_tasks / emit_cell_tasks cut and filter
   * the targets); every rank walks the same phase sequence */
  const int dist = world > 1 && level < d && (opts->dist_top == 1 || (opts->dist_top == 2 && p->sep_size[p->tree[1]] >= CHOL_DIST_MIN));
  if (dist) { B->dist_world = world; B->dist_rank = rank; B->dist_plan = p; }
  int fused_last = -1; /* the level's last fused launch, if nothing was launched after it */
  int fuse = opts->fuse; /* POTRF + TRSM of a step in one launch, if every block fits its TRSM role */
  for (int q = 0; q < nh; q++) if (pivot_block_width(opts, p->sep_size[p->tree[hs[q]]]) > CHOL_FUSE_MAXN) fuse = 0;
  /* pivots: small ones whole in step 0; big ones in pivot_block_width()-column blocks, each step =
   * POTRF of the diagonal block, TRSM of every row below it (rows of the pivot and filled ancestor rows
   * alike), rank-nb update of the remaining columns of those rows */
  for (int st = 0; st < steps; st++) {
    const int p0 = w->n_potrf, t0 = w->n_trsm, k0 = w->n_task, km0 = w->n_task_mt, b0 = w->n_bcast;
    /* a step with thousands of strips (the wide fronts of a large problem) is a throughput problem: POTRF launch, then the
     * one-wave-per-strip TRSM with twelve strips per workgroup, instead of the fused launch's latency design */
    int thru = 0;
    if (fuse && opts->trsm_wt_min > 0) { /* (distributed top levels too: the owner's strips) */
      int64_t est = 0; int fits = 1;
      for (int q = 0; q < nh; q++) {
        const int s = p->tree[hs[q]], n = p->sep_size[s], bw = pivot_block_width(opts, n), c0 = st * bw;
        if (c0 >= n) continue;
        const int nb = n - c0 < bw ? n - c0 : bw;
        if (nb > CHOL_TRSM_WT_MAXN) fits = 0;
        est += (p->panel_rows[s] - c0 - nb + CHOL_TRSM_ROWS - 1) / CHOL_TRSM_ROWS;
      }
      thru = fits && est >= opts->trsm_wt_min;
    }
    for (int q = 0; q < nh; q++) {
      const int h = hs[q], s = p->tree[h], n = p->sep_size[s], ld = p->panel_ld[s];
      const int bw = pivot_block_width(opts, n);
      const int c0 = st * bw;
      if (c0 >= n) continue;
      const int nb = n - c0 < bw ? n - c0 : bw;
      const int64_t diag = p->panel_off[s] + c0 + (int64_t)c0 * ld;          /* element (c0, c0) of the pivot */
      const int64_t dinv = p->dinv_off[s] + (int64_t)(c0 / CHOL_NB) * CHOL_NB * CHOL_NB;
      const int64_t colbase = (int64_t)c0 * ld;                              /* column c0 of the panel */
      const int below = n - c0 - nb;                                         /* pivot rows under the diagonal block */
      row_run *runs; const int nr_ = ancestor_runs(p, h, opts->merge_targets, snap, first, count, &runs);
      const int mine = !dist || dist_owner(p, s, st, world) == rank;
      /* a leaf: L(i, k) = 0 for i - k > band inside the diagonal block, and an ancestor row is zero in front of its first entry of A */
      int band = n, *run_first = NULL;
      int nr = nr_;
      if (rfirst && rfirst[s]) {
        band = 0;
        for (int r = 0; r < n; r++) if (r - rfirst[s][r] > band) band = r - rfirst[s][r];
        /* the ancestor row runs in pieces of 64 rows (a macro tile's height), each with the first entry of ITS rows: a run is live for a column block as a
         * whole otherwise, though most of a face's rows start late */
        const int ph = fine_pieces ? 64 : 1 << 30; /* piece height */
        int np_ = 0;
        for (int r = 0; r < nr_; r++) np_ += fine_pieces ? (runs[r].m + 63) / 64 : 1;
        row_run *pieces = malloc((size_t)(np_ > 0 ? np_ : 1) * sizeof(row_run));
        run_first = malloc((size_t)(np_ > 0 ? np_ : 1) * sizeof(int));
        np_ = 0;
        for (int r = 0; r < nr_; r++)
          for (int r0 = 0; r0 < runs[r].m; r0 += ph) {
            row_run pc = { runs[r].off + r0, runs[r].m - r0 < ph ? runs[r].m - r0 : ph, runs[r].pos + r0 };
            run_first[np_] = rows_first(rfirst[s], (int)((pc.off - p->panel_off[s]) % ld), pc.m, n);
            pieces[np_++] = pc;
          }
        free(runs); runs = pieces; nr = np_;
      }
#define RUN_LIVE(r_, col_end_) (!run_first || run_first[r_] < (col_end_)) /* the run has an entry in front of column col_end_ */
      if (dist) { /* the column block travels from its owner to every rank once it is factored and solved */
        if (w->n_bcast == B->cap_b) { B->cap_b = B->cap_b ? 2 * B->cap_b : 16; w->bcast = realloc(w->bcast, B->cap_b * sizeof(chol_bcast)); }
        chol_bcast bc = { p->panel_off[s] + colbase, (int64_t)nb * ld, dist_owner(p, s, st, world), h };
        w->bcast[w->n_bcast++] = bc;
        B->tgt_sep = s;
      }
      if (mine) {
        chol_potrf_desc pd = { diag, dinv, nb, ld, s, c0, 0, 0, { 0 } };
        push_potrf(B, pd);
        const int flag = w->n_potrf - 1 - p0; /* this block's POTRF descriptor within the step */
        if (below > 0) push_trsm_run(B, diag, dinv, p->panel_off[s] + (c0 + nb) + colbase, nb, ld, below < band ? below : band, flag);
        for (int r = 0; r < nr; r++) if (RUN_LIVE(r, c0 + nb)) push_trsm_run(B, diag, dinv, runs[r].off + colbase, nb, ld, runs[r].m, flag);
        if (thru) pad_trsm_group(B, t0, CHOL_TRSM_WT_GROUP, diag, dinv, diag, nb, ld, flag);
        else if (fuse) pad_trsm_group(B, t0, 3, diag, dinv, diag, nb, ld, flag);
        else if (opts->trsm_group > 0) pad_trsm_group(B, t0, opts->trsm_group, diag, dinv, diag, nb, ld, flag);
        else if (nb <= CHOL_TRSM_W_MAXN) pad_trsm_group(B, t0, 4, diag, dinv, diag, nb, ld, flag);
      }
      if (below > 0) {
        /* Trailing update, in SUPER-BLOCKS of `super` column blocks: after a column block only the remaining columns of its
         * own super-block receive its rank-nb update (they are factored next); the columns beyond wait for the end of the
         * super-block and receive ONE update of rank (columns of the super-block) -- the same sums, but the bulk of the
         * flops (the trailing matrix of a wide front) runs at K = super * nb instead of nb, where the macro-tile kernel is
         * nearer its MFMA rate (35 TF/s at K = 144, 44 at K = 512, fp64).  Pivots of up to `super` blocks are unchanged. */
        const int G = opts->super_blocks > 1 ? opts->super_blocks : 1;
        const int sb0 = (st / G) * G;                                           /* first block of this super-block */
        const int cs0 = sb0 * bw;                                               /* its first column */
        const int cse = (sb0 + G) * bw < n ? (sb0 + G) * bw : n;                /* one past its last column */
        const int last_in_sb = (st % G == G - 1) || c0 + nb >= n || c0 + nb >= cse;
        /* (1) narrow: columns [c0+nb, cse) of the rows below, K = nb */
        const int ncol = cse - (c0 + nb);
        const int ncol_e = ncol < band ? ncol : band;                           /* (a leaf: the columns the block's band reaches) */
        B->tgt_col0 = c0 + nb;
        if (ncol > 0) {
          const int64_t x_piv = p->panel_off[s] + (c0 + nb) + colbase;         /* solved pivot rows under the block, k = nb */
          chol_upd_src sp = { x_piv, x_piv, ld, ld, nb, 0, 0, 0 };
          const int sidx = push_src(B, sp);
          push_tasks(B, p->panel_off[s] + (c0 + nb) + (int64_t)(c0 + nb) * ld, ld, ncol_e, ncol_e, 1, sidx, sidx + 1); /* rows inside the super-block: lower triangle */
          const int beyond = below - ncol < band - ncol ? below - ncol : band - ncol; /* pivot rows beyond the super-block (inside the band) x its remaining columns */
          if (beyond > 0) {
            chol_upd_src sq = { x_piv + ncol, x_piv, ld, ld, nb, 0, 0, 0 };
            const int si = push_src(B, sq);
            push_tasks(B, p->panel_off[s] + cse + (int64_t)(c0 + nb) * ld, ld, beyond, ncol_e, 0, si, si + 1);
          }
          for (int r = 0; r < nr; r++) {
            if (!RUN_LIVE(r, c0 + nb)) continue;
            chol_upd_src sa = { runs[r].off + colbase, x_piv, ld, ld, nb, 0, 0, 0 };
            const int si = push_src(B, sa);
            push_tasks(B, runs[r].off + (int64_t)(c0 + nb) * ld, ld, runs[r].m, ncol_e, 0, si, si + 1);
          }
        }
        /* (2) wide, at the end of the super-block: columns [cse, n), K = cse - cs0 */
        B->tgt_col0 = cse;
        if (last_in_sb && cse < n) {
          /* (a leaf: only the last `band` columns of the super-block reach the columns beyond it, and only `band` of those) */
          const int bk = (band + 15) / 16 * 16;
          const int K = cse - cs0 < bk ? cse - cs0 : bk, ce0 = cse - K, rest = n - cse < band ? n - cse : band;
          const int64_t x_sb = p->panel_off[s] + cse + (int64_t)ce0 * ld;      /* pivot rows beyond the super-block, its columns */
          chol_upd_src sp = { x_sb, x_sb, ld, ld, K, 0, 0, 0 };
          const int sidx = push_src(B, sp);
          push_tasks(B, p->panel_off[s] + cse + (int64_t)cse * ld, ld, rest, rest, 1, sidx, sidx + 1);
          for (int r = 0; r < nr; r++) {
            if (!RUN_LIVE(r, cse)) continue;
            int cr0 = ce0; /* (a leaf: the piece's rows are zero in front of their first entry) */
            if (run_first && (run_first[r] & ~15) > cr0) cr0 = run_first[r] & ~15;
            chol_upd_src sa = { runs[r].off + (int64_t)cr0 * ld, x_sb + (int64_t)(cr0 - ce0) * ld, ld, ld, cse - cr0, 0, 0, 0 };
            const int si = push_src(B, sa);
            push_tasks(B, runs[r].off + (int64_t)cse * ld, ld, runs[r].m, rest, 0, si, si + 1);
          }
        }
      }
#undef RUN_LIVE
      free(run_first);
      free(runs);
    }
    flush_targets(B);
    if (dist) { /* owner's POTRF + TRSM, the broadcast of the step's column blocks (collective: the same list on every rank),
                 * then this rank's share of the trailing update */
      if (thru) {
        push_phase(B, 0, p0, w->n_potrf - p0);
        push_phase(B, 7, t0, w->n_trsm - t0);
      } else if (fuse) {
        if (w->n_phase == B->cap_ph) { B->cap_ph = B->cap_ph ? 2 * B->cap_ph : 16; w->phase = realloc(w->phase, B->cap_ph * sizeof(chol_phase)); }
        chol_phase ph = { 5, p0, w->n_potrf - p0, t0, w->n_trsm - t0, k0, 0 };
        if (ph.n > 0) w->phase[w->n_phase++] = ph;
      } else {
        push_phase(B, 0, p0, w->n_potrf - p0);
        int wide = 0;
        for (int i = t0; i < w->n_trsm; i++) if (w->trsm[i].n > CHOL_TRSM_W_MAXN) wide = 1;
        push_phase(B, opts->trsm_group > 0 ? 7 : wide ? 1 : 4, t0, w->n_trsm - t0); /* (trsm_group: the fp32 schedule's throughput TRSM) */
      }
      push_phase_f(B, 6, b0, w->n_bcast - b0, 1);
      push_phase(B, 2, k0, w->n_task - k0);
      push_phase(B, 3, km0, w->n_task_mt - km0);
      continue;
    }
    if (thru) {
      push_phase(B, 0, p0, w->n_potrf - p0);
      push_phase(B, 7, t0, w->n_trsm - t0);
      push_phase(B, 2, k0, w->n_task - k0);
      fused_last = -1;
    } else if (fuse) { /* one launch: the strips follow their pivot's POTRF column by column, the 16x16 update tasks of the
                 * step (trailing columns of a split pivot) wait for the strips inside the same launch */
      if (w->n_phase == B->cap_ph) { B->cap_ph = B->cap_ph ? 2 * B->cap_ph : 16; w->phase = realloc(w->phase, B->cap_ph * sizeof(chol_phase)); }
      const int ride = w->n_task - k0 <= opts->fuse_update_max;
      chol_phase ph = { 5, p0, w->n_potrf - p0, t0, w->n_trsm - t0, k0, ride ? w->n_task - k0 : 0 };
      fused_last = -1;
      if (ph.n > 0) { fused_last = w->n_phase; w->phase[w->n_phase++] = ph; }
      if (ph.n <= 0 || !ride) { if (w->n_task > k0) fused_last = -1; push_phase(B, 2, k0, w->n_task - k0); }
    } else {
      push_phase(B, 0, p0, w->n_potrf - p0);
      /* strips whose pivot block is narrow enough take the one-wave-per-strip kernel */
      int wide = 0;
      for (int i = t0; i < w->n_trsm; i++) if (w->trsm[i].n > CHOL_TRSM_W_MAXN) wide = 1;
      push_phase(B, opts->trsm_group > 0 ? 7 : wide ? 1 : 4, t0, w->n_trsm - t0);
      push_phase(B, 2, k0, w->n_task - k0);
    }
    push_phase(B, 3, km0, w->n_task_mt - km0);
  }
  /* extend-add of the level: tuples in program order (par bottom-up, gp from par to the root, tiles i, j),
   * grouped by target tile */
  int cap_u = 256, ntu = 0;
  upd_tuple *tu = malloc(cap_u * sizeof(upd_tuple));
  int64_t seq = 0;
  for (int q = 0; q < nh; q++) {
    const int h = hs[q], s = p->tree[h], n = p->sep_size[s];
    for (int hp = h / 2; hp >= 1; hp /= 2) {
      int par = p->tree[hp];
      int bb = BIDX(p, par, s);
      const chol_block *Bb = &p->blk[bb];
      for (int hg = hp; hg >= 1; hg /= 2) {
        int gp = p->tree[hg];
        int ba = BIDX(p, gp, s), bc = BIDX(p, gp, par);
        const chol_block *Ba = &p->blk[ba], *Bc = &p->blk[bc];
        for (int i = 0; i < count[ba]; i++) {
          const cholamd_filled *fa = &snap[first[ba] + i];
          for (int j = 0; j < count[bb]; j++) {
            const cholamd_filled *fb_ = &snap[first[bb] + j];
            if (gp == par && fb_->cluster > fa->cluster) continue; /* col > row skipped, blas.rg:396-431 */
            if (ntu == cap_u) { cap_u *= 2; tu = realloc(tu, cap_u * sizeof(upd_tuple)); }
            upd_tuple *u = &tu[ntu++];
            /* the C tile rectangle = rows of the A tile x rows of the B tile */
            int crow = fa->lo_x - Bc->lo_x, ccol = fb_->lo_x - Bc->lo_y;
            u->key = ((int64_t)bc << 40) | ((int64_t)crow << 20) | (int64_t)ccol;
            u->seq = seq++;
            u->c_off = chol_block_row(Bc, crow) + (int64_t)ccol * Bc->ld; u->ldc = Bc->ld;
            u->a_off = chol_block_row(Ba, fa->lo_x - Ba->lo_x); u->lda = Ba->ld;
            u->b_off = chol_block_row(Bb, fb_->lo_x - Bb->lo_x); u->ldb = Bb->ld;
            u->m = fa->hi_x - fa->lo_x + 1; u->n = fb_->hi_x - fb_->lo_x + 1; u->k = n;
            u->syrk = (gp == par && fb_->cluster == fa->cluster);
            u->bc = bc; u->crow = crow; u->ccol = ccol; u->src_sep = s;
          }
        }
      }
    }
  }
  qsort(tu, ntu, sizeof(upd_tuple), cmp_tuple);
  {
    const int k0 = w->n_task, km0 = w->n_task_mt;
    if (tuples_are_small(opts, tu, ntu)) emit_cell_tasks(B, p, tu, ntu);
    else {
      /* One target per cluster-tile pair of the reference would leave the 64x64 macro tiles a quarter full where the cluster tiles are
       * 32 rows (the generated problems: 54 % of the macro-tile work of 60^3 was 32 x 32 targets).  Targets of one block that are
       * neighbours IN STORAGE -- in the target and, with the same shift, in every source (row compaction makes the kept tiles of a
       * block consecutive) -- and receive the same sources in the same order are merged, along the columns first, then along the rows:
       * every element keeps its sources and their order (bit-identical sums), the tiles fill up.  SYRK targets stay alone. */
      int ng = 0;
      tgt_group *G = malloc((size_t)(ntu > 0 ? ntu : 1) * sizeof(tgt_group));
      for (int i = 0; i < ntu;) {
        int e = i + 1;
        while (e < ntu && tu[e].key == tu[i].key) e++;
        tgt_group g = { i, e, tu[i].bc, tu[i].crow, tu[i].ccol, tu[i].m, tu[i].n, tu[i].syrk, tu[i].c_off, tu[i].ldc, 0 };
        G[ng++] = g;
        i = e;
      }
      if (opts->merge_targets) {
        /* columns: the groups are sorted by (block, first row, first column) */
        for (int a = 0; a < ng;) {
          int b = a + 1;
          while (b < ng && !G[a].syrk && !G[b].syrk && G[b].bc == G[a].bc && G[b].crow == G[a].crow && G[b].m == G[a].m && G[b].ccol == G[a].ccol + G[a].n &&
                 same_sources(tu, &G[a], &G[b], 0, G[a].n)) { G[a].n += G[b].n; G[b].dead = 1; b++; }
          a = b;
        }
        int nl = 0;
        for (int a = 0; a < ng; a++) if (!G[a].dead) G[nl++] = G[a];
        ng = nl;
        /* rows: by (block, first column, columns, first row) */
        qsort(G, ng, sizeof(tgt_group), cmp_group_rows);
        for (int a = 0; a < ng;) {
          int b = a + 1;
          while (b < ng && !G[a].syrk && !G[b].syrk && G[b].bc == G[a].bc && G[b].ccol == G[a].ccol && G[b].n == G[a].n && G[b].c_off == G[a].c_off + G[a].m &&
                 same_sources(tu, &G[a], &G[b], G[a].m, 0)) { G[a].m += G[b].m; G[b].dead = 1; b++; }
          a = b;
        }
        nl = 0;
        for (int a = 0; a < ng; a++) if (!G[a].dead) G[nl++] = G[a];
        ng = nl;
        qsort(G, ng, sizeof(tgt_group), cmp_group_first);
      }
      if (rfirst && !dist && !getenv("CHOLAMD_NO_LEAF_ORDER")) {
        /* leaf level: every macro tile streams 2 x 64 x K operand entries out of its source leaf's panel, and in target order the tiles one leaf feeds are
         * spread over the whole list -- 77 GB fetched for 3 GB of leaf panels at 100^3, the launch at 6.3 TB/s.  In source order the tiles of a leaf run
         * on one XCD at about the same time and share its panel through the L2 (targets are independent: any order gives the same sums) */
        qsort_r(G, ng, sizeof(tgt_group), cmp_group_source, tu);
      }
      for (int a = 0; a < ng; a++) {
        if (rfirst && !dist && !G[a].syrk && rfirst[tu[G[a].i].src_sep]) {
          /* sources out of LEAF panels: the target in 64 x 64 pieces (the macro tiles it would be cut into anyway, in the same 8 x 8 block order), every
           * piece with its own sources, each starting at the first column both of the piece's row sets have entries from */
          for (int rb = 0; rb < G[a].m; rb += 64 * MT_ORDER)
          for (int cb = 0; cb < G[a].n; cb += 64 * MT_ORDER)
          for (int r0 = rb; r0 < G[a].m && r0 < rb + 64 * MT_ORDER; r0 += 64)
          for (int c0 = cb; c0 < G[a].n && c0 < cb + 64 * MT_ORDER; c0 += 64) {
            const int mm = G[a].m - r0 < 64 ? G[a].m - r0 : 64, nn = G[a].n - c0 < 64 ? G[a].n - c0 : 64;
            const int sb = w->n_src;
            for (int q = G[a].i; q < G[a].e; q++) {
              chol_upd_src sd = { tu[q].a_off + r0, tu[q].b_off + c0, tu[q].lda, tu[q].ldb, tu[q].k, 0, 0, 0 };
              const int s_ = tu[q].src_sep;
              const int fa_ = rows_first(rfirst[s_], (int)((sd.a_off - p->panel_off[s_]) % sd.lda), mm, sd.k);
              const int fb_ = rows_first(rfirst[s_], (int)((sd.b_off - p->panel_off[s_]) % sd.ldb), nn, sd.k);
              int k0_ = (fa_ > fb_ ? fa_ : fb_) & ~15;
              if (k0_ > ((sd.k - 1) & ~15)) k0_ = (sd.k - 1) & ~15; /* (at least the last sixteen columns: no empty source) */
              if (k0_ > 0) { sd.a_off += (int64_t)k0_ * sd.lda; sd.b_off += (int64_t)k0_ * sd.ldb; sd.k -= k0_; }
              push_src(B, sd);
            }
            push_tasks(B, G[a].c_off + r0 + (int64_t)c0 * G[a].ldc, G[a].ldc, mm, nn, 0, sb, w->n_src);
          }
          continue;
        }
        const int sb = w->n_src;
        for (int q = G[a].i; q < G[a].e; q++) {
          chol_upd_src sd = { tu[q].a_off, tu[q].b_off, tu[q].lda, tu[q].ldb, tu[q].k, 0, 0, 0 };
          if (rfirst && rfirst[tu[q].src_sep]) { /* a leaf's panel: the source starts at the first column both row sets have entries from */
            const int s_ = tu[q].src_sep;
            const int fa_ = rows_first(rfirst[s_], (int)((tu[q].a_off - p->panel_off[s_]) % tu[q].lda), G[a].m, tu[q].k);
            const int fb_ = rows_first(rfirst[s_], (int)((tu[q].b_off - p->panel_off[s_]) % tu[q].ldb), G[a].n, tu[q].k);
            int k0_ = (fa_ > fb_ ? fa_ : fb_) & ~15;
            if (k0_ > ((tu[q].k - 1) & ~15)) k0_ = (tu[q].k - 1) & ~15; /* (at least the last sixteen columns: no empty source) */
            if (k0_ > 0) { sd.a_off += (int64_t)k0_ * sd.lda; sd.b_off += (int64_t)k0_ * sd.ldb; sd.k -= k0_; }
          }
          push_src(B, sd);
        }
        if (dist) { B->tgt_sep = p->blk[G[a].bc].c; B->tgt_col0 = G[a].ccol; }
        push_tasks(B, G[a].c_off, G[a].ldc, G[a].m, G[a].n, G[a].syrk, sb, w->n_src);
      }
      free(G);
    }
    flush_targets(B);
    if (dist) push_phase(B, 2, k0, w->n_task - k0); else
    /* the 16x16 tasks of the extend-add ride in the level's last fused launch when that one carries no tasks of
     * its own and nothing was launched after it */
    if (fused_last >= 0 && fused_last == w->n_phase - 1 && w->phase[fused_last].n3 == 0 && w->n_task - k0 <= opts->fuse_update_max) {
      w->phase[fused_last].first3 = k0;
      w->phase[fused_last].n3 = w->n_task - k0;
    } else push_phase(B, 2, k0, w->n_task - k0);
    push_phase(B, 3, km0, w->n_task_mt - km0);
  }
  if (rfirst) { for (int s = 1; s <= p->nsep; s++) free(rfirst[s]); free(rfirst); }
  free(tu); free(first); free(count); free(hs); free(B->pend); B->pend = NULL; B->cap_pend = 0;
  return 0;
}


/* ---------------------------------------------------------------------------------------- */
/* The whole factorisation as ONE program launch (single GPU, small problems): k_program.     */
/*                                                                                            */
/* Same work as the per-level lists -- the same pivot blocks, strips and, per output element, */
/* the same sources in the same order -- as a queue of jobs for resident workgroups, with the  */
/* kernel boundaries replaced by counters:                                                     */
/*   prog[pb]     columns a pivot block's POTRF job has published (its strips follow it)       */
/*   strips[pb]   TRSM jobs of the pivot block that have finished                              */
/*   upd[s]       update jobs into panel s that have finished; updd[s]: into its diagonal block */
/*   chan + e     strips of a follow channel that have stored column tile e                    */
/* Queue order per tree level (bottom-up) and column-block step: POTRF jobs, TRSM jobs (the     */
/* strips a follower waits for first), the POTRF job of the pivot's NEXT column block (it       */
/* follows the strips of the rows it owns), the trailing-update jobs; after the last step the    */
/* POTRF jobs of the PARENTS (they follow their children's strips of the parent rows), then the  */
/* extend-add jobs of the level.  A job waits for counters raised by jobs ahead of it in the     */
/* queue, except a follower, which is queued ahead of some of the strips it follows; the host     */
/* self-check (chol_program_check) simulates the queue with a bounded number of resident          */
/* workgroups to show that this cannot dead-lock.                                                 */
/* ---------------------------------------------------------------------------------------- */
#ifndef PROG_JOB_TASKS
#define PROG_JOB_TASKS 12    /* light 16x16 update tasks per update job (one per wave) */
#endif
#ifndef PROG_HEAVY_STEPS
#define PROG_HEAVY_STEPS 48   /* MFMA k-steps (of 4 columns) from which a task is split over four waves */
#endif
#define PROG_MAX_TASKS 24000 /* beyond this the per-level launches (macro tiles) are the better schedule */
typedef struct { int potrf, c_prog, c_strips, n_groups, c0, nb, emitted; int ch_below, ns_below; int ch_par, ns_par; int64_t par_off;
  int ch_rest, ns_rest; /* chunked block (stage_chunk): channel of every strip that is not a followed one */
} pblock;
typedef struct { chol_program *pg; int cap_j, cap_w, cap_e, cap_c;
  int *c_updp, *cnt_updp; /* per separator: update jobs into the block (parent, separator) -- what the followed strips of the parent's rows wait for */
} pbuild;
static int new_ctr(pbuild *P, int total)
{
  chol_program *g = P->pg;
  if (g->n_ctr == P->cap_c) { P->cap_c = P->cap_c ? 2 * P->cap_c : 256; g->ctr_total = realloc(g->ctr_total, P->cap_c * sizeof(int)); }
  g->ctr_total[g->n_ctr] = total;
  return g->n_ctr++;
}
static void add_wait(pbuild *P, int ctr, int value)
{
  chol_program *g = P->pg;
  if (value <= 0) return;
  if (g->n_wait == P->cap_w) { P->cap_w = P->cap_w ? 2 * P->cap_w : 1024; g->wait = realloc(g->wait, P->cap_w * sizeof(chol_wait)); }
  chol_wait wt = { ctr, value };
  g->wait[g->n_wait++] = wt;
}
static int add_ext(pbuild *P, chol_ext e)
{
  chol_program *g = P->pg;
  if (g->n_ext == P->cap_e) { P->cap_e = P->cap_e ? 2 * P->cap_e : 64; g->ext = realloc(g->ext, P->cap_e * sizeof(chol_ext)); }
  g->ext[g->n_ext] = e;
  return g->n_ext++;
}
/* the column tiles of one followed source block (k columns from `off`, counters chan + e) with the expected arrival position key0 + e */
typedef struct { chol_ext x; int key, seq; } ext_item;
static int cmp_ext_item(const void *a, const void *b)
{
  const ext_item *x = a, *y = b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->seq < y->seq ? -1 : (x->seq > y->seq);
}
static void push_ext_items(ext_item **items, int *n, int *cap, int64_t off, int ld, int k, int chan, int nstrip, int key0, int e_from)
{
  for (int e = e_from; e * CHOL_NB < k; e++) {
    if (*n == *cap) { *cap = *cap ? 2 * *cap : 64; *items = realloc(*items, *cap * sizeof(ext_item)); }
    ext_item *it = &(*items)[(*n)];
    it->x.off = off + (int64_t)e * CHOL_NB * ld; it->x.ld = ld; it->x.ncol = k - e * CHOL_NB < CHOL_NB ? k - e * CHOL_NB : CHOL_NB;
    it->x.ctr = chan + e; it->x.need = nstrip;
    it->key = key0 + e; it->seq = *n;
    (*n)++;
  }
}
/* a job whose waits are the entries added since wait_first */
static chol_job *add_job(pbuild *P, int kind, int first, int n, int wait_first)
{
  chol_program *g = P->pg;
  if (g->n_job == P->cap_j) { P->cap_j = P->cap_j ? 2 * P->cap_j : 1024; g->job = realloc(g->job, P->cap_j * sizeof(chol_job)); }
  chol_job j = { kind, first, n, wait_first, g->n_wait - wait_first, { -1, -1 }, 0, g->n_ext, 0, 0, g->n_wait - wait_first };
  g->job[g->n_job] = j;
  return &g->job[g->n_job++];
}
void chol_program_free(chol_program *g)
{
  free(g->job); free(g->wait); free(g->ext); free(g->ctr_total);
  memset(g, 0, sizeof *g);
}
/* strips of the filled rows of block (anc, s) that lie in rows [lo, hi) of anc (relative to the block), for the column block
 * at column c0: appended to the TRSM list with `chan`; returns the number of strips */
static int push_block_rows(builder *B, const plan_t *p, int anc, int s, int lo, int hi, const cholamd_filled *snap, const int64_t *first, const int *count,
                           int64_t diag, int64_t dinv, int nb, int ld, int64_t colbase, int flag, int chan)
{
  chol_level_work *w = B->w;
  const int b = BIDX(p, anc, s);
  const chol_block *Bk = &p->blk[b];
  const int t_before = w->n_trsm;
  int run_lo = -1, run_hi = -1; /* current merged run, rows relative to the block */
  for (int q = 0; q <= count[b]; q++) {
    int a0 = -1, a1 = -1;
    if (q < count[b]) {
      const cholamd_filled *f = &snap[first[b] + q];
      a0 = f->lo_x - Bk->lo_x; a1 = f->hi_x - Bk->lo_x + 1;
      if (a0 < lo) a0 = lo;
      if (a1 > hi) a1 = hi;
      if (a0 >= a1) continue;
      if (run_hi == a0) { run_hi = a1; continue; }
    }
    if (run_lo >= 0) {
      push_trsm_run(B, diag, dinv, chol_block_row(Bk, run_lo) + colbase, nb, ld, run_hi - run_lo, flag);
    }
    run_lo = a0; run_hi = a1;
  }
  for (int i = t_before; i < w->n_trsm; i++) w->trsm[i].chan = chan;
  return w->n_trsm - t_before;
}
/* TRSM jobs over the strips [t0, n_trsm): groups of three, never mixing channels */
static int emit_trsm_jobs(pbuild *P, builder *B, int t0, int c_upd, int need_upd, int c_strips,
                          int ch_par, int c_updp, int need_updp, int ch_below, int c_updd, int need_updd)
{ /* a job waits for the update jobs into the rows its strips solve: the followed strips of the parent's rows (channel ch_par) for
   * those into the block (parent, separator), the followed strips of the pivot's next column block (ch_below) for those into the
   * diagonal block, every other job for all the update jobs into the panel (ch_par / ch_below < 0: that for every job) */
  chol_level_work *w = B->w;
  int njobs = 0;
  for (int i = t0; i < w->n_trsm;) {
    int e = i + 1;
    while (e < w->n_trsm && e - i < 3 && w->trsm[e].chan == w->trsm[i].chan) e++;
    const int wf = P->pg->n_wait;
    const int ch = w->trsm[i].chan;
    if (ch >= 0 && ch == ch_par && c_updp >= 0) add_wait(P, c_updp, need_updp);
    else if (ch >= 0 && ch == ch_below && ch_par >= -1 && c_updp >= 0) add_wait(P, c_updd, need_updd);
    else add_wait(P, c_upd, need_upd);
    chol_job *j = add_job(P, 1, i, e - i, wf);
    j->sig[0] = c_strips; j->sig_add = 1;
    njobs++;
    i = e;
  }
  return njobs;
}
/* update jobs over the tasks [k0, n_task): consecutive tasks of one target block, at most PROG_JOB_TASKS each; every job waits
 * for the strips of its source pivot blocks (src_ctr / src_need pairs) and for the update jobs of earlier phases into its panel */
static void emit_update_jobs(pbuild *P, builder *B, const plan_t *p, int k0, const int *c_upd, const int *c_updd, const int *snap_upd, int *cnt_upd, int *cnt_updd,
                             const int *src_ctr, const int *src_need, int n_src_ctr, const int *stage_sep /* NULL: every wait gates the whole job */)
{ /* stage_sep (extend-add of a level): src_ctr[i] is the strips counter of a pivot block of separator stage_sep[i], the list in the
   * order the blocks are expected to finish.  Those waits are STAGED: the job starts as soon as the earlier update jobs into its
   * panel are through, every task takes its sources in that order and looks at wait i only before the first source that needs it --
   * what is left once the last source pivot has been solved is that pivot's contribution alone, not every descendant's */
  chol_level_work *w = B->w;
  for (int i = k0; i < w->n_task;) {
    /* a task is heavy when its sources add up to PROG_HEAVY_STEPS MFMA k-steps or more (one wave would take them one memory
     * round trip after the other): heavy tasks go three to a job, four waves each; light ones twelve to a job, one wave each */
#define TASK_STEPS(T_) ({ int st_ = 0; for (int q_ = w->task[T_].src_begin; q_ < w->task[T_].src_end; q_++) st_ += (w->src[q_].k + 3) / 4; st_; })
    const int heavy = TASK_STEPS(i) >= PROG_HEAVY_STEPS;
    const int lim = heavy ? 3 : PROG_JOB_TASKS;
    int e = i + 1;
    while (e < w->n_task && e - i < lim && w->task[e].blk == w->task[i].blk && (TASK_STEPS(e) >= PROG_HEAVY_STEPS) == heavy) e++;
#undef TASK_STEPS
    const chol_block *Bc = &p->blk[w->task[i].blk];
    const int wf = P->pg->n_wait;
    if (stage_sep) add_wait(P, c_upd[Bc->c], snap_upd[Bc->c]);
    const int n_pre = P->pg->n_wait - wf;
    for (int q = 0; q < n_src_ctr; q++) add_wait(P, src_ctr[q], src_need[q]);
    if (!stage_sep) add_wait(P, c_upd[Bc->c], snap_upd[Bc->c]);
    chol_job *j = add_job(P, 2, i, e - i, wf);
    j->mode = heavy;
    if (stage_sep) {
      j->n_pre = n_pre;
      for (int t = i; t < e; t++) {
        chol_upd_task *tk = &w->task[t];
        for (int q = tk->src_begin; q < tk->src_end; q++) { /* -(separator) - 1 -> number of staged waits that must hold: through its last pivot
                                                              * block; sub-tiles of one target share their sources: converted once */
          if (w->src[q].stage > 0) continue;
          int st = n_src_ctr;
          if (w->src[q].stage < 0) { /* tag 64 separator + block (block 63: the whole pivot): the last listed block of the separator up to that one */
            const int tag = -w->src[q].stage - 1;
            for (int z = n_src_ctr - 1; z >= 0; z--) if (stage_sep[z] / 64 == tag / 64 && stage_sep[z] % 64 <= tag % 64) { st = z + 1; break; }
          }
          w->src[q].stage = st;
        }
        for (int a = tk->src_begin + 1; a < tk->src_end; a++) { /* stable insertion sort by stage */
          const chol_upd_src x = w->src[a];
          int b = a - 1;
          while (b >= tk->src_begin && w->src[b].stage > x.stage) { w->src[b + 1] = w->src[b]; b--; }
          w->src[b + 1] = x;
        }
      }
    } else for (int t = i; t < e; t++) for (int q = w->task[t].src_begin; q < w->task[t].src_end; q++) w->src[q].stage = 0;
    const int hpc = p->heap_of[Bc->c] / 2;
    const int is_par = P->c_updp && Bc->r != Bc->c && hpc >= 1 && Bc->r == p->tree[hpc];
    j->sig[0] = c_upd[Bc->c]; j->sig[1] = Bc->r == Bc->c ? c_updd[Bc->c] : is_par ? P->c_updp[Bc->c] : -1; j->sig_add = 1;
    cnt_upd[Bc->c]++;
    if (Bc->r == Bc->c) cnt_updd[Bc->c]++;
    if (is_par) P->cnt_updp[Bc->c]++;
    i = e;
  }
}

/* Tile-level skyline of the LEAF pivots.  A leaf's diagonal block receives nothing from below, so its factor stays inside the envelope
 * of A: L(i, j) = 0 for j < first(i), first(i) = the first column of row i of tril(P A P^T) inside the block.  fcol[tile row] = the
 * first 16-column tile that can be non-zero in any of the tile's rows (lapl_3375's 259-column leaf: 56 of its 153 tiles, a band of
 * three below the diagonal).  Returns a malloc'ed array over the separators (NULL for a separator with descendants or no pivot). */
static unsigned char **leaf_skylines(const plan_t *p)
{
  const int ns = p->nsep, L = p->levels;
  unsigned char **sky = calloc(ns + 1, sizeof(unsigned char *));
  int **first = calloc(ns + 1, sizeof(int *));
  for (int h = 1 << (L - 1); h <= ns; h++) {
    const int s = p->tree[h], n = p->sep_size[s];
    if (n <= 0) continue;
    first[s] = malloc(n * sizeof(int));
    for (int r = 0; r < n; r++) first[s][r] = r; /* the diagonal */
  }
  for (int64_t e = 0; e < p->nnz_a; e++) { /* entries of tril(A) by arena offset: the leaf panels' diagonal blocks */
    const int64_t off = p->a_dst[e];
    int lo = 1, hi = ns; /* the panel that holds the offset: panels are laid out by label */
    while (lo < hi) { const int mid = (lo + hi + 1) / 2; if (p->panel_off[mid] <= off) lo = mid; else hi = mid - 1; }
    const int s = lo;
    if (!first[s]) continue;
    const int64_t rel = off - p->panel_off[s];
    const int ld = p->panel_ld[s], row = (int)(rel % ld), col = (int)(rel / ld);
    if (row < p->sep_size[s] && col < first[s][row]) first[s][row] = col;
  }
  for (int s = 1; s <= ns; s++) {
    if (!first[s]) continue;
    const int n = p->sep_size[s], T = (n + CHOL_NB - 1) / CHOL_NB;
    sky[s] = malloc(T);
    for (int i = 0; i < T; i++) {
      int f = n;
      for (int r = i * CHOL_NB; r < n && r < (i + 1) * CHOL_NB; r++) if (first[s][r] < f) f = first[s][r];
      sky[s][i] = (unsigned char)(f / CHOL_NB);
    }
    free(first[s]);
  }
  free(first);
  return sky;
}
static void free_skylines(unsigned char **sky, int ns) { if (!sky) return; for (int s = 1; s <= ns; s++) free(sky[s]); free(sky); }
/* first[s][r] for every row r of a LEAF's panel (diagonal block and the rows into the ancestors): the first column with an entry of tril(P A P^T) --
 * the row of L is zero in front of it (nothing reaches a leaf from below); n for a row without entries.  NULL for other separators. */
static int **leaf_row_first(const plan_t *p)
{
  const int ns = p->nsep, L = p->levels;
  int **first = calloc(ns + 1, sizeof(int *));
  for (int h = 1 << (L - 1); h <= ns; h++) {
    const int s = p->tree[h], n = p->sep_size[s], ld = p->panel_ld[s];
    if (n <= 0) continue;
    first[s] = malloc((size_t)ld * sizeof(int));
    for (int r = 0; r < ld; r++) first[s][r] = r < n ? r : n;
  }
  for (int64_t e = 0; e < p->nnz_a; e++) {
    const int64_t off = p->a_dst[e];
    int lo = 1, hi = ns;
    while (lo < hi) { const int mid = (lo + hi + 1) / 2; if (p->panel_off[mid] <= off) lo = mid; else hi = mid - 1; }
    const int s = lo;
    if (!first[s]) continue;
    const int64_t rel = off - p->panel_off[s];
    const int ld = p->panel_ld[s], row = (int)(rel % ld), col = (int)(rel / ld);
    if (col < first[s][row]) first[s][row] = col;
  }
  return first;
}

/* Early part of a wide follower's sources: dense 16x16 cell tasks  C(I, J) -= sum over the sources of E_I E_J^T  over the lower
 * triangle of the lim x lim target block at c_off, E = rows of the target in a source panel (off = row 0 in the first of k
 * columns).  They become ordinary update jobs on other CUs that wait for the sources' channel counters (the strips publish their
 * column tiles in order: the last column needed stands for all) and for the earlier update jobs into the panel; the follower
 * itself takes only the last column tiles of each source (follow_external) and sees these through its own tiles. */
typedef struct { int64_t off; int ld, k, ctr, need; } early_src; /* ctr: the channel counter of the source's LAST early column tile (tiles before it: ctr - 1, ...) */
#ifndef EARLY_CHUNK
#define EARLY_CHUNK 2 /* column tiles per staged source of an early job (1 / 2 / 3 at follow_tail 3: 193.5 / 194.5 / 197.3 us on lapl_3375) */
#endif
static void emit_early_cells(pbuild *P, builder *B, const plan_t *p, int64_t c_off, int ldc, int lim, int blk, const early_src *es, int nes,
                             const int *c_upd, const int *c_updd, int *snap_upd, int *cnt_upd, int *cnt_updd)
{
  chol_level_work *w = B->w;
  /* the staged waits of these jobs: the sources in chunks of EARLY_CHUNK column tiles, chunk by chunk over the sources (the order
   * in which the strips publish them, near enough); stage = position in this list + 1 */
  int nst = 0, maxch = 0;
  for (int q = 0; q < nes; q++) if (es[q].k > 0) { const int ch = ((es[q].k + CHOL_NB - 1) / CHOL_NB + EARLY_CHUNK - 1) / EARLY_CHUNK; nst += ch; if (ch > maxch) maxch = ch; }
  if (nst == 0) return;
  int *sc = malloc(nst * sizeof(int)), *sn = malloc(nst * sizeof(int)), *sq = malloc(nst * sizeof(int)), *sch = malloc(nst * sizeof(int));
  nst = 0;
  for (int c = 0; c < maxch; c++)
    for (int q = 0; q < nes; q++) {
      if (es[q].k <= 0) continue;
      const int et = (es[q].k + CHOL_NB - 1) / CHOL_NB;
      if (c * EARLY_CHUNK >= et) continue;
      const int last = (c + 1) * EARLY_CHUNK < et ? (c + 1) * EARLY_CHUNK - 1 : et - 1; /* last column tile of the chunk */
      sc[nst] = es[q].ctr - (et - 1) + last; sn[nst] = es[q].need; sq[nst] = q; sch[nst] = c; nst++;
    }
  const int staged = B->o->staged;
  const int k0 = w->n_task, Tl = (lim + CHOL_NB - 1) / CHOL_NB;
  for (int I = 0; I < Tl; I++)
    for (int J = 0; J <= I; J++) {
      const int sb = w->n_src;
      for (int z = 0; z < nst; z++) {
        const early_src *e = &es[sq[z]];
        const int col0 = sch[z] * EARLY_CHUNK * CHOL_NB, kc = e->k - col0 < EARLY_CHUNK * CHOL_NB ? e->k - col0 : EARLY_CHUNK * CHOL_NB;
        chol_upd_src sd = { e->off + CHOL_NB * I + (int64_t)col0 * e->ld, e->off + CHOL_NB * J + (int64_t)col0 * e->ld, e->ld, e->ld, kc, 0, staged ? z + 1 : 0, 0 };
        push_src(B, sd);
      }
      if (w->n_task == B->cap_k) { B->cap_k = B->cap_k ? 2 * B->cap_k : 256; w->task = realloc(w->task, B->cap_k * sizeof(chol_upd_task)); }
      chol_upd_task *t = &w->task[w->n_task++];
      memset(t, 0, sizeof *t);
      t->c_off = c_off + CHOL_NB * I + (int64_t)(CHOL_NB * J) * ldc; t->ldc = ldc;
      t->mv = (short)(lim - CHOL_NB * I < CHOL_NB ? lim - CHOL_NB * I : CHOL_NB);
      t->nv = (short)(lim - CHOL_NB * J < CHOL_NB ? lim - CHOL_NB * J : CHOL_NB);
      t->lower = I == J;
      t->src_begin = sb; t->src_end = w->n_src;
      t->blk = blk;
    }
  const int col_sep = p->blk[blk].c;
  snap_upd[col_sep] = cnt_upd[col_sep];
  emit_update_jobs(P, B, p, k0, c_upd, c_updd, snap_upd, cnt_upd, cnt_updd, sc, sn, nst, staged ? sq : NULL); /* sq: only "staged" (the stages are set) */
  free(sc); free(sn); free(sq); free(sch);
}

/* the options as the program launch sees them: pivots up to CHOL_PROG_SPLIT_MIN columns whole unless split_min was set by the caller */
static chol_sched_opts chol_program_opts(const chol_sched_opts *o)
{
  chol_sched_opts r = *o;
  if (r.split_min == CHOL_SPLIT_MIN && r.split_nb == CHOL_SPLIT_NB) r.split_min = CHOL_PROG_SPLIT_MIN;
  return r;
}
int chol_build_program(const plan_t *p, const chol_sched_opts *opts, chol_level_work *w, chol_program *pg)
{
  memset(w, 0, sizeof *w);
  memset(pg, 0, sizeof *pg);
  w->level = -1;
  chol_sched_opts dflt, prog;
  if (!opts) { chol_sched_opts_default(&dflt); opts = &dflt; }
  prog = chol_program_opts(opts); opts = &prog;
  builder Bd; memset(&Bd, 0, sizeof Bd); Bd.w = w; Bd.o = opts; Bd.force_fine = 1;
  builder *B = &Bd;
  pbuild Pd; memset(&Pd, 0, sizeof Pd); Pd.pg = pg;
  pbuild *P = &Pd;
  const int L = p->levels, ns = p->nsep;
  int rc = 0;
  /* column-block width per separator.  A LEAF whose factor is banded at tile level (skyline: at most four tiles below the diagonal)
   * is factored as ONE block up to CHOL_RR_MAXN columns: its POTRF skips the zero tiles, so one CU keeps up with the pivot chain, and
   * its strips keep one L tile per step (trsm_rr_body, band form) -- no split, no transition between column blocks */
  unsigned char **sky = opts->skyline ? leaf_skylines(p) : calloc(ns + 1, sizeof(unsigned char *));
  int *bw_of = malloc((ns + 1) * sizeof(int)), *band_of = calloc(ns + 1, sizeof(int));
  for (int s = 1; s <= ns; s++) {
    const int n = p->sep_size[s];
    bw_of[s] = pivot_block_width(opts, n);
    if (sky[s] && n > CHOL_FUSE_MAXN && n <= CHOL_RR_MAXN && pivot_blocks(opts, n) > 1) {
      int band = 1;
      for (int i = 0; i * CHOL_NB < n; i++) if (i - sky[s][i] > band) band = i - sky[s][i];
      if (band <= 4) { bw_of[s] = n; band_of[s] = band; }
    }
  }
  B->bw_of = bw_of;
  /* stage width of a separator's extend-add sources: a banded leaf factored as one block is handed on in chunks of stage_chunk column
   * tiles (all its strips publish their column tiles); everything else block by block */
  int *sw_of = malloc((ns + 1) * sizeof(int));
  for (int s = 1; s <= ns; s++) {
    sw_of[s] = bw_of[s];
    if (band_of[s] && opts->staged && opts->stage_chunk > 0 && opts->stage_chunk * CHOL_NB < p->sep_size[s]) sw_of[s] = opts->stage_chunk * CHOL_NB;
  }
  B->sw_of = sw_of;
  /* eligibility: every pivot block fits both fused roles */
  for (int s = 1; s <= ns && !rc; s++) {
    const int bw = bw_of[s];
    if ((bw > CHOL_FUSE_MAXN && !band_of[s]) || bw > CHOL_RR_MAXN) { chol_set_error("program launch: pivot block of %d columns (separator %d) exceeds %d", bw, s, CHOL_FUSE_MAXN); rc = CHOLAMD_ERR_ARG; }
  }
  { /* cheap early refusal of large problems: 16x16 tiles of the trailing updates of split pivots, and of every panel once per
     * tree level below it (extend-add cells) */
    double est = 0;
    for (int s = 1; s <= ns; s++) {
      const int n = p->sep_size[s], rows = p->panel_rows[s];
      if (n <= 0) continue;
      const int bw = bw_of[s], nbk = (n + bw - 1) / bw;
      for (int st = 0; st + 1 < nbk; st++) est += (double)((rows - (st + 1) * bw) / 16 + 1) * ((n - (st + 1) * bw) / 16 + 1);
      est += (double)(rows / 16 + 1) * (n / 16 + 1) * (L - 1 - p->level_of[s]);
    }
    if (est > 2.0 * PROG_MAX_TASKS) { chol_set_error("program launch: about %.0f update tasks", est); rc = CHOLAMD_ERR_ARG; }
  }
  if (rc) { free_skylines(sky, ns); free(bw_of); free(band_of); free(sw_of); return rc; }
  const int follow = opts->follow && opts->cells;
  pg->follow = follow;
  int64_t *first = malloc(p->nblk * sizeof(int64_t));
  int *count = malloc(p->nblk * sizeof(int));
  /* per panel: counters and running counts of update jobs; per separator: its pivot blocks */
  int *c_upd = malloc((ns + 1) * sizeof(int)), *c_updd = malloc((ns + 1) * sizeof(int));
  if (opts->fine_upd && opts->follow && opts->cells) { P->c_updp = malloc((ns + 1) * sizeof(int)); P->cnt_updp = calloc(ns + 1, sizeof(int)); }
  int *cnt_upd = calloc(ns + 1, sizeof(int)), *cnt_updd = calloc(ns + 1, sizeof(int)), *snap_upd = calloc(ns + 1, sizeof(int));
  int *nblk_of = calloc(ns + 1, sizeof(int));
  pblock **pb = calloc(ns + 1, sizeof(pblock *));
  int *follow_lim = calloc(p->nblk, sizeof(int));
  for (int s = 1; s <= ns; s++) {
    c_upd[s] = new_ctr(P, 0); c_updd[s] = new_ctr(P, 0);
    if (P->c_updp) P->c_updp[s] = new_ctr(P, 0);
    const int n = p->sep_size[s];
    const int bw = bw_of[s], nbk = n > 0 ? (n + bw - 1) / bw : 0;
    nblk_of[s] = nbk;
    pb[s] = calloc(nbk > 0 ? nbk : 1, sizeof(pblock));
    for (int st = 0; st < nbk; st++) {
      pblock *b = &pb[s][st];
      b->c0 = st * bw; b->nb = n - b->c0 < bw ? n - b->c0 : bw;
      b->potrf = -1; b->ch_below = b->ch_par = b->ch_rest = -1;
      b->c_prog = new_ctr(P, (b->nb + CHOL_NB - 1) / CHOL_NB);
      b->c_strips = new_ctr(P, 0);
    }
  }
  /* expected position (in 16-column steps) at which a separator's pivot chain starts: the longest chain below it; leaves start at 0.
   * A follower consumes its children's column tiles in the order of these positions */
  int *est_start = calloc(ns + 1, sizeof(int));
  for (int h = ns; h >= 1; h--) {
    int st0 = 0;
    for (int c = 2 * h; c <= 2 * h + 1 && c <= ns; c++) {
      const int k = p->tree[c];
      const int end = est_start[k] + (p->sep_size[k] + CHOL_NB - 1) / CHOL_NB;
      if (end > st0) st0 = end;
    }
    est_start[p->tree[h]] = st0;
  }
  /* POTRF job of block st of separator s; exts were added just before (ext_first .. n_ext) */
  #define EMIT_POTRF(s_, st_, ext_first_)                                                                        \
    do {                                                                                                          \
      pblock *b_ = &pb[s_][st_];                                                                                  \
      const int ld_ = p->panel_ld[s_];                                                                            \
      const int64_t diag_ = p->panel_off[s_] + b_->c0 + (int64_t)b_->c0 * ld_;                                    \
      const int64_t dinv_ = p->dinv_off[s_] + (int64_t)(b_->c0 / CHOL_NB) * CHOL_NB * CHOL_NB;                    \
      chol_potrf_desc pd_ = { diag_, dinv_, b_->nb, ld_, s_, b_->c0, b_->c_prog, 0, { 0 } };                      \
      if (sky[s_]) /* the block's skyline, relative to its own first tile column */                              \
        for (int i_ = 0; i_ * CHOL_NB < b_->nb && i_ < 24; i_++) {                                                \
          const int f_ = sky[s_][b_->c0 / CHOL_NB + i_] - b_->c0 / CHOL_NB;                                       \
          pd_.sky[i_] = (unsigned char)(f_ > 0 ? f_ : 0);                                                         \
        }                                                                                                         \
      push_potrf(B, pd_);                                                                                         \
      b_->potrf = w->n_potrf - 1;                                                                                 \
      const int wf_ = pg->n_wait;                                                                                 \
      add_wait(P, c_updd[s_], cnt_updd[s_]);                                                                      \
      chol_job *j_ = add_job(P, 0, b_->potrf, 1, wf_);                                                            \
      j_->ext_first = (ext_first_); j_->n_ext = pg->n_ext - (ext_first_);                                         \
      b_->emitted = 1;                                                                                            \
    } while (0)

  for (int level = L - 1; level >= 0 && !rc; level--) {
    const int lbl = L - 1 - level;
    index_snapshot(p, lbl, first, count);
    const cholamd_filled *snap = p->snap[lbl];
    const int h0 = 1 << level, h1 = (1 << (level + 1)) - 1;
    int *hs = malloc((h1 - h0 + 1) * sizeof(int)), nh = 0, steps = 0;
    for (int h = h0; h <= h1; h++) {
      const int s = p->tree[h];
      if (count[BIDX(p, s, s)] == 0 || p->sep_size[s] == 0) continue;
      hs[nh++] = h;
      if (nblk_of[s] > steps) steps = nblk_of[s];
    }
    for (int st = 0; st < steps; st++) {
      /* (A) POTRF jobs of this step that no follower placement has emitted yet */
      for (int q = 0; q < nh; q++) {
        const int s = p->tree[hs[q]];
        if (st < nblk_of[s] && !pb[s][st].emitted) EMIT_POTRF(s, st, pg->n_ext);
      }
      /* (B) TRSM jobs: per pivot block the followed rows first (next column block of the pivot, rows of the parent's first
       *     column block), then everything else */
      for (int q = 0; q < nh; q++) {
        const int h = hs[q], s = p->tree[h], n = p->sep_size[s], ld = p->panel_ld[s];
        if (st >= nblk_of[s]) continue;
        pblock *b = &pb[s][st];
        const int c0 = b->c0, nb = b->nb;
        const int64_t diag = p->panel_off[s] + c0 + (int64_t)c0 * ld;
        const int64_t dinv = p->dinv_off[s] + (int64_t)(c0 / CHOL_NB) * CHOL_NB * CHOL_NB;
        const int64_t colbase = (int64_t)c0 * ld;
        const int below = n - c0 - nb;
        const int t0 = w->n_trsm;
        B->cur_band = band_of[s]; /* strips of a banded leaf factored as one block */
        /* rows of the next column block of this pivot */
        int nb1 = 0;
        if (below > 0) {
          nb1 = pb[s][st + 1].nb;
          const int fol = follow && (nb1 + CHOL_NB - 1) / CHOL_NB <= CHOL_FOLLOW_MAXT;
          if (fol) { b->ns_below = (nb1 + CHOL_TRSM_ROWS - 1) / CHOL_TRSM_ROWS; b->ch_below = new_ctr(P, b->ns_below); for (int e = 1; e < (nb + CHOL_NB - 1) / CHOL_NB; e++) new_ctr(P, b->ns_below); }
          push_trsm_run(B, diag, dinv, p->panel_off[s] + (c0 + nb) + colbase, nb, ld, nb1, b->c_prog);
          for (int i = w->n_trsm - (nb1 + CHOL_TRSM_ROWS - 1) / CHOL_TRSM_ROWS; i < w->n_trsm; i++) w->trsm[i].chan = fol ? b->ch_below : -1;
        }
        /* rows of the parent's first column block */
        int par_lim = 0;
        const int hp = h / 2;
        if (hp >= 1 && follow) {
          const int par = p->tree[hp];
          if (nblk_of[par] > 0 && (pb[par][0].nb + CHOL_NB - 1) / CHOL_NB <= CHOL_FOLLOW_MAXT && count[BIDX(p, par, s)] > 0) {
            par_lim = pb[par][0].nb;
            const int tb = w->n_trsm;
            const int nstr = push_block_rows(B, p, par, s, 0, par_lim, snap, first, count, diag, dinv, nb, ld, colbase, b->c_prog, -2);
            if (nstr > 0) {
              b->ns_par = nstr; b->ch_par = new_ctr(P, nstr);
              for (int e = 1; e < (nb + CHOL_NB - 1) / CHOL_NB; e++) new_ctr(P, nstr);
              for (int i = tb; i < w->n_trsm; i++) w->trsm[i].chan = b->ch_par;
              b->par_off = p->blk[BIDX(p, par, s)].off + colbase; /* row 0 of the parent block in column c0 of the panel */
            } else par_lim = 0;
          }
        }
        /* the rest: remaining pivot rows, remaining parent rows, every other ancestor */
        const int t_rest = w->n_trsm;
        if (below - nb1 > 0) push_trsm_run(B, diag, dinv, p->panel_off[s] + (c0 + nb + nb1) + colbase, nb, ld, below - nb1, b->c_prog);
        for (int ha = hp; ha >= 1; ha /= 2) {
          const int anc = p->tree[ha];
          if (BIDX(p, anc, s) < 0) continue;
          push_block_rows(B, p, anc, s, ha == hp ? par_lim : 0, p->sep_size[anc], snap, first, count, diag, dinv, nb, ld, colbase, b->c_prog, -1);
        }
        if (sw_of[s] < bw_of[s] && w->n_trsm > t_rest) { /* chunked: these strips publish their column tiles too */
          b->ns_rest = w->n_trsm - t_rest; b->ch_rest = new_ctr(P, b->ns_rest);
          for (int e = 1; e < (nb + CHOL_NB - 1) / CHOL_NB; e++) new_ctr(P, b->ns_rest);
          for (int i = t_rest; i < w->n_trsm; i++) w->trsm[i].chan = b->ch_rest;
        }
        B->cur_band = 0;
        b->n_groups = emit_trsm_jobs(P, B, t0, c_upd[s], cnt_upd[s], b->c_strips, b->ch_par, P->c_updp ? P->c_updp[s] : -1, P->c_updp ? P->cnt_updp[s] : 0, b->ch_below, c_updd[s], cnt_updd[s]);
        pg->ctr_total[b->c_strips] = b->n_groups;
      }
      /* (C) the next column block of a split pivot follows the strips of its own rows */
      for (int q = 0; q < nh; q++) {
        const int s = p->tree[hs[q]];
        if (st + 1 >= nblk_of[s] || pb[s][st].ch_below < 0) continue;
        const pblock *b = &pb[s][st];
        const int ld = p->panel_ld[s];
        ext_item *items = NULL; int nit = 0, capit = 0;
        const int64_t e_off = p->panel_off[s] + (b->c0 + b->nb) + (int64_t)b->c0 * ld;
        const int nb1 = pb[s][st + 1].nb, nt = (b->nb + CHOL_NB - 1) / CHOL_NB;
        int bt = 0; /* column tiles of this block that reach the next block's diagonal block through early update jobs */
        if (opts->follow_tail_split > 0 && (nb1 + CHOL_NB - 1) / CHOL_NB > CHOL_FOLLOW_ALL_MAXT && nt > opts->follow_tail_split) bt = nt - opts->follow_tail_split;
        if (bt > 0 && sky[s]) { /* a leaf: columns left of the next block's skyline bring nothing into its diagonal block */
          int fmin = 1 << 20;
          for (int i = (b->c0 + b->nb) / CHOL_NB; i * CHOL_NB < b->c0 + b->nb + nb1; i++) if (sky[s][i] < fmin) fmin = sky[s][i];
          if (fmin - b->c0 / CHOL_NB >= bt) bt = 0; /* (the tail items stay: they are cheap, and the follower's list must not be empty) */
        }
        if (bt > 0) {
          const early_src es = { e_off, ld, bt * CHOL_NB, b->ch_below + bt - 1, b->ns_below };
          emit_early_cells(P, B, p, p->panel_off[s] + (b->c0 + b->nb) + (int64_t)(b->c0 + b->nb) * ld, ld, nb1, BIDX(p, s, s), &es, 1, c_upd, c_updd, snap_upd, cnt_upd, cnt_updd);
        }
        const int ef = pg->n_ext;
        push_ext_items(&items, &nit, &capit, e_off, ld, b->nb, b->ch_below, b->ns_below, 0, bt);
        for (int i = 0; i < nit; i++) add_ext(P, items[i].x);
        free(items);
        EMIT_POTRF(s, st + 1, ef);
      }
      /* (D) trailing update of the step: columns right of the block in the pivot rows below it (lower triangle) and in the
       *     ancestor rows; the diagonal block of a following next column block is left out (follow_external does it) */
      {
        memcpy(snap_upd, cnt_upd, (ns + 1) * sizeof(int));
        for (int q = 0; q < nh; q++) {
          const int h = hs[q], s = p->tree[h], n = p->sep_size[s], ld = p->panel_ld[s];
          if (st >= nblk_of[s]) continue;
          const pblock *b = &pb[s][st];
          const int c0 = b->c0, nb = b->nb, below = n - c0 - nb;
          if (below <= 0) continue;
          const int64_t colbase = (int64_t)c0 * ld;
          const int64_t x_piv = p->panel_off[s] + (c0 + nb) + colbase; /* solved pivot rows below the block, k = nb */
          /* fast part: (followed rows of the parent) x (columns of the NEXT column block) -- what that block's followed strips of
           * the parent's rows wait for.  Its operands are channel strips on both sides (the parent-row strips, the strips of the next
           * block's own rows), so these jobs wait for the channels' last column instead of every strip of the block, and are queued
           * ahead of the rest of the step's trailing update */
          const int hpp_ = h / 2, par_ = hpp_ >= 1 ? p->tree[hpp_] : -1, blkp_ = par_ >= 0 ? BIDX(p, par_, s) : -1;
          const int fast = P->c_updp && b->ch_par >= 0 && b->ch_below >= 0 && blkp_ >= 0 && count[blkp_] > 0;
          const int fast_cols = fast ? pb[s][st + 1].nb : 0; /* columns of the parent's followed rows the fast jobs cover */
          if (fast) {
            const chol_block *Bk = &p->blk[blkp_];
            const int plim = pb[par_][0].nb, kf0 = w->n_task, ntl_ = (nb + CHOL_NB - 1) / CHOL_NB;
            B->cur_blk = blkp_;
            int run_lo = -1, run_hi = -1;
            for (int q2 = 0; q2 <= count[blkp_]; q2++) {
              int a0 = -1, a1 = -1;
              if (q2 < count[blkp_]) {
                const cholamd_filled *f = &snap[first[blkp_] + q2];
                a0 = f->lo_x - Bk->lo_x; a1 = f->hi_x - Bk->lo_x + 1;
                if (a1 > plim) a1 = plim;
                if (a0 >= a1) continue;
                if (run_hi == a0) { run_hi = a1; continue; }
              }
              if (run_lo >= 0) { /* sources in chunks of EARLY_CHUNK column tiles, chunk c behind the waits 2 c, 2 c + 1 (both channels) */
                const int sb_ = w->n_src;
                for (int c = 0; c * EARLY_CHUNK < ntl_; c++) {
                  const int col0 = c * EARLY_CHUNK * CHOL_NB, kc = nb - col0 < EARLY_CHUNK * CHOL_NB ? nb - col0 : EARLY_CHUNK * CHOL_NB;
                  chol_upd_src sa = { chol_block_row(Bk, run_lo) + colbase + (int64_t)col0 * ld, x_piv + (int64_t)col0 * ld, ld, ld, kc, 0, opts->staged ? 2 * c + 2 : 0, 0 };
                  push_src(B, sa);
                }
                push_tasks(B, chol_block_row(Bk, run_lo) + (int64_t)(c0 + nb) * ld, ld, run_hi - run_lo, fast_cols, 0, sb_, w->n_src);
              }
              run_lo = a0; run_hi = a1;
            }
            flush_targets(B);
            int scf[2 * 32], snf[2 * 32], nwf = 0;
            for (int c = 0; c * EARLY_CHUNK < ntl_ && nwf < 62; c++) {
              const int last = (c + 1) * EARLY_CHUNK < ntl_ ? (c + 1) * EARLY_CHUNK - 1 : ntl_ - 1;
              scf[nwf] = b->ch_par + last; snf[nwf++] = b->ns_par;
              scf[nwf] = b->ch_below + last; snf[nwf++] = b->ns_below;
            }
            emit_update_jobs(P, B, p, kf0, c_upd, c_updd, snap_upd, cnt_upd, cnt_updd, scf, snf, nwf, opts->staged ? scf : NULL); /* non-NULL: staged (the stages are set) */
          }
          const int k0 = w->n_task;
          B->cur_blk = BIDX(p, s, s);
          if (b->ch_below >= 0) {
            const int nb1 = pb[s][st + 1].nb, r2 = below - nb1;
            if (r2 > 0) {
              const int64_t x2 = x_piv + nb1;
              chol_upd_src s1 = { x2, x_piv, ld, ld, nb, 0, 0, 0 }, s2 = { x2, x2, ld, ld, nb, 0, 0, 0 };
              const int i1 = push_src(B, s1);
              push_tasks(B, p->panel_off[s] + (c0 + nb + nb1) + (int64_t)(c0 + nb) * ld, ld, r2, nb1, 0, i1, i1 + 1);
              const int i2 = push_src(B, s2);
              push_tasks(B, p->panel_off[s] + (c0 + nb + nb1) + (int64_t)(c0 + nb + nb1) * ld, ld, r2, r2, 1, i2, i2 + 1);
            }
          } else {
            chol_upd_src sp = { x_piv, x_piv, ld, ld, nb, 0, 0, 0 };
            const int sidx = push_src(B, sp);
            push_tasks(B, p->panel_off[s] + (c0 + nb) + (int64_t)(c0 + nb) * ld, ld, below, below, 1, sidx, sidx + 1);
          }
          /* ancestor rows.  `blk` of these tasks: the true block for the followed rows of the parent (rows [0, par_lim) of block
           * (parent, s): their update jobs are what the followed strips of the later column blocks wait for, counter updp), a
           * non-parent, non-diagonal block of the panel for every other ancestor row (they raise the panel counter only) */
          const int hpp = h / 2, hgp = h / 4;
          const int par = hpp >= 1 ? p->tree[hpp] : -1;
          const int blk_par = par >= 0 ? BIDX(p, par, s) : -1;
          const int blk_rest = hgp >= 1 && BIDX(p, p->tree[hgp], s) >= 0 ? BIDX(p, p->tree[hgp], s) : blk_par;
          const int fine = P->c_updp && b->ch_par >= 0 && blk_par >= 0 && count[blk_par] > 0;
          const int par_lim = fine ? pb[par][0].nb : 0;
          if (fine) { /* followed parent rows: filled row runs of block (parent, s) within [0, par_lim) */
            const chol_block *Bk = &p->blk[blk_par];
            B->cur_blk = blk_par;
            int run_lo = -1, run_hi = -1;
            for (int q2 = 0; q2 <= count[blk_par]; q2++) {
              int a0 = -1, a1 = -1;
              if (q2 < count[blk_par]) {
                const cholamd_filled *f = &snap[first[blk_par] + q2];
                a0 = f->lo_x - Bk->lo_x; a1 = f->hi_x - Bk->lo_x + 1;
                if (a1 > par_lim) a1 = par_lim;
                if (a0 >= a1) continue;
                if (run_hi == a0) { run_hi = a1; continue; }
              }
              if (run_lo >= 0 && below > fast_cols) { /* the columns beyond the fast part */
                chol_upd_src sa = { chol_block_row(Bk, run_lo) + colbase, x_piv + fast_cols, ld, ld, nb, 0, 0, 0 };
                const int si = push_src(B, sa);
                push_tasks(B, chol_block_row(Bk, run_lo) + (int64_t)(c0 + nb + fast_cols) * ld, ld, run_hi - run_lo, below - fast_cols, 0, si, si + 1);
              }
              run_lo = a0; run_hi = a1;
            }
          }
          B->cur_blk = blk_rest;
          for (int ha = hpp; ha >= 1; ha /= 2) { /* every other ancestor row: the rest of the parent block, the other ancestors */
            const int anc = p->tree[ha], ba = BIDX(p, anc, s);
            if (ba < 0) continue;
            const chol_block *Bk = &p->blk[ba];
            const int lo = (fine && ha == hpp) ? par_lim : 0;
            int run_lo = -1, run_hi = -1;
            for (int q2 = 0; q2 <= count[ba]; q2++) {
              int a0 = -1, a1 = -1;
              if (q2 < count[ba]) {
                const cholamd_filled *f = &snap[first[ba] + q2];
                a0 = f->lo_x - Bk->lo_x; a1 = f->hi_x - Bk->lo_x + 1;
                if (a0 < lo) a0 = lo;
                if (a0 >= a1) continue;
                if (run_hi == a0) { run_hi = a1; continue; }
              }
              if (run_lo >= 0) {
                chol_upd_src sa = { chol_block_row(Bk, run_lo) + colbase, x_piv, ld, ld, nb, 0, 0, 0 };
                const int si = push_src(B, sa);
                push_tasks(B, chol_block_row(Bk, run_lo) + (int64_t)(c0 + nb) * ld, ld, run_hi - run_lo, below, 0, si, si + 1);
              }
              run_lo = a0; run_hi = a1;
            }
          }
          flush_targets(B);
          const int sc[1] = { b->c_strips }, sn[1] = { b->n_groups };
          emit_update_jobs(P, B, p, k0, c_upd, c_updd, snap_upd, cnt_upd, cnt_updd, sc, sn, 1, NULL);
        }
      }
    }
    if (level == 0) { free(hs); break; }
    /* (E) POTRF jobs of the parents' first column blocks: they follow their children's strips of the parent rows; children in
     *     ascending pivot width (the one expected to finish last is consumed last), column blocks in order */
    memset(follow_lim, 0, p->nblk * sizeof(int));
    if (follow)
      for (int hp = h0 / 2; hp <= h1 / 2; hp++) {
        const int par = p->tree[hp];
        if (nblk_of[par] == 0 || pb[par][0].emitted || count[BIDX(p, par, par)] == 0) continue;
        int kids[2] = { p->tree[2 * hp], p->tree[2 * hp + 1] };
        if (p->sep_size[kids[0]] > p->sep_size[kids[1]]) { const int t = kids[0]; kids[0] = kids[1]; kids[1] = t; }
        const int ef = pg->n_ext;
        int all = 1; /* follow only if every contributing child has its channel (then the cells are left out of the extend-add) */
        for (int c = 0; c < 2; c++)
          for (int st = 0; st < nblk_of[kids[c]]; st++)
            if (count[BIDX(p, par, kids[c])] > 0 && pb[kids[c]][st].ch_par < 0) all = 0;
        if (!all) continue;
        ext_item *items = NULL; int nit = 0, capit = 0;
        const int lim = pb[par][0].nb, wide = opts->follow_tail > 0 && (lim + CHOL_NB - 1) / CHOL_NB > CHOL_FOLLOW_ALL_MAXT;
        early_src es[2 * 32]; int nes = 0;
        for (int c = 0; c < 2; c++) {
          int total = 0;
          for (int st = 0; st < nblk_of[kids[c]]; st++) total += (pb[kids[c]][st].nb + CHOL_NB - 1) / CHOL_NB;
          const int bt = wide && total > opts->follow_tail ? total - opts->follow_tail : 0; /* the child's leading column tiles that go the early way */
          int before = 0; /* column tiles of the child's earlier blocks */
          for (int st = 0; st < nblk_of[kids[c]]; st++) {
            const pblock *b = &pb[kids[c]][st];
            const int nt = (b->nb + CHOL_NB - 1) / CHOL_NB;
            const int et = bt - before < 0 ? 0 : bt - before > nt ? nt : bt - before; /* early column tiles of this block */
            if (b->ch_par >= 0) {
              if (et > 0 && nes < 64) { const early_src e1 = { b->par_off, p->panel_ld[kids[c]], et * CHOL_NB < b->nb ? et * CHOL_NB : b->nb, b->ch_par + et - 1, b->ns_par }; es[nes++] = e1; }
              push_ext_items(&items, &nit, &capit, b->par_off, p->panel_ld[kids[c]], b->nb, b->ch_par, b->ns_par, est_start[kids[c]] + before, et);
            }
            before += nt;
          }
        }
        if (nes > 0) {
          const chol_block *Bp = &p->blk[BIDX(p, par, par)];
          emit_early_cells(P, B, p, Bp->off, Bp->ld, lim, BIDX(p, par, par), es, nes, c_upd, c_updd, snap_upd, cnt_upd, cnt_updd);
        }
        qsort(items, nit, sizeof(ext_item), cmp_ext_item);
        for (int i = 0; i < nit; i++) add_ext(P, items[i].x);
        free(items);
        if (pg->n_ext == ef) continue;
        follow_lim[BIDX(p, par, par)] = pb[par][0].nb;
        EMIT_POTRF(par, 0, ef);
      }
    /* (F) extend-add of the level (tuples in program order, grouped by target), without the followed cells */
    {
      int cap_u = 256, ntu = 0;
      upd_tuple *tu = malloc(cap_u * sizeof(upd_tuple));
      int64_t seq = 0;
      for (int q = 0; q < nh; q++) {
        const int h = hs[q], s = p->tree[h], n = p->sep_size[s];
        for (int hp = h / 2; hp >= 1; hp /= 2) {
          const int par = p->tree[hp], bb = BIDX(p, par, s);
          const chol_block *Bb = &p->blk[bb];
          for (int hg = hp; hg >= 1; hg /= 2) {
            const int gp = p->tree[hg], ba = BIDX(p, gp, s), bc = BIDX(p, gp, par);
            const chol_block *Ba = &p->blk[ba], *Bc = &p->blk[bc];
            for (int i = 0; i < count[ba]; i++) {
              const cholamd_filled *fa = &snap[first[ba] + i];
              for (int j = 0; j < count[bb]; j++) {
                const cholamd_filled *fb_ = &snap[first[bb] + j];
                if (gp == par && fb_->cluster > fa->cluster) continue; /* col > row skipped, blas.rg:396-431 */
                if (ntu == cap_u) { cap_u *= 2; tu = realloc(tu, cap_u * sizeof(upd_tuple)); }
                upd_tuple *u = &tu[ntu++];
                const int crow = fa->lo_x - Bc->lo_x, ccol = fb_->lo_x - Bc->lo_y;
                u->key = ((int64_t)bc << 40) | ((int64_t)crow << 20) | (int64_t)ccol;
                u->seq = seq++;
                u->c_off = chol_block_row(Bc, crow) + (int64_t)ccol * Bc->ld; u->ldc = Bc->ld;
                u->a_off = chol_block_row(Ba, fa->lo_x - Ba->lo_x); u->lda = Ba->ld;
                u->b_off = chol_block_row(Bb, fb_->lo_x - Bb->lo_x); u->ldb = Bb->ld;
                u->m = fa->hi_x - fa->lo_x + 1; u->n = fb_->hi_x - fb_->lo_x + 1; u->k = n;
                u->syrk = (gp == par && fb_->cluster == fa->cluster);
                u->bc = bc; u->crow = crow; u->ccol = ccol; u->src_sep = s;
              }
            }
          }
        }
      }
      qsort(tu, ntu, sizeof(upd_tuple), cmp_tuple);
      const int k0 = w->n_task;
      memcpy(snap_upd, cnt_upd, (ns + 1) * sizeof(int));
      int any_follow = 0;
      for (int b = 0; b < p->nblk; b++) any_follow |= follow_lim[b] > 0;
      if (any_follow && !tuples_are_small(opts, tu, ntu)) { chol_set_error("program launch: followers need the grid-cell extend-add"); rc = CHOLAMD_ERR_ARG; }
      if (!rc) {
        if (tuples_are_small(opts, tu, ntu) || any_follow) { B->follow_lim = follow_lim; emit_cell_tasks(B, p, tu, ntu); B->follow_lim = NULL; }
        else {
          for (int i = 0; i < ntu;) {
            int e = i + 1;
            while (e < ntu && tu[e].key == tu[i].key) e++;
            const int sb = w->n_src;
            for (int q = i; q < e; q++) { chol_upd_src sd = { tu[q].a_off, tu[q].b_off, tu[q].lda, tu[q].ldb, tu[q].k, 0, -(64 * tu[q].src_sep + 63) - 1, 0 }; push_src(B, sd); }
            B->cur_blk = tu[i].bc;
            push_tasks(B, tu[i].c_off, tu[i].ldc, tu[i].m, tu[i].n, tu[i].syrk, sb, w->n_src);
            i = e;
          }
          flush_targets(B);
        }
        /* a job's sources are the separators of this level under its target's column separator: wait for all their strips.
         * Jobs are queued by urgency: targets deepest in the tree first (their pivots are factored next), diagonal blocks
         * before the rows below them; the order of the tasks in memory is irrelevant to the queue */
        typedef struct { int i, e, prio; } blk_range;
        int nrg = 0, caprg = 64;
        blk_range *rg = malloc(caprg * sizeof(blk_range));
        for (int i = k0; i < w->n_task;) {
          int e = i + 1;
          while (e < w->n_task && w->task[e].blk == w->task[i].blk) e++;
          const chol_block *Bc = &p->blk[w->task[i].blk];
          if (nrg == caprg) { caprg *= 2; rg = realloc(rg, caprg * sizeof(blk_range)); }
          rg[nrg].i = i; rg[nrg].e = e;
          rg[nrg].prio = (L - p->level_of[Bc->c]) * 4 * L + (Bc->r == Bc->c ? 0 : 1 + (p->level_of[Bc->c] - p->level_of[Bc->r])); /* ascending = more urgent first */
          nrg++;
          i = e;
        }
        for (int a = 1; a < nrg; a++) { /* stable insertion sort (a few dozen ranges) */
          blk_range t = rg[a];
          int b = a - 1;
          while (b >= 0 && rg[b].prio > t.prio) { rg[b + 1] = rg[b]; b--; }
          rg[b + 1] = t;
        }
        for (int q = 0; q < nrg; q++) {
          const int i = rg[q].i, e = rg[q].e;
          const chol_block *Bc = &p->blk[w->task[i].blk];
          const int hpar = p->heap_of[Bc->c], dl = level - p->level_of[Bc->c];
          int nsc = 0, capsc = 1;
          for (int hh = hpar << dl; hh < ((hpar + 1) << dl); hh++) capsc += nblk_of[p->tree[hh]] + 2 * ((p->sep_size[p->tree[hh]] + CHOL_NB - 1) / CHOL_NB);
          int *sc = malloc(capsc * sizeof(int)), *sn = malloc(capsc * sizeof(int)), *ssep = malloc(capsc * sizeof(int)), *sest = malloc(capsc * sizeof(int));
          for (int hh = hpar << dl; hh < ((hpar + 1) << dl); hh++) {
            const int s = p->tree[hh];
            int through = 0;
            for (int st = 0; st < nblk_of[s]; st++) {
              const pblock *b = &pb[s][st];
              if (sw_of[s] < bw_of[s] && b->n_groups > 0) { /* chunked (one block): per chunk the channels' last column tile of the chunk */
                const int nt = (b->nb + CHOL_NB - 1) / CHOL_NB, ct = sw_of[s] / CHOL_NB;
                for (int c = 0; c * ct < nt; c++) {
                  const int last = (c + 1) * ct < nt ? (c + 1) * ct - 1 : nt - 1;
                  if (b->ch_par >= 0) { sc[nsc] = b->ch_par + last; sn[nsc] = b->ns_par; ssep[nsc] = 64 * s + c; sest[nsc] = est_start[s] + last + 1; nsc++; }
                  if (b->ch_rest >= 0) { sc[nsc] = b->ch_rest + last; sn[nsc] = b->ns_rest; ssep[nsc] = 64 * s + c; sest[nsc] = est_start[s] + last + 1; nsc++; }
                }
                through += nt;
                continue;
              }
              through += (b->nb + CHOL_NB - 1) / CHOL_NB;
              if (b->n_groups > 0) { sc[nsc] = b->c_strips; sn[nsc] = b->n_groups; ssep[nsc] = 64 * s + st; sest[nsc] = est_start[s] + through; nsc++; }
            }
          }
          for (int a = 1; a < nsc; a++) { /* expected order of completion (stable) */
            const int c0_ = sc[a], n0_ = sn[a], s0_ = ssep[a], e0_ = sest[a];
            int b = a - 1;
            while (b >= 0 && sest[b] > e0_) { sc[b + 1] = sc[b]; sn[b + 1] = sn[b]; ssep[b + 1] = ssep[b]; sest[b + 1] = sest[b]; b--; }
            sc[b + 1] = c0_; sn[b + 1] = n0_; ssep[b + 1] = s0_; sest[b + 1] = e0_;
          }
          /* emit_update_jobs walks [i, n_task): hand it exactly this block's tasks */
          const int keep = w->n_task;
          w->n_task = e;
          emit_update_jobs(P, B, p, i, c_upd, c_updd, snap_upd, cnt_upd, cnt_updd, sc, sn, nsc, opts->staged ? ssep : NULL);
          w->n_task = keep;
          free(ssep); free(sest);
          free(sc); free(sn);
        }
        free(rg);
      }
      free(tu);
    }
    free(hs);
    if (w->n_task > PROG_MAX_TASKS && !rc) { chol_set_error("program launch: more than %d update tasks", PROG_MAX_TASKS); rc = CHOLAMD_ERR_ARG; }
  }
  #undef EMIT_POTRF
  if (!rc && w->n_task_mt > 0) { chol_set_error("program launch: macro-tile update phases"); rc = CHOLAMD_ERR_ARG; }
  if (!rc && w->n_task > PROG_MAX_TASKS) { chol_set_error("program launch: %d update tasks", w->n_task); rc = CHOLAMD_ERR_ARG; }
  for (int s = 1; s <= ns; s++) { pg->ctr_total[c_upd[s]] = cnt_upd[s]; pg->ctr_total[c_updd[s]] = cnt_updd[s]; if (P->c_updp) pg->ctr_total[P->c_updp[s]] = P->cnt_updp[s]; }
  free(P->c_updp); free(P->cnt_updp); P->c_updp = P->cnt_updp = NULL;
  for (int s = 1; s <= ns; s++) free(pb[s]);
  free_skylines(sky, ns); free(bw_of); free(band_of); free(sw_of);
  free(est_start); free(pb); free(nblk_of); free(first); free(count); free(c_upd); free(c_updd); free(cnt_upd); free(cnt_updd); free(snap_upd); free(follow_lim);
  free(B->pend);
  if (rc) { chol_level_work_free(w); chol_program_free(pg); }
  return rc;
}


/* Host-side self-check of the program (CPU tests): simulated with `workers` resident workgroups that take jobs in queue
 * order and with every counter raised only when its job has COMPLETED (stricter than the device, where strips follow a
 * POTRF column by column), every job must become runnable -- no dead-lock for any interleaving; counters total up; the
 * pivot blocks and TRSM rows are those of the per-level lists. */
typedef struct { int64_t a, b, c, d; } quad;
static int cmp_quad(const void *x, const void *y) { return memcmp(x, y, sizeof(quad)); }
int chol_program_check(const plan_t *p, const chol_sched_opts *opts, int workers)
{
  chol_level_work w;
  chol_program g;
  int rc = chol_build_program(p, opts, &w, &g);
  if (rc) return rc;
  rc = chol_program_check_built(p, opts, workers, &w, &g);
  chol_level_work_free(&w);
  chol_program_free(&g);
  return rc;
}
/* the same on a program the caller has built (and keeps): build_levels() checks the program it is about to upload */
int chol_program_check_built(const plan_t *p, const chol_sched_opts *opts, int workers, const chol_level_work *wp, const chol_program *gp)
{
  const chol_level_work w = *wp;
  const chol_program g = *gp;
  int rc = 0;
  int *val = calloc(g.n_ctr > 0 ? g.n_ctr : 1, sizeof(int));
  char *state = calloc(g.n_job > 0 ? g.n_job : 1, 1); /* 0 queued, 1 running, 2 done */
  int next = 0, running = 0, done = 0;
  while (done < g.n_job && !rc) {
    while (running < workers && next < g.n_job) { state[next++] = 1; running++; }
    int progressed = 0;
    for (int j = 0; j < next; j++) {
      if (state[j] != 1) continue;
      const chol_job *jb = &g.job[j];
      int ok = 1;
      for (int q = 0; q < jb->n_wait && ok; q++) {
        const chol_wait *wt = &g.wait[jb->wait_first + q];
        if (wt->ctr < 0 || wt->ctr >= g.n_ctr || wt->value > g.ctr_total[wt->ctr]) { chol_set_error("program job %d waits for %d on counter %d (total %d)", j, wt->value, wt->ctr, wt->ctr >= 0 && wt->ctr < g.n_ctr ? g.ctr_total[wt->ctr] : -1); rc = CHOLAMD_ERR_ARG; ok = 0; break; }
        if (val[wt->ctr] < wt->value) ok = 0;
      }
      if (rc) break;
      if (ok && jb->kind == 0)
        for (int x = 0; x < jb->n_ext && ok; x++) {
          const chol_ext *e = &g.ext[jb->ext_first + x];
          if (val[e->ctr] < e->need) ok = 0;
        }
      if (ok && jb->kind == 1 && val[w.trsm[jb->first].flag] < g.ctr_total[w.trsm[jb->first].flag]) ok = 0; /* its pivot block is factored */
      if (!ok) continue;
      /* complete */
      if (jb->kind == 0) val[w.potrf[jb->first].ctr] = g.ctr_total[w.potrf[jb->first].ctr];
      if (jb->kind == 1)
        for (int i = jb->first; i < jb->first + jb->n; i++)
          if (w.trsm[i].chan >= 0 && w.trsm[i].m > 0)
            for (int t = 0; t < (w.trsm[i].n + CHOL_NB - 1) / CHOL_NB; t++) val[w.trsm[i].chan + t]++;
      for (int q = 0; q < 2; q++) if (jb->sig[q] >= 0) val[jb->sig[q]] += jb->sig_add;
      state[j] = 2; running--; done++; progressed = 1;
    }
    if (!rc && !progressed && (running == workers || next == g.n_job)) {
      int j = 0;
      while (j < next && state[j] != 1) j++;
      chol_set_error("program dead-locks with %d resident workgroups: job %d (kind %d) can never run; %d of %d jobs done", workers, j, j < g.n_job ? g.job[j].kind : -1, done, g.n_job);
      rc = CHOLAMD_ERR_ARG;
    }
  }
  for (int c = 0; c < g.n_ctr && !rc; c++)
    if (val[c] != g.ctr_total[c]) { chol_set_error("program counter %d ends at %d, total %d", c, val[c], g.ctr_total[c]); rc = CHOLAMD_ERR_ARG; }
  free(val); free(state);
  if (!rc) { /* same pivot blocks and TRSM rows as the level lists */
    quad *a = NULL, *b = NULL;
    int64_t na = 0, nb = 0, ca = 0, cb = 0;
#define PUSHQ(V_, N_, C_, A_, B_, C2_, D_) do { if (N_ == C_) { C_ = C_ ? 2 * C_ : 4096; V_ = realloc(V_, C_ * sizeof(quad)); } quad q_ = { A_, B_, C2_, D_ }; V_[N_++] = q_; } while (0)
    for (int l = 0; l < p->levels && !rc; l++) {
      chol_level_work lw;
      chol_sched_opts lo = *opts; lo.leaf_envelope = 0; /* (the program keeps the leaves' zero rows in its strips: compare with the lists that do too) */
      rc = chol_build_level_work(p, &lo, l, 0, 1, &lw);
      if (rc) break;
      /* per 16-column tile (the program may factor a banded leaf as one block where the level lists split it): the diagonal tiles
       * factored, and per solved row the first element of every column tile */
      /* coverage per (row, 16-column tile) of the panels -- the program may factor a banded leaf as one block where the level lists
       * split it (rows of the later blocks are then panel rows of the POTRF, not TRSM rows): the first element of every column tile of
       * every row a POTRF block factors (diagonal tile and below) or a strip solves; each exactly once on both sides */
      for (int i = 0; i < lw.n_potrf; i++) for (int ct = 0; ct * CHOL_NB < lw.potrf[i].n; ct++) for (int r = ct * CHOL_NB; r < lw.potrf[i].n; r++) PUSHQ(a, na, ca, -1, lw.potrf[i].a_off + r + (int64_t)ct * CHOL_NB * lw.potrf[i].lda, 0, 0);
      for (int i = 0; i < lw.n_trsm; i++) for (int r = 0; r < lw.trsm[i].m; r++) for (int ct = 0; ct * CHOL_NB < lw.trsm[i].n; ct++) PUSHQ(a, na, ca, -1, lw.trsm[i].b_off + r + (int64_t)ct * CHOL_NB * lw.trsm[i].ldb, 0, 0);
      chol_level_work_free(&lw);
    }
    for (int i = 0; i < w.n_potrf; i++) for (int ct = 0; ct * CHOL_NB < w.potrf[i].n; ct++) for (int r = ct * CHOL_NB; r < w.potrf[i].n; r++) PUSHQ(b, nb, cb, -1, w.potrf[i].a_off + r + (int64_t)ct * CHOL_NB * w.potrf[i].lda, 0, 0);
    for (int i = 0; i < w.n_trsm; i++) for (int r = 0; r < w.trsm[i].m; r++) for (int ct = 0; ct * CHOL_NB < w.trsm[i].n; ct++) PUSHQ(b, nb, cb, -1, w.trsm[i].b_off + r + (int64_t)ct * CHOL_NB * w.trsm[i].ldb, 0, 0);
#undef PUSHQ
    if (!rc) {
      qsort(a, na, sizeof(quad), cmp_quad);
      qsort(b, nb, sizeof(quad), cmp_quad);
      if (na != nb || (na > 0 && memcmp(a, b, na * sizeof(quad)))) { chol_set_error("program pivot blocks / TRSM rows differ from the level lists (%lld vs %lld items)", (long long)nb, (long long)na); rc = CHOLAMD_ERR_ARG; }
    }
    free(a); free(b);
  }
  return rc;
}
int cholamd_plan_program_check(const cholamd_plan *p, int follow, int workers)
{
  chol_sched_opts o;
  chol_sched_opts_default(&o);
  o.follow = follow;
  return chol_program_check(p, &o, workers);
}
int cholamd_plan_program_check_opts(const cholamd_plan *p, int follow_tail, int split_min, int split_nb, int workers)
{ /* the same for other follower tails / pivot splits (negative: the default); the other switches as the environment sets them
   * (CHOLAMD_NO_SKYLINE, CHOLAMD_STAGE_CHUNK, ...: what a device object created now would use) */
  chol_sched_opts o;
  chol_sched_opts_from_env(&o);
  if (follow_tail >= 0) o.follow_tail = follow_tail;
  if (split_min >= 0) o.split_min = split_min;
  if (split_nb >= 0) o.split_nb = split_nb;
  return chol_program_check(p, &o, workers);
}
int64_t cholamd_plan_program_jobs(const cholamd_plan *p, int follow, int64_t cap, int *out)
{ /* diagnostic: per job 8 ints: kind, separator (POTRF / TRSM: the pivot's label; update: target block's column label), target
   * block's row label (update) or column offset of the pivot block, first, n, sig0, sig1, n_wait; then the wait list as
   * (job, ctr, value) triples appended after the jobs: returns the total number of ints */
  chol_sched_opts o;
  chol_sched_opts_default(&o);
  o.follow = follow;
  chol_level_work w; chol_program g;
  if (chol_build_program(p, &o, &w, &g)) return -1;
  const int64_t need = (int64_t)8 * g.n_job + (int64_t)3 * g.n_wait + 1;
  if (cap >= need) {
    int64_t k = 0;
    for (int j = 0; j < g.n_job; j++) {
      const chol_job *jb = &g.job[j];
      int a = 0, b = 0;
      if (jb->kind == 0) { a = w.potrf[jb->first].sep; b = w.potrf[jb->first].col0; }
      else if (jb->kind == 1) { for (int i = 0; i < w.n_potrf; i++) if (w.potrf[i].ctr == w.trsm[jb->first].flag) { a = w.potrf[i].sep; b = w.potrf[i].col0; } }
      else { a = p->blk[w.task[jb->first].blk].c; b = p->blk[w.task[jb->first].blk].r; }
      out[k++] = jb->kind; out[k++] = a; out[k++] = b; out[k++] = jb->first; out[k++] = jb->n; out[k++] = jb->sig[0]; out[k++] = jb->kind == 1 ? w.trsm[jb->first].chan : jb->sig[1]; out[k++] = jb->n_wait;
    }
    for (int j = 0; j < g.n_job; j++)
      for (int q = 0; q < g.job[j].n_wait; q++) { out[k++] = j; out[k++] = g.wait[g.job[j].wait_first + q].ctr; out[k++] = g.wait[g.job[j].wait_first + q].value; }
    out[k++] = g.n_job;
  }
  chol_level_work_free(&w); chol_program_free(&g);
  return need;
}
int cholamd_follow_rounds(int n_ext, int tile_columns, int cap, int *rounds_out, int *own_at_out)
{ /* the rounds in which a follower of `tile_columns` column tiles consumes a list of n_ext followed column tiles (the kernel's own functions,
   * chol_plan.h): returns the number of rounds, their sizes in rounds_out[0 .. cap), the item in front of which its own tiles go in */
  int n = 0;
  for (int i = 0; i < n_ext; i += chol_follow_round(i, n_ext, tile_columns), n++)
    if (n < cap && rounds_out) rounds_out[n] = chol_follow_round(i, n_ext, tile_columns);
  if (own_at_out) *own_at_out = chol_follow_own_at(n_ext, tile_columns);
  return n;
}
int64_t cholamd_plan_program_followers(const cholamd_plan *p, int64_t cap, int *out)
{ /* per following POTRF job of the program launch: job index, followed column tiles, column tiles of its pivot block, wait-list entries (3 ints) */
  chol_sched_opts o;
  chol_sched_opts_default(&o);
  chol_level_work w; chol_program g;
  if (chol_build_program(p, &o, &w, &g)) return -1;
  int64_t n = 0;
  for (int j = 0; j < g.n_job; j++) {
    const chol_job *jb = &g.job[j];
    if (jb->kind != 0 || jb->n_ext <= 0) continue;
    if (4 * n + 3 < cap) { out[4 * n] = j; out[4 * n + 1] = jb->n_ext; out[4 * n + 2] = (w.potrf[jb->first].n + CHOL_NB - 1) / CHOL_NB; out[4 * n + 3] = jb->n_wait; }
    n++;
  }
  chol_level_work_free(&w); chol_program_free(&g);
  return n;
}
int cholamd_plan_program_counts(const cholamd_plan *p, int follow, int out[6])
{ /* jobs, POTRF jobs that follow, update tasks, TRSM strips, counters, followed panels */
  chol_sched_opts o;
  chol_sched_opts_default(&o);
  o.follow = follow;
  chol_level_work w; chol_program g;
  int rc = chol_build_program(p, &o, &w, &g);
  if (rc) return rc;
  int nf = 0;
  for (int j = 0; j < g.n_job; j++) nf += g.job[j].kind == 0 && g.job[j].n_ext > 0;
  out[0] = g.n_job; out[1] = nf; out[2] = w.n_task; out[3] = w.n_trsm; out[4] = g.n_ctr; out[5] = g.n_ext;
  chol_level_work_free(&w); chol_program_free(&g);
  return 0;
}

/* volumes of one level's lists for (rank, world) under dist_top = 0 / 1 / 2 (auto): what the CPU tests sum over the ranks.
 * out: POTRF columns, TRSM elements (rows x columns), update volume (target elements x source depth), broadcast entries,
 * broadcast doubles, a checksum of the broadcast list (identical on every rank: the sequence is collective) */
static int level_work_volume(const cholamd_plan *p, const chol_sched_opts *o, int level, int rank, int world, int64_t out[6], int64_t counts[3])
{
  chol_level_work w;
  int rc = chol_build_level_work(p, o, level, rank, world, &w);
  if (rc) return rc;
  memset(out, 0, 6 * sizeof(int64_t));
  if (counts) { counts[0] = w.n_task; counts[1] = w.n_task_mt; counts[2] = 0; for (int i = 0; i < w.n_trsm; i++) counts[2] += w.trsm[i].m > 0; }
  for (int i = 0; i < w.n_potrf; i++) out[0] += w.potrf[i].n;
  for (int i = 0; i < w.n_trsm; i++) out[1] += (int64_t)w.trsm[i].m * w.trsm[i].n;
  for (int pass = 0; pass < 2; pass++) {
    const chol_upd_task *T = pass ? w.task_mt : w.task;
    const int nt = pass ? w.n_task_mt : w.n_task;
    for (int i = 0; i < nt; i++) {
      const chol_upd_task *t = &T[i];
      int64_t full = 0;
      for (int c = 0; c < t->nv; c++) full += t->lower ? (t->mv - c > 0 ? t->mv - c : 0) : t->mv;
      for (int q = t->src_begin; q < t->src_end; q++) {
        const chol_upd_src *sd = &w.src[q];
        const int r0 = sd->range & 255, r1 = (sd->range >> 8) & 255, c0 = (sd->range >> 16) & 255, c1 = (sd->range >> 24) & 255;
        out[2] += (sd->range ? (int64_t)(r1 - r0) * (c1 - c0) : full) * sd->k;
      }
    }
  }
  out[3] = w.n_bcast;
  for (int i = 0; i < w.n_bcast; i++) { out[4] += w.bcast[i].count; out[5] = (int64_t)((uint64_t)out[5] * 1000003u + (uint64_t)w.bcast[i].off * 31u + (uint64_t)w.bcast[i].owner + 7u * (uint64_t)w.bcast[i].count); /* a hash of the sequence: wraps (unsigned) */ }
  int nb6 = 0;
  for (int i = 0; i < w.n_phase; i++) if (w.phase[i].kind == 6) nb6 += w.phase[i].n;
  if (nb6 != w.n_bcast) { chol_set_error("internal: %d broadcast entries, %d in phases", w.n_bcast, nb6); rc = CHOLAMD_ERR_ARG; }
  chol_level_work_free(&w);
  return rc;
}
int cholamd_plan_level_work_volume(const cholamd_plan *p, int level, int rank, int world, int dist_top, int64_t out[6])
{
  chol_sched_opts o;
  chol_sched_opts_default(&o);
  o.dist_top = dist_top;
  return level_work_volume(p, &o, level, rank, world, out, NULL);
}
/* the same volumes under the level schedule's merging switch and macro-tile threshold (single GPU), plus the list lengths:
 * out[6..8] = 16x16 tasks, 64x64 macro-tile tasks, TRSM strips */
int cholamd_plan_level_work_volume_opts(const cholamd_plan *p, int level, int merge_targets, int mt_min_tiles, int64_t out[9])
{
  chol_sched_opts o;
  chol_sched_opts_default(&o);
  o.merge_targets = merge_targets != 0;
  o.leaf_envelope = 0; /* (the dense work: a leaf's sources are cut at the first entry of the TARGET's rows, which depends on how targets are merged) */
  if (mt_min_tiles >= 0) o.mt_min_tiles = mt_min_tiles;
  return level_work_volume(p, &o, level, 0, 1, out, out + 6);
}

/* Volume of the extend-add exchange for (rank, world) under dist_top = 0 / 1 / 2 (auto), in arena elements: out[0] received, out[1]
 * sent, out[2] the tail, out[3] column-block pieces (0: replicated top levels, one all-reduce of the tail -- a ring moves
 * 2 (world - 1) / world of it each way).  Distributed top levels: a rank sends the column blocks it does not own of the top separators ON ITS
 * OWN ROOT PATH once, to their owners (rank 0: of every top separator -- it holds A's entries), and receives, for every block it owns, one copy
 * from each other rank under the block's separator and from rank 0 (chol_top_contributors; chol_api.cpp, exchange_owned). */
int cholamd_plan_exchange_volume(const cholamd_plan *p, int rank, int world, int dist_top, int64_t out[4])
{
  chol_sched_opts o;
  chol_sched_opts_default(&o);
  o.dist_top = dist_top;
  const int d = chol_split_level(world);
  if (world < 1 || (1 << d) != world || d > p->levels - 1 || rank < 0 || rank >= world) { chol_set_error("bad partition rank %d of %d", rank, world); return CHOLAMD_ERR_ARG; }
  out[0] = out[1] = out[3] = 0;
  out[2] = world > 1 ? p->arena - p->panel_off[p->nsep - (world - 1) + 1] : 0;
  for (int lvl = d - 1; lvl >= 0; lvl--) {
    chol_level_work w;
    int rc = chol_build_level_work(p, &o, lvl, rank, world, &w);
    if (rc) return rc;
    for (int i = 0; i < w.n_bcast; i++) {
      out[3]++;
      const unsigned c = chol_top_contributors(w.bcast[i].heap, world);
      if (w.bcast[i].owner == rank) out[0] += w.bcast[i].count * __builtin_popcount(c & ~(1u << rank));
      else if (c & (1u << rank)) out[1] += w.bcast[i].count;
    }
    chol_level_work_free(&w);
  }
  if (out[3] == 0 && world > 1) out[0] = out[1] = 2 * out[2] * (world - 1) / world;
  return 0;
}

int cholamd_plan_exchange_pieces(const cholamd_plan *p, int world, int dist_top, int max, int64_t (*out)[4])
{
  chol_sched_opts o;
  chol_sched_opts_default(&o);
  o.dist_top = dist_top;
  const int d = chol_split_level(world);
  if (world < 1 || (1 << d) != world || d > p->levels - 1) { chol_set_error("bad partition into %d", world); return CHOLAMD_ERR_ARG; }
  int n = 0;
  for (int lvl = d - 1; lvl >= 0; lvl--) {
    chol_level_work w;
    int rc = chol_build_level_work(p, &o, lvl, 0, world, &w);
    if (rc) return rc;
    for (int i = 0; i < w.n_bcast; i++, n++)
      if (n < max) { out[n][0] = w.bcast[i].off; out[n][1] = w.bcast[i].count; out[n][2] = w.bcast[i].owner; out[n][3] = w.bcast[i].heap; }
    chol_level_work_free(&w);
  }
  return n;
}

void chol_level_work_free(chol_level_work *w)
{
  free(w->potrf); free(w->trsm); free(w->task); free(w->task_mt); free(w->src); free(w->phase); free(w->bcast);
  memset(w, 0, sizeof *w);
}

/* ---------------------------------------------------------------------------------------- */
/* solve phase lists (mmat.rg:1394-1479), level by level                                      */
/*   forward  (bottom-up): TRSV per separator, then target-centric GEMV into every ancestor   */
/*   backward (top-down) : per separator gather from all ancestors (GEMV Trans), then TRSV^T  */
/* ---------------------------------------------------------------------------------------- */
/* the stored row runs of a block: maximal runs of consecutive kept 16-row tiles (one run = the whole block without compaction).
 * Calls f(ctx, first row, rows, arena offset of the first row) per run; rows in [lo, hi) only */
typedef struct { int row0, m; int64_t off; } blk_run;
static int block_runs(const chol_block *B, int lo, int hi, blk_run **out)
{
  int cap = 8, n = 0;
  blk_run *r = malloc(cap * sizeof(blk_run));
  if (hi > B->rows) hi = B->rows;
  if (!B->tmap) {
    if (lo < hi) { r[0].row0 = lo; r[0].m = hi - lo; r[0].off = B->off + lo; n = 1; }
  } else
    for (int t = lo / CHOL_NB; t * CHOL_NB < hi; t++) {
      if (B->tmap[t] < 0) continue;
      const int a = t * CHOL_NB > lo ? t * CHOL_NB : lo, b = (t + 1) * CHOL_NB < hi ? (t + 1) * CHOL_NB : hi;
      if (n > 0 && r[n - 1].row0 + r[n - 1].m == a) { r[n - 1].m += b - a; continue; }
      if (n == cap) { cap *= 2; r = realloc(r, cap * sizeof(blk_run)); }
      r[n].row0 = a; r[n].m = b - a; r[n].off = chol_block_row(B, a); n++;
    }
  *out = r;
  return n;
}

int cholamd_plan_solve_counts(const cholamd_plan *p, int level, int rank, int world, int64_t out[5])
{ /* host-side view of one rank's solve lists of a level: separators, (ancestor, separator) row runs, forward / backward row chunks, columns solved */
  if (level < 0 || level >= p->levels || world < 1 || (world & (world - 1)) || rank < 0 || rank >= world || chol_split_level(world) > p->levels - 1) { chol_set_error("bad solve partition"); return CHOLAMD_ERR_ARG; }
  chol_solve_level w;
  int rc = chol_build_solve_level_part(p, level, rank, world, &w);
  if (rc) return rc;
  out[0] = w.n_trsv; out[1] = w.n_bw; out[2] = w.n_ifw; out[3] = w.n_ibw; out[4] = 0;
  for (int i = 0; i < w.n_trsv; i++) out[4] += w.trsv[i].n;
  chol_solve_level_free(&w);
  return 0;
}
int chol_build_solve_level(const plan_t *p, int level, chol_solve_level *w) { return chol_build_solve_level_part(p, level, 0, 1, w); }
int cholamd_plan_solve_skips(const cholamd_plan *p, int level, int *seps, int *runs)
{
  if (!p || level < 0 || level >= p->levels || !seps || !runs) { chol_set_error("solve_skips: bad arguments"); return CHOLAMD_ERR_ARG; }
  chol_solve_level w;
  int rc = chol_build_solve_level(p, level, &w);
  if (rc) return rc;
  for (int t = 0; t < w.n_trsv; t++) { seps[3 * t] = w.trsv[t].x_off; seps[3 * t + 1] = w.trsv[t].n; seps[3 * t + 2] = w.trsv[t].band; }
  for (int q = 0; q < w.n_bw; q++) { runs[5 * q] = w.bw[q].x_off; runs[5 * q + 1] = w.bw[q].m; runs[5 * q + 2] = w.bw[q].y_off; runs[5 * q + 3] = w.bw[q].n; runs[5 * q + 4] = w.bw[q].c_lo; }
  chol_solve_level_free(&w);
  return 0;
}
/* the same for rank `rank` of `world` (distributed solve, mmat.rg:1394-1479 sharded like the factorisation): below the cut only the separators of
 * the rank's own subtrees -- their TRSVs and their panels into every ancestor, the shared top included; the levels above the cut complete (every
 * rank holds the whole factored top and solves it redundantly: only vectors travel) */
int chol_build_solve_level_part(const plan_t *p, int level, int rank, int world, chol_solve_level *w)
{
  memset(w, 0, sizeof *w);
  const int cut = chol_split_level(world);
#define SOLVE_MINE(s_) (world <= 1 || level < cut || chol_owner_of(p, (s_), world) == rank)
  const int h0 = 1 << level, h1 = (1 << (level + 1)) - 1, cnt = h1 - h0 + 1;
  int capb = cnt * (level > 0 ? level : 1);
  w->trsv = malloc(cnt * sizeof(chol_trsv_desc));
  w->bw_start = malloc((cnt + 1) * sizeof(int));
  w->bw = malloc((size_t)capb * sizeof(chol_gemv_desc));
  for (int h = h0; h <= h1; h++) {
    int s = p->tree[h];
    if (!SOLVE_MINE(s)) continue;
    chol_trsv_desc t = { p->panel_off[s], p->sep_size[s], p->panel_ld[s], p->sep_off[s], s, p->dinv_off[s], 0 };
    if (p->sep_size[s] > w->max_n) w->max_n = p->sep_size[s];
    w->bw_start[w->n_trsv] = w->n_bw;
    w->trsv[w->n_trsv++] = t;
    for (int hp = h / 2; hp >= 1; hp /= 2) {
      int par = p->tree[hp];
      const chol_block *B = &p->blk[BIDX(p, par, s)];
      if (B->rows == 0 || B->cols == 0) continue;
      blk_run *rr; const int nr = block_runs(B, 0, B->rows, &rr); /* one descriptor per stored row run */
      for (int q = 0; q < nr; q++)
        for (int r0 = 0; r0 < rr[q].m; r0 += CHOL_SOLVE_BW_ROWS) { /* rows of a run are consecutive in memory and in the ancestor's vector */
          if (w->n_bw == capb) { capb *= 2; w->bw = realloc(w->bw, (size_t)capb * sizeof(chol_gemv_desc)); }
          const int m = rr[q].m - r0 < CHOL_SOLVE_BW_ROWS ? rr[q].m - r0 : CHOL_SOLVE_BW_ROWS;
          chol_gemv_desc g = { rr[q].off + r0, m, B->cols, B->ld, p->sep_off[par] + rr[q].row0 + r0, p->sep_off[s], 0 };
          w->bw[w->n_bw++] = g;
        }
      free(rr);
    }
  }
  w->bw_start[w->n_trsv] = w->n_bw;
  w->max_rows_under_span = w->max_n;
  int **rfirst = NULL; /* leaf level: first entry of every panel row */
  if (level == p->levels - 1 && w->n_trsv > 0 && !getenv("CHOLAMD_SOLVE_NO_BAND")) { /* leaves: the element band of the tile skyline (a tile (ti, tj) with tj < sky[ti] is zero) */
    rfirst = leaf_row_first(p);
    for (int t = 0; t < w->n_trsv; t++) { /* ... and the first non-zero column of every row run into the ancestors */
      const int s = w->trsv[t].sep;
      if (!rfirst[s]) continue;
      for (int q = w->bw_start[t]; q < w->bw_start[t + 1]; q++) {
        const int pr = (int)((w->bw[q].a_off - p->panel_off[s]) % p->panel_ld[s]);
        int f = w->bw[q].n;
        for (int r = 0; r < w->bw[q].m; r++) if (rfirst[s][pr + r] < f) f = rfirst[s][pr + r];
        w->bw[q].c_lo = f & ~15;
      }
    }
    unsigned char **sky = leaf_skylines(p);
    w->max_rows_under_span = 0;
    for (int t = 0; t < w->n_trsv; t++) {
      const int s = w->trsv[t].sep, n = w->trsv[t].n;
      int band = 0;
      if (sky[s]) {
        int md = 0;
        for (int i = 0; i * CHOL_NB < n; i++) if (i - sky[s][i] > md) md = i - sky[s][i];
        band = CHOL_NB * md + CHOL_NB - 1;
        if (band >= n) band = 0;
      }
      w->trsv[t].band = band;
      if ((band > 0 ? band : n) > w->max_rows_under_span) w->max_rows_under_span = band > 0 ? band : n;
    }
    w->banded = w->max_rows_under_span <= 256; /* (SSPAN of chol_kernels.hip) */
    free_skylines(sky, p->nsep);
  }
  /* forward: for every ancestor separator `par` (levels above), chunks of 256 rows; a source = the stored rows of a block inside the
   * chunk (y_off = first row of the run, relative to the target separator) */
  int cap = 16, capg = 16;
  w->fw = malloc(cap * sizeof(chol_gemv_desc));
  w->grp_start = malloc((capg + 1) * sizeof(int));
  w->grp_rows = malloc(2 * capg * sizeof(int));
  for (int pl = level - 1; pl >= 0; pl--)
    for (int hp = 1 << pl; hp < (1 << (pl + 1)); hp++) {
      int par = p->tree[hp];
      if (p->sep_size[par] == 0) continue;
      for (int row0 = 0; row0 < p->sep_size[par]; row0 += 256) {
        if (w->n_grp == capg) { capg *= 2; w->grp_start = realloc(w->grp_start, (capg + 1) * sizeof(int)); w->grp_rows = realloc(w->grp_rows, 2 * capg * sizeof(int)); }
        w->grp_start[w->n_grp] = w->n_fw;
        w->grp_rows[2 * w->n_grp] = row0; w->grp_rows[2 * w->n_grp + 1] = p->sep_off[par];
        w->n_grp++;
        for (int h = hp << (level - pl); h < ((hp + 1) << (level - pl)); h++) {
          int s = p->tree[h];
          if (!SOLVE_MINE(s)) continue;
          const chol_block *B = &p->blk[BIDX(p, par, s)];
          if (B->rows == 0 || B->cols == 0) continue;
          blk_run *rr; const int nr = block_runs(B, row0, row0 + 256, &rr);
          for (int q = 0; q < nr; q++) {
            if (w->n_fw == cap) { cap *= 2; w->fw = realloc(w->fw, cap * sizeof(chol_gemv_desc)); }
            chol_gemv_desc g = { rr[q].off, rr[q].m, B->cols, B->ld, p->sep_off[s], rr[q].row0, 0 };
            w->fw[w->n_fw++] = g;
          }
          free(rr);
        }
      }
    }
  w->grp_start[w->n_grp] = w->n_fw;
  /* row chunks of the blocks for the source-centric kernels of the driver-level solve */
  { /* backward: per separator, 64-column chunks x ranges of its row runs holding about R rows; R as large as keeps CHOL_SOLVE_BW_ITEMS workgroups */
    int64_t R = (int64_t)1 << 30;
    int n = 0, *it = NULL;
    for (;;) {
      n = 0;
      for (int pass = 0; pass < 2; pass++) { /* count, then fill */
        int k = 0;
        for (int t = 0; t < w->n_trsv; t++) {
          const int nc = (w->trsv[t].n + CHOL_SOLVE_BW_COLS - 1) / CHOL_SOLVE_BW_COLS;
          for (int q0 = w->bw_start[t]; q0 < w->bw_start[t + 1];) {
            int q1 = q0; int64_t rows = 0;
            while (q1 < w->bw_start[t + 1] && (q1 == q0 || rows + w->bw[q1].m <= R)) rows += w->bw[q1++].m;
            if (pass) for (int c = 0; c < nc; c++) { it[4 * k] = q0; it[4 * k + 1] = q1; it[4 * k + 2] = c * CHOL_SOLVE_BW_COLS; it[4 * k + 3] = 0; k++; }
            else k += nc;
            q0 = q1;
          }
        }
        if (!pass) {
          n = k;
          if (n < CHOL_SOLVE_BW_ITEMS && R > CHOL_SOLVE_BW_ROWS) break; /* too few: halve R and count again */
          it = malloc((size_t)(n > 0 ? 4 * n : 4) * sizeof(int));
        }
      }
      if (it) break;
      R = R > ((int64_t)1 << 20) ? ((int64_t)1 << 20) : R / 2;
    }
    w->ibw = it; w->n_ibw = n;
  }
  for (int pass = 0; pass < 1; pass++) {
    const int rows = pass ? CHOL_SOLVE_BW_ROWS : CHOL_SOLVE_FW_ROWS;
    int n = 0;
    for (int i = 0; i < w->n_bw; i++) n += ((w->bw[i].m + rows - 1) / rows) * ((w->bw[i].n + CHOL_SOLVE_COLS - 1) / CHOL_SOLVE_COLS);
    int *it = malloc((size_t)(n > 0 ? 4 * n : 4) * sizeof(int)), k = 0;
    for (int t = 0; t < w->n_trsv; t++) {
      const int s = w->trsv[t].sep;
      for (int i = w->bw_start[t]; i < w->bw_start[t + 1]; i++)
        for (int c0 = 0; c0 < w->bw[i].n; c0 += CHOL_SOLVE_COLS)
          for (int r0 = 0; r0 < w->bw[i].m; r0 += rows) {
            int f = 0;
            if (rfirst && rfirst[s]) { /* the first non-zero column of these rows */
              const int pr = (int)((w->bw[i].a_off - p->panel_off[s]) % p->panel_ld[s]);
              f = w->bw[i].n;
              for (int r = r0; r < r0 + rows && r < w->bw[i].m; r++) if (rfirst[s][pr + r] < f) f = rfirst[s][pr + r];
              f &= ~15;
            }
            it[4 * k] = i; it[4 * k + 1] = r0; it[4 * k + 2] = c0; it[4 * k + 3] = f; k++;
          }
    }
    if (pass) { w->ibw = it; w->n_ibw = n; } else { w->ifw = it; w->n_ifw = n; }
  }
  if (rfirst) { for (int s = 1; s <= p->nsep; s++) free(rfirst[s]); free(rfirst); }
#undef SOLVE_MINE
  return 0;
}

void chol_solve_level_free(chol_solve_level *w)
{
  free(w->trsv); free(w->fw); free(w->grp_start); free(w->grp_rows); free(w->bw); free(w->bw_start); free(w->ifw); free(w->ibw);
  memset(w, 0, sizeof *w);
}

/* ---------------------------------------------------------------------------------------- */
/* host-side views of the multi-GPU partition (testable without a device)                     */
/* ---------------------------------------------------------------------------------------- */
int cholamd_plan_fill_host_part(const cholamd_plan *p, double *arena, int rank, int world, int64_t *tail_offset_out)
{
  const int d = chol_split_level(world);
  if (world < 1 || (1 << d) != world || d > p->levels - 1 || rank < 0 || rank >= world) { chol_set_error("bad partition rank %d of %d", rank, world); return CHOLAMD_ERR_ARG; }
  const int64_t tail = world > 1 ? p->panel_off[p->nsep - (world - 1) + 1] : p->arena;
  if (tail_offset_out) *tail_offset_out = tail;
  memset(arena, 0, (size_t)p->arena * sizeof(double));
  for (int64_t e = 0; e < p->nnz_a; e++)
    if (rank == 0 || p->a_dst[e] < tail) arena[p->a_dst[e]] = p->a_val[e];
  return 0;
}

/* how full the 64x64 macro-tile tasks of one level are: out = { tasks, tasks with 64 x 64 valid elements, sum over tasks of valid elements x K,
 * sum over tasks of 4096 x K } (K = total depth of the task's sources): out[2] / out[3] is the share of the kernel's MFMA work that lands on
 * valid elements */
int cholamd_plan_level_mt_hist(const cholamd_plan *p, int level, int64_t out[16])
{ /* K-weighted count of the macro-tile tasks by (ceil(mv / 16) - 1, ceil(nv / 16) - 1) */
  chol_level_work w;
  int rc = chol_build_level_work(p, NULL, level, 0, 1, &w);
  if (rc) return rc;
  for (int i = 0; i < 16; i++) out[i] = 0;
  for (int i = 0; i < w.n_task_mt; i++) {
    const chol_upd_task *t = &w.task_mt[i];
    int64_t K = 0;
    for (int q = t->src_begin; q < t->src_end; q++) K += w.src[q].k;
    out[((t->mv + 15) / 16 - 1) * 4 + (t->nv + 15) / 16 - 1] += K;
  }
  chol_level_work_free(&w);
  return 0;
}
int cholamd_plan_level_mt_fill(const cholamd_plan *p, int level, int64_t out[4])
{
  chol_level_work w;
  chol_sched_opts o; chol_sched_opts_from_env(&o);
  int rc = chol_build_level_work(p, &o, level, 0, 1, &w);
  if (rc) return rc;
  out[0] = w.n_task_mt; out[1] = out[2] = out[3] = 0;
  if (getenv("CHOLAMD_PRINT_PHASES"))
    for (int i = 0; i < w.n_phase; i++)
      if (w.phase[i].kind == 3) {
        double work = 0; int64_t nsrc = 0, kmax = 0;
        for (int q = w.phase[i].first; q < w.phase[i].first + w.phase[i].n; q++) {
          const chol_upd_task *t = &w.task_mt[q];
          for (int z = t->src_begin; z < t->src_end; z++) { work += 2.0 * 4096 * w.src[z].k; if (w.src[z].k > kmax) kmax = w.src[z].k; }
          nsrc += t->src_end - t->src_begin;
        }
        fprintf(stderr, "level %d phase %d: %d macro-tile tasks, %.3g padded flop, %.2f sources per task, K max %ld\n", level, i, w.phase[i].n, work, (double)nsrc / w.phase[i].n, (long)kmax);
      }
  for (int i = 0; i < w.n_task_mt; i++) {
    const chol_upd_task *t = &w.task_mt[i];
    int64_t K = 0;
    for (int q = t->src_begin; q < t->src_end; q++) K += w.src[q].k;
    const int64_t valid = t->lower ? (int64_t)t->mv * (t->mv + 1) / 2 : (int64_t)t->mv * t->nv;
    out[1] += t->mv == 64 && t->nv == 64;
    out[2] += valid * K; out[3] += (t->lower ? 64 * 65 / 2 : 4096) * K;
  }
  chol_level_work_free(&w);
  return 0;
}

int cholamd_plan_level_work_counts(const cholamd_plan *p, int level, int rank, int world, int out[4])
{
  chol_level_work w;
  int rc = chol_build_level_work(p, NULL, level, rank, world, &w);
  if (rc) return rc;
  int strips = 0; /* without the placeholders that fill the groups of four */
  for (int i = 0; i < w.n_trsm; i++) strips += w.trsm[i].m > 0;
  out[0] = w.n_potrf; out[1] = strips; out[2] = w.n_task + w.n_task_mt; out[3] = w.n_src;
  chol_level_work_free(&w);
  return 0;
}
