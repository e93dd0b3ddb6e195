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
} upd_tuple;

static int cmp_tuple(const void *x, const void *y)
{
  const upd_tuple *a = x, *b = y;
  if (a->key != b->key) return a->key < b->key ? -1 : 1;
  return a->seq < b->seq ? -1 : (a->seq > b->seq);
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
typedef struct { int64_t c_off; int ldc, m, n, syrk, src_begin, src_end; } upd_target;
typedef struct { chol_level_work *w; const chol_sched_opts *o; int cap_p, cap_t, cap_k, cap_km, cap_s, cap_ph; upd_target *pend; int n_pend, cap_pend; } builder;
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
  o->mt_min_tiles = CHOL_MT_MIN_TILES; o->cells = 1;
}
void chol_sched_opts_from_env(chol_sched_opts *o)
{
  chol_sched_opts_default(o);
  o->split_min = env_int("CHOLAMD_SPLIT_MIN", o->split_min);
  o->split_nb = env_int("CHOLAMD_SPLIT_NB", o->split_nb);
  o->fuse = !env_int("CHOLAMD_NO_FUSE", 0);
  o->fuse_update_max = env_int("CHOLAMD_FUSE_UPDATE_MAX", o->fuse_update_max);
  o->mt_min_tiles = env_int("CHOLAMD_MT_MIN_TILES", o->mt_min_tiles);
  o->cells = !env_int("CHOLAMD_NO_CELLS", 0);
}
static int split_nb(const chol_sched_opts *o) { int v = o->split_nb; if (v > CHOL_RR_MAXN) v = CHOL_RR_MAXN; v = (v + 15) / 16 * 16; if (v < 16) v = 16; return v; }
static int pivot_blocks(const chol_sched_opts *o, int n) { return n > o->split_min || n > CHOL_RR_MAXN ? (n + split_nb(o) - 1) / split_nb(o) : 1; }
static int pivot_block_width(const chol_sched_opts *o, int n) { const int nb = pivot_blocks(o, n); return nb == 1 ? n : ((n + nb - 1) / nb + 15) / 16 * 16; }

/* a phase whose 16x16 sub-tile count reaches this goes to 64x64 macro tiles for its larger targets:
 * below it the 16x16 split-K workgroups are what fills the 256 CUs, above it their 4x operand
 * re-reads are what costs */

static void push_phase(builder *B, int kind, int first, int n)
{
  if (n <= 0) return;
  chol_level_work *w = B->w;
  if (w->n_phase == B->cap_ph) { B->cap_ph = B->cap_ph ? 2 * B->cap_ph : 16; w->phase = realloc(w->phase, B->cap_ph * sizeof(chol_phase)); }
  chol_phase ph = { kind, first, n, 0, 0, 0, 0 };
  w->phase[w->n_phase++] = ph;
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
    chol_trsm_desc td = { l_off, dinv_off, b_off + r0, n, ld, mm, ld, flag, 0 };
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
    chol_trsm_desc td = { l_off, dinv_off, b_off, n, ld, 0, ld, flag, 0 };
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
static void emit_tasks(builder *B, int macro, int64_t c_off, int ldc, int m, int n, int syrk, int src_begin, int src_end)
{
  chol_level_work *w = B->w;
  const int ts = macro ? 64 : 16;
  const int tr = (m + ts - 1) / ts, tc = (n + ts - 1) / ts;
  for (int a = 0; a < tr; a++)
    for (int b = 0; b < tc; b++) {
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
      t->ar = a * ts; t->br = b * ts;
    }
}

/* targets of the current update phase; flush_targets() chooses the tile shape once the phase is complete */
static void push_tasks(builder *B, int64_t c_off, int ldc, int m, int n, int syrk, int src_begin, int src_end)
{
  if (B->n_pend == B->cap_pend) { B->cap_pend = B->cap_pend ? 2 * B->cap_pend : 256; B->pend = realloc(B->pend, B->cap_pend * sizeof(upd_target)); }
  upd_target t = { c_off, ldc, m, n, syrk, src_begin, src_end };
  B->pend[B->n_pend++] = t;
}
static void flush_targets(builder *B)
{
  int64_t fine = 0;
  for (int i = 0; i < B->n_pend; i++) {
    const int64_t tr = (B->pend[i].m + 15) / 16, tc = (B->pend[i].n + 15) / 16;
    fine += B->pend[i].syrk ? tr * (tr + 1) / 2 : tr * tc;
  }
  const int big = fine >= B->o->mt_min_tiles;
  for (int i = 0; i < B->n_pend; i++) {
    const upd_target *t = &B->pend[i];
    emit_tasks(B, big && (t->m > 16 || t->n > 16), t->c_off, t->ldc, t->m, t->n, t->syrk, t->src_begin, t->src_end);
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
typedef struct { int64_t key, seq, c_off, a_off, b_off; int ldc, lda, ldb, k, mv, nv, lower, range; } cell_piece;
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
        if (np == cap) { cap *= 2; pc = realloc(pc, (size_t)cap * sizeof(cell_piece)); }
        cell_piece *q = &pc[np++];
        const int r0 = (u->crow > 16 * I ? u->crow : 16 * I) - 16 * I, r1 = (u->crow + u->m < 16 * I + 16 ? u->crow + u->m : 16 * I + 16) - 16 * I;
        const int c0 = (u->ccol > 16 * J ? u->ccol : 16 * J) - 16 * J, c1 = (u->ccol + u->n < 16 * J + 16 ? u->ccol + u->n : 16 * J + 16) - 16 * J;
        q->key = ((int64_t)u->bc << 40) | ((int64_t)I << 20) | (int64_t)J;
        q->seq = u->seq;
        q->c_off = Bc->off + 16 * I + (int64_t)(16 * J) * Bc->ld; q->ldc = Bc->ld;
        q->mv = Bc->rows - 16 * I < 16 ? Bc->rows - 16 * I : 16;
        q->nv = Bc->cols - 16 * J < 16 ? Bc->cols - 16 * J : 16;
        q->lower = diag && I == J;
        /* operand rows as seen from row 0 / column 0 of the cell (only [r0, r1) / [c0, c1) are read) */
        q->a_off = u->a_off + (16 * I - u->crow); q->lda = u->lda;
        q->b_off = u->b_off + (16 * J - u->ccol); q->ldb = u->ldb;
        q->k = u->k;
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
      chol_upd_src sd = { pc[q].a_off, pc[q].b_off, pc[q].lda, pc[q].ldb, pc[q].k, r0 | (r1 << 8) | (c0 << 16) | (c1 << 24) };
      push_src(B, sd);
      q = f;
    }
    if (w->n_task == B->cap_k) { B->cap_k = B->cap_k ? 2 * B->cap_k : 256; w->task = realloc(w->task, B->cap_k * sizeof(chol_upd_task)); }
    chol_upd_task *t = &w->task[w->n_task++];
    memset(t, 0, sizeof *t);
    t->c_off = pc[i].c_off; t->ldc = pc[i].ldc;
    t->mv = (short)pc[i].mv; t->nv = (short)pc[i].nv;
    t->lower = pc[i].lower;
    t->src_begin = sb; t->src_end = w->n_src;
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
typedef struct { int64_t off; int m; } row_run;
/* which: 0 = every ancestor, 1 = the parent only, 2 = every ancestor but the parent */
static int ancestor_runs_of(const plan_t *p, int h, int which, const cholamd_filled *snap, const int64_t *first, const int *count, row_run **out)
{
  const int s = p->tree[h];
  int cap = 16, n = 0;
  row_run *r = malloc(cap * sizeof(row_run));
  for (int hp = h / 2; hp >= 1; hp /= 2) {
    if ((which == 1 && hp != h / 2) || (which == 2 && hp == h / 2)) continue;
    const int b = BIDX(p, p->tree[hp], s);
    const chol_block *Bk = &p->blk[b];
    for (int q = 0; q < count[b]; q++) {
      const cholamd_filled *f = &snap[first[b] + q];
      const int64_t off = Bk->off + (f->lo_x - Bk->lo_x);
      const int m = f->hi_x - f->lo_x + 1;
      if (n > 0 && r[n - 1].off + r[n - 1].m == off) { r[n - 1].m += m; continue; }
      if (n == cap) { cap *= 2; r = realloc(r, cap * sizeof(row_run)); }
      r[n].off = off; r[n].m = m; n++;
    }
  }
  *out = r;
  return n;
}
static int ancestor_runs(const plan_t *p, int h, const cholamd_filled *snap, const int64_t *first, const int *count, row_run **out)
{
  return ancestor_runs_of(p, h, 0, snap, first, count, out);
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
  int fused_last = -1; /* the level's last fused launch, if nothing was launched after it */
  int fuse = opts->fuse; /* POTRF + TRSM of a step in one launch, if every block fits its TRSM role */
  for (int q = 0; q < nh; q++) if (pivot_block_width(opts, p->sep_size[p->tree[hs[q]]]) > CHOL_FUSE_MAXN) fuse = 0;
  /* pivots: small ones whole in step 0; big ones in pivot_block_width()-column blocks, each step =
   * POTRF of the diagonal block, TRSM of every row below it (rows of the pivot and filled ancestor rows
   * alike), rank-nb update of the remaining columns of those rows */
  for (int st = 0; st < steps; st++) {
    const int p0 = w->n_potrf, t0 = w->n_trsm, k0 = w->n_task, km0 = w->n_task_mt;
    for (int q = 0; q < nh; q++) {
      const int h = hs[q], s = p->tree[h], n = p->sep_size[s], ld = p->panel_ld[s];
      const int bw = pivot_block_width(opts, n);
      const int c0 = st * bw;
      if (c0 >= n) continue;
      const int nb = n - c0 < bw ? n - c0 : bw;
      const int64_t diag = p->panel_off[s] + c0 + (int64_t)c0 * ld;          /* element (c0, c0) of the pivot */
      const int64_t dinv = p->dinv_off[s] + (int64_t)(c0 / CHOL_NB) * CHOL_NB * CHOL_NB;
      chol_potrf_desc pd = { diag, dinv, nb, ld, s, c0 };
      push_potrf(B, pd);
      const int64_t colbase = (int64_t)c0 * ld;                              /* column c0 of the panel */
      const int below = n - c0 - nb;                                         /* pivot rows under the diagonal block */
      const int flag = w->n_potrf - 1 - p0; /* this block's POTRF descriptor within the step */
      if (below > 0) push_trsm_run(B, diag, dinv, p->panel_off[s] + (c0 + nb) + colbase, nb, ld, below, flag);
      row_run *runs; const int nr = ancestor_runs(p, h, snap, first, count, &runs);
      for (int r = 0; r < nr; r++) push_trsm_run(B, diag, dinv, runs[r].off + colbase, nb, ld, runs[r].m, flag);
      if (fuse) pad_trsm_group(B, t0, 3, diag, dinv, diag, nb, ld, flag);
      else if (nb <= CHOL_TRSM_W_MAXN) pad_trsm_group(B, t0, 4, diag, dinv, diag, nb, ld, flag);
      if (below > 0) { /* trailing columns [c0+nb, n): lower triangle of the pivot rows, everything of the ancestor rows */
        const int64_t x_piv = p->panel_off[s] + (c0 + nb) + colbase;         /* X rows = solved pivot rows, k = nb */
        chol_upd_src sp = { x_piv, x_piv, ld, ld, nb, 0 };
        const int sidx = push_src(B, sp);
        push_tasks(B, p->panel_off[s] + (c0 + nb) + (int64_t)(c0 + nb) * ld, ld, below, below, 1, sidx, sidx + 1);
        for (int r = 0; r < nr; r++) {
          chol_upd_src sa = { runs[r].off + colbase, x_piv, ld, ld, nb, 0 };
          const int si = push_src(B, sa);
          push_tasks(B, runs[r].off + (int64_t)(c0 + nb) * ld, ld, runs[r].m, below, 0, si, si + 1);
        }
      }
      free(runs);
    }
    flush_targets(B);
    if (fuse) { /* one launch: the strips follow their pivot's POTRF column by column, the 16x16 update tasks of the
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
      push_phase(B, wide ? 1 : 4, t0, w->n_trsm - t0);
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
            u->c_off = Bc->off + crow + (int64_t)ccol * Bc->ld; u->ldc = Bc->ld;
            u->a_off = Ba->off + (fa->lo_x - Ba->lo_x); u->lda = Ba->ld;
            u->b_off = Bb->off + (fb_->lo_x - Bb->lo_x); u->ldb = Bb->ld;
            u->m = fa->hi_x - fa->lo_x + 1; u->n = fb_->hi_x - fb_->lo_x + 1; u->k = n;
            u->syrk = (gp == par && fb_->cluster == fa->cluster);
            u->bc = bc; u->crow = crow; u->ccol = ccol;
          }
        }
      }
    }
  }
  qsort(tu, ntu, sizeof(upd_tuple), cmp_tuple);
  {
    const int k0 = w->n_task, km0 = w->n_task_mt;
    if (tuples_are_small(opts, tu, ntu)) emit_cell_tasks(B, p, tu, ntu);
    else
    for (int i = 0; i < ntu;) {
      int e = i + 1;
      while (e < ntu && tu[e].key == tu[i].key) e++;
      const int sb = w->n_src;
      for (int q = i; q < e; q++) {
        chol_upd_src sd = { tu[q].a_off, tu[q].b_off, tu[q].lda, tu[q].ldb, tu[q].k, 0 };
        push_src(B, sd);
      }
      push_tasks(B, tu[i].c_off, tu[i].ldc, tu[i].m, tu[i].n, tu[i].syrk, sb, w->n_src);
      i = e;
    }
    flush_targets(B);
    /* the 16x16 tasks of the extend-add ride in the level's last fused launch when that one carries no tasks of
     * its own and nothing was launched after it */
    if (fused_last >= 0 && fused_last == w->n_phase - 1 && w->phase[fused_last].n3 == 0 && w->n_task - k0 <= opts->fuse_update_max) {
      w->phase[fused_last].first3 = k0;
      w->phase[fused_last].n3 = w->n_task - k0;
    } else push_phase(B, 2, k0, w->n_task - k0);
    push_phase(B, 3, km0, w->n_task_mt - km0);
  }
  free(tu); free(first); free(count); free(hs); free(B->pend); B->pend = NULL; B->cap_pend = 0;
  return 0;
}

void chol_level_work_free(chol_level_work *w)
{
  free(w->potrf); free(w->trsm); free(w->task); free(w->task_mt); free(w->src); free(w->phase);
  memset(w, 0, sizeof *w);
}

/* ---------------------------------------------------------------------------------------- */
/* solve phase lists (mmat.rg:1394-1479), level by level                                      */
/*   forward  (bottom-up): TRSV per separator, then target-centric GEMV into every ancestor   */
/*   backward (top-down) : per separator gather from all ancestors (GEMV Trans), then TRSV^T  */
/* ---------------------------------------------------------------------------------------- */
int chol_build_solve_level(const plan_t *p, int level, chol_solve_level *w)
{
  memset(w, 0, sizeof *w);
  const int h0 = 1 << level, h1 = (1 << (level + 1)) - 1, cnt = h1 - h0 + 1;
  w->trsv = malloc(cnt * sizeof(chol_trsv_desc));
  w->bw_start = malloc((cnt + 1) * sizeof(int));
  w->bw = malloc((size_t)(cnt * (level > 0 ? level : 1)) * sizeof(chol_gemv_desc));
  for (int h = h0; h <= h1; h++) {
    int s = p->tree[h];
    chol_trsv_desc t = { p->panel_off[s], p->sep_size[s], p->panel_ld[s], p->sep_off[s], s, p->dinv_off[s] };
    if (p->sep_size[s] > w->max_n) w->max_n = p->sep_size[s];
    w->bw_start[w->n_trsv] = w->n_bw;
    w->trsv[w->n_trsv++] = t;
    for (int hp = h / 2; hp >= 1; hp /= 2) {
      int par = p->tree[hp];
      const chol_block *B = &p->blk[BIDX(p, par, s)];
      if (B->rows == 0 || B->cols == 0) continue;
      chol_gemv_desc g = { B->off, B->rows, B->cols, B->ld, p->sep_off[par], p->sep_off[s] };
      w->bw[w->n_bw++] = g;
    }
  }
  w->bw_start[w->n_trsv] = w->n_bw;
  /* forward: for every ancestor separator `par` (levels above), chunks of 256 rows */
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
          const chol_block *B = &p->blk[BIDX(p, par, s)];
          if (B->rows == 0 || B->cols == 0) continue;
          if (w->n_fw == cap) { cap *= 2; w->fw = realloc(w->fw, cap * sizeof(chol_gemv_desc)); }
          chol_gemv_desc g = { B->off, B->rows, B->cols, B->ld, p->sep_off[s], p->sep_off[par] };
          w->fw[w->n_fw++] = g;
        }
      }
    }
  w->grp_start[w->n_grp] = w->n_fw;
  /* row chunks of the blocks for the source-centric kernels of the driver-level solve */
  for (int pass = 0; pass < 2; pass++) {
    const int rows = pass ? CHOL_SOLVE_BW_ROWS : CHOL_SOLVE_FW_ROWS;
    int n = 0;
    for (int i = 0; i < w->n_bw; i++) n += (w->bw[i].m + rows - 1) / rows;
    int *it = malloc((size_t)(n > 0 ? 2 * n : 2) * sizeof(int)), k = 0;
    for (int i = 0; i < w->n_bw; i++)
      for (int r0 = 0; r0 < w->bw[i].m; r0 += rows) { it[2 * k] = i; it[2 * k + 1] = r0; k++; }
    if (pass) { w->ibw = it; w->n_ibw = n; } else { w->ifw = it; w->n_ifw = n; }
  }
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
