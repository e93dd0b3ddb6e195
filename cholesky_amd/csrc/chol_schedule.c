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

int chol_build_level_work(const plan_t *p, int level, int rank, int world, chol_level_work *w)
{
  memset(w, 0, sizeof *w);
  w->level = level;
  const int L = p->levels, lbl = L - 1 - level;
  const int d = chol_split_level(world);
  if (world < 1 || (1 << d) != world || d > L - 1 || rank < 0 || rank >= world) { chol_set_error("bad partition rank %d of %d", rank, world); return CHOLAMD_ERR_ARG; }
  const int h0 = 1 << level, h1 = (1 << (level + 1)) - 1;
  int64_t *first = malloc(p->nblk * sizeof(int64_t));
  int *count = malloc(p->nblk * sizeof(int));
  index_snapshot(p, lbl, first, count);
  const cholamd_filled *snap = p->snap[lbl];

  int cap_p = h1 - h0 + 1, cap_t = 64, cap_u = 256;
  w->potrf = malloc(cap_p * sizeof(chol_potrf_desc));
  w->trsm = malloc(cap_t * sizeof(chol_trsm_desc));
  upd_tuple *tu = malloc(cap_u * sizeof(upd_tuple));
  int ntu = 0;
  int64_t seq = 0;

  for (int h = h0; h <= h1; h++) {
    int s = p->tree[h];
    if (level >= d && chol_owner_of(p, s, world) != rank) continue;
    int bs = BIDX(p, s, s);
    if (count[bs] == 0 || p->sep_size[s] == 0) continue;
    const int n = p->sep_size[s], ld = p->panel_ld[s];
    chol_potrf_desc pd = { p->panel_off[s], p->dinv_off[s], n, ld, s, 0 };
    w->potrf[w->n_potrf++] = pd;
    /* TRSM row runs over all ancestor blocks of panel(s): tiles come in increasing row order */
    {
      int64_t run_off = 0; int run_m = 0;
      for (int hp = h / 2;; hp /= 2) {
        int nb_tiles = 0; int64_t fb = 0; const chol_block *B = NULL;
        if (hp >= 1) { int b = BIDX(p, p->tree[hp], s); B = &p->blk[b]; nb_tiles = count[b]; fb = first[b]; }
        for (int q = 0; q < nb_tiles + (hp == 0); q++) {
          int64_t off = -1; int m = 0;
          if (hp >= 1) {
            const cholamd_filled *f = &snap[fb + q];
            off = B->off + (f->lo_x - B->lo_x);
            m = f->hi_x - f->lo_x + 1;
            if (run_m > 0 && off == run_off + run_m) { run_m += m; continue; }
          }
          for (int r0 = 0; r0 < run_m; r0 += CHOL_TRSM_ROWS) { /* flush the finished run in chunks */
            if (w->n_trsm == cap_t) { cap_t *= 2; w->trsm = realloc(w->trsm, cap_t * sizeof(chol_trsm_desc)); }
            int mm = run_m - r0 < CHOL_TRSM_ROWS ? run_m - r0 : CHOL_TRSM_ROWS;
            chol_trsm_desc td = { p->panel_off[s], p->dinv_off[s], run_off + r0, n, ld, mm, ld };
            w->trsm[w->n_trsm++] = td;
          }
          run_off = off; run_m = m;
        }
        if (hp == 0) break;
      }
    }
    /* update tuples in program order: par bottom-up, gp from par to the root, tiles i, j */
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
          }
        }
      }
    }
  }
  /* group by target tile, keeping program order inside a group */
  qsort(tu, ntu, sizeof(upd_tuple), cmp_tuple);
  int ngroups = 0, ntask = 0;
  for (int i = 0; i < ntu; i++)
    if (i == 0 || tu[i].key != tu[i - 1].key) {
      ngroups++;
      int tr = (tu[i].m + 15) / 16, tc = (tu[i].n + 15) / 16;
      ntask += tu[i].syrk ? tr * (tr + 1) / 2 : tr * tc;
    }
  w->src = malloc((ntu > 0 ? ntu : 1) * sizeof(chol_upd_src));
  w->task = malloc((ntask > 0 ? ntask : 1) * sizeof(chol_upd_task));
  w->n_src = ntu;
  for (int i = 0; i < ntu; i++) {
    chol_upd_src sd = { tu[i].a_off, tu[i].b_off, tu[i].lda, tu[i].ldb, tu[i].k, 0 };
    w->src[i] = sd;
  }
  for (int i = 0; i < ntu;) {
    int e = i + 1;
    while (e < ntu && tu[e].key == tu[i].key) e++;
    const upd_tuple *g = &tu[i];
    int tr = (g->m + 15) / 16, tc = (g->n + 15) / 16;
    for (int a = 0; a < tr; a++)
      for (int b = 0; b < tc; b++) {
        if (g->syrk && b > a) continue;
        chol_upd_task *t = &w->task[w->n_task++];
        t->c_off = g->c_off + a * 16 + (int64_t)b * 16 * g->ldc;
        t->ldc = g->ldc;
        t->mv = (short)(g->m - a * 16 < 16 ? g->m - a * 16 : 16);
        t->nv = (short)(g->n - b * 16 < 16 ? g->n - b * 16 : 16);
        t->lower = (g->syrk && a == b);
        t->src_begin = i; t->src_end = e;
        t->ar = a * 16; t->br = b * 16;
      }
    i = e;
  }
  (void)ngroups;
  free(tu); free(first); free(count);
  return 0;
}

void chol_level_work_free(chol_level_work *w)
{
  free(w->potrf); free(w->trsm); free(w->task); free(w->src);
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
    chol_trsv_desc t = { p->panel_off[s], p->sep_size[s], p->panel_ld[s], p->sep_off[s], s };
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
  return 0;
}

void chol_solve_level_free(chol_solve_level *w)
{
  free(w->trsv); free(w->fw); free(w->grp_start); free(w->grp_rows); free(w->bw); free(w->bw_start);
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
  int rc = chol_build_level_work(p, level, rank, world, &w);
  if (rc) return rc;
  out[0] = w.n_potrf; out[1] = w.n_trsm; out[2] = w.n_task; out[3] = w.n_src;
  chol_level_work_free(&w);
  return 0;
}
