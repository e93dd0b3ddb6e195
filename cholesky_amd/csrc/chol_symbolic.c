/* Symbolic phase of libcholamd (host, plain C): separator tree, block/panel layout, tile layout,
 * A -> panel scatter map, per-level fill prediction, the reference-order BLAS call list and the
 * device work lists derived from it.
 *
 * Specification = the reference's symbolic tasks (mmat.rg:299-499, 529-695, 834-849, 896-1028)
 * and its level schedule (mmat.rg:1227-1355); this is a re-design, not a transcription:
 *   * A is scattered from its COO entries through the inverse permutation (O(nnz)), instead of
 *     probing a hash table for every element of every block (mmat.rg:501-527, 576-609);
 *   * tile fill is kept as one byte-matrix per block at the block's current interval and
 *     propagated with outer products of the pivot column's fill vectors;
 *   * storage is one contiguous panel per separator (chol_plan.h), not an N x N region.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "chol_plan.h"

typedef struct cholamd_plan plan_t;

#define BIDX(p, r, c) ((p)->blk_index[(size_t)(r) * ((p)->nsep + 1) + (c)])

static int ilog2(int v) { int l = 0; while ((1 << (l + 1)) <= v) l++; return l; }

int chol_ntiles(const plan_t *p, int sep, int t)
{
  return t < p->cl[sep].n_int ? p->cl[sep].len[t] - 1 : -1;
}
const chol_block *chol_plan_block(const plan_t *p, int r, int c)
{
  int i = BIDX(p, r, c);
  return i < 0 ? NULL : &p->blk[i];
}
static int interval_of_level(const plan_t *p, int lvl)
{
  int t = p->levels - 2 - lvl; /* mmat.rg:1350-1354 */
  return t < 0 ? 0 : t;
}

/* ---------------------------------------------------------------------------------------- */
/* per-block fill state                                                                       */
/* ---------------------------------------------------------------------------------------- */
typedef struct {
  int t;             /* interval the flags refer to; -1 = wiped */
  int nr, nc;
  unsigned char *f;  /* nr*nc, 1 = filled */
} fillmat;

static void fill_resize(fillmat *m, int t, int nr, int nc)
{
  free(m->f);
  m->t = t; m->nr = nr; m->nc = nc;
  m->f = calloc((size_t)(nr > 0 ? nr : 1) * (nc > 0 ? nc : 1), 1);
}

static int tile_of(const chol_clusters *c, int t, int off)
{ /* index i with start[t][i] <= off < start[t][i+1] */
  int lo = 0, hi = c->len[t] - 1;
  while (hi - lo > 1) {
    int mid = (lo + hi) / 2;
    if (c->start[t][mid] <= off) lo = mid; else hi = mid;
  }
  return lo;
}

static void push_op(plan_t *p, int op, int level, int m, int n, int k, int ax, int ay, int az, int bx, int by, int bz, int cx, int cy, int cz)
{
  if (p->nops == p->cap_ops) {
    p->cap_ops = p->cap_ops ? 2 * p->cap_ops : 4096;
    p->ops = realloc(p->ops, (size_t)p->cap_ops * sizeof(cholamd_op));
  }
  cholamd_op o = { op, level, m, n, k, ax, ay, az, bx, by, bz, cx, cy, cz };
  p->ops[p->nops++] = o;
  double f = 0;
  switch (op) { /* POTRF n^3/3, TRSM m n^2, SYRK n(n+1)k, GEMM 2mnk */
    case 0: f = (double)n * n * n / 3.0; break;
    case 1: f = (double)m * n * n; break;
    case 2: f = (double)n * (n + 1) * k; break;
    case 3: f = 2.0 * m * n * k; break;
  }
  if (level < 16) { p->calls[level][op]++; p->flops[level][op] += f; }
}

static void tile_bounds(const plan_t *p, const chol_block *b, int t, int row, int col, int *lo_x, int *lo_y, int *hi_x, int *hi_y)
{
  const chol_clusters *cr = &p->cl[b->r], *cc = &p->cl[b->c];
  *lo_x = b->lo_x + cr->start[t][row];
  *hi_x = b->lo_x + cr->start[t][row + 1] - 1;
  *lo_y = b->lo_y + cc->start[t][col];
  *hi_y = b->lo_y + cc->start[t][col + 1] - 1;
}

/* ---------------------------------------------------------------------------------------- */
/* Scalar symbolic factorisation of P A P^T (elimination tree + row-subtree traversal): exact   */
/* nnz(L) and sum of squared column counts, the algorithmic-work definitions of SURVEY 8d.      */
/* ---------------------------------------------------------------------------------------- */
static void scalar_symbolic(plan_t *p, const int *px, const int *py)
{
  const int n = p->n;
  const int64_t nnz = p->nnz_a;
  int64_t *rowptr = calloc((size_t)n + 2, sizeof(int64_t));
  int *cols = malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(int));
  for (int64_t e = 0; e < nnz; e++) if (px[e] != py[e]) rowptr[px[e] + 2]++;
  for (int i = 0; i < n; i++) rowptr[i + 2] += rowptr[i + 1];
  for (int64_t e = 0; e < nnz; e++) if (px[e] != py[e]) cols[rowptr[px[e] + 1]++] = py[e];
  /* rowptr[i] .. rowptr[i+1] now delimit the strictly-lower entries of row i */
  int *parent = malloc(n * sizeof(int)), *anc = malloc(n * sizeof(int)), *mark = malloc(n * sizeof(int));
  int64_t *cc = calloc(n, sizeof(int64_t));
  for (int i = 0; i < n; i++) { /* Liu's elimination tree with path compression */
    parent[i] = -1; anc[i] = -1;
    for (int64_t q = rowptr[i]; q < rowptr[i + 1]; q++) {
      int j = cols[q];
      while (j != -1 && j < i) {
        int nx = anc[j];
        anc[j] = i;
        if (nx == -1) parent[j] = i;
        j = nx;
      }
    }
  }
  for (int i = 0; i < n; i++) mark[i] = -1;
  for (int i = 0; i < n; i++) { /* row i of L = union of etree paths from its entries up to i */
    mark[i] = i;
    cc[i]++; /* diagonal */
    for (int64_t q = rowptr[i]; q < rowptr[i + 1]; q++)
      for (int j = cols[q]; j != -1 && j < i && mark[j] != i; j = parent[j]) { mark[j] = i; cc[j]++; }
  }
  p->nnz_l = 0; p->fmin = 0.0;
  for (int i = 0; i < n; i++) { p->nnz_l += cc[i]; p->fmin += (double)cc[i] * (double)cc[i]; }
  free(rowptr); free(cols); free(parent); free(anc); free(mark); free(cc);
}

static int cmp_dv(const void *x, const void *y)
{
  const int64_t a = *(const int64_t *)x, b = *(const int64_t *)y;
  return a < b ? -1 : (a > b);
}

/* ---------------------------------------------------------------------------------------- */
/* Arena layout: one column-major panel per separator, by ascending label: the separator's own rows, then bottom-up the rows of every
 * ancestor block -- all of them for the parent (a follower reads its children's strips of its whole diagonal-block height), the
 * kept 16-row tiles for the ancestors above it (keep[b], from the fill analysis; NULL entries / CHOLAMD_COMPACT=0: all rows). */
static void layout_panels(plan_t *p, unsigned char **keep)
{
  const int ns = p->nsep;
  const char *e = getenv("CHOLAMD_COMPACT");
  p->compact = !(e && e[0] == '0');
  int64_t off = 0, ws = 0, dense = 0;
  for (int c = 1; c <= ns; c++) {
    int rows = 0, rows_dense = 0;
    for (int h = p->heap_of[c]; h >= 1; h /= 2) {
      chol_block *b = &p->blk[BIDX(p, p->tree[h], c)];
      const int T = (b->rows + CHOL_NB - 1) / CHOL_NB;
      rows_dense += b->rows;
      if (p->compact && keep[BIDX(p, p->tree[h], c)]) {
        const unsigned char *k = keep[BIDX(p, p->tree[h], c)];
        b->tmap = malloc((size_t)(T > 0 ? T : 1) * sizeof(int));
        b->crows = 0;
        int nk = 0;
        for (int t = 0; t < T; t++) {
          b->tmap[t] = k[t] ? nk++ : -1;
          if (k[t]) b->crows += b->rows - t * CHOL_NB < CHOL_NB ? b->rows - t * CHOL_NB : CHOL_NB;
        }
      }
      b->off = rows; /* relative to the panel for now */
      rows += b->crows;
    }
    p->panel_rows[c] = rows;
    p->panel_ld[c] = (rows + 3) & ~3; /* 32-byte aligned columns */
    p->panel_off[c] = off;
    for (int h = p->heap_of[c]; h >= 1; h /= 2) {
      chol_block *b = &p->blk[BIDX(p, p->tree[h], c)];
      b->off += off; b->ld = p->panel_ld[c];
    }
    off += (int64_t)p->panel_ld[c] * p->sep_size[c];
    off = (off + 15) & ~(int64_t)15; /* 128-byte aligned panels */
    dense += (int64_t)((rows_dense + 3) & ~3) * p->sep_size[c];
    dense = (dense + 15) & ~(int64_t)15;
    p->dinv_off[c] = ws;
    ws += (int64_t)((p->sep_size[c] + CHOL_NB - 1) / CHOL_NB) * CHOL_NB * CHOL_NB;
  }
  p->arena = off; p->ws_doubles = ws; p->arena_dense = dense;
}

int chol_plan_finish(plan_t *p, int nz, const int *a_row, const int *a_col, const double *a_val)
{
  const int ns = p->nsep, L = p->levels, n = p->n;
  /* tree: heap index i (level floor(log2 i)) carries label nsep - (i - 1)   (mmat.rg:834-849) */
  p->tree = calloc(ns + 2, sizeof(int));
  p->heap_of = calloc(ns + 2, sizeof(int));
  p->level_of = calloc(ns + 2, sizeof(int));
  for (int i = 1; i <= ns; i++) {
    int lab = ns - (i - 1);
    p->tree[i] = lab; p->heap_of[lab] = i; p->level_of[lab] = ilog2(i);
  }
  /* cluster invariants the level schedule relies on */
  for (int s = 1; s <= ns; s++) {
    chol_clusters *c = &p->cl[s];
    int need = interval_of_level(p, p->level_of[s]);
    if (c->n_int < need + 1) { chol_set_error("separator %d has %d intervals, needs %d", s, c->n_int, need + 1); return CHOLAMD_ERR_INVARIANT; }
    if (c->len[need] != 2) { chol_set_error("separator %d must be a single tile at interval %d", s, need); return CHOLAMD_ERR_INVARIANT; }
    c->start = calloc(c->n_int, sizeof(int *));
    for (int t = 0; t < c->n_int; t++) {
      c->start[t] = malloc(c->len[t] * sizeof(int));
      for (int i = 0; i < c->len[t]; i++) {
        int v = c->raw[t][i];
        for (int u = t - 1; u >= 0; u--) { /* chained boundaries, mmat.rg:400-422 */
          if (v < 0 || v >= c->len[u]) { chol_set_error("separator %d interval %d: boundary index out of range", s, t); return CHOLAMD_ERR_FORMAT; }
          v = c->raw[u][v];
        }
        c->start[t][i] = v;
        if (i > 0 && v <= c->start[t][i - 1]) { chol_set_error("separator %d interval %d: boundaries not increasing", s, t); return CHOLAMD_ERR_FORMAT; }
      }
      if (c->start[t][0] != 0 || c->start[t][c->len[t] - 1] != p->sep_size[s]) {
        chol_set_error("separator %d interval %d: boundaries do not span the separator", s, t);
        return CHOLAMD_ERR_FORMAT;
      }
    }
  }
  /* panels + blocks.  (r,c) allocated iff r == c or r is a proper ancestor of c (mmat.rg:740-767) */
  p->panel_off = calloc(ns + 2, sizeof(int64_t));
  p->panel_ld = calloc(ns + 2, sizeof(int));
  p->panel_rows = calloc(ns + 2, sizeof(int));
  p->dinv_off = calloc(ns + 2, sizeof(int64_t));
  p->blk_index = malloc((size_t)(ns + 1) * (ns + 1) * sizeof(int));
  for (size_t i = 0; i < (size_t)(ns + 1) * (ns + 1); i++) p->blk_index[i] = -1;
  p->nblk = 0;
  for (int c = 1; c <= ns; c++) p->nblk += p->level_of[c] + 1;
  p->blk = calloc(p->nblk, sizeof(chol_block));
  { /* block geometry; the arena layout follows the fill analysis (layout_panels) */
    int k = 0;
    for (int r = 1; r <= ns; r++)
      for (int c = 1; c <= r; c++) {
        /* r ancestor-or-self of c ? */
        int hr = p->heap_of[r], hc = p->heap_of[c], dl = p->level_of[c] - p->level_of[r];
        if (dl < 0 || (hc >> dl) != hr) continue;
        chol_block *b = &p->blk[k];
        b->r = r; b->c = c;
        b->lo_x = p->sep_off[r]; b->hi_x = p->sep_off[r] + p->sep_size[r] - 1;
        b->lo_y = p->sep_off[c]; b->hi_y = p->sep_off[c] + p->sep_size[c] - 1;
        b->rows = p->sep_size[r]; b->cols = p->sep_size[c];
        b->tmap = NULL; b->crows = b->rows;
        BIDX(p, r, c) = k++;
      }
    if (k != p->nblk) { chol_set_error("internal: block count mismatch"); return CHOLAMD_ERR_INVARIANT; }
  }
  /* scatter map of tril(A) and initial (interval-0) fill  (fill_block, mmat.rg:529-633) */
  fillmat *F = calloc(p->nblk, sizeof(fillmat));
  for (int b = 0; b < p->nblk; b++) { F[b].f = NULL; fill_resize(&F[b], 0, chol_ntiles(p, p->blk[b].r, 0), chol_ntiles(p, p->blk[b].c, 0)); }
  p->a_dst = malloc((size_t)(nz > 0 ? nz : 1) * sizeof(int64_t));
  p->a_val = malloc((size_t)(nz > 0 ? nz : 1) * sizeof(double));
  p->nnz_a = 0; p->dropped = 0;
  int *px = malloc((size_t)(nz > 0 ? nz : 1) * sizeof(int)), *py = malloc((size_t)(nz > 0 ? nz : 1) * sizeof(int));
  unsigned char **keep = calloc(p->nblk, sizeof(unsigned char *)); /* per block: the 16-row tiles the panel must store */
  for (int e = 0; e < nz; e++) {
    int i = a_row[e], j = a_col[e];
    if (i < 0 || j < 0 || i >= n || j >= n) {
      chol_set_error("matrix entry %d out of range", e);
      for (int b = 0; b < p->nblk; b++) free(F[b].f);
      free(F); free(px); free(py); free(keep); /* a_dst / a_val belong to the plan: cholamd_plan_destroy frees them */
      return CHOLAMD_ERR_FORMAT;
    }
    if (a_val[e] == 0.0) continue; /* explicit zeros are invisible to the reference (mnd.c:168-195) */
    int pi = p->iperm[i], pj = p->iperm[j];
    int x = pi > pj ? pi : pj, y = pi > pj ? pj : pi;
    int bi = BIDX(p, p->sep_of_pos[x], p->sep_of_pos[y]);
    if (bi < 0) { p->dropped++; continue; } /* not in an ancestor/descendant block: the ordering is not a valid ND */
    const chol_block *b = &p->blk[bi];
    p->a_dst[p->nnz_a] = bi; /* the block for now: the arena offset once the panels are laid out */
    p->a_val[p->nnz_a] = a_val[e];
    px[p->nnz_a] = x; py[p->nnz_a] = y;
    p->nnz_a++;
    int tr = tile_of(&p->cl[b->r], 0, x - b->lo_x), tc = tile_of(&p->cl[b->c], 0, y - b->lo_y);
    F[bi].f[(size_t)tr * F[bi].nc + tc] = 1;
  }
  scalar_symbolic(p, px, py);
  { /* A as a full symmetric CSR in ORIGINAL dof order (both triangles, explicit zeros skipped, entries the schedule
     * dropped included): the operator of the fp64 residual r = b - A x of the iterative refinement */
    int64_t *ptr = calloc((size_t)n + 1, sizeof(int64_t));
    for (int e = 0; e < nz; e++) {
      if (a_val[e] == 0.0) continue;
      ptr[a_row[e] + 1]++;
      if (a_row[e] != a_col[e]) ptr[a_col[e] + 1]++;
    }
    for (int i = 0; i < n; i++) ptr[i + 1] += ptr[i];
    p->csr_col = malloc((size_t)(ptr[n] > 0 ? ptr[n] : 1) * sizeof(int));
    p->csr_val = malloc((size_t)(ptr[n] > 0 ? ptr[n] : 1) * sizeof(double));
    int64_t *fillp = malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
    memcpy(fillp, ptr, (size_t)n * sizeof(int64_t));
    for (int e = 0; e < nz; e++) {
      if (a_val[e] == 0.0) continue;
      const int i = a_row[e], j = a_col[e];
      p->csr_col[fillp[i]] = j; p->csr_val[fillp[i]++] = a_val[e];
      if (i != j) { p->csr_col[fillp[j]] = i; p->csr_val[fillp[j]++] = a_val[e]; }
    }
    free(fillp);
    p->csr_ptr = ptr;
  }
  /* per-level fill prediction, snapshots and the reference-order call list
   * (compute_filled_clusters mmat.rg:896-1028 interleaved with the schedule mmat.rg:1227-1355) */
  p->snap_n = calloc(L, sizeof(int64_t));
  p->snap = calloc(L, sizeof(cholamd_filled *));
  p->nnz_tiles = 0;
  int t = 0, lbl = 0;
  for (int lvl = L - 1; lvl >= 0; lvl--, lbl++) {
    const int h0 = 1 << lvl, h1 = (1 << (lvl + 1)) - 1;
    /* the panels of this level are final in structure now: the rows a panel stores for an ancestor above its parent are the 16-row
     * tiles its filled tiles touch (every later access -- strips, update sources, solves -- stays inside them) */
    for (int h = h0; h <= h1; h++)
      for (int hp = h / 4; hp >= 1; hp /= 2) {
        const int b = BIDX(p, p->tree[hp], p->tree[h]);
        const fillmat *Fb = &F[b];
        const chol_clusters *cr = &p->cl[p->tree[hp]];
        keep[b] = calloc((size_t)(p->blk[b].rows + CHOL_NB - 1) / CHOL_NB + 1, 1);
        if (Fb->t != t) continue;
        for (int i = 0; i < Fb->nr; i++) {
          int any = 0;
          for (int j = 0; j < Fb->nc; j++) any |= Fb->f[(size_t)i * Fb->nc + j];
          if (!any || cr->start[t][i + 1] <= cr->start[t][i]) continue;
          for (int q = cr->start[t][i] / CHOL_NB; q <= (cr->start[t][i + 1] - 1) / CHOL_NB; q++) keep[b][q] = 1;
        }
      }
    /* symbolic update of the ancestors' blocks */
    for (int h = h0; h <= h1; h++) {
      int s = p->tree[h];
      if (chol_ntiles(p, s, t) != 1) {
        chol_set_error("separator %d not a single tile when eliminated", s);
        for (int b = 0; b < p->nblk; b++) { free(F[b].f); free(keep[b]); }
        free(F); free(keep); free(px); free(py);
        return CHOLAMD_ERR_INVARIANT;
      }
      for (int hp = h / 2; hp >= 1; hp /= 2) {
        int par = p->tree[hp];
        const fillmat *Fb = &F[BIDX(p, par, s)];
        for (int hg = hp; hg >= 1; hg /= 2) {
          int gp = p->tree[hg];
          const fillmat *Fa = &F[BIDX(p, gp, s)];
          fillmat *Fc = &F[BIDX(p, gp, par)];
          for (int i = 0; i < Fa->nr; i++) {
            if (!Fa->f[i]) continue;
            for (int j = 0; j < Fb->nr; j++) {
              if (!Fb->f[j]) continue;
              if (gp == par && j > i) continue;
              Fc->f[(size_t)i * Fc->nc + j] = 1;
            }
          }
        }
      }
    }
    /* snapshot `lbl`: every filled tile of every live block, with its rectangle at interval t */
    int64_t cnt = 0;
    for (int b = 0; b < p->nblk; b++)
      if (F[b].t == t)
        for (size_t z = 0; z < (size_t)F[b].nr * F[b].nc; z++) cnt += F[b].f[z];
    p->snap[lbl] = malloc((size_t)(cnt > 0 ? cnt : 1) * sizeof(cholamd_filled));
    p->snap_n[lbl] = cnt;
    cnt = 0;
    for (int b = 0; b < p->nblk; b++) {
      if (F[b].t != t) continue;
      const chol_block *B = &p->blk[b];
      for (int row = 0; row < F[b].nr; row++)
        for (int col = 0; col < F[b].nc; col++) {
          if (!F[b].f[(size_t)row * F[b].nc + col]) continue;
          cholamd_filled *q = &p->snap[lbl][cnt++];
          q->filled = 0; q->sep_x = B->r; q->sep_y = B->c; q->interval = lbl; q->cluster = row * F[b].nc + col;
          tile_bounds(p, B, t, row, col, &q->lo_x, &q->lo_y, &q->hi_x, &q->hi_y);
        }
    }
    /* the reference's BLAS calls of this level, in program order */
    for (int h = h0; h <= h1; h++) { /* fused_dpotrf sweep, mmat.rg:1240-1257 */
      int s = p->tree[h], ns_ = p->sep_size[s];
      if (F[BIDX(p, s, s)].f[0] && ns_ > 0) {
        push_op(p, 0, lvl, ns_, ns_, 0, s, s, 0, 0, 0, 0, 0, 0, 0);
        p->nnz_tiles += (int64_t)ns_ * (ns_ + 1) / 2;
      }
    }
    for (int h = h0; h <= h1; h++) { /* fused_dtrsm sweep, mmat.rg:1259-1291 */
      int s = p->tree[h];
      if (!F[BIDX(p, s, s)].f[0]) continue;
      for (int hp = h / 2; hp >= 1; hp /= 2) {
        int par = p->tree[hp];
        const fillmat *Fb = &F[BIDX(p, par, s)];
        const chol_clusters *cp = &p->cl[par];
        for (int j = 0; j < Fb->nr; j++)
          if (Fb->f[j]) {
            int m = cp->start[t][j + 1] - cp->start[t][j];
            push_op(p, 1, lvl, m, p->sep_size[s], 0, s, s, 0, par, s, j, 0, 0, 0);
            p->nnz_tiles += (int64_t)m * p->sep_size[s];
          }
      }
    }
    for (int h = h0; h <= h1; h++) { /* fused_dsyrk / fused_dgemm sweep, mmat.rg:1293-1347 */
      int s = p->tree[h], k = p->sep_size[s];
      for (int hp = h / 2; hp >= 1; hp /= 2) {
        int par = p->tree[hp];
        const fillmat *Fb = &F[BIDX(p, par, s)];
        const chol_clusters *cp = &p->cl[par];
        int ccs = chol_ntiles(p, par, t);
        for (int hg = hp; hg >= 1; hg /= 2) {
          int gp = p->tree[hg];
          const fillmat *Fa = &F[BIDX(p, gp, s)];
          const chol_clusters *cg = &p->cl[gp];
          for (int i = 0; i < Fa->nr; i++) {
            if (!Fa->f[i]) continue;
            int m = cg->start[t][i + 1] - cg->start[t][i];
            for (int j = 0; j < Fb->nr; j++) {
              if (!Fb->f[j]) continue;
              int nn = cp->start[t][j + 1] - cp->start[t][j];
              if (gp == par) {
                if (j < i) push_op(p, 3, lvl, m, nn, k, gp, s, i, par, s, j, gp, par, i * ccs + j);
                else if (j == i) push_op(p, 2, lvl, m, m, k, gp, s, i, par, s, j, gp, par, i * ccs + j);
              } else {
                push_op(p, 3, lvl, m, nn, k, gp, s, i, par, s, j, gp, par, i * ccs + j);
              }
            }
          }
        }
      }
    }
    /* coarsen to the next interval (merge_filled_clusters, mmat.rg:635-695) */
    if (lvl <= L - 2) {
      t++;
      if (t < L) {
        for (int b = 0; b < p->nblk; b++) {
          const chol_block *B = &p->blk[b];
          int nr = chol_ntiles(p, B->r, t), nc = chol_ntiles(p, B->c, t);
          if (nr < 0 || nc < 0 || F[b].t != t - 1) { free(F[b].f); F[b].f = NULL; F[b].t = -1; F[b].nr = F[b].nc = 0; continue; }
          fillmat old = F[b];
          F[b].f = NULL;
          fill_resize(&F[b], t, nr, nc);
          const int *rr = p->cl[B->r].raw[t], *rc = p->cl[B->c].raw[t];
          for (int row = 0; row < nr; row++)
            for (int col = 0; col < nc; col++) {
              unsigned char any = 0;
              for (int i = rr[row]; i < rr[row + 1] && !any; i++)
                for (int j = rc[col]; j < rc[col + 1]; j++)
                  if (old.f[(size_t)i * old.nc + j]) { any = 1; break; }
              F[b].f[(size_t)row * nc + col] = any;
            }
          free(old.f);
        }
      }
    }
  }
  for (int b = 0; b < p->nblk; b++) free(F[b].f);
  free(F);
  layout_panels(p, keep);
  for (int b = 0; b < p->nblk; b++) free(keep[b]);
  free(keep);
  /* scatter map of tril(A) */
  for (int64_t e = 0; e < p->nnz_a; e++) {
    const chol_block *b = &p->blk[p->a_dst[e]];
    const int64_t row = chol_block_row(b, px[e] - b->lo_x);
    if (row < 0) { chol_set_error("internal: matrix entry (%d, %d) outside the stored rows of block (%d, %d)", px[e], py[e], b->r, b->c); free(px); free(py); return CHOLAMD_ERR_INVARIANT; }
    p->a_dst[e] = row + (int64_t)(py[e] - b->lo_y) * b->ld;
  }
  free(px); free(py);
  { /* ascending arena offsets: coalesced device scatter, and the entries of the shared top of the
     * tree (the tail of the arena) form a suffix that non-root ranks skip (multi-GPU fill) */
    typedef struct { int64_t d; double v; } dv_t;
    dv_t *t = malloc((size_t)(p->nnz_a > 0 ? p->nnz_a : 1) * sizeof(dv_t));
    for (int64_t e = 0; e < p->nnz_a; e++) { t[e].d = p->a_dst[e]; t[e].v = p->a_val[e]; }
    qsort(t, (size_t)p->nnz_a, sizeof(dv_t), cmp_dv);
    for (int64_t e = 0; e < p->nnz_a; e++) { p->a_dst[e] = t[e].d; p->a_val[e] = t[e].v; }
    free(t);
  }
  return 0;
}

/* ---------------------------------------------------------------------------------------- */
/* construction from files / arrays                                                           */
/* ---------------------------------------------------------------------------------------- */
static int setup_clusters(plan_t *p, const int *idx, const int *interval, const int *sep, int64_t count)
{
  const int ns = p->nsep;
  p->cl = calloc(ns + 2, sizeof(chol_clusters));
  /* count intervals and lengths */
  for (int64_t i = 0; i < count; i++) {
    int s = sep[i], t = interval[i];
    if (s < 1 || s > ns || t < 0 || t > 62) { chol_set_error("cluster triple %ld out of range", (long)i); return CHOLAMD_ERR_FORMAT; }
    if (t + 1 > p->cl[s].n_int) p->cl[s].n_int = t + 1;
  }
  for (int s = 1; s <= ns; s++) {
    chol_clusters *c = &p->cl[s];
    if (c->n_int < 1) { chol_set_error("separator %d has no cluster list", s); return CHOLAMD_ERR_FORMAT; }
    c->len = calloc(c->n_int, sizeof(int));
    c->raw = calloc(c->n_int, sizeof(int *));
  }
  for (int64_t i = 0; i < count; i++) p->cl[sep[i]].len[interval[i]]++;
  for (int s = 1; s <= ns; s++)
    for (int t = 0; t < p->cl[s].n_int; t++) {
      if (p->cl[s].len[t] < 2) { chol_set_error("separator %d interval %d has fewer than two boundaries", s, t); return CHOLAMD_ERR_FORMAT; }
      p->cl[s].raw[t] = malloc(p->cl[s].len[t] * sizeof(int));
      p->cl[s].len[t] = 0;
    }
  for (int64_t i = 0; i < count; i++) {
    chol_clusters *c = &p->cl[sep[i]];
    c->raw[interval[i]][c->len[interval[i]]++] = idx[i];
  }
  return 0;
}

static int setup_ordering(plan_t *p, const int *perm, const int *sep_of_pos)
{
  const int n = p->n, ns = p->nsep;
  if (ns != (1 << p->levels) - 1 || p->levels < 1 || p->levels > 30) { chol_set_error("num_separators %d != 2^%d - 1", ns, p->levels); return CHOLAMD_ERR_FORMAT; }
  p->perm = malloc(n * sizeof(int));
  p->iperm = malloc(n * sizeof(int));
  p->sep_of_pos = malloc(n * sizeof(int));
  p->sep_size = calloc(ns + 2, sizeof(int));
  p->sep_off = calloc(ns + 2, sizeof(int));
  for (int i = 0; i < n; i++) p->iperm[i] = -1;
  for (int q = 0; q < n; q++) {
    int d = perm[q], s = sep_of_pos[q];
    if (d < 0 || d >= n || p->iperm[d] != -1) { chol_set_error("ordering is not a permutation (position %d)", q); return CHOLAMD_ERR_FORMAT; }
    if (s < 1 || s > ns || (q > 0 && s < sep_of_pos[q - 1])) { chol_set_error("separator labels must be non-decreasing along the ordering (position %d)", q); return CHOLAMD_ERR_FORMAT; }
    p->perm[q] = d; p->iperm[d] = q; p->sep_of_pos[q] = s; p->sep_size[s]++;
  }
  int acc = 0;
  for (int s = 1; s <= ns; s++) { p->sep_off[s] = acc; acc += p->sep_size[s]; }
  return 0;
}

int cholamd_plan_create_from_arrays(int n, int levels, const int *perm, const int *sep_sizes,
                                    const int *cl_idx, const int *cl_interval, const int *cl_sep, int64_t cl_count,
                                    int64_t nz, const int *a_row, const int *a_col, const double *a_val,
                                    const char *banner, cholamd_plan **out)
{
  *out = NULL;
  if (nz > 2147483647) { chol_set_error("nz too large"); return CHOLAMD_ERR_ARG; }
  plan_t *p = calloc(1, sizeof(plan_t));
  p->n = n; p->nz_file = (int)nz; p->levels = levels; p->nsep = (1 << levels) - 1;
  snprintf(p->banner, sizeof p->banner, "%s", banner ? banner : "%%MatrixMarket matrix coordinate real symmetric");
  p->typecode[0] = 'M'; p->typecode[1] = 'C'; p->typecode[2] = 'R'; p->typecode[3] = strstr(p->banner, "hermitian") ? 'H' : 'S';
  int *sep_of_pos = malloc((n > 0 ? n : 1) * sizeof(int));
  int q = 0, rc = 0;
  for (int s = 1; s <= p->nsep && rc == 0; s++)
    for (int i = 0; i < sep_sizes[s - 1]; i++) {
      if (q >= n) { rc = CHOLAMD_ERR_FORMAT; break; }
      sep_of_pos[q++] = s;
    }
  if (rc || q != n) { free(sep_of_pos); cholamd_plan_destroy(p); chol_set_error("separator sizes do not sum to n"); return CHOLAMD_ERR_FORMAT; }
  rc = setup_ordering(p, perm, sep_of_pos);
  free(sep_of_pos);
  if (!rc) rc = setup_clusters(p, cl_idx, cl_interval, cl_sep, cl_count);
  if (!rc) rc = chol_plan_finish(p, (int)nz, a_row, a_col, a_val);
  if (rc) { cholamd_plan_destroy(p); return rc; }
  *out = p;
  return 0;
}

int cholamd_plan_create(const char *matrix_file, const char *separator_file, const char *clusters_file, cholamd_plan **out)
{
  *out = NULL;
  plan_t *p = calloc(1, sizeof(plan_t));
  int rc = 0, M = 0, N = 0, NZ = 0;
  int *idx = NULL, *sp = NULL, *ci = NULL, *ct = NULL, *cs = NULL, *ar = NULL, *ac = NULL;
  double *av = NULL;
  /* read_matrix_banner, mmat.rg:76-100 */
  FILE *fp = fopen(matrix_file, "r");
  if (!fp) { chol_set_error("cannot open matrix file %s: %s", matrix_file, strerror(errno)); rc = CHOLAMD_ERR_IO; goto done; }
  {
    char line[1025];
    if (fgets(line, sizeof line, fp)) {
      size_t l = strlen(line);
      while (l && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0;
      snprintf(p->banner, sizeof p->banner, "%s", line);
    }
    rewind(fp);
  }
  if (mm_read_banner(fp, &p->typecode) != 0) { fclose(fp); chol_set_error("%s: unable to read banner", matrix_file); rc = CHOLAMD_ERR_FORMAT; goto done; }
  if (mm_read_mtx_crd_size(fp, &M, &N, &NZ) != 0) { fclose(fp); chol_set_error("%s: unable to read matrix size", matrix_file); rc = CHOLAMD_ERR_FORMAT; goto done; }
  fclose(fp);
  if (M != N || M <= 0 || NZ < 0) { chol_set_error("%s: need a square matrix (M=%d N=%d nz=%d)", matrix_file, M, N, NZ); rc = CHOLAMD_ERR_FORMAT; goto done; }
  p->n = N; p->nz_file = NZ;
  idx = malloc(N * sizeof(int)); sp = malloc(N * sizeof(int));
  cholamd_sepinfo info;
  if ((rc = cholamd_read_separators(separator_file, N, idx, sp, &info)) != 0) goto done;
  p->levels = info.levels; p->nsep = info.num_separators;
  if ((rc = setup_ordering(p, idx, sp)) != 0) goto done;
  {
    int64_t count = 0;
    int r = cholamd_read_clusters(clusters_file, NULL, NULL, NULL, 0, &count);
    if (r < 0) { rc = r; goto done; }
    ci = malloc((count + 1) * sizeof(int)); ct = malloc((count + 1) * sizeof(int)); cs = malloc((count + 1) * sizeof(int));
    r = cholamd_read_clusters(clusters_file, ci, ct, cs, count, &count);
    if (r < 0) { rc = r; goto done; }
    p->max_int_size = r;
    if ((rc = setup_clusters(p, ci, ct, cs, count)) != 0) goto done;
  }
  ar = malloc((NZ + 1) * sizeof(int)); ac = malloc((NZ + 1) * sizeof(int)); av = malloc((NZ + 1) * sizeof(double));
  if ((rc = cholamd_read_matrix(matrix_file, NZ, ar, ac, av)) != 0) goto done;
  rc = chol_plan_finish(p, NZ, ar, ac, av);
done:
  free(idx); free(sp); free(ci); free(ct); free(cs); free(ar); free(ac); free(av);
  if (rc) { cholamd_plan_destroy(p); return rc; }
  *out = p;
  return 0;
}

void cholamd_plan_destroy(cholamd_plan *p)
{
  if (!p) return;
  if (p->cl)
    for (int s = 1; s <= p->nsep; s++) {
      chol_clusters *c = &p->cl[s];
      for (int t = 0; t < c->n_int; t++) { if (c->raw) free(c->raw[t]); if (c->start) free(c->start[t]); }
      free(c->raw); free(c->start); free(c->len);
    }
  if (p->snap) for (int l = 0; l < p->levels; l++) free(p->snap[l]);
  free(p->cl); free(p->snap); free(p->snap_n); free(p->perm); free(p->iperm); free(p->sep_of_pos); free(p->sep_size);
  if (p->blk) for (int b = 0; b < p->nblk; b++) free(p->blk[b].tmap);
  free(p->sep_off); free(p->tree); free(p->heap_of); free(p->level_of); free(p->blk); free(p->blk_index);
  free(p->panel_off); free(p->panel_ld); free(p->panel_rows); free(p->dinv_off); free(p->a_dst); free(p->a_val); free(p->ops);
  free(p->csr_ptr); free(p->csr_col); free(p->csr_val);
  free(p);
}

/* ---------------------------------------------------------------------------------------- */
/* queries                                                                                    */
/* ---------------------------------------------------------------------------------------- */
int cholamd_plan_n(const cholamd_plan *p) { return p->n; }
int cholamd_plan_nz(const cholamd_plan *p) { return p->nz_file; }
int cholamd_plan_levels(const cholamd_plan *p) { return p->levels; }
int cholamd_plan_num_separators(const cholamd_plan *p) { return p->nsep; }
int cholamd_plan_max_int_size(const cholamd_plan *p) { return p->max_int_size; }
int cholamd_plan_num_blocks(const cholamd_plan *p) { return p->nblk; }
int64_t cholamd_plan_arena_doubles(const cholamd_plan *p) { return p->arena; }
int64_t cholamd_plan_arena_dense_doubles(const cholamd_plan *p) { return p->arena_dense; }
int cholamd_plan_block_tile_map(const cholamd_plan *p, int r, int c, int *out)
{
  const chol_block *B = chol_plan_block(p, r, c);
  if (!B) return CHOLAMD_ERR_ARG;
  const int T = (B->rows + CHOL_NB - 1) / CHOL_NB;
  for (int t = 0; t < T; t++) out[t] = B->tmap ? B->tmap[t] : t;
  return T;
}
int64_t cholamd_plan_dropped_entries(const cholamd_plan *p) { return p->dropped; }
const char *cholamd_plan_banner(const cholamd_plan *p) { return p->banner; }
void cholamd_plan_perm(const cholamd_plan *p, int *out) { memcpy(out, p->perm, p->n * sizeof(int)); }
void cholamd_plan_sep_sizes(const cholamd_plan *p, int *out) { for (int s = 1; s <= p->nsep; s++) out[s - 1] = p->sep_size[s]; }
void cholamd_plan_sep_offsets(const cholamd_plan *p, int *out) { for (int s = 1; s <= p->nsep; s++) out[s - 1] = p->sep_off[s]; }
void cholamd_plan_tree(const cholamd_plan *p, int *out) { for (int i = 1; i <= p->nsep; i++) out[i - 1] = p->tree[i]; }
void cholamd_plan_blocks(const cholamd_plan *p, int *out)
{
  for (int b = 0; b < p->nblk; b++) {
    const chol_block *B = &p->blk[b];
    int *q = out + 9 * (size_t)b;
    q[0] = B->r; q[1] = B->c; q[2] = B->lo_x; q[3] = B->lo_y; q[4] = B->hi_x; q[5] = B->hi_y; q[6] = B->ld;
    q[7] = (int)(B->off & 0xffffffff); q[8] = (int)(B->off >> 32);
  }
}
int64_t cholamd_plan_snapshot_count(const cholamd_plan *p, int lbl) { return (lbl >= 0 && lbl < p->levels) ? p->snap_n[lbl] : 0; }
void cholamd_plan_snapshot(const cholamd_plan *p, int lbl, cholamd_filled *out)
{
  if (lbl >= 0 && lbl < p->levels) memcpy(out, p->snap[lbl], (size_t)p->snap_n[lbl] * sizeof(cholamd_filled));
}
int64_t cholamd_plan_num_ops(const cholamd_plan *p) { return p->nops; }
void cholamd_plan_ops(const cholamd_plan *p, cholamd_op *out) { memcpy(out, p->ops, (size_t)p->nops * sizeof(cholamd_op)); }
void cholamd_plan_counts(const cholamd_plan *p, int level, int64_t calls[4], double flops[4])
{
  for (int k = 0; k < 4; k++) { calls[k] = 0; flops[k] = 0; }
  for (int l = 0; l < p->levels && l < 16; l++)
    if (level < 0 || level == l)
      for (int k = 0; k < 4; k++) { calls[k] += p->calls[l][k]; flops[k] += p->flops[l][k]; }
}
double cholamd_plan_flops(const cholamd_plan *p)
{
  int64_t c[4]; double f[4];
  cholamd_plan_counts(p, -1, c, f);
  return f[0] + f[1] + f[2] + f[3];
}
int64_t cholamd_plan_nnz_a(const cholamd_plan *p) { return p->nnz_a; }
int64_t cholamd_plan_nnz_l(const cholamd_plan *p) { return p->nnz_l; }
int64_t cholamd_plan_nnz_tiles(const cholamd_plan *p) { return p->nnz_tiles; }
double cholamd_plan_fmin(const cholamd_plan *p) { return p->fmin; }
int64_t cholamd_plan_alg_bytes(const cholamd_plan *p) { return 8 * (p->nnz_a + p->nnz_l); }

int cholamd_plan_fill_host(const cholamd_plan *p, double *arena)
{
  memset(arena, 0, (size_t)p->arena * sizeof(double));
  for (int64_t e = 0; e < p->nnz_a; e++) arena[p->a_dst[e]] = p->a_val[e];
  return 0;
}

/* element (i, j) of a block; rows without storage are the zeros of the reference's dense block */
static double block_value(const chol_block *B, const double *arena, int i, int j)
{
  const int64_t row = chol_block_row(B, i);
  return row < 0 ? 0.0 : arena[row + (int64_t)j * B->ld];
}

int cholamd_plan_arena_to_dense(const cholamd_plan *p, const double *arena, double *dense)
{
  const size_t n = p->n;
  memset(dense, 0, n * n * sizeof(double));
  for (int b = 0; b < p->nblk; b++) {
    const chol_block *B = &p->blk[b];
    for (int i0 = 0; i0 < B->rows; i0 += CHOL_NB) { /* tile by tile: a tile without storage is zero */
      const int64_t row = chol_block_row(B, i0);
      const int m = B->rows - i0 < CHOL_NB ? B->rows - i0 : CHOL_NB;
      if (row < 0) continue;
      for (int j = 0; j < B->cols; j++)
        memcpy(dense + (size_t)B->lo_x + i0 + (size_t)(B->lo_y + j) * n, arena + row + (int64_t)j * B->ld, (size_t)m * sizeof(double));
    }
  }
  return 0;
}

int cholamd_plan_write_matrix(const cholamd_plan *p, const double *arena, const char *file, int full_precision)
{
  /* write_matrix, mmat.rg:102-147: blocks in colour order, row-major inside a block, 1-based */
  FILE *fp = fopen(file, "w");
  if (!fp) { chol_set_error("cannot write %s: %s", file, strerror(errno)); return CHOLAMD_ERR_IO; }
  int64_t nnz = 0;
  for (int b = 0; b < p->nblk; b++) {
    const chol_block *B = &p->blk[b];
    for (int j = 0; j < B->cols; j++)
      for (int i = 0; i < B->rows; i++) nnz += block_value(B, arena, i, j) != 0.0;
  }
  MM_typecode tc; memcpy(tc, p->typecode, 4);
  mm_write_banner(fp, tc);
  mm_write_mtx_crd_size(fp, p->n, p->n, (int)nnz);
  for (int b = 0; b < p->nblk; b++) {
    const chol_block *B = &p->blk[b];
    for (int i = 0; i < B->rows; i++)
      for (int j = 0; j < B->cols; j++) {
        double v = block_value(B, arena, i, j);
        if (v != 0.0) fprintf(fp, full_precision ? "%d %d %.17g\n" : "%d %d %0.8g\n", B->lo_x + i + 1, B->lo_y + j + 1, v);
      }
  }
  fclose(fp);
  return 0;
}

/* ---------------------------------------------------------------------------------------- */
/* -d structured op log (blas.rg:308, 340, 405; Block lines mmat.rg:331,352)                   */
/* ---------------------------------------------------------------------------------------- */
static const cholamd_filled *find_tile(const cholamd_plan *p, int lbl, int sx, int sy, int z)
{
  const cholamd_filled *v = p->snap[lbl];
  for (int64_t i = 0; i < p->snap_n[lbl]; i++)
    if (v[i].sep_x == sx && v[i].sep_y == sy && v[i].cluster == z) return &v[i];
  return NULL;
}

/* What the reference's symbolic phase prints with -d, in its order: the Block lines of partition_matrix (mmat.rg:331,352),
 * then per tree level (bottom-up) the Cluster lines of partition_separators (every tile of every block among the separators
 * of tree levels <= that level at the interval in force, labelled with the interval label; mmat.rg:396,432,447) and the Fill
 * lines of the level's snapshot (filled tiles only; mmat.rg:1010).  verify.debug_factor reads the Block and Cluster lines. */
int cholamd_plan_write_debug_header(const cholamd_plan *p, FILE *f)
{
  for (int b = 0; b < p->nblk; b++) {
    const chol_block *B = &p->blk[b];
    fprintf(f, "Block: {'Block': (%d, %d), 'Lo': (%d, %d), 'Hi': (%d, %d)}\n", B->r, B->c, B->lo_x, B->lo_y, B->hi_x, B->hi_y);
  }
  const int L = p->levels;
  for (int lvl = L - 1; lvl >= 0; lvl--) {
    const int lbl = L - 1 - lvl, t = interval_of_level(p, lvl);
    for (int b = 0; b < p->nblk; b++) { /* blocks are ordered by (row label, column label) */
      const chol_block *B = &p->blk[b];
      if (p->level_of[B->r] > lvl || p->level_of[B->c] > lvl) continue;
      const int nr = chol_ntiles(p, B->r, t), nc = chol_ntiles(p, B->c, t);
      if (nr < 0 || nc < 0) continue;
      fprintf(f, "\t\tPartitioning (%d, %d) Cluster: %d Rows: %d Cols: %d\n", B->r, B->c, t, nr, nc);
      const int *rs = p->cl[B->r].start[t], *cs = p->cl[B->c].start[t];
      for (int i = 0; i < nr; i++) {
        for (int j = 0; j < nc; j++) {
          const int lox = B->lo_x + rs[i], hix = B->lo_x + rs[i + 1] - 1, loy = B->lo_y + cs[j], hiy = B->lo_y + cs[j + 1] - 1;
          fprintf(f, "\t\tCluster: {'Block': (%d, %d), 'color': (%d, %d, %d), 'Lo': (%d, %d), 'Hi': (%d, %d), 'size': (%d, %d), 'vol': %d, 'Interval': %d}\n",
                  B->r, B->c, B->r, B->c, i * nc + j, lox, loy, hix, hiy, hix - lox + 1, hiy - loy + 1, (hix - lox + 1) * (hiy - loy + 1), lbl);
        }
        fprintf(f, "\n");
      }
    }
    const cholamd_filled *v = p->snap[lbl];
    for (int64_t i = 0; i < p->snap_n[lbl]; i++)
      fprintf(f, "Fill: {'Level': %d, 'Interval': %d, 'Block': (%d, %d), 'Cluster': (%d, %d, %d), 'Filled': %d, 'Lo': (%d, %d), 'Hi': (%d, %d), 'Size': (%d, %d)}\n",
              lvl, lbl, v[i].sep_x, v[i].sep_y, v[i].sep_x, v[i].sep_y, v[i].cluster, v[i].filled, v[i].lo_x, v[i].lo_y, v[i].hi_x, v[i].hi_y,
              v[i].hi_x - v[i].lo_x + 1, v[i].hi_y - v[i].lo_y + 1);
  }
  return 0;
}
/* write_blocks' second file (mmat.rg:174-218): a header line naming the task, then every allocated block as
 * "Color: r c size: RxC bounds.lo: .. bounds.hi: .. vol: V" followed by its rows, values as "%0.2f, " (a blank in front of
 * non-negative ones) */
int cholamd_plan_write_blocks_txt(const cholamd_plan *p, const double *arena, const char *file, const char *header)
{
  FILE *fp = fopen(file, "w");
  if (!fp) { chol_set_error("cannot write %s: %s", file, strerror(errno)); return CHOLAMD_ERR_IO; }
  fprintf(fp, "%s\n", header);
  for (int b = 0; b < p->nblk; b++) {
    const chol_block *B = &p->blk[b];
    if (B->rows * B->cols == 0) continue;
    fprintf(fp, "Color: %d %d size: %dx%d bounds.lo: %d %d bounds.hi: %d %d vol: %d\n", B->r, B->c, B->rows, B->cols, B->lo_x, B->lo_y, B->hi_x, B->hi_y, B->rows * B->cols);
    for (int i = 0; i < B->rows; i++) {
      for (int j = 0; j < B->cols; j++) {
        const double v = block_value(B, arena, i, j);
        fprintf(fp, v < 0 ? "%0.2f, " : " %0.2f, ", v);
      }
      fprintf(fp, "\n");
    }
  }
  fclose(fp);
  return 0;
}
/* the number of tiles of separator `sep` at the interval in force at tree level `level` (col_cluster_size of the fused tasks) */
int cholamd_plan_ntiles_at_level(const cholamd_plan *p, int sep, int level) { return chol_ntiles(p, sep, interval_of_level(p, level)); }

int cholamd_plan_write_debug_log(const cholamd_plan *p, FILE *f)
{
  for (int b = 0; b < p->nblk; b++) {
    const chol_block *B = &p->blk[b];
    fprintf(f, "Block: {'Block': (%d, %d), 'Lo': (%d, %d), 'Hi': (%d, %d)}\n", B->r, B->c, B->lo_x, B->lo_y, B->hi_x, B->hi_y);
  }
  for (int64_t i = 0; i < p->nops; i++) {
    const cholamd_op *o = &p->ops[i];
    int lbl = p->levels - 1 - o->level;
    const cholamd_filled *a = find_tile(p, lbl, o->a_sx, o->a_sy, o->a_z);
    const cholamd_filled *b = o->op ? find_tile(p, lbl, o->b_sx, o->b_sy, o->b_z) : NULL;
    const cholamd_filled *c = o->op >= 2 ? find_tile(p, lbl, o->c_sx, o->c_sy, o->c_z) : NULL;
    if (!a || (o->op && !b) || (o->op >= 2 && !c)) { chol_set_error("internal: op %ld references an unfilled tile", (long)i); return CHOLAMD_ERR_INVARIANT; }
#define SZ(q) (q)->hi_x - (q)->lo_x + 1, (q)->hi_y - (q)->lo_y + 1
    if (o->op == 0)
      fprintf(f, "POTRF: {'A': (%d, %d, %d), 'A_Lo': (%d, %d), 'A_Hi': (%d, %d), 'SizeA': (%d, %d), 'Block': (%d, %d), 'Level': %d, 'Interval': %d}\n",
              a->sep_x, a->sep_y, a->cluster, a->lo_x, a->lo_y, a->hi_x, a->hi_y, SZ(a), a->sep_x, a->sep_y, o->level, lbl);
    else if (o->op == 1)
      fprintf(f, "TRSM: {'A': (%d, %d, %d), 'A_Lo': (%d, %d), 'A_Hi': (%d, %d), 'SizeA': (%d, %d), 'B': (%d, %d, %d), 'B_Lo': (%d, %d), 'B_Hi': (%d, %d), 'SizeB': (%d, %d), 'Block': (%d, %d), 'Level': %d, 'Interval': %d}\n",
              a->sep_x, a->sep_y, a->cluster, a->lo_x, a->lo_y, a->hi_x, a->hi_y, SZ(a),
              b->sep_x, b->sep_y, b->cluster, b->lo_x, b->lo_y, b->hi_x, b->hi_y, SZ(b), b->sep_x, b->sep_y, o->level, lbl);
    else
      fprintf(f, "GEMM: {'A': (%d, %d, %d), 'A_Lo': (%d, %d), 'A_Hi': (%d, %d), 'sizeA': (%d, %d), 'B': (%d, %d, %d), 'B_Lo': (%d, %d), 'B_Hi': (%d, %d), 'sizeB': (%d, %d), 'C': (%d, %d, %d), 'C_Lo': (%d, %d), 'C_Hi': (%d, %d), 'sizeC': (%d, %d), 'Block': (%d, %d), 'Level': %d, 'Interval': %d}\n",
              a->sep_x, a->sep_y, a->cluster, a->lo_x, a->lo_y, a->hi_x, a->hi_y, SZ(a),
              b->sep_x, b->sep_y, b->cluster, b->lo_x, b->lo_y, b->hi_x, b->hi_y, SZ(b),
              c->sep_x, c->sep_y, c->cluster, c->lo_x, c->lo_y, c->hi_x, c->hi_y, SZ(c), c->sep_x, c->sep_y, o->level, lbl);
#undef SZ
  }
  return 0;
}
