/* Laplacian + geometric nested-dissection + cluster generator.  Not present in the
 * reference (its `_ord_` / `_clust_` files come from an external tool); emits the reference's file
 * formats so generated problems can be fed to the reference, the oracle and
 * this library alike.  Everything is deterministic (no randomness, no seed).
 */
#define _GNU_SOURCE
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "chol_plan.h"

struct cholamd_problem {
  int nx, ny, nz, n, levels, nsep;
  int *perm;       /* dof at permuted position */
  int *sep_sizes;  /* by label 1..nsep (index label-1) */
  int64_t ncl; int *cl_idx, *cl_interval, *cl_sep;
  int64_t nz_a; int *a_row, *a_col; double *a_val;
};

typedef struct { int lo[3], hi[3]; } box_t; /* half-open */


/* dofs of a box in lexicographic (x fastest) order */
static int emit_box(const struct cholamd_problem *g, const box_t *b, int *out)
{
  int k = 0;
  for (int z = b->lo[2]; z < b->hi[2]; z++)
    for (int y = b->lo[1]; y < b->hi[1]; y++)
      for (int x = b->lo[0]; x < b->hi[0]; x++) out[k++] = x + g->nx * (y + g->ny * z);
  return k;
}

int cholamd_generate_laplacian(int nx, int ny, int nz, int levels, int tile, cholamd_problem **out)
{
  *out = NULL;
  if (nx < 1 || ny < 1 || nz < 1 || levels < 1 || levels > 20 || tile < 1) { chol_set_error("generator: bad arguments"); return CHOLAMD_ERR_ARG; }
  struct cholamd_problem *g = calloc(1, sizeof *g);
  g->nx = nx; g->ny = ny; g->nz = nz; g->n = nx * ny * nz; g->levels = levels; g->nsep = (1 << levels) - 1;
  const int ns = g->nsep, n = g->n;
  /* heap-ordered boxes: node i is split by its separator plane into children 2i, 2i+1; at the last
   * level the node's whole box is its "separator" (a leaf subdomain) */
  box_t *dom = calloc(ns + 2, sizeof(box_t)), *sepb = calloc(ns + 2, sizeof(box_t));
  dom[1] = (box_t){ { 0, 0, 0 }, { nx, ny, nz } };
  for (int i = 1; i <= ns; i++) {
    int lvl = 0;
    while ((1 << (lvl + 1)) <= i) lvl++;
    box_t b = dom[i];
    if (lvl == levels - 1) { sepb[i] = b; continue; }
    int ax = 0;
    for (int a = 1; a < 3; a++) if (b.hi[a] - b.lo[a] > b.hi[ax] - b.lo[ax]) ax = a;
    const int len = b.hi[ax] - b.lo[ax];
    box_t s = b, l = b, r = b;
    if (len >= 3) {
      const int mid = b.lo[ax] + len / 2;
      s.lo[ax] = mid; s.hi[ax] = mid + 1; l.hi[ax] = mid; r.lo[ax] = mid + 1;
    } else if (len == 2) { /* no room for a separating plane: one side becomes the separator */
      s.lo[ax] = b.lo[ax] + 1; l.hi[ax] = b.lo[ax] + 1; r.lo[ax] = r.hi[ax] = b.hi[ax];
    } else { /* a single layer: it is the separator, both children are empty */
      l.hi[ax] = l.lo[ax]; r.lo[ax] = r.hi[ax];
    }
    sepb[i] = s; dom[2 * i] = l; dom[2 * i + 1] = r;
  }
  /* labels: heap index i -> nsep - (i - 1)  (mmat.rg:834-849); positions by ascending label */
  g->perm = malloc((n > 0 ? n : 1) * sizeof(int));
  g->sep_sizes = calloc(ns + 1, sizeof(int));
  int pos = 0;
  for (int label = 1; label <= ns; label++) {
    const int i = ns - label + 1;
    const int k = emit_box(g, &sepb[i], g->perm + pos);
    g->sep_sizes[label - 1] = k;
    pos += k;
  }
  if (pos != n) { chol_set_error("generator: internal dof count mismatch"); free(dom); free(sepb); cholamd_problem_destroy(g); return CHOLAMD_ERR_INVARIANT; }
  for (int label = 1; label <= ns; label++)
    if (g->sep_sizes[label - 1] == 0) {
      chol_set_error("generator: separator %d is empty (grid too small for %d levels)", label, levels);
      free(dom); free(sepb); cholamd_problem_destroy(g);
      return CHOLAMD_ERR_ARG;
    }
  /* clusters: separator at tree level l needs intervals 0 .. need = max(0, levels-2-l) with exactly one
   * tile at interval `need`.  Interval 0 cuts the separator into ceil(size / tile) nearly equal runs
   * (1 run if need == 0); each further interval halves the count; the last one is forced to one tile. */
  int64_t cap = 0;
  for (int label = 1; label <= ns; label++) cap += g->sep_sizes[label - 1] / 1 + 4 * (levels + 2);
  g->cl_idx = malloc(cap * sizeof(int)); g->cl_interval = malloc(cap * sizeof(int)); g->cl_sep = malloc(cap * sizeof(int));
  g->ncl = 0;
  for (int label = 1; label <= ns; label++) {
    const int i = ns - label + 1;
    int lvl = 0;
    while ((1 << (lvl + 1)) <= i) lvl++;
    const int need = levels - 2 - lvl > 0 ? levels - 2 - lvl : 0;
    const int size = g->sep_sizes[label - 1];
    int m_prev = 0;
    for (int t = 0; t <= need; t++) {
      int m; /* tiles at interval t */
      if (t == need) m = 1;
      else if (t == 0) { m = (size + tile - 1) / tile; if (m < 1) m = 1; }
      else { m = (m_prev + 1) / 2; if (m < 1) m = 1; }
      if (t == 0 && m > size) m = size;
      const int units = t == 0 ? size : m_prev; /* what this interval's boundaries count */
      for (int q = 0; q <= m; q++) {
        g->cl_idx[g->ncl] = (int)(((int64_t)units * q) / m);
        g->cl_interval[g->ncl] = t; g->cl_sep[g->ncl] = label; g->ncl++;
      }
      m_prev = m;
    }
  }
  /* matrix: lower triangle, sorted by (col, row) like the fixtures */
  const int dim = (nx > 1) + (ny > 1) + (nz > 1);
  const double diag = 2.0 * (dim > 0 ? dim : 1);
  g->a_row = malloc((size_t)(4 * (int64_t)n + 4) * sizeof(int)); g->a_col = malloc((size_t)(4 * (int64_t)n + 4) * sizeof(int));
  g->a_val = malloc((size_t)(4 * (int64_t)n + 4) * sizeof(double));
  int64_t e = 0;
  for (int z = 0; z < nz; z++)
    for (int y = 0; y < ny; y++)
      for (int x = 0; x < nx; x++) {
        const int j = x + nx * (y + ny * z);
        g->a_row[e] = j; g->a_col[e] = j; g->a_val[e++] = diag;
        if (x + 1 < nx) { g->a_row[e] = j + 1; g->a_col[e] = j; g->a_val[e++] = -1.0; }
        if (y + 1 < ny) { g->a_row[e] = j + nx; g->a_col[e] = j; g->a_val[e++] = -1.0; }
        if (z + 1 < nz) { g->a_row[e] = j + nx * ny; g->a_col[e] = j; g->a_val[e++] = -1.0; }
      }
  g->nz_a = e;
  free(dom); free(sepb);
  *out = g;
  return 0;
}

void cholamd_problem_destroy(cholamd_problem *g)
{
  if (!g) return;
  free(g->perm); free(g->sep_sizes); free(g->cl_idx); free(g->cl_interval); free(g->cl_sep); free(g->a_row); free(g->a_col); free(g->a_val);
  free(g);
}
int cholamd_problem_n(const cholamd_problem *g) { return g->n; }
int cholamd_problem_nz(const cholamd_problem *g) { return (int)g->nz_a; }
void cholamd_problem_rhs(const cholamd_problem *g, double *b)
{
  for (int i = 0; i < g->n; i++) b[i] = 1.0 + (double)((7919LL * i) % 10);
}

int cholamd_plan_create_from_problem(const cholamd_problem *g, cholamd_plan **out)
{
  return cholamd_plan_create_from_arrays(g->n, g->levels, g->perm, g->sep_sizes, g->cl_idx, g->cl_interval, g->cl_sep, g->ncl,
                                         g->nz_a, g->a_row, g->a_col, g->a_val, "%%MatrixMarket matrix coordinate real hermitian", out);
}

int cholamd_problem_write(const cholamd_problem *g, const char *prefix)
{
  char path[1100];
  FILE *f;
  /* matrix, SURVEY A.1: no comment lines, lower triangle, 1-based */
  snprintf(path, sizeof path, "%s.mtx", prefix);
  if (!(f = fopen(path, "w"))) { chol_set_error("cannot write %s: %s", path, strerror(errno)); return CHOLAMD_ERR_IO; }
  fprintf(f, "%%%%MatrixMarket matrix coordinate real hermitian\n%d %d %ld\n", g->n, g->n, (long)g->nz_a);
  for (int64_t e = 0; e < g->nz_a; e++) fprintf(f, "%d %d %.1f\n", g->a_row[e] + 1, g->a_col[e] + 1, g->a_val[e]);
  fclose(f);
  /* separators, SURVEY A.2 */
  snprintf(path, sizeof path, "%s_ord_%d.txt", prefix, g->levels);
  if (!(f = fopen(path, "w"))) { chol_set_error("cannot write %s: %s", path, strerror(errno)); return CHOLAMD_ERR_IO; }
  fprintf(f, "%d %d\n", g->levels, g->nsep);
  int pos = 0;
  for (int label = 1; label <= g->nsep; label++) {
    fprintf(f, "%d;", label - 1);
    for (int k = 0; k < g->sep_sizes[label - 1]; k++) fprintf(f, "%d,", g->perm[pos++]);
    fprintf(f, "\n");
  }
  fclose(f);
  /* clusters, SURVEY A.3 */
  snprintf(path, sizeof path, "%s_clust_%d.txt", prefix, g->levels);
  if (!(f = fopen(path, "w"))) { chol_set_error("cannot write %s: %s", path, strerror(errno)); return CHOLAMD_ERR_IO; }
  fprintf(f, "%d %d\n", g->levels, g->nsep);
  int64_t q = 0;
  for (int label = 1; label <= g->nsep; label++) {
    fprintf(f, "%d;", label - 1);
    int t = 0;
    while (q < g->ncl && g->cl_sep[q] == label) {
      if (g->cl_interval[q] != t) { fprintf(f, ";"); t = g->cl_interval[q]; }
      fprintf(f, "%d,", g->cl_idx[q]);
      q++;
    }
    fprintf(f, ";\n");
  }
  fclose(f);
  /* rhs, SURVEY A.4: three header lines, then n values */
  snprintf(path, sizeof path, "%s_B.mtx", prefix);
  if (!(f = fopen(path, "w"))) { chol_set_error("cannot write %s: %s", path, strerror(errno)); return CHOLAMD_ERR_IO; }
  fprintf(f, "%%%%MatrixMarket matrix array integer general\n%%\n%d 1\n", g->n);
  for (int i = 0; i < g->n; i++) fprintf(f, "%d\n", 1 + (int)((7919LL * i) % 10));
  fclose(f);
  return 0;
}
