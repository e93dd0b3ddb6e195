/*
 * oracle/chol_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
 *
 * A plain-C, single-threaded, CPU restatement of the reference's block-sparse supernodal
 * Cholesky (syamajala/cholesky: mmat.rg + blas.rg + mnd.c), written from the survey of the
 * reference's behaviour.  It follows the reference LITERALLY (dense index spaces, linear
 * searches, tile ids z = row*ncols+col, "filled == 0 means filled") so that it is easy to audit
 * against the .rg sources, and it is deliberately independent of the product's host code in
 * cholesky_amd/csrc (different data structures, different algorithms).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product path never links, imports or calls anything in oracle/.
 *
 * Parity pinning: this oracle is checked (tests/test_oracle.py) against golden vectors produced
 * by the reference's own verify.py (permute_matrix + scipy cholesky / solve) for all four of the
 * reference's fixtures -- see tests/golden/make_golden.py.
 *
 * Reference sections restated (file:line in /root/reference):
 *   parsers .................. mnd.c:22-69 (separators), :71-150 (clusters), :152-199 (matrix),
 *                              :201-229 (rhs vector); banner/size mmio.c:96-179, :189-217
 *   separator tree ........... mmat.rg:834-849
 *   block rectangles ......... mmat.rg:299-362, allocated blocks :740-767
 *   tile rectangles .......... mmat.rg:364-451 (partition_separator), :453-499
 *   initial scatter + fill ... mmat.rg:529-633 (fill_block)
 *   symbolic fill per level .. mmat.rg:896-1028 (compute_filled_clusters), :635-695 (merge)
 *   numeric level schedule ... mmat.rg:1227-1355
 *   fused leaf tasks ......... blas.rg:292-315 (potrf), :317-351 (trsm), :353-436 (syrk),
 *                              :438-504 (gemm)
 *   BLAS call semantics ...... blas.rg:71, :99, :139, :187, :226, :263
 *   solve .................... mmat.rg:1364-1495
 *   factor writer ............ mmat.rg:102-147
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------ */
/* BLAS back ends.  Default: the oracle's own straightforward kernels.  Optionally (for the CPU */
/* baseline timing only) an OpenBLAS found on the box is bound with dlopen, because OpenBLAS is */
/* what the reference links (blas.rg:18-22, mmat.rg:1057).                                      */
/* ------------------------------------------------------------------------------------------ */
enum { ColMajor = 102, NoTrans = 111, Trans = 112, Upper = 121, Lower = 122, NonUnit = 131, Left = 141, Right = 142 };

typedef int (*potrf_fn)(int, char, int, double *, int);
typedef void (*trsm_fn)(int, int, int, int, int, int, int, double, const double *, int, double *, int);
typedef void (*syrk_fn)(int, int, int, int, int, double, const double *, int, double, double *, int);
typedef void (*gemm_fn)(int, int, int, int, int, int, double, const double *, int, const double *, int, double, double *, int);
typedef void (*trsv_fn)(int, int, int, int, int, const double *, int, double *, int);
typedef void (*gemv_fn)(int, int, int, int, double, const double *, int, const double *, int, double, double *, int);

/* LAPACKE_dpotrf(ColMajor,'L',n,a,lda): unblocked lower Cholesky; returns info like LAPACK. */
static int own_potrf(int layout, char uplo, int n, double *a, int lda)
{
  (void)layout; (void)uplo;
  for (int j = 0; j < n; j++) {
    double d = a[j + (size_t)j * lda];
    for (int k = 0; k < j; k++) d -= a[j + (size_t)k * lda] * a[j + (size_t)k * lda];
    if (!(d > 0.0)) return j + 1;
    d = sqrt(d);
    a[j + (size_t)j * lda] = d;
    for (int i = j + 1; i < n; i++) {
      double s = a[i + (size_t)j * lda];
      for (int k = 0; k < j; k++) s -= a[i + (size_t)k * lda] * a[j + (size_t)k * lda];
      a[i + (size_t)j * lda] = s / d;
    }
  }
  return 0;
}

/* cblas_dtrsm(ColMajor, Right, Lower, Trans, NonUnit, m, n, 1.0, A, lda, B, ldb): B <- B * A^-T */
static void own_trsm(int layout, int side, int uplo, int trans, int diag, int m, int n, double alpha,
                     const double *A, int lda, double *B, int ldb)
{
  (void)layout; (void)side; (void)uplo; (void)trans; (void)diag; (void)alpha;
  for (int j = 0; j < n; j++) {
    for (int k = 0; k < j; k++) {
      double l = A[j + (size_t)k * lda];
      if (l != 0.0)
        for (int i = 0; i < m; i++) B[i + (size_t)j * ldb] -= B[i + (size_t)k * ldb] * l;
    }
    double d = A[j + (size_t)j * lda];
    for (int i = 0; i < m; i++) B[i + (size_t)j * ldb] /= d;
  }
}

/* cblas_dsyrk(ColMajor, Lower, NoTrans, n, k, -1, A, lda, 1, C, ldc): C <- C - A A^T (lower) */
static void own_syrk(int layout, int uplo, int trans, int n, int k, double alpha, const double *A, int lda,
                     double beta, double *C, int ldc)
{
  (void)layout; (void)uplo; (void)trans; (void)alpha; (void)beta;
  for (int j = 0; j < n; j++)
    for (int p = 0; p < k; p++) {
      double b = A[j + (size_t)p * lda];
      if (b != 0.0)
        for (int i = j; i < n; i++) C[i + (size_t)j * ldc] -= A[i + (size_t)p * lda] * b;
    }
}

/* cblas_dgemm(ColMajor, NoTrans, Trans, m, n, k, -1, A, lda, B, ldb, 1, C, ldc): C <- C - A B^T */
static void own_gemm(int layout, int ta, int tb, int m, int n, int k, double alpha, const double *A, int lda,
                     const double *B, int ldb, double beta, double *C, int ldc)
{
  (void)layout; (void)ta; (void)tb; (void)alpha; (void)beta;
  for (int j = 0; j < n; j++)
    for (int p = 0; p < k; p++) {
      double b = B[j + (size_t)p * ldb];
      if (b != 0.0)
        for (int i = 0; i < m; i++) C[i + (size_t)j * ldc] -= A[i + (size_t)p * lda] * b;
    }
}

/* cblas_dtrsv(ColMajor, Lower, NoTrans|Trans, NonUnit, n, A, lda, x, 1) */
static void own_trsv(int layout, int uplo, int trans, int diag, int n, const double *A, int lda, double *x, int incx)
{
  (void)layout; (void)uplo; (void)diag;
  if (trans == NoTrans) {
    for (int j = 0; j < n; j++) {
      double v = x[(size_t)j * incx] / A[j + (size_t)j * lda];
      x[(size_t)j * incx] = v;
      for (int i = j + 1; i < n; i++) x[(size_t)i * incx] -= v * A[i + (size_t)j * lda];
    }
  } else {
    for (int j = n - 1; j >= 0; j--) {
      double s = x[(size_t)j * incx];
      for (int i = j + 1; i < n; i++) s -= A[i + (size_t)j * lda] * x[(size_t)i * incx];
      x[(size_t)j * incx] = s / A[j + (size_t)j * lda];
    }
  }
}

/* cblas_dgemv(ColMajor, NoTrans|Trans, m, n, -1, A, lda, x, 1, 1, y, 1) */
static void own_gemv(int layout, int trans, int m, int n, double alpha, const double *A, int lda, const double *x,
                     int incx, double beta, double *y, int incy)
{
  (void)layout; (void)alpha; (void)beta;
  if (trans == NoTrans) {
    for (int j = 0; j < n; j++) {
      double v = x[(size_t)j * incx];
      for (int i = 0; i < m; i++) y[(size_t)i * incy] -= A[i + (size_t)j * lda] * v;
    }
  } else {
    for (int j = 0; j < n; j++) {
      double s = 0.0;
      for (int i = 0; i < m; i++) s += A[i + (size_t)j * lda] * x[(size_t)i * incx];
      y[(size_t)j * incy] -= s;
    }
  }
}

static potrf_fn B_potrf = own_potrf;
static trsm_fn B_trsm = own_trsm;
static syrk_fn B_syrk = own_syrk;
static gemm_fn B_gemm = own_gemm;
static trsv_fn B_trsv = own_trsv;
static gemv_fn B_gemv = own_gemv;
static char B_name[256] = "oracle-own-C-kernels";

static void *try_sym(void *h, const char *a, const char *b)
{
  void *p = dlsym(h, a);
  if (!p) p = dlsym(h, b);
  return p;
}

/* Bind an OpenBLAS (LP64) shared object; returns 0 on success.  openblas_set_num_threads(1) is
 * applied as the reference does (mmat.rg:1057). */
int orc_use_openblas(const char *path)
{
  void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) return -1;
  void *p = try_sym(h, "scipy_LAPACKE_dpotrf", "LAPACKE_dpotrf");
  void *t = try_sym(h, "scipy_cblas_dtrsm", "cblas_dtrsm");
  void *s = try_sym(h, "scipy_cblas_dsyrk", "cblas_dsyrk");
  void *g = try_sym(h, "scipy_cblas_dgemm", "cblas_dgemm");
  void *v = try_sym(h, "scipy_cblas_dtrsv", "cblas_dtrsv");
  void *m = try_sym(h, "scipy_cblas_dgemv", "cblas_dgemv");
  void (*setn)(int) = (void (*)(int))try_sym(h, "scipy_openblas_set_num_threads", "openblas_set_num_threads");
  if (!p || !t || !s || !g || !v || !m) return -2;
  B_potrf = (potrf_fn)p; B_trsm = (trsm_fn)t; B_syrk = (syrk_fn)s; B_gemm = (gemm_fn)g;
  B_trsv = (trsv_fn)v; B_gemv = (gemv_fn)m;
  if (setn) setn(1);
  snprintf(B_name, sizeof B_name, "OpenBLAS(dlopen:%s),1-thread", path);
  return 0;
}
void orc_use_own_kernels(void)
{
  B_potrf = own_potrf; B_trsm = own_trsm; B_syrk = own_syrk; B_gemm = own_gemm; B_trsv = own_trsv; B_gemv = own_gemv;
  snprintf(B_name, sizeof B_name, "oracle-own-C-kernels");
}
const char *orc_backend_name(void) { return B_name; }

/* ------------------------------------------------------------------------------------------ */
/* Data model (literal)                                                                         */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int lo_x, lo_y, hi_x, hi_y; } rect2;  /* inclusive, x = row, y = col, permuted coords */

typedef struct {         /* fspace Filled, blas.rg:55-61 */
  int filled;            /* 0 == filled, 1 == empty (sic) */
  int sep_x, sep_y;      /* block colour (row separator, col separator), labels 1..nsep */
  int interval;          /* interval label of the snapshot */
  int cluster;           /* tile id z = row*ncols + col */
  rect2 bounds;
} Filled;

typedef struct {
  int allocated;         /* find_index_space_2d */
  rect2 bounds;          /* partition_matrix */
  int rows, cols;
  double *data;          /* own instance, col-major, ld = rows (effect of the mapper, cholesky.cc:65-73) */
  int ntile_ids;         /* #allocated tile ids = ntiles0(row)*ntiles0(col) (find_index_space_3d) */
  int *filled;           /* working flags per tile id (0 filled / 1 empty) */
  rect2 *tbounds;        /* cluster_bounds per tile id (current partition_separator result) */
} Block;

typedef struct {
  int n_int;             /* #intervals */
  int *len;              /* boundary-list length per interval */
  int **b;               /* boundary lists */
} Clusters;

typedef struct { Filled *v; int n; } FilledList;

typedef struct {
  int op; /* 0 potrf 1 trsm 2 syrk 3 gemm */
  int level, m, n, k;
  int a_sx, a_sy, a_z, b_sx, b_sy, b_z, c_sx, c_sy, c_z;
} OpRec;

typedef struct {
  int M, N, NZ, levels, nsep, max_int_size;
  char banner[128];
  int *sep_of_pos;       /* SepIndex.sep per permuted position (label) */
  int *dof_of_pos;       /* SepIndex.idx: original dof at permuted position == perm */
  int *sep_size, *sep_off;   /* per label 1..nsep */
  int *tree_node;        /* heap index 1..nsep -> label */
  Clusters *cl;          /* per label */
  /* matrix entries (original coords, lower triangle as in the file) in a dense lookup */
  double *Aorig;         /* N*N dense, row-major [i*N+j] with i>=j as stored in file */
  Block *blk;            /* (nsep+1)*(nsep+1) */
  /* snapshots: filled_clusters_region2 grouped as [interval_lbl][block] */
  FilledList *snap;      /* levels * (nsep+1)*(nsep+1) */
  /* op log of the last factorisation */
  OpRec *ops; int nops, cap_ops;
  int log_ops;
  int quiet;   /* level-parallel runs: log_op keeps no statistics (shared counters) */
  double flops[4]; long calls[4];
  double level_flops[16][4]; long level_calls[16][4];
  int info;              /* first non-zero potrf info */
} Orc;

#define BLK(o, r, c) ((o)->blk[(size_t)(r) * ((o)->nsep + 1) + (c)])
#define SNAP(o, lbl, r, c) ((o)->snap[((size_t)(lbl) * ((o)->nsep + 1) + (r)) * ((o)->nsep + 1) + (c)])

static int ntiles(const Orc *o, int sep, int interval)
{
  if (interval >= o->cl[sep].n_int) return -1; /* "volume == 0" */
  return o->cl[sep].len[interval] - 1;
}
/* resolve boundary `idx` of interval t down to a dof offset (mmat.rg:400-422) */
static int resolve(const Orc *o, int sep, int t, int idx)
{
  int v = o->cl[sep].b[t][idx];
  for (int i = t - 1; i >= 0; i--) v = o->cl[sep].b[i][v];
  return v;
}

/* ---------------------------------------------------------------------------------------- */
/* Parsers                                                                                    */
/* ---------------------------------------------------------------------------------------- */
static int read_banner(Orc *o, const char *file)
{
  /* mm_read_banner + mm_read_mtx_crd_size behaviour (mmio.c:96-179, 189-217): banner line with
   * 5 tokens, then skip '%' comment lines, then "M N NZ".  Validity (mm_is_valid) is NOT checked,
   * so "real hermitian" is accepted. */
  FILE *fp = fopen(file, "r");
  if (!fp) return -1;
  char line[1100];
  if (!fgets(line, sizeof line, fp)) { fclose(fp); return -2; }
  char t0[64], t1[64], t2[64], t3[64], t4[64];
  if (sscanf(line, "%63s %63s %63s %63s %63s", t0, t1, t2, t3, t4) != 5) { fclose(fp); return -3; }
  if (strncmp(t0, "%%MatrixMarket", 14) != 0) { fclose(fp); return -4; }
  size_t L = strlen(line);
  while (L && (line[L - 1] == '\n' || line[L - 1] == '\r')) line[--L] = 0;
  snprintf(o->banner, sizeof o->banner, "%s", line);
  do {
    if (!fgets(line, sizeof line, fp)) { fclose(fp); return -5; }
  } while (line[0] == '%');
  if (sscanf(line, "%d %d %d", &o->M, &o->N, &o->NZ) != 3) { fclose(fp); return -6; }
  fclose(fp);
  return 0;
}

static int read_separators(Orc *o, const char *file)
{
  /* mnd.c:22-69.  Line 0: "levels num_separators" (single digit levels: atoi(&line[0]),
   * atoi(&line[2])).  Then "k;d0,d1,...,dn," per separator; label = k+1; dofs fill consecutive
   * permuted positions in file order. */
  FILE *fp = fopen(file, "r");
  if (!fp) return -1;
  char *line = NULL; size_t cap = 0; ssize_t rd; int i = 0, pos = 0;
  o->sep_of_pos = calloc(o->M, sizeof(int));
  o->dof_of_pos = calloc(o->M, sizeof(int));
  while ((rd = getline(&line, &cap, fp)) != -1) {
    if (i == 0) {
      o->levels = atoi(&line[0]);
      o->nsep = atoi(&line[2]);
      i++;
      continue;
    }
    char *save = NULL;
    char *tok = strtok_r(line, ";", &save);
    if (!tok) break;
    int separator = atoi(tok) + 1;
    tok = strtok_r(NULL, ",", &save);
    while (tok != NULL) {
      if (isspace((unsigned char)*tok)) break;
      if (pos >= o->M) { free(line); fclose(fp); return -2; }
      o->sep_of_pos[pos] = separator;
      o->dof_of_pos[pos] = atoi(tok);
      pos++;
      tok = strtok_r(NULL, ",", &save);
    }
    i++;
  }
  free(line);
  fclose(fp);
  if (pos != o->M) return -3;
  return 0;
}

static int read_clusters(Orc *o, const char *file)
{
  /* mnd.c:71-150.  Tokens split on ",; "; a token equal to "0" (after the first) starts the next
   * interval; the trailing "\n" token terminates the line.  Return value max_int_size restated
   * with its quirk (last interval counts the newline token). */
  FILE *fp = fopen(file, "r");
  if (!fp) return -1;
  char *line = NULL; size_t cap = 0; ssize_t rd; int i = 0;
  int max_int_size = -1;
  o->cl = calloc(o->nsep + 1, sizeof(Clusters));
  while ((rd = getline(&line, &cap, fp)) != -1) {
    if (i == 0) { i++; continue; }
    char *save = NULL;
    char *tok = strtok_r(line, "; ", &save);
    if (!tok || !isdigit((unsigned char)*tok)) continue;
    int separator = atoi(tok) + 1;
    if (separator < 1 || separator > o->nsep) { free(line); fclose(fp); return -2; }
    Clusters *c = &o->cl[separator];
    c->n_int = 0; c->len = calloc(o->levels + 2, sizeof(int)); c->b = calloc(o->levels + 2, sizeof(int *));
    int interval = 0, dofs = 0;
    int capv = 16; int *vals = malloc(capv * sizeof(int)); int nv = 0;
    tok = strtok_r(NULL, ",; ", &save);
    while (tok != NULL) {
      int row = atoi(tok);
      dofs++;
      tok = strtok_r(NULL, ",; ", &save);
      if (tok == NULL) {
        if (dofs > max_int_size) max_int_size = dofs;
      } else {
        if (nv == capv) { capv *= 2; vals = realloc(vals, capv * sizeof(int)); }
        vals[nv++] = row;
        if (strcmp("0", tok) == 0) {
          if (dofs > max_int_size) max_int_size = dofs;
          c->len[interval] = nv; c->b[interval] = vals;
          interval++; dofs = 0;
          capv = 16; vals = malloc(capv * sizeof(int)); nv = 0;
        }
      }
    }
    if (nv > 0) { c->len[interval] = nv; c->b[interval] = vals; interval++; } else free(vals);
    c->n_int = interval;
    i++;
  }
  free(line);
  fclose(fp);
  o->max_int_size = max_int_size;
  return 0;
}

static int read_matrix(Orc *o, const char *file)
{
  /* mnd.c:152-199: skip exactly two lines, then NZ lines "%lu %lu %lg", 1-based.  The reference
   * stores entries in an open-addressing hash keyed by i*cols+j (empty <=> val == 0, so explicit
   * zeros are not representable); the oracle keeps the same observable behaviour with a dense
   * N*N lookup (entry (i,j) as written in the file; lookups are always (max,min), mmat.rg:581-585). */
  FILE *fp = fopen(file, "r");
  if (!fp) return -1;
  char buff[1100];
  if (!fgets(buff, sizeof buff, fp) || !fgets(buff, sizeof buff, fp)) { fclose(fp); return -2; }
  o->Aorig = calloc((size_t)o->M * o->N, sizeof(double));
  for (int n = 0; n < o->NZ; n++) {
    unsigned long i = 0, j = 0; double val = 0.0;
    if (fscanf(fp, "%lu %lu %lg\n", &i, &j, &val) != 3) { fclose(fp); return -3; }
    i -= 1; j -= 1;
    if (i >= (unsigned long)o->M || j >= (unsigned long)o->N) { fclose(fp); return -4; }
    o->Aorig[i * (size_t)o->N + j] = val;
  }
  fclose(fp);
  return 0;
}

/* read_vector, mnd.c:201-229: three header lines skipped blindly, then n values */
int orc_read_vector(const char *file, int n, double *out)
{
  FILE *fp = fopen(file, "r");
  if (!fp) return -1;
  char buff[1100];
  for (int i = 0; i < 3; i++) if (!fgets(buff, sizeof buff, fp)) { fclose(fp); return -2; }
  for (int i = 0; i < n; i++) {
    double v = 0.0;
    if (fscanf(fp, "%lg\n", &v) != 1) { fclose(fp); return -3; }
    out[i] = v;
  }
  fclose(fp);
  return 0;
}

/* ---------------------------------------------------------------------------------------- */
/* Symbolic phase                                                                             */
/* ---------------------------------------------------------------------------------------- */
static int lvl_lo(int lvl) { return 1 << lvl; }            /* first heap index of a level */
static int lvl_hi(int lvl) { return (1 << (lvl + 1)) - 1; } /* last heap index of a level  */

static void build_separator_tree(Orc *o)
{
  /* mmat.rg:834-849 */
  o->tree_node = calloc(o->nsep + 2, sizeof(int));
  int num = o->nsep, i = 1;
  for (int level = 0; level < o->levels; level++)
    for (int e = 0; e < (1 << level); e++) { o->tree_node[i] = num; num--; i++; }
}

static void find_index_space_2d(Orc *o)
{
  /* mmat.rg:740-767 */
  for (int lvl = 0; lvl < o->levels; lvl++)
    for (int si = lvl_lo(lvl); si <= lvl_hi(lvl); si++) {
      int row_sep = o->tree_node[si];
      BLK(o, row_sep, row_sep).allocated = 1;
      for (int clvl = lvl + 1; clvl < o->levels; clvl++)
        for (int ci = si << (clvl - lvl); ci < ((si + 1) << (clvl - lvl)); ci++)
          BLK(o, row_sep, o->tree_node[ci]).allocated = 1;
    }
}

static void partition_matrix(Orc *o)
{
  /* mmat.rg:299-362 */
  int prev = o->M - 1;
  for (int lvl = 0; lvl < o->levels; lvl++)
    for (int si = lvl_lo(lvl); si <= lvl_hi(lvl); si++) {
      int sep = o->tree_node[si];
      int size = o->sep_size[sep];
      rect2 b = { prev - (size - 1), prev - (size - 1), prev, prev };
      BLK(o, sep, sep).bounds = b;
      prev -= size;
      int pi = si;
      for (int pl = lvl - 1; pl >= 0; pl--) {
        pi = pi / 2;
        int par = o->tree_node[pi];
        rect2 pb = BLK(o, par, par).bounds;
        rect2 cb = { pb.lo_x, b.lo_y, pb.hi_x, b.hi_y };
        BLK(o, par, sep).bounds = cb;
      }
    }
}

/* partition_separator, mmat.rg:364-451: tile rectangles of one block at `interval` */
static void partition_separator(Orc *o, int row_sep, int col_sep, int interval)
{
  Block *B = &BLK(o, row_sep, col_sep);
  for (int z = 0; z < B->ntile_ids; z++) { rect2 e = { 0, 0, -1, -1 }; B->tbounds[z] = e; }
  int rcs = ntiles(o, row_sep, interval), ccs = ntiles(o, col_sep, interval);
  int px = B->bounds.lo_x, py = B->bounds.lo_y;
  for (int row = 0; row < rcs; row++) {
    int top = resolve(o, row_sep, interval, row), bottom = resolve(o, row_sep, interval, row + 1);
    for (int col = 0; col < ccs; col++) {
      int left = resolve(o, col_sep, interval, col), right = resolve(o, col_sep, interval, col + 1);
      int z = row * ccs + col;
      rect2 r = { px, py, px + (bottom - top - 1), py + (right - left - 1) };
      if (z < B->ntile_ids) B->tbounds[z] = r;
      py += right - left;
    }
    px += bottom - top;
    py = B->bounds.lo_y;
  }
}

/* partition_separators, mmat.rg:453-499 */
static void partition_separators(Orc *o, int depth, int interval)
{
  for (int lvl = 0; lvl <= depth; lvl++)
    for (int si = lvl_lo(lvl); si <= lvl_hi(lvl); si++) {
      int row = o->tree_node[si];
      partition_separator(o, row, row, interval);
      for (int clvl = lvl + 1; clvl <= depth; clvl++)
        for (int ci = si << (clvl - lvl); ci < ((si + 1) << (clvl - lvl)); ci++)
          partition_separator(o, row, o->tree_node[ci], interval);
    }
}

/* fill_block, mmat.rg:529-633: zero the block, scatter A, mark filled interval-0 tiles */
static int fill_block(Orc *o, int row_sep, int col_sep, int mark)
{
  Block *B = &BLK(o, row_sep, col_sep);
  memset(B->data, 0, (size_t)B->rows * B->cols * sizeof(double));
  int rcs = ntiles(o, row_sep, 0), ccs = ntiles(o, col_sep, 0);
  int nz = 0;
  for (int col = 0; col < ccs; col++) {
    int left = o->cl[col_sep].b[0][col], right = o->cl[col_sep].b[0][col + 1];
    for (int row = 0; row < rcs; row++) {
      int top = o->cl[row_sep].b[0][row], bottom = o->cl[row_sep].b[0][row + 1];
      int z = row * ccs + col, nnz = 0;
      for (int i = top; i < bottom; i++)
        for (int j = left; j < right; j++) {
          int idxi = o->dof_of_pos[B->bounds.lo_x + i];
          int idxj = o->dof_of_pos[B->bounds.lo_y + j];
          if (idxj > idxi) { int t = idxi; idxi = idxj; idxj = t; }
          double val = o->Aorig[(size_t)idxi * o->N + idxj];
          int gx = B->bounds.lo_x + i, gy = B->bounds.lo_y + j; /* global permuted coords */
          if (row_sep == col_sep && gy <= gx) {
            if (val != 0.0) { B->data[i + (size_t)j * B->rows] = val; nnz++; }
          } else if (row_sep != col_sep) {
            if (val != 0.0) { B->data[i + (size_t)j * B->rows] = val; nnz++; }
          }
        }
      if (mark && nnz > 0) B->filled[z] = 0;
      nz += nnz;
    }
  }
  return nz;
}

/* merge_filled_clusters, mmat.rg:635-695 */
static void merge_filled_clusters(Orc *o, int interval)
{
  int ns = o->nsep;
  for (int r = 1; r <= ns; r++)
    for (int c = 1; c <= ns; c++) {
      Block *B = &BLK(o, r, c);
      if (!B->allocated) continue;
      int *old = malloc(B->ntile_ids * sizeof(int));
      memcpy(old, B->filled, B->ntile_ids * sizeof(int));
      for (int z = 0; z < B->ntile_ids; z++) B->filled[z] = 1;
      int rcs = ntiles(o, r, interval), ccs = ntiles(o, c, interval);
      if (rcs >= 0 && ccs >= 0) {
        int prev_cols = ntiles(o, c, interval - 1);
        for (int row = 0; row < rcs; row++) {
          int top = o->cl[r].b[interval][row], bottom = o->cl[r].b[interval][row + 1];
          for (int col = 0; col < ccs; col++) {
            int left = o->cl[c].b[interval][col], right = o->cl[c].b[interval][col + 1];
            int nz = row * ccs + col;
            for (int i = top; i < bottom; i++)
              for (int j = left; j < right; j++)
                if (old[i * prev_cols + j] == 0) B->filled[nz] = 0;
          }
        }
      }
      free(old);
    }
}

/* compute_filled_clusters, mmat.rg:896-1028 */
static void compute_filled_clusters(Orc *o)
{
  int interval = 0, interval_lbl = 0, ns = o->nsep;
  for (int lvl = o->levels - 1; lvl >= 0; lvl--) {
    partition_separators(o, lvl, interval);
    for (int si = lvl_lo(lvl); si <= lvl_hi(lvl); si++) {
      int sep = o->tree_node[si];
      int pi = si;
      for (int pl = lvl - 1; pl >= 0; pl--) {
        pi = pi / 2;
        int par = o->tree_node[pi];
        int gi = pi;
        for (int gl = pl; gl >= 0; gl--) {
          int gp = o->tree_node[gi];
          int ccs = ntiles(o, par, interval);
          int A_clusters = ntiles(o, gp, interval) * ntiles(o, sep, interval);
          int B_clusters = ntiles(o, par, interval) * ntiles(o, sep, interval);
          Block *A = &BLK(o, gp, sep), *Bb = &BLK(o, par, sep), *C = &BLK(o, gp, par);
          for (int i = 0; i < A_clusters; i++) {
            if (A->filled[i] != 0) continue;
            for (int j = 0; j < B_clusters; j++) {
              if (Bb->filled[j] != 0) continue;
              if (gp == par && !(j <= i)) continue;
              int cz = i * ccs + j;
              if (cz < C->ntile_ids && C->filled[cz] == 1) C->filled[cz] = 0;
            }
          }
          gi = gi / 2;
        }
      }
    }
    /* snapshot (mmat.rg:1000-1016) */
    for (int r = 1; r <= ns; r++)
      for (int c = 1; c <= ns; c++) {
        Block *B = &BLK(o, r, c);
        if (!B->allocated) continue;
        FilledList *fl = &SNAP(o, interval_lbl, r, c);
        fl->v = malloc((B->ntile_ids ? B->ntile_ids : 1) * sizeof(Filled));
        fl->n = 0;
        for (int z = 0; z < B->ntile_ids; z++)
          if (B->filled[z] == 0) {
            Filled f = { 0, r, c, interval_lbl, z, B->tbounds[z] };
            fl->v[fl->n++] = f;
          }
      }
    interval_lbl++;
    if (lvl <= o->levels - 2) {
      interval++;
      if (interval < o->levels) merge_filled_clusters(o, interval);
    }
  }
}

/* ---------------------------------------------------------------------------------------- */
/* Construction                                                                               */
/* ---------------------------------------------------------------------------------------- */
void orc_free(Orc *o);

Orc *orc_load(const char *mtx, const char *ord, const char *clust, int *err)
{
  Orc *o = calloc(1, sizeof(Orc));
  int e;
#define FAIL(code) do { if (err) *err = (code); orc_free(o); return NULL; } while (0)
  if ((e = read_banner(o, mtx)) != 0) FAIL(100 + (-e));
  if ((e = read_separators(o, ord)) != 0) FAIL(200 + (-e));
  if (o->nsep != (1 << o->levels) - 1) FAIL(250);
  int ns = o->nsep;
  o->sep_size = calloc(ns + 2, sizeof(int));
  o->sep_off = calloc(ns + 2, sizeof(int));
  for (int p = 0; p < o->M; p++) o->sep_size[o->sep_of_pos[p]]++;
  { int acc = 0; for (int s = 1; s <= ns; s++) { o->sep_off[s] = acc; acc += o->sep_size[s]; } }
  /* positions must be grouped by ascending label for the reference's layout to make sense */
  for (int p = 1; p < o->M; p++) if (o->sep_of_pos[p] < o->sep_of_pos[p - 1]) FAIL(260);
  build_separator_tree(o);
  if ((e = read_clusters(o, clust)) != 0) FAIL(300 + (-e));
  for (int s = 1; s <= ns; s++) {
    if (o->cl[s].n_int < 1) FAIL(350);
    if (o->cl[s].b[0][o->cl[s].len[0] - 1] != o->sep_size[s]) FAIL(351);
  }
  /* invariants the reference relies on but never checks: a separator at tree level l
   * has intervals 0..max(0,levels-2-l) and exactly ONE tile at the last of them */
  for (int i = 1; i <= ns; i++) {
    int lvl = 0; while ((1 << (lvl + 1)) <= i) lvl++;
    int need = o->levels - 2 - lvl; if (need < 0) need = 0;
    int s = o->tree_node[i];
    if (o->cl[s].n_int < need + 1) FAIL(352);
    if (o->cl[s].len[need] != 2) FAIL(353);
  }
  if ((e = read_matrix(o, mtx)) != 0) FAIL(400 + (-e));
  o->blk = calloc((size_t)(ns + 1) * (ns + 1), sizeof(Block));
  o->snap = calloc((size_t)o->levels * (ns + 1) * (ns + 1), sizeof(FilledList));
  find_index_space_2d(o);
  partition_matrix(o);
  for (int r = 1; r <= ns; r++)
    for (int c = 1; c <= ns; c++) {
      Block *B = &BLK(o, r, c);
      if (!B->allocated) continue;
      B->rows = B->bounds.hi_x - B->bounds.lo_x + 1;
      B->cols = B->bounds.hi_y - B->bounds.lo_y + 1;
      B->data = calloc((size_t)B->rows * B->cols + 1, sizeof(double));
      B->ntile_ids = ntiles(o, r, 0) * ntiles(o, c, 0);          /* find_index_space_3d, mmat.rg:697-738 */
      B->filled = malloc((B->ntile_ids + 1) * sizeof(int));
      B->tbounds = malloc((B->ntile_ids + 1) * sizeof(rect2));
      for (int z = 0; z < B->ntile_ids; z++) B->filled[z] = 1;
    }
  for (int r = 1; r <= ns; r++)
    for (int c = 1; c <= ns; c++)
      if (BLK(o, r, c).allocated) fill_block(o, r, c, 1);          /* mmat.rg:1175-1183 */
  compute_filled_clusters(o);                                     /* mmat.rg:1200-1203 */
  if (err) *err = 0;
  return o;
#undef FAIL
}

void orc_free(Orc *o)
{
  if (!o) return;
  int ns = o->nsep;
  if (o->blk)
    for (size_t i = 0; i < (size_t)(ns + 1) * (ns + 1); i++) { free(o->blk[i].data); free(o->blk[i].filled); free(o->blk[i].tbounds); }
  if (o->snap)
    for (size_t i = 0; i < (size_t)o->levels * (ns + 1) * (ns + 1); i++) free(o->snap[i].v);
  if (o->cl)
    for (int s = 1; s <= ns; s++) { for (int t = 0; t < o->cl[s].n_int; t++) free(o->cl[s].b[t]); free(o->cl[s].b); free(o->cl[s].len); }
  free(o->blk); free(o->snap); free(o->cl); free(o->sep_of_pos); free(o->dof_of_pos); free(o->sep_size);
  free(o->sep_off); free(o->tree_node); free(o->Aorig); free(o->ops); free(o);
}

/* ---------------------------------------------------------------------------------------- */
/* Numeric phase: fused leaf tasks + level schedule                                           */
/* ---------------------------------------------------------------------------------------- */
static double *raw_ptr(Block *B, rect2 r, int *ld)   /* get_raw_ptr_2d, blas.rg:35-43 */
{
  *ld = B->rows;
  return B->data + (r.lo_x - B->bounds.lo_x) + (size_t)(r.lo_y - B->bounds.lo_y) * B->rows;
}

static void log_op(Orc *o, int op, int level, int m, int n, int k, const Filled *a, const Filled *b, const Filled *c)
{
  double f = 0;
  switch (op) {
    case 0: f = (double)n * n * n / 3.0; break;          /* POTRF n^3/3 */
    case 1: f = (double)m * n * n; break;                /* TRSM m n^2  */
    case 2: f = (double)n * (n + 1) * k; break;          /* SYRK n(n+1)k */
    case 3: f = 2.0 * m * n * k; break;                  /* GEMM 2mnk */
  }
  if (o->quiet) return; /* level-parallel runs: no shared counters */
  o->flops[op] += f; o->calls[op]++;
  if (level < 16) { o->level_flops[level][op] += f; o->level_calls[level][op]++; }
  if (!o->log_ops) return;
  if (o->nops == o->cap_ops) { o->cap_ops = o->cap_ops ? 2 * o->cap_ops : 4096; o->ops = realloc(o->ops, o->cap_ops * sizeof(OpRec)); }
  OpRec r = { op, level, m, n, k, a ? a->sep_x : 0, a ? a->sep_y : 0, a ? a->cluster : 0,
              b ? b->sep_x : 0, b ? b->sep_y : 0, b ? b->cluster : 0, c ? c->sep_x : 0, c ? c->sep_y : 0, c ? c->cluster : 0 };
  o->ops[o->nops++] = r;
}

static void fused_dpotrf(Orc *o, Block *rA, FilledList *fa, int level)
{
  /* blas.rg:292-315 */
  for (int i = 0; i < fa->n; i++) {
    Filled *a = &fa->v[i];
    int m = a->bounds.hi_x - a->bounds.lo_x + 1, ld;
    double *A = raw_ptr(rA, a->bounds, &ld);
    if (m != 0) {                                            /* blas.rg:68 */
      int info = B_potrf(ColMajor, 'L', m, A, ld);          /* blas.rg:71 (info ignored there) */
      if (info != 0 && o->info == 0) o->info = info;
      log_op(o, 0, level, m, m, 0, a, NULL, NULL);
    }
  }
}

static void fused_dtrsm(Orc *o, Block *rA, Block *rB, FilledList *fa, FilledList *fb, int level)
{
  /* blas.rg:317-351 */
  for (int i = 0; i < fa->n; i++) {
    Filled *a = &fa->v[i];
    int lda; double *A = raw_ptr(rA, a->bounds, &lda);
    for (int j = 0; j < fb->n; j++) {
      Filled *b = &fb->v[j];
      int m = b->bounds.hi_x - b->bounds.lo_x + 1, n = b->bounds.hi_y - b->bounds.lo_y + 1, ldb;
      double *Bp = raw_ptr(rB, b->bounds, &ldb);
      B_trsm(ColMajor, Right, Lower, Trans, NonUnit, m, n, 1.0, A, lda, Bp, ldb);   /* blas.rg:99 */
      log_op(o, 1, level, m, n, 0, a, b, NULL);
    }
  }
}

static void fused_update(Orc *o, Block *rA, Block *rB, Block *rC, FilledList *fa, FilledList *fb, FilledList *fc,
                         int col_cluster_size, int level, int is_syrk)
{
  /* blas.rg:353-436 (is_syrk) and :438-504 */
  for (int i = 0; i < fa->n; i++) {
    Filled *a = &fa->v[i];
    int row = a->cluster;
    int sAx = a->bounds.hi_x - a->bounds.lo_x + 1, sAy = a->bounds.hi_y - a->bounds.lo_y + 1;
    for (int j = 0; j < fb->n; j++) {
      Filled *b = &fb->v[j];
      int col = b->cluster;
      int sBx = b->bounds.hi_x - b->bounds.lo_x + 1;
      int cz = row * col_cluster_size + col;
      Filled *c = NULL;
      for (int k = 0; k < fc->n; k++)                        /* linear search, blas.rg:385-392 */
        if (fc->v[k].sep_x == a->sep_x && fc->v[k].sep_y == b->sep_x && fc->v[k].cluster == cz) { c = &fc->v[k]; break; }
      if (!c) continue;
      int sCx = c->bounds.hi_x - c->bounds.lo_x + 1, sCy = c->bounds.hi_y - c->bounds.lo_y + 1;
      if (sCx <= 0 || sCy <= 0) continue;                    /* vol == 0 */
      int lda, ldb, ldc;
      double *A = raw_ptr(rA, a->bounds, &lda), *Bp = raw_ptr(rB, b->bounds, &ldb), *C = raw_ptr(rC, c->bounds, &ldc);
      if (is_syrk) {
        if (col < row) {
          B_gemm(ColMajor, NoTrans, Trans, sAx, sBx, sAy, -1.0, A, lda, Bp, ldb, 1.0, C, ldc);   /* blas.rg:412 */
          log_op(o, 3, level, sAx, sBx, sAy, a, b, c);
        } else if (col == row) {
          B_syrk(ColMajor, Lower, NoTrans, sCx, sAy, -1.0, A, lda, 1.0, C, ldc);                /* blas.rg:429 */
          log_op(o, 2, level, sCx, sCx, sAy, a, b, c);
        }
      } else {
        B_gemm(ColMajor, NoTrans, Trans, sAx, sBx, sAy, -1.0, A, lda, Bp, ldb, 1.0, C, ldc);     /* blas.rg:497 */
        log_op(o, 3, level, sAx, sBx, sAy, a, b, c);
      }
    }
  }
}

static void refill(Orc *o)
{
  int ns = o->nsep;                                           /* mmat.rg:1216-1224 */
  for (int r = 1; r <= ns; r++)
    for (int c = 1; c <= ns; c++)
      if (BLK(o, r, c).allocated) fill_block(o, r, c, 0);
}

static void factor_levels(Orc *o)
{
  /* mmat.rg:1227-1355 */
  int interval = 0, interval_lbl = 0;
  for (int lvl = o->levels - 1; lvl >= 0; lvl--) {
    for (int si = lvl_lo(lvl); si <= lvl_hi(lvl); si++) {
      int sep = o->tree_node[si];
      fused_dpotrf(o, &BLK(o, sep, sep), &SNAP(o, interval_lbl, sep, sep), lvl);
    }
    for (int si = lvl_lo(lvl); si <= lvl_hi(lvl); si++) {
      int sep = o->tree_node[si], pi = si;
      for (int pl = lvl - 1; pl >= 0; pl--) {
        pi = pi / 2;
        int par = o->tree_node[pi];
        fused_dtrsm(o, &BLK(o, sep, sep), &BLK(o, par, sep), &SNAP(o, interval_lbl, sep, sep), &SNAP(o, interval_lbl, par, sep), lvl);
      }
    }
    for (int si = lvl_lo(lvl); si <= lvl_hi(lvl); si++) {
      int sep = o->tree_node[si], pi = si;
      for (int pl = lvl - 1; pl >= 0; pl--) {
        pi = pi / 2;
        int par = o->tree_node[pi], gi = pi;
        for (int gl = pl; gl >= 0; gl--) {
          int gp = o->tree_node[gi];
          int ccs = ntiles(o, par, interval);
          fused_update(o, &BLK(o, gp, sep), &BLK(o, par, sep), &BLK(o, gp, par), &SNAP(o, interval_lbl, gp, sep),
                       &SNAP(o, interval_lbl, par, sep), &SNAP(o, interval_lbl, gp, par), ccs, lvl, gp == par);
          gi = gi / 2;
        }
      }
    }
    interval_lbl++;
    if (lvl <= o->levels - 2) interval++;
  }
}

static double now_s(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* The same level loop with the task parallelism the reference runs with (`-ll:cpu N`, test_matrices.py:27: 3 workers; OpenBLAS
 * one thread per call, mmat.rg:1057): inside a level the POTRF tasks of the separators are independent, so are the TRSM tasks,
 * and the SYRK / GEMM tasks conflict only through their target block C = (gp, par) ("reads writes(rC)", blas.rg:364-367), where
 * Legion serialises them in program order.  Restated with OpenMP: parallel loops over the separators for the first two sweeps
 * and over the TARGET BLOCKS for the third, each block receiving its contributions in program order -- the same values as the
 * sequential loop, bit for bit. */
typedef struct { int sep, par, gp; } UpdTask;
static void factor_levels_parallel(Orc *o, int workers)
{
  int interval = 0, interval_lbl = 0;
  const int ns = o->nsep;
  UpdTask *tk = malloc((size_t)(ns + 1) * 64 * sizeof(UpdTask));
  int *order = malloc((size_t)(ns + 1) * 64 * sizeof(int));
  for (int lvl = o->levels - 1; lvl >= 0; lvl--) {
    const int lo = lvl_lo(lvl), hi = lvl_hi(lvl);
#pragma omp parallel for schedule(dynamic, 1) num_threads(workers)
    for (int si = lo; si <= hi; si++) {
      int sep = o->tree_node[si];
      fused_dpotrf(o, &BLK(o, sep, sep), &SNAP(o, interval_lbl, sep, sep), lvl);
    }
    /* one TRSM task per (separator, ancestor): all independent */
    int nt = 0;
    for (int si = lo; si <= hi; si++) {
      int pi = si;
      for (int pl = lvl - 1; pl >= 0; pl--) { pi = pi / 2; tk[nt].sep = o->tree_node[si]; tk[nt].par = o->tree_node[pi]; tk[nt].gp = 0; nt++; }
    }
#pragma omp parallel for schedule(dynamic, 1) num_threads(workers)
    for (int t = 0; t < nt; t++)
      fused_dtrsm(o, &BLK(o, tk[t].sep, tk[t].sep), &BLK(o, tk[t].par, tk[t].sep), &SNAP(o, interval_lbl, tk[t].sep, tk[t].sep),
                  &SNAP(o, interval_lbl, tk[t].par, tk[t].sep), lvl);
    /* update tasks in program order, then grouped by target block (stable): a group runs on one thread, in order */
    nt = 0;
    for (int si = lo; si <= hi; si++) {
      int sep = o->tree_node[si], pi = si;
      for (int pl = lvl - 1; pl >= 0; pl--) {
        pi = pi / 2;
        int par = o->tree_node[pi], gi = pi;
        for (int gl = pl; gl >= 0; gl--) { tk[nt].sep = sep; tk[nt].par = par; tk[nt].gp = o->tree_node[gi]; nt++; gi = gi / 2; }
      }
    }
    int ng = 0; /* order[] = task indices grouped by (gp, par); gstart via a second pass */
    int *gstart = malloc((size_t)(nt + 1) * sizeof(int));
    char *done = calloc((size_t)nt + 1, 1);
    for (int t = 0; t < nt; t++) {
      if (done[t]) continue;
      gstart[ng++] = 0;
      for (int u = t; u < nt; u++)
        if (!done[u] && tk[u].gp == tk[t].gp && tk[u].par == tk[t].par) done[u] = 1;
    }
    /* fill order / gstart */
    memset(done, 0, (size_t)nt + 1);
    int pos = 0; ng = 0;
    for (int t = 0; t < nt; t++) {
      if (done[t]) continue;
      gstart[ng++] = pos;
      for (int u = t; u < nt; u++)
        if (!done[u] && tk[u].gp == tk[t].gp && tk[u].par == tk[t].par) { done[u] = 1; order[pos++] = u; }
    }
    gstart[ng] = pos;
    const int ccs_interval = interval;
#pragma omp parallel for schedule(dynamic, 1) num_threads(workers)
    for (int g = 0; g < ng; g++)
      for (int q = gstart[g]; q < gstart[g + 1]; q++) {
        const UpdTask *u = &tk[order[q]];
        fused_update(o, &BLK(o, u->gp, u->sep), &BLK(o, u->par, u->sep), &BLK(o, u->gp, u->par), &SNAP(o, interval_lbl, u->gp, u->sep),
                     &SNAP(o, interval_lbl, u->par, u->sep), &SNAP(o, interval_lbl, u->gp, u->par), ntiles(o, u->par, ccs_interval), lvl, u->gp == u->par);
      }
    free(gstart); free(done);
    interval_lbl++;
    if (lvl <= o->levels - 2) interval++;
  }
  free(tk); free(order);
}
double orc_factor_parallel(Orc *o, int workers)
{
  o->log_ops = 0; o->nops = 0; o->info = 0; o->quiet = 1;
  refill(o);
  double t0 = now_s();
  factor_levels_parallel(o, workers < 1 ? 1 : workers);
  double dt = now_s() - t0;
  o->quiet = 0;
  return dt;
}

/* One reference "iteration" (mmat.rg:1212-1358): re-fill, then the level loop.  Returns the
 * seconds spent in the level loop only (the timed region of SURVEY 8d). */
double orc_factor(Orc *o, int log_ops)
{
  o->log_ops = log_ops; o->nops = 0; o->info = 0;
  memset(o->flops, 0, sizeof o->flops); memset(o->calls, 0, sizeof o->calls);
  memset(o->level_flops, 0, sizeof o->level_flops); memset(o->level_calls, 0, sizeof o->level_calls);
  refill(o);
  double t0 = now_s();
  factor_levels(o);
  return now_s() - t0;
}

/* ---------------------------------------------------------------------------------------- */
/* Solve, mmat.rg:1364-1495                                                                   */
/* ---------------------------------------------------------------------------------------- */
void orc_solve(Orc *o, const double *b_in, double *x_out)
{
  int N = o->N, ns = o->nsep;
  double *Bv = malloc(N * sizeof(double));
  for (int p = 0; p < N; p++) Bv[p] = b_in[o->dof_of_pos[p]];             /* fill_b, :769-783 */
  for (int lvl = o->levels - 1; lvl >= 0; lvl--)                            /* forward, :1395-1435 */
    for (int si = lvl_lo(lvl); si <= lvl_hi(lvl); si++) {
      int sep = o->tree_node[si];
      Block *P = &BLK(o, sep, sep);
      B_trsv(ColMajor, Lower, NoTrans, NonUnit, P->rows, P->data, P->rows, Bv + o->sep_off[sep], 1);
      int pi = si;
      for (int pl = lvl - 1; pl >= 0; pl--) {
        pi = pi / 2;
        int par = o->tree_node[pi];
        Block *A = &BLK(o, par, sep);
        B_gemv(ColMajor, NoTrans, A->rows, A->cols, -1.0, A->data, A->rows, Bv + o->sep_off[sep], 1, 1.0, Bv + o->sep_off[par], 1);
      }
    }
  for (int pl = 0; pl < o->levels; pl++)                                    /* backward, :1438-1479 */
    for (int pi = lvl_lo(pl); pi <= lvl_hi(pl); pi++) {
      int par = o->tree_node[pi];
      Block *P = &BLK(o, par, par);
      B_trsv(ColMajor, Lower, Trans, NonUnit, P->rows, P->data, P->rows, Bv + o->sep_off[par], 1);
      for (int lvl = pl + 1; lvl < o->levels; lvl++)
        for (int si = pi << (lvl - pl); si < ((pi + 1) << (lvl - pl)); si++) {
          int sep = o->tree_node[si];
          Block *A = &BLK(o, par, sep);
          if (A->rows * A->cols != 0)
            B_gemv(ColMajor, Trans, A->rows, A->cols, -1.0, A->data, A->rows, Bv + o->sep_off[par], 1, 1.0, Bv + o->sep_off[sep], 1);
        }
    }
  for (int p = 0; p < N; p++) x_out[o->dof_of_pos[p]] = Bv[p];              /* :1483-1491 */
  (void)ns;
  free(Bv);
}

/* ---------------------------------------------------------------------------------------- */
/* Accessors for the tests                                                                    */
/* ---------------------------------------------------------------------------------------- */
int orc_N(Orc *o) { return o->N; }
int orc_NZ(Orc *o) { return o->NZ; }
int orc_levels(Orc *o) { return o->levels; }
int orc_nsep(Orc *o) { return o->nsep; }
int orc_max_int_size(Orc *o) { return o->max_int_size; }
int orc_info(Orc *o) { return o->info; }
const char *orc_banner(Orc *o) { return o->banner; }
void orc_perm(Orc *o, int *out) { memcpy(out, o->dof_of_pos, o->N * sizeof(int)); }
void orc_sep_sizes(Orc *o, int *out) { for (int s = 1; s <= o->nsep; s++) out[s - 1] = o->sep_size[s]; }
void orc_sep_offsets(Orc *o, int *out) { for (int s = 1; s <= o->nsep; s++) out[s - 1] = o->sep_off[s]; }
void orc_tree(Orc *o, int *out) { for (int i = 1; i <= o->nsep; i++) out[i - 1] = o->tree_node[i]; }
int orc_num_blocks(Orc *o)
{
  int n = 0;
  for (int r = 1; r <= o->nsep; r++) for (int c = 1; c <= o->nsep; c++) n += BLK(o, r, c).allocated;
  return n;
}
/* out: per allocated block (row-major over (r,c)): r, c, lo_x, lo_y, hi_x, hi_y */
void orc_blocks(Orc *o, int *out)
{
  int k = 0;
  for (int r = 1; r <= o->nsep; r++)
    for (int c = 1; c <= o->nsep; c++) {
      Block *B = &BLK(o, r, c);
      if (!B->allocated) continue;
      out[k++] = r; out[k++] = c; out[k++] = B->bounds.lo_x; out[k++] = B->bounds.lo_y; out[k++] = B->bounds.hi_x; out[k++] = B->bounds.hi_y;
    }
}
/* number of filled tiles in the snapshot `lbl`, all blocks */
int orc_snapshot_count(Orc *o, int lbl)
{
  int n = 0;
  for (int r = 1; r <= o->nsep; r++) for (int c = 1; c <= o->nsep; c++) n += SNAP(o, lbl, r, c).n;
  return n;
}
/* out: per filled tile: sep_x, sep_y, cluster, lo_x, lo_y, hi_x, hi_y  (7 ints) */
void orc_snapshot(Orc *o, int lbl, int *out)
{
  int k = 0;
  for (int r = 1; r <= o->nsep; r++)
    for (int c = 1; c <= o->nsep; c++) {
      FilledList *fl = &SNAP(o, lbl, r, c);
      for (int i = 0; i < fl->n; i++) {
        Filled *f = &fl->v[i];
        out[k++] = f->sep_x; out[k++] = f->sep_y; out[k++] = f->cluster;
        out[k++] = f->bounds.lo_x; out[k++] = f->bounds.lo_y; out[k++] = f->bounds.hi_x; out[k++] = f->bounds.hi_y;
      }
    }
}
void orc_counts(Orc *o, long *calls4, double *flops4) { for (int i = 0; i < 4; i++) { calls4[i] = o->calls[i]; flops4[i] = o->flops[i]; } }
void orc_level_counts(Orc *o, int level, long *calls4, double *flops4)
{
  for (int i = 0; i < 4; i++) { calls4[i] = o->level_calls[level][i]; flops4[i] = o->level_flops[level][i]; }
}
int orc_num_ops(Orc *o) { return o->nops; }
/* out: 14 ints per op: op, level, m, n, k, a(sx,sy,z), b(sx,sy,z), c(sx,sy,z) */
void orc_ops(Orc *o, int *out)
{
  for (int i = 0; i < o->nops; i++) {
    OpRec *r = &o->ops[i]; int *q = out + 14 * (size_t)i;
    q[0] = r->op; q[1] = r->level; q[2] = r->m; q[3] = r->n; q[4] = r->k; q[5] = r->a_sx; q[6] = r->a_sy; q[7] = r->a_z;
    q[8] = r->b_sx; q[9] = r->b_sy; q[10] = r->b_z; q[11] = r->c_sx; q[12] = r->c_sy; q[13] = r->c_z;
  }
}
/* Dense N x N col-major copy of the block storage (zeros elsewhere) */
void orc_dense(Orc *o, double *out)
{
  int N = o->N;
  memset(out, 0, (size_t)N * N * sizeof(double));
  for (int r = 1; r <= o->nsep; r++)
    for (int c = 1; c <= o->nsep; c++) {
      Block *B = &BLK(o, r, c);
      if (!B->allocated) continue;
      for (int j = 0; j < B->cols; j++)
        for (int i = 0; i < B->rows; i++)
          out[(size_t)(B->bounds.lo_x + i) + (size_t)(B->bounds.lo_y + j) * N] = B->data[i + (size_t)j * B->rows];
    }
}
/* nnz(L) as the reference would count it: non-zeros of all allocated blocks (write_matrix) */
long orc_nnz(Orc *o)
{
  long n = 0;
  for (int r = 1; r <= o->nsep; r++)
    for (int c = 1; c <= o->nsep; c++) {
      Block *B = &BLK(o, r, c);
      if (!B->allocated) continue;
      for (size_t i = 0; i < (size_t)B->rows * B->cols; i++) n += (B->data[i] != 0.0);
    }
  return n;
}
/* write_matrix, mmat.rg:102-147: banner copied, "M N nnz", then "row col %0.8g" block by block,
 * row-major inside a block, 1-based permuted coordinates.  `fmt17` != 0 writes %.17g instead. */
int orc_write_matrix(Orc *o, const char *file, int fmt17)
{
  FILE *fp = fopen(file, "w");
  if (!fp) return -1;
  fprintf(fp, "%s\n", o->banner);
  fprintf(fp, "%d %d %ld\n", o->M, o->N, orc_nnz(o));
  for (int r = 1; r <= o->nsep; r++)
    for (int c = 1; c <= o->nsep; c++) {
      Block *B = &BLK(o, r, c);
      if (!B->allocated) continue;
      for (int i = 0; i < B->rows; i++)
        for (int j = 0; j < B->cols; j++) {
          double v = B->data[i + (size_t)j * B->rows];
          if (v != 0) fprintf(fp, fmt17 ? "%d %d %.17g\n" : "%d %d %0.8g\n", B->bounds.lo_x + i + 1, B->bounds.lo_y + j + 1, v);
        }
    }
  fclose(fp);
  return 0;
}

/* ---------------------------------------------------------------------------------------- */
/* Stand-alone access to the oracle's BLAS restatements (checker for the L-A entry points)    */
/* ---------------------------------------------------------------------------------------- */
int orc_blas_potrf(int n, double *a, int lda) { return B_potrf(ColMajor, 'L', n, a, lda); }
void orc_blas_trsm(int m, int n, const double *a, int lda, double *b, int ldb) { B_trsm(ColMajor, Right, Lower, Trans, NonUnit, m, n, 1.0, a, lda, b, ldb); }
void orc_blas_syrk(int n, int k, const double *a, int lda, double *c, int ldc) { B_syrk(ColMajor, Lower, NoTrans, n, k, -1.0, a, lda, 1.0, c, ldc); }
void orc_blas_gemm(int m, int n, int k, const double *a, int lda, const double *b, int ldb, double *c, int ldc)
{
  B_gemm(ColMajor, NoTrans, Trans, m, n, k, -1.0, a, lda, b, ldb, 1.0, c, ldc);
}
void orc_blas_trsv(int trans, int n, const double *a, int lda, double *x) { B_trsv(ColMajor, Lower, trans, NonUnit, n, a, lda, x, 1); }
void orc_blas_gemv(int trans, int m, int n, const double *a, int lda, const double *x, double *y) { B_gemv(ColMajor, trans, m, n, -1.0, a, lda, x, 1, 1.0, y, 1); }
