/* File ingest of libcholamd: Matrix-Market banner/size I/O and the separator / cluster / matrix /
 * vector readers.  Behavioural contract = the reference's mmio.c (the four entry points mmat.rg
 * calls) and mnd.c (file formats); the implementation is new and fills plain
 * arrays instead of Legion accessors.  Unlike the reference every fopen/parse is checked.
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <errno.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "chol_plan.h"

/* ---------------------------------------------------------------------------------------- */
static __thread char g_err[512];

void chol_set_error(const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}
const char *cholamd_last_error(void) { return g_err; }
const char *cholamd_version(void) { return "cholamd 0.1 (gfx950)"; }

/* ---------------------------------------------------------------------------------------- */
/* Matrix-Market banner: "%%MatrixMarket <object> <format> <field> <symmetry>"                */
/* typecode letters (mmio.h:88-100): [0] M  [1] C|A  [2] R|C|P|I  [3] G|S|H|K                  */
/* ---------------------------------------------------------------------------------------- */
struct word_code { const char *word; char code; };
static const struct word_code k_format[] = { { "coordinate", 'C' }, { "array", 'A' }, { NULL, 0 } };
static const struct word_code k_field[] = { { "real", 'R' }, { "complex", 'C' }, { "pattern", 'P' }, { "integer", 'I' }, { NULL, 0 } };
static const struct word_code k_symm[] = { { "general", 'G' }, { "symmetric", 'S' }, { "hermitian", 'H' }, { "skew-symmetric", 'K' }, { NULL, 0 } };

static char lookup_code(const struct word_code *tab, const char *w)
{
  for (; tab->word; tab++)
    if (strcmp(tab->word, w) == 0) return tab->code;
  return 0;
}
static const char *lookup_word(const struct word_code *tab, char c)
{
  for (; tab->word; tab++)
    if (tab->code == c) return tab->word;
  return NULL;
}
static void lower_inplace(char *s)
{
  for (; *s; s++) *s = (char)tolower((unsigned char)*s);
}

int mm_read_banner(FILE *f, MM_typecode *matcode)
{
  char line[1025], tag[64], obj[64], fmt[64], fld[64], sym[64];
  (*matcode)[0] = (*matcode)[1] = (*matcode)[2] = ' ';
  (*matcode)[3] = 'G';
  if (!f || !fgets(line, sizeof line, f)) return MM_PREMATURE_EOF;
  if (sscanf(line, "%63s %63s %63s %63s %63s", tag, obj, fmt, fld, sym) != 5) return MM_PREMATURE_EOF;
  lower_inplace(obj); lower_inplace(fmt); lower_inplace(fld); lower_inplace(sym);
  if (strncmp(tag, "%%MatrixMarket", 14) != 0) return MM_NO_HEADER;
  if (strcmp(obj, "matrix") != 0) return MM_UNSUPPORTED_TYPE;
  (*matcode)[0] = 'M';
  char c;
  if (!(c = lookup_code(k_format, fmt))) return MM_UNSUPPORTED_TYPE;
  (*matcode)[1] = c;
  if (!(c = lookup_code(k_field, fld))) return MM_UNSUPPORTED_TYPE;
  (*matcode)[2] = c;
  if (!(c = lookup_code(k_symm, sym))) return MM_UNSUPPORTED_TYPE;
  (*matcode)[3] = c;
  return 0; /* deliberately no validity check: "real hermitian" passes, as in the reference */
}

int mm_read_mtx_crd_size(FILE *f, int *M, int *N, int *nz)
{
  char line[1025];
  *M = *N = *nz = 0;
  for (;;) { /* skip comment lines, then tolerate blank lines before the size line */
    if (!fgets(line, sizeof line, f)) return MM_PREMATURE_EOF;
    if (line[0] == '%') continue;
    if (sscanf(line, "%d %d %d", M, N, nz) == 3) return 0;
    const char *q = line;
    while (*q && isspace((unsigned char)*q)) q++;
    if (*q) { /* a non-blank line that is not "M N nz": keep scanning tokens like the reference does */
      int got = fscanf(f, "%d %d %d", M, N, nz);
      if (got == EOF) return MM_PREMATURE_EOF;
      if (got == 3) return 0;
    }
  }
}

char *mm_typecode_to_str(MM_typecode matcode)
{
  char buf[128];
  const char *fmt = lookup_word(k_format, matcode[1]);
  const char *fld = lookup_word(k_field, matcode[2]);
  const char *sym = lookup_word(k_symm, matcode[3]);
  if (matcode[0] != 'M' || !fmt || !fld || !sym) return NULL;
  snprintf(buf, sizeof buf, "matrix %s %s %s", fmt, fld, sym);
  return strdup(buf);
}

int mm_write_banner(FILE *f, MM_typecode matcode)
{
  char *s = mm_typecode_to_str(matcode);
  if (!s) return MM_COULD_NOT_WRITE_FILE;
  int rc = fprintf(f, "%%%%MatrixMarket %s\n", s);
  free(s);
  return rc < 0 ? MM_COULD_NOT_WRITE_FILE : 0;
}

int mm_write_mtx_crd_size(FILE *f, int M, int N, int nz)
{
  return fprintf(f, "%d %d %d\n", M, N, nz) < 0 ? MM_COULD_NOT_WRITE_FILE : 0;
}

/* ---------------------------------------------------------------------------------------- */
/* separators: line 0 "levels num_separators"; then "k;d0,d1,...,dn,"            */
/* ---------------------------------------------------------------------------------------- */
int cholamd_read_separators(const char *file, int dim, int *idx_out, int *sep_out, cholamd_sepinfo *info)
{
  FILE *fp = fopen(file, "r");
  if (!fp) { chol_set_error("cannot open separator file %s: %s", file, strerror(errno)); return CHOLAMD_ERR_IO; }
  char *line = NULL; size_t cap = 0; int lineno = 0, pos = 0, rc = 0;
  info->levels = info->num_separators = 0;
  while (getline(&line, &cap, fp) != -1) {
    if (lineno++ == 0) {
      /* the reference reads atoi(&line[0]) and atoi(&line[2]) (single-digit levels, mnd.c:42-43);
       * a general two-integer parse is a superset of that */
      if (sscanf(line, "%d %d", &info->levels, &info->num_separators) != 2) { rc = CHOLAMD_ERR_FORMAT; break; }
      continue;
    }
    char *semi = strchr(line, ';');
    if (!semi) continue; /* blank trailer */
    int label = atoi(line) + 1;
    const char *q = semi + 1;
    while (*q) {
      while (*q == ',' ) q++;
      if (!*q || isspace((unsigned char)*q)) break;
      char *end;
      long v = strtol(q, &end, 10);
      if (end == q) { rc = CHOLAMD_ERR_FORMAT; break; }
      if (pos >= dim) { rc = CHOLAMD_ERR_FORMAT; break; }
      idx_out[pos] = (int)v;
      sep_out[pos] = label;
      pos++;
      q = end;
    }
    if (rc) break;
  }
  free(line);
  fclose(fp);
  if (rc == 0 && pos != dim) rc = CHOLAMD_ERR_FORMAT;
  if (rc) chol_set_error("separator file %s: malformed (read %d of %d dofs)", file, pos, dim);
  return rc;
}

/* ---------------------------------------------------------------------------------------- */
/* clusters: line 0 ignored; then "k;b0,b1,..,;c0,c1,..,;...;" one boundary list per interval  */
/* (SURVEY A.3).  The reference tokenises on ",; " and starts a new interval at every token    */
/* "0"; since each list is written as its own ';'-terminated group that starts with 0, parsing */
/* the groups directly yields the same triples.  max_int_size reproduces the reference's       */
/* return value including its quirk (the newline token is counted in the last interval).       */
/* ---------------------------------------------------------------------------------------- */
int cholamd_read_clusters(const char *file, int *idx_out, int *interval_out, int *sep_out, int64_t cap_out, int64_t *count_out)
{
  FILE *fp = fopen(file, "r");
  if (!fp) { chol_set_error("cannot open clusters file %s: %s", file, strerror(errno)); return CHOLAMD_ERR_IO; }
  char *line = NULL; size_t cap = 0; int lineno = 0, max_int = -1, rc = 0;
  int64_t count = 0;
  while (getline(&line, &cap, fp) != -1) {
    if (lineno++ == 0) continue;
    char *semi = strchr(line, ';');
    if (!semi) continue;
    int label = atoi(line) + 1;
    int interval = 0, in_list = 0, last_len = 0;
    const char *q = semi + 1;
    for (;;) {
      while (*q == ',' || *q == ' ') q++;
      if (*q == ';') { /* end of one boundary list */
        if (in_list > max_int) max_int = in_list;
        if (in_list > 0) { interval++; last_len = in_list; }
        in_list = 0;
        q++;
        continue;
      }
      if (!*q || *q == '\n' || *q == '\r') {
        /* the reference counts the trailing newline token into the running interval length */
        if (last_len + 1 > max_int) max_int = last_len + 1;
        break;
      }
      char *end;
      long v = strtol(q, &end, 10);
      if (end == q) { rc = CHOLAMD_ERR_FORMAT; break; }
      if (in_list == 0 && v != 0) { rc = CHOLAMD_ERR_FORMAT; break; } /* every list starts with 0 */
      if (count < cap_out) { idx_out[count] = (int)v; interval_out[count] = interval; sep_out[count] = label; }
      count++;
      in_list++;
      q = end;
    }
    if (rc) break;
  }
  free(line);
  fclose(fp);
  if (count_out) *count_out = count;
  if (rc) { chol_set_error("clusters file %s: malformed", file); return rc; }
  return max_int;
}

/* ---------------------------------------------------------------------------------------- */
/* matrix body: exactly two lines skipped, then nz "i j val" lines, 1-based      */
/* ---------------------------------------------------------------------------------------- */
int cholamd_read_matrix(const char *file, int nz, int *row_out, int *col_out, double *val_out)
{
  FILE *fp = fopen(file, "r");
  if (!fp) { chol_set_error("cannot open matrix file %s: %s", file, strerror(errno)); return CHOLAMD_ERR_IO; }
  char buf[1025];
  if (!fgets(buf, sizeof buf, fp) || !fgets(buf, sizeof buf, fp)) { fclose(fp); chol_set_error("%s: truncated header", file); return CHOLAMD_ERR_FORMAT; }
  for (int k = 0; k < nz; k++) {
    unsigned long i, j; double v;
    if (fscanf(fp, "%lu %lu %lg", &i, &j, &v) != 3 || i == 0 || j == 0) {
      fclose(fp);
      chol_set_error("%s: entry %d of %d malformed", file, k, nz);
      return CHOLAMD_ERR_FORMAT;
    }
    row_out[k] = (int)(i - 1); col_out[k] = (int)(j - 1); val_out[k] = v;
  }
  fclose(fp);
  return 0;
}

/* rhs vector: three header lines skipped blindly, then n values */
int cholamd_read_vector(const char *file, int n, double *out)
{
  FILE *fp = fopen(file, "r");
  if (!fp) { chol_set_error("cannot open vector file %s: %s", file, strerror(errno)); return CHOLAMD_ERR_IO; }
  char buf[1025];
  for (int i = 0; i < 3; i++)
    if (!fgets(buf, sizeof buf, fp)) { fclose(fp); chol_set_error("%s: truncated header", file); return CHOLAMD_ERR_FORMAT; }
  for (int i = 0; i < n; i++)
    if (fscanf(fp, "%lg", &out[i]) != 1) { fclose(fp); chol_set_error("%s: value %d of %d missing", file, i, n); return CHOLAMD_ERR_FORMAT; }
  fclose(fp);
  return 0;
}

/* write_solution, mmat.rg:785-798 */
int cholamd_write_solution(const char *file, const double *x, int n, int full_precision)
{
  FILE *fp = fopen(file, "w");
  if (!fp) { chol_set_error("cannot write %s: %s", file, strerror(errno)); return CHOLAMD_ERR_IO; }
  for (int i = 0; i < n; i++) fprintf(fp, full_precision ? "%.17g\n" : "%0.8g\n", x[i]);
  fclose(fp);
  return 0;
}
