/* cholamd_mmat -- the C driver with mmat.rg's command line (mmat.rg:1059-1093):
 *
 *   -i matrix.mtx  -s separators.txt  -c clusters.txt  [-b rhs.mtx] [-o solution] [-m factor.mtx]
 *   [-p permuted.mtx] [-d debug_dir] [--iterations n]
 * plus:  --gpu id (first device, default 0), --gpus n (subtree-sharded over n devices of this node, one RCCL all-reduce;
 *        n a power of two), --repeat n (= --iterations), --full-precision (write %.17g instead of the reference's
 *        %0.8g), --precision fp64|mixed (mixed: fp32 factor + fp64 iterative refinement of the solve).
 * Unknown flags (the reference passes -fflow/-ll:cpu/-fcuda/-ll:csize through to Legion) are ignored.
 *
 * Flow = main() of mmat.rg:1056-1496 with the numeric phase on the GPU.  Progress lines keep the
 * reference's wording.  There is no CPU numeric path: without a HIP device the program fails.
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "cholamd.h"

static double now_s(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 1; } while (0)

int main(int argc, char **argv)
{
  const char *matrix_file = "", *separator_file = "", *clusters_file = "", *b_file = "", *solution_file = "", *factor_file = "",
             *permuted_file = "", *debug_path = "";
  int debug = 0, iterations = 1, gpu = 0, full = 0, gpus = 1;
  const char *precision = "fp64";
  for (int i = 0; i < argc; i++) {
    const char *next = i + 1 < argc ? argv[i + 1] : "";
    if (!strcmp(argv[i], "-i")) matrix_file = next;
    else if (!strcmp(argv[i], "-s")) separator_file = next;
    else if (!strcmp(argv[i], "-c")) clusters_file = next;
    else if (!strcmp(argv[i], "-m")) factor_file = next;
    else if (!strcmp(argv[i], "-p")) permuted_file = next;
    else if (!strcmp(argv[i], "-o")) solution_file = next;
    else if (!strcmp(argv[i], "-b")) b_file = next;
    else if (!strcmp(argv[i], "-d")) { debug_path = next; debug = 1; }
    else if (!strcmp(argv[i], "--iterations")) iterations = atoi(next);
    else if (!strcmp(argv[i], "--repeat")) iterations = atoi(next);
    else if (!strcmp(argv[i], "--gpu")) gpu = atoi(next);
    else if (!strcmp(argv[i], "--gpus")) gpus = atoi(next);
    else if (!strcmp(argv[i], "--precision")) precision = next;
    else if (!strcmp(argv[i], "--full-precision")) full = 1;
  }
  printf("Iterations: %d\n", iterations);
  if (!*matrix_file || !*separator_file || !*clusters_file) DIE("usage: %s -i A.mtx -s ord.txt -c clust.txt [-b B.mtx -o x.txt] [-m L.mtx] [-p PAPt.mtx] [-d dir] [--iterations n]", argv[0]);

  double t0 = now_s();
  cholamd_plan *plan = NULL;
  if (cholamd_plan_create(matrix_file, separator_file, clusters_file, &plan)) DIE("plan: %s", cholamd_last_error());
  const int n = cholamd_plan_n(plan), levels = cholamd_plan_levels(plan);
  printf("M: %d N: %d nz: %d typecode: %s\n", n, n, cholamd_plan_nz(plan), cholamd_plan_banner(plan));
  printf("levels: %d\nseparators: %d\nMax Interval Size: %d\n", levels, cholamd_plan_num_separators(plan), cholamd_plan_max_int_size(plan));
  printf("Blocks ispace: %d\n", cholamd_plan_num_blocks(plan));
  if (cholamd_plan_dropped_entries(plan))
    fprintf(stderr, "warning: %ld matrix entries lie outside every ancestor/descendant block and were dropped\n", (long)cholamd_plan_dropped_entries(plan));
  double t_sym = now_s() - t0;

  const int64_t na = cholamd_plan_arena_doubles(plan);
  double *h_arena = malloc((size_t)na * sizeof(double));
  if (!h_arena) DIE("out of memory");
  if (*permuted_file) { /* mmat.rg:1187-1189 */
    cholamd_plan_fill_host(plan, h_arena);
    printf("saving matrix to: %s\n\n", permuted_file);
    if (cholamd_plan_write_matrix(plan, h_arena, permuted_file, full)) DIE("%s", cholamd_last_error());
  }
  if (debug) { /* Block / Cluster / Fill lines of the symbolic phase on stdout, where verify.debug_factor's log comes from */
    if (gpus > 1 || !strcmp(precision, "mixed")) DIE("-d (debug mode) is a single-GPU fp64 path");
    cholamd_plan_write_debug_header(plan, stdout);
  }

  const int mixed = !strcmp(precision, "mixed");
  if (!mixed && strcmp(precision, "fp64")) DIE("--precision %s: fp64 or mixed", precision);
  if (mixed && gpus > 1) DIE("--precision mixed is a single-GPU path");
  if (gpus < 1 || gpus > 64 || (gpus & (gpus - 1))) DIE("--gpus must be a power of two");
  if (cholamd_device_count() < gpu + gpus) DIE("--gpus %d from device %d: only %d HIP devices are visible", gpus, gpu, cholamd_device_count());
  /* one device object + arena per GPU; devs[0] ends up with the complete factor */
  cholamd_device *devs[64] = { NULL };
  double *arenas[64] = { NULL };
  cholamd_comm *comms[64] = { NULL };
  for (int g = 0; g < gpus; g++) {
    if (cholamd_device_create(plan, gpu + g, &devs[g])) DIE("device %d: %s", gpu + g, cholamd_last_error());
    if (gpus > 1 && cholamd_device_set_partition(devs[g], g, gpus)) DIE("partition: %s", cholamd_last_error());
    if (cholamd_device_alloc(devs[g], na, &arenas[g])) DIE("alloc: %s", cholamd_last_error());
  }
  if (gpus > 1 && cholamd_comm_create_all(devs, gpus, comms)) DIE("rccl: %s", cholamd_last_error());
  cholamd_device *dev = devs[0];
  double *d_arena = arenas[0];
  printf("Done fill.\n");
  double t_factor = 0;
  for (int it = 0; it < iterations; it++) { /* mmat.rg:1212-1358 */
    for (int g = 0; g < gpus; g++)
      if ((mixed ? cholamd_device_fill_f32(devs[g], (float *)arenas[g], NULL) : cholamd_device_fill(devs[g], arenas[g], NULL)) || cholamd_device_sync(devs[g], NULL)) DIE("fill: %s", cholamd_last_error());
    for (int lvl = levels - 1, interval = 0; lvl >= 0; lvl--) {
      printf("Factoring Level: %d Interval: %d Iteration: %d\n", lvl, interval, it);
      if (lvl <= levels - 2) interval++;
    }
    double t1 = now_s();
    if (debug) { /* one fused task at a time, op lines on stdout, write_blocks dumps in debug_path (mmat.rg:1255,1288,1342) */
      if (cholamd_factor_debug(dev, d_arena, debug_path, full, NULL)) DIE("factor (debug): %s", cholamd_last_error());
    } else
    if (mixed ? cholamd_factor_f32(dev, (float *)d_arena, NULL) : cholamd_factor_multi(devs, arenas, comms, gpus, NULL)) DIE("factor: %s", cholamd_last_error());
    for (int g = 0; g < gpus; g++) if (cholamd_device_sync(devs[g], NULL)) DIE("factor: %s", cholamd_last_error());
    t_factor = now_s() - t1;
    for (int g = 0; g < gpus; g++) {
      int sep = 0, info = cholamd_factor_info(devs[g], &sep);
      if (info < 0) DIE("factor: %s", cholamd_last_error()); /* internal failure (stall watchdog / HIP error): no output file is written */
      if (info > 0) fprintf(stderr, "warning: leading minor %d of separator %d is not positive definite\n", info, sep);
    }
    printf("Done factoring Iteration: %d.\n", it);
  }
  if (gpus > 1) { /* the subtrees' panels -> device 0 */
    if (cholamd_gather_factor(devs, arenas, gpus, NULL) || cholamd_device_sync(dev, NULL)) DIE("gather: %s", cholamd_last_error());
  }
  const double flops = cholamd_plan_flops(plan);
  fprintf(stderr, "[cholamd] %d GPU(s), symbolic %.3f ms, numeric factorisation %.3f ms, F_ref %.6g flop, %.3f GF/s, B_alg %ld bytes\n",
          gpus, 1e3 * t_sym, 1e3 * t_factor, flops, flops / t_factor * 1e-9, (long)cholamd_plan_alg_bytes(plan));

  if (*factor_file) { /* mmat.rg:1360-1362 */
    if (cholamd_device_download(dev, h_arena, d_arena, na, NULL)) DIE("download: %s", cholamd_last_error());
    if (mixed) { /* the device arena holds na floats (in the first half of the buffer): widen in place, back to front */
      const float *f = (const float *)h_arena;
      for (int64_t i = na - 1; i >= 0; i--) h_arena[i] = (double)f[i];
    }
    printf("saving matrix to: %s\n\n", factor_file);
    if (cholamd_plan_write_matrix(plan, h_arena, factor_file, full)) DIE("%s", cholamd_last_error());
  }
  if (*b_file) { /* mmat.rg:1364-1495 */
    double *b = malloc((size_t)n * sizeof(double)), *x = malloc((size_t)n * sizeof(double)), *d_b = NULL, *d_x = NULL;
    if (cholamd_read_vector(b_file, n, b)) DIE("%s", cholamd_last_error());
    if (cholamd_device_alloc(dev, n, &d_b) || cholamd_device_alloc(dev, n, &d_x) || cholamd_device_upload(dev, d_b, b, n, NULL)) DIE("%s", cholamd_last_error());
    printf("Forward Substitution\nBackward Substitution\n");
    if (mixed) {
      int iters = 0; double rel = 0.0;
      if (cholamd_solve_refine(dev, (const float *)d_arena, d_b, d_x, 30, 1e-14, &iters, &rel, NULL)) DIE("solve: %s", cholamd_last_error());
      fprintf(stderr, "[cholamd] iterative refinement: %d corrections, ||b - A x|| / ||b|| = %.3e\n", iters, rel);
      if (cholamd_device_download(dev, x, d_x, n, NULL)) DIE("solve: %s", cholamd_last_error());
    } else if (cholamd_solve(dev, d_arena, d_b, d_x, NULL) || cholamd_device_download(dev, x, d_x, n, NULL)) DIE("solve: %s", cholamd_last_error());
    printf("Done solve.\n");
    if (*solution_file) {
      printf("Saving solution to: %s\n", solution_file);
      if (cholamd_write_solution(solution_file, x, n, full)) DIE("%s", cholamd_last_error());
    }
    cholamd_device_free(dev, d_b); cholamd_device_free(dev, d_x);
    free(b); free(x);
  }
  for (int g = 0; g < gpus; g++) {
    cholamd_comm_destroy(comms[g]);
    cholamd_device_free(devs[g], arenas[g]);
    cholamd_device_destroy(devs[g]);
  }
  cholamd_plan_destroy(plan);
  free(h_arena);
  return 0;
}
