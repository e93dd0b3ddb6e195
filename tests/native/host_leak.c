/* Leak check of libcholamd's HOST code (ingest, symbolic phase, schedules, program builder + self-check, generator) without an
 * interpreter in the process: `make asan` builds this against the sanitizer build and runs it with LeakSanitizer on and NO
 * suppressions that could hide the library's own allocations.  No GPU call is made (device entry points need a HIP device). */
#include <stdio.h>
#include <stdlib.h>
#include "cholamd.h"

static int run_plan(cholamd_plan *p)
{
  const int L = cholamd_plan_levels(p);
  int out4[4], out6[6];
  int64_t vol[6];
  for (int world = 1; world <= 8 && (world == 1 || L > 3); world *= 2)
    for (int r = 0; r < world; r++)
      for (int lvl = 0; lvl < L; lvl++) {
        if (cholamd_plan_level_work_counts(p, lvl, r, world, out4)) return 1;
        if (cholamd_plan_level_work_volume(p, lvl, r, world, 1, vol)) return 1;
      }
  for (int workers = 4; workers <= 256; workers *= 8) (void)cholamd_plan_program_check(p, 1, workers); /* may refuse large problems: also a path to check */
  (void)cholamd_plan_program_check(p, 0, 16);
  (void)cholamd_plan_program_counts(p, 1, out6);
  double *arena = malloc((size_t)cholamd_plan_arena_doubles(p) * sizeof(double));
  if (!arena) return 1;
  const int rc = cholamd_plan_fill_host(p, arena);
  free(arena);
  return rc;
}

int main(int argc, char **argv)
{
  if (argc < 4) { fprintf(stderr, "usage: host_leak matrix separators clusters [more triples]\n"); return 2; }
  for (int a = 1; a + 2 < argc; a += 3) {
    cholamd_plan *p = NULL;
    if (cholamd_plan_create(argv[a], argv[a + 1], argv[a + 2], &p)) { fprintf(stderr, "plan: %s\n", cholamd_last_error()); return 1; }
    if (run_plan(p)) { fprintf(stderr, "run: %s\n", cholamd_last_error()); return 1; }
    cholamd_plan_destroy(p);
    /* error paths free what they allocated */
    if (cholamd_plan_create(argv[a], argv[a + 2], argv[a + 1], &p) == 0) cholamd_plan_destroy(p);
    if (cholamd_plan_create("/nonexistent.mtx", argv[a + 1], argv[a + 2], &p) == 0) cholamd_plan_destroy(p);
  }
  cholamd_problem *g = NULL;
  if (cholamd_generate_laplacian(10, 9, 8, 4, 16, &g)) { fprintf(stderr, "generate: %s\n", cholamd_last_error()); return 1; }
  cholamd_plan *p = NULL;
  if (cholamd_plan_create_from_problem(g, &p)) { fprintf(stderr, "problem plan: %s\n", cholamd_last_error()); return 1; }
  if (run_plan(p)) return 1;
  cholamd_plan_destroy(p);
  cholamd_problem_destroy(g);
  printf("host_leak: ok\n");
  return 0;
}
