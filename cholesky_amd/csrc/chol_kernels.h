/* Launchers of the HIP kernels in chol_kernels.hip (internal). */
#ifndef CHOL_KERNELS_H
#define CHOL_KERNELS_H
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include "chol_plan.h"
#ifdef __cplusplus
extern "C" {
#endif
int chol_launch_scatter(double *arena, const int64_t *dst, const double *val, int64_t nnz, hipStream_t st);
int chol_launch_potrf(double *base, double *ws, const chol_potrf_desc *descs, int n, int *info, hipStream_t st);
int chol_launch_potrf_big(double *base, double *ws, const chol_potrf_desc *descs, int n, int *info, hipStream_t st);
int chol_launch_potrf_trsm(double *base, double *ws, const chol_potrf_desc *pdescs, int n_potrf, const chol_trsm_desc *tdescs, int n_trsm,
                           const chol_upd_task *tasks, const chol_upd_src *srcs, int n_task,
                           int *info, int *progress, int progress_base, int *done, int done_target, hipStream_t st);
int chol_launch_program(double *base, double *ws, const chol_job *jobs, int njobs, const chol_wait *waits, const chol_potrf_desc *pdescs, const chol_trsm_desc *tdescs,
                        const chol_upd_task *tasks, const chol_upd_src *srcs, const chol_ext *exts, int *ctr, const int *ctr_total, int epoch, int *head, int head_base,
                        int grid, int *info, int *info_next, unsigned long long *trace, hipStream_t st);
int chol_launch_trsm_w(double *base, const double *ws, const chol_trsm_desc *descs, int n, hipStream_t st);
int chol_launch_trsm_wt(double *base, const double *ws, const chol_trsm_desc *descs, int n, hipStream_t st);
int chol_launch_trsm_big(double *base, const double *ws, const chol_trsm_desc *descs, int n, hipStream_t st);
int chol_launch_dinv(const double *L, int n, int ldl, double *W, hipStream_t st);
int chol_launch_trsm(double *base, const double *ws, const chol_trsm_desc *descs, int n, hipStream_t st);
int chol_launch_update(double *base, const chol_upd_task *tasks, const chol_upd_src *srcs, int ntask, hipStream_t st);
int chol_launch_update_mt(double *base, const chol_upd_task *tasks, const chol_upd_src *srcs, int ntask, int64_t arena_elems, hipStream_t st);
int chol_launch_permute(const double *in, const int *perm, double *out, int n, int inverse, hipStream_t st);
int chol_launch_solve_dinv(const double *base, const chol_trsv_desc *descs, int n, int max_n, double *W, hipStream_t st);
int chol_launch_solve_trsv(const double *base, const chol_trsv_desc *descs, int n, int max_n, int max_under, const double *W, double *y, int backward, int *flags, int *gen, const double *W256, double *xt, hipStream_t st);
/* explicit inverses of the 256-column diagonal spans of a level's separators (levels of at most 8 separators wider than a span): W256[(separator * spans + span) * 65536
 * + column * 256 + row], from the 16x16 inverses W16 of chol_launch_solve_dinv; passed to chol_launch_solve_trsv (with xt: 8 x 256 doubles of scratch) they turn
 * the span solve of the step launches into a matrix-vector product over sixteen workgroups (k_solve_stepw); NULL: k_solve_step */
#define CHOL_STEPW_MAX_SEPS 128 /* most separators of a level that takes the step launches with explicit span inverses: flags = 16 ints, xt = 256 doubles per separator */
int chol_launch_solve_inv256(const double *base, const chol_trsv_desc *descs, int n, int max_n, const double *W16, double *W256, hipStream_t st);
int chol32_launch_solve_inv256(const float *base, const chol_trsv_desc *descs, int n, int max_n, const double *W16, double *W256, hipStream_t st);
int chol_launch_solve_offdiag(const double *base, const chol_gemv_desc *blocks, const int *items, int n_items, double *y, int backward, hipStream_t st);
int chol_launch_trsv_fwd(const double *base, const chol_trsv_desc *descs, int n, double *y, hipStream_t st);
int chol_launch_gemv_fwd(const double *base, const chol_gemv_desc *descs, const int *grp_start, const int *grp_rows, int ngroups, double *y, hipStream_t st);
int chol_launch_bwd(const double *base, const chol_trsv_desc *descs, const chol_gemv_desc *gd, const int *gstart, int n, double *y, hipStream_t st);
/* fp32 factor (chol_kernels_f32.hip) and the fp64 refinement helpers */
int chol32_launch_scatter(float *arena, const int64_t *dst, const double *val, int64_t nnz, hipStream_t st);
int chol32_launch_potrf(float *base, float *ws, const chol_potrf_desc *descs, int n, int *info, hipStream_t st);
int chol32_launch_trsm(float *base, const float *ws, const chol_trsm_desc *descs, int n, hipStream_t st);
int chol32_launch_trsm_wt(float *base, const float *ws, const chol_trsm_desc *descs, int n, hipStream_t st);
int chol32_launch_update(float *base, const chol_upd_task *tasks, const chol_upd_src *srcs, int ntask, hipStream_t st);
int chol32_launch_update_mt(float *base, const chol_upd_task *tasks, const chol_upd_src *srcs, int ntask, int64_t arena_elems, hipStream_t st);
int chol32_launch_solve_dinv(const float *base, const chol_trsv_desc *descs, int n, int max_n, double *W, hipStream_t st);
/* flags / gen: STEP flags of the device object (one int per separator of a top level, zero at allocation) and its launch counter (host) -- the step launches
 * of the wide top separators (k_solve_step); NULL: launch by launch */
int chol32_launch_solve_trsv(const float *base, const chol_trsv_desc *descs, int n, int max_n, int max_under, const double *W, double *y, int backward, int *flags, int *gen, const double *W256, double *xt, hipStream_t st);
int chol32_launch_solve_offdiag(const float *base, const chol_gemv_desc *blocks, const int *items, int n_items, double *y, int backward, hipStream_t st);
int chol_launch_residual(const int64_t *ptr, const int *col, const double *val, const double *b, const double *x, double *r, int n, double *partial, hipStream_t st);
int chol_launch_axpy1(double *x, const double *dx, int n, hipStream_t st);
/* diagnostic instance of the program launch (k_program<true>): 4 stamps per job, then CHOL_TRACE_X per job -- [0] follower: own tiles' wait over,
 * [1] its items, [2 + i] round of item i begun; [48 + k] POTRF job: column k published / TRSM job (first strip): column tile k on its channel;
 * [72 + k] POTRF job: the factor wave starts column k / TRSM job: the POTRF's column k seen */
#define CHOL_TRACE_X 96
#ifdef __cplusplus
}
#endif
#endif
