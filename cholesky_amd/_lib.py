"""ctypes binding of libcholamd.so (include/cholamd.h).  No compute happens in Python.

The shared library is built in-tree by `make` (or `__graft_entry__.build()`); importing this module
without it raises -- there is no Python/CPU fallback for the numeric path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CHOLAMD_LIB") or os.path.join(_HERE, "lib", "libcholamd.so")  # CHOLAMD_LIB: the sanitizer build of `make asan`
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "cholamd.h")


class Filled(C.Structure):
    """fspace Filled (blas.rg:55-61)."""
    _fields_ = [("filled", C.c_int), ("sep_x", C.c_int), ("sep_y", C.c_int), ("interval", C.c_int), ("cluster", C.c_int),
                ("lo_x", C.c_int), ("lo_y", C.c_int), ("hi_x", C.c_int), ("hi_y", C.c_int)]


class Op(C.Structure):
    _fields_ = [("op", C.c_int), ("level", C.c_int), ("m", C.c_int), ("n", C.c_int), ("k", C.c_int),
                ("a_sx", C.c_int), ("a_sy", C.c_int), ("a_z", C.c_int), ("b_sx", C.c_int), ("b_sy", C.c_int), ("b_z", C.c_int),
                ("c_sx", C.c_int), ("c_sy", C.c_int), ("c_z", C.c_int)]


class Region(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("ld", C.c_int), ("lo_x", C.c_int), ("lo_y", C.c_int), ("hi_x", C.c_int), ("hi_y", C.c_int),
                ("tile_row", C.POINTER(C.c_int))]


class SepInfo(C.Structure):
    _fields_ = [("levels", C.c_int), ("num_separators", C.c_int)]


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `make` (or __graft_entry__.build()) first; "
                          "cholesky_amd has no fallback path without its HIP library")
    # One HIP runtime per process: torch bundles its own libamdhip64.so.7; when it is loaded first
    # the dynamic linker resolves libcholamd's dependency to that same copy by soname.  Loading
    # /opt/rocm's copy first and torch's second leaves torch without a visible GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, ci, cd, i64 = C.c_void_p, C.c_int, C.c_double, C.c_int64
    L.cholamd_last_error.restype = C.c_char_p
    L.cholamd_version.restype = C.c_char_p
    L.cholamd_read_separators.argtypes = [C.c_char_p, ci, vp, vp, C.POINTER(SepInfo)]
    L.cholamd_read_clusters.argtypes = [C.c_char_p, vp, vp, vp, i64, C.POINTER(i64)]
    L.cholamd_read_matrix.argtypes = [C.c_char_p, ci, vp, vp, vp]
    L.cholamd_read_vector.argtypes = [C.c_char_p, ci, vp]
    L.cholamd_write_solution.argtypes = [C.c_char_p, vp, ci, ci]
    L.cholamd_plan_create.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(vp)]
    L.cholamd_plan_create_from_arrays.argtypes = [ci, ci, vp, vp, vp, vp, vp, i64, i64, vp, vp, vp, C.c_char_p, C.POINTER(vp)]
    L.cholamd_plan_destroy.argtypes = [vp]
    L.cholamd_generate_laplacian.argtypes = [ci, ci, ci, ci, ci, C.POINTER(vp)]
    L.cholamd_problem_destroy.argtypes = [vp]
    L.cholamd_problem_n.argtypes = [vp]
    L.cholamd_problem_nz.argtypes = [vp]
    L.cholamd_problem_write.argtypes = [vp, C.c_char_p]
    L.cholamd_plan_create_from_problem.argtypes = [vp, C.POINTER(vp)]
    L.cholamd_problem_rhs.argtypes = [vp, vp]
    for f in ("n", "nz", "levels", "num_separators", "max_int_size", "num_blocks"):
        getattr(L, "cholamd_plan_" + f).argtypes = [vp]
    for f in ("arena_doubles", "dropped_entries", "num_ops", "nnz_a", "nnz_l", "alg_bytes"):
        fn = getattr(L, "cholamd_plan_" + f)
        fn.argtypes = [vp]
        fn.restype = i64
    L.cholamd_plan_nnz_tiles.argtypes = [vp]
    L.cholamd_plan_nnz_tiles.restype = i64
    L.cholamd_plan_fmin.argtypes = [vp]
    L.cholamd_plan_fmin.restype = cd
    L.cholamd_plan_banner.argtypes = [vp]
    L.cholamd_plan_banner.restype = C.c_char_p
    for f in ("perm", "sep_sizes", "sep_offsets", "tree", "blocks", "ops"):
        getattr(L, "cholamd_plan_" + f).argtypes = [vp, vp]
    L.cholamd_plan_snapshot_count.argtypes = [vp, ci]
    L.cholamd_plan_snapshot_count.restype = i64
    L.cholamd_plan_snapshot.argtypes = [vp, ci, vp]
    L.cholamd_plan_counts.argtypes = [vp, ci, vp, vp]
    L.cholamd_plan_flops.argtypes = [vp]
    L.cholamd_plan_flops.restype = cd
    L.cholamd_plan_fill_host.argtypes = [vp, vp]
    L.cholamd_plan_arena_to_dense.argtypes = [vp, vp, vp]
    L.cholamd_plan_fill_host_part.argtypes = [vp, vp, ci, ci, C.POINTER(i64)]
    L.cholamd_plan_level_work_counts.argtypes = [vp, ci, ci, ci, vp]
    L.cholamd_plan_level_work_volume.argtypes = [vp, ci, ci, ci, ci, vp]
    L.cholamd_plan_program_check.argtypes = [vp, ci, ci]
    L.cholamd_plan_program_check_opts.argtypes = [vp, ci, ci, ci, ci]
    L.cholamd_plan_program_counts.argtypes = [vp, ci, vp]
    L.cholamd_plan_program_jobs.argtypes = [vp, ci, i64, vp]
    L.cholamd_plan_program_jobs.restype = i64
    L.cholamd_plan_write_matrix.argtypes = [vp, vp, C.c_char_p, ci]
    L.cholamd_plan_write_debug_log.argtypes = [vp, vp]
    L.cholamd_device_create.argtypes = [vp, ci, C.POINTER(vp)]
    L.cholamd_device_destroy.argtypes = [vp]
    L.cholamd_device_set_partition.argtypes = [vp, ci, ci]
    L.cholamd_device_set_option.argtypes = [vp, C.c_char_p, ci]
    L.cholamd_comm_unique_id.argtypes = [vp]
    L.cholamd_comm_create.argtypes = [vp, ci, ci, vp, C.POINTER(vp)]
    L.cholamd_comm_create_local.argtypes = [vp, ci, vp]
    L.cholamd_factor_multi.argtypes = [vp, vp, vp, ci, vp]
    L.cholamd_gather_factor.argtypes = [vp, vp, ci, vp]
    L.cholamd_comm_adopt.argtypes = [vp, ci, ci, C.POINTER(vp)]
    L.cholamd_comm_destroy.argtypes = [vp]
    L.cholamd_comm_destroy.restype = None
    L.cholamd_comm_allreduce.argtypes = [vp, vp, i64, vp]
    L.cholamd_device_tail_offset.argtypes = [vp]
    L.cholamd_device_tail_offset.restype = i64
    L.cholamd_exchange_tail.argtypes = [vp, vp, vp, vp]
    L.cholamd_factor_sharded.argtypes = [vp, vp, vp, vp]
    L.cholamd_factor_sharded_f32.argtypes = [vp, vp, vp, vp]
    L.cholamd_gather_to_root.argtypes = [vp, vp, ci, vp, vp]
    L.cholamd_exchange_volume.argtypes = [vp, vp]
    L.cholamd_plan_exchange_volume.argtypes = [vp, ci, ci, ci, vp]
    L.cholamd_plan_exchange_pieces.argtypes = [vp, ci, ci, ci, vp]
    L.cholamd_plan_solve_counts.argtypes = [vp, ci, ci, ci, vp]
    L.cholamd_plan_solve_skips.argtypes = [vp, ci, vp, vp]
    L.cholamd_solve_sharded.argtypes = [vp, vp, vp, vp, vp, vp]
    L.cholamd_solve_sharded_f32.argtypes = [vp, vp, vp, vp, vp, vp]
    L.cholamd_solve_refine_sharded.argtypes = [vp, vp, vp, vp, ci, C.c_double, vp, vp, vp, vp]
    L.cholamd_solve_multi.argtypes = [vp, vp, vp, vp, vp, ci, vp]
    L.cholamd_solve_refine_multi.argtypes = [vp, vp, vp, vp, ci, C.c_double, vp, vp, vp, ci, vp]
    L.cholamd_follow_rounds.argtypes = [ci, ci, ci, vp, vp]
    L.cholamd_plan_program_followers.argtypes = [vp, i64, vp]
    L.cholamd_plan_program_followers.restype = i64
    L.cholamd_factor_multi_f32.argtypes = [vp, vp, vp, ci, vp]
    L.cholamd_gather_factor_f32.argtypes = [vp, vp, ci, vp]
    L.cholamd_device_alloc.argtypes = [vp, i64, C.POINTER(vp)]
    L.cholamd_device_free.argtypes = [vp, vp]
    L.cholamd_device_upload.argtypes = [vp, vp, vp, i64, vp]
    L.cholamd_device_download.argtypes = [vp, vp, vp, i64, vp]
    L.cholamd_device_sync.argtypes = [vp, vp]
    L.cholamd_device_fill.argtypes = [vp, vp, vp]
    L.cholamd_factor.argtypes = [vp, vp, vp]
    L.cholamd_factor_levels.argtypes = [vp, vp, ci, ci, vp]
    L.cholamd_device_program_trace.argtypes = [vp, vp, vp, i64, vp, C.POINTER(ci)]
    L.cholamd_factor_info.argtypes = [vp, C.POINTER(ci)]
    L.cholamd_solve.argtypes = [vp, vp, vp, vp, vp]
    L.cholamd_device_fill_f32.argtypes = [vp, vp, vp]
    L.cholamd_factor_f32.argtypes = [vp, vp, vp]
    L.cholamd_factor_levels_f32.argtypes = [vp, vp, ci, ci, vp]
    L.cholamd_solve_f32.argtypes = [vp, vp, vp, vp, vp]
    L.cholamd_solve_refine.argtypes = [vp, vp, vp, vp, ci, cd, C.POINTER(ci), C.POINTER(cd), vp]
    L.cholamd_residual.argtypes = [vp, vp, vp, vp, C.POINTER(cd), vp]
    L.cholamd_device_alloc_arena.argtypes = [vp, ci, C.POINTER(vp), C.POINTER(C.c_int64)]
    L.cholamd_device_free_arena.argtypes = [vp, vp]
    L.cholamd_device_set_timing.argtypes = [vp, ci]
    L.cholamd_device_get_timing.argtypes = [vp, vp, vp]
    L.cholamd_device_get_timing_ex.argtypes = [vp, vp, vp]
    L.cholamd_device_bcast_phases.argtypes = [vp]
    L.cholamd_comm_count.argtypes = [vp, C.POINTER(ci)]
    L.cholamd_device_event_overhead.argtypes = [vp, vp, C.POINTER(C.c_float)]
    RP, FP = C.POINTER(Region), C.POINTER(Filled)
    L.cholamd_plan_region.argtypes = [vp, vp, ci, ci, RP]
    L.cholamd_plan_level_work_volume_opts.argtypes = [vp, ci, ci, ci, vp]
    L.cholamd_plan_level_mt_fill.argtypes = [vp, ci, vp]
    L.cholamd_plan_arena_dense_doubles.argtypes = [vp]
    L.cholamd_plan_arena_dense_doubles.restype = C.c_int64
    L.cholamd_plan_block_tile_map.argtypes = [vp, ci, ci, vp]
    L.cholamd_fused_dpotrf.argtypes = [RP, FP, ci, ci, ci, ci, vp]
    L.cholamd_fused_dtrsm.argtypes = [RP, RP, FP, ci, FP, ci, ci, ci, ci, vp]
    L.cholamd_fused_dsyrk.argtypes = [RP, RP, RP, FP, ci, FP, ci, FP, ci, ci, ci, ci, ci, vp]
    L.cholamd_fused_dgemm.argtypes = [RP, RP, RP, FP, ci, FP, ci, FP, ci, ci, ci, ci, ci, vp]
    L.cholamd_LAPACKE_dpotrf.argtypes = [ci, C.c_char, ci, vp, ci]
    L.cholamd_cblas_dtrsm.argtypes = [ci, ci, ci, ci, ci, ci, ci, cd, vp, ci, vp, ci]
    L.cholamd_cblas_dtrsm.restype = None
    L.cholamd_cblas_dgemm.argtypes = [ci, ci, ci, ci, ci, ci, cd, vp, ci, vp, ci, cd, vp, ci]
    L.cholamd_cblas_dgemm.restype = None
    L.cholamd_cblas_dsyrk.argtypes = [ci, ci, ci, ci, ci, cd, vp, ci, cd, vp, ci]
    L.cholamd_cblas_dsyrk.restype = None
    L.cholamd_cblas_dtrsv.argtypes = [ci, ci, ci, ci, ci, vp, ci, vp, ci]
    L.cholamd_cblas_dtrsv.restype = None
    L.cholamd_cblas_dgemv.argtypes = [ci, ci, ci, ci, cd, vp, ci, vp, ci, cd, vp, ci]
    L.cholamd_cblas_dgemv.restype = None
    L.cholamd_openblas_set_num_threads.argtypes = [ci]
    L.cholamd_openblas_set_num_threads.restype = None
    L.cholamd_dpotrf_dev.argtypes = [ci, vp, ci, vp, vp]
    L.cholamd_dtrsm_dev.argtypes = [ci, ci, vp, ci, vp, ci, vp]
    L.cholamd_dgemm_dev.argtypes = [ci, ci, ci, vp, ci, vp, ci, vp, ci, vp]
    L.cholamd_dsyrk_dev.argtypes = [ci, ci, vp, ci, vp, ci, vp]
    L.cholamd_dtrsv_dev.argtypes = [ci, ci, vp, ci, vp, vp]
    L.cholamd_dgemv_dev.argtypes = [ci, ci, ci, vp, ci, vp, vp, vp]
    _lib = L
    return L


class CholamdError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        msg = load().cholamd_last_error().decode(errors="replace")
        raise CholamdError(f"{what} failed with code {rc}: {msg}")
