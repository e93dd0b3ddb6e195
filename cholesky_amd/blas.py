"""Mirror of the reference's operator interface for the hot path.

Two levels, both thin ctypes calls into libcholamd.so:

* ``LAPACKE_dpotrf / cblas_dtrsm / cblas_dsyrk / cblas_dgemm / cblas_dtrsv / cblas_dgemv`` -- the C
  symbols Terra links (blas.rg:71, 99, 139, 187, 226, 263), numpy (host) arrays in and out,
  column-major ("F") like the Legion instances;
* ``fused_dpotrf / fused_dtrsm / fused_dsyrk / fused_dgemm`` -- the leaf tasks of blas.rg:292-504
  with the same argument meaning: regions (device block instances) + lists of Filled records.
"""
import ctypes as C

import numpy as np

from ._lib import Filled, Region, check, load

ColMajor, NoTrans, Trans, Upper, Lower, NonUnit, Left, Right = 102, 111, 112, 121, 122, 131, 141, 142


def _f(a):
    """float64, unit row stride (column-major, possibly with a leading dimension > rows)."""
    ok = isinstance(a, np.ndarray) and a.dtype == np.float64
    if ok and a.ndim == 2 and a.size:
        ok = (a.shape[0] == 1 or a.strides[0] == 8) and (a.shape[1] == 1 or (a.strides[1] % 8 == 0 and a.strides[1] >= 8 * a.shape[0]))
    elif ok and a.ndim == 1 and a.size > 1:
        ok = a.strides[0] == 8
    if not ok:
        raise TypeError("expected a float64 column-major numpy array (unit row stride)")
    return a


def _status(what):
    rc = load().cholamd_blas_status()
    if rc:
        check(rc, what)


def LAPACKE_dpotrf(a, n=None, lda=None, uplo="L"):
    a = _f(a)
    n = a.shape[0] if n is None else n
    lda = a.strides[1] // 8 if lda is None and a.ndim == 2 and a.shape[1] > 1 else (lda or max(1, a.shape[0]))
    info = load().cholamd_LAPACKE_dpotrf(ColMajor, uplo.encode(), n, a.ctypes.data, lda)
    if info < 0:
        check(info, "LAPACKE_dpotrf")
    return info


def _ld(a):
    return a.strides[1] // 8 if a.ndim == 2 and a.shape[1] > 1 else max(1, a.shape[0])


def cblas_dtrsm(A, B, side=Right, uplo=Lower, trans=Trans, diag=NonUnit, alpha=1.0):
    A, B = _f(A), _f(B)
    m, n = B.shape
    load().cholamd_cblas_dtrsm(ColMajor, side, uplo, trans, diag, m, n, alpha, A.ctypes.data, _ld(A), B.ctypes.data, _ld(B))
    _status("cblas_dtrsm")


def cblas_dgemm(A, B, Cm, transa=NoTrans, transb=Trans, alpha=-1.0, beta=1.0):
    A, B, Cm = _f(A), _f(B), _f(Cm)
    m, n = Cm.shape
    k = A.shape[1]
    load().cholamd_cblas_dgemm(ColMajor, transa, transb, m, n, k, alpha, A.ctypes.data, _ld(A), B.ctypes.data, _ld(B), beta, Cm.ctypes.data, _ld(Cm))
    _status("cblas_dgemm")


def cblas_dsyrk(A, Cm, uplo=Lower, trans=NoTrans, alpha=-1.0, beta=1.0):
    A, Cm = _f(A), _f(Cm)
    n, k = A.shape
    load().cholamd_cblas_dsyrk(ColMajor, uplo, trans, n, k, alpha, A.ctypes.data, _ld(A), beta, Cm.ctypes.data, _ld(Cm))
    _status("cblas_dsyrk")


def cblas_dtrsv(A, x, uplo=Lower, trans=NoTrans, diag=NonUnit):
    A, x = _f(A), _f(x)
    load().cholamd_cblas_dtrsv(ColMajor, uplo, trans, diag, A.shape[0], A.ctypes.data, _ld(A), x.ctypes.data, 1)
    _status("cblas_dtrsv")


def cblas_dgemv(A, x, y, trans=NoTrans, alpha=-1.0, beta=1.0):
    A, x, y = _f(A), _f(x), _f(y)
    m, n = A.shape
    load().cholamd_cblas_dgemv(ColMajor, trans, m, n, alpha, A.ctypes.data, _ld(A), x.ctypes.data, 1, beta, y.ctypes.data, 1)
    _status("cblas_dgemv")


def openblas_set_num_threads(n):
    load().cholamd_openblas_set_num_threads(int(n))


# ---- task level ---------------------------------------------------------------------------------
def region(ptr, ld, lo_x, lo_y, hi_x, hi_y):
    """A dense block instance: every row lo_x..hi_x stored from ptr."""
    return Region(C.c_void_p(ptr), ld, lo_x, lo_y, hi_x, hi_y, None)


def plan_region(plan, arena_ptr, r, c):
    """Block instance (r, c) of an arena laid out by `plan` (row-compacted above the parent block: Region.tile_row)."""
    rg = Region()
    check(load().cholamd_plan_region(plan.h, C.c_void_p(arena_ptr), r, c, C.byref(rg)), "plan_region")
    return rg


def _arr(filled_list):
    n = len(filled_list)
    buf = (Filled * max(n, 1))(*filled_list)
    return buf, n


def _ret(rc, what):
    if rc < 0:
        check(rc, what)
    return rc


def fused_dpotrf(rA, filled_rA, level, interval, debug=False, stream=None):
    fa, na = _arr(filled_rA)
    return _ret(load().cholamd_fused_dpotrf(C.byref(rA), fa, na, level, interval, int(debug), stream), "fused_dpotrf")


def fused_dtrsm(rA, rB, filled_rA, filled_rB, level, interval, debug=False, stream=None):
    fa, na = _arr(filled_rA)
    fb, nb = _arr(filled_rB)
    return _ret(load().cholamd_fused_dtrsm(C.byref(rA), C.byref(rB), fa, na, fb, nb, level, interval, int(debug), stream), "fused_dtrsm")


def fused_dsyrk(rA, rB, rC, filled_rA, filled_rB, filled_rC, col_cluster_size, level, interval, debug=False, stream=None):
    fa, na = _arr(filled_rA)
    fb, nb = _arr(filled_rB)
    fc, nc = _arr(filled_rC)
    return _ret(load().cholamd_fused_dsyrk(C.byref(rA), C.byref(rB), C.byref(rC), fa, na, fb, nb, fc, nc, col_cluster_size, level, interval,
                                           int(debug), stream), "fused_dsyrk")


def fused_dgemm(rA, rB, rC, filled_rA, filled_rB, filled_rC, col_cluster_size, level, interval, debug=False, stream=None):
    fa, na = _arr(filled_rA)
    fb, nb = _arr(filled_rB)
    fc, nc = _arr(filled_rC)
    return _ret(load().cholamd_fused_dgemm(C.byref(rA), C.byref(rB), C.byref(rC), fa, na, fb, nb, fc, nc, col_cluster_size, level, interval,
                                           int(debug), stream), "fused_dgemm")
