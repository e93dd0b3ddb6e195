"""ctypes front end of the CPU oracle (oracle/chol_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by the product package (cholesky_amd/).
"""
import ctypes as C
import glob
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """(Re)build liboracle.so (and oracle/_ref when the reference tree is present)."""
    subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(so):
        build()
    L = C.CDLL(so)
    L.orc_load.restype = C.c_void_p
    L.orc_load.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]
    L.orc_free.argtypes = [C.c_void_p]
    L.orc_factor.restype = C.c_double
    L.orc_factor.argtypes = [C.c_void_p, C.c_int]
    L.orc_factor_parallel.restype = C.c_double
    L.orc_factor_parallel.argtypes = [C.c_void_p, C.c_int]
    L.orc_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_read_vector.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
    L.orc_use_openblas.argtypes = [C.c_char_p]
    L.orc_backend_name.restype = C.c_char_p
    L.orc_banner.restype = C.c_char_p
    L.orc_banner.argtypes = [C.c_void_p]
    L.orc_nnz.restype = C.c_long
    L.orc_nnz.argtypes = [C.c_void_p]
    L.orc_write_matrix.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    for f in ("orc_N", "orc_NZ", "orc_levels", "orc_nsep", "orc_max_int_size", "orc_info", "orc_num_blocks", "orc_num_ops"):
        getattr(L, f).argtypes = [C.c_void_p]
    for f in ("orc_perm", "orc_sep_sizes", "orc_sep_offsets", "orc_tree", "orc_blocks", "orc_ops", "orc_dense"):
        getattr(L, f).argtypes = [C.c_void_p, C.c_void_p]
    L.orc_snapshot_count.argtypes = [C.c_void_p, C.c_int]
    L.orc_snapshot.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.orc_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_level_counts.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    _LIB = L
    return L


def find_openblas():
    """An LP64 OpenBLAS on this box (scipy bundles one); None if absent."""
    try:
        import scipy
        cands = glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas*.so"))
        cands = [c for c in cands if "64_" not in os.path.basename(c)]
        if cands:
            return os.path.abspath(cands[0])
    except Exception:
        pass
    for pat in ("/usr/lib/x86_64-linux-gnu/libopenblas.so*", "/usr/lib64/libopenblas.so*"):
        c = glob.glob(pat)
        if c:
            return c[0]
    return None


class Oracle:
    """One loaded problem: parse + symbolic phase done at construction (mmat.rg:1097-1203)."""

    OPS = ("POTRF", "TRSM", "SYRK", "GEMM")

    def __init__(self, mtx, ord_, clust):
        self.L = lib()
        err = C.c_int(0)
        self.h = self.L.orc_load(os.fsencode(mtx), os.fsencode(ord_), os.fsencode(clust), C.byref(err))
        if not self.h:
            raise RuntimeError(f"oracle load failed, code {err.value}")
        self.N = self.L.orc_N(self.h)
        self.NZ = self.L.orc_NZ(self.h)
        self.levels = self.L.orc_levels(self.h)
        self.nsep = self.L.orc_nsep(self.h)

    def __del__(self):
        try:
            if self.h:
                self.L.orc_free(self.h)
                self.h = None
        except Exception:
            pass

    def _ints(self, fn, n):
        a = np.zeros(n, dtype=np.int32)
        getattr(self.L, fn)(self.h, a.ctypes.data)
        return a

    @property
    def banner(self):
        return self.L.orc_banner(self.h).decode()

    @property
    def perm(self):
        return self._ints("orc_perm", self.N)

    @property
    def sep_sizes(self):
        return self._ints("orc_sep_sizes", self.nsep)

    @property
    def sep_offsets(self):
        return self._ints("orc_sep_offsets", self.nsep)

    @property
    def tree(self):
        return self._ints("orc_tree", self.nsep)

    @property
    def blocks(self):
        n = self.L.orc_num_blocks(self.h)
        return self._ints("orc_blocks", 6 * n).reshape(n, 6)

    def snapshot(self, lbl):
        n = self.L.orc_snapshot_count(self.h, lbl)
        a = np.zeros(7 * n, dtype=np.int32)
        self.L.orc_snapshot(self.h, lbl, a.ctypes.data)
        return a.reshape(n, 7)

    def factor_parallel(self, workers):
        """The level loop with `workers` task-parallel threads (the reference's -ll:cpu N; BLAS one thread per call): the
        same values as factor(), bit for bit.  Returns seconds in the level loop."""
        return self.L.orc_factor_parallel(self.h, int(workers))

    def factor(self, log_ops=False):
        """One reference iteration (re-fill + level loop); returns seconds in the level loop."""
        return self.L.orc_factor(self.h, int(log_ops))

    @property
    def info(self):
        return self.L.orc_info(self.h)

    def counts(self):
        c = np.zeros(4, dtype=np.int64)
        f = np.zeros(4, dtype=np.float64)
        self.L.orc_counts(self.h, c.ctypes.data, f.ctypes.data)
        return c, f

    def level_counts(self, level):
        c = np.zeros(4, dtype=np.int64)
        f = np.zeros(4, dtype=np.float64)
        self.L.orc_level_counts(self.h, level, c.ctypes.data, f.ctypes.data)
        return c, f

    def ops(self):
        n = self.L.orc_num_ops(self.h)
        return self._ints("orc_ops", 14 * n).reshape(n, 14)

    def dense(self):
        """Dense N x N (numpy, row/col = permuted coords) copy of the block storage."""
        a = np.zeros((self.N, self.N), dtype=np.float64, order="F")
        self.L.orc_dense(self.h, a.ctypes.data)
        return a

    def nnz(self):
        return self.L.orc_nnz(self.h)

    def solve(self, b):
        b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1)
        x = np.zeros(self.N, dtype=np.float64)
        self.L.orc_solve(self.h, b.ctypes.data, x.ctypes.data)
        return x

    def write_matrix(self, path, full_precision=False):
        if self.L.orc_write_matrix(self.h, os.fsencode(path), int(full_precision)) != 0:
            raise IOError(path)


def read_vector(path, n):
    out = np.zeros(n, dtype=np.float64)
    rc = lib().orc_read_vector(os.fsencode(path), n, out.ctypes.data)
    if rc != 0:
        raise IOError(f"{path}: rc={rc}")
    return out


def use_openblas(path=None):
    path = path or find_openblas()
    if path is None:
        return False
    return lib().orc_use_openblas(os.fsencode(path)) == 0


def use_own_kernels():
    lib().orc_use_own_kernels()


def backend_name():
    return lib().orc_backend_name().decode()


# ---- stand-alone BLAS restatements (numpy float64, order='F') -------------------------------
def _ld(a):
    return a.strides[1] // 8 if a.ndim == 2 and a.shape[1] > 1 else max(1, a.shape[0])


def blas_potrf(a):
    return lib().orc_blas_potrf(a.shape[0], C.c_void_p(a.ctypes.data), _ld(a))


def blas_trsm(A, B):
    lib().orc_blas_trsm(B.shape[0], B.shape[1], C.c_void_p(A.ctypes.data), _ld(A), C.c_void_p(B.ctypes.data), _ld(B))


def blas_syrk(A, Cm):
    lib().orc_blas_syrk(A.shape[0], A.shape[1], C.c_void_p(A.ctypes.data), _ld(A), C.c_void_p(Cm.ctypes.data), _ld(Cm))


def blas_gemm(A, B, Cm):
    lib().orc_blas_gemm(Cm.shape[0], Cm.shape[1], A.shape[1], C.c_void_p(A.ctypes.data), _ld(A), C.c_void_p(B.ctypes.data), _ld(B),
                        C.c_void_p(Cm.ctypes.data), _ld(Cm))


def blas_trsv(A, x, trans=111):
    lib().orc_blas_trsv(trans, A.shape[0], C.c_void_p(A.ctypes.data), _ld(A), C.c_void_p(x.ctypes.data))


def blas_gemv(A, x, y, trans=111):
    lib().orc_blas_gemv(trans, A.shape[0], A.shape[1], C.c_void_p(A.ctypes.data), _ld(A), C.c_void_p(x.ctypes.data), C.c_void_p(y.ctypes.data))
