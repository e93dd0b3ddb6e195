"""GPU parity of the BLAS-level C-ABI (L-A of include/cholamd.h) against the oracle's restatement of
the six BLAS calls the reference issues (blas.rg:71, 99, 139, 187, 226, 263).

fp64 tolerance: |got - want| <= 1e-12 * scale, scale = magnitude of the accumulated products
(k * max|a| * max|b|); results are not bit-identical because the MFMA accumulation order differs
from a scalar loop."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as orc  # noqa: E402


def F(a):
    return np.asfortranarray(a, dtype=np.float64)


def padded(rng, m, n, ld):
    """m x n view with leading dimension ld inside a bigger F-ordered buffer (poisoned padding)."""
    buf = np.full((ld, max(n, 1)), 777.0, order="F")
    buf[:m, :n] = rng.standard_normal((m, n))
    return buf, buf[:m, :n]


@pytest.fixture(scope="module")
def blas():
    import cholesky_amd
    orc.use_own_kernels()
    return cholesky_amd.blas


def test_mfma_layout_identity_asymmetric(blas):
    """A = I, asymmetric B: a row/column swap in the fragment maps cannot hide (guide section 3)."""
    n = 16
    A = F(np.eye(n))
    B = F(np.arange(n * n, dtype=np.float64).reshape(n, n) + 0.25 * np.arange(n)[:, None])
    C = F(np.zeros((n, n)))
    blas.cblas_dgemm(A, B, C)  # C -= A B^T = -B^T
    assert np.array_equal(C, -B.T)


SHAPES = [(1, 1, 1), (3, 2, 5), (16, 16, 4), (17, 15, 14), (33, 31, 35), (97, 97, 259), (6, 11, 84), (1, 97, 14), (64, 48, 3)]


@pytest.mark.parametrize("m,n,k", SHAPES)
def test_dgemm(blas, m, n, k):
    rng = np.random.default_rng(m * 1000 + n * 10 + k)
    _, A = padded(rng, m, k, m + 3)
    _, B = padded(rng, n, k, n + 5)
    cb, C = padded(rng, m, n, m + 2)
    want = C.copy(order="F")
    orc.blas_gemm(F(A), F(B), want)
    blas.cblas_dgemm(A, B, C)
    assert np.abs(C - want).max() <= 1e-12 * max(1.0, k)
    assert (cb[m:, :] == 777.0).all()  # padding rows untouched


@pytest.mark.parametrize("n,k", [(1, 1), (5, 3), (16, 16), (17, 33), (97, 259), (40, 7)])
def test_dsyrk_lower_only(blas, n, k):
    rng = np.random.default_rng(n * 100 + k)
    _, A = padded(rng, n, k, n + 1)
    _, C = padded(rng, n, n, n + 4)
    C0 = C.copy()
    want = C.copy(order="F")
    orc.blas_syrk(F(A), want)
    blas.cblas_dsyrk(A, C)
    assert np.abs(np.tril(C) - np.tril(want)).max() <= 1e-12 * max(1.0, k)
    assert np.array_equal(np.triu(C, 1), np.triu(C0, 1))  # strict upper triangle is not referenced


@pytest.mark.parametrize("m,n", [(1, 1), (4, 3), (10, 14), (32, 32), (33, 33), (97, 259), (7, 100), (70, 65)])
def test_dtrsm(blas, m, n):
    rng = np.random.default_rng(m * 100 + n)
    Lfull = np.tril(rng.standard_normal((n, n))) + np.diag(2.0 + rng.random(n) * n ** 0.5)
    lb, Lm = padded(rng, n, n, n + 2)
    Lm[:, :] = Lfull + np.triu(np.full((n, n), 55.0), 1)  # garbage above the diagonal must be ignored
    _, B = padded(rng, m, n, m + 1)
    want = B.copy(order="F")
    orc.blas_trsm(F(np.tril(Lm)), want)
    blas.cblas_dtrsm(Lm, B)
    scale = np.abs(want).max() + 1.0
    assert np.abs(B - want).max() <= 1e-11 * scale


@pytest.mark.parametrize("n", [1, 2, 14, 31, 32, 33, 64, 97, 225, 259, 300])
def test_dpotrf(blas, n):
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, n))
    S = G @ G.T + n * np.eye(n)
    ab, A = padded(rng, n, n, n + 3)
    A[:, :] = np.tril(S) + np.triu(np.full((n, n), -9.0), 1)  # upper part must not be read or written
    want = F(np.tril(S))
    assert orc.blas_potrf(want) == 0
    info = blas.LAPACKE_dpotrf(A)
    assert info == 0
    assert np.abs(np.tril(A) - np.tril(want)).max() <= 1e-12 * np.abs(want).max()
    assert (np.triu(A, 1) == np.triu(np.full((n, n), -9.0), 1)).all()
    assert (ab[n:, :] == 777.0).all()


def test_dpotrf_not_positive_definite_reports_info(blas):
    """LAPACK info semantics (the reference discards it, blas.rg:71; the build returns it)."""
    n = 40
    rng = np.random.default_rng(5)
    G = rng.standard_normal((n, n))
    S = G @ G.T + n * np.eye(n)
    S[17, 17] = -1.0
    A = F(np.tril(S))
    want = F(np.tril(S))
    assert orc.blas_potrf(want) == 18
    assert blas.LAPACKE_dpotrf(A) == 18


@pytest.mark.parametrize("n", [1, 5, 32, 33, 100, 259])
@pytest.mark.parametrize("trans", [111, 112])
def test_dtrsv(blas, n, trans):
    rng = np.random.default_rng(n + trans)
    Lm = F(np.tril(rng.standard_normal((n, n))) + np.diag(3.0 + rng.random(n) * n ** 0.5))
    x = rng.standard_normal(n)
    want = x.copy()
    orc.blas_trsv(Lm, want, trans)
    blas.cblas_dtrsv(Lm, x, trans=trans)
    assert np.abs(x - want).max() <= 1e-11 * (np.abs(want).max() + 1.0)


@pytest.mark.parametrize("m,n", [(1, 1), (3, 7), (225, 259), (300, 20), (20, 300)])
@pytest.mark.parametrize("trans", [111, 112])
def test_dgemv(blas, m, n, trans):
    rng = np.random.default_rng(m * 7 + n + trans)
    A = F(rng.standard_normal((m, n)))
    x = rng.standard_normal(n if trans == 111 else m)
    y = rng.standard_normal(m if trans == 111 else n)
    want = y.copy()
    orc.blas_gemv(A, x, want, trans)
    blas.cblas_dgemv(A, x, y, trans=trans)
    assert np.abs(y - want).max() <= 1e-12 * max(m, n)


def test_unsupported_parameters_fail_loudly(blas):
    import cholesky_amd
    A = F(np.eye(4))
    B = F(np.ones((4, 4)))
    with pytest.raises(cholesky_amd.CholamdError):
        blas.cblas_dtrsm(A, B, side=blas.Left)
    with pytest.raises(cholesky_amd.CholamdError):
        blas.cblas_dgemm(A, B, B.copy(order="F"), alpha=1.0)
    with pytest.raises(cholesky_amd.CholamdError):
        blas.LAPACKE_dpotrf(A, uplo="U")


def test_empty_inputs(blas):
    """m == 0 POTRF is skipped (blas.rg:68); empty GEMM/TRSM are no-ops."""
    assert blas.LAPACKE_dpotrf(F(np.zeros((0, 0))), n=0, lda=1) == 0
    C = F(np.ones((3, 3)))
    blas.cblas_dgemm(F(np.zeros((3, 0))), F(np.zeros((3, 0))), C)
    assert (C == 1.0).all()
