"""Pin the CPU oracle (oracle/chol_oracle.c) to the reference.

Golden vectors come from the reference's own verify.py (tests/golden/make_golden.py): the permuted
matrix (verify.permute_matrix), L = scipy cholesky of it (verify.check_matrix) and x (verify.check_solution).
Known-answer integers come from SURVEY.md Appendix C.
"""
import os

import numpy as np
import pytest

from conftest import CASES, KNOWN, case_paths
from oracle import oracle as orc


@pytest.fixture(scope="module")
def oracles():
    orc.build()
    orc.use_own_kernels()
    out = {}
    for case in CASES:
        m, o, c, _ = case_paths(case)
        out[case] = orc.Oracle(m, o, c)
    return out


@pytest.mark.parametrize("case", list(CASES))
def test_permuted_matrix_matches_reference(case, oracles, golden):
    O = oracles[case]
    O.L.orc_factor  # noqa: B018  (symbol must exist)
    g = golden(case)
    # before any factorisation the block storage holds P A P^T (lower, ancestor/descendant blocks)
    m, o, c, _ = case_paths(case)
    fresh = orc.Oracle(m, o, c)
    assert np.array_equal(fresh.dense(), g["pmat"])
    assert fresh.nnz() == KNOWN[case][2]
    assert sorted(fresh.perm.tolist()) == list(range(fresh.N))


@pytest.mark.parametrize("case", list(CASES))
def test_factor_matches_reference_golden(case, oracles, golden):
    O = oracles[case]
    O.factor(log_ops=True)
    assert O.info == 0
    g = golden(case)
    L = O.dense()
    assert np.abs(L - g["L"]).max() <= 1e-13
    calls, flops = O.counts()
    assert tuple(int(v) for v in calls) == KNOWN[case][0]
    assert flops.sum() == pytest.approx(KNOWN[case][1], rel=1e-5)
    assert O.nnz() == KNOWN[case][3]
    # residual ||L L^T - P A P^T||_F / ||A||_F
    A = g["pmat"] + np.tril(g["pmat"], -1).T
    Lt = np.tril(L)
    assert np.linalg.norm(Lt @ Lt.T - A) / np.linalg.norm(A) <= 1e-14


@pytest.mark.parametrize("case", list(CASES))
def test_solve_matches_reference_golden(case, oracles, golden):
    O = oracles[case]
    O.factor()
    g = golden(case)
    b = orc.read_vector(case_paths(case)[3], O.N)
    assert np.array_equal(b, g["b"])
    x = O.solve(b)
    assert np.abs(x - g["x"]).max() <= 1e-11


def test_per_level_counts_3375(oracles):
    """SURVEY Appendix C, per-level call counts of lapl_3375 (level 4 = leaves ... level 0 = root)."""
    O = oracles["lapl_3375x3375"]
    O.factor()
    want = {4: (16, 193, 193, 1229), 3: (8, 162, 162, 1610), 2: (4, 56, 56, 371), 1: (2, 14, 14, 42), 0: (1, 0, 0, 0)}
    for lvl, w in want.items():
        c, _ = O.level_counts(lvl)
        assert tuple(int(v) for v in c) == w


def test_separator_sizes_3375(oracles):
    O = oracles["lapl_3375x3375"]
    sizes = O.sep_sizes
    tree = O.tree  # heap order -> label
    by_level = [[int(sizes[tree[i - 1] - 1]) for i in range(1 << l, 1 << (l + 1))] for l in range(O.levels)]
    assert by_level[0] == [225]
    assert by_level[1] == [98, 97]
    assert by_level[2] == [51, 34, 59, 34]
    assert by_level[3] == [26, 39, 33, 14, 26, 35, 30, 20]
    assert by_level[4] == [86, 174, 259, 206, 161, 177, 115, 102, 163, 143, 236, 201, 218, 115, 97, 101]
    assert O.L.orc_num_blocks(O.h) == 129


def test_openblas_backend_agrees(oracles, golden):
    """The OpenBLAS back end (what the reference links) gives the same factor to rounding."""
    if not orc.use_openblas():
        pytest.skip("no OpenBLAS on this box")
    try:
        O = oracles["lapl_400x400"]
        O.factor()
        assert np.abs(O.dense() - golden("lapl_400x400")["L"]).max() <= 1e-13
    finally:
        orc.use_own_kernels()


def test_factor_writer_roundtrip(tmp_path, oracles, golden):
    """write_matrix format (mmat.rg:102-147) is readable by scipy.io.mmread as check_matrix does."""
    import scipy.io

    O = oracles["lapl_25x25"]
    O.factor()
    p = tmp_path / "factored.mtx"
    O.write_matrix(str(p))
    M = np.tril(scipy.io.mmread(str(p)).toarray())
    assert np.allclose(M, golden("lapl_25x25")["L"], rtol=1e-4, atol=1e-4)  # the reference's own gate (verify.py:286)
    O.write_matrix(str(p), full_precision=True)
    M = np.tril(scipy.io.mmread(str(p)).toarray())
    assert np.abs(M - golden("lapl_25x25")["L"]).max() <= 1e-14


def test_mmio_reference_build_agrees():
    """oracle/_ref/libmmio_ref.so is the reference's own mmio.c compiled where it lies; the oracle's
    banner/size reader must agree with it on every fixture."""
    import ctypes as C

    so = os.path.join(os.path.dirname(orc.__file__), "_ref", "libmmio_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built (reference tree absent)")
    ref = C.CDLL(so)
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    ref.mm_read_banner.argtypes = [C.c_void_p, C.c_char_p]
    ref.mm_read_mtx_crd_size.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    for case in CASES:
        m, o, c, _ = case_paths(case)
        fp = libc.fopen(m.encode(), b"r")
        tc = C.create_string_buffer(4)
        assert ref.mm_read_banner(fp, tc) == 0
        M, N, NZ = C.c_int(), C.c_int(), C.c_int()
        assert ref.mm_read_mtx_crd_size(fp, C.byref(M), C.byref(N), C.byref(NZ)) == 0
        libc.fclose(fp)
        O = orc.Oracle(m, o, c)
        assert (O.N, O.NZ) == (N.value, NZ.value) and M.value == N.value
        assert tc.raw == b"MCRH"  # matrix coordinate real hermitian


def test_level_parallel_oracle_is_bit_identical():
    """The cpu_baseline's task-parallel variants (3 workers as the reference's tests run, all cores) compute the same factor, bit for bit."""
    m, o, c, _ = case_paths("lapl_400x400")
    O = orc.Oracle(m, o, c)
    O.factor()
    ref = O.dense().copy()
    for workers in (1, 3, 8):
        O.factor_parallel(workers)
        assert np.array_equal(O.dense(), ref)
