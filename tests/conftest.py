import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# the reference's four fixtures (test_matrices.py:49-142); data files copied verbatim
CASES = {
    "lapl_9x9": ("lapl_3_2.mtx", "lapl_3_2_ord_2.txt", "lapl_3_2_clust_2.txt", "B_9x1.mtx"),
    "lapl_25x25": ("lapl_5_2.mtx", "lapl_5_2_ord_3.txt", "lapl_5_2_clust_3.txt", "B_25x1.mtx"),
    "lapl_400x400": ("lapl_20_2.mtx", "lapl_20_2_ord_5.txt", "lapl_20_2_clust_5.txt", "B_400x1.mtx"),
    "lapl_3375x3375": ("lapl_15_3.mtx", "lapl_15_3_ord_5.txt", "lapl_15_3_clust_5.txt", "B_3375x1.mtx"),
}

# SURVEY Appendix C known answers: calls POTRF/TRSM/SYRK/GEMM, F_ref, nnz(tril A), nnz(L)
KNOWN = {
    "lapl_9x9": ((3, 2, 2, 0), 153.0, 21, 28),
    "lapl_25x25": ((7, 16, 16, 14), 980.0 + 1.0 / 3.0, 65, 117),
    "lapl_400x400": ((31, 139, 139, 304), 202207.0, 1160, 5069),
    "lapl_3375x3375": ((31, 425, 425, 3252), 1.48552e8, 12825, 353683),
}


def case_paths(case):
    m, o, c, b = CASES[case]
    d = os.path.join(GOLDEN, case)
    return os.path.join(d, m), os.path.join(d, o), os.path.join(d, c), os.path.join(d, b)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(case):
        if case not in cache:
            g = np.load(os.path.join(GOLDEN, case, "golden.npz"))
            n = int(g["n"])
            L = np.zeros((n, n))
            L[g["L_row"].astype(int), g["L_col"].astype(int)] = g["L_val"]
            P = np.zeros((n, n))
            P[g["pmat_row"], g["pmat_col"]] = g["pmat_val"]
            cache[case] = {"n": n, "L": L, "pmat": P, "x": g["x"], "b": g["b"], "block_nnz": g["block_nnz"]}
        return cache[case]

    return load
