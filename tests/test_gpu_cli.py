"""End-to-end test of the C driver with mmat.rg's command line (the reference's test_matrices.py
pattern: run the program, then check_matrix / check_solution on its output files)."""
import os
import subprocess

import numpy as np
import pytest
import scipy.io

from conftest import CASES, ROOT, case_paths

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "cholesky_amd", "bin", "cholamd_mmat")


@pytest.mark.parametrize("case", list(CASES))
def test_cli_matches_reference_gates(case, tmp_path, golden):
    m, o, c, b = case_paths(case)
    fac, sol, perm = tmp_path / "factored.mtx", tmp_path / "solution.txt", tmp_path / "permuted.mtx"
    args = [BIN, "-i", m, "-s", o, "-c", c, "-b", b, "-o", str(sol), "-m", str(fac), "-p", str(perm),
            "-fflow", "0", "-ll:cpu", "3", "-fcuda", "0"]  # Legion flags of test_matrices.py:27 are ignored
    r = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert "Iterations: 1" in out and "Done fill." in out and "Done factoring Iteration: 0." in out and "Done solve." in out
    g = golden(case)
    # verify.check_matrix: tril(mmread(L)) vs scipy cholesky of the permuted matrix, rtol = atol = 1e-4
    L = np.tril(scipy.io.mmread(str(fac)).toarray())
    assert np.allclose(g["L"], L, rtol=1e-4, atol=1e-4)
    # verify.check_solution
    x = np.genfromtxt(str(sol)).reshape(-1)
    assert np.allclose(g["x"], x, rtol=1e-4, atol=1e-4)
    # -p: permuted matrix
    P = np.tril(scipy.io.mmread(str(perm)).toarray())
    assert np.allclose(P, g["pmat"], atol=1e-7)
    assert open(fac).readline().strip() == "%%MatrixMarket matrix coordinate real hermitian"


def test_cli_full_precision_and_iterations(tmp_path, golden):
    m, o, c, b = case_paths("lapl_400x400")
    fac, sol = tmp_path / "f.mtx", tmp_path / "x.txt"
    r = subprocess.run([BIN, "-i", m, "-s", o, "-c", c, "-b", b, "-o", str(sol), "-m", str(fac), "--iterations", "3", "--full-precision"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Done factoring Iteration: 2." in r.stdout
    g = golden("lapl_400x400")
    L = np.tril(scipy.io.mmread(str(fac)).toarray())
    assert np.abs(L - g["L"]).max() <= 1e-12
    assert np.abs(np.genfromtxt(str(sol)) - g["x"]).max() <= 1e-10


def test_cli_reports_missing_input(tmp_path):
    r = subprocess.run([BIN, "-i", str(tmp_path / "nope.mtx"), "-s", "a", "-c", "b"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "cannot open" in r.stderr
