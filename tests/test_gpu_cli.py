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


def test_cli_mixed_precision_and_repeat(tmp_path, golden):
    """--precision mixed: fp32 factor (written widened), solution refined to fp64 accuracy; --repeat = --iterations."""
    m, o, c, b = case_paths("lapl_3375x3375")
    fac, sol = tmp_path / "f.mtx", tmp_path / "x.txt"
    r = subprocess.run([BIN, "-i", m, "-s", o, "-c", c, "-b", b, "-o", str(sol), "-m", str(fac), "--precision", "mixed", "--repeat", "2", "--full-precision"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Done factoring Iteration: 1." in r.stdout and "iterative refinement" in r.stderr
    g = golden("lapl_3375x3375")
    L = np.tril(scipy.io.mmread(str(fac)).toarray())
    assert np.abs(L - g["L"]).max() <= 2e-5
    assert np.abs(np.genfromtxt(str(sol)) - g["x"]).max() <= 1e-10 * max(1.0, np.abs(g["x"]).max())


def test_cli_gpus_flag_is_honest(tmp_path):
    """--gpus N needs N devices (the test box has one): the program must refuse, not fall back to one GPU;
    --gpus 1 is the plain path; a non-power-of-two is an argument error."""
    m, o, c, b = case_paths("lapl_400x400")
    import torch
    n = torch.cuda.device_count()
    r = subprocess.run([BIN, "-i", m, "-s", o, "-c", c, "--gpus", str(2 * n if n & (n - 1) == 0 else 16)], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "HIP devices are visible" in r.stderr
    r = subprocess.run([BIN, "-i", m, "-s", o, "-c", c, "--gpus", "3"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "power of two" in r.stderr
    sol = tmp_path / "x.txt"
    r = subprocess.run([BIN, "-i", m, "-s", o, "-c", c, "-b", b, "-o", str(sol), "--gpus", "1"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    if n >= 2:  # a node: the sharded path through cholamd_factor_multi (one process, RCCL group of all-reduces)
        sol2 = tmp_path / "x2.txt"
        r = subprocess.run([BIN, "-i", m, "-s", o, "-c", c, "-b", b, "-o", str(sol2), "--gpus", "2", "--full-precision"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert np.allclose(np.genfromtxt(str(sol)), np.genfromtxt(str(sol2)), rtol=1e-6, atol=1e-6)
