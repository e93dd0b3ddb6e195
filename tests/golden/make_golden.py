#!/usr/bin/env python3
"""Generate golden vectors from the reference's OWN oracle (verify.py).

Run ONLY in the authoring container, where /root/reference exists:

    python tests/golden/make_golden.py

For every fixture of the reference's test-suite (test_matrices.py:51-142) it calls
  * verify.permute_matrix (verify.py:127-213)  -> permuted lower matrix PAP^T
  * scipy.linalg.cholesky(lower=True)          -> L     (what verify.check_matrix compares, verify.py:278-287)
  * scipy.linalg.solve(A, b)                   -> x     (what verify.check_solution compares, verify.py:290-302)
and stores them as compressed COO / dense vectors in  tests/golden/<case>/golden.npz.

Only DATA is written (inputs are the reference's fixture files copied verbatim next to
this script; outputs are numbers).  No reference source text is copied.
"""
import os
import sys

import numpy as np
import scipy.io
import scipy.linalg

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    "lapl_9x9": ("lapl_3_2.mtx", "lapl_3_2_ord_2.txt", "lapl_3_2_clust_2.txt", "B_9x1.mtx"),
    "lapl_25x25": ("lapl_5_2.mtx", "lapl_5_2_ord_3.txt", "lapl_5_2_clust_3.txt", "B_25x1.mtx"),
    "lapl_400x400": ("lapl_20_2.mtx", "lapl_20_2_ord_5.txt", "lapl_20_2_clust_5.txt", "B_400x1.mtx"),
    "lapl_3375x3375": ("lapl_15_3.mtx", "lapl_15_3_ord_5.txt", "lapl_15_3_clust_5.txt", "B_3375x1.mtx"),
}


def coo(mat, tol=0.0):
    r, c = np.nonzero(np.abs(mat) > tol)
    return r.astype(np.int32), c.astype(np.int32), mat[r, c].astype(np.float64)


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; goldens can only be regenerated in the authoring container")
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    import verify  # the reference's oracle module (imports fine: numpy/scipy/pandas only)

    for case, (mtx, ord_, clust, bfile) in CASES.items():
        d = os.path.join(REF, "tests", case)
        nzs, pmat = verify.permute_matrix(os.path.join(d, mtx), os.path.join(d, ord_))
        L = scipy.linalg.cholesky(pmat, lower=True)
        A = scipy.io.mmread(os.path.join(d, mtx)).toarray()
        b = np.asarray(scipy.io.mmread(os.path.join(d, bfile)), dtype=np.float64)
        x = scipy.linalg.solve(A, b)
        pr, pc, pv = coo(pmat)
        # keep every entry of L that is not an exact zero (structural zeros of the dense factor
        # are exact zeros because scipy works on the block-sparse permuted matrix densely;
        # tiny fill values are kept)
        lr, lc, lv = coo(L)
        nz_keys = np.array([[k[0], k[1], v] for k, v in sorted(nzs.items())], dtype=np.int64)
        out = os.path.join(HERE, case, "golden.npz")
        np.savez_compressed(
            out,
            n=np.int64(A.shape[0]),
            pmat_row=pr, pmat_col=pc, pmat_val=pv,
            L_row=lr.astype(np.int16 if A.shape[0] < 32768 else np.int32),
            L_col=lc.astype(np.int16 if A.shape[0] < 32768 else np.int32),
            L_val=lv,
            x=x.reshape(-1), b=b.reshape(-1),
            block_nnz=nz_keys,
        )
        resid = np.linalg.norm(L @ L.T - (pmat + np.tril(pmat, -1).T)) / np.linalg.norm(A)
        print(f"{case}: n={A.shape[0]} nnz(pmat)={pv.size} nnz(L)={lv.size} "
              f"resid={resid:.2e} -> {out} ({os.path.getsize(out)} bytes)")
    os.chdir(cwd)


if __name__ == "__main__":
    main()
