"""Development aid: run LAPACKE_dpotrf (GPU) on a range of sizes; use under rocprofv3 --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cholesky_amd as ca
rng = np.random.default_rng(0)
for n in [16, 32, 48, 64, 96, 128, 160, 192, 224, 256, 272]:
    G = rng.standard_normal((n, n))
    A = np.asfortranarray(np.tril(G @ G.T + n * np.eye(n)))
    for _ in range(3):
        B = A.copy(order="F")
        assert ca.blas.LAPACKE_dpotrf(B) == 0
    print(n, "ok")
