"""Scaling curve on generated 3-D Laplacians (20^3 ... 100^3): plan build time, arena size, factorisation
time and GF/s of F_ref, residual of a solve.  One GPU.

    python scripts/scale_gen.py 20:5 30:6 40:6 60:8        # N:levels (tile 64)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import cholesky_amd as ca

for spec in sys.argv[1:]:
    parts = spec.split(":")
    n, lv = int(parts[0]), int(parts[1])
    tile = int(parts[2]) if len(parts) > 2 else 64
    t0 = time.perf_counter()
    prob = ca.Problem(n, n, n, lv, tile)
    plan = prob.plan()
    t_plan = time.perf_counter() - t0
    gb = plan.arena_doubles * 8 / 1e9
    print(f"{n}^3 levels={lv} tile={tile}: N={plan.n} seps={plan.nsep} arena={gb:.2f} GB F_ref={plan.flops:.4g} nnz(L)={plan.nnz_l} "
          f"plan+symbolic {t_plan:.1f} s", flush=True)
    t0 = time.perf_counter()
    dev = ca.Device(plan, 0)
    print(f"   device schedule upload {time.perf_counter() - t0:.1f} s", flush=True)
    a = dev.new_arena()
    times = []
    for rep in range(3):
        dev.fill(a)
        dev.sync()
        t0 = time.perf_counter()
        dev.factor(a)
        dev.sync()
        times.append(time.perf_counter() - t0)
    info = dev.info()
    dt = min(times)
    b = prob.rhs()
    d_b = torch.from_numpy(b).cuda()
    d_x = torch.empty_like(d_b)
    dev.solve(a, d_b, d_x)
    dev.sync()
    t0 = time.perf_counter()
    dev.solve(a, d_b, d_x)
    dev.sync()
    t_solve = time.perf_counter() - t0
    x = d_x.cpu().numpy()
    # residual with the generated operator: 7-point Laplacian, diag 6, off-diag -1 (natural ordering)
    X = x.reshape(n, n, n)  # index x + n (y + n z): axes (z, y, x), the operator is symmetric in them
    r = 6.0 * X
    r[1:, :, :] -= X[:-1, :, :]; r[:-1, :, :] -= X[1:, :, :]
    r[:, 1:, :] -= X[:, :-1, :]; r[:, :-1, :] -= X[:, 1:, :]
    r[:, :, 1:] -= X[:, :, :-1]; r[:, :, :-1] -= X[:, :, 1:]
    res = np.linalg.norm(r.ravel() - b) / np.linalg.norm(b)
    print(f"   factor {dt*1e3:.2f} ms = {plan.flops/dt*1e-12:.2f} TF/s of F_ref, info={info}; solve {t_solve*1e3:.2f} ms ({plan.arena_doubles*8*2/t_solve*1e-12:.2f} TB/s over the arena, twice), |Ax-b|/|b| = {res:.2e}", flush=True)
    del dev, a
