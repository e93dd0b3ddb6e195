"""Development aid: parity and time of the three launch structures (program + followers, program, level by level) on the
reference fixtures (or gen:NX,NY,NZ,levels,tile problems against the level-by-level factor)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

import cholesky_amd as ca
from conftest import CASES, GOLDEN, case_paths

cases = sys.argv[1:] or list(CASES)
for case in cases:
    if case.startswith("gen:"):
        dims = [int(v) for v in case[4:].split(",")]
        plan = ca.Problem(*dims).plan()
        ref = None
    else:
        m, o, c, b = case_paths(case)
        plan = ca.Plan(m, o, c)
        g = np.load(os.path.join(GOLDEN, case, "golden.npz"))
        ref = np.zeros((plan.n, plan.n))
        ref[g["L_row"].astype(int), g["L_col"].astype(int)] = g["L_val"]
    base = None
    for name, opts in (("levels", {"program": 0}), ("program", {"follow": 0}), ("program+follow", {})):
        dev = ca.Device(plan, 0)
        for k, v in opts.items():
            dev.set_option(k, v)
        reps = 50 if plan.arena_doubles * 8 * 53 < 6e9 else 3
        arenas = [dev.new_arena() for _ in range(reps + 3)]
        for a in arenas:
            dev.fill(a)
        dev.sync()
        for a in arenas[:3]:
            dev.factor(a)
        dev.sync()
        info = dev.info()
        t0 = time.perf_counter()
        for a in arenas[3:]:
            dev.factor(a)
        dev.sync()
        dt = (time.perf_counter() - t0) / reps
        L = arenas[-1].cpu().numpy()
        same = all(torch.equal(arenas[3], a) for a in arenas[4:])
        if ref is not None:
            err = float(np.abs(np.tril(plan.arena_to_dense(L)) - ref).max())
        else:
            if base is None:
                base = L
            err = float(np.abs(L - base).max() / max(1.0, np.abs(base).max()))
        print(f"{case:16s} {name:15s} info {info}  max|dL| {err:.2e}  deterministic {same}  {dt * 1e6:8.1f} us  {plan.flops / dt * 1e-9:9.1f} GF/s", flush=True)
        del arenas, dev
