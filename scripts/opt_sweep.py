"""Development aid: time and parity of the program launch on a fixture under a sweep of one schedule option.
python scripts/opt_sweep.py <case> <option> v1 v2 ... [other=val ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

import cholesky_amd as ca
from conftest import GOLDEN, case_paths

case, opt = sys.argv[1], sys.argv[2]
vals = [a for a in sys.argv[3:] if "=" not in a]
fixed = dict(a.split("=") for a in sys.argv[3:] if "=" in a)
m, o, c, b = case_paths(case)
plan = ca.Plan(m, o, c)
g = np.load(os.path.join(GOLDEN, case, "golden.npz"))
ref = np.zeros((plan.n, plan.n))
ref[g["L_row"].astype(int), g["L_col"].astype(int)] = g["L_val"]
for v in vals:
    dev = ca.Device(plan, 0)
    for k, x in fixed.items():
        dev.set_option(k, int(x))
    dev.set_option(opt, int(v))
    reps = 50
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
    same = all(torch.equal(arenas[3], a) for a in arenas[4:])
    err = float(np.abs(np.tril(plan.arena_to_dense(arenas[-1].cpu().numpy())) - ref).max())
    print(f"{case} {opt}={v:>4} {fixed} info {info} max|dL| {err:.2e} deterministic {same} {dt * 1e6:8.1f} us", flush=True)
    del arenas, dev
