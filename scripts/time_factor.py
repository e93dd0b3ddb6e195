"""Quick timing of the level schedule on one GPU (development aid; bench.py is the contract)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import cholesky_amd as ca

case = sys.argv[1] if len(sys.argv) > 1 else "lapl_3375x3375"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if case.startswith("gen:"):  # gen:NXxNYxNZ:levels:tile
    _, dims, lv, tile = case.split(":")
    nx, ny, nz = (int(v) for v in dims.split("x"))
    plan = ca.Problem(nx, ny, nz, int(lv), int(tile)).plan()
else:
    G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", case)
    files = sorted(os.listdir(G))
    mtx = [f for f in files if f.startswith("lapl") and f.endswith(".mtx")][0]
    ordf = [f for f in files if "_ord_" in f][0]
    clf = [f for f in files if "_clust_" in f][0]
    plan = ca.Plan(os.path.join(G, mtx), os.path.join(G, ordf), os.path.join(G, clf))
dev = ca.Device(plan, 0)
n_ar = max(1, min(reps + 3, int(6e9 // (plan.arena_doubles * 8))))
arenas = [dev.new_arena() for _ in range(n_ar)]
reps = max(1, n_ar - 3) if n_ar >= 4 else 1
for a in arenas:
    dev.fill(a)
dev.sync()
for a in arenas[:min(3, n_ar)]:
    dev.factor(a)
dev.sync()
if n_ar < 4:
    dev.fill(arenas[0]); dev.sync()
t0 = time.perf_counter()
for a in (arenas[3:] if n_ar >= 4 else arenas[:1]):
    dev.factor(a)
dev.sync()
dt = (time.perf_counter() - t0) / reps
print(f"{case}: {dt*1e6:.1f} us per factorisation, {plan.flops/dt*1e-9:.2f} GF/s (F_ref={plan.flops:.4g})")
dev.set_timing(1)
k3 = min(3, n_ar)
for a in arenas[:k3]:
    dev.fill(a)
dev.sync()
for a in arenas[:k3]:
    dev.factor(a)
dev.sync()
t = dev.get_timing()
for k, (ms, n) in t.items():
    if n:
        print(f"  {k:7s} {n:3d} launches, {ms/n*1e3:8.1f} us avg, {ms/k3*1e3:8.1f} us per factorisation")
