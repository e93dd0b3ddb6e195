"""Development aid: N program-launch factorisations of a fixture back to back; every one's info is checked (a stalled launch is a failed
factorisation, CHOLAMD_ERR_STALL) and every factor compared bit for bit with the first.  python scripts/stress.py [case] [N] [opt=val ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

import cholesky_amd as ca
from conftest import case_paths

case = sys.argv[1] if len(sys.argv) > 1 else "lapl_3375x3375"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
m, o, c, _ = case_paths(case)
plan = ca.Plan(m, o, c)
dev = ca.Device(plan, 0)
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    dev.set_option(k, int(v))
ref = dev.new_arena()
dev.fill(ref); dev.factor(ref); dev.sync()
assert dev.info() == (0, 0), dev.info()
a = dev.new_arena()
bad = slow = 0
t_all = time.perf_counter()
for i in range(N):
    dev.fill(a)
    dev.sync()
    t0 = time.perf_counter()
    dev.factor(a)
    dev.sync()
    dt = time.perf_counter() - t0
    try:
        info = dev.info()
    except Exception as e:  # negative (internal) code: a stall
        info = ("error", str(e)[:80])
    same = bool(torch.equal(a, ref))
    if info != (0, 0) or not same or dt > 5e-3:
        bad += info != (0, 0) or not same
        slow += dt > 5e-3
        print(f"  factorisation {i}: info {info} identical {same} {dt * 1e3:.2f} ms", flush=True)
print(f"{case}: {N} factorisations, {bad} failed or different, {slow} slower than 5 ms, {time.perf_counter() - t_all:.1f} s")
sys.exit(1 if bad or slow else 0)
