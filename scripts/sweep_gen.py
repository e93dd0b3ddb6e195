"""Development aid: factor + solve of odd-shaped generated problems on the GPU, each checked against the CPU oracle."""
import os, sys, itertools, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import cholesky_amd as ca
from oracle import oracle as orc
orc.use_own_kernels()
rng = np.random.default_rng(1)
cases = [(5,4,3,2,4),(9,9,9,4,8),(10,7,13,3,16),(16,16,8,4,24),(11,11,11,5,8),(20,10,6,3,32),(14,14,14,2,40),(13,17,11,4,16),(21,5,5,5,4),(15,15,15,3,64),(8,8,8,6,4),(19,19,3,4,12)]
bad = 0
for (nx,ny,nz,lv,tile) in cases:
    with tempfile.TemporaryDirectory() as td:
        try:
            prob = ca.Problem(nx,ny,nz,lv,tile)
        except Exception as e:
            print((nx,ny,nz,lv,tile), "generator refused:", str(e)[:80]); continue
        m,o,c,b = prob.write(os.path.join(td,"g"))
        plan = prob.plan()
        O = orc.Oracle(m,o,c); O.factor()
        dev = ca.Device(plan,0); a = dev.new_arena(); dev.fill(a); dev.factor(a); dev.sync()
        L = np.tril(plan.arena_to_dense(a.cpu().numpy())); Lo = np.tril(O.dense())
        err = np.abs(L-Lo).max()/np.abs(Lo).max()
        bv = prob.rhs(); d_b = torch.from_numpy(bv).cuda(); d_x = torch.empty_like(d_b); dev.solve(a,d_b,d_x); dev.sync()
        xe = np.abs(d_x.cpu().numpy()-O.solve(bv)).max()
        ok = err <= 1e-11 and xe <= 1e-9 and dev.info()==(0,0)
        bad += not ok
        print((nx,ny,nz,lv,tile), "n=%d seps=%d  |dL|=%.1e |dx|=%.1e %s" % (plan.n, plan.nsep, err, xe, "ok" if ok else "FAIL"), flush=True)
print("failures:", bad)
