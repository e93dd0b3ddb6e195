"""Mixed precision (BASELINE config 5): fp32 factor + fp64 iterative refinement.

Not in the reference (its arithmetic is fp64 CBLAS throughout), so the oracle is the fp64 restatement: the fp32 factor must
agree with the oracle's L to single-precision accuracy, and the refined solution with the oracle's fp64 x to 1e-10 -- the
same tolerance the fp64 path is held to.  At BASELINE's full size (100^3 = 10^6 unknowns, one GPU) the size-independent
property ||b - A x|| / ||b|| <= 1e-10 is checked for both the fp64 factorisation and the mixed-precision path."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from conftest import case_paths  # noqa: E402
from oracle import oracle as orc  # noqa: E402

TOL_X = 1e-10


@pytest.mark.parametrize("dims", [(12, 12, 12, 4, 16), (18, 18, 18, 3, 48), (7, 5, 3, 3, 4), (24, 24, 12, 2, 64)])
def test_fp32_factor_and_refinement_match_the_fp64_oracle(dims, tmp_path):
    import torch
    import cholesky_amd as ca
    orc.use_own_kernels()
    nx, ny, nz, levels, tile = dims
    prob = ca.Problem(nx, ny, nz, levels, tile)
    m, o, c, _ = prob.write(os.path.join(tmp_path, "gen"))
    plan = prob.plan()
    O = orc.Oracle(m, o, c)
    O.factor()
    dev = ca.Device(plan, 0)
    a32 = dev.new_arena_f32()
    dev.fill_f32(a32)
    dev.factor_f32(a32)
    dev.sync()
    assert dev.info() == (0, 0)
    L32 = np.tril(plan.arena_to_dense(a32.cpu().numpy().astype(np.float64)))
    Lo = np.tril(O.dense())
    scale = np.abs(Lo).max()
    assert np.abs(L32 - Lo).max() <= 2e-5 * scale                      # an fp32 factor of an fp64-representable matrix
    assert np.count_nonzero(L32) <= np.count_nonzero(Lo) + plan.n       # no fill outside the reference's structure (bar rounding to zero)
    bvec = prob.rhs()
    xo = O.solve(bvec)
    d_b = torch.from_numpy(bvec).cuda()
    d_x = torch.empty_like(d_b)
    # one fp32-factor solve alone is only single-precision accurate ...
    dev.solve_f32(a32, d_b, d_x)
    dev.sync()
    rel0 = dev.residual(d_b, d_x)
    assert 1e-12 < rel0 < 1e-3
    # ... refinement in fp64 recovers the fp64 solution
    iters, rel = dev.solve_refine(a32, d_b, d_x, max_iter=20, tol=1e-13)
    x = d_x.cpu().numpy()
    assert rel <= 1e-12 and iters <= 8, (iters, rel)
    assert np.abs(x - xo).max() <= TOL_X * max(1.0, np.abs(xo).max())
    # the residual the library reports is the true one (A rebuilt on the host from the plan's fill)
    A = plan.arena_to_dense(plan.fill_host())
    A = A + np.tril(A, -1).T
    perm = plan.perm
    Ao = np.zeros_like(A)
    Ao[np.ix_(perm, perm)] = A
    true_rel = np.linalg.norm(bvec - Ao @ x) / np.linalg.norm(bvec)
    assert abs(true_rel - rel) <= 1e-13 + 0.5 * true_rel


def test_mixed_precision_on_the_reference_fixture(golden):
    """lapl_3375x3375 (BASELINE configs 3/4's matrix): fp32 factor + refinement reaches the reference's golden x."""
    import torch
    import cholesky_amd as ca
    case = "lapl_3375x3375"
    m, o, c, b = case_paths(case)
    plan = ca.Plan(m, o, c)
    dev = ca.Device(plan, 0)
    a32 = dev.new_arena_f32()
    dev.fill_f32(a32)
    dev.factor_f32(a32)
    dev.sync()
    assert dev.info() == (0, 0)
    g = golden(case)
    L32 = np.tril(plan.arena_to_dense(a32.cpu().numpy().astype(np.float64)))
    assert np.abs(L32 - g["L"]).max() <= 2e-5
    d_b = torch.from_numpy(ca.plan.read_vector(b, plan.n)).cuda()
    d_x = torch.empty_like(d_b)
    iters, rel = dev.solve_refine(a32, d_b, d_x, max_iter=20, tol=1e-13)
    assert rel <= 1e-12 and iters <= 8, (iters, rel)
    assert np.abs(d_x.cpu().numpy() - g["x"]).max() <= TOL_X * max(1.0, np.abs(g["x"]).max())


def test_fp32_factor_is_deterministic_and_reports_non_spd():
    import torch
    import cholesky_amd as ca
    m, o, c, _ = case_paths("lapl_400x400")
    plan = ca.Plan(m, o, c)
    dev = ca.Device(plan, 0)
    a, b2 = dev.new_arena_f32(), dev.new_arena_f32()
    for t in (a, b2):
        dev.fill_f32(t)
        dev.factor_f32(t)
    dev.sync()
    assert torch.equal(a, b2)
    host = plan.fill_host().astype(np.float32)
    blocks = plan.blocks
    root = blocks[(blocks[:, 0] == plan.nsep) & (blocks[:, 1] == plan.nsep)][0]
    off, ld = int(root[7]), int(root[6])
    host[off + 2 + 2 * ld] = -50.0  # third diagonal entry of the root pivot
    t = torch.from_numpy(host).cuda()
    dev.factor_f32(t)
    dev.sync()
    info, sep = dev.info()
    assert sep == plan.nsep and info == 3


@pytest.mark.parametrize("precision", ["mixed", "fp64"])
def test_config5_matrix_100_cubed_on_one_gpu(precision):
    """BASELINE config 5's matrix (synthetic 100^3 7-point Laplacian, 10^6 unknowns, 10 tree levels) on ONE MI355X:
    factor + solve, ||b - A x|| / ||b|| <= 1e-10 (verify.py's residual gate as BASELINE states it).  The fp64 arena is
    188 GB, the fp32 one 94 GB; the 8-GPU part of the configuration needs a node."""
    import torch
    import cholesky_amd as ca
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    prob = ca.Problem(100, 100, 100, 10, 64)
    plan = prob.plan()
    need = plan.arena_doubles * (4 if precision == "mixed" else 8)
    if free < need + (8 << 30):
        pytest.skip(f"needs {need / 1e9:.0f} GB of free HBM")
    dev = ca.Device(plan, 0)
    d_b = torch.from_numpy(prob.rhs()).cuda()
    d_x = torch.empty_like(d_b)
    if precision == "mixed":
        a = dev.new_arena_f32()
        dev.fill_f32(a)
        dev.factor_f32(a)
        dev.sync()
        assert dev.info() == (0, 0)
        iters, rel = dev.solve_refine(a, d_b, d_x, max_iter=30, tol=1e-11)
        assert iters <= 12, iters
    else:
        a = dev.new_arena()
        dev.fill(a)
        dev.factor(a)
        dev.sync()
        assert dev.info() == (0, 0)
        dev.solve(a, d_b, d_x)
        rel = dev.residual(d_b, d_x)
    assert rel <= 1e-10, rel
    del a, dev
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dims,world,dist_top", [((20, 20, 20, 4, 32), 2, 1), ((20, 20, 20, 4, 32), 4, 1), ((24, 24, 24, 5, 32), 8, 1), ((24, 24, 24, 5, 32), 4, 0)])
def test_mixed_precision_sharded_over_the_local_communicator(dims, world, dist_top, tmp_path):
    """BASELINE config 5 as written -- mixed precision x multi-GPU: the fp32 schedule partitioned by subtrees, the extend-add exchange
    (owner-directed under dist_top) and the broadcasts on floats (cholamd_factor_multi_f32; the rank objects share the one GPU
    through the local communicator), the complete fp32 factor gathered on rank 0 and refined there in fp64.  The sharded fp32 factor
    agrees with the single-GPU fp32 factor to fp32 rounding (the exchange adds the ranks' contributions in another order), the
    refined x with the oracle's fp64 x to 1e-10."""
    import torch
    import cholesky_amd as ca
    from cholesky_amd import _lib
    from cholesky_amd.device import factor_multi
    import ctypes as C
    orc.use_own_kernels()
    prob = ca.Problem(*dims)
    m, o, c, _ = prob.write(os.path.join(tmp_path, "gen"))
    plan = prob.plan()
    O = orc.Oracle(m, o, c)
    O.factor()
    one = ca.Device(plan, 0)
    ref32 = one.new_arena_f32()
    one.fill_f32(ref32)
    one.factor_f32(ref32)
    one.sync()
    assert one.info() == (0, 0)
    devs, arenas = [], []
    for r in range(world):
        dev = ca.Device(plan, 0)
        dev.set_option("dist_top", dist_top)
        dev.set_partition(r, world)
        a = dev.new_arena_f32()
        dev.fill_f32(a)
        devs.append(dev)
        arenas.append(a)
    factor_multi(devs, arenas, local=True)
    for dev in devs:
        assert dev.info() == (0, 0)
    # the refinement with the fp32 factor LEFT on the ranks (cholamd_solve_refine_multi: only vectors travel), before anything is gathered
    from cholesky_amd.device import solve_refine_multi
    bvec = prob.rhs()
    xo = O.solve(bvec)
    d_b = torch.from_numpy(bvec).cuda()
    xs = [torch.full_like(d_b, float("nan")) for _ in range(world)]
    iters, rel = solve_refine_multi(devs, arenas, [d_b] * world, xs, max_iter=20, tol=1e-13, local=True)
    assert rel <= 1e-12 and iters <= 8, (iters, rel)
    for r in range(world):
        assert np.abs(xs[r].cpu().numpy() - xo).max() <= TOL_X * max(1.0, np.abs(xo).max()), r
    L = _lib.load()
    n = world
    hd = (C.c_void_p * n)(*[d.h for d in devs])
    ha = (C.c_void_p * n)(*[C.c_void_p(a.data_ptr()) for a in arenas])
    _lib.check(L.cholamd_gather_factor_f32(hd, ha, n, None), "cholamd_gather_factor_f32")
    devs[0].sync()
    full32 = arenas[0].cpu().numpy().astype(np.float64)
    r32 = ref32.cpu().numpy().astype(np.float64)
    scale = max(1.0, np.abs(r32).max())
    assert np.abs(full32 - r32).max() <= 2e-5 * scale
    Lo = np.tril(O.dense())
    assert np.abs(np.tril(plan.arena_to_dense(full32)) - Lo).max() <= 2e-5 * np.abs(Lo).max()
    # refinement on rank 0 with the gathered fp32 factor (the full-tree lists on a partitioned device)
    d_x = torch.empty_like(d_b)
    iters, rel = devs[0].solve_refine(arenas[0], d_b, d_x, max_iter=20, tol=1e-13)
    assert rel <= 1e-12 and iters <= 8, (iters, rel)
    assert np.abs(d_x.cpu().numpy() - xo).max() <= TOL_X * max(1.0, np.abs(xo).max())


def test_the_leaf_skips_of_the_solve_change_nothing(monkeypatch):
    """16 leaves of 864 columns with a band of 159 rows (24^3, 5 levels): the solve reads `band` rows under a span and a panel row from its
    first entry of A on, and an fp32 factor takes every leaf's triangle in ONE launch (k_solve_leaf32).  CHOLAMD_SOLVE_NO_BAND=1 (read when a
    device object builds its solve lists) switches all of that off: same x from the same factor, fp64 and fp32."""
    import torch
    import cholesky_amd as ca
    prob = ca.Problem(24, 24, 24, 5, 32)
    plan = prob.plan()
    seps, runs = plan.solve_skips(plan.levels - 1)
    assert len(seps) == 16 and (seps[:, 1] > 256).all() and (seps[:, 2] > 0).all() and (seps[:, 2] <= 256).all() and runs[:, 4].any()
    d_b = torch.from_numpy(prob.rhs()).cuda()
    xs = {}
    for skip in (True, False):
        if skip:
            monkeypatch.delenv("CHOLAMD_SOLVE_NO_BAND", raising=False)
        else:
            monkeypatch.setenv("CHOLAMD_SOLVE_NO_BAND", "1")
        dev = ca.Device(plan, 0)
        a = dev.new_arena()
        dev.fill(a)
        dev.factor(a)
        a32 = dev.new_arena_f32()
        dev.fill_f32(a32)
        dev.factor_f32(a32)
        dev.sync()
        assert dev.info() == (0, 0)
        x64, x32 = torch.empty_like(d_b), torch.empty_like(d_b)
        dev.solve(a, d_b, x64)
        dev.solve_f32(a32, d_b, x32)
        dev.sync()
        assert dev.residual(d_b, x64) <= 1e-12
        xs[skip] = (x64.cpu().numpy(), x32.cpu().numpy())
    for k in (0, 1):  # (the sums are atomic: equal to rounding, not bit for bit)
        assert np.abs(xs[True][k] - xs[False][k]).max() <= 1e-12 * np.abs(xs[False][k]).max()
