"""GPU parity of the hot path (driver level L-C and task level L-B of include/cholamd.h).

The HIP factorisation of every reference fixture is compared with
  * the CPU oracle on the same inputs (oracle/chol_oracle.c) and
  * the golden L / x produced by the reference's own verify.py (tests/golden/*/golden.npz),
to an fp64 tolerance of 1e-12 (absolute, entries of L are O(1)); BASELINE's gate is a residual
<= 1e-10.  Size-independent properties are checked at the largest size: ||L L^T - P A P^T||_F /
||A||_F and ||A x - b|| / ||b||.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from conftest import CASES, KNOWN, case_paths  # noqa: E402
from oracle import oracle as orc  # noqa: E402

TOL_L = 1e-12
TOL_RESID = 1e-10  # BASELINE.json north_star


@pytest.fixture(scope="module")
def ca():
    import cholesky_amd
    orc.use_own_kernels()
    return cholesky_amd


@pytest.fixture(scope="module")
def runs(ca):
    """Factor every fixture once on the GPU; keep plan, device, arenas on the host."""
    import torch
    out = {}
    for case in CASES:
        m, o, c, b = case_paths(case)
        plan = ca.Plan(m, o, c)
        dev = ca.Device(plan, 0)
        arena = dev.new_arena()
        dev.fill(arena)
        dev.sync()
        filled = arena.cpu().numpy().copy()
        dev.factor(arena)
        dev.sync()
        info = dev.info()
        bvec = ca.plan.read_vector(b, plan.n)
        d_b = torch.from_numpy(bvec).cuda()
        d_x = torch.empty_like(d_b)
        dev.solve(arena, d_b, d_x)
        dev.sync()
        out[case] = dict(plan=plan, dev=dev, arena_t=arena, filled=filled, L=arena.cpu().numpy().copy(), info=info,
                         x=d_x.cpu().numpy().copy(), b=bvec)
    return out


@pytest.mark.parametrize("case", list(CASES))
def test_device_scatter_equals_host_fill_and_reference_permutation(case, runs, golden):
    r = runs[case]
    plan = r["plan"]
    assert np.array_equal(r["filled"], plan.fill_host())               # device A scatter == host fill_block
    assert np.array_equal(plan.arena_to_dense(r["filled"]), golden(case)["pmat"])  # == verify.permute_matrix


@pytest.mark.parametrize("case", list(CASES))
def test_factor_matches_oracle_and_reference_golden(case, runs, golden):
    r = runs[case]
    plan = r["plan"]
    assert r["info"] == (0, 0)
    L = np.tril(plan.arena_to_dense(r["L"]))
    m, o, c, _ = case_paths(case)
    O = orc.Oracle(m, o, c)
    O.factor()
    assert np.abs(L - np.tril(O.dense())).max() <= TOL_L          # vs CPU oracle
    assert np.abs(L - golden(case)["L"]).max() <= TOL_L           # vs reference verify.py + scipy
    assert np.count_nonzero(L) == KNOWN[case][3]                  # same non-zero structure as the reference's L
    # the reference's own gate (verify.check_matrix: allclose rtol=atol=1e-4)
    assert np.allclose(golden(case)["L"], L, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("case", list(CASES))
def test_residuals(case, runs, golden):
    r = runs[case]
    g = golden(case)
    L = np.tril(r["plan"].arena_to_dense(r["L"]))
    A = g["pmat"] + np.tril(g["pmat"], -1).T
    assert np.linalg.norm(L @ L.T - A) / np.linalg.norm(A) <= TOL_RESID
    # solution: golden x, and ||A x - b|| / ||b|| with A in ORIGINAL ordering
    perm = r["plan"].perm
    Aorig = np.zeros_like(A)
    Aorig[np.ix_(perm, perm)] = A
    assert np.abs(r["x"] - g["x"]).max() <= 1e-10 * max(1.0, np.abs(g["x"]).max())
    assert np.linalg.norm(Aorig @ r["x"] - r["b"]) / np.linalg.norm(r["b"]) <= TOL_RESID
    assert np.allclose(g["x"], r["x"], rtol=1e-4, atol=1e-4)     # verify.check_solution's gate


def test_factor_is_deterministic(runs):
    """Target-centric updates use no atomics: two runs give bit-identical factors."""
    r = runs["lapl_3375x3375"]
    dev, arena = r["dev"], r["arena_t"]
    dev.fill(arena)
    dev.factor(arena)
    dev.sync()
    assert np.array_equal(arena.cpu().numpy(), r["L"])


def test_refactor_many_times_stable(runs):
    """The reference's --iterations loop (mmat.rg:1212-1358): re-fill then factor, repeatedly."""
    r = runs["lapl_400x400"]
    dev, arena = r["dev"], r["arena_t"]
    for _ in range(5):
        dev.fill(arena)
        dev.factor(arena)
    dev.sync()
    assert np.array_equal(arena.cpu().numpy(), r["L"])


def test_non_spd_matrix_reports_separator_and_column(ca, runs):
    """A pivot that is not positive definite yields LAPACK-style info + the separator label."""
    import torch
    r = runs["lapl_25x25"]
    plan, dev = r["plan"], r["dev"]
    host = plan.fill_host()
    blocks = plan.blocks
    root = blocks[(blocks[:, 0] == plan.nsep) & (blocks[:, 1] == plan.nsep)][0]
    off, ld = int(root[7]), int(root[6])
    host[off + 1 + 1 * ld] = -50.0  # second diagonal entry of the root pivot
    arena = torch.from_numpy(host).cuda()
    dev.factor(arena)
    dev.sync()
    info, sep = dev.info()
    assert sep == plan.nsep and info == 2
    # the info words alternate between two slots from one program launch to the next (each launch clears the other slot: no memset
    # per factorisation): good and failing factorisations in any order report their own outcome
    good = dev.new_arena()
    for bad_turn in (False, False, True, False, True, True, False):
        if bad_turn:
            a = torch.from_numpy(host).cuda()
        else:
            dev.fill(good)
            a = good
        dev.factor(a)
        dev.sync()
        assert dev.info() == ((2, plan.nsep) if bad_turn else (0, 0))
    dev.set_option("program", 0)  # the level-by-level path in between (slot 0, cleared by a memset)
    dev.factor(torch.from_numpy(host).cuda())
    dev.sync()
    assert dev.info() == (2, plan.nsep)
    dev.set_option("program", 1)
    dev.fill(good)
    dev.factor(good)
    dev.sync()
    assert dev.info() == (0, 0)


# ------------------------------------------------------------------------------------------------
# task level: replay mmat.rg:1227-1355 literally through the fused_* entry points
# ------------------------------------------------------------------------------------------------
def _snap_by_block(plan, lbl):
    from cholesky_amd import Filled
    buf, n = plan.snapshot(lbl)
    d = {}
    for i in range(n):
        f = buf[i]
        d.setdefault((f.sep_x, f.sep_y), []).append(Filled(f.filled, f.sep_x, f.sep_y, f.interval, f.cluster, f.lo_x, f.lo_y, f.hi_x, f.hi_y))
    return d


@pytest.mark.parametrize("case", ["lapl_9x9", "lapl_25x25", "lapl_400x400", "lapl_3375x3375"])
def test_fused_tasks_replay_reference_main_loop(case, ca, runs):
    import torch
    blas = ca.blas
    r = runs[case]
    plan = r["plan"]
    arena = torch.from_numpy(plan.fill_host()).cuda()
    base = arena.data_ptr()
    def reg(rs, cs):  # the block instance inside the arena (row-compacted above the parent block: Region.tile_row)
        return blas.plan_region(plan, base, rs, cs)

    tree = plan.tree  # heap index-1 -> label
    levels = plan.levels
    sizes = plan.sep_sizes
    interval, lbl = 0, 0
    ntiles = _ntiles_table(case)
    for lvl in range(levels - 1, -1, -1):
        filled = _snap_by_block(plan, lbl)
        heap = range(1 << lvl, 1 << (lvl + 1))
        for h in heap:  # POTRF sweep
            s = int(tree[h - 1])
            assert blas.fused_dpotrf(reg(s, s), filled.get((s, s), []), lvl, lbl) == 0
        for h in heap:  # TRSM sweep
            s = int(tree[h - 1])
            hp = h // 2
            while hp >= 1:
                par = int(tree[hp - 1])
                blas.fused_dtrsm(reg(s, s), reg(par, s), filled.get((s, s), []), filled.get((par, s), []), lvl, lbl)
                hp //= 2
        for h in heap:  # SYRK / GEMM sweep
            s = int(tree[h - 1])
            hp = h // 2
            while hp >= 1:
                par = int(tree[hp - 1])
                ccs = ntiles[par][interval]
                hg = hp
                while hg >= 1:
                    gp = int(tree[hg - 1])
                    args = (reg(gp, s), reg(par, s), reg(gp, par), filled.get((gp, s), []), filled.get((par, s), []), filled.get((gp, par), []), ccs, lvl, lbl)
                    (blas.fused_dsyrk if gp == par else blas.fused_dgemm)(*args)
                    hg //= 2
                hp //= 2
        lbl += 1
        if lvl <= levels - 2:
            interval += 1
    got = arena.cpu().numpy()
    assert np.abs(got - r["L"]).max() <= TOL_L
    assert sizes.sum() == plan.n


def test_fused_task_refuses_a_tile_without_storage(ca, runs):
    """A block instance of the arena above the parent block stores only the 16-row tiles a filled tile touches: a Filled record
    that points into a tile without storage is an argument error, not a write somewhere else."""
    import torch
    from cholesky_amd import CholamdError
    from cholesky_amd._lib import Filled
    blas = ca.blas
    plan = runs["lapl_3375x3375"]["plan"]
    off = plan.sep_offsets
    tree = plan.tree
    found = None
    for b in plan.blocks:  # a compacted block with a tile that is not stored
        r, c = int(b[0]), int(b[1])
        if r == c or plan.heap_of(r) == plan.heap_of(c) // 2:
            continue
        tm = plan.block_tile_map(r, c)
        gone = np.nonzero(tm < 0)[0]
        if len(gone) and (tm >= 0).any():
            found = (r, c, int(gone[0]))
            break
    assert found is not None
    r, c, t = found
    arena = torch.zeros(plan.arena_doubles, dtype=torch.float64, device="cuda")
    rA, rB = blas.plan_region(plan, arena.data_ptr(), c, c), blas.plan_region(plan, arena.data_ptr(), r, c)
    n_c = int(plan.sep_sizes[c - 1])
    fa = [Filled(0, c, c, 0, 0, int(off[c - 1]), int(off[c - 1]), int(off[c - 1]) + n_c - 1, int(off[c - 1]) + n_c - 1)]
    lo = int(off[r - 1]) + 16 * t
    fb = [Filled(0, r, c, 0, 0, lo, int(off[c - 1]), lo + 3, int(off[c - 1]) + n_c - 1)]
    with pytest.raises(CholamdError, match="does not store"):
        blas.fused_dtrsm(rA, rB, fa, fb, 0, 0)


def _ntiles_table(case):
    """clusters[sep][interval].volume - 1 for every separator, parsed from the clusters fixture."""
    table = {}
    with open(case_paths(case)[2]) as f:
        for line in f.read().splitlines()[1:]:
            if ";" not in line:
                continue
            k, rest = line.split(";", 1)
            lists = [x for x in rest.split(";") if any(ch.isdigit() for ch in x)]
            table[int(k) + 1] = [len([v for v in lst.split(",") if v.strip() != ""]) - 1 for lst in lists]
    return table


@pytest.mark.parametrize("opts", [
    {"follow": 0},                                     # the one-launch program without followers (update jobs carry every contribution)
    {"follow_tail": 0},                                # followers take every column tile of their sources themselves (no early update jobs)
    {"follow_tail": 2},                                # ... only the last two (most of the contribution through early jobs)
    {"follow_tail": 7},
    {"skyline": 0},                                    # leaf pivots without their skylines: dense POTRF, every leaf split at split_min
    {"stage_chunk": 4},                                # unsplit banded leaves handed to the extend-add four column tiles at a time
    {"fine_upd": 0},                                   # followed strips that wait for every update job into their panel
    {"staged": 0},                                     # extend-add jobs that wait for every source before they start (no staged waits)
    {"follow": 0, "staged": 1},                        # without followers every contribution goes through (staged) update jobs
    {"split_min": 96, "split_nb": 96, "follow_tail": 1}, # three / four column blocks per leaf: chains of followers with early jobs
    {"program": 0},                                    # level by level: fused POTRF+TRSM launches + update launches
    {"program": 0, "fuse": 0},                         # separate POTRF / TRSM launches (k_potrf_rr, k_trsm_rr, k_trsm_w)
    {"program": 0, "trsm_wt_min": 1},                  # every step's strips through the throughput TRSM (k_trsm_wt, after a POTRF launch of its own)
    {"program": 0, "trsm_wt_min": 1, "split_min": 64, "split_nb": 48}, # ... with three-tile pivot blocks
    {"program": 0, "fuse_update_max": 100000},         # update tasks inside the fused launch
    {"program": 0, "split_min": 272, "split_nb": 256}, # pivots factored whole (k_potrf_rr up to 17 tiles)
    {"split_min": 64, "split_nb": 64},                 # many column-block steps per pivot (program: chains of followers)
    {"program": 0, "split_min": 64, "split_nb": 64},
    {"split_min": 192, "split_nb": 192},               # program with blocks too wide to follow (12 tiles): plain POTRF jobs
    {"cells": 0},                                      # extend-add by the reference's cluster tiles instead of grid cells (no followers)
    {"program": 0, "cells": 0},
    {"solve_reference_shape": 1},                      # the per-call (deterministic) solve kernels
], ids=lambda o: "+".join(f"{k}={v}" for k, v in o.items()))
def test_alternative_launch_paths_keep_parity(opts, ca, golden):
    """Every schedule / launch variant selectable through cholamd_device_set_option factors lapl_3375 to the same L
    (reference golden, 1e-12) and solves to the same x."""
    import torch
    case = "lapl_3375x3375"
    m, o, c, b = case_paths(case)
    plan = ca.Plan(m, o, c)
    dev = ca.Device(plan, 0)
    for k, v in opts.items():
        dev.set_option(k, v)
    arena = dev.new_arena()
    dev.fill(arena)
    dev.factor(arena)
    dev.sync()
    assert dev.info() == (0, 0)
    g = golden(case)
    L = np.tril(plan.arena_to_dense(arena.cpu().numpy()))
    assert np.abs(L - g["L"]).max() <= TOL_L
    d_b = torch.from_numpy(ca.plan.read_vector(b, plan.n)).cuda()
    d_x = torch.empty_like(d_b)
    dev.solve(arena, d_b, d_x)
    dev.sync()
    assert np.abs(d_x.cpu().numpy() - g["x"]).max() <= 1e-9


def test_unknown_option_is_refused(ca, runs):
    with pytest.raises(ca.CholamdError):
        runs["lapl_9x9"]["dev"].set_option("no_such_switch", 1)


@pytest.mark.parametrize("case", ["lapl_400x400", "lapl_3375x3375"])
def test_role_tables_of_the_schedule_equal_the_kernels_own(case, ca, monkeypatch):
    """The POTRF role's tables (tile -> slot and wave, per-step work masks) come with the descriptors (chol_potrf_table, built with the
    schedule); a descriptor without one has them built by the workgroup's prologue.  Same tables, so bit-identical factors, for the program
    launch and the level-by-level launches."""
    import torch
    m, o, c, _ = case_paths(case)
    plan = ca.Plan(m, o, c)
    for opts in ({}, {"program": 0}):
        res = []
        for no_tables in (False, True):
            if no_tables:
                monkeypatch.setenv("CHOLAMD_NO_ROLE_TABLES", "1")
            else:
                monkeypatch.delenv("CHOLAMD_NO_ROLE_TABLES", raising=False)
            dev = ca.Device(plan, 0)
            for k, v in opts.items():
                dev.set_option(k, v)
            a = dev.new_arena()
            dev.fill(a)
            dev.factor(a)
            dev.sync()
            assert dev.info()[0] == 0
            res.append(a.clone())
        assert torch.equal(res[0], res[1]), opts
    monkeypatch.delenv("CHOLAMD_NO_ROLE_TABLES", raising=False)
