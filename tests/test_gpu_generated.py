"""Generated problems (cholamd_generate_laplacian): parity of the HIP path with the oracle on inputs the
reference never shipped, including pivots larger than the register-resident kernels take (> 272 columns:
factored in 256-column blocks by the level schedule) and ragged / tiny grids."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as orc  # noqa: E402

# (nx, ny, nz, levels, tile)
PROBLEMS = [
    (3, 3, 1, 2, 8),        # the 9x9 fixture's matrix
    (7, 5, 3, 3, 4),        # ragged, tiny tiles
    (12, 12, 12, 4, 16),    # 1728 dof
    (18, 18, 18, 3, 48),    # root separator 18*18 = 324 > 272: blocked big pivot
    (24, 24, 12, 2, 64),    # root 24*12 = 288, two big leaves of 24*12*11.. > 272 columns each
]


@pytest.mark.parametrize("dims", PROBLEMS)
def test_generated_problem_matches_oracle(dims, tmp_path):
    import torch
    import cholesky_amd as ca
    orc.use_own_kernels()
    nx, ny, nz, levels, tile = dims
    prob = ca.Problem(nx, ny, nz, levels, tile)
    m, o, c, b = prob.write(os.path.join(tmp_path, "gen"))
    plan = prob.plan()
    O = orc.Oracle(m, o, c)
    O.factor(log_ops=True)
    assert np.array_equal(plan.ops(), O.ops())
    dev = ca.Device(plan, 0)
    arena = dev.new_arena()
    dev.fill(arena)
    dev.factor(arena)
    dev.sync()
    assert dev.info() == (0, 0)
    L = np.tril(plan.arena_to_dense(arena.cpu().numpy()))
    Lo = np.tril(O.dense())
    scale = np.abs(Lo).max()
    assert np.abs(L - Lo).max() <= 1e-11 * scale
    A = plan.arena_to_dense(plan.fill_host())
    A = A + np.tril(A, -1).T
    assert np.linalg.norm(L @ L.T - A) / np.linalg.norm(A) <= 1e-10
    # solve against the generated right-hand side
    bvec = prob.rhs()
    assert np.array_equal(bvec, orc.read_vector(b, prob.n))
    d_b = torch.from_numpy(bvec).cuda()
    d_x = torch.empty_like(d_b)
    dev.solve(arena, d_b, d_x)
    dev.sync()
    x = d_x.cpu().numpy()
    xo = O.solve(bvec)
    assert np.abs(x - xo).max() <= 1e-9 * max(1.0, np.abs(xo).max())


@pytest.mark.parametrize("dims", [(18, 18, 18, 3, 48), (24, 24, 12, 2, 64), (12, 12, 12, 4, 16)])
def test_throughput_trsm_matches_the_fused_path(dims):
    """Level schedule with every step's strips through k_trsm_wt (option trsm_wt_min = 1: POTRF launch, then one wave per strip,
    twelve strips per workgroup sharing the pivot block's LDS image) against the default fused launches: the same factor."""
    import cholesky_amd as ca
    prob = ca.Problem(*dims)
    plan = prob.plan()
    res = []
    for wt in (0, 1):
        dev = ca.Device(plan, 0)
        dev.set_option("program", 0)
        dev.set_option("trsm_wt_min", wt)
        arena = dev.new_arena()
        dev.fill(arena)
        dev.factor(arena)
        dev.sync()
        assert dev.info() == (0, 0)
        res.append(np.tril(plan.arena_to_dense(arena.cpu().numpy())))
    scale = np.abs(res[0]).max()
    assert np.abs(res[0] - res[1]).max() <= 1e-12 * scale


def test_big_pivot_info_is_pivot_relative():
    """A non-positive pivot inside a later 256-column block of a big pivot reports its column within the pivot."""
    import torch
    import cholesky_amd as ca
    prob = ca.Problem(18, 18, 18, 3, 48)
    plan = prob.plan()
    dev = ca.Device(plan, 0)
    host = plan.fill_host()
    blocks = plan.blocks
    root = blocks[(blocks[:, 0] == plan.nsep) & (blocks[:, 1] == plan.nsep)][0]
    off, ld = int(root[7]), int(root[6])
    col = 300  # in the second column block of the 324-column root pivot
    host[off + col + col * ld] = -1e6
    arena = torch.from_numpy(host).cuda()
    dev.factor(arena)
    dev.sync()
    info, sep = dev.info()
    assert sep == plan.nsep and info == col + 1


@pytest.mark.parametrize("opts", [
    {"split_min": 64, "split_nb": 64, "super_blocks": 2},   # root pivot 324 -> 6 column blocks, 3 super-blocks: narrow + wide trailing updates
    {"split_min": 64, "split_nb": 64, "super_blocks": 4},
    {"split_min": 64, "split_nb": 64, "super_blocks": 1},   # one update per column block (round 1's schedule)
    {"split_min": 96, "split_nb": 96, "super_blocks": 3, "fuse": 0},
], ids=lambda o: "+".join(f"{k}={v}" for k, v in o.items()))
def test_super_block_trailing_updates_match_oracle(opts, tmp_path):
    """Wide pivots are factored in column blocks; the trailing matrix beyond a SUPER-block of them gets one update of rank
    (columns of the super-block) instead of one per block.  Same factor as the oracle for every grouping."""
    import cholesky_amd as ca
    orc.use_own_kernels()
    prob = ca.Problem(18, 18, 18, 3, 48)
    m, o, c, _ = prob.write(os.path.join(tmp_path, "gen"))
    plan = prob.plan()
    O = orc.Oracle(m, o, c)
    O.factor()
    dev = ca.Device(plan, 0)
    for k, v in opts.items():
        dev.set_option(k, v)
    arena = dev.new_arena()
    dev.fill(arena)
    dev.factor(arena)
    dev.sync()
    assert dev.info() == (0, 0)
    L = np.tril(plan.arena_to_dense(arena.cpu().numpy()))
    Lo = np.tril(O.dense())
    assert np.abs(L - Lo).max() <= 1e-11 * np.abs(Lo).max()
    # the fp32 schedule (blocks of 128 columns) uses the same grouping
    a32 = dev.new_arena_f32()
    dev.fill_f32(a32)
    dev.factor_f32(a32)
    dev.sync()
    L32 = np.tril(plan.arena_to_dense(a32.cpu().numpy().astype(np.float64)))
    assert np.abs(L32 - Lo).max() <= 2e-5 * np.abs(Lo).max()


@pytest.mark.parametrize("dims", [(20, 20, 20, 4, 16), (24, 24, 12, 3, 32), (18, 18, 18, 3, 48)])
def test_merged_targets_and_row_compaction_keep_the_factor(dims):
    """Level schedule on 64x64 macro tiles (mt_min_tiles = 1 forces them on a small problem): extend-add targets and panel row runs merged
    where they are neighbours in storage (option merge_targets) against the unmerged lists.  Every element keeps its sources and
    their order; a merged target may move from the 16x16 kernel (four waves split K) to the macro-tile kernel (K in 16-deep chunks), so the
    factors agree to rounding (1e-13 of the largest entry), bit for bit where the targets were macro tiles before."""
    import cholesky_amd as ca
    orc.use_own_kernels()
    prob = ca.Problem(*dims)
    plan = prob.plan()
    assert plan.arena_doubles <= plan.arena_dense_doubles
    res = []
    for merge in (0, 1):
        dev = ca.Device(plan, 0)
        dev.set_option("program", 0)
        dev.set_option("mt_min_tiles", 1)
        dev.set_option("merge_targets", merge)
        arena = dev.new_arena()
        dev.fill(arena)
        dev.factor(arena)
        dev.sync()
        assert dev.info() == (0, 0)
        res.append(arena.cpu().numpy())
    assert np.abs(res[0] - res[1]).max() <= 1e-13 * np.abs(res[0]).max()
    L = np.tril(plan.arena_to_dense(res[1]))
    A = plan.arena_to_dense(plan.fill_host())
    A = A + np.tril(A, -1).T
    assert np.linalg.norm(L @ L.T - A) / np.linalg.norm(A) <= 1e-10


@pytest.mark.parametrize("dims", [(24, 24, 24, 5, 32), (30, 30, 10, 3, 64), (12, 12, 12, 4, 16)])
@pytest.mark.parametrize("f32", [False, True])
def test_the_leaf_envelope_of_the_level_schedule_keeps_the_factor(dims, f32):
    """Option leaf_envelope (level schedule): the leaves' TRSM strips, trailing updates and extend-add sources leave out what is structurally zero
    (a leaf's factor stays inside the envelope of A).  The whole arena -- every stored entry, the skipped zeros included -- equals the arena of the
    dense lists; fewer tasks and strips are launched."""
    import cholesky_amd as ca
    prob = ca.Problem(*dims)
    plan = prob.plan()
    res, counts = [], []
    for env in (0, 1):
        dev = ca.Device(plan, 0)
        dev.set_option("program", 0)
        dev.set_option("mt_min_tiles", 1)
        dev.set_option("leaf_envelope", env)
        if f32:
            arena = dev.new_arena_f32()
            dev.fill_f32(arena)
            dev.factor_f32(arena)
        else:
            arena = dev.new_arena()
            dev.fill(arena)
            dev.factor(arena)
        dev.sync()
        assert dev.info() == (0, 0)
        res.append(arena.cpu().numpy().astype(np.float64))
    scale = np.abs(res[0]).max()
    assert np.abs(res[0] - res[1]).max() <= (2e-6 if f32 else 1e-13) * scale  # (shorter sums: another rounding of the same values, no other difference)
    assert np.array_equal(res[0] == 0.0, res[1] == 0.0)                         # the same zero pattern, entry by entry
