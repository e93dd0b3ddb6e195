"""Multi-rank tests of the subtree sharding.

* CPU (gloo, world_size 2 and 8): the host-side partition logic -- per-rank fills whose tails sum to the
  full fill, and per-rank work lists that tile the full schedule -- with a real all-reduce.
* GPU (marker gpu; gloo, 2 and 4 ranks sharing the one GPU of the test box): the sharded
  factorisation equals the single-GPU one to 1e-12.  World 8 (BASELINE config 4's rank count) runs its
  eight partitions in ONE process (the box admits at most 6 processes on the card), the exchange being a
  device-side sum of the eight tails.
* RCCL: the C-ABI communicator is exercised for real -- a one-rank ncclAllReduce on the one GPU, and the
  complete cholamd_factor_sharded path on two GPUs when the box has them (skipped otherwise)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import case_paths


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cpu_worker(rank, world, port, case, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cholesky_amd as ca
    plan = ca.Plan(*case_paths(case)[:3])
    arena, tail = plan.fill_host_part(rank, world)
    t = torch.from_numpy(arena[tail:].copy())
    dist.all_reduce(t)
    full = plan.fill_host()
    ok_tail = bool(np.array_equal(t.numpy(), full[tail:]))
    d = world.bit_length() - 1
    counts = torch.tensor([plan.level_work_counts(l, rank, world) for l in range(plan.levels)], dtype=torch.int64)
    below = counts.clone()
    below[:d] = 0  # levels above the cut are replicated, not shared
    dist.all_reduce(below)
    whole = torch.tensor([plan.level_work_counts(l) for l in range(plan.levels)], dtype=torch.int64)
    ok_lists = bool(torch.equal(below[d:, :2], whole[d:, :2])) and bool(torch.equal(counts[:d], whole[:d]))
    # update sources below the cut are partitioned by source separator: their counts add up too
    ok_src = bool(torch.equal(below[d:, 3], whole[d:, 3]))
    # top levels distributed by column blocks (dist_top = 1): pivot columns, solved elements and update volume of the ranks add up
    # to the undivided lists'; every rank holds the same broadcast sequence
    vol = torch.tensor([plan.level_work_volume(l, rank, world, 1) for l in range(d)], dtype=torch.int64).reshape(d, 6)
    tot = vol[:, :3].clone()
    dist.all_reduce(tot)
    lo, hi = vol[:, 3:].clone(), vol[:, 3:].clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    one = torch.tensor([plan.level_work_volume(l, 0, 1, 0)[:3] for l in range(d)], dtype=torch.int64).reshape(d, 3)
    ok_dist = bool(torch.equal(tot, one)) and bool(torch.equal(lo, hi)) and bool((vol[:, 3] > 0).all())
    if rank == 0:
        q.put((ok_tail, ok_lists, ok_src and ok_dist, tail))
    dist.destroy_process_group()


@pytest.mark.parametrize("case,world", [("lapl_400x400", 2), ("lapl_3375x3375", 2), ("lapl_3375x3375", 8)])
def test_partition_logic_gloo(case, world):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_cpu_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    ok_tail, ok_lists, ok_src, tail = q.get()
    assert ok_tail and ok_lists and ok_src and tail > 0


def _solve_worker(rank, world, port, case, q):
    """One rank of the distributed solve restated with dense numpy blocks of the oracle's L (the vector flow of cholamd_solve_sharded: own subtrees
    forward, sum of the top's part of y, the top redundantly, own subtrees backward, sum of the solution), and its lists against the undivided ones."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cholesky_amd as ca
    from oracle import oracle as orc
    orc.use_own_kernels()
    m, o, c, bfile = case_paths(case)
    plan = ca.Plan(m, o, c)
    O = orc.Oracle(m, o, c)
    O.factor()
    Lf = np.tril(O.dense())
    pb = ca.plan.read_vector(bfile, plan.n)[plan.perm]
    levels, d = plan.levels, world.bit_length() - 1
    tree, off, size = plan.tree, plan.sep_offsets, plan.sep_sizes  # tree[h - 1] = label of heap index h; off / size by label - 1
    rng = lambda s: slice(int(off[s - 1]), int(off[s - 1] + size[s - 1]))
    mine = lambda h: (h.bit_length() - 1) < d or (h >> ((h.bit_length() - 1) - d)) - world == rank
    y = np.zeros(plan.n)
    for h in range(1, plan.nsep + 1):
        if (mine(h) and (h.bit_length() - 1) >= d) or ((h.bit_length() - 1) < d and rank == 0):
            y[rng(tree[h - 1])] = pb[rng(tree[h - 1])]

    def forward(lvls):
        for lv in lvls:
            for h in range(1 << lv, 1 << (lv + 1)):
                if not mine(h):
                    continue
                s = rng(tree[h - 1])
                y[s] = np.linalg.solve(Lf[s, s], y[s])
                a = h // 2
                while a >= 1:
                    ra = rng(tree[a - 1])
                    y[ra] -= Lf[ra, s] @ y[s]
                    a //= 2

    def backward(lvls):
        for lv in lvls:
            for h in range(1 << lv, 1 << (lv + 1)):
                if not mine(h):
                    continue
                s = rng(tree[h - 1])
                a = h // 2
                while a >= 1:
                    ra = rng(tree[a - 1])
                    y[s] -= Lf[ra, s].T @ y[ra]
                    a //= 2
                y[s] = np.linalg.solve(Lf[s, s].T, y[s])

    t0 = int(off[tree[world - 2] - 1]) if world > 1 else plan.n  # the top separators (heap 1 .. world - 1) are the last labels: a contiguous tail
    forward(range(levels - 1, d - 1, -1))
    top = torch.from_numpy(y[t0:].copy())
    dist.all_reduce(top)
    y[t0:] = top.numpy()
    forward(range(d - 1, -1, -1))
    backward(range(0, levels))
    if rank != 0:
        y[t0:] = 0.0
    full = torch.from_numpy(y.copy())
    dist.all_reduce(full)
    xo = O.solve(ca.plan.read_vector(bfile, plan.n))[plan.perm]
    ok_x = bool(np.abs(full.numpy() - xo).max() <= 1e-10 * max(1.0, np.abs(xo).max()))
    cnt = torch.tensor([plan.solve_counts(l, rank, world) for l in range(levels)], dtype=torch.int64)
    whole = torch.tensor([plan.solve_counts(l) for l in range(levels)], dtype=torch.int64)
    below = cnt.clone()
    below[:d] = 0
    dist.all_reduce(below)
    ok_lists = bool(torch.equal(below[d:], whole[d:])) and bool(torch.equal(cnt[:d], whole[:d]))  # shares tile the lists under the cut, the top is whole on every rank
    if rank == 0:
        q.put((ok_x, ok_lists, t0))
    dist.destroy_process_group()


@pytest.mark.parametrize("case,world", [("lapl_400x400", 2), ("lapl_400x400", 8), ("lapl_3375x3375", 2)])
def test_distributed_solve_logic_gloo(case, world):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_solve_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    ok_x, ok_lists, t0 = q.get()
    assert ok_x and ok_lists and t0 > 0


@pytest.mark.parametrize("dims,world", [((20, 20, 20, 4, 32), 2), ((20, 20, 20, 4, 32), 4), ((30, 30, 10, 5, 32), 8)])
def test_distributed_top_lists_tile_the_schedule(dims, world):
    """Generated problems with top separators of several column blocks: under dist_top the ranks' POTRF blocks, TRSM strips and
    update targets of the top levels partition the undivided lists (volumes add up), each rank's broadcast list is the same,
    and what is broadcast is every column block of every top panel exactly once."""
    import cholesky_amd as ca
    plan = ca.Problem(*dims).plan()
    d = world.bit_length() - 1
    sizes, tree = plan.sep_sizes, plan.tree
    for lvl in range(d):
        vols = np.array([plan.level_work_volume(lvl, r, world, 1) for r in range(world)], dtype=np.int64)
        one = np.array(plan.level_work_volume(lvl, 0, 1, 0), dtype=np.int64)
        assert np.array_equal(vols[:, :3].sum(axis=0), one[:3])
        assert (vols[:, 3:] == vols[0, 3:]).all()
        assert one[3] == 0
        rep = np.array(plan.level_work_volume(lvl, world - 1, world, 0), dtype=np.int64)
        assert np.array_equal(rep[:3], one[:3]) and rep[3] == 0  # replicated top: every rank does all of it
        # the owners really differ: no rank holds all the pivot columns of a level with more than one column block
        if vols[0, 3] > 1:
            assert vols[:, 0].max() < one[0]
    # below the cut nothing changes
    for lvl in range(d, plan.levels):
        for r in range(world):
            assert plan.level_work_volume(lvl, r, world, 1) == plan.level_work_volume(lvl, r, world, 0)
    assert sizes[tree[1]] > 144  # the root really has several column blocks


def _gpu_worker(rank, world, port, case, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cholesky_amd as ca
    from cholesky_amd import parallel
    plan = ca.Plan(*case_paths(case)[:3])
    dev = ca.Device(plan, 0)  # every rank on the one GPU of the test box
    dev.set_partition(rank, world)
    tail = parallel.tail_offset(plan, world)
    arena = dev.new_arena()
    dev.fill(arena)
    parallel.factor_sharded(dev, arena, world, tail, via_host=True)
    dev.sync()
    info = dev.info()
    # gather: every rank owns the panels of its subtrees + the (replicated) top
    mine = arena.cpu()
    gathered = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, gathered, dst=0)
    if rank == 0:
        q.put((info, [g.numpy() for g in gathered], tail))
    dist.destroy_process_group()


@pytest.mark.parametrize("case,world", [("lapl_3375x3375", 2), ("lapl_3375x3375", 4), ("lapl_3375x3375", 8), ((24, 24, 24, 5, 32), 8), ((30, 30, 10, 5, 32), 4)])
def test_owner_directed_exchange_volume(case, world):
    """The extend-add exchange under the distributed top levels is OWNER-DIRECTED (SURVEY 5 communication row, 8e
    "Collective") and, since round 4, PATH-AWARE (blas.rg:385-395: a C tile outside the fill is never touched): a rank
    sends each column block it does not own once IF its subtree hangs under the block's separator (A's entries of a block start on its owner:
    cholamd_device_fill), an owner receives one copy from every such rank.  Restated here from the piece list and the tree alone; with
    replicated top levels (dist_top 0) it stays the all-reduce."""
    import cholesky_amd as ca
    plan = ca.Plan(*case_paths(case)[:3]) if isinstance(case, str) else ca.Problem(*case).plan()
    d = world.bit_length() - 1
    vols = [plan.exchange_volume(r, world, 1) for r in range(world)]
    tail = vols[0][2]
    assert all(v[2] == tail and v[3] == vols[0][3] and v[3] > 0 for v in vols)
    pieces = plan.exchange_pieces(world, 1)
    assert len(pieces) == vols[0][3]
    # the pieces are the broadcast lists of the levels above the cut: sizes from the same (checked) lists
    total = sum(plan.level_work_volume(lvl, 0, world, 1)[4] for lvl in range(d))
    assert int(pieces[:, 1].sum()) == total and 0.98 * tail <= total <= tail  # they cover the tail (bar the alignment gaps between panels)

    def under(heap):  # ranks whose subtree root (heap index world + g) has `heap` on its path to the root
        return {g for g in range(world) if any(((world + g) >> k) == heap for k in range(1, d + 1))}

    recv, sent = [0] * world, [0] * world
    for off, count, owner, heap in pieces.tolist():
        assert 1 <= heap < world and 0 <= owner < world
        for g in under(heap):
            if g != owner:
                sent[g] += count
                recv[owner] += count
    assert [v[0] for v in vols] == recv and [v[1] for v in vols] == sent
    dense_sent = [total - int(pieces[pieces[:, 2] == r, 1].sum()) for r in range(world)]  # round 3: every block a rank does not own
    assert all(sent[r] <= dense_sent[r] for r in range(world))
    if world >= 4:  # most ranks are under few of the top separators (the root's panel, which every rank is under, is the largest)
        assert sum(sent) < (0.8 if world == 4 else 0.6) * sum(dense_sent)
    ring = 2 * tail * (world - 1) // world                   # what the all-reduce moves per rank, each way
    assert all(plan.exchange_volume(r, world, 0) == (ring, ring, tail, 0) for r in range(world))
    assert 2 * sum(v[0] for v in vols) <= world * ring * 1.0001  # over all ranks: less than half of what the ring all-reduce moves


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_sharded_factorisation_matches_single_gpu(world):
    import cholesky_amd as ca
    case = "lapl_3375x3375"
    plan = ca.Plan(*case_paths(case)[:3])
    dev = ca.Device(plan, 0)
    arena = dev.new_arena()
    dev.fill(arena)
    dev.factor(arena)
    dev.sync()
    ref = arena.cpu().numpy()

    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    info, parts, tail = q.get()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert info == (0, 0)
    # assemble: panel of separator s comes from its owner; the top from rank 0
    blocks = plan.blocks
    d = world.bit_length() - 1
    tree = plan.tree
    owner = {}
    for h in range(1, plan.nsep + 1):
        lvl = h.bit_length() - 1
        owner[int(tree[h - 1])] = 0 if lvl < d else (h >> (lvl - d)) - (1 << d)
    diag = {int(b[1]): int(b[7]) for b in blocks if b[0] == b[1]}
    order = sorted(diag)
    out = np.zeros_like(ref)
    for i, s in enumerate(order):
        lo = diag[s]
        hi = diag[order[i + 1]] if i + 1 < len(order) else plan.arena_doubles
        out[lo:hi] = parts[owner[s]][lo:hi]
    assert np.abs(out - ref).max() <= 1e-12
    # every rank holds the same top factor
    for r in range(1, world):
        assert np.abs(parts[r][tail:] - parts[0][tail:]).max() <= 1e-12


def _assemble(plan, parts, world, ref_like):
    """Panel of separator s from its owner's arena, the top from rank 0."""
    d = world.bit_length() - 1
    tree = plan.tree
    owner = {}
    for h in range(1, plan.nsep + 1):
        lvl = h.bit_length() - 1
        owner[int(tree[h - 1])] = 0 if lvl < d else (h >> (lvl - d)) - (1 << d)
    diag = {int(b[1]): int(b[7]) for b in plan.blocks if b[0] == b[1]}
    order = sorted(diag)
    out = np.zeros_like(ref_like)
    for i, s in enumerate(order):
        lo = diag[s]
        hi = diag[order[i + 1]] if i + 1 < len(order) else plan.arena_doubles
        out[lo:hi] = parts[owner[s]][lo:hi]
    return out


@pytest.mark.gpu
def test_sharded_world8_in_one_process():
    """BASELINE config 4's partition (8 ranks, tree cut at level 3): the eight rank programs run one after the
    other on the one GPU, the exchange is the device-side sum of the eight tails; result == single-GPU factor."""
    import cholesky_amd as ca
    from cholesky_amd import parallel
    case, world = "lapl_3375x3375", 8
    plan = ca.Plan(*case_paths(case)[:3])
    one = ca.Device(plan, 0)
    ref_t = one.new_arena()
    one.fill(ref_t)
    one.factor(ref_t)
    one.sync()
    ref = ref_t.cpu().numpy()
    d = parallel.split_level(world)
    tail = parallel.tail_offset(plan, world)
    devs, arenas = [], []
    for r in range(world):
        dev = ca.Device(plan, 0)
        dev.set_partition(r, world)
        assert dev.tail_offset() == tail
        a = dev.new_arena()
        dev.fill(a)
        dev.factor_levels(a, plan.levels - 1, d)
        devs.append(dev)
        arenas.append(a)
    torch.cuda.synchronize()
    total = torch.stack([a[tail:] for a in arenas]).sum(dim=0)
    for dev, a in zip(devs, arenas):
        a[tail:] = total
        dev.factor_levels(a, d - 1, 0)
        dev.sync()
        assert dev.info() == (0, 0)
    parts = [a.cpu().numpy() for a in arenas]
    assert np.abs(_assemble(plan, parts, world, ref) - ref).max() <= 1e-12
    for r in range(1, world):
        assert np.array_equal(parts[r][tail:], parts[0][tail:])


@pytest.mark.gpu
@pytest.mark.parametrize("case,world,dist_top,wt", [("lapl_3375x3375", 2, 1, -1), ("lapl_3375x3375", 8, 1, -1), ("lapl_3375x3375", 4, 0, -1),
                                                    ((20, 20, 20, 4, 32), 4, 1, -1), ((24, 24, 24, 5, 32), 8, 1, -1), ((36, 36, 30, 3, 64), 2, 2, -1),
                                                    ((24, 24, 24, 5, 32), 4, 1, 1), ("lapl_3375x3375", 2, 0, 1)])  # wt = 1: every step's strips through k_trsm_wt
def test_factor_multi_distributed_top(case, world, dist_top, wt):
    """cholamd_factor_multi over a LOCAL communicator (the rank objects share the one GPU): subtree levels, ordered device-side sum
    of the tails, then the top levels distributed by column blocks (owner POTRF + TRSM, broadcast, owned updates) -- or
    replicated (dist_top 0) -- against the single-GPU factor; every rank ends with the same complete top."""
    import cholesky_amd as ca
    from cholesky_amd import parallel
    from cholesky_amd.device import factor_multi
    plan = ca.Plan(*case_paths(case)[:3]) if isinstance(case, str) else ca.Problem(*case).plan()
    one = ca.Device(plan, 0)
    ref_t = one.new_arena()
    one.fill(ref_t)
    one.factor(ref_t)
    one.sync()
    assert one.info() == (0, 0)
    ref = ref_t.cpu().numpy()
    del ref_t
    tail = parallel.tail_offset(plan, world)
    devs, arenas = [], []
    for r in range(world):
        dev = ca.Device(plan, 0)
        dev.set_option("dist_top", dist_top)
        if wt >= 0:
            dev.set_option("trsm_wt_min", wt)
        dev.set_partition(r, world)
        a = dev.new_arena()
        dev.fill(a)
        devs.append(dev)
        arenas.append(a)
    if dist_top == 2:
        assert plan.level_work_volume(0, 0, world, 2)[3] > 0  # root >= 1024 columns: distributed automatically
    factor_multi(devs, arenas, local=True)
    for dev in devs:
        assert dev.info() == (0, 0)
    scale = max(1.0, np.abs(ref).max())
    top0 = arenas[0][tail:].cpu().numpy()
    for r in range(1, world):
        assert np.array_equal(arenas[r][tail:].cpu().numpy(), top0)
    parts = [a.cpu().numpy() for a in arenas]
    assert np.abs(_assemble(plan, parts, world, ref) - ref).max() <= 1e-12 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("case,world,dist_top,elem", [((40, 40, 40, 6, 64), 4, 1, 8), ("lapl_3375x3375", 8, 0, 8), ((30, 30, 30, 5, 32), 4, 1, 4)])
def test_rank_arenas_back_only_their_own_panels(case, world, dist_top, elem):
    """Per-rank arenas (cholamd_device_alloc_arena): the address range is complete, memory of its own only under the rank's panels and the shared
    top (the other ranks' panels alias one scratch chunk).  The multi-rank factorisation over them -- local communicator, the rank objects share the
    GPU -- equals the single-GPU factor; on a problem large enough for the 2 MB backing steps the ranks other than 0 take a fraction of the arena."""
    import cholesky_amd as ca
    from cholesky_amd import parallel
    from cholesky_amd.device import factor_multi
    plan = ca.Plan(*case_paths(case)[:3]) if isinstance(case, str) else ca.Problem(*case).plan()
    one = ca.Device(plan, 0)
    if elem == 8:
        ref_t = one.new_arena()
        one.fill(ref_t)
        one.factor(ref_t)
    else:
        ref_t = one.new_arena_f32()
        one.fill_f32(ref_t)
        one.factor_f32(ref_t)
    one.sync()
    assert one.info() == (0, 0)
    ref = ref_t.cpu().numpy()
    del ref_t
    tail = parallel.tail_offset(plan, world)
    devs, arenas = [], []
    for r in range(world):
        dev = ca.Device(plan, 0)
        dev.set_option("dist_top", dist_top)
        dev.set_partition(r, world)
        a = dev.alloc_arena(elem)
        (dev.fill if elem == 8 else dev.fill_f32)(a)
        devs.append(dev)
        arenas.append(a)
    full = plan.arena_doubles * elem
    assert arenas[0].backed_bytes == full  # rank 0 gathers and solves: everything
    if full > (1 << 30):
        assert all(a.backed_bytes < 0.6 * full for a in arenas[1:])
    factor_multi(devs, arenas, local=True)
    for dev in devs:
        assert dev.info() == (0, 0)
    parts = [a.numpy() for a in arenas]
    for r in range(1, world):
        assert np.array_equal(parts[r][tail:], parts[0][tail:])
    scale = max(1.0, np.abs(ref).max())
    tol = 1e-12 if elem == 8 else 2e-5
    assert np.abs(_assemble(plan, parts, world, ref) - ref).max() <= tol * scale
    for a in arenas:
        a.free()


@pytest.mark.gpu
def test_rccl_one_rank_allreduce_and_sharded_entry():
    """libcholamd's RCCL binding runs: unique id, ncclCommInitRank, an in-place ncclAllReduce on the stream, and
    cholamd_factor_sharded (world 1 = the plain level loop) through the communicator-taking entry point."""
    import cholesky_amd as ca
    plan = ca.Plan(*case_paths("lapl_400x400")[:3])
    dev = ca.Device(plan, 0)
    comm = ca.Comm(dev, 1, 0, ca.Comm.unique_id())
    t = torch.arange(1000, dtype=torch.float64, device="cuda")
    comm.allreduce(t)
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float64))
    a, b = dev.new_arena(), dev.new_arena()
    dev.fill(a)
    dev.fill(b)
    dev.factor(a)
    dev.factor_sharded(b, comm)
    dev.sync()
    assert torch.equal(a, b)


def _rccl_worker(rank, world, port, case, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)  # carries the unique id only
    import cholesky_amd as ca
    from cholesky_amd import parallel
    plan = ca.Plan(*case_paths(case)[:3])
    dev = ca.Device(plan, rank)  # one GPU per rank
    dev.set_partition(rank, world)
    comm = parallel.make_comm(dev, world, rank)
    torch.cuda.set_device(rank)
    arena = dev.new_arena()
    dev.fill(arena)
    parallel.factor_sharded(dev, arena, world, dev.tail_offset(), comm=comm)
    dev.sync()
    info = dev.info()
    mine = arena.cpu()
    gathered = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, gathered, dst=0)
    if rank == 0:
        q.put((info, [g.numpy() for g in gathered]))
    del comm
    dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_factorisation_over_rccl_two_gpus():
    """The product path of BASELINE config 4 at world 2: one process per GPU, cholamd_factor_sharded with a real
    RCCL all-reduce between the devices.  Needs two GPUs."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (the round's test box has one)")
    import cholesky_amd as ca
    case, world = "lapl_3375x3375", 2
    plan = ca.Plan(*case_paths(case)[:3])
    dev = ca.Device(plan, 0)
    arena = dev.new_arena()
    dev.fill(arena)
    dev.factor(arena)
    dev.sync()
    ref = arena.cpu().numpy()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    info, parts = q.get()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert info == (0, 0)
    assert np.abs(_assemble(plan, parts, world, ref) - ref).max() <= 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("case,world,mixed", [("lapl_3375x3375", 8, False), ((24, 24, 24, 5, 32), 8, False), ((24, 24, 24, 5, 32), 4, True)])
def test_only_the_ranks_under_a_top_separator_touch_its_blocks(case, world, mixed):
    """What the path-aware exchange (exchange_owned: grouped ncclSend / ncclRecv, not runnable on a one-GPU box) relies on, checked on the device:
    after the subtree levels, BEFORE the exchange, a rank's copy of a column block of the shared top is non-zero only if its subtree hangs under
    the block's separator or it owns the block (whose entries of A its fill scattered); and the copies add up to what the single-GPU run holds
    there at the same point (A_top minus every contribution of the levels under the cut)."""
    import cholesky_amd as ca
    from cholesky_amd import parallel
    plan = ca.Plan(*case_paths(case)[:3]) if isinstance(case, str) else ca.Problem(*case).plan()
    d = world.bit_length() - 1
    L, split = plan.levels, parallel.split_level(world)
    pieces = plan.exchange_pieces(world, 1).tolist()

    def under(heap):
        return {g for g in range(world) if any(((world + g) >> k) == heap for k in range(1, d + 1))}

    def run(dev):
        a = dev.new_arena_f32() if mixed else dev.new_arena()
        (dev.fill_f32 if mixed else dev.fill)(a)
        if mixed:
            dev.factor_levels_f32(a, L - 1, split)
        else:
            dev.factor_levels(a, L - 1, split)
        dev.sync()
        assert dev.info() == (0, 0)
        return a.cpu().numpy().astype(np.float64)

    one = ca.Device(plan, 0)
    one.set_option("dist_top", 1)
    if mixed and not hasattr(one, "factor_levels_f32"):
        pytest.skip("no level-range entry point for the fp32 factor in the Python mirror")
    ref = run(one)
    parts = []
    for r in range(world):
        dev = ca.Device(plan, 0)
        dev.set_option("dist_top", 1)
        dev.set_partition(r, world)
        parts.append(run(dev))
    if mixed:  # the fp32 schedule cuts its own column blocks (128 columns): its piece list is the device's
        pieces = None
    checked = 0
    for off, count, owner, heap in (pieces or []):
        tot = np.zeros(count)
        for r in range(world):
            blk = parts[r][off:off + count]
            if r not in under(heap) and r != owner:
                assert not blk.any(), (r, heap, owner)
            tot += blk
        assert np.abs(tot - ref[off:off + count]).max() <= 1e-12 * max(1.0, np.abs(ref[off:off + count]).max())
        checked += 1
    if not mixed:
        assert checked == len(pieces) > 0
    else:  # without the fp32 piece list: the tails add up to the single-GPU tail
        tail = parallel.tail_offset(plan, world)
        tot = sum(p[tail:] for p in parts)
        assert np.abs(tot - ref[tail:]).max() <= 2e-5 * max(1.0, np.abs(ref[tail:]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("case,world,dist_top", [("lapl_3375x3375", 2, 0), ("lapl_3375x3375", 8, 1), ((20, 20, 20, 4, 32), 4, 1), ((24, 24, 24, 5, 32), 8, 1),
                                                 ((24, 24, 24, 5, 32), 4, 0)])
def test_distributed_solve_over_the_local_communicator(case, world, dist_top):
    """The solve sharded like the factorisation (mmat.rg:1394-1479): every rank sweeps its own subtrees, the top is solved
    redundantly, only two vector sums travel (cholamd_solve_multi over the local communicator -- the rank objects share the one GPU); no rank
    ever holds the other ranks' panels.  Against the single-GPU solve of the same right-hand side and, through the residual, against A."""
    import torch
    import cholesky_amd as ca
    from cholesky_amd.device import factor_multi, solve_multi
    if isinstance(case, str):
        m, o, c, bfile = case_paths(case)
        plan = ca.Plan(m, o, c)
        bvec = ca.plan.read_vector(bfile, plan.n)
    else:
        prob = ca.Problem(*case)
        plan = prob.plan()
        bvec = prob.rhs()
    one = ca.Device(plan, 0)
    ref = one.new_arena()
    one.fill(ref)
    one.factor(ref)
    d_b = torch.from_numpy(bvec).cuda()
    x1 = torch.empty_like(d_b)
    one.solve(ref, d_b, x1)
    one.sync()
    devs, arenas = [], []
    for r in range(world):
        dev = ca.Device(plan, 0)
        dev.set_option("dist_top", dist_top)
        dev.set_partition(r, world)
        a = dev.new_arena()
        dev.fill(a)
        devs.append(dev)
        arenas.append(a)
    factor_multi(devs, arenas, local=True)
    xs = [torch.full_like(d_b, float("nan")) for _ in range(world)]
    solve_multi(devs, arenas, [d_b] * world, xs, local=True)
    xr = x1.cpu().numpy()
    scale = max(1.0, np.abs(xr).max())
    for r in range(world):
        assert np.abs(xs[r].cpu().numpy() - xr).max() <= 1e-10 * scale, r   # every rank ends with the whole solution
    assert one.residual(d_b, xs[world - 1]) <= 1e-10
    # a second right-hand side through the same lists, and the full-tree solve of a partitioned device is still there (gathered factor)
    b2 = torch.from_numpy(np.cos(np.arange(plan.n) * 0.37) + 2.0).cuda()
    solve_multi(devs, arenas, [b2] * world, xs, local=True)
    one.solve(ref, b2, x1)
    one.sync()
    assert np.abs(xs[0].cpu().numpy() - x1.cpu().numpy()).max() <= 1e-10 * max(1.0, float(x1.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("dist_top", [1, 0])
def test_one_process_two_gpus_over_rccl(dist_top):
    """cholamd_factor_multi and cholamd_solve_multi with cholamd_comm_create_all (one process, one GPU per rank, RCCL between them) and the top levels
    distributed: the path in which the owners' sums have to wait for the OUTERMOST RCCL group and the grouped ncclSend / ncclRecv of
    the path-aware exchange run for real.  Needs two GPUs (skipped on the round's one-GPU boxes)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (the round's test box has one)")
    import cholesky_amd as ca
    from cholesky_amd.device import factor_multi, solve_multi
    prob = ca.Problem(24, 24, 24, 5, 32)
    plan = prob.plan()
    one = ca.Device(plan, 0)
    ref = one.new_arena()
    one.fill(ref)
    one.factor(ref)
    d_b = torch.from_numpy(prob.rhs()).cuda(0)
    x1 = torch.empty_like(d_b)
    one.solve(ref, d_b, x1)
    one.sync()
    world = 2
    devs, arenas, bs, xs = [], [], [], []
    for r in range(world):
        dev = ca.Device(plan, r)
        dev.set_option("dist_top", dist_top)
        dev.set_partition(r, world)
        with torch.cuda.device(r):
            a = dev.new_arena()
            dev.fill(a)
            bs.append(d_b.to(f"cuda:{r}"))
            xs.append(torch.empty_like(bs[-1]))
        devs.append(dev)
        arenas.append(a)
    factor_multi(devs, arenas, local=False)
    for dev in devs:
        assert dev.info() == (0, 0)
    refh = ref.cpu().numpy()
    parts = [a.cpu().numpy() for a in arenas]
    assert np.abs(_assemble(plan, parts, world, refh) - refh).max() <= 1e-12 * max(1.0, np.abs(refh).max())
    solve_multi(devs, arenas, bs, xs, local=False)
    xr = x1.cpu().numpy()
    for r in range(world):
        assert np.abs(xs[r].cpu().numpy() - xr).max() <= 1e-10 * max(1.0, np.abs(xr).max())
