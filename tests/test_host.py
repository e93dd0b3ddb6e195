"""CPU-only tests of the product's host side: ingest, symbolic analysis and schedule of
libcholamd.so against the oracle (which is pinned to the reference in test_oracle.py), plus the
C-ABI surface check (every symbol include/cholamd.h declares is exported; no compute calls)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import CASES, KNOWN, ROOT, case_paths
from oracle import oracle as orc


@pytest.fixture(scope="module")
def ca():
    import __graft_entry__
    if not os.path.exists(os.path.join(ROOT, "cholesky_amd", "lib", "libcholamd.so")):
        __graft_entry__.build()
    import cholesky_amd
    return cholesky_amd


@pytest.fixture(scope="module")
def plans(ca):
    return {case: ca.Plan(*case_paths(case)[:3]) for case in CASES}


@pytest.fixture(scope="module")
def oracles():
    orc.use_own_kernels()
    out = {}
    for case in CASES:
        O = orc.Oracle(*case_paths(case)[:3])
        O.factor(log_ops=True)
        out[case] = O
    return out


def test_header_symbols_are_exported(ca):
    """Every function declared in include/cholamd.h is exported by libcholamd.so."""
    from cholesky_amd import _lib
    text = open(_lib.HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b((?:cholamd|mm)_[A-Za-z0-9_]+)\s*\(", text))
    assert len(names) > 60
    lib = C.CDLL(_lib.LIB_PATH)
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def test_no_device_means_loud_failure(ca):
    """Without a GPU the compute entry points fail with an error code; nothing computes on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    a = np.asfortranarray(np.eye(4) * 4.0)
    with pytest.raises(ca.CholamdError):
        ca.blas.LAPACKE_dpotrf(a)
    assert a[0, 0] == 4.0  # untouched
    plan = ca.Plan(*case_paths("lapl_9x9")[:3])
    with pytest.raises(ca.CholamdError):
        ca.Device(plan, 0)


@pytest.mark.parametrize("case", list(CASES))
def test_plan_matches_oracle(case, plans, oracles):
    P, O = plans[case], oracles[case]
    assert (P.n, P.nz, P.levels, P.nsep) == (O.N, O.NZ, O.levels, O.nsep)
    assert P.banner == O.banner
    assert np.array_equal(P.perm, O.perm)
    assert np.array_equal(P.sep_sizes, O.sep_sizes)
    assert np.array_equal(P.sep_offsets, O.sep_offsets)
    assert np.array_equal(P.tree, O.tree)
    assert np.array_equal(P.blocks[:, :6], O.blocks)
    assert P.max_int_size == O.L.orc_max_int_size(O.h)
    for lbl in range(P.levels):
        assert np.array_equal(P.snapshot_array(lbl), O.snapshot(lbl)), f"snapshot {lbl}"
    assert np.array_equal(P.ops(), O.ops())  # same BLAS calls, same program order


@pytest.mark.parametrize("case", list(CASES))
def test_known_answers(case, plans):
    P = plans[case]
    calls, flops = P.counts()
    assert tuple(int(c) for c in calls) == KNOWN[case][0]
    assert flops.sum() == pytest.approx(KNOWN[case][1], rel=1e-5)
    assert P.flops == pytest.approx(KNOWN[case][1], rel=1e-5)
    assert P.nnz_a == KNOWN[case][2]
    assert P.nnz_l == KNOWN[case][3]
    assert P.alg_bytes == 8 * (KNOWN[case][2] + KNOWN[case][3])
    assert P.dropped == 0
    assert P.nnz_tiles >= P.nnz_l


def test_fmin_known_answers(plans):
    want = {"lapl_9x9": 96, "lapl_25x25": 603, "lapl_400x400": 7.792e4, "lapl_3375x3375": 5.273e7}
    for case, v in want.items():
        assert plans[case].fmin == pytest.approx(v, rel=1e-3)


@pytest.mark.parametrize("case", list(CASES))
def test_host_fill_is_the_reference_permuted_matrix(case, plans, golden):
    P = plans[case]
    assert np.array_equal(P.arena_to_dense(P.fill_host()), golden(case)["pmat"])


def test_panel_layout(plans):
    """One contiguous panel per separator; the top of the tree is a contiguous tail of the arena."""
    P = plans["lapl_3375x3375"]
    b = P.blocks
    ns = P.nsep
    # diagonal blocks start their panels; panels ordered by label
    diag = b[b[:, 0] == b[:, 1]]
    assert np.all(np.diff(diag[:, 7]) > 0)
    # block (r, c) lies in panel c: same ld as (c, c) and offset between panel start and next panel
    for row in b:
        r, c = int(row[0]), int(row[1])
        d = diag[diag[:, 1] == c][0]
        assert row[6] == d[6] and row[7] >= d[7]
        tm = P.block_tile_map(r, c)
        rows = int(row[4] - row[2] + 1)
        stored = sum(min(16, rows - 16 * t) for t in range(len(tm)) if tm[t] >= 0)
        kept = tm[tm >= 0]
        assert np.array_equal(kept, np.arange(len(kept)))  # kept tiles keep their order, no holes in storage
        if r == c or P.heap_of(r) == P.heap_of(c) // 2:
            assert stored == rows  # the separator's own rows and its parent's are stored in full
        if c < ns and stored > 0:
            nxt = diag[diag[:, 1] == c + 1][0]
            assert row[7] + (row[5] - row[3]) * row[6] + (stored - 1) < nxt[7]
    assert P.arena_dense_doubles * 8 < 16e6  # 13.9 MB of allocated blocks + alignment
    assert P.arena_doubles * 8 < 9.0e6       # row compaction: 8.9 MB


@pytest.mark.parametrize("case", list(CASES))
def test_row_compaction_keeps_every_filled_tile(case, plans):
    """Every filled tile of every snapshot (the only rows the reference's tasks touch, blas.rg:385-395) lies in stored rows that are
    consecutive in storage; the arena never exceeds the uncompacted one."""
    P = plans[case]
    assert P.arena_doubles <= P.arena_dense_doubles
    off = P.sep_offsets
    maps = {}
    for lbl in range(P.levels):
        for sx, sy, _, lo_x, _, hi_x, _ in P.snapshot_array(lbl):
            tm = maps.setdefault((sx, sy), P.block_tile_map(int(sx), int(sy)))
            t0, t1 = (lo_x - off[sx - 1]) // 16, (hi_x - off[sx - 1]) // 16
            assert tm[t0] >= 0 and np.array_equal(tm[t0:t1 + 1], tm[t0] + np.arange(t1 - t0 + 1))


def test_writers_roundtrip(tmp_path, ca, plans, golden):
    import scipy.io
    P = plans["lapl_25x25"]
    arena = P.fill_host()
    p = tmp_path / "permuted.mtx"
    P.write_matrix(arena, str(p))
    first = open(p).readline().strip()
    assert first == "%%MatrixMarket matrix coordinate real hermitian"
    M = scipy.io.mmread(str(p)).toarray()
    assert np.array_equal(np.tril(M), golden("lapl_25x25")["pmat"])
    x = np.linspace(-1, 1, 25)
    s = tmp_path / "sol.txt"
    ca.plan.write_solution(str(s), x)
    assert np.allclose(np.genfromtxt(str(s)), x, atol=1e-7)
    ca.plan.write_solution(str(s), x, full_precision=True)
    assert np.array_equal(np.genfromtxt(str(s)), x)


def test_debug_log_is_replayable(tmp_path, plans):
    """The -d structured op log has the reference's dict-literal line format (blas.rg:308,340,405):
    every line after the tag evaluates as a Python dict with the keys verify.debug_factor reads."""
    P = plans["lapl_25x25"]
    p = tmp_path / "oplog.txt"
    P.write_debug_log(str(p))
    tags = {"Block": 0, "POTRF": 0, "TRSM": 0, "GEMM": 0}
    for line in open(p):
        tag, rest = line.split(":", 1)
        d = eval(rest)  # noqa: S307 - same consumption as verify.py:26-29
        tags[tag] += 1
        if tag != "Block":
            assert {"A", "A_Lo", "A_Hi", "Block", "Level", "Interval"} <= set(d)
    assert tags["POTRF"] == 7 and tags["TRSM"] == 16 and tags["GEMM"] == 30 and tags["Block"] == P.num_blocks


def test_ingest_errors_are_reported(tmp_path, ca):
    m, o, c, _ = case_paths("lapl_9x9")
    with pytest.raises(ca.CholamdError):
        ca.Plan(str(tmp_path / "missing.mtx"), o, c)
    bad = tmp_path / "bad_ord.txt"
    bad.write_text("2 3\n0;0,3,6,\n1;2,5,8,\n")  # third separator missing
    with pytest.raises(ca.CholamdError):
        ca.Plan(m, str(bad), c)
    badc = tmp_path / "bad_clust.txt"
    badc.write_text("2 3\n0;0,3,;\n1;0,3,;\n2;0,1,3,;\n")  # root not a single tile when eliminated
    with pytest.raises(ca.CholamdError):
        ca.Plan(m, o, str(badc))


def test_mm_reader_matches_reference_build(ca):
    """The library's mm_read_banner / mm_read_mtx_crd_size agree with the reference's own mmio.c
    (compiled where it lies into oracle/_ref) on every fixture."""
    so = os.path.join(ROOT, "oracle", "_ref", "libmmio_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built (reference tree absent)")
    from cholesky_amd import _lib
    ref, mine, libc = C.CDLL(so), C.CDLL(_lib.LIB_PATH), C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    for lib in (ref, mine):
        lib.mm_read_banner.argtypes = [C.c_void_p, C.c_char_p]
        lib.mm_read_mtx_crd_size.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 3
    for case in CASES:
        for path in (case_paths(case)[0], case_paths(case)[3]):
            res = []
            for lib in (ref, mine):
                fp = libc.fopen(path.encode(), b"r")
                tc = C.create_string_buffer(4)
                rc = lib.mm_read_banner(fp, tc)
                M, N, NZ = C.c_int(), C.c_int(), C.c_int()
                rc2 = lib.mm_read_mtx_crd_size(fp, C.byref(M), C.byref(N), C.byref(NZ)) if tc.raw[1:2] == b"C" else 0
                libc.fclose(fp)
                res.append((rc, tc.raw, rc2, M.value, N.value, NZ.value))
            assert res[0] == res[1], path


def test_partition_work_lists(ca, plans):
    """Multi-GPU subtree sharding: every separator below the cut is owned by exactly one
    rank and the shared top is the contiguous tail of the arena."""
    P = plans["lapl_3375x3375"]
    tree = P.tree
    for world in (1, 2, 4, 8):
        d = world.bit_length() - 1
        owners = {}
        for h in range(1, P.nsep + 1):
            lvl = h.bit_length() - 1
            owners[int(tree[h - 1])] = -1 if lvl < d else (h >> (lvl - d)) - (1 << d)
        for rank in range(world):
            mine = [s for s, o in owners.items() if o == rank]
            assert len(mine) == (P.nsep - (world - 1)) // world
        top = sorted(s for s, o in owners.items() if o == -1)
        assert top == list(range(P.nsep - (world - 1) + 1, P.nsep + 1))


def test_bench_gpus_flag_never_mislabels(tmp_path):
    """bench.py --gpus N (VERDICT r1 / ADVICE): a world size that differs from --gpus is an error, and without a launcher the
    ranks are started as children before anything touches the GPU -- on this GPU-less box they must all fail loudly."""
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, bench, "--gpus", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing to report a mislabelled run" in r.stderr and "{" not in r.stdout
    r = subprocess.run([sys.executable, bench, "--gpus", "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "power of two" in r.stderr


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("follow", [1, 0])
def test_program_launch_is_live_and_covers_the_level_lists(case, follow):
    """cholamd_factor() runs small problems as ONE launch of resident workgroups drawing jobs from a queue (k_program).  Host-side
    check of that program: with as few as 4 resident workgroups and every counter raised only when its job has completed (stricter
    than the device), no job starves; counters total up; pivot blocks and TRSM rows are those of the per-level lists."""
    import cholesky_amd as ca
    m, o, c, _ = case_paths(case)
    P = ca.Plan(m, o, c)
    for workers in (256, 16, 4):
        P.program_check(follow, workers)
    cnt = P.program_counts(follow)
    assert cnt["jobs"] > 0 and (cnt["followers"] > 0) == bool(follow)


def test_program_launch_is_live_under_follower_tails_and_splits():
    """The followers' early update jobs (option follow_tail) and the item lists of every pivot split keep the queue live: tails
    0 (followers take everything) ... 6, pivots in 64-column blocks (chains of followers) and in 192-column blocks."""
    import cholesky_amd as ca
    m, o, c, _ = case_paths("lapl_3375x3375")
    P = ca.Plan(m, o, c)
    for tail in (0, 1, 2, 4, 6):
        for workers in (256, 8):
            P.program_check_opts(follow_tail=tail, workers=workers)
    P.program_check_opts(follow_tail=1, split_min=64, split_nb=64, workers=16)
    P.program_check_opts(follow_tail=3, split_min=96, split_nb=96, workers=16)
    P.program_check_opts(split_min=192, split_nb=192, workers=16)
    # the leaf skylines off (leaves split like every other pivot); unsplit banded leaves handed to the extend-add in chunks of
    # column tiles (every strip of such a leaf publishes its column tiles): the switches come from the environment
    for env in ({"CHOLAMD_NO_SKYLINE": "1"}, {"CHOLAMD_STAGE_CHUNK": "4"}, {"CHOLAMD_STAGE_CHUNK": "2"}):
        os.environ.update(env)
        try:
            P.program_check_opts(workers=16)
            P.program_check_opts(follow_tail=1, workers=64)
        finally:
            for k in env:
                del os.environ[k]
    G = ca.Problem(14, 14, 14, 4, 16).plan()
    for tail in (0, 2, 4):
        G.program_check_opts(follow_tail=tail, workers=16)


def test_program_launch_on_generated_problems():
    import cholesky_amd as ca
    for dims in [(7, 5, 3, 3, 4), (12, 12, 12, 4, 16), (10, 9, 8, 5, 8)]:
        P = ca.Problem(*dims).plan()
        P.program_check(1, 32)
        P.program_check(0, 32)
    # a pivot block wider than the fused roles take does not qualify: the level-by-level launches serve it
    big = ca.Problem(24, 24, 12, 2, 64).plan()
    with pytest.raises(ca.CholamdError):
        big.program_check(1, 256)


def test_follower_round_grouping_is_a_function_of_the_list_alone():
    """How a following POTRF job groups its followed column tiles into rounds (one or two per round) is computed by every wave of the
    workgroup for itself; the waves count barriers by it, so it must depend on nothing but the list (its length) and the block's
    tile columns -- a grouping by what happened to be published once dead-locked.  The kernel takes the grouping from
    chol_follow_round / chol_follow_own_at (chol_plan.h); the same functions are checked here for every follower of every fixture's
    program and exhaustively for small cases."""
    import ctypes as C
    from cholesky_amd import _lib
    L = _lib.load()

    def rounds(n_ext, T):
        buf = (C.c_int * 64)()
        own = C.c_int(-1)
        n = L.cholamd_follow_rounds(n_ext, T, 64, buf, C.byref(own))
        return [buf[i] for i in range(min(n, 64))], own.value

    for T in range(1, 11):
        for n_ext in range(1, 41):
            r, own = rounds(n_ext, T)
            assert (r, own) == rounds(n_ext, T)                                   # a pure function: the same answer every time
            assert sum(r) == n_ext and all(x in (1, 2) for x in r)                 # the rounds tile the list, in order
            assert all(x == 1 for x in r) or 2 * T <= 17                           # two column tiles share an LDS buffer of 17 tiles
            starts = [sum(r[:i]) for i in range(len(r))]
            assert own in starts and own <= max(n_ext - 2, 0)                      # the own tiles go in in front of a round, never inside a pair
            assert own == max(s0 for s0 in starts if s0 <= max(n_ext - 2, 0))      # ... the last such round
    import cholesky_amd as ca
    for case in ("lapl_25x25", "lapl_400x400", "lapl_3375x3375"):
        plan = ca.Plan(*case_paths(case)[:3])
        buf = (C.c_int * 4096)()
        nf = L.cholamd_plan_program_followers(plan.h, 4096, buf)
        assert nf > 0
        for i in range(nf):
            job, n_ext, T, n_wait = buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]
            r, own = rounds(n_ext, T)
            assert 1 <= T <= 10 and n_ext >= 1 and sum(r) == n_ext, (case, job)


@pytest.mark.parametrize("dims", [(20, 20, 20, 4, 16), (24, 24, 12, 3, 32), (30, 30, 10, 5, 32)])
def test_merged_targets_cover_the_same_work_with_fewer_tasks(dims, ca):
    """Level schedule, option merge_targets: extend-add targets / panel row runs that are neighbours in storage become one target / run.
    The pivots, the solved elements and the update volume (target elements x source depth) do not change, the lists get shorter."""
    plan = ca.Problem(*dims).plan()
    shorter = 0
    for lvl in range(plan.levels):
        a = plan.level_work_volume_opts(lvl, merge_targets=0, mt_min_tiles=1)
        b = plan.level_work_volume_opts(lvl, merge_targets=1, mt_min_tiles=1)
        assert a[:3] == b[:3]                      # POTRF columns, TRSM elements, update volume
        assert b[6] + b[7] <= a[6] + a[7] and b[8] <= a[8]   # update tasks (16x16 + macro tiles: a merged target may move from one list to the other), strips
        shorter += (a[6] + a[7] - b[6] - b[7]) + (a[8] - b[8])
    assert shorter > 0


def test_macro_tile_fill_statistics(ca, monkeypatch):
    """cholamd_plan_level_mt_fill: how full the 64 x 64 macro tiles of a level's update lists are (the statistic behind option merge_targets:
    a generated problem with 32-row cluster tiles is half empty without the merging)."""
    monkeypatch.setenv("CHOLAMD_NO_LEAF_ENVELOPE", "1")  # (with the leaves' structural zeros left out this problem's phases stay under the macro-tile threshold)
    plan = ca.Problem(30, 30, 30, 4, 32).plan()
    tasks = full = valid = total = 0
    for lvl in range(plan.levels):
        t, f, v, w = plan.level_mt_fill(lvl)
        assert 0 <= f <= t and 0 <= v <= w
        tasks, full, valid, total = tasks + t, full + f, valid + v, total + w
    assert tasks > 0 and valid / total > 0.6


def _role_table_reference(n, sky):
    """Independent restatement of the POTRF role's deal and step masks (chol_kernels.hip, potrf_rr_body): tiles of columns >= 2 of the T x T
    lower tile grid in reverse column-major order to 9 heavy + 2 light waves; per (step, wave) the slots with work, the panel tiles to solve."""
    NW, NH, SLOTS, MAXT = 11, 9, 12, 17
    T = (n + 15) // 16
    tiles = [(i, j) for j in range(T) for i in range(j, T)]
    ntl, ntl2 = len(tiles), (T - 2) * (T - 1) // 2
    mask = np.zeros((MAXT, NW), dtype=np.int32)
    ij = np.full((SLOTS, NW), 0xFFFF, dtype=np.uint16)
    km = np.zeros((SLOTS, NW), dtype=np.uint8)

    def owner(idx):
        if ntl2 <= 36:
            off = idx % NH
            return off + off // 3, idx // NH
        cyc, pos = divmod(idx, 51)
        starts = [0, 11, 20, 31, 40, 51]
        rnd = max(r for r in range(5) if pos >= starts[r])
        off = pos - starts[rnd]
        if off < NH:
            return off + off // 3, cyc * 5 + rnd
        return (3 if off == NH else 7), cyc * 3 + rnd // 2

    for t in range(2 * T - 1, ntl):
        ti, tj = tiles[t]
        w, s = owner(ntl - 1 - t)
        ij[s, w] = ti | (tj << 8)
        kmin = max(sky[min(ti, 23)], sky[min(tj, 23)])
        km[s, w] = kmin | (0x80 if tj < sky[min(ti, 23)] else 0)  # bit 7: left of the skyline, zero in A
        last = tj - 2 if ti == tj else tj - 1
        for k in range(min(kmin, last), last + 1):
            mask[k, w] |= 1 << s
    for k in range(T):
        for i in range(k + 2, T):
            if sky[min(i, 23)] <= k:
                hv = i % NH
                mask[k, hv + hv // 3] |= 1 << (12 if i < k + 2 + NH else 13)
    return mask, ij, km


@pytest.mark.parametrize("n,band", [(16, 0), (33, 0), (98, 0), (144, 0), (160, 0), (176, 0), (259, 3), (272, 2), (272, 0)])
def test_potrf_role_table(n, band):
    """The table the schedule ships behind a POTRF descriptor (chol_potrf_table) against the restatement above."""
    from cholesky_amd import _lib
    L = C.CDLL(_lib.LIB_PATH)
    sky = np.zeros(24, dtype=np.uint8)
    if band:
        sky[:] = [max(0, i - band) for i in range(24)]
    out = np.zeros(2048, dtype=np.uint8)
    L.chol_potrf_table.restype = None
    L.chol_potrf_table(C.c_int(n), sky.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    mask, ij, km = _role_table_reference(n, [int(v) for v in sky])
    got_mask = out[:17 * 11 * 4].view(np.int32).reshape(17, 11)
    got_ij = out[17 * 11 * 4:17 * 11 * 4 + 12 * 11 * 2].view(np.uint16).reshape(12, 11)
    got_km = out[17 * 11 * 4 + 12 * 11 * 2:17 * 11 * 4 + 12 * 11 * 3].reshape(12, 11)
    assert np.array_equal(got_ij, ij) and np.array_equal(got_km, km) and np.array_equal(got_mask, mask)
    # every tile of columns >= 2 has exactly one (slot, wave)
    T = (n + 15) // 16
    owned = sorted(int(v) for v in got_ij.ravel() if v != 0xFFFF)
    assert owned == sorted(i | (j << 8) for j in range(2, T) for i in range(j, T))


@pytest.mark.parametrize("case", ["lapl_400x400", "lapl_3375x3375"])
def test_what_the_solve_skips_is_zero_in_the_reference_factor(case, plans, oracles):
    """The solve does not read a leaf's rows beyond its band under a span, nor a leaf panel row in front of its first entry of A
    (cholamd_plan_solve_skips).  In the oracle's factor (the reference's algorithm on dense blocks) every such entry is exactly zero, and
    above the leaves nothing is skipped."""
    P, O = plans[case], oracles[case]
    Ld = np.tril(O.dense())
    skipped = 0
    for level in range(P.levels):
        seps, runs = P.solve_skips(level)
        if level < P.levels - 1:
            assert not seps[:, 2].any() and not runs[:, 4].any()
            continue
        for off, n, band in seps:
            if band > 0:
                D = Ld[off:off + n, off:off + n]
                i, j = np.indices(D.shape)
                assert not D[i - j > band].any()
                skipped += int((i - j > band).sum())
        for x_off, m, y_off, n, c_lo in runs:
            assert 0 <= c_lo <= n and c_lo % 16 == 0
            assert not Ld[x_off:x_off + m, y_off:y_off + c_lo].any()
            skipped += m * c_lo
    assert skipped > 0


def test_leaf_envelope_shrinks_the_leaf_level_lists_only(ca):
    """Level schedule, option leaf_envelope (default on): the leaves' TRSM strips and update tasks leave out the structural zeros -- less solved
    elements and less update volume at the leaf level than the dense lists (cholamd_plan_level_work_volume_opts builds those), the same pivots, and
    nothing changes above the leaves."""
    plan = ca.Problem(24, 24, 24, 4, 32).plan()
    for lvl in range(plan.levels):
        on = plan.level_work_volume(lvl)                          # default options
        off = plan.level_work_volume_opts(lvl, merge_targets=1)   # the same with leaf_envelope = 0
        assert on[0] == off[0]                                    # POTRF columns
        if lvl == plan.levels - 1:
            assert on[1] < 0.8 * off[1] and on[2] < 0.8 * off[2], (on, off)
        else:
            assert tuple(on[:3]) == tuple(off[:3])
