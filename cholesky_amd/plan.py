"""Host-side plan: ingest + symbolic analysis (the part of mmat.rg's main before the level loop,
mmat.rg:1097-1209), computed by the C library."""
import ctypes as C
import os

import numpy as np

from ._lib import Filled, Op, check, load

OP_NAMES = ("POTRF", "TRSM", "SYRK", "GEMM")


class Plan:
    def __init__(self, matrix_file=None, separator_file=None, clusters_file=None, _handle=None):
        self.L = load()
        if _handle is not None:
            self.h = _handle
        else:
            h = C.c_void_p()
            check(self.L.cholamd_plan_create(os.fsencode(matrix_file), os.fsencode(separator_file), os.fsencode(clusters_file), C.byref(h)),
                  "cholamd_plan_create")
            self.h = h
        g = self.L
        self.n = g.cholamd_plan_n(self.h)
        self.nz = g.cholamd_plan_nz(self.h)
        self.levels = g.cholamd_plan_levels(self.h)
        self.nsep = g.cholamd_plan_num_separators(self.h)
        self.num_blocks = g.cholamd_plan_num_blocks(self.h)
        self.arena_doubles = g.cholamd_plan_arena_doubles(self.h)
        self.flops = g.cholamd_plan_flops(self.h)          # F_ref
        self.alg_bytes = g.cholamd_plan_alg_bytes(self.h)  # B_alg
        self.nnz_a = g.cholamd_plan_nnz_a(self.h)
        self.nnz_l = g.cholamd_plan_nnz_l(self.h)
        self.nnz_tiles = g.cholamd_plan_nnz_tiles(self.h)
        self.fmin = g.cholamd_plan_fmin(self.h)
        self.dropped = g.cholamd_plan_dropped_entries(self.h)
        self.max_int_size = g.cholamd_plan_max_int_size(self.h)

    @classmethod
    def from_arrays(cls, n, levels, perm, sep_sizes, cl_idx, cl_interval, cl_sep, a_row, a_col, a_val, banner=None):
        L = load()
        arr = lambda a, t: np.ascontiguousarray(a, dtype=t)  # noqa: E731
        perm, sep_sizes = arr(perm, np.int32), arr(sep_sizes, np.int32)
        cl_idx, cl_interval, cl_sep = arr(cl_idx, np.int32), arr(cl_interval, np.int32), arr(cl_sep, np.int32)
        a_row, a_col, a_val = arr(a_row, np.int32), arr(a_col, np.int32), arr(a_val, np.float64)
        h = C.c_void_p()
        check(L.cholamd_plan_create_from_arrays(n, levels, perm.ctypes.data, sep_sizes.ctypes.data, cl_idx.ctypes.data,
                                                cl_interval.ctypes.data, cl_sep.ctypes.data, len(cl_idx), len(a_val),
                                                a_row.ctypes.data, a_col.ctypes.data, a_val.ctypes.data,
                                                banner.encode() if banner else None, C.byref(h)), "cholamd_plan_create_from_arrays")
        return cls(_handle=h)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.cholamd_plan_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _ints(self, fn, n):
        a = np.zeros(n, dtype=np.int32)
        getattr(self.L, fn)(self.h, a.ctypes.data)
        return a

    @property
    def banner(self):
        return self.L.cholamd_plan_banner(self.h).decode()

    @property
    def perm(self):
        return self._ints("cholamd_plan_perm", self.n)

    @property
    def sep_sizes(self):
        return self._ints("cholamd_plan_sep_sizes", self.nsep)

    @property
    def sep_offsets(self):
        return self._ints("cholamd_plan_sep_offsets", self.nsep)

    @property
    def tree(self):
        return self._ints("cholamd_plan_tree", self.nsep)

    @property
    def blocks(self):
        """rows: r, c, lo_x, lo_y, hi_x, hi_y, ld, arena offset (int64)"""
        raw = self._ints("cholamd_plan_blocks", 9 * self.num_blocks).reshape(-1, 9).astype(np.int64)
        off = (raw[:, 7] & 0xFFFFFFFF) | (raw[:, 8] << 32)
        return np.concatenate([raw[:, :7], off[:, None]], axis=1)

    def heap_of(self, label):
        """Heap index (root = 1, children of h = 2 h, 2 h + 1) of a separator label."""
        return int(np.nonzero(self.tree == label)[0][0]) + 1

    @property
    def arena_dense_doubles(self):
        """What the arena would hold with every ancestor row stored (no row compaction)."""
        return int(self.L.cholamd_plan_arena_dense_doubles(self.h))

    def block_tile_map(self, r, c):
        """Stored position of every 16-row tile of block (r, c) in its panel (-1: the tile has no storage)."""
        rows = int(self.sep_sizes[r - 1])
        out = np.zeros((rows + 15) // 16 + 1, dtype=np.int32)
        n = self.L.cholamd_plan_block_tile_map(self.h, r, c, out.ctypes.data)
        if n < 0:
            raise ValueError(f"no block ({r}, {c})")
        return out[:n]

    def snapshot(self, interval_lbl):
        n = self.L.cholamd_plan_snapshot_count(self.h, interval_lbl)
        buf = (Filled * max(n, 1))()
        self.L.cholamd_plan_snapshot(self.h, interval_lbl, buf)
        return buf, n

    def snapshot_array(self, interval_lbl):
        buf, n = self.snapshot(interval_lbl)
        a = np.frombuffer(buf, dtype=np.int32).reshape(-1, 9)[:n]
        return a[:, [1, 2, 4, 5, 6, 7, 8]].copy()  # sep_x, sep_y, cluster, lo_x, lo_y, hi_x, hi_y

    def ops(self):
        n = self.L.cholamd_plan_num_ops(self.h)
        buf = (Op * max(n, 1))()
        self.L.cholamd_plan_ops(self.h, buf)
        return np.frombuffer(buf, dtype=np.int32).reshape(-1, 14)[:n].copy()

    def counts(self, level=-1):
        c = np.zeros(4, dtype=np.int64)
        f = np.zeros(4, dtype=np.float64)
        self.L.cholamd_plan_counts(self.h, level, c.ctypes.data, f.ctypes.data)
        return c, f

    def fill_host(self):
        """fill_block for every block: the arena holding P A P^T (host array)."""
        a = np.zeros(self.arena_doubles, dtype=np.float64)
        check(self.L.cholamd_plan_fill_host(self.h, a.ctypes.data), "cholamd_plan_fill_host")
        return a

    def fill_host_part(self, rank, world):
        """Host arena as rank `rank` of `world` starts from (A's shared-top entries on rank 0 only) and
        the offset of the shared tail of the arena."""
        a = np.zeros(self.arena_doubles, dtype=np.float64)
        tail = C.c_int64(0)
        check(self.L.cholamd_plan_fill_host_part(self.h, a.ctypes.data, rank, world, C.byref(tail)), "cholamd_plan_fill_host_part")
        return a, tail.value

    def level_work_counts(self, level, rank=0, world=1):
        out = np.zeros(4, dtype=np.int32)
        check(self.L.cholamd_plan_level_work_counts(self.h, level, rank, world, out.ctypes.data), "cholamd_plan_level_work_counts")
        return tuple(int(v) for v in out)

    def exchange_volume(self, rank, world, dist_top=2):
        """(received, sent, tail, pieces) in arena elements of the extend-add exchange of (rank, world): cholamd_plan_exchange_volume."""
        out = np.zeros(4, dtype=np.int64)
        check(self.L.cholamd_plan_exchange_volume(self.h, rank, world, dist_top, out.ctypes.data), "cholamd_plan_exchange_volume")
        return tuple(int(v) for v in out)

    def solve_counts(self, level, rank=0, world=1):
        """(separators, row runs, forward chunks, backward chunks, columns solved) of one rank's solve lists of a level (cholamd_plan_solve_counts)."""
        out = np.zeros(5, dtype=np.int64)
        check(self.L.cholamd_plan_solve_counts(self.h, level, rank, world, out.ctypes.data), "cholamd_plan_solve_counts")
        return tuple(int(v) for v in out)

    def solve_skips(self, level):
        """(seps, runs) of cholamd_plan_solve_skips: rows (first position, columns, band) per separator and (first row position, rows, first column
        position, columns, c_lo) per (ancestor, separator) row run of the level's solve lists."""
        cnt = self.solve_counts(level)
        seps = np.zeros((max(cnt[0], 1), 3), dtype=np.int32)
        runs = np.zeros((max(cnt[1], 1), 5), dtype=np.int32)
        check(self.L.cholamd_plan_solve_skips(self.h, level, seps.ctypes.data, runs.ctypes.data), "cholamd_plan_solve_skips")
        return seps[:cnt[0]], runs[:cnt[1]]

    def exchange_pieces(self, world, dist_top=2):
        """The column-block pieces of that exchange as rows (arena offset, elements, owner rank, heap index of the top separator)."""
        out = np.zeros((4096, 4), dtype=np.int64)
        n = self.L.cholamd_plan_exchange_pieces(self.h, world, dist_top, len(out), out.ctypes.data)
        if n < 0:
            check(n, "cholamd_plan_exchange_pieces")
        return out[:min(n, len(out))]

    def level_work_volume(self, level, rank=0, world=1, dist_top=2):
        """(POTRF columns, TRSM elements, update volume, broadcast entries, broadcast doubles, broadcast checksum) of one level's
        lists for (rank, world) with the top levels replicated (dist_top=0), distributed by column blocks (1) or automatic (2)."""
        out = np.zeros(6, dtype=np.int64)
        check(self.L.cholamd_plan_level_work_volume(self.h, level, rank, world, dist_top, out.ctypes.data), "cholamd_plan_level_work_volume")
        return tuple(int(v) for v in out)

    def level_work_volume_opts(self, level, merge_targets=1, mt_min_tiles=-1):
        """level_work_volume (single GPU) under option merge_targets / mt_min_tiles + (16x16 tasks, macro-tile tasks, TRSM strips)."""
        out = np.zeros(9, dtype=np.int64)
        check(self.L.cholamd_plan_level_work_volume_opts(self.h, level, int(merge_targets), int(mt_min_tiles), out.ctypes.data), "cholamd_plan_level_work_volume_opts")
        return tuple(int(v) for v in out)

    def level_mt_fill(self, level):
        """(macro-tile tasks, tasks with all 64 x 64 elements valid, valid elements x depth, tile elements x depth) of one level's lists."""
        out = np.zeros(4, dtype=np.int64)
        check(self.L.cholamd_plan_level_mt_fill(self.h, level, out.ctypes.data), "cholamd_plan_level_mt_fill")
        return tuple(int(v) for v in out)

    def program_check(self, follow=True, workers=64):
        """Host-side self-check of the one-launch program (raises CholamdError on a dead-lock or a mismatch)."""
        check(self.L.cholamd_plan_program_check(self.h, int(follow), int(workers)), "cholamd_plan_program_check")

    def program_check_opts(self, follow_tail=-1, split_min=-1, split_nb=-1, workers=64):
        """program_check under other follower tails / pivot splits (negative: the default)."""
        check(self.L.cholamd_plan_program_check_opts(self.h, follow_tail, split_min, split_nb, int(workers)), "cholamd_plan_program_check_opts")

    def program_counts(self, follow=True):
        out = np.zeros(6, dtype=np.int32)
        check(self.L.cholamd_plan_program_counts(self.h, int(follow), out.ctypes.data), "cholamd_plan_program_counts")
        return dict(zip(("jobs", "followers", "tasks", "strips", "counters", "followed_panels"), (int(v) for v in out)))

    def program_jobs(self, follow=True):
        """Diagnostic: (jobs [n, 8] = kind, sep, aux, first, n, sig0, sig1, n_wait; waits [m, 3] = job, counter, value)."""
        need = self.L.cholamd_plan_program_jobs(self.h, int(follow), 0, None)
        buf = np.zeros(need, dtype=np.int32)
        self.L.cholamd_plan_program_jobs(self.h, int(follow), need, buf.ctypes.data)
        nj = int(buf[-1])
        return buf[:8 * nj].reshape(nj, 8), buf[8 * nj:-1].reshape(-1, 3)

    def arena_to_dense(self, arena):
        arena = np.ascontiguousarray(arena, dtype=np.float64)
        assert arena.size == self.arena_doubles
        d = np.zeros((self.n, self.n), dtype=np.float64, order="F")
        check(self.L.cholamd_plan_arena_to_dense(self.h, arena.ctypes.data, d.ctypes.data), "cholamd_plan_arena_to_dense")
        return d

    def write_matrix(self, arena, path, full_precision=False):
        arena = np.ascontiguousarray(arena, dtype=np.float64)
        check(self.L.cholamd_plan_write_matrix(self.h, arena.ctypes.data, os.fsencode(path), int(full_precision)), "cholamd_plan_write_matrix")

    def write_debug_log(self, path):
        libc = C.CDLL(None)
        libc.fopen.restype = C.c_void_p
        libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
        libc.fclose.argtypes = [C.c_void_p]
        fp = libc.fopen(os.fsencode(path), b"w")
        if not fp:
            raise IOError(path)
        try:
            check(self.L.cholamd_plan_write_debug_log(self.h, fp), "cholamd_plan_write_debug_log")
        finally:
            libc.fclose(fp)


def read_vector(path, n):
    out = np.zeros(n, dtype=np.float64)
    check(load().cholamd_read_vector(os.fsencode(path), n, out.ctypes.data), "cholamd_read_vector")
    return out


def write_solution(path, x, full_precision=False):
    x = np.ascontiguousarray(x, dtype=np.float64)
    check(load().cholamd_write_solution(os.fsencode(path), x.ctypes.data, x.size, int(full_precision)), "cholamd_write_solution")


class Problem:
    """Generated Laplacian + nested-dissection ordering + clusters (cholamd_generate_laplacian)."""

    def __init__(self, nx, ny=1, nz=1, levels=3, tile=32):
        self.L = load()
        h = C.c_void_p()
        check(self.L.cholamd_generate_laplacian(nx, ny, nz, levels, tile, C.byref(h)), "cholamd_generate_laplacian")
        self.h = h
        self.n = self.L.cholamd_problem_n(h)
        self.nz = self.L.cholamd_problem_nz(h)
        self.levels = levels

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.cholamd_problem_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def write(self, prefix):
        """<prefix>.mtx, _ord_<levels>.txt, _clust_<levels>.txt, _B.mtx in the reference's formats."""
        check(self.L.cholamd_problem_write(self.h, os.fsencode(prefix)), "cholamd_problem_write")
        lv = self.levels
        return f"{prefix}.mtx", f"{prefix}_ord_{lv}.txt", f"{prefix}_clust_{lv}.txt", f"{prefix}_B.mtx"

    def plan(self):
        h = C.c_void_p()
        check(self.L.cholamd_plan_create_from_problem(self.h, C.byref(h)), "cholamd_plan_create_from_problem")
        return Plan(_handle=h)

    def rhs(self):
        b = np.zeros(self.n, dtype=np.float64)
        self.L.cholamd_problem_rhs(self.h, b.ctypes.data)
        return b
