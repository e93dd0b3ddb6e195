"""Device-side driver objects: the level schedule of mmat.rg:1227-1355 on one MI355X.

torch is used for what it is good at here -- device buffers, streams, torch.distributed (RCCL) --
and every numeric step is a call into libcholamd.so with raw pointers."""
import ctypes as C

import numpy as np

from ._lib import check, load


def _stream_ptr(stream):
    if stream is None:
        return None
    if hasattr(stream, "cuda_stream"):
        return C.c_void_p(stream.cuda_stream)
    return C.c_void_p(int(stream))


class RankArena:
    """A device arena owned by the library (Device.alloc_arena): data_ptr() like a tensor, numpy() = a host copy of the whole address range."""

    def __init__(self, dev, ptr, elem_bytes, backed_bytes):
        self.dev, self._ptr, self.elem_bytes, self.backed_bytes = dev, ptr, elem_bytes, backed_bytes
        self.nbytes = dev.plan.arena_doubles * elem_bytes

    def data_ptr(self):
        return self._ptr

    def numpy(self):
        n = self.dev.plan.arena_doubles
        out = np.empty(n, dtype=np.float64 if self.elem_bytes == 8 else np.float32)
        words = out.nbytes // 8  # cholamd_device_download counts doubles (the fp32 arena of an even number of floats, or its last float stays behind)
        check(self.dev.L.cholamd_device_download(self.dev.h, out.ctypes.data, C.c_void_p(self._ptr), words, None), "download")
        return out

    def free(self):
        if self._ptr:
            check(self.dev.L.cholamd_device_free_arena(self.dev.h, C.c_void_p(self._ptr)), "cholamd_device_free_arena")
            self._ptr = 0


class Device:
    KINDS = ("potrf", "trsm", "update", "other")

    def __init__(self, plan, device_id=0):
        self.L = load()
        self.plan = plan
        self.device_id = device_id
        h = C.c_void_p()
        check(self.L.cholamd_device_create(plan.h, device_id, C.byref(h)), "cholamd_device_create")
        self.h = h
        self._owned = []

    def __del__(self):
        try:
            if getattr(self, "h", None):
                for p in self._owned:
                    self.L.cholamd_device_free(self.h, p)
                self.L.cholamd_device_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # -- memory ---------------------------------------------------------------------------------
    def alloc(self, doubles):
        """Library-owned device buffer (hipMalloc); returns the raw pointer as int."""
        p = C.c_void_p()
        check(self.L.cholamd_device_alloc(self.h, int(doubles), C.byref(p)), "cholamd_device_alloc")
        self._owned.append(p)
        return p.value

    def new_arena(self):
        """A torch fp64 CUDA tensor of the arena size (caller-owned buffer, like Legion's regions)."""
        import torch
        return torch.empty(self.plan.arena_doubles, dtype=torch.float64, device=f"cuda:{self.device_id}")

    def alloc_arena(self, elem_bytes=8):
        """This rank's arena by cholamd_device_alloc_arena: complete address range, memory of its own only under the rank's panels and the shared
        top (the other ranks' panels alias one scratch chunk); rank 0 / single GPU: a plain allocation.  Returns a RankArena."""
        p, backed = C.c_void_p(), C.c_int64(0)
        check(self.L.cholamd_device_alloc_arena(self.h, int(elem_bytes), C.byref(p), C.byref(backed)), "cholamd_device_alloc_arena")
        return RankArena(self, p.value, int(elem_bytes), backed.value)

    @staticmethod
    def ptr(t):
        return C.c_void_p(t if isinstance(t, int) else t.data_ptr())

    def upload(self, dptr, host, stream=None):
        host = np.ascontiguousarray(host, dtype=np.float64)
        check(self.L.cholamd_device_upload(self.h, self.ptr(dptr), host.ctypes.data, host.size, _stream_ptr(stream)), "upload")

    def download(self, dptr, doubles, stream=None):
        out = np.empty(int(doubles), dtype=np.float64)
        check(self.L.cholamd_device_download(self.h, out.ctypes.data, self.ptr(dptr), out.size, _stream_ptr(stream)), "download")
        return out

    def sync(self, stream=None):
        check(self.L.cholamd_device_sync(self.h, _stream_ptr(stream)), "sync")

    # -- the path -------------------------------------------------------------------------------
    def fill(self, arena, stream=None):
        """A scatter on the device: fill_block for every block (mmat.rg:1216-1224)."""
        check(self.L.cholamd_device_fill(self.h, self.ptr(arena), _stream_ptr(stream)), "cholamd_device_fill")

    def factor(self, arena, stream=None):
        """The level loop (mmat.rg:1227-1355), asynchronous on `stream`."""
        check(self.L.cholamd_factor(self.h, self.ptr(arena), _stream_ptr(stream)), "cholamd_factor")

    def factor_levels(self, arena, level_hi, level_lo, stream=None):
        check(self.L.cholamd_factor_levels(self.h, self.ptr(arena), level_hi, level_lo, _stream_ptr(stream)), "cholamd_factor_levels")

    def set_partition(self, rank, world):
        check(self.L.cholamd_device_set_partition(self.h, rank, world), "cholamd_device_set_partition")

    def dist_top_active(self):
        """True when the partitioned schedule holds broadcast phases (top levels distributed by column blocks)."""
        return self.L.cholamd_device_bcast_phases(self.h) > 0

    def set_option(self, name, value):
        """Schedule / kernel-selection switch of this device object (cholamd_device_set_option)."""
        check(self.L.cholamd_device_set_option(self.h, name.encode(), int(value)), "cholamd_device_set_option")

    def program_trace(self, arena, stream=None):
        """Diagnostic: one program launch with per-job stamps -> int64 array [jobs, 5] = kind, drawn, waits over, ended (10 ns ticks), workgroup."""
        n = C.c_int(0)
        check(self.L.cholamd_device_program_trace(self.h, self.ptr(arena), _stream_ptr(stream), 0, None, C.byref(n)), "cholamd_device_program_trace")
        out = np.zeros((n.value, 5), dtype=np.int64)
        check(self.L.cholamd_device_program_trace(self.h, self.ptr(arena), _stream_ptr(stream), out.size, out.ctypes.data, C.byref(n)), "cholamd_device_program_trace")
        return out

    def tail_offset(self):
        """First double of the shared top of the tree in the arena under the current partition."""
        return int(self.L.cholamd_device_tail_offset(self.h))

    def exchange_tail(self, arena, comm, stream=None):
        """The extend-add exchange alone: in-place RCCL all-reduce (sum) of the arena tail (cholamd_exchange_tail)."""
        check(self.L.cholamd_exchange_tail(self.h, self.ptr(arena), comm.h, _stream_ptr(stream)), "cholamd_exchange_tail")

    def factor_sharded(self, arena, comm, stream=None):
        """This rank's part of a sharded factorisation (cholamd_factor_sharded): local levels, RCCL exchange, top levels."""
        check(self.L.cholamd_factor_sharded(self.h, self.ptr(arena), comm.h if comm is not None else None, _stream_ptr(stream)), "cholamd_factor_sharded")

    def factor_sharded_f32(self, arena32, comm, stream=None):
        """The same with the fp32 factor (cholamd_factor_sharded_f32): mixed precision x multi-GPU."""
        check(self.L.cholamd_factor_sharded_f32(self.h, self.ptr(arena32), comm.h if comm is not None else None, _stream_ptr(stream)), "cholamd_factor_sharded_f32")

    def gather_to_root(self, arena, comm, stream=None):
        """The subtree panels this rank owns -> rank 0's arena (cholamd_gather_to_root); fp64 or fp32 arena by the tensor's dtype."""
        check(self.L.cholamd_gather_to_root(self.h, self.ptr(arena), int(arena.element_size()), comm.h, _stream_ptr(stream)), "cholamd_gather_to_root")

    def exchange_volume(self):
        """(received, sent, tail, pieces) in elements of the extend-add exchange of this rank's partition (cholamd_exchange_volume)."""
        out = np.zeros(4, dtype=np.int64)
        check(self.L.cholamd_exchange_volume(self.h, out.ctypes.data), "cholamd_exchange_volume")
        return tuple(int(v) for v in out)

    def info(self):
        sep = C.c_int(0)
        rc = self.L.cholamd_factor_info(self.h, C.byref(sep))
        if rc < 0:
            check(rc, "cholamd_factor_info")
        return rc, sep.value

    def solve(self, arena, b, x, stream=None):
        check(self.L.cholamd_solve(self.h, self.ptr(arena), self.ptr(b), self.ptr(x), _stream_ptr(stream)), "cholamd_solve")

    # -- mixed precision: fp32 factor + fp64 iterative refinement (BASELINE config 5) -----------------
    def new_arena_f32(self):
        import torch
        return torch.empty(self.plan.arena_doubles, dtype=torch.float32, device=f"cuda:{self.device_id}")

    def fill_f32(self, arena32, stream=None):
        check(self.L.cholamd_device_fill_f32(self.h, self.ptr(arena32), _stream_ptr(stream)), "cholamd_device_fill_f32")

    def factor_f32(self, arena32, stream=None):
        check(self.L.cholamd_factor_f32(self.h, self.ptr(arena32), _stream_ptr(stream)), "cholamd_factor_f32")

    def factor_levels_f32(self, arena32, level_hi, level_lo, stream=None):
        check(self.L.cholamd_factor_levels_f32(self.h, self.ptr(arena32), level_hi, level_lo, _stream_ptr(stream)), "cholamd_factor_levels_f32")

    def solve_f32(self, arena32, b, x, stream=None):
        check(self.L.cholamd_solve_f32(self.h, self.ptr(arena32), self.ptr(b), self.ptr(x), _stream_ptr(stream)), "cholamd_solve_f32")

    def solve_refine(self, arena32, b, x, max_iter=20, tol=1e-12, stream=None):
        """x = A^-1 b by iterative refinement on the fp32 factor; returns (corrections applied, ||b - A x|| / ||b||)."""
        it, rel = C.c_int(0), C.c_double(0.0)
        check(self.L.cholamd_solve_refine(self.h, self.ptr(arena32), self.ptr(b), self.ptr(x), int(max_iter), float(tol),
                                          C.byref(it), C.byref(rel), _stream_ptr(stream)), "cholamd_solve_refine")
        return it.value, rel.value

    def solve_sharded(self, arena, b, x, comm, stream=None):
        """This rank's part of the distributed solve (cholamd_solve_sharded / _f32 by the arena's element type): only vectors travel."""
        import torch
        f32 = arena.elem_bytes == 4 if isinstance(arena, RankArena) else arena.dtype == torch.float32
        fn = self.L.cholamd_solve_sharded_f32 if f32 else self.L.cholamd_solve_sharded
        check(fn(self.h, self.ptr(arena), self.ptr(b), self.ptr(x), comm.h if comm is not None else None, _stream_ptr(stream)), "cholamd_solve_sharded")

    def solve_refine_sharded(self, arena32, b, x, comm, max_iter=20, tol=1e-12, stream=None):
        """cholamd_solve_refine_sharded: the fp64 refinement with the fp32 factor left on the ranks; returns (corrections, relres)."""
        it, rel = C.c_int(0), C.c_double(0.0)
        check(self.L.cholamd_solve_refine_sharded(self.h, self.ptr(arena32), self.ptr(b), self.ptr(x), max_iter, tol, C.byref(it), C.byref(rel),
                                                  comm.h if comm is not None else None, _stream_ptr(stream)), "cholamd_solve_refine_sharded")
        return int(it.value), float(rel.value)

    def residual(self, b, x, r=None, stream=None):
        """||b - A x|| / ||b|| in fp64 on the device (A = the matrix file's entries)."""
        rel = C.c_double(0.0)
        check(self.L.cholamd_residual(self.h, self.ptr(b), self.ptr(x), self.ptr(r) if r is not None else None, C.byref(rel), _stream_ptr(stream)), "cholamd_residual")
        return rel.value

    def set_timing(self, on):
        check(self.L.cholamd_device_set_timing(self.h, int(on)), "set_timing")

    def event_overhead_ms(self, stream=None):
        v = C.c_float(0)
        check(self.L.cholamd_device_event_overhead(self.h, _stream_ptr(stream), C.byref(v)), "event_overhead")
        return float(v.value)

    def get_timing(self):
        ms = np.zeros(4, dtype=np.float32)
        cnt = np.zeros(4, dtype=np.int32)
        check(self.L.cholamd_device_get_timing(self.h, ms.ctypes.data, cnt.ctypes.data), "get_timing")
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(self.KINDS)}

    def get_timing_ex(self):
        """The compute kinds plus the exchange kinds of a sharded run (cholamd_device_get_timing_ex)."""
        ms = np.zeros(8, dtype=np.float32)
        cnt = np.zeros(8, dtype=np.int32)
        check(self.L.cholamd_device_get_timing_ex(self.h, ms.ctypes.data, cnt.ctypes.data), "get_timing_ex")
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(self.KINDS + ("exchange", "bcast"))}


class Comm:
    """An RCCL communicator owned by libcholamd (cholamd_comm_create: ncclCommInitRank on the device's GPU)."""

    def __init__(self, dev, world, rank, unique_id):
        self.L = load()
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), 128)
        check(self.L.cholamd_comm_create(dev.h, world, rank, buf, C.byref(h)), "cholamd_comm_create")
        self.h, self.world, self.rank = h, world, rank

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        check(load().cholamd_comm_unique_id(buf), "cholamd_comm_unique_id")
        return bytes(buf.raw)

    def count(self):
        """ncclCommCount of the communicator behind the handle."""
        n = C.c_int(0)
        check(self.L.cholamd_comm_count(self.h, C.byref(n)), "cholamd_comm_count")
        return int(n.value)

    def allreduce(self, t, stream=None):
        """In-place fp64 sum of a CUDA tensor over the ranks (cholamd_comm_allreduce)."""
        check(self.L.cholamd_comm_allreduce(self.h, C.c_void_p(t.data_ptr()), t.numel(), _stream_ptr(stream)), "cholamd_comm_allreduce")

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.cholamd_comm_destroy(self.h)
                self.h = None
        except Exception:
            pass


def factor_multi(devs, arenas, local=True, streams=None):
    """One process driving the n rank objects `devs` (devs[g] partitioned as rank g of n): cholamd_factor_multi with a LOCAL
    communicator (device-side sums and peer copies: the ranks may share a GPU) or an RCCL one (ncclCommInitAll, one GPU per rank).
    fp32 arenas (torch.float32) take the fp32 schedule (cholamd_factor_multi_f32)."""
    import torch
    L = load()
    f32 = arenas[0].elem_bytes == 4 if isinstance(arenas[0], RankArena) else arenas[0].dtype == torch.float32
    n = len(devs)
    hd = (C.c_void_p * n)(*[d.h for d in devs])
    ha = (C.c_void_p * n)(*[C.c_void_p(a.data_ptr()) for a in arenas])
    hc = (C.c_void_p * n)()
    hs = (C.c_void_p * n)(*[_stream_ptr(s) for s in streams]) if streams is not None else None
    check((L.cholamd_comm_create_local if local else L.cholamd_comm_create_all)(hd, n, hc), "cholamd_comm_create")
    try:
        check((L.cholamd_factor_multi_f32 if f32 else L.cholamd_factor_multi)(hd, ha, hc, n, hs), "cholamd_factor_multi")
        for d in devs:
            d.sync()
    finally:
        for c in hc:
            L.cholamd_comm_destroy(c)


def _multi_call(devs, local, body):
    L = load()
    n = len(devs)
    hd = (C.c_void_p * n)(*[d.h for d in devs])
    hc = (C.c_void_p * n)()
    check((L.cholamd_comm_create_local if local else L.cholamd_comm_create_all)(hd, n, hc), "cholamd_comm_create")
    try:
        out = body(L, n, hd, hc)
        for d in devs:
            d.sync()
        return out
    finally:
        for c in hc:
            L.cholamd_comm_destroy(c)


def solve_multi(devs, arenas, bs, xs, local=True):
    """One process driving the n rank objects through the distributed solve (cholamd_solve_multi): the factor stays where
    cholamd_factor_multi left it, every rank ends with the whole solution in xs[g]."""
    def body(L, n, hd, hc):
        ha = (C.c_void_p * n)(*[C.c_void_p(a.data_ptr()) for a in arenas])
        hb = (C.c_void_p * n)(*[C.c_void_p(b.data_ptr()) for b in bs])
        hx = (C.c_void_p * n)(*[C.c_void_p(x.data_ptr()) for x in xs])
        check(L.cholamd_solve_multi(hd, ha, hb, hx, hc, n, None), "cholamd_solve_multi")
    return _multi_call(devs, local, body)


def solve_refine_multi(devs, arenas32, bs, xs, max_iter=20, tol=1e-12, local=True):
    """cholamd_solve_refine_multi: fp64 refinement with the fp32 factor left on the ranks; returns (corrections, relres)."""
    def body(L, n, hd, hc):
        ha = (C.c_void_p * n)(*[C.c_void_p(a.data_ptr()) for a in arenas32])
        hb = (C.c_void_p * n)(*[C.c_void_p(b.data_ptr()) for b in bs])
        hx = (C.c_void_p * n)(*[C.c_void_p(x.data_ptr()) for x in xs])
        it, rel = C.c_int(0), C.c_double(0.0)
        check(L.cholamd_solve_refine_multi(hd, ha, hb, hx, max_iter, tol, C.byref(it), C.byref(rel), hc, n, None), "cholamd_solve_refine_multi")
        return int(it.value), float(rel.value)
    return _multi_call(devs, local, body)
