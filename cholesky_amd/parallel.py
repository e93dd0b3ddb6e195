"""Multi-GPU driver of the hot path: subtree sharding + one extend-add exchange.

One process per GPU.  The separator tree is cut at level d = log2(world): rank g factors the subtrees under
its level-d separator, the contributions to the shared top of the tree accumulate in the rank's own copy of
the arena tail, ONE all-reduce (sum) over that contiguous tail is the exchange, then every rank factors the
top levels.  The whole of it -- launches and the RCCL all-reduce -- is ONE C-ABI call
(cholamd_factor_sharded); torch.distributed only carries the RCCL unique id to the ranks (make_comm) and,
in the CPU-side tests, stands in for the exchange through a host tensor (via_host, gloo)."""
import torch
import torch.distributed as dist


def split_level(world):
    d = world.bit_length() - 1
    if world < 1 or (1 << d) != world:
        raise ValueError("world size must be a power of two")
    return d


def tail_offset(plan, world):
    """Arena offset (doubles) where the panels of the shared top of the tree start."""
    if world == 1:
        return plan.arena_doubles
    first_top = plan.nsep - (world - 1) + 1
    for b in plan.blocks:
        if b[0] == first_top and b[1] == first_top:
            return int(b[7])
    raise RuntimeError("top panel not found")


def make_comm(dev, world, rank, group=None):
    """libcholamd's own RCCL communicator for this rank: rank 0 draws the unique id (ncclGetUniqueId), the
    process group broadcasts its 128 bytes, every rank joins (ncclCommInitRank on its GPU)."""
    from .device import Comm
    box = [Comm.unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    return Comm(dev, world, rank, box[0])


def factor_sharded(dev, arena, world, tail, stream=None, group=None, via_host=False, comm=None):
    """One factorisation sharded over `world` ranks; `dev` must have set_partition(rank, world) applied
    and `arena` filled by dev.fill (rank-aware).

    comm (make_comm): the product path -- cholamd_factor_sharded, everything in order on `stream`.
    via_host: the exchange through a CPU tensor and torch.distributed (gloo; tests without one GPU per rank).
    Otherwise the exchange is torch.distributed's all_reduce on the device tensor, issued on `stream`."""
    levels = dev.plan.levels
    if world == 1:
        dev.factor(arena, stream)
        return
    if comm is not None:
        dev.factor_sharded(arena, comm, stream)
        return
    d = split_level(world)
    if dev.dist_top_active():
        # the top levels distributed by column blocks hold ncclBroadcast phases: they need libcholamd's communicator
        raise RuntimeError("factor_sharded without comm= (via_host / torch all_reduce) runs the REPLICATED top levels only: "
                           "call dev.set_option('dist_top', 0) before dev.set_partition(rank, world), or pass comm=make_comm(...)")
    dev.factor_levels(arena, levels - 1, d, stream)
    t = arena[tail:]
    if via_host:
        dev.sync(stream)
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
        torch.cuda.current_stream().synchronize()  # the copy ran on torch's current stream: done before `stream` goes on
    else:
        # the collective is enqueued on torch's CURRENT stream: make that `stream` for the call, so that it is
        # ordered after the local levels and before the top levels whatever stream the caller passed
        s = stream if isinstance(stream, torch.cuda.Stream) else (torch.cuda.current_stream() if stream is None else torch.cuda.ExternalStream(int(stream)))
        with torch.cuda.stream(s):
            dist.all_reduce(t, group=group)
    dev.factor_levels(arena, d - 1, 0, stream)
