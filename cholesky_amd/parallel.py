"""Multi-GPU driver of the hot path: subtree sharding + one extend-add exchange.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on a real node, "gloo" in
the tests).  The separator tree is cut at level d = log2(world): rank g factors the subtrees under
its level-d separator, the contributions to the shared top of the tree accumulate in the rank's own
copy of the arena tail, ONE all-reduce (sum) over that contiguous tail is the exchange, then every
rank factors the top levels.  All numerics are libcholamd launches; this module only orders them."""
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


def factor_sharded(dev, arena, world, tail, stream=None, group=None, via_host=False):
    """One factorisation sharded over `world` ranks; `dev` must have set_partition(rank, world) applied
    and `arena` filled by dev.fill (rank-aware).  via_host routes the exchange through a CPU tensor
    (gloo without CUDA support)."""
    levels = dev.plan.levels
    if world == 1:
        dev.factor(arena, stream)
        return
    d = split_level(world)
    dev.factor_levels(arena, levels - 1, d, stream)
    t = arena[tail:]
    if via_host:
        dev.sync(stream)
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=group)
    dev.factor_levels(arena, d - 1, 0, stream)
