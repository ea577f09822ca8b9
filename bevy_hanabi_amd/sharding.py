"""Multi-GPU partitioning of the hot path: one process per GPU, no data-path
collective. Effect instances share nothing (batch.rs:153-173 refuses to merge anything with
parent/child links), so the units are:

  * many instances of an effect (C4): instance i lives on rank i mod world;
  * one giant effect (C2/C3/C5): capacity slabs. Rank g owns global slots [g*C, (g+1)*C) and
    passes slot_base = g*C to `hnb_effect_create`, so the PRNG (seeded by the GLOBAL slot index,
    vfx_init.wgsl:138 / vfx_update.wgsl:138) draws exactly what a single-GPU run would; a spawn
    request is split in rank order against each slab's free capacity.

The only collective is an all-reduce (RCCL on GPUs, gloo in the CPU tests) of per-effect
alive counters, for reporting.
"""
from typing import List, Sequence, Tuple


def slab_plan(total_capacity: int, world: int) -> List[Tuple[int, int]]:
    """[(slot_base, capacity)] per rank; the remainder goes to the lowest ranks."""
    if world <= 0 or total_capacity < 0:
        raise ValueError("world must be positive and capacity non-negative")
    base, rem = divmod(total_capacity, world)
    out, at = [], 0
    for g in range(world):
        c = base + (1 if g < rem else 0)
        out.append((at, c))
        at += c
    return out


def split_spawn(spawn_count: int, free_per_rank: Sequence[int]) -> List[int]:
    """Deterministic split of one frame's spawn request: rank order, capped by free slab capacity
    (the single-GPU semantics `min(spawn, capacity - alive)` of vfx_init.wgsl:118-121 hold for the union)."""
    out, left = [], int(spawn_count)
    for free in free_per_rank:
        n = min(left, int(free))
        out.append(n)
        left -= n
    return out


def instance_plan(n_instances: int, world: int) -> List[List[int]]:
    """Instance indices per rank: instance i -> rank i mod world."""
    return [list(range(g, n_instances, world)) for g in range(world)]


def allreduce_alive(local_counts: Sequence[int], device=None):
    """Sum per-effect alive counters over ranks (reporting only). Uses the default process group."""
    import torch
    import torch.distributed as dist

    t = torch.tensor(list(local_counts), dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(x) for x in t.tolist()]
