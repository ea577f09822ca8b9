#!/usr/bin/env python3
"""Headline benchmark: particle-updates/sec of the firework effect (BASELINE.json configs[1]).

A "step" is one simulated frame (one pass of the hot path) over a resident 16,777,216-particle
effect: frame inputs upload + k_update (age, LinearDrag, Accel, Euler, kill test, alive/dead list
rebuild). The burst spawn frame (k_init) runs during warm-up; all particles stay alive during the
timed frames (minimum lifetime 0.8 s > (warmup + steps)/60 s for the default step counts).

N > 1: one process per GPU (torch.distributed / RCCL). The effect is sharded by capacity slab:
rank g simulates slots [g*C, (g+1)*C) of a logical N*C-particle effect (global slot index feeds the
PRNG, so the union equals a single-GPU run); there is no data-path collective, only an all-reduce of
the alive counters for reporting. Weak scaling.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bevy_hanabi_amd as bh  # noqa: E402
from bevy_hanabi_amd import effects, sharding  # noqa: E402

CAPACITY = 1 << 24
BYTES_PER_UPDATE = 68  # reads pos12+vel12+age4+lifetime4, writes pos12+vel12+age4, + 8 B alive-list entry
HBM_PEAK_GBS = 8000.0  # MI355X spec (guides/MI355X_MICROARCH.md)
# HBM bytes per k_update_slots_stream launch at capacity 16,777,216 from the PMC passes committed under
# profiles/ (r01i_summary.md: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate rocprofv3 runs; with lifetime
# culling the LIFETIME plane is not read in the timed frames, and completely alive chunks skip the alive bytes:
# 56.06 B per particle = pos + vel + age read, pos + vel + age written).
# Counters cannot be collected from inside this script; the figure is per launch of this workload.
PMC_TRAFFIC_BYTES = {1 << 24: 9.406e8}
PMC_TRAFFIC_SOURCE = "profiles/r01i_summary.md"
DT = 1.0 / 60.0
MIN_LIFETIME = 0.8  # firework.rs: lifetime = uniform(0.8, 1.2)
TIMING_PERIOD = 5   # HIP events bracket the kernels of every 5th timed frame (each costs ~20 us of stream bubbles)


def frame_dt(total_frames):
    """1/60 s like the reference's example; shrunk only if a long --steps run would outlive the
    youngest particle (the metric is defined on frames where all 16M particles are alive)."""
    return DT if total_frames * DT < MIN_LIFETIME * 0.95 else MIN_LIFETIME * 0.95 / total_frames


def frame_seed(f):
    # harness-defined per-frame seed list: pcg_hash(0xC0FFEE + f)
    x = (0xC0FFEE + f) & 0xFFFFFFFF
    state = (x * 747796405 + 2891336453) & 0xFFFFFFFF
    word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
    return ((word >> 22) ^ word) & 0xFFFFFFFF


def cpu_baseline(sample_capacity=1 << 22, frames=24):
    """The oracle (a restatement of the reference's WGSL semantics; the reference has no CPU
    simulation path) timed with OpenMP on the host cores, on a bounded sample of the same workload."""
    import oracle

    oracle.build()
    asset = effects.firework_trails(sample_capacity)
    fx = oracle.OracleEffect(bh.serialize_asset(asset), omp=True)
    fx.step(DT, sample_capacity, frame_seed(0))  # spawn frame (not timed)
    t0 = time.perf_counter()
    for f in range(1, frames + 1):
        fx.step(DT, 0, frame_seed(f), time=f * DT)
    t = time.perf_counter() - t0
    assert fx.alive_count() == sample_capacity
    return {"value": sample_capacity * frames / t, "unit": "particle-updates/s", "cores": oracle.omp_threads(), "kind": "port",
            "sample": f"{sample_capacity} particles x {frames} frames of the same firework effect ({t:.1f} s wall, OpenMP oracle)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--capacity", type=int, default=CAPACITY)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for dry runs)")
    ap.add_argument("--force-device", type=int, default=None, help="dry runs: every rank uses this GPU instead of LOCAL_RANK")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    device_index = local_rank if args.force_device is None else args.force_device
    reduce_device = "cuda" if args.backend == "nccl" else "cpu"
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(device_index)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    n_gpus = world if distributed else 1
    if args.gpus != n_gpus and rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run for N>1", file=sys.stderr)

    cap = args.capacity
    asset = effects.firework_trails(cap)
    ctx = bh.Context(device_index)
    prog = ctx.create_program(bh.lower(asset))
    slot_base, _ = sharding.slab_plan(cap * n_gpus, n_gpus)[rank]  # rank g owns global slots [g*cap, (g+1)*cap)
    fx = prog.create_effect(slot_base=slot_base)

    dt = frame_dt(1 + args.warmup + args.steps)

    def step(f, spawn=0):
        ctx.frame_begin(dt, f * dt)
        fx.set_frame(spawn, frame_seed(f))
        ctx.simulate()

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()

    # warm-up: frame 0 is the burst (k_init + k_update), then untimed update frames
    ctx.enable_kernel_timing(1)
    step(0, cap)
    init_ms = ctx.kernel_timing()["init_ms_avg"]
    ctx.enable_kernel_timing(0)
    for f in range(1, args.warmup + 1):
        step(f)
    barrier()
    ctx.enable_kernel_timing(TIMING_PERIOD)
    t0 = time.perf_counter()
    for f in range(args.warmup + 1, args.warmup + 1 + args.steps):
        step(f)
    barrier()
    elapsed = time.perf_counter() - t0
    timing = ctx.kernel_timing()
    ctx.enable_kernel_timing(0)

    alive = fx.alive_count()
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the only collective of the design: alive-particle counters, for reporting
        alive_total = sharding.allreduce_alive([alive], device=reduce_device)[0]
    else:
        alive_total = alive
    assert alive_total == cap * n_gpus, f"expected every particle alive during the timed frames, got {alive_total}"

    if rank == 0:
        updates = float(cap) * n_gpus * args.steps
        value = updates / elapsed
        k_ms = timing["update_ms_avg"]
        achieved = cap * BYTES_PER_UPDATE / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        out = {
            "metric": "particle-updates/sec", "value": value, "unit": "particle-updates/s", "n_gpus": n_gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"firework.rs trails EffectAsset, capacity={cap:_} per GPU, burst spawner, all particles alive",
                       "capacity_per_gpu": cap, "dt": dt, "sharding": f"capacity slab x{n_gpus}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": PMC_TRAFFIC_BYTES.get(cap), "traffic_unit": "B/launch", "traffic_source": PMC_TRAFFIC_SOURCE, "kernel": "k_update_slots_stream<ProgDragAccel>", "kernel_ms_avg": k_ms, "compact_ms_avg": timing["compact_ms_avg"],
                         "kernel_samples": timing["frames"], "timing": f"HIP events on the simulation stream, every {TIMING_PERIOD}th timed frame", "bytes_per_update": BYTES_PER_UPDATE,
                         "hbm_gbs_whole_step": updates / n_gpus * BYTES_PER_UPDATE / elapsed / 1e9},
        }
        # the burst frame's init kernel (not part of the metric): 44 B per spawned particle
        out["init"] = {"kernel": "k_init", "kernel_ms": init_ms, "spawned": cap, "bytes_per_spawn": 44,
                       "achieved_gbs": cap * 44 / (init_ms * 1e-3) / 1e9 if init_ms > 0 else 0.0, "kernels": prog.kernel_info().split("\n")[0]}
        if not args.no_cpu_baseline and n_gpus == 1:   # rank 0 at N=1 only: the host cores are shared by the ranks otherwise
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    ctx.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
