#!/usr/bin/env python3
"""Benchmark of the particle hot path on MI355X: particle-updates/sec + achieved HBM GB/s (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5] [--scaling weak|strong|both]

A "step" is one simulated frame (one pass of the hot path: per-frame inputs upload, init where the frame spawns, update +
kill + list maintenance, ribbon sort where the layout has RIBBON_ID) over particle state resident in HBM.

Configurations (synthetic scalings of the reference's example assets):
  c2 (default, the headline)  examples/firework.rs `trails` effect, capacity 16,777,216 per GPU, burst spawn during warm-up,
                              every particle alive in the timed frames; sharded by CAPACITY SLAB (rank g owns global slots
                              [g*C, (g+1)*C), `slot_base` feeds the PRNG so the union equals a single-GPU run);
  c3                          examples/force_field.rs, capacity 8,388,608 per GPU, burst, capacity slabs;
  c4                          examples/instancing.rs: independent instances x 65,536, sharded BY INSTANCE
                              (sharding.instance_plan: instance i on rank i mod N); 512 instances per GPU, i.e. BASELINE's
                              4096 instances at N = 8 (`--instances 4096` puts the whole configuration on one GPU);
  c5                          examples/ribbon.rs, capacity 4,194,304 per GPU, rate spawner in steady spawn/kill churn,
                              ribbon sort included in the step.

N > 1: `python bench.py --gpus N` launches itself under `python -m torch.distributed.run` (one process per GPU, RCCL); when the
driver already started it that way (WORLD_SIZE in the environment) it just runs its rank. There is no data-path collective: the
only collective is the all-reduce of the alive-particle counters for reporting (plus the MAX of the elapsed times). The JSON
line is the weak-scaling run (per-GPU work fixed); with N > 1 a strong-scaling run (the N = 1 workload split over the ranks) is
reported inside it under "strong".
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# CPU baseline: OpenMP threads are bound to cores. libgomp reads these when it is first loaded (which `import torch` may already
# trigger), so they are set before any import that could; unbound threads measured 6-15x slower on the 2-socket host.
os.environ.setdefault("OMP_PROC_BIND", "spread")
os.environ.setdefault("OMP_PLACES", "cores")

import numpy as np  # noqa: E402
import torch  # noqa: E402,F401  (before the HIP library of this package: torch bundles its own libamdhip64, and whichever copy is loaded
#                                 first is the one that owns the devices — loaded second, torch reports "No HIP GPUs are available")

HBM_PEAK_GBS = 8000.0  # MI355X spec (guides/MI355X_MICROARCH.md)
DT = 1.0 / 60.0
MIN_LIFETIME = 0.8  # firework.rs: lifetime = uniform(0.8, 1.2)
TIMING_PERIOD = 5   # HIP events bracket the kernels of every 5th timed frame (each costs ~20 us of stream bubbles)
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")  # written by tools/prof_bench.sh from the PMC passes

# algorithmic bytes per particle update, SURVEY.md §8(d): attributes read + attributes written + 8 B alive-list entry
CONFIGS = {
    "c2": dict(capacity=1 << 24, bytes_per_update=68, bytes_per_spawn=44, kernel="k_update_slots_stream<ProgDragAccel>",
               workload="firework.rs trails EffectAsset, capacity={cap:_} per GPU, burst spawner, all particles alive"),
    "c3": dict(capacity=1 << 23, bytes_per_update=68, bytes_per_spawn=40, kernel="k_update_slots_stream<ProgForceField>",
               workload="force_field.rs EffectAsset (2x ConformToSphere + KillAabb + KillSphere), capacity={cap:_} per GPU, burst"),
    "c4": dict(capacity=65536, bytes_per_update=68, bytes_per_spawn=40, kernel="k_update_slots_stream<ProgAgeEuler>", instances=512,
               workload="instancing.rs: {inst} independent effect instances x {cap:_} per GPU (one launch), burst, all alive"),
    "c5": dict(capacity=1 << 22, bytes_per_update=20, bytes_per_spawn=36, kernel="k_update_slots_stream<ProgAge>",
               workload="ribbon.rs EffectAsset, capacity={cap:_} per GPU, rate spawner in steady spawn/kill churn, ribbon sort in the step"),
}


def frame_dt(total_frames):
    """1/60 s like the reference's example; shrunk only if a long --steps run would outlive the
    youngest particle (the c2 metric is defined on frames where all particles are alive)."""
    return DT if total_frames * DT < MIN_LIFETIME * 0.95 else MIN_LIFETIME * 0.95 / total_frames


def pcg_hash(x):
    x &= 0xFFFFFFFF
    state = (x * 747796405 + 2891336453) & 0xFFFFFFFF
    word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
    return ((word >> 22) ^ word) & 0xFFFFFFFF


def frame_seed(f):
    # harness-defined per-frame seed list: pcg_hash(0xC0FFEE + f)
    return pcg_hash(0xC0FFEE + f)


def instance_seed(f, i):
    """Per-frame seed of (global) instance i: independent of how the instances are sharded."""
    return (frame_seed(f) ^ (i * 2654435761)) & 0xFFFFFFFF


def instance_transform(i):
    """instancing.rs places its instances on a grid; here 64 columns, 10 units apart."""
    return [1, 0, 0, 10.0 * (i % 64), 0, 1, 0, 0.0, 0, 0, 1, 10.0 * (i // 64)]


def load_traffic(config, capacity, n_inst):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/traffic.json, written by
    tools/prof_bench.sh: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate rocprofv3 runs). Counters cannot be
    collected from inside this process; None when no pass was recorded for this workload."""
    try:
        with open(TRAFFIC_FILE) as f:
            t = json.load(f)
        e = t.get(f"{config}:{capacity}x{n_inst}")
        return (e["bytes_per_launch"], e["source"]) if e else (None, None)
    except (OSError, ValueError, KeyError):
        return None, None


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline
# ------------------------------------------------------------------------------------------------------------------
def cpu_baseline(capacity, frames=60, check_frames=2):
    """A tuned CPU port of the lowered firework update (oracle/cpu_soa.c: packed SoA planes, OpenMP over 4096-particle blocks,
    -O3 -march=native -ffp-contract=off) on the FULL configuration, timed on the host cores. The reference has no CPU
    simulation path, so this is a port, not bevy_hanabi code. Before it is timed the port is checked
    bit-for-bit against the oracle (hanabi_oracle.c, the restatement of the WGSL semantics) on the same particles.
    Threads are bound (OMP_PROC_BIND=spread over OMP_PLACES=cores unless the environment says otherwise: unbound threads
    measured 6-15x slower on the 2-socket host) and the thread count is the best of {1/8, 1/4, 1/2} of the hardware threads."""
    import bevy_hanabi_amd as bh
    import oracle
    from bevy_hanabi_amd import effects

    oracle.build()
    t_all = time.perf_counter()
    orc = oracle.OracleEffect(bh.serialize_asset(effects.firework_trails(capacity)), omp=True)
    orc.step(DT, capacity, frame_seed(0))  # burst frame (spawn + first update), by the oracle
    alive = np.zeros(capacity, np.uint8)
    alive[orc.alive_list()] = 1
    soa = oracle.CpuSoaEffect(orc.read_attr(2), orc.read_attr(3), orc.read_attr(4).reshape(-1), orc.read_attr(5).reshape(-1), alive)
    f32 = np.float32

    def ops_for(dt):
        dt = f32(dt)
        drag = max(f32(0.0), f32(1.0) - f32(4.0) * dt)             # LinearDragModifier(4): max(0., (1.) - ((4.) * (dt)))
        accel = np.array([-0.0, -16.0, -0.0], np.float32) * dt      # AccelModifier: (vec3(-0.,-16.,-0.)) * dt
        return [(oracle.HCS_AGE_TICK, (dt,)), (oracle.HCS_VEL_SCALE, (drag,)), (oracle.HCS_VEL_ADD, accel), (oracle.HCS_EULER, (dt,))]

    for f in range(1, check_frames + 1):
        orc.step(DT, 0, frame_seed(f), time=f * DT)
        soa.update(ops_for(DT))
    for attr, mine in ((2, soa.pos), (3, soa.vel), (4, soa.age)):
        if not np.array_equal(orc.read_attr(attr).view(np.uint32).reshape(-1), mine.view(np.uint32).reshape(-1)):
            raise RuntimeError(f"cpu_baseline: the SoA port differs from the oracle on attribute {attr}")
    orc.close()
    state = [a.copy() for a in (soa.pos, soa.vel, soa.age, soa.life, soa.alive)]
    hw = os.cpu_count() or 1
    tried = {}
    # the timed frames use a small dt so that no particle reaches its lifetime however many frames are timed (the arithmetic
    # per frame is the same); every thread count gets a fresh first-touch copy of the state
    ops = ops_for(1e-4)
    for threads in sorted({max(1, hw // 8), max(1, hw // 4), max(1, hw // 2)}):
        oracle.CpuSoaEffect.set_threads(threads)
        soa = oracle.CpuSoaEffect(*state)
        for _ in range(3):
            soa.update(ops)
        best = None
        for _rep in range(3):
            t0 = time.perf_counter()
            died = 0
            for _ in range(frames):
                died += soa.update(ops)
            t = time.perf_counter() - t0
            assert died == 0, "cpu_baseline: particles died in the timed frames"
            best = t if best is None else min(best, t)
        tried[threads] = capacity * frames / best
    threads = max(tried, key=tried.get)
    return {"value": tried[threads], "unit": "particle-updates/s", "cores": threads, "kind": "port",
            "hbm_equiv_gbs": tried[threads] * 56 / 1e9, "threads_tried": {str(k): v for k, v in tried.items()},
            "sample": f"{capacity} particles x {frames} frames (best of 3 repeats) of the same firework update (all alive), packed-SoA OpenMP "
                      f"port (oracle/cpu_soa.c, -O3 -march=native, threads bound {os.environ['OMP_PROC_BIND']}/{os.environ['OMP_PLACES']}); "
                      f"checked bit-equal to the oracle on all {capacity} particles x {check_frames} frames first; {time.perf_counter() - t_all:.1f} s in total. "
                      "Note: the 0.5 GB of state fits in the host's last-level caches (2 x 256 MB L3 on the GPU box)"}


# ------------------------------------------------------------------------------------------------------------------
# one configuration on one rank
# ------------------------------------------------------------------------------------------------------------------
class Dist:
    def __init__(self, args):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.on = self.world > 1
        self.device_index = self.local_rank if args.force_device is None else args.force_device
        self.reduce_device = "cuda" if args.backend == "nccl" else "cpu"
        self.backend = args.backend
        if self.on:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            torch.cuda.set_device(self.device_index)
            if args.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", self.device_index))
            else:
                dist.init_process_group(args.backend, rank=self.rank, world_size=self.world)

    def barrier(self, ctx):
        import torch
        ctx.synchronize()
        torch.cuda.synchronize()
        if self.on:
            import torch.distributed as dist
            dist.barrier()

    def max_time(self, elapsed):
        if not self.on:
            return elapsed
        import torch
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=self.reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_counts(self, counts):
        # the only collective of the design: alive-particle counters, for reporting
        from bevy_hanabi_amd import sharding
        return sharding.allreduce_alive(counts, device=self.reduce_device) if self.on else [int(c) for c in counts]

    def close(self):
        if self.on:
            import torch.distributed as dist
            dist.destroy_process_group()


def run_config(name, args, D, strong=False):
    """Runs one configuration on this rank; returns the result dict on rank 0 (None elsewhere)."""
    import bevy_hanabi_amd as bh
    from bevy_hanabi_amd import effects, sharding

    cfg = CONFIGS[name]
    n = D.world
    steps, warmup = args.steps, args.warmup
    base_cap = args.capacity or cfg["capacity"]
    ctx = bh.Context(D.device_index)
    per_inst_cap = base_cap
    spawn_plan = None  # c5: per-frame spawn counts
    xf_of = None

    if name == "c4":
        inst_per_gpu = args.instances or cfg["instances"]
        total_inst = inst_per_gpu if strong else inst_per_gpu * n
        mine = sharding.instance_plan(total_inst, n)[D.rank]          # instance i -> rank i mod N
        asset = effects.instancing(per_inst_cap)
        prog = ctx.create_program(bh.lower(asset))
        fxs = [prog.create_effect() for _ in mine]
        gids = list(mine)
        gid_mix = (np.asarray(gids, dtype=np.uint64) * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)   # instance_seed(f, i), vectorised over i
        xf_of = np.array([instance_transform(i) for i in gids], dtype=np.float32)
        local_particles = per_inst_cap * len(fxs)
        sharding_desc = f"by instance: {total_inst} instances over {n} rank(s), {len(fxs)} on rank 0"
    else:
        if strong:
            total_cap = base_cap
            slot_base, per_inst_cap = sharding.slab_plan(total_cap, n)[D.rank]
        else:
            total_cap = base_cap * n
            slot_base = sharding.slab_plan(total_cap, n)[D.rank][0]   # rank g owns global slots [g*cap, (g+1)*cap)
        asset = {"c2": effects.firework_trails, "c3": effects.force_field, "c5": effects.ribbon}[name](per_inst_cap)
        prog = ctx.create_program(bh.lower(asset))
        fxs = [prog.create_effect(slot_base=slot_base)]
        gids = [0]
        local_particles = per_inst_cap
        sharding_desc = f"capacity slab x{n}"
        if name == "c5":
            sp = bh.EffectSpawner(asset.spawner)
            rng = bh.Pcg32()
            warmup = max(warmup, 120)    # 1.5 s lifetime at 60 Hz: 90 frames to reach the steady state
            spawn_plan = [sp.tick(DT, rng) for _ in range(1 + warmup + steps)]

    dt = frame_dt(1 + warmup + steps) if name == "c2" else DT

    def spawn_of(f):
        if spawn_plan is not None:
            return spawn_plan[f]
        return per_inst_cap if f == 0 else 0

    def step(f):
        ctx.frame_begin(dt, f * dt)
        if name == "c4":
            s = spawn_of(f)
            # (numpy, not a Python loop over 512 instances: the harness must not be what the step waits for)
            prog.set_frames(np.full(len(fxs), s, dtype=np.uint32), (gid_mix ^ np.uint64(frame_seed(f))).astype(np.uint32), xf_of)
        elif name == "c5":   # the emitter moves (ribbon.rs Shape::tick, Lissajou)
            t = f * dt * 6.5
            fxs[0].set_frame(spawn_of(f), frame_seed(f), [1, 0, 0, 25.0 * np.cos(3.0 * t), 0, 1, 0, 25.0 * np.sin(2.0 * t), 0, 0, 1, 0.0])
        else:
            fxs[0].set_frame(spawn_of(f), frame_seed(f))
        ctx.simulate()

    # warm-up: frame 0 is the burst (k_init + k_update) for c2/c3/c4, then untimed frames
    ctx.enable_kernel_timing(1)
    step(0)
    init_ms = ctx.kernel_timing()["init_ms_avg"]
    ctx.enable_kernel_timing(0)
    for f in range(1, warmup + 1):
        step(f)
    D.barrier(ctx)
    alive0 = sum(fx.alive_count() for fx in fxs)
    ctx.enable_kernel_timing(TIMING_PERIOD)
    D.barrier(ctx)
    t0 = time.perf_counter()
    for f in range(warmup + 1, warmup + 1 + steps):
        step(f)
    D.barrier(ctx)
    elapsed = time.perf_counter() - t0
    timing = ctx.kernel_timing()
    ctx.enable_kernel_timing(0)
    alive1 = sum(fx.alive_count() for fx in fxs)
    last_max_update = sum(fx.metadata()["max_update"] for fx in fxs[:8]) if name == "c5" else None
    kinfo = prog.kernel_info().split("\n")[0]
    ctx.close()

    elapsed = D.max_time(elapsed)
    alive0_total, alive1_total = D.sum_counts([alive0, alive1])
    if name in ("c2", "c4"):
        expect = local_particles if not D.on else None
        assert expect is None or (alive0 == expect and alive1 == expect), f"{name}: expected every particle alive during the timed frames, got {alive0}, {alive1}"
    if D.rank != 0:
        return None
    # particles processed by the update stage per frame (max_update). c2/c3/c4: constant when nothing dies; c5 (churn): the
    # update processes alive-before + this frame's spawns, reported by the last frame's metadata in the steady state.
    if name == "c5":
        per_frame_local = float(last_max_update)
        per_frame_total = per_frame_local * (alive1_total / max(alive1, 1))
    else:
        per_frame_total = (alive0_total + alive1_total) / 2.0
        per_frame_local = (alive0 + alive1) / 2.0
    updates = per_frame_total * steps
    value = updates / elapsed
    bpu = cfg["bytes_per_update"]
    k_ms = timing["update_ms_avg"]
    achieved = per_frame_local * bpu / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    traffic, traffic_src = load_traffic(name, per_inst_cap, len(fxs))
    out = {
        "metric": "particle-updates/sec", "value": value, "unit": "particle-updates/s", "n_gpus": n,
        "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": cfg["workload"].format(cap=per_inst_cap, inst=len(fxs)), "name": name, "capacity_per_gpu": local_particles,
                   "instances_per_gpu": len(fxs), "dt": dt, "sharding": sharding_desc, "alive_before": alive0_total, "alive_after": alive1_total,
                   "updates_per_frame": per_frame_total},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_unit": "B/launch", "traffic_source": traffic_src,
                     "kernel": cfg["kernel"], "kernel_ms_avg": k_ms, "lists_ms_avg": timing["compact_ms_avg"],
                     "kernel_samples": timing["frames"], "timing": f"HIP events on the simulation stream, every {TIMING_PERIOD}th timed frame (rank 0)",
                     "bytes_per_update": bpu, "hbm_gbs_whole_step": updates / n * bpu / elapsed / 1e9,
                     "frac_of_aggregate_peak_whole_step": updates * bpu / elapsed / 1e9 / (HBM_PEAK_GBS * n)},
        "kernels": kinfo,
    }
    if init_ms > 0 and name != "c5":  # the burst frame's init kernel (not part of the metric)
        bps = cfg["bytes_per_spawn"]
        out["init"] = {"kernel": "k_init", "kernel_ms": init_ms, "spawned": local_particles, "bytes_per_spawn": bps,
                       "achieved_gbs": local_particles * bps / (init_ms * 1e-3) / 1e9, "frac": local_particles * bps / (init_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    return out


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--scaling", choices=["weak", "strong", "both"], default="both")
    ap.add_argument("--capacity", type=int, default=None, help="particles per effect instance (default: the configuration's)")
    ap.add_argument("--instances", type=int, default=None, help="c4: instances per GPU (weak) / in total (strong); default 512")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scene", action="store_true", help="N = 1: append the small-effects scene (26 example effects in one context) as \"small_effects_scene\"")
    ap.add_argument("--no-extra-configs", action="store_true", help="N = 1, c2: do not append the c3/c4/c5 lines under \"configs\"")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for dry runs)")
    ap.add_argument("--force-device", type=int, default=None, help="dry runs: every rank uses this GPU instead of LOCAL_RANK")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    D = Dist(args)
    if args.gpus != D.world and D.rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE={D.world}", file=sys.stderr)

    out = None
    if args.scaling in ("weak", "both") or not D.on:
        out = run_config(args.config, args, D, strong=False)
    if D.on and args.scaling in ("strong", "both"):
        s = run_config(args.config, args, D, strong=True)
        if D.rank == 0:
            if out is None:
                out = s
            else:
                out["strong"] = {"value": s["value"], "ms_per_step": s["ms_per_step"], "capacity_per_gpu": s["config"]["capacity_per_gpu"],
                                 "instances_per_gpu": s["config"]["instances_per_gpu"], "kernel_ms_avg": s["roofline"]["kernel_ms_avg"],
                                 "roofline_frac": s["roofline"]["frac"], "frac_of_aggregate_peak_whole_step": s["roofline"]["frac_of_aggregate_peak_whole_step"],
                                 "workload": "the N = 1 workload split over the ranks"}
    if not D.on and args.config == "c2" and not args.no_extra_configs:
        # the other single-GPU configurations of BASELINE.json, same process, short runs: kernel time + roofline each
        extra = {}
        for name in ("c3", "c4", "c5"):
            sub = argparse.Namespace(**vars(args))
            sub.capacity, sub.steps, sub.warmup = None, min(args.steps, 30), min(args.warmup, 5)
            try:
                r = run_config(name, sub, D)
                extra[name] = {"value": r["value"], "ms_per_step": r["ms_per_step"], "workload": r["config"]["workload"],
                               "updates_per_frame": r["config"]["updates_per_frame"], "roofline": r["roofline"], "init": r.get("init"), "kernels": r["kernels"]}
            except Exception as e:  # the headline line must not be lost to a side configuration
                extra[name] = {"error": f"{type(e).__name__}: {e}"}
        out["configs"] = extra
    if not D.on and args.scene:
        # the launch-bound end of the path: 26 different small effects in one context (tools/scene_bench.py, profiles/r02u_scene.md). Opt-in: the
        # small effects share kernel instantiations with the headline workload, and the default command's rocprofv3 statistics are
        # meant to show the headline's kernel on its own
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import scene_bench
            out["small_effects_scene"] = scene_bench.run(1, 300, device=D.device_index, quiet=True)
        except Exception as e:
            out["small_effects_scene"] = {"error": f"{type(e).__name__}: {e}"}
    if D.rank == 0 and not args.no_cpu_baseline and not D.on and args.config == "c2":
        # rank 0 at N = 1 only: the host cores are shared by the ranks otherwise
        try:
            out["cpu_baseline"] = cpu_baseline(args.capacity or CONFIGS["c2"]["capacity"])
        except Exception as e:
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    if D.rank == 0:
        print(json.dumps(out), flush=True)
    D.close()


if __name__ == "__main__":
    main()
