#!/usr/bin/env python3
"""Benchmark of the particle hot path on MI355X: particle-updates/sec + achieved HBM GB/s (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config NAME] [--scaling weak|strong|both]

A "step" is one simulated frame (one pass of the hot path: per-frame inputs upload, init where the frame spawns, update +
kill + list maintenance, ribbon sort where the layout has RIBBON_ID) over particle state resident in HBM. After W warm-up
frames the script times WINDOWS (default 25) windows of exactly K steps each, every one bracketed by a barrier and a device
synchronisation; `ms_per_step` / `value` are the MEDIAN window, the others are listed under "windows" (min, all).

Configurations (synthetic scalings of the reference's example assets):
  c2 (default, the headline)  examples/firework.rs `trails` effect, capacity 16,777,216 per GPU, burst spawn during warm-up,
                              every particle alive in the timed frames (BASELINE.json configs[1]); sharded by CAPACITY SLAB; the
                              LIBRARY'S DEFAULT OPTIONS (HNB_AGE_COHORT_AUTO: the asset's render modifiers read AGE, so every plane -
                              AGE included - is current after every frame; asserted through hnb_effect_device_view);
  c2_mixed                    the same program and capacity in its GENERAL state: a rate spawner of capacity / mean lifetime
                              particles per second in steady state - per-particle ages, lifetimes loaded, spawns into
                              recycled slots and deaths in every frame, list kernels in every frame;
  c2_dieoff                   frames 48..70 of the burst (1/60 s frames): the die-off, 4 % of the capacity lost per frame;
  c2_reburst                  SpawnerSettings::burst(capacity, period) (src/spawn.rs:472) as a 4-frame cycle, every frame timed: the RE-BURST of
                              16,777,216 particles into the dead stack the previous die-off left in killing order (vfx_init.wgsl:141-143),
                              then three frames of 0.45 s in which a third of them and then all of them die;
  c2_lean                     c2 with HNB_AGE_COHORT_LEAN (a headless host that never reads AGE on the device: the plane is stale
                              between hnb_effect_materialise calls) - what rounds 2-5 reported as the headline;
  c2_events                   the REAL examples/firework.rs at scale: three linked effects - rockets whose update emits GPU spawn
                              events, a sparkle trail (5 events per rocket and frame) and the trails (1000 events per dying rocket,
                              capacity 16,777,216) - in steady state, event buffers sized for it;
  c2_interop                  c2 with HNB_AGE_COHORT_OFF: per-particle ages in the plane (what AUTO gives effects below 65,536 slots);
  c2_view                     c2 + a stand-in renderer (a consumer kernel through hnb_effect_device_view) behind every frame, reported apart;
  c3                          examples/force_field.rs, capacity 8,388,608 per GPU, burst, capacity slabs;
  c4                          examples/instancing.rs: independent instances x 65,536, sharded BY INSTANCE
                              (sharding.instance_plan: instance i on rank i mod N); 512 instances per GPU, i.e. BASELINE's
                              4096 instances at N = 8 (`--instances 4096` puts the whole configuration on one GPU);
  c5                          examples/ribbon.rs, capacity 4,194,304 per GPU, rate spawner in steady spawn/kill churn,
                              ribbon sort included in the step.
At N = 1 the default command runs c2 as the JSON line and appends the others under "configs".

Roofline (every configuration): the dominant kernel's HBM-side traffic is MEASURED in this run - the script re-runs itself
twice under `rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE` (separate passes; FETCH_SIZE doubled as
guides/MI355X_MICROARCH.md prescribes for gfx950), cutting the dispatch list at marker kernels - and
    roofline.achieved = measured bytes per launch / the kernel's average duration (HIP events, this process),
    roofline.frac     = achieved / 8 TB/s                      (never above 1: these are bytes that crossed the fabric),
    roofline.algorithmic = SURVEY.md §8(d)'s bytes per update (68 B for the firework) x updates / the same time: what the
                        launch is WORTH, which exceeds what it moves where the design elides traffic (age cohorts,
                        lifetime culling, no list traffic in frames without casualties).
Without rocprofv3 the committed profiles/traffic.json is used if its kernel-source stamp matches this tree, else the
fraction falls back to a stated per-configuration byte model (`traffic_source` says which).

Parity gate (BASELINE.md §3 "same run"): before a configuration's number is accepted, the state its timed frames produced is compared bit
for bit with the CPU oracle (oracle/, the restatement of the reference's WGSL semantics): burst configurations (c2, c2_interop, c3, c4) on one
slab of slots of the FULL-SIZE effect after all timed frames (a burst fills slot i with PRNG stream i: the slab IS a small effect with that
slot_base); churn configurations (c2_mixed, c2_dieoff, c2_events, c5) by replaying the same regime (same dt, warm-up and frame count) at a reduced
capacity against the oracle, full state - AND, on the state the timed frames themselves left at full size: hnb_effect_check (fault flag, the two
lists a permutation of the slots, alive bytes, age < lifetime, all on the device) and hnb_effect_compare against a second context that replayed the
same frames with every proof and hint switched off (PLAIN_OPTIONS), bit for bit. Slab / capacity are sized to the oracle's measured speed (a few
seconds per configuration). The line carries "parity": {"checked": [...], "ok": true}; if a comparison fails the line carries "value": null and
the exit code is 1.

Output: the LAST stdout line is the short result (<= 4 KB: the c2 headline with roofline, cpu_baseline, parity and a one-row summary of every
other configuration); the complete record (every window, stage, counter and per-kernel figure) goes to profiles/bench_full.json (and
gpurun_out/bench_full.json when that directory exists) and to stderr.

N > 1: `python bench.py --gpus N` launches itself under `python -m torch.distributed.run` (one process per GPU, RCCL); when the
driver already started it that way (WORLD_SIZE in the environment) it just runs its rank. There is no data-path collective: the
only collective is the all-reduce of the alive-particle counters for reporting - through the product's own communicator (rank 0 takes an id from
hnb_comm_unique_id, it travels over the torch process group, every rank calls hnb_comm_create_rank; hnb_comm_allreduce_alive = one grouped
ncclAllReduce of librccl on the simulation stream) - plus the MAX of the elapsed times (torch.distributed). The JSON
line is the weak-scaling run (per-GPU work fixed); with N > 1 a strong-scaling run (the N = 1 workload split over the ranks) is
reported inside it under "strong".
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import shutil
import socket
import statistics
import subprocess
import sys
import tempfile
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
MEAN_LIFETIME = 1.0
TIMING_PERIOD = 15  # HIP events bracket the kernels of every 15th timed frame (each costs ~20 us of stream bubbles: at every 5th frame, as until round 5, that was 5 % of a C5 step and 2 % of a c2 step - profiles/r06f_timing_period.log; 750 timed frames still give 50 samples)
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")  # written by `bench.py --write-traffic` from the PMC passes
REBURST_CYCLE, REBURST_DT = 4, 0.45   # c2_reburst: frame 0 of a cycle bursts `capacity` particles (dt 1/60), frames 1..3 advance 0.45 s each: alive after them 100 %, ~71 %, 0
DIEOFF_FIRST, DIEOFF_LAST, DIEOFF_END = 48, 70, 76   # c2_dieoff: timed frames [48, 70]; by frame 76 nothing is alive (1.2 s < 73 / 60 s)

# algorithmic bytes per particle update, SURVEY.md §8(d): attributes read + attributes written + 8 B alive-list entry.
# model_bytes: what the dominant kernel of the configuration is DESIGNED to move per updated particle (fallback when no PMC pass
# is available; DESIGN.md "Roofline" derives each figure).
CONFIGS = {
    "c2": dict(capacity=1 << 24, bytes_per_update=68, bytes_per_spawn=44, model_bytes=52, kernel="k_update_slots_stream<ProgDragAccel, cohort> (age_current)",
               workload="firework.rs trails EffectAsset, capacity={cap:_} per GPU, burst spawner, all particles alive, library default options"),
    "c2_lean": dict(capacity=1 << 24, bytes_per_update=68, bytes_per_spawn=44, model_bytes=48, kernel="k_update_slots_stream<ProgDragAccel, cohort>",
                    workload="as c2 with hnb_ctx_set_option(HNB_OPT_AGE_COHORT, LEAN): a headless host, the AGE plane is stale between hnb_effect_materialise calls (the headline of rounds 2-5)"),
    "c2_reburst": dict(capacity=1 << 24, bytes_per_update=68, bytes_per_spawn=44, model_bytes=62, kernel="k_update_slots_stream<ProgDragAccel, cohort> (age_current)",
                       workload="firework.rs trails EffectAsset, capacity={cap:_} per GPU, SpawnerSettings::burst(capacity, period) as a 4-frame cycle, all frames timed: re-burst into the "
                                "dead stack of the previous die-off (slot-major k_init_slots), then 3 frames of 0.45 s: a third dies, then everybody"),
    "c2_mixed": dict(capacity=1 << 24, bytes_per_update=68, bytes_per_spawn=44, model_bytes=62, kernel="k_update_slots_stream<ProgDragAccel, cohort>",
                     workload="firework.rs trails EffectAsset, capacity={cap:_} per GPU, rate spawner (capacity / mean lifetime per second) in steady state: "
                              "mixed ages, spawns into recycled slots + deaths + list kernels every frame"),
    "c2_dieoff": dict(capacity=1 << 24, bytes_per_update=68, bytes_per_spawn=44, model_bytes=62, kernel="k_update_slots_stream<ProgDragAccel, cohort>",
                      workload="firework.rs trails EffectAsset, capacity={cap:_} per GPU, burst; frames 48..70 at 1/60 s: the die-off (list kernels every frame)"),
    "c2_interop": dict(capacity=1 << 24, bytes_per_update=68, bytes_per_spawn=44, model_bytes=56, kernel="k_update_slots_stream<ProgDragAccel>",
                       workload="firework.rs trails EffectAsset, capacity={cap:_} per GPU, burst, all alive, hnb_ctx_set_option(HNB_OPT_AGE_COHORT, OFF) (AGE plane current every frame)"),
    "c2_view": dict(capacity=1 << 24, bytes_per_update=68, bytes_per_spawn=44, model_bytes=52, kernel="k_update_slots_stream<ProgDragAccel, cohort> (age_current)",
                    workload="c2 (library defaults) + a stand-in RENDERER behind every frame: a consumer kernel that gathers position / age / lifetime by list row through hnb_effect_device_view "
                             "(not simulation time: `consumer_ms` is its own cost, `sim_only_ms` the step without it)"),
    "c2_events": dict(capacity=1 << 24, bytes_per_update=68, bytes_per_spawn=44, model_bytes=62, kernel="k_update_slots_stream<ProgDragAccel, cohort> (trails)",
                      workload="the real examples/firework.rs: rocket (capacity 32_768, 16_000 rockets/s) -> sparkle_trail (1_048_576; 5 spawn events per rocket and frame) + "
                               "trails (capacity={cap:_}; 1000 spawn events per dying rocket), GPU spawn events, steady state"),
    "c3": dict(capacity=1 << 23, bytes_per_update=68, bytes_per_spawn=40, model_bytes=56, kernel="k_update_slots_stream<ProgForceField>",
               workload="force_field.rs EffectAsset (2x ConformToSphere + KillAabb + KillSphere), capacity={cap:_} per GPU, burst"),
    "c4": dict(capacity=65536, bytes_per_update=68, bytes_per_spawn=40, model_bytes=36, kernel="k_update_slots_stream<ProgAgeEuler, cohort>", instances=512,
               workload="instancing.rs: {inst} independent effect instances x {cap:_} per GPU (one launch), burst, all alive"),
    "c5": dict(capacity=1 << 22, bytes_per_update=20, bytes_per_spawn=36, model_bytes=9, kernel="k_update_slots_stream_age",
               workload="ribbon.rs EffectAsset, capacity={cap:_} per GPU, rate spawner in steady spawn/kill churn, ribbon sort in the step"),
}
EXTRA_CONFIGS = ("c2_lean", "c2_interop", "c2_view", "c2_mixed", "c2_dieoff", "c2_reburst", "c2_events", "c3", "c4", "c5")   # appended to the N = 1 default line
# what identifies the dominant kernel of a configuration in a rocprofv3 dispatch list
KERNEL_MATCH = {"c2_events": ("k_update_slots_stream",), "c2": ("k_update_slots_stream",), "c2_mixed": ("k_update_slots_stream",), "c2_dieoff": ("k_update_slots_stream",),
                "c2_interop": ("k_update_slots_stream",), "c2_view": ("k_update_slots_stream",), "c2_lean": ("k_update_slots_stream",), "c2_reburst": ("k_update_slots_stream",), "c3": ("k_update_slots_stream",), "c4": ("k_update_slots_stream",), "c5": ("k_update_slots_stream",)}


def frame_dt(total_frames, safe_seconds=MIN_LIFETIME * 0.95):
    """1/60 s like the reference's example; shrunk only if a long run would outlive the youngest
    particle (the c2 metric is defined on frames where all particles are alive)."""
    return DT if total_frames * DT < safe_seconds else safe_seconds / total_frames


# c3 / c4 (lifetimes 10 s and 12 s, c3 also kills by position): the simulated time within which the burst is known to stay complete - what
# 506 frames of 1/60 s cover. (Round 4 raised the default to 25 windows of 30 steps = 756 frames = 12.6 s: c4's particles were all dead and half
# of c3's before the last window; every run of that round that was looked at had been started with --steps 20.)
BURST_SAFE_SECONDS = {"c3": 8.4, "c4": 8.4}


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


def kernel_source_stamp():
    """sha256 over the sources every kernel of the library is built from: ties a recorded traffic figure to a kernel build."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "bevy_hanabi_amd", "csrc")
    for name in ("hnb_math.h", "hnb_vm.h", "hnb_dev.h", "hnb_kernels.hip.h", "hnb_sort.hip.h", "hnb_jit.h", "hanabi_amd.hip"):
        with open(os.path.join(d, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=5).stdout.strip() or None
    except Exception:
        return None


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline
# ------------------------------------------------------------------------------------------------------------------
def host_physical_cores():
    try:
        cores, phys, core = set(), None, None
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("physical id"):
                    phys = ln.split(":")[1].strip()
                elif ln.startswith("core id"):
                    core = ln.split(":")[1].strip()
                elif not ln.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
        return len(cores) or None
    except OSError:
        return None


def cpu_limits():
    """What this process may actually use of the host: the affinity mask and the container's CPU quota (cgroup v2 cpu.max / v1 cfs quota). A box whose
    container is capped at a few dozen CPUs' worth of time makes an OpenMP run with one thread per physical core SLOWER than one with a quarter of
    them (round 4: 32 threads 2.3e10, 64 1.0e10, 128 5.2e9 updates/s on 2 x 64 cores): the threads beyond the quota are throttled, not idle."""
    out = {"affinity_cpus": None, "cgroup_cpu_quota": None}
    try:
        out["affinity_cpus"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
            out["cgroup_cpu_quota"] = None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, p = int(f.read()), int(g.read())
                out["cgroup_cpu_quota"] = None if q <= 0 else q / p
        except (OSError, ValueError):
            pass
    return out


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
    limits = cpu_limits()
    usable = min(x for x in (hw, limits["affinity_cpus"], limits["cgroup_cpu_quota"]) if x)
    tried = {}
    # the timed frames use a small dt so that no particle reaches its lifetime however many frames are timed (the arithmetic
    # per frame is the same); every thread count gets a fresh first-touch copy of the state
    ops = ops_for(1e-4)
    # (one thread per physical core is hw / 2 with SMT; the quota-sized count is tried too: threads beyond a container's CPU quota are throttled)
    for threads in sorted({max(1, hw // 8), max(1, hw // 4), max(1, hw // 2), max(1, int(usable)), max(1, int(usable) // 2)}):
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
    # `cores`: what the run could actually use - the thread count, capped by the container's CPU quota (on this pool's GPU boxes cgroup cpu.max allows 16 CPUs'
    # worth of time on a 2 x 64-core host: 64 threads then run at a quarter speed each, which is why more threads stopped helping in every earlier round)
    quota = limits.get("cgroup_cpu_quota")
    cores = int(min(threads, quota)) if quota else threads
    pool = None
    try:   # the same measurement on other boxes of the pool (committed records): this one is a sample of a 5x spread, not a constant of the machine type
        with open(os.path.join(ROOT, "profiles", "cpu_baseline_boxes.json")) as f:
            vals = [r["value"] for r in json.load(f)["records"]] + [tried[threads]]
        pool = {"best": max(vals), "worst": min(vals), "boxes": len(vals), "source": "profiles/cpu_baseline_boxes.json + this run"}
    except (OSError, ValueError, KeyError):
        pass
    return {"value": tried[threads], "unit": "particle-updates/s", "cores": cores, "threads": threads, "host_logical_cpus": hw, "pool": pool,
            "host_physical_cores": host_physical_cores(), "kind": "port",
            "algorithmic_gbs": tried[threads] * CONFIGS["c2"]["bytes_per_update"] / 1e9, "bytes_per_update": CONFIGS["c2"]["bytes_per_update"],
            "threads_tried": {str(k): v for k, v in tried.items()}, "limits": limits,
            "sample": f"{capacity} particles x {frames} frames (best of 3 repeats) of the same firework update (all alive), packed-SoA OpenMP "
                      f"port (oracle/cpu_soa.c, -O3 -march=native, threads bound {os.environ['OMP_PROC_BIND']}/{os.environ['OMP_PLACES']}); "
                      f"checked bit-equal to the oracle on all {capacity} particles x {check_frames} frames first; {time.perf_counter() - t_all:.1f} s in total. "
                      "`threads` = the OpenMP thread count that was fastest, `cores` = that count capped by the container's CPU quota (`limits`); the 0.5 GB of state fits in the host's last-level caches (2 x 256 MB L3 on the GPU box)"}


# ------------------------------------------------------------------------------------------------------------------
# distributed plumbing
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

    def make_comm(self, w):
        """N > 1: the product's OWN communicator for this workload's context: rank 0 takes a unique id from hnb_comm_unique_id,
        the 128 bytes travel over the torch process group that already exists (it also carries the MAX of the elapsed times, nothing else), every rank calls
        hnb_comm_create_rank: ncclCommInitRank of librccl (or of --comm-lib, for dry runs of several ranks on one GPU)."""
        if not self.on:
            return None
        import torch
        import torch.distributed as dist
        # (a rank whose communicator cannot be created must not leave the others inside ncclCommInitRank or kill the measurement: every rank learns
        # whether every rank succeeded; if not, the totals fall back to the torch process group and the line says so - `comm.error`)
        err = None
        try:
            box = [w.bh.Comm.unique_id() if self.rank == 0 else None]
        except Exception as e:   # noqa: BLE001
            box, err = [None], f"hnb_comm_unique_id: {e}"
        dist.broadcast_object_list(box, src=0)
        comm = None
        if box[0] is not None:
            try:
                comm = w.bh.Comm.rank(w.ctx, box[0], self.rank, self.world)
            except Exception as e:   # noqa: BLE001
                err = f"hnb_comm_create_rank: {e}"
        else:
            err = err or "rank 0 could not create a unique id"
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int64, device=self.reduce_device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if comm is not None:
                comm.destroy()
            self.comm_error = err or "another rank could not create its communicator"
            return None
        return comm

    def alive_total(self, w):
        """The only collective of the design: the alive-particle counters, for reporting. N > 1: hnb_comm_allreduce_alive (one grouped ncclAllReduce of
        n_effects x u64 on the simulation stream) over the workload's communicator; every rank passes the same number of entries (NULL = 0)."""
        if not self.on:
            return w.alive()
        if w.comm is None:   # (make_comm failed on some rank: the measurement goes on, the line reports it)
            import torch
            import torch.distributed as dist
            t = torch.tensor([w.alive()], dtype=torch.int64, device=self.reduce_device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return int(t.item())
        fxs = list(w.fxs) + [None] * (w.n_effects_max - len(w.fxs))
        return int(sum(w.comm.allreduce_alive([fxs])))

    def close(self):
        if self.on:
            import torch.distributed as dist
            dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------------
# one configuration on one rank
# ------------------------------------------------------------------------------------------------------------------
class Workload:
    """One configuration set up on one rank: builds the effect(s), knows how to play frame f."""

    def __init__(self, name, args, D, strong=False, options=None):
        import bevy_hanabi_amd as bh
        from bevy_hanabi_amd import effects, sharding

        self.name, self.D, self.bh = name, D, bh
        cfg = self.cfg = CONFIGS[name]
        n = D.world
        base_cap = args.capacity or cfg["capacity"]
        self.ctx = bh.Context(D.device_index)
        # Every configuration runs the library's DEFAULT options (HNB_AGE_COHORT_AUTO: the plane of every attribute the asset's render modifiers read is
        # current after every frame) except the two that are named for the option they set:
        self.options = "default"
        if name == "c2_interop":
            self.ctx.set_option("age_cohort", 0)   # HNB_AGE_COHORT_OFF, fixed in the program at creation: per-particle ages in the plane
            self.options = "age_cohort=OFF"
        elif name == "c2_lean":
            self.ctx.set_option("age_cohort", 1)   # HNB_AGE_COHORT_LEAN: a headless host, nobody reads AGE on the device between frames
            self.options = "age_cohort=LEAN"
        for k, v in (options or {}).items():   # (the parity gate's plain replay: every proof and hint off; a test's broken proof)
            self.ctx.set_option(k, v)
        self.per_inst_cap = base_cap
        self.spawner = self.rng = None
        self.xf_of = None
        self.family = "c2" if name.startswith("c2") else name

        self.shadow = None       # parity gate: callable(f, dt, [(spawn, seed, transform) per effect]) stepped in lockstep with the GPU
        self.assets, self.event_caps, self.slot_base = [], [None, None, None], 0
        if name == "c4":
            inst_per_gpu = args.instances or cfg["instances"]
            total_inst = inst_per_gpu if strong else inst_per_gpu * n
            mine = sharding.instance_plan(total_inst, n)[D.rank]          # instance i -> rank i mod N
            asset = effects.instancing(self.per_inst_cap)
            self.assets = [asset]
            self.gids = list(mine)
            self.prog = self.ctx.create_program(bh.lower(asset))
            self.fxs = [self.prog.create_effect() for _ in mine]
            gids = list(mine)
            self.gid_mix = (np.asarray(gids, dtype=np.uint64) * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)   # instance_seed(f, i), vectorised over i
            self.xf_of = np.array([instance_transform(i) for i in gids], dtype=np.float32)
            self.local_particles = self.per_inst_cap * len(self.fxs)
            self.sharding_desc = f"by instance: {total_inst} instances over {n} rank(s), {len(self.fxs)} on rank 0"
        elif name == "c2_events":
            # examples/firework.rs:41-251 scaled so that the trails effect fills its 16,777,216 slots to ~95 %: 16,000 rockets per second,
            # each alive for 0.8 .. 1.2 s, 1000 trail particles per explosion living 0.8 .. 1.2 s. Parent and children share the context (one GPU).
            cap = self.per_inst_cap
            self.rocket_asset = effects.firework_rocket(32768, 5, 1000)
            self.rocket_asset.spawner = bh.SpawnerSettings.rate(16000.0 * cap / (1 << 24))
            assets = [self.rocket_asset, effects.firework_sparkle_trail(max(4096, cap // 16)), effects.firework_trails_child(cap)]
            self.assets = assets
            self.event_caps = [None, (0, 0, max(4096, cap // 128)), (0, 1, max(4096, cap // 32))]   # (parent, channel, event capacity)
            self.progs = [self.ctx.create_program(bh.lower(a)) for a in assets]
            self.prog = self.progs[2]
            self.fxs = [p.create_effect() for p in self.progs]
            self.fxs[1].set_parent(self.fxs[0], 0, self.event_caps[1][2])     # ~84k sparkle events per frame at full size
            self.fxs[2].set_parent(self.fxs[0], 1, self.event_caps[2][2])     # ~270k explosion events per frame at full size
            self.local_particles = cap
            self.sharding_desc = "one context (a parent and its children live on one GPU); replicas at N > 1"
            self.spawner, self.rng = bh.EffectSpawner(self.rocket_asset.spawner), bh.Pcg32()
        else:
            if strong:
                total_cap = base_cap
                slot_base, self.per_inst_cap = sharding.slab_plan(total_cap, n)[D.rank]
            else:
                total_cap = base_cap * n
                slot_base = sharding.slab_plan(total_cap, n)[D.rank][0]   # rank g owns global slots [g*cap, (g+1)*cap)
            cap = self.per_inst_cap
            if name == "c2_mixed":
                asset = effects.firework_trails(cap, bh.SpawnerSettings.rate(float(cap) / MEAN_LIFETIME))
            elif self.family == "c2":
                asset = effects.firework_trails(cap)
            else:
                asset = {"c3": effects.force_field, "c5": effects.ribbon}[name](cap)
            self.prog = self.ctx.create_program(bh.lower(asset))
            self.fxs = [self.prog.create_effect(slot_base=slot_base)]
            self.assets, self.slot_base = [asset], slot_base
            self.local_particles = cap
            self.sharding_desc = f"capacity slab x{n}"
            if name in ("c5", "c2_mixed"):
                self.spawner, self.rng = bh.EffectSpawner(asset.spawner), bh.Pcg32()
        self.n_effects_max = len(self.fxs) if name != "c4" else max(len(g) for g in sharding.instance_plan(total_inst, n))   # (entries of the all-reduce: the same on every rank)
        self.comm = None
        self.consumer = self.consumer_out = None
        if name.endswith("_view"):   # a renderer's vertex-stage reads behind every frame (tests/device_view/consumer.hip: consumer_render_like)
            import ctypes as C
            from bevy_hanabi_amd import runtime
            lib = C.CDLL(os.path.join(ROOT, "tests", "device_view", "libconsumer.so"))
            lib.consumer_render_like.argtypes = [C.POINTER(runtime.DeviceView), C.c_void_p]
            self.consumer, self._byref = lib, C.byref
            self.consumer_out = torch.empty(self.per_inst_cap * 4, dtype=torch.float32, device=torch.device("cuda", D.device_index))
            torch.cuda.synchronize()
        self.dt = DT
        self.f = 0   # next frame index

    # -- frame inputs -------------------------------------------------------------------------------------------
    def spawn_of(self, f):
        if self.spawner is not None:
            return self.spawner.tick(self.dt, self.rng)
        if self.name == "c2_dieoff":
            return self.per_inst_cap if f % DIEOFF_END == 0 else 0      # a burst every DIEOFF_END frames: every pass replays the same die-off
        if self.name == "c2_reburst":
            return self.per_inst_cap if f % REBURST_CYCLE == 0 else 0   # SpawnerSettings::burst(capacity, period): the period is one cycle
        return self.per_inst_cap if f == 0 else 0

    def dt_of(self, f):
        if self.name == "c2_reburst":
            return DT if f % REBURST_CYCLE == 0 else REBURST_DT
        return self.dt

    def inputs_of(self, f, s):
        """Per-frame inputs (spawn count, seed, transform) of the effects of this rank in frame f, given the spawner's count s.
        c4 is vectorised in step(); here only the instances the parity gate asks for (inputs_of_instance)."""
        if self.name == "c2_events":
            return [(s, frame_seed(f), None), (0, frame_seed(1000 + f), None), (0, frame_seed(2000 + f), None)]
        if self.name == "c5":   # the emitter moves (ribbon.rs Shape::tick, Lissajou)
            t = f * self.dt * 6.5
            return [(s, frame_seed(f), [1, 0, 0, 25.0 * np.cos(3.0 * t), 0, 1, 0, 25.0 * np.sin(2.0 * t), 0, 0, 1, 0.0])]
        return [(s, frame_seed(f % DIEOFF_END if self.name == "c2_dieoff" else f), None)]

    def step(self):
        f, ctx = self.f, self.ctx
        dt = self.dt_of(f)
        ctx.frame_begin(dt, f * dt)
        s = self.spawn_of(f)
        if self.name == "c4":
            # (numpy, not a Python loop over 512 instances: the harness must not be what the step waits for)
            self.prog.set_frames(np.full(len(self.fxs), s, dtype=np.uint32), (self.gid_mix ^ np.uint64(frame_seed(f))).astype(np.uint32), self.xf_of)
        else:
            inputs = self.inputs_of(f, s)
            for fx, (sp, seed, xf) in zip(self.fxs, inputs):
                fx.set_frame(sp, seed, xf)
            if self.shadow is not None:
                self.shadow(f, dt, inputs)
        ctx.simulate()
        if self.consumer is not None:   # enqueued on the simulation stream right behind the frame: no synchronisation, no host copy
            v = self.fxs[0].device_view()
            assert v.stale_attr_mask == 0, "HNB_AGE_COHORT_AUTO left AGE stale for an asset whose render modifiers read it"
            rc = self.consumer.consumer_render_like(self._byref(v), self.consumer_out.data_ptr())
            assert rc == 0, f"consumer_render_like: {rc}"
        self.f += 1

    def alive(self):
        return sum(fx.alive_count() for fx in self.fxs)

    def close(self):
        if self.comm is not None:
            self.comm.destroy()
            self.comm = None
        self.ctx.close()


# ------------------------------------------------------------------------------------------------------------------
# parity gate: the state the timed frames produced, against the CPU oracle, before a number is accepted
# ------------------------------------------------------------------------------------------------------------------
PARITY_BUDGET_S = 0.5          # oracle time per configuration (the slab / the reduced capacity is sized to the oracle's measured speed)
PARITY_KEYS = ("capacity", "alive_count", "max_update", "max_spawn", "indirect_write_index", "particle_counter", "instance_count", "dead_count")
BURST_PARITY = ("c2", "c2_lean", "c2_interop", "c2_view", "c3", "c4")
_ORACLE_RATE = None


def oracle_rate():
    """particle-updates/s of the OpenMP oracle on this host (firework update), measured once on 65,536 particles. (The churn regimes cost the
    oracle 4-5x as much per update - spawns, list maintenance, event ordering: PARITY_BUDGET_S is set with that in mind.)"""
    global _ORACLE_RATE
    if _ORACLE_RATE is None:
        import bevy_hanabi_amd as bh
        import oracle
        from bevy_hanabi_amd import effects
        oracle.build()
        n = 65536
        o = oracle.OracleEffect(bh.serialize_asset(effects.firework_trails(n)), omp=True)
        o.step(1e-3, n, 1)
        t0 = time.perf_counter()
        for f in range(1, 5):
            o.step(1e-3, 0, 1 + f, time=f * 1e-3)
        _ORACLE_RATE = 4 * n / max(time.perf_counter() - t0, 1e-6)
        o.close()
    return _ORACLE_RATE


def _first_difference(ref, got, is_float):
    """None if the two u32 planes are equal (two NaNs of a float attribute compare equal whatever their bits: WGSL leaves the pattern
    unspecified and x86 / gfx950 default NaNs differ in sign), else a description of the first differing element."""
    if np.array_equal(ref, got):
        return None
    diff = ref != got
    if is_float:
        isnan = lambda x: ((x & 0x7F800000) == 0x7F800000) & ((x & 0x007FFFFF) != 0)
        diff &= ~(isnan(ref) & isnan(got))
        if not diff.any():
            return None
    i = int(np.argwhere(diff)[0][0])
    return f"{int(diff.sum())} differing words, first at row {i}: oracle {ref[i]} device {got[i]}"


def _stored_attrs(asset):
    return [a for a in asset.particle_layout() if a.id >= 2]


def parity_burst_slab(w, D):
    """c2, c2_interop, c3, c4 after ALL the frames played so far (burst, warm-up, every timed window): slots [B, B + S) of the full-size
    effect (c4: of one instance) against an oracle effect of capacity S and slot_base B that is fed the same frames. A burst of `capacity`
    particles into a fresh effect gives slot i the PRNG stream of particle index slot_base + i and nothing a particle does afterwards
    depends on another slot, so the slab of the big effect IS that small effect (the capacity-slab argument of SURVEY.md 8e)."""
    import bevy_hanabi_amd as bh
    import oracle
    from bevy_hanabi_amd import effects
    t0 = time.perf_counter()
    frames, cap = w.f, w.per_inst_cap
    S = int(PARITY_BUDGET_S * oracle_rate() / max(frames, 1)) // 4096 * 4096
    S = max(4096, min(65536, S, cap // 4096 * 4096 or cap))
    B = min(cap // 2 // 4096 * 4096, cap - S)
    inst = len(w.fxs) // 3 if w.name == "c4" else 0            # c4: a slab of one instance in the middle of the batch
    make = {"c2": effects.firework_trails, "c3": effects.force_field, "c4": effects.instancing}[w.family]
    asset = make(S)
    base = (0 if w.name == "c4" else w.slot_base) + B
    o = oracle.OracleEffect(bh.serialize_asset(asset), base, omp=True)
    xf = None if w.name != "c4" else instance_transform(w.gids[inst])
    for f in range(frames):
        seed = instance_seed(f, w.gids[inst]) if w.name == "c4" else frame_seed(f)
        o.step(w.dt, S if f == 0 else 0, seed, time=f * w.dt, transform=xf)
    fx = w.fxs[inst]
    problems = []
    for a in _stored_attrs(asset):
        d = _first_difference(o.read_attr(a.id).view(np.uint32), fx.read_attr(a.id).view(np.uint32)[B:B + S], a.value_type.elem == bh.ScalarType.Float)
        if d:
            problems.append(f"{a.name}: {d}")
    alive = fx.alive_list()
    mine = np.sort(alive[(alive >= B) & (alive < B + S)] - B)
    if not np.array_equal(mine, np.sort(o.alive_list())):
        problems.append(f"alive slots of the slab: device {len(mine)}, oracle {o.alive_count()}")
    n_alive = o.alive_count()
    o.close()
    return {"config": w.name, "kind": "slab of the full-size effect after the timed frames", "slots": [B, B + S], "instance": inst if w.name == "c4" else None,
            "frames": frames, "alive_in_slab": n_alive, "attrs": [a.name for a in _stored_attrs(asset)], "seconds": time.perf_counter() - t0,
            "ok": not problems, "problems": problems}


# every proof, hint and shortcut hnb_ctx_set_option can switch off: what is left is one init, one update, k_count_rows + k_compact per
# program and frame, per-particle ages and lifetimes, direct spawn stores, one stream
PLAIN_OPTIONS = {"horizon": 0, "age_cohort": 0, "cull_lifetime": 0, "skip_lists": 0, "stream_hints": 0, "overlap_updates": 0,
                 "suffix_proof": 0, "ring_lists": 0, "alternate": 0, "transpose": 0, "scene_merge": 0, "slot_init": 0, "direct_upload": 0}


def parity_timed_state(w, args, D, options=None):
    """c2_mixed, c2_dieoff, c2_events, c5 at FULL size, on the state the timed frames left in `w`:
    (a) hnb_effect_check on every effect - fault flag clear, alive rows + dead rows a permutation of the slots, alive bytes consistent with the
        lists, every alive particle younger than its lifetime - evaluated on the device;
    (b) a second context replays the same frames (same dt, inputs, spawner sequence) with every proof and hint off (PLAIN_OPTIONS) and
        hnb_effect_compare holds the two effects against each other bit for bit: counters, both lists, every plane of every slot.
    The oracle leg (the same regime at reduced capacity, full state) is parity_regime."""
    t0 = time.perf_counter()
    problems = []
    checks = [fx.check() for fx in w.fxs]
    for i, c in enumerate(checks):
        if not c["ok"]:
            problems.append(f"effect {i} invariants: {c}")
    sub = argparse.Namespace(**vars(args))
    sub.capacity = w.per_inst_cap
    w2 = Workload(w.name, sub, D, options=dict(PLAIN_OPTIONS, **(options or {})))
    w2.dt = w.dt
    for _ in range(w.f):
        w2.step()
    w2.ctx.synchronize()
    diffs = [a.compare(b) for a, b in zip(w.fxs, w2.fxs)]
    for i, d in enumerate(diffs):
        if not d["equal"]:
            problems.append(f"effect {i} differs from the plain replay: {d}")
    plain_kernels = w2.prog.kernel_info().split("\n")[0]
    w2.close()
    return {"kind": "timed state: invariants on the device + plain-path differential at full size", "frames": w.f, "capacity": w.per_inst_cap,
            "checks": checks, "diffs": diffs, "plain_kernels": plain_kernels, "seconds": time.perf_counter() - t0, "ok": not problems, "problems": problems}


def parity_regime(name, args, D, frames):
    """c2_mixed, c2_dieoff, c2_events, c5: the same regime - same dt, same warm-up, same number of frames, the same spawner - replayed at a
    reduced capacity on the device and, frame by frame, by the oracle; then the FULL state (counters, both lists, every plane of every slot)."""
    import bevy_hanabi_amd as bh
    import oracle
    t0 = time.perf_counter()
    full = CONFIGS[name]["capacity"] if not args.capacity else args.capacity
    if name == "c2_dieoff":
        frames = min(frames, 2 * DIEOFF_END)     # two passes: the second bursts into the dead list the first die-off left
    if name == "c2_reburst":
        frames = min(frames, 3 * REBURST_CYCLE + 1)   # three cycles and the burst of the fourth: re-bursts into the dead stacks the die-offs left (few frames: a capacity at which the slot-major init engages)
    cap = int(PARITY_BUDGET_S * oracle_rate() / max(frames, 1))
    cap = max(16384, min(1 << (max(cap, 1).bit_length() - 1), max(16384, full // 16)))
    sub = argparse.Namespace(**vars(args))
    sub.capacity = cap
    w = Workload(name, sub, D)
    orcs = [oracle.OracleEffect(bh.serialize_asset(a), w.slot_base, omp=True) for a in w.assets]
    for o, link in zip(orcs, w.event_caps):
        if link is not None:
            o.set_parent(orcs[link[0]], link[1], link[2])

    def shadow(f, dt, inputs):   # the reference's frame order: every init pass, parents first, then every update pass
        for o, (sp, seed, xf) in zip(orcs, inputs):
            o.init_pass(dt, sp, seed, time=f * dt, transform=xf)
        for o, (sp, seed, xf) in zip(orcs, inputs):
            o.update_pass(dt, seed, time=f * dt, transform=xf)

    w.shadow = shadow
    for _ in range(frames):
        w.step()
    w.ctx.synchronize()
    problems, alive = [], []
    for i, (o, fx, asset) in enumerate(zip(orcs, w.fxs, w.assets)):
        m, c = fx.metadata(), o.counters()
        alive.append(c["alive_count"])
        if {k: m[k] for k in PARITY_KEYS} != c:
            problems.append(f"effect {i} counters: device {[m[k] for k in PARITY_KEYS]} oracle {[c[k] for k in PARITY_KEYS]}")
            continue
        if m["fault"]:
            problems.append(f"effect {i}: device fault flag {m['fault']}")
        for what, a, b in (("alive list", o.alive_list(), fx.alive_list()), ("dead list", o.dead_list(), fx.dead_list())):
            d = _first_difference(a, b, False)
            if d:
                problems.append(f"effect {i} {what}: {d}")
        for a in _stored_attrs(asset):
            d = _first_difference(o.read_attr(a.id).view(np.uint32), fx.read_attr(a.id).view(np.uint32), a.value_type.elem == bh.ScalarType.Float)
            if d:
                problems.append(f"effect {i} {a.name}: {d}")
    kinfo = w.prog.kernel_info().split("\n")
    for o in orcs:
        o.close()
    w.close()
    return {"config": name, "kind": "the same regime at reduced capacity, full state", "capacity": cap, "frames": frames, "alive_at_end": alive,
            "kernels": kinfo[0], "seconds": time.perf_counter() - t0, "ok": not problems, "problems": problems}


_COMM_STUCK = False


def comm_alive_total(w, timeout_s=90.0):
    """N = 1: the alive total of the headline configuration through the library's OWN collective - a one-context communicator built with the real
    librccl (hnb_comm_set_library(NULL, HNB_COMM_LIB_SINGLE_RANK) at start-up: ncclCommInitAll over one device, a grouped ncclAllReduce(ncclUint64,
    ncclSum) on the simulation stream) - so that the record of a one-GPU run shows that hnb_comm_* works, not only torch.distributed at N > 1.
    Guarded by a timeout (a collective library that cannot initialise on this host must not cost the line)."""
    global _COMM_STUCK
    import threading
    out = {}

    def body():
        try:
            comm = w.bh.Comm.local([w.ctx])
            desc = comm.describe()
            totals = comm.allreduce_alive([list(w.fxs)])
            comm.destroy()
            kind, _, rest = desc.partition(" ")
            out.update({"library": rest.split(" ranks=")[0] if kind == "rccl" else kind, "ranks": 1, "effects": len(totals), "alive_total": int(sum(totals))})
        except Exception as e:
            out["error"] = f"{type(e).__name__}: {e}"[:200]

    t = threading.Thread(target=body, daemon=True)
    t.start()
    t.join(timeout_s)
    if t.is_alive():
        _COMM_STUCK = True
        return {"error": f"hnb_comm_* did not return within {timeout_s:.0f} s"}
    return out


def warmup_frames(name, requested):
    if name == "c5":
        return max(requested, 120)     # 1.5 s lifetime at 60 Hz: 90 frames to reach the steady state
    if name == "c2_events":
        return max(requested, 300)     # a rocket's life (<= 1.2 s) + a trail particle's (<= 1.2 s) = 144 frames to fill, then as long again to mix
    if name == "c2_mixed":
        return max(requested, 240)     # lifetimes of 0.8 .. 1.2 s: four mean lifetimes until ages, slots and list order are mixed
    return requested


def run_config(name, args, D, strong=False, pmc=None):
    """Runs one configuration on this rank; returns the result dict on rank 0 (None elsewhere).
    pmc: None (normal run) or a dict {"begin": tag, "end": tag}: the short run a rocprofv3 counter pass wraps - marker kernels
    bracket a few steady-state frames, nothing is timed."""
    w = Workload(name, args, D, strong)
    if pmc is None:
        w.comm = D.make_comm(w)
    cfg, ctx, n = w.cfg, w.ctx, D.world
    steps, windows = args.steps, max(1, args.windows)
    warmup = warmup_frames(name, args.warmup)
    if name == "c2_dieoff":
        steps = DIEOFF_LAST - DIEOFF_FIRST + 1           # a window is the die-off itself, not K steps
    elif name == "c2_reburst":
        steps = max(1, (steps + REBURST_CYCLE - 1) // REBURST_CYCLE) * REBURST_CYCLE   # whole cycles; the windows start on a burst frame
        warmup = 2 * REBURST_CYCLE - 1
        windows = min(windows, 12)
    elif name in ("c2", "c2_lean", "c2_interop", "c2_view"):
        w.dt = frame_dt(1 + warmup + steps * windows)    # nobody may die in the timed frames
    elif name in BURST_SAFE_SECONDS:
        w.dt = frame_dt(1 + warmup + steps * windows, BURST_SAFE_SECONDS[name])

    if pmc is not None:   # ---- counter pass: no timing, a handful of steady frames between two markers
        frames = 6
        if name == "c2_dieoff":
            for _ in range(DIEOFF_FIRST):
                w.step()
            frames = steps
        elif name == "c2_reburst":
            for _ in range(2 * REBURST_CYCLE):
                w.step()
            frames = 2 * REBURST_CYCLE
        else:
            w.step()
            for _ in range(warmup if name in ("c5", "c2_mixed", "c2_events") else 3):
                w.step()
        ctx.synchronize()
        ctx.profile_marker(pmc["begin"])
        for _ in range(frames):
            w.step()
        ctx.profile_marker(pmc["end"])
        ctx.synchronize()
        alive = w.alive()
        w.close()
        return {"frames": frames, "alive_after": alive}

    # ---- warm-up: frame 0 is the burst (k_init + k_update) for the burst configurations, then untimed frames
    ctx.enable_kernel_timing(1)
    w.step()
    init_ms = ctx.kernel_timing()["init_ms_avg"]
    ctx.enable_kernel_timing(0)
    dieoff_counts = None
    if name == "c2_dieoff":
        # pass 0 (untimed): the alive count in front of every frame of the window, read back frame by frame. The simulation is
        # deterministic and every pass replays the same burst (same seeds, every slot spawned), so the timed passes see the same counts.
        dieoff_counts = []
        while w.f < DIEOFF_END:
            if DIEOFF_FIRST <= w.f <= DIEOFF_LAST:
                dieoff_counts.append(w.alive())
            w.step()
        assert w.alive() == 0, "c2_dieoff: particles left after the die-off"
    elif name == "c2_reburst":
        cycle_updates = []                               # max_update of every frame of a cycle (read back in the untimed warm-up; every cycle replays it)
        for _ in range(warmup):
            w.step()
            if len(cycle_updates) < REBURST_CYCLE and w.f > REBURST_CYCLE:
                cycle_updates.append(w.fxs[0].metadata()["max_update"])
        assert w.f % REBURST_CYCLE == 0 and w.alive() == 0 and len(cycle_updates) == REBURST_CYCLE, (w.f, cycle_updates)
        cycle_updates = cycle_updates[-1:] + cycle_updates[:-1]   # (recorded from frame 1 of a cycle on: rotate the burst frame to the front)
    else:
        for _ in range(warmup):
            w.step()
    D.barrier(ctx)
    alive0 = w.alive()
    alive0_total = D.alive_total(w)
    m0 = [fx.metadata() for fx in w.fxs[:8]]
    window_s, window_updates = [], []
    ctx.enable_kernel_timing(getattr(args, "timing_period", TIMING_PERIOD) if name != "c2_dieoff" else 0)
    for _win in range(windows):
        if name == "c2_dieoff":
            while w.f % DIEOFF_END != DIEOFF_FIRST:      # untimed: burst + flight up to the first death
                w.step()
            ctx.enable_kernel_timing(1)                  # (kernel events only inside the window, every frame of it)
        D.barrier(ctx)
        t0 = time.perf_counter()
        for _ in range(steps):
            w.step()
        D.barrier(ctx)
        window_s.append(D.max_time(time.perf_counter() - t0))
        if name == "c2_dieoff":
            timing_d = ctx.kernel_timing()
            ctx.enable_kernel_timing(0)
            window_updates.append(float(sum(dieoff_counts)))
            while w.f % DIEOFF_END != 0:                 # let the rest die: the next pass bursts into an empty effect
                w.step()
            w.dieoff_timing = getattr(w, "dieoff_timing", []) + [timing_d]
    timing = ctx.kernel_timing() if name != "c2_dieoff" else None
    consumer_ms = None
    if w.consumer is not None:   # the stand-in renderer alone (kernel-bound: 0.13 ms a launch), so that the row can say what of its step is simulation
        ctx.enable_kernel_timing(0)
        v = w.fxs[0].device_view()
        for _ in range(3):
            w.consumer.consumer_render_like(w._byref(v), w.consumer_out.data_ptr())
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            w.consumer.consumer_render_like(w._byref(v), w.consumer_out.data_ptr())
        ctx.synchronize()
        consumer_ms = (time.perf_counter() - t0) / 50 * 1e3
    per_program = None
    if name == "c2_events":   # the context's averages mix three programs: report each, the roofline kernel is the trails' update
        per_program = {k: p.kernel_timing() for k, p in zip(("rocket", "sparkle_trail", "trails"), w.progs)}
        timing = dict(per_program["trails"])
        timing["init_ms_avg"] = sum(v["init_ms_avg"] for v in per_program.values())
        timing["compact_ms_avg"] = sum(v["compact_ms_avg"] for v in per_program.values())
    ctx.enable_kernel_timing(0)
    if name == "c2_dieoff":
        ts = w.dieoff_timing
        tot = sum(t["frames"] for t in ts) or 1
        timing = {k: sum(t[k] * t["frames"] for t in ts) / tot for k in ("update_ms_avg", "compact_ms_avg", "init_ms_avg")}
        timing["frames"] = tot
    alive1 = w.alive()
    alive1_total = D.alive_total(w)
    m1 = [fx.metadata() for fx in w.fxs[:8]]
    stale_mask = w.fxs[0].device_view().stale_attr_mask   # after the timed frames: which planes a device-side reader would find stale (0 under the default options)
    if w.options == "default":
        assert stale_mask == 0, f"{name}: the library's default options left attribute planes stale after the timed frames (mask {stale_mask:#x})"
    comm_info = None
    if D.on and w.comm is None:
        comm_info = {"error": getattr(D, "comm_error", "no communicator"), "ranks": D.world, "alive_total": alive1_total, "via": "torch.distributed (fallback: hnb_comm_create_rank failed)"}
    if D.on and w.comm is not None:   # N > 1: what the totals above were reduced through (hnb_comm_describe)
        kind, _, rest = w.comm.describe().partition(" ")
        comm_info = {"library": rest.split(" ranks=")[0] if kind == "rccl" else kind, "ranks": int(rest.split(" ranks=")[1].split()[0]) if " ranks=" in rest else D.world,
                     "effects": w.n_effects_max, "alive_total": alive1_total, "via": "hnb_comm_create_rank + hnb_comm_allreduce_alive"}
    if getattr(args, "comm", False) and not D.on and not strong and name == args.config:
        comm_info = comm_alive_total(w)
        if "alive_total" in comm_info:
            assert comm_info["alive_total"] == alive1, f"hnb_comm_allreduce_alive says {comm_info['alive_total']}, the effects' counters {alive1}"
    w_kernel_info = w.prog.kernel_info()
    kinfo = w_kernel_info.split("\n")[0]
    # how the frames' parameter blocks reached the device (HNB_OPT_DIRECT_UPLOAD) and how often hnb_simulate found the device behind (it is device-bound then)
    submission = {}
    for ln in w_kernel_info.split("\n"):
        if ln.startswith("frame parameters (context)"):
            wd = ln.split()
            submission["frames_written_by_host"] = int(wd[wd.index("in") + 1]); submission["frames_copied"] = int(wd[wd.index("copied") + 2])
        if ln.startswith("hnb_simulate waited"):
            wd = ln.split()
            submission["frames_waited_for_device"] = int(wd[wd.index("in") + 1]); submission["frames"] = int(wd[wd.index("of") + 1]); submission["waited_us"] = int(wd[wd.index("frames,") + 1])
    parity = None
    if args.parity and D.rank == 0 and not strong:
        try:
            if name in BURST_PARITY:
                parity = parity_burst_slab(w, D)
            else:
                timed = parity_timed_state(w, args, D)
                w.close()
                parity = parity_regime(name, args, D, w.f)
                parity["timed_state"] = timed
                parity["ok"] = parity["ok"] and timed["ok"]
                parity["problems"] = parity["problems"] + timed["problems"]
        except Exception as e:
            parity = {"config": name, "ok": False, "problems": [f"{type(e).__name__}: {e}"]}
    w.close()

    if name in ("c2", "c2_lean", "c2_interop", "c2_view", "c4"):
        expect = w.local_particles if not D.on else None
        assert expect is None or (alive0 == expect and alive1 == expect), f"{name}: expected every particle alive during the timed frames, got {alive0}, {alive1}"
    if D.rank != 0:
        return None
    # particles processed by the update stage per frame (max_update). Burst configurations: constant when nothing dies. Churn (c5,
    # c2_mixed): the update processes alive-before + this frame's spawns = the last frame's max_update in the steady state.
    scale = alive1_total / max(alive1, 1) if D.on else 1.0
    if name in ("c5", "c2_mixed", "c2_events"):
        per_frame_local = float(sum(m["max_update"] for m in m1)) * (len(w.fxs) / max(1, len(m1)))
        per_frame_total = per_frame_local * scale
    elif name == "c2_dieoff":
        per_frame_local = sum(dieoff_counts) / steps
        per_frame_total = per_frame_local * n
    elif name == "c2_reburst":
        per_frame_local = sum(cycle_updates) / REBURST_CYCLE
        per_frame_total = per_frame_local * n
    else:
        per_frame_total = (alive0_total + alive1_total) / 2.0
        per_frame_local = (alive0 + alive1) / 2.0
    med = statistics.median(window_s)
    updates = per_frame_total * steps
    value = updates / med
    bpu = cfg["bytes_per_update"]
    k_ms = timing["update_ms_avg"]
    spawned = sum(b["particle_counter"] - a["particle_counter"] for a, b in zip(m0, m1)) / max(1, steps * windows)
    out = {
        "metric": "particle-updates/sec", "value": value, "unit": "particle-updates/s", "n_gpus": n,
        "steps": steps, "warmup": warmup, "ms_per_step": med / steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": cfg["workload"].format(cap=w.per_inst_cap, inst=len(w.fxs)), "name": name, "capacity_per_gpu": w.local_particles,
                   "instances_per_gpu": len(w.fxs), "dt": w.dt, "options": w.options, "stale_attr_mask_after": stale_mask, "sharding": w.sharding_desc, "alive_before": alive0_total, "alive_after": alive1_total,
                   "updates_per_frame": per_frame_total, "spawns_per_frame": spawned, "submission": submission},
        "windows": {"n": windows, "steps_each": steps, "ms_per_step": [s / steps * 1e3 for s in window_s], "median_ms_per_step": med / steps * 1e3,
                    "min_ms_per_step": min(window_s) / steps * 1e3, "value_best_window": updates / min(window_s),
                    "timed_region_s": sum(window_s)},
        "stages": {"init_ms_avg": timing["init_ms_avg"], "update_ms_avg": k_ms, "lists_ms_avg": timing["compact_ms_avg"],
                   "sum_ms": timing["init_ms_avg"] + k_ms + timing["compact_ms_avg"], "samples": timing["frames"],
                   "timing": f"HIP events on the simulation stream, every {getattr(args, 'timing_period', TIMING_PERIOD) if name != 'c2_dieoff' else 1}th timed frame (rank 0); lists = count + compact (+ ribbon sort, + event ordering)"},
        "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel": cfg["kernel"], "kernel_ms_avg": k_ms,
                     "step_kernels_ms": {"init": timing["init_ms_avg"] if name in ("c5", "c2_mixed", "c2_events", "c2_reburst") else 0.0, "update": k_ms, "lists": timing["compact_ms_avg"]},
                     "updates_per_launch": per_frame_local,
                     "algorithmic": {"bytes_per_update": bpu, "gbs_kernel": per_frame_local * bpu / (k_ms * 1e-3) / 1e9 if k_ms > 0 else None,
                                     "gbs_whole_step": updates / n * bpu / med / 1e9,
                                     "whole_step_over_peak": updates * bpu / med / 1e9 / (HBM_PEAK_GBS * n),
                                     "note": "SURVEY.md §8(d) bytes x updates / time: what the work is worth, not what was moved (above the moved figure wherever the design elides traffic)"}},
        "kernels": kinfo,
        "parity": parity,
    }
    if name in ("c2", "c2_lean", "c2_view", "c2_interop"):
        # SURVEY.md 8(d)'s 68 B per update against what this configuration's update kernel is built to move: every elided stream named with its bytes
        # (all of them proved per chunk and frame and verified by the parity gate; the PMC figure `moved_bytes_per_update` is the measured counterpart)
        el = {"lifetime_read": 4, "alive_list_read_write": 8}                    # lifetime culling (chunk bound); no casualty possible: lists untouched
        if name != "c2_interop":
            el["age_read"] = 4                                                   # age cohorts: the chunk's common age is one word
            if w.options == "age_cohort=LEAN":
                el["age_write"] = 4                                              # ... and LEAN leaves the plane stale
        out["roofline"]["algorithmic"]["elided_bytes_per_update"] = el
        out["roofline"]["algorithmic"]["designed_bytes_per_update"] = bpu - sum(el.values())
    if consumer_ms is not None:
        out["consumer_ms"] = consumer_ms
        out["sim_only_ms"] = out["ms_per_step"] - consumer_ms
        out["consumer_note"] = "consumer_ms: the stand-in renderer's kernel alone (back to back, no frames); sim_only_ms = ms_per_step - consumer_ms; value is the end-to-end rate"
    if name == "c2_reburst":
        bps = cfg["bytes_per_spawn"]
        rb = timing["init_ms_avg"]
        out["reburst"] = {"cycle": f"{REBURST_CYCLE} frames: burst of `capacity` (dt 1/60 s), then {REBURST_CYCLE - 1} x {REBURST_DT} s", "updates_per_frame_of_cycle": cycle_updates,
                          "init_kernel": "k_init_slots (slot-major; hnb_kernels.hip.h)", "init_kernel_ms": rb, "spawned": w.local_particles, "bytes_per_spawn": bps,
                          "init_frac": w.local_particles * bps / (rb * 1e-3) / 1e9 / HBM_PEAK_GBS if rb > 0 else None,
                          "row_major_before": "2.06 ms for the same re-burst in round 5 (profiles/r05_nursery/r05b_reburst.log)"}
    if comm_info is not None:
        out["comm"] = comm_info
    if per_program is not None:
        out["stages"]["per_program"] = per_program
        out["stages"]["note"] = "init / lists: sums over the three programs (lists of the rocket include its spawn-event ordering: k_emit_count + k_emit_events); update: the trails' kernel; sum_ms leaves out the two small update kernels (per_program has them)"
    if init_ms > 0 and name not in ("c5", "c2_mixed", "c2_events"):  # the burst frame's init kernel (not part of the metric; c2_reburst: the FRESH burst, its re-bursts are under "reburst")
        bps = cfg["bytes_per_spawn"]
        out["init"] = {"kernel": "k_init_slots" if "slot-major init" in w_kernel_info else "k_init", "kernel_ms": init_ms, "spawned": w.local_particles, "bytes_per_spawn": bps,
                       "achieved_gbs": w.local_particles * bps / (init_ms * 1e-3) / 1e9, "frac": w.local_particles * bps / (init_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    return out


# ------------------------------------------------------------------------------------------------------------------
# HBM traffic: counter passes of this script under rocprofv3
# ------------------------------------------------------------------------------------------------------------------
def pmc_child(args, D):
    """`bench.py --pmc-child`: the configurations, a few steady frames each between two marker kernels. Run under rocprofv3 --pmc."""
    names = args.pmc_configs.split(",")
    info = {}
    for i, name in enumerate(names):
        sub = argparse.Namespace(**vars(args))
        sub.capacity = args.capacity if name == args.config else None
        info[name] = run_config(name, sub, D, pmc={"begin": 100 + 2 * i, "end": 101 + 2 * i})
        info[name]["begin"], info[name]["end"] = 100 + 2 * i, 101 + 2 * i
    print("PMCINFO " + json.dumps(info), flush=True)


def parse_counter_csv(path, counter, info):
    """-> {config: {"dominant_kib": median per launch, "frame_kib": all kernels per frame, "per_kernel_kib": {...}}}"""
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") == counter:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["Grid_Size"]), float(r["Counter_Value"])))
    rows.sort()
    out = {}
    # (several launches of one kernel template per frame - c2_events: the sparkle trail and the trails run the same update kernel -: the
    # dominant launch is the one with the largest grid)
    for name, meta in info.items():
        inside, sect = False, []
        for _id, kname, grid, val in rows:
            if "k_marker" in kname:
                if grid == meta["begin"]:
                    inside = True
                elif grid == meta["end"]:
                    inside = False
                continue
            if inside:
                sect.append((kname, val, grid))
        if not sect:
            continue
        per = {}
        for kname, val, _g in sect:
            short = kname.split("(")[0].replace("void ", "").replace("hnb::", "")
            per.setdefault(short, []).append(val)
        match = [(v, g) for k, v, g in sect if any(m in k for m in KERNEL_MATCH[name])]
        gmax = max((g for _v, g in match), default=0)
        dom = [v for v, g in match if g == gmax]
        out[name] = {"dominant_kib": statistics.median(dom) if dom else None, "frame_kib": sum(v for _k, v, _g in sect) / meta["frames"],
                     "per_kernel_kib": {k[:120]: {"median": statistics.median(vs), "launches_per_frame": len(vs) / meta["frames"]} for k, vs in per.items()}}
    return out


def parse_kernel_trace(path, info):
    """rocprofv3 --kernel-trace CSV of the marker-cut child -> {config: {"dominant_ms": avg duration of the dominant kernel's launches,
    "dominant_n": launches, "per_kernel_ms": {kernel: {"avg", "min", "max", "launches_per_frame"}}}}: the per-configuration kernel durations
    a reader can re-derive `frac` from without trusting this process's HIP events."""
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["Grid_Size_X"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6))
    rows.sort()
    out = {}
    for name, meta in info.items():
        inside, sect = False, []
        for _id, kname, grid, ms in rows:
            if "k_marker" in kname:
                inside = grid == meta["begin"] if grid in (meta["begin"], meta["end"]) else inside
                continue
            if inside:
                sect.append((kname, ms, grid))
        if not sect:
            continue
        per = {}
        for kname, ms, _g in sect:
            per.setdefault(kname.split("(")[0].replace("void ", "").replace("hnb::", "")[:120], []).append(ms)
        match = [(ms, g) for k, ms, g in sect if any(m in k for m in KERNEL_MATCH[name])]
        gmax = max((g for _ms, g in match), default=0)
        dom = [ms for ms, g in match if g == gmax]
        out[name] = {"dominant_ms": sum(dom) / len(dom) if dom else None, "dominant_n": len(dom), "frames": meta["frames"],
                     "per_kernel_ms": {k: {"avg": sum(v) / len(v), "min": min(v), "max": max(v), "launches_per_frame": len(v) / meta["frames"]} for k, v in per.items()}}
    return out


def measure_traffic(args, names):
    """Two rocprofv3 counter passes of this script (FETCH_SIZE and WRITE_SIZE cannot share a pass). Returns
    {config: {bytes_per_launch, frame_bytes, ...}} or raises."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    res = {}
    tmp = tempfile.mkdtemp(prefix="hnb_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE", "DURATION"):
            outdir = os.path.join(tmp, counter)
            # (DURATION: a third pass with the kernel trace alone - counter collection perturbs the durations it would report)
            cmd = [exe, "--kernel-trace"] + (["--pmc", counter] if counter != "DURATION" else []) + ["--output-format", "csv", "-d", outdir, "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", "--pmc-configs", ",".join(names), "--config", args.config]
            if args.capacity:
                cmd += ["--capacity", str(args.capacity)]
            if args.instances:
                cmd += ["--instances", str(args.instances)]
            env = dict(os.environ)
            env["TMPDIR"] = "/tmp"
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=args.pmc_timeout, cwd="/tmp", env=env)
            line = next((ln for ln in p.stdout.splitlines() if ln.startswith("PMCINFO ")), None)
            if p.returncode != 0 or line is None:
                raise RuntimeError(f"counter pass {counter} failed (rc {p.returncode}): {(p.stderr or p.stdout)[-300:]}")
            info = json.loads(line[len("PMCINFO "):])
            if counter == "DURATION":
                files = glob.glob(os.path.join(outdir, "**", "*kernel_trace.csv"), recursive=True)
                if not files:
                    raise RuntimeError("duration pass: no kernel_trace.csv")
                res[counter] = parse_kernel_trace(files[0], info)
                if args.keep_pmc:
                    os.makedirs(args.keep_pmc, exist_ok=True)
                    with open(os.path.join(args.keep_pmc, "kernel_durations.json"), "w") as fo:
                        json.dump(res[counter], fo, indent=1)
                continue
            files = glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                raise RuntimeError(f"counter pass {counter}: no counter_collection.csv")
            res[counter] = parse_counter_csv(files[0], counter, info)
            if args.keep_pmc:
                os.makedirs(args.keep_pmc, exist_ok=True)
                keep = os.path.join(args.keep_pmc, f"pmc_{counter}.csv")
                with open(files[0]) as fi, open(keep, "w") as fo:   # only the library's kernels, names shortened
                    rd = csv.DictReader(fi)
                    cols = ["Dispatch_Id", "Grid_Size", "Workgroup_Size", "VGPR_Count", "Kernel_Name", "Counter_Name", "Counter_Value"]
                    wr = csv.DictWriter(fo, fieldnames=cols)
                    wr.writeheader()
                    for r in rd:
                        if r.get("Counter_Name") == counter and ("hnb::" in r["Kernel_Name"] or "k_marker" in r["Kernel_Name"]):
                            r = {c: r.get(c, "") for c in cols}
                            r["Kernel_Name"] = r["Kernel_Name"].split("(")[0][:140]
                            wr.writerow(r)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    for name in names:
        f, wv = res["FETCH_SIZE"].get(name), res["WRITE_SIZE"].get(name)
        if not f or not wv or f["dominant_kib"] is None or wv["dominant_kib"] is None:
            continue
        out[name] = {"bytes_per_launch": f["dominant_kib"] * 1024 * 2 + wv["dominant_kib"] * 1024,
                     "fetch_size_kib": f["dominant_kib"], "write_size_kib": wv["dominant_kib"],
                     "frame_bytes": f["frame_kib"] * 1024 * 2 + wv["frame_kib"] * 1024,
                     "per_kernel": {k: {"fetch_kib": v["median"], "write_kib": wv["per_kernel_kib"].get(k, {}).get("median"), "launches_per_frame": v["launches_per_frame"]}
                                    for k, v in f["per_kernel_kib"].items()},
                     "rocprof": res.get("DURATION", {}).get(name)}
    return out


def attach_roofline(result, name, traffic, source):
    """Completes result['roofline'] from the measured (or recorded, or modelled) traffic of the dominant kernel."""
    r = result["roofline"]
    k_ms, n_upd = r["kernel_ms_avg"], r["updates_per_launch"]
    if traffic is not None:
        b = traffic["bytes_per_launch"]
        r["traffic"] = b
        r["traffic_detail"] = {"fetch_size_kib": traffic["fetch_size_kib"], "write_size_kib": traffic["write_size_kib"],
                               "formula": "FETCH_SIZE KiB x 1024 x 2 (gfx950) + WRITE_SIZE KiB x 1024, median launch of the dominant kernel; counters sit between L2 and the fabric (Infinity-Cache hits included)",
                               "whole_frame_bytes": traffic.get("frame_bytes"), "per_kernel": traffic.get("per_kernel")}
    else:
        b = CONFIGS[name]["model_bytes"] * n_upd
        r["traffic"] = None
    if traffic is not None and traffic.get("rocprof") and traffic["rocprof"].get("dominant_ms"):
        rp = traffic["rocprof"]
        r["rocprof_detail"] = {"launches": rp["dominant_n"], "per_kernel_ms": rp["per_kernel_ms"]}
        # rocprofv3 --kernel-trace, same marker-cut frames as the counters - in a CHILD process, which allocates its slabs anew: where a block lands in
        # physical memory changes the same kernel by 3-10 % (alloc_slab_block), and the child's placement is not the timed process's. A kernel
        # duration above the whole timed step describes a different placement, not this run: it is kept in the detail and not put beside the step.
        if rp["dominant_ms"] <= result["ms_per_step"]:
            r["kernel_ms_rocprof"] = rp["dominant_ms"]
            r["frac_rocprof"] = traffic["bytes_per_launch"] / (rp["dominant_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        else:
            r["rocprof_detail"]["other_placement"] = {"kernel_ms": rp["dominant_ms"], "note": "the counter child's slab placement was slower than the timed process's whole step: not comparable"}
    r["traffic_unit"], r["traffic_source"] = "B/launch", source
    r["moved_bytes_per_update"] = b / n_upd if n_upd else None
    r["achieved"] = b / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    r["frac"] = r["achieved"] / HBM_PEAK_GBS
    r["algorithmic"]["ratio_moved_to_algorithmic"] = (b / n_upd) / CONFIGS[name]["bytes_per_update"] if n_upd else None
    if traffic is not None and traffic.get("frame_bytes"):
        r["whole_step"] = {"moved_bytes": traffic["frame_bytes"], "gbs": traffic["frame_bytes"] / (result["ms_per_step"] * 1e-3) / 1e9,
                           "frac": traffic["frame_bytes"] / (result["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "note": "all kernels of a frame (PMC) / the driver-timed step"}


def load_recorded_traffic(stamp):
    """profiles/traffic.json, only if it was recorded from this kernel build."""
    try:
        with open(TRAFFIC_FILE) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None, "no profiles/traffic.json"
    if t.get("kernel_source_stamp") != stamp:
        return None, f"profiles/traffic.json is stale (recorded for kernel sources {t.get('kernel_source_stamp')}, this tree is {stamp})"
    return t.get("configs", {}), f"profiles/traffic.json (recorded at {t.get('head')}, kernel sources {stamp})"


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


def threads_launch(args):
    """`--launcher threads`: the single-process host (examples/multi_gpu.c): N contexts, N submit threads, capacity-slab sharding of one giant
    effect, alive counters all-reduced through hnb_comm_* (RCCL). Python only lowers the asset and relays the C program's numbers."""
    import bevy_hanabi_amd as bh
    from bevy_hanabi_amd import build as hb
    from bevy_hanabi_amd import effects

    if args.config not in ("c2", "c3"):
        print("--launcher threads shards ONE burst effect by capacity slab: --config c2 or c3", file=sys.stderr)
        return 2
    cfg = CONFIGS[args.config]
    cap = args.capacity or cfg["capacity"]
    n = args.gpus
    exe = hb.build_examples()
    asset = effects.firework_trails(cap) if args.config == "c2" else effects.force_field(cap)
    windows = max(1, args.windows)
    dt = frame_dt(1 + args.warmup + args.steps * windows) if args.config == "c2" else DT
    devices = ",".join(str(args.force_device if args.force_device is not None else g) for g in range(n))
    with tempfile.NamedTemporaryFile(suffix=".blob", delete=False) as f:
        f.write(bh.lower(asset))
        blob = f.name
    try:
        p = subprocess.run([exe, blob, devices, str(args.warmup), str(args.steps), str(windows), repr(dt)], capture_output=True, text=True, timeout=900)
    finally:
        os.unlink(blob)
    if p.returncode != 0:
        print(p.stderr, file=sys.stderr)
        return p.returncode
    r = json.loads(p.stdout.strip().splitlines()[-1])
    bpu = cfg["bytes_per_update"]
    out = {"metric": "particle-updates/sec", "value": r["updates_per_s"], "unit": "particle-updates/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": cfg["workload"].format(cap=cap, inst=1), "name": args.config, "capacity_per_gpu": cap, "dt": dt, "sharding": f"capacity slab x{n}",
                      "alive_after": r["alive_total"], "devices": r["devices"]},
           "launcher": "threads: one process, one HnbContext + one submit thread per GPU (examples/multi_gpu.c, C99 over include/hanabi_amd.h); alive counters all-reduced by hnb_comm_allreduce_alive (RCCL)",
           "windows": {"n": windows, "steps_each": args.steps, "ms_per_step": r["window_ms_per_step"], "median_ms_per_step": r["ms_per_step"], "min_ms_per_step": r["min_ms_per_step"]},
           "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS * n, "unit": "GB/s", "traffic": None, "traffic_source": "not collected by this launcher (see the N = 1 line)",
                        "algorithmic": {"bytes_per_update": bpu, "gbs_whole_step": r["updates_per_s"] * bpu / 1e9, "whole_step_over_peak": r["updates_per_s"] * bpu / 1e9 / (HBM_PEAK_GBS * n)}},
           "build": {"head": git_head(), "kernel_source_stamp": kernel_source_stamp()}}
    print(json.dumps(out), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--windows", type=int, default=25, help="timed windows of --steps frames each; the line reports the median window")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--scaling", choices=["weak", "strong", "both"], default="both")
    ap.add_argument("--capacity", type=int, default=None, help="particles per effect instance (default: the configuration's)")
    ap.add_argument("--instances", type=int, default=None, help="c4: instances per GPU (weak) / in total (strong); default 512")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", dest="parity", action="store_false", default=True,
                    help="skip the oracle comparison of the state the timed frames produced (A/B tooling only: the line then says parity.ok = null)")
    ap.add_argument("--full-json", default=os.path.join(ROOT, "profiles", "bench_full.json"), help="where the complete record goes (the last stdout line is the short one)")
    ap.add_argument("--scene", dest="scene", action="store_true", default=True,
                    help="N = 1: append the small-effects scene (26 example effects in one context, tools/scene_bench.py) as \"small_effects_scene\" (default)")
    ap.add_argument("--no-scene", dest="scene", action="store_false")
    ap.add_argument("--no-extra-configs", action="store_true", help="N = 1, c2: do not append the other configurations under \"configs\"")
    ap.add_argument("--pmc", choices=["auto", "off"], default="auto", help="auto: measure HBM traffic in this run (two rocprofv3 counter passes of this script)")
    ap.add_argument("--pmc-timeout", type=int, default=420)
    ap.add_argument("--keep-pmc", default=None, help="directory: keep the filtered counter CSVs of the two passes (profiles/)")
    ap.add_argument("--write-traffic", action="store_true", help="record the measured traffic in profiles/traffic.json (stamped with HEAD and the kernel sources)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-configs", default="c2", help=argparse.SUPPRESS)
    ap.add_argument("--launcher", choices=["ranks", "threads"], default="ranks",
                    help="N > 1: ranks = one process per GPU (torch.distributed.run; alive totals through hnb_comm_create_rank); threads = ONE process, one HnbContext and one "
                         "submit thread per GPU (examples/multi_gpu.c: C99 over the C ABI, RCCL through hnb_comm_*), c2 / c3 only")
    ap.add_argument("--no-comm", dest="comm", action="store_false", default=True,
                    help="N = 1: do not take the headline's alive total through hnb_comm_allreduce_alive (a one-rank communicator of the real librccl)")
    ap.add_argument("--timing-period", type=int, default=TIMING_PERIOD, help="HIP events bracket the kernels of every n-th timed frame (stage / kernel times; each such frame costs ~20 us of stream bubbles)")
    ap.add_argument("--comm-lib", default=None, help="N > 1 dry runs: the collective library hnb_comm_* loads instead of librccl (tests/fake_rccl/libfake_rccl.so takes several ranks on one GPU)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for dry runs)")
    ap.add_argument("--force-device", type=int, default=None, help="dry runs: every rank uses this GPU instead of LOCAL_RANK")
    args = ap.parse_args()

    if args.launcher == "threads" and "WORLD_SIZE" not in os.environ and not args.pmc_child:
        sys.exit(threads_launch(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    D = Dist(args)
    if args.pmc_child:
        pmc_child(args, D)
        return
    if D.on:   # (before the first hnb_comm_* call of the process)
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
        if args.comm_lib:
            from bevy_hanabi_amd import runtime as _rt
            _rt.comm_set_library(os.path.abspath(args.comm_lib), duplicate_devices=True)
    if args.comm and not D.on:   # (before the first hnb_comm_* call of the process)
        try:
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
            os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")   # (RCCL's own messages: not into the stream whose last line is the result)
            from bevy_hanabi_amd import runtime as _rt
            _rt.comm_set_library(None, single_rank=True)
        except Exception as e:
            print(f"note: hnb_comm_set_library: {e}", file=sys.stderr)
            args.comm = False
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
                                 "algorithmic_whole_step_over_aggregate_peak": s["roofline"]["algorithmic"]["whole_step_over_peak"],
                                 "workload": "the N = 1 workload split over the ranks"}
    results = {args.config: out}
    if not D.on and args.config == "c2" and not args.no_extra_configs:
        # the other single-GPU configurations, same process, same window protocol
        for name in EXTRA_CONFIGS:
            sub = argparse.Namespace(**vars(args))
            sub.capacity = None
            try:
                results[name] = run_config(name, sub, D)
            except Exception as e:  # the headline line must not be lost to a side configuration
                results[name] = {"error": f"{type(e).__name__}: {e}"}
    if D.rank == 0:
        # ---- HBM traffic of the dominant kernels: measured now, else recorded for this kernel build, else modelled
        stamp = kernel_source_stamp()
        names = [k for k, v in results.items() if v and "error" not in v]
        measured, why = None, ("--pmc off" if args.pmc == "off" else "N > 1: counter passes are a single-GPU measurement")
        if not D.on and args.pmc == "auto":
            try:
                t0 = time.perf_counter()
                measured = measure_traffic(args, names)
                why = f"measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this script ({time.perf_counter() - t0:.0f} s), kernel sources {stamp}"
            except Exception as e:
                why = f"{type(e).__name__}: {e}"
        recorded, rec_src = (None, None) if measured else load_recorded_traffic(stamp)
        for name in names:
            if measured and name in measured:
                attach_roofline(results[name], name, measured[name], why)
            elif recorded and name in recorded:
                attach_roofline(results[name], name, recorded[name], rec_src)
            else:
                attach_roofline(results[name], name, None, f"model: {CONFIGS[name]['model_bytes']} B per update by design (no counter pass: {why}; {rec_src or ''})")
        if measured and args.write_traffic:
            with open(TRAFFIC_FILE, "w") as f:
                json.dump({"kernel_source_stamp": stamp, "head": git_head(), "configs": measured}, f, indent=1)
        out = results[args.config]
        out["build"] = {"head": git_head(), "kernel_source_stamp": stamp}
        extra = {k: v for k, v in results.items() if k != args.config}
        if extra:
            out["configs"] = {k: (v if "error" in v else {kk: v[kk] for kk in ("value", "ms_per_step", "windows", "stages", "roofline", "init", "kernels", "parity") if kk in v}
                                  | {"workload": v["config"]["workload"], "updates_per_frame": v["config"]["updates_per_frame"], "spawns_per_frame": v["config"]["spawns_per_frame"]})
                              | {kk: v[kk] for kk in ("consumer_ms", "sim_only_ms", "reburst") if kk in v}
                              for k, v in extra.items()}
    if not D.on and args.scene and args.config == "c2" and not args.no_extra_configs and not args.pmc_child:
        # the launch-bound end of the path: 26 different small effects in one context (tools/scene_bench.py, profiles/r03v_scene.log). Their
        # passes run in the job-table kernels (k_init_jobs, k_update_jobs, k_count_rows_multi, k_compact_multi): the statistics of the
        # headline's kernels in a rocprofv3 run of this command stay the headline's own
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import scene_bench
            out["small_effects_scene"] = scene_bench.run(1, 300, device=D.device_index, quiet=True)
            # ... and with the merged launches on the byte-code interpreters (HNB_OPT_SET_MODULE off: what a set without a compiled module gets)
            interp = scene_bench.run(1, 300, device=D.device_index, quiet=True, set_module=0)
            out["small_effects_scene"]["ms_per_frame_wall_interpreters"] = interp["ms_per_frame_wall"]
        except Exception as e:
            out["small_effects_scene"] = {"error": f"{type(e).__name__}: {e}"}
    if D.rank == 0 and not args.no_cpu_baseline and not D.on and args.config == "c2":
        # rank 0 at N = 1 only: the host cores are shared by the ranks otherwise
        try:
            out["cpu_baseline"] = cpu_baseline(args.capacity or CONFIGS["c2"]["capacity"])
        except Exception as e:
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    rc = 0
    if D.rank == 0:
        short = short_line(out, args)
        text = json.dumps(out)
        for path in [args.full_json] + ([os.path.join(ROOT, "gpurun_out", "bench_full.json")] if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else []):
            try:
                os.makedirs(os.path.dirname(path), exist_ok=True)
                with open(path, "w") as f:
                    f.write(text + "\n")
            except OSError as e:
                print(f"note: could not write {path}: {e}", file=sys.stderr)
        print(text, file=sys.stderr, flush=True)          # the complete record, for a log; stdout carries the short line LAST
        # The collective library (librccl, loaded for `comm`) prints a banner and warnings through C stdio into this process's stdout: buffered there, they
        # would land BEHIND this line when the process exits (seen: "Librccl path : ..." as the last line of a run). Flush C stdio first, start on a fresh line.
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print("\n" + encode_line(short), flush=True)
        rc = 0 if short["parity"]["ok"] is not False else 1
    D.close()
    if _COMM_STUCK or (args.comm and not D.on):   # (a thread may still be inside the collective library; and nothing it prints at exit may follow the result line)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(rc)
    sys.exit(rc)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def _r(x, digits=4):
    """floats rounded to `digits` significant digits (the short line only; bench_full.json keeps everything)"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    return x


def encode_line(short):
    return json.dumps(short, separators=(",", ":"))


def short_line(full, args):
    """The LAST stdout line: what the driver parses (<= 4 KB; tests/test_bench_line.py builds a worst case and asserts < 6000 bytes).
    The headline configuration with roofline, cpu_baseline and parity, plus one row per other configuration; everything else is in the
    complete record (--full-json)."""
    ro, win = full.get("roofline", {}), full.get("windows", {})
    ms = win.get("ms_per_step") or [full.get("ms_per_step")]
    parity_all = [full.get("parity")] + [v.get("parity") for v in full.get("configs", {}).values() if isinstance(v, dict)]
    parity_all = [p for p in parity_all if p]
    failed = [p["config"] for p in parity_all if not p.get("ok")]
    errored = [k for k, v in full.get("configs", {}).items() if "error" in v]
    parity = {"checked": [p["config"] for p in parity_all if p.get("ok")], "ok": (not failed) if (parity_all or args.parity) else None, "burst": "full-size slab after the timed frames vs oracle/, bit-exact",
              "churn": "timed state: invariants + plain-path differential at full size; oracle at reduced capacity"}
    timed = [p.get("timed_state") for p in parity_all if p.get("timed_state")]
    if timed:   # (the churn configurations' full-size leg: how many effects were checked / compared on the device)
        parity["timed_state"] = {"checked": sum(len(t["checks"]) for t in timed), "compared": sum(len(t["diffs"]) for t in timed), "ok": all(t["ok"] for t in timed)}
    if failed:
        parity["failed"] = {p["config"]: (p.get("problems") or ["?"])[0][:56] for p in parity_all if not p.get("ok")}
    if not args.parity:
        parity["skipped"] = "--no-parity"
    short = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    if parity["ok"] is False:
        short["value"] = None            # a number whose state differs from the oracle's is not a result
        short["refused"] = "parity gate failed: " + ", ".join(failed)
    cfg = full.get("config", {})
    short["config"] = {k: (cfg.get(k)[:120] if k == "workload" else cfg.get(k)) for k in ("workload", "name", "capacity_per_gpu", "instances_per_gpu", "dt", "options", "stale_attr_mask_after", "sharding", "updates_per_frame") if k in cfg}
    short["windows"] = {"n": win.get("n"), "steps_each": win.get("steps_each"), "ms_per_step_min_median_max": [_r(min(ms)), _r(statistics.median(ms)), _r(max(ms))],
                        "timed_region_s": _r(win.get("timed_region_s"))}
    short["roofline"] = {"bound": ro.get("bound"), "kernel": ro.get("kernel"), "kernel_ms_avg": _r(ro.get("kernel_ms_avg")), "kernel_ms_rocprof": _r(ro.get("kernel_ms_rocprof")),
                         "traffic": ro.get("traffic"), "traffic_source": (ro.get("traffic_source") or "")[:20],
                         "moved_bytes_per_update": _r(ro.get("moved_bytes_per_update")), "achieved": _r(ro.get("achieved")), "peak": ro.get("peak"), "unit": ro.get("unit"),
                         "frac": _r(ro.get("frac")), "frac_rocprof": _r(ro.get("frac_rocprof")),
                         # (flat: a record that keeps only the scalar members of this object still says what the kernel elides - SURVEY.md 8(d)'s 68 B x
                         #  updates / step time / peak above 1 means the timed kernel does not move the per-particle age / lifetime / list bytes)
                         "algorithmic_bytes_per_update": ro.get("algorithmic", {}).get("bytes_per_update"),
                         "algorithmic_whole_step_over_peak": _r(ro.get("algorithmic", {}).get("whole_step_over_peak")),
                         # (which of the 68 B the kernel is built not to move, named with their bytes; the sum of the frame's kernels - one, the update, in the headline's timed frames)
                         "elided": " + ".join(f"{k} {v}" for k, v in (ro.get("algorithmic", {}).get("elided_bytes_per_update") or {}).items()) or None,
                         "designed_bytes_per_update": ro.get("algorithmic", {}).get("designed_bytes_per_update"),
                         "step_kernels_ms": _r(sum((ro.get("step_kernels_ms") or {}).values()), 4) if ro.get("step_kernels_ms") else None,
                         "whole_step_frac": _r(ro.get("whole_step", {}).get("frac"))}
    cb = full.get("cpu_baseline")
    if cb:
        short["cpu_baseline"] = cb if "error" in cb else {"value": _r(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "threads": cb["threads"], "host_physical_cores": cb.get("host_physical_cores"),
                                                          "cpu_model": cpu_model(), "cpu_quota": (cb.get("limits") or {}).get("cgroup_cpu_quota"), "pool_best": _r((cb.get("pool") or {}).get("best")), "pool_worst": _r((cb.get("pool") or {}).get("worst")), "kind": cb["kind"], "sample": cb["sample"][:24]}
    short["parity"] = parity
    if full.get("comm"):   # (N = 1: the alive total went through hnb_comm_allreduce_alive over a one-rank communicator of the real librccl)
        short["comm"] = {k: v for k, v in full["comm"].items() if k != "effects"}
    rows = {}
    for k, v in full.get("configs", {}).items():
        if "error" in v:
            rows[k] = {"error": v["error"][:120]}
            continue
        r2 = v.get("roofline", {})
        # (value, ms_per_step: median window; kernel_ms: HIP events (the rocprofv3 kernel-trace figure of every row is in the complete record); frac = moved bytes / kernel time / 8 TB/s;
        #  B_upd = moved bytes per update; ws_frac = all kernels' moved bytes / step time / peak; ws68 = 68 B x updates / step time / peak;
        #  stages_ms = [init, update, lists])
        rows[k] = {"value": _r(v.get("value")), "ms_per_step": _r(v.get("ms_per_step")), "kernel_ms": _r(r2.get("kernel_ms_avg")),
                   "frac": _r(r2.get("frac"), 3), "B_upd": _r(r2.get("moved_bytes_per_update"), 3),
                   "ws_frac": _r(r2.get("whole_step", {}).get("frac"), 3), "ws68": _r(r2.get("algorithmic", {}).get("whole_step_over_peak"), 3),
                   "stages_ms": [_r(v.get("stages", {}).get(x), 3) for x in ("init_ms_avg", "update_ms_avg", "lists_ms_avg")]}   # (parity: "parity".checked / .failed name the configuration)
        if v.get("consumer_ms") is not None:
            rows[k]["consumer_ms"], rows[k]["sim_only_ms"] = _r(v["consumer_ms"], 3), _r(v.get("sim_only_ms"), 3)
        if v.get("reburst"):
            rows[k]["reburst_init_ms"], rows[k]["reburst_init_frac"] = _r(v["reburst"]["init_kernel_ms"], 3), _r(v["reburst"]["init_frac"], 3)
        if v.get("init") and k in ("c3", "c4"):   # (the burst frame's k_init against 8 TB/s on its algorithmic bytes: c2's is "burst_init" below)
            rows[k]["init_frac"] = _r(v["init"].get("frac"), 3)
    if rows:
        short["configs"] = rows
    if full.get("init"):
        short["burst_init"] = {"kernel_ms": _r(full["init"].get("kernel_ms")), "frac": _r(full["init"].get("frac"), 3)}
    sc = full.get("small_effects_scene")
    if sc:
        short["small_effects_scene"] = sc if "error" in sc else {"effects": sc.get("effects"), "ms_per_frame_wall": _r(sc.get("ms_per_frame_wall")), "ms_in_simulate": _r(sc.get("ms_per_frame_in_simulate")),
                                                                                "ms_interpreters": _r(sc.get("ms_per_frame_wall_interpreters")), "in_set_module": sc.get("programs_served_by_the_set_module")}
    if full.get("strong"):
        short["strong"] = {k: _r(v) for k, v in full["strong"].items() if k != "workload"}
    short["build"] = full.get("build")
    short["full_record"] = os.path.relpath(args.full_json, ROOT)
    if errored:
        short["config_errors"] = errored
    return short


if __name__ == "__main__":
    main()
