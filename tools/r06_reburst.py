"""Round 6: init kernel time of a fresh burst and of a RE-burst after a complete die-off (16.7M firework trails), slot-major init on / off;
and the c2 update under AUTO (age plane kept current in the kernel) / LEAN / OFF.   python tools/r06_reburst.py [capacity]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import bevy_hanabi_amd as bh  # noqa: E402
from bevy_hanabi_amd import effects  # noqa: E402
from bench import frame_seed  # noqa: E402

cap = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24


def init_ms(ctx, fx, f, dt, spawn):
    ctx.enable_kernel_timing(1)
    ctx.frame_begin(dt, f * dt)
    fx.set_frame(spawn, frame_seed(f))
    ctx.simulate()
    t = ctx.kernel_timing()
    ctx.enable_kernel_timing(0)
    return t["init_ms_avg"], t["update_ms_avg"], t["compact_ms_avg"]


for slot_init in (1, 0):
    ctx = bh.Context(0)
    ctx.set_option("slot_init", slot_init)
    prog = ctx.create_program(bh.lower(effects.firework_trails(cap)))
    fx = prog.create_effect()
    out = []
    f = 0
    for cycle in range(3):
        out.append(("burst" if cycle == 0 else "re-burst", init_ms(ctx, fx, f, 1 / 60, cap))); f += 1
        for _ in range(75):     # 1/60 s frames: the die-off of frames 48..72 leaves the dead stack in killing order
            ctx.frame_begin(1 / 60, f / 60); fx.set_frame(0, frame_seed(f)); ctx.simulate(); f += 1
        assert fx.alive_count() == 0
    # a partial re-fill: half of the slots after a die-off
    out.append(("partial half", init_ms(ctx, fx, f, 1 / 60, cap // 2)))
    print(f"slot_init={slot_init} capacity={cap}: " + "; ".join(f"{k}: init {a:.4f} ms (44 B/spawn: {44 * (cap if 'half' not in k else cap // 2) / a / 1e6 / 8000:.3f} of 8 TB/s), update {b:.4f}, lists {c:.4f}" for k, (a, b, c) in out), flush=True)
    print(prog.kernel_info().split("\n")[0], "|", [l for l in prog.kernel_info().split("\n") if "slot-major" in l])
    ctx.close()

for mode, name in ((3, "AUTO"), (1, "LEAN"), (0, "OFF")):
    ctx = bh.Context(0)
    ctx.set_option("age_cohort", mode)
    prog = ctx.create_program(bh.lower(effects.firework_trails(cap)))
    fx = prog.create_effect()
    dt = 1e-3
    ctx.frame_begin(dt, 0.0); fx.set_frame(cap, frame_seed(0)); ctx.simulate()
    for f in range(1, 20):
        ctx.frame_begin(dt, f * dt); fx.set_frame(0, frame_seed(f)); ctx.simulate()
    ctx.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for f in range(100):
            ctx.frame_begin(dt, f * dt); fx.set_frame(0, frame_seed(f)); ctx.simulate()
        ctx.synchronize()
        best = min(best, (time.perf_counter() - t0) / 100)
    print(f"c2 update, age_cohort={name}: {best * 1e3:.4f} ms per frame, stale mask {fx.device_view().stale_attr_mask}", flush=True)
    ctx.close()
