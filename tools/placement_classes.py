"""Development tool: update-kernel time of the 16,777,216-particle firework on N slabs alive at the same time (one context each,
placement search off), walking the chunks in one direction every frame and in alternating directions (HNB_OPT_ALTERNATE)."""
import os, sys
os.environ["HNB_SLAB_CANDIDATES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevy_hanabi_amd as bh
from bevy_hanabi_amd import effects
from bench import frame_seed, frame_dt

cap = 1 << 24
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
blob = bh.lower(effects.firework_trails(cap))
dt = frame_dt(400)
items = []
for i in range(n):
    ctx = bh.Context(0)
    prog = ctx.create_program(blob)
    fx = prog.create_effect()
    ctx.frame_begin(dt, 0.0); fx.set_frame(cap, frame_seed(0)); ctx.simulate(); ctx.synchronize()
    items.append((ctx, prog, fx, [1]))

def measure(ctx, fx, fcount, frames=30):
    ctx.enable_kernel_timing(1)
    for _ in range(frames):
        f = fcount[0]; fcount[0] += 1
        ctx.frame_begin(dt, f * dt); fx.set_frame(0, frame_seed(f)); ctx.simulate()
    ctx.synchronize()
    t = ctx.kernel_timing()["update_ms_avg"]
    ctx.enable_kernel_timing(0)
    return t

print("slab  one-direction  alternating  one-direction  alternating   (ms, 30 frames each)")
for i, (ctx, prog, fx, fc) in enumerate(items):
    row = []
    for alt in (0, 1, 0, 1):
        ctx.set_option(2, alt)
        measure(ctx, fx, fc, frames=4)
        row.append(measure(ctx, fx, fc))
    print("%3d   %.4f         %.4f       %.4f         %.4f" % (i, *row), flush=True)
