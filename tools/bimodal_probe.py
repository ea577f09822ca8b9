"""Development tool: is the headline kernel time bimodal per process, per context or per allocation?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevy_hanabi_amd as bh
from bevy_hanabi_amd import effects
from bench import frame_seed, frame_dt

cap = 1 << 24
blob = bh.lower(effects.firework_trails(cap))
dt = frame_dt(80)

def measure(ctx, prog, fx, frames=40):
    ctx.frame_begin(dt, 0.0); fx.set_frame(cap, frame_seed(0)); ctx.simulate()
    for f in range(1, 6):
        ctx.frame_begin(dt, f * dt); fx.set_frame(0, frame_seed(f)); ctx.simulate()
    ctx.synchronize()
    ctx.enable_kernel_timing(1)
    for f in range(6, 6 + frames):
        ctx.frame_begin(dt, f * dt); fx.set_frame(0, frame_seed(f)); ctx.simulate()
    ctx.synchronize()
    t = ctx.kernel_timing(); ctx.enable_kernel_timing(0)
    return t["update_ms_avg"]

mode = sys.argv[1] if len(sys.argv) > 1 else "ctx"
if mode == "ctx":      # a new context (new arena) every time
    for i in range(8):
        ctx = bh.Context(0); prog = ctx.create_program(blob); fx = prog.create_effect()
        print("new ctx   %d: %.4f ms" % (i, measure(ctx, prog, fx)), flush=True)
        prog.destroy(); ctx.close()
elif mode == "prog":   # same context, new program + effect (arena blocks are reused)
    ctx = bh.Context(0)
    for i in range(8):
        prog = ctx.create_program(blob); fx = prog.create_effect()
        print("new prog  %d: %.4f ms" % (i, measure(ctx, prog, fx)), flush=True)
        prog.destroy()
    ctx.close()
elif mode == "same":   # same effect measured repeatedly
    ctx = bh.Context(0); prog = ctx.create_program(blob); fx = prog.create_effect()
    measure(ctx, prog, fx)
    for i in range(8):
        ctx.enable_kernel_timing(1)
        for f in range(100 + i * 40, 140 + i * 40):
            ctx.frame_begin(dt, 0.0); fx.set_frame(0, frame_seed(f)); ctx.simulate()
        ctx.synchronize()
        print("same fx   %d: %.4f ms" % (i, ctx.kernel_timing()["update_ms_avg"]), flush=True)
        ctx.enable_kernel_timing(0)
    ctx.close()
elif mode == "multi":  # one context, several programs alive at once (different memory), each measured alone, twice
    ctx = bh.Context(0)
    items = []
    for i in range(5):
        prog = ctx.create_program(blob); fx = prog.create_effect()
        items.append((prog, fx))
    for rep in range(2):
        for i, (prog, fx) in enumerate(items):
            # freeze the others so that only this effect is simulated
            for j, (_, other) in enumerate(items):
                other.set_simulated(j == i)
            print("rep %d effect %d: %.4f ms" % (rep, i, measure(ctx, prog, fx, frames=25)), flush=True)
    ctx.close()
