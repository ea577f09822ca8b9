"""Development tool: whole-step time with and without the per-kernel HIP events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevy_hanabi_amd as bh
from bevy_hanabi_amd import effects
from bench import frame_seed, DT
cap = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
ctx = bh.Context(0)
prog = ctx.create_program(bh.lower(effects.firework_trails(cap)))
fx = prog.create_effect()
def step(f, spawn=0):
    ctx.frame_begin(DT, f * DT); fx.set_frame(spawn, frame_seed(f)); ctx.simulate()
step(0, cap)
for f in range(1, 6): step(f)
ctx.synchronize()
f = 6
for timing in (False, True, False, True):
    ctx.enable_kernel_timing(timing)
    t0 = time.perf_counter()
    for _ in range(30):
        step(f); f += 1
    ctx.synchronize()
    el = time.perf_counter() - t0
    t_host0 = time.perf_counter()
    print(f"events={timing}: {el/30*1e3:.4f} ms/step", ctx.kernel_timing() if timing else "")
ctx.enable_kernel_timing(False)
# host-side submit cost only
t0 = time.perf_counter()
for _ in range(30):
    step(f); f += 1
t1 = time.perf_counter()
ctx.synchronize()
print(f"host submit {((t1-t0)/30)*1e6:.1f} us/step; alive {fx.alive_count()}")
