"""Development tool: run only the burst frame (k_init) a few times for profiling."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevy_hanabi_amd as bh
from bevy_hanabi_amd import effects
from bench import frame_seed, DT
cap = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
which = sys.argv[2] if len(sys.argv) > 2 else "firework"
asset = {"firework": effects.firework_trails, "force_field": effects.force_field, "instancing": effects.instancing}[which](cap)
ctx = bh.Context(0)
prog = ctx.create_program(bh.lower(asset))
ctx.enable_kernel_timing(1)
for rep in range(3):
    fx = prog.create_effect()
    ctx.frame_begin(DT, 0.0); fx.set_frame(cap, frame_seed(0)); ctx.simulate()
    ctx.synchronize()
    assert fx.alive_count() == cap
    fx.destroy()
print(which, cap, ctx.kernel_timing())
