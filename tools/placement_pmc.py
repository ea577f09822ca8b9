"""Development tool (GPU box): per-slab hardware counters of the firework update on N 16,777,216-particle slabs alive at once,
each walked 12 frames in one direction and 12 frames in alternating directions. Run under
`rocprofv3 --kernel-trace --pmc <counters>`; tools/placement_pmc_report.py turns the counter CSVs into a table."""
import os, sys
os.environ["HNB_SLAB_CANDIDATES"] = "1"
os.environ["HNB_CTX_OPTIONS"] = "skip_lists=0"   # (read by the Python binding, not by the library)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevy_hanabi_amd as bh
from bevy_hanabi_amd import effects
from bench import frame_seed, frame_dt

cap = 1 << 24
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
blob = bh.lower(effects.firework_trails(cap))
dt = frame_dt(400)
items = []
for i in range(n):
    ctx = bh.Context(0)
    prog = ctx.create_program(blob)
    fx = prog.create_effect()
    ctx.frame_begin(dt, 0.0); fx.set_frame(cap, frame_seed(0)); ctx.simulate(); ctx.synchronize()
    items.append((ctx, prog, fx))
f = 1
for ctx, prog, fx in items:
    for alt in (0, 1):
        ctx.set_option(2, alt)
        for _ in range(12):
            ctx.frame_begin(dt, f * dt); fx.set_frame(0, frame_seed(f)); ctx.simulate(); f += 1
        ctx.synchronize()
print("done", n)
