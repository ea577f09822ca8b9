"""Development tool: the firework trails program in a spawn / die steady state (rate spawner), per-stage kernel times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import bevy_hanabi_amd as bh
from bevy_hanabi_amd import effects
from bench import frame_seed, DT

cap = int(os.environ.get("CAP", str(1 << 24)))
warm = int(os.environ.get("WARM", "300"))
frames = int(os.environ.get("FRAMES", "60"))
order = os.environ.get("ORDER", "spawn")
ctx = bh.Context(0)
ctx.set_list_order(order)
asset = effects.firework_trails(cap, bh.SpawnerSettings.rate(float(cap) / 1.0))
prog = ctx.create_program(bh.lower(asset)); fx = prog.create_effect()
sp = bh.EffectSpawner(asset.spawner); rng = bh.Pcg32()
f = 0
def step():
    global f
    ctx.frame_begin(DT, f * DT); fx.set_frame(sp.tick(DT, rng), frame_seed(f)); ctx.simulate(); f += 1
for _ in range(warm): step()
ctx.synchronize()
m0 = fx.metadata()
for rep in range(3):
    ctx.enable_kernel_timing(1)
    t0 = time.perf_counter()
    for _ in range(frames): step()
    ctx.synchronize()
    el = time.perf_counter() - t0
    tm = ctx.kernel_timing(); ctx.enable_kernel_timing(0)
    m = fx.metadata()
    n = m["max_update"]
    k = tm["update_ms_avg"] + tm["compact_ms_avg"] + tm["init_ms_avg"]
    print(f"mixed cap {cap} order {order}: alive {m['alive_count']} max_update {n} spawned/frame {(m['particle_counter'] - m0['particle_counter']) / frames / (rep + 1):.0f} | wall(events on) {el / frames * 1e3:.4f} ms | init {tm['init_ms_avg']:.4f} update {tm['update_ms_avg']:.4f} lists {tm['compact_ms_avg']:.4f} = {k:.4f} ms -> {n * 68 / (k * 1e-3) / 1e12:.3f} TB/s @68B = {n * 68 / (k * 1e-3) / 8e12:.3f} of 8 TB/s", flush=True)
# untimed-events wall
t0 = time.perf_counter()
for _ in range(frames): step()
ctx.synchronize()
el = time.perf_counter() - t0
m = fx.metadata()
print(f"wall without events: {el / frames * 1e3:.4f} ms/frame -> {m['max_update'] * 68 / (el / frames) / 8e12:.3f} of 8 TB/s @68B; {m['max_update'] / (el / frames):.3e} updates/s")
print(prog.kernel_info())
ctx.close()
