"""Round 6: is a small launch bound by fetching its CODE from HBM? The init kernel of a small effect (256 spawns per frame into 8192 slots: microseconds of work,
3 KB of code) timed with HIP events (a) alone in its context, (b) in a context whose other effect streams 1 GB per frame through the L2s (16.7M firework trails).
    python tools/r06_cold_code.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import bevy_hanabi_amd as bh  # noqa: E402
from bevy_hanabi_amd import effects  # noqa: E402
from bench import frame_seed  # noqa: E402


def run(with_big, frames=200):
    ctx = bh.Context(0)
    small = ctx.create_program(bh.lower(effects.firework_trails(8192, bh.SpawnerSettings.rate(256 * 60.0))))
    fs = small.create_effect()
    fb = None
    if with_big:
        big = ctx.create_program(bh.lower(effects.firework_trails(1 << 24)))
        fb = big.create_effect()
    dt = 1e-3
    for f in range(frames + 20):
        if f == 20:
            ctx.synchronize()
            ctx.enable_kernel_timing(1)
        ctx.frame_begin(dt, f * dt)
        fs.set_frame(256, frame_seed(f))
        if fb is not None:
            fb.set_frame((1 << 24) if f == 0 else 0, frame_seed(1000 + f))
        ctx.simulate()
    t = small.kernel_timing()
    ctx.enable_kernel_timing(0)
    ctx.close()
    return t


for rep in range(2):
    for wb in (False, True):
        t = run(wb)
        print(f"rep {rep} small effect {'beside the 16.7M effect' if wb else 'alone'}: init {t['init_ms_avg'] * 1e3:.2f} us, update {t['update_ms_avg'] * 1e3:.2f} us, lists {t['compact_ms_avg'] * 1e3:.2f} us ({t['frames']} frames)", flush=True)
