"""Development probe for the pipelined general path (DESIGN.md "What comes next"): how much of the c2_mixed frame's serial chain
(k_init -> update -> k_count_rows -> k_compact) can overlap with ANOTHER frame's work on this part?

Two contexts on one GPU, each with the firework trails effect in the spawn / die steady state, each on its own stream. Measured:
  A alone, B alone          one effect per frame (the chain as it is today)
  A and B interleaved       frame f of A and frame f of B enqueued back to back on their two streams: the device is free to run A's
                            lists beside B's update, B's init beside A's update, ... - the overlap a pipelined schedule of ONE effect
                            would get, without its data hazards
If the pair takes clearly less than the sum (e.g. 1.6x one frame), the chain's idle bandwidth is there to be had; if it takes the sum,
the update kernel already saturates what the other kernels need and pipelining buys nothing.
  CAP=16777216 python tools/overlap_probe.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

import bevy_hanabi_amd as bh  # noqa: E402
from bench import DT, frame_seed  # noqa: E402
from bevy_hanabi_amd import effects  # noqa: E402

cap = int(os.environ.get("CAP", str(1 << 24)))
warm = int(os.environ.get("WARM", "300"))
frames = int(os.environ.get("FRAMES", "120"))


class One:
    def __init__(self):
        self.ctx = bh.Context(0)
        asset = effects.firework_trails(cap, bh.SpawnerSettings.rate(float(cap) / 1.0))
        self.prog = self.ctx.create_program(bh.lower(asset))
        self.fx = self.prog.create_effect()
        self.sp, self.rng, self.f = bh.EffectSpawner(asset.spawner), bh.Pcg32(), 0

    def step(self):
        self.ctx.frame_begin(DT, self.f * DT)
        self.fx.set_frame(self.sp.tick(DT, self.rng), frame_seed(self.f))
        self.ctx.simulate()
        self.f += 1


def timed(effects_, n):
    for e in effects_:
        e.ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for e in effects_:
            e.step()
    for e in effects_:
        e.ctx.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


a, b = One(), One()
for _ in range(warm):
    a.step()
    b.step()
for rep in range(3):
    ta, tb, tab = timed([a], frames), timed([b], frames), timed([a, b], frames)
    print(f"cap {cap}: A alone {ta:.4f} ms/frame, B alone {tb:.4f}, A and B interleaved {tab:.4f} per pair = {tab / (ta + tb):.3f} of the sum "
          f"({tab / 2:.4f} per effect frame)", flush=True)
print("alive", a.fx.metadata()["alive_count"], b.fx.metadata()["alive_count"])
a.ctx.close()
b.ctx.close()
