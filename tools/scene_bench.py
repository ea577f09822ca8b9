"""A scene of many different small effects in one context: every single-entity effect of the reference's examples/
(bevy_hanabi_amd/reference_examples.py), COPIES of each program, all simulated by one hnb_simulate per frame.

This is the launch-bound regime (a few thousand particles per effect): what it measures is the fixed cost per program and
frame — host time inside simulate(), and wall time per frame with the device kept busy.
  python tools/scene_bench.py [copies] [frames]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

import bevy_hanabi_amd as bh  # noqa: E402
from bevy_hanabi_amd import reference_examples as rx  # noqa: E402

DT = 1.0 / 60.0


def run(copies=1, frames=600, device=0, quiet=False):
    ctx = bh.Context(device)
    players = []
    for c in range(copies):
        for name, entries in sorted(rx.catalog().items()):
            if any(e.parent is not None for e in entries):
                continue
            for e in entries:
                prog = ctx.create_program(bh.lower(e.asset))
                players.append({"e": e, "fx": prog.create_effect(), "sp": bh.EffectSpawner(e.asset.spawner), "rng": bh.Pcg32(), "seed": 17 + len(players)})
    if not quiet:
        print(f"scene: {len(players)} effects / programs in one context")

    def frame(f):
        t = f * DT
        ctx.frame_begin(DT, t)
        for p in players:
            e = p["e"]
            for k, v in e.drive(f, t, p["sp"]).items():
                p["fx"].set_property(k, v)
            xf = e.transform(f, t) if callable(e.transform) else e.transform
            p["seed"] = bh.next_prng_seed(p["seed"])
            p["fx"].set_frame(p["sp"].tick(DT, p["rng"]), p["seed"], None if xf is None else np.asarray(xf, dtype=np.float32))
        t0 = time.perf_counter()
        ctx.simulate()
        return time.perf_counter() - t0

    for f in range(60):
        frame(f)
    ctx.synchronize()
    host = 0.0
    t0 = time.perf_counter()
    for f in range(60, 60 + frames):
        host += frame(f)
    ctx.synchronize()
    wall = time.perf_counter() - t0
    alive = sum(p["fx"].alive_count() for p in players)
    if not quiet:
        print(f"{frames} frames: {wall / frames * 1e3:.3f} ms per frame wall (python driving included), {host / frames * 1e3:.3f} ms inside simulate(); "
              f"{host / frames / len(players) * 1e6:.1f} us of simulate() per effect and frame; {alive} particles alive at the end")
    ctx.close()
    return {"effects": len(players), "frames": frames, "ms_per_frame_wall": wall / frames * 1e3, "ms_per_frame_in_simulate": host / frames * 1e3,
            "us_of_simulate_per_effect": host / frames / len(players) * 1e6, "alive_at_end": alive,
            "workload": "every single-entity effect of the reference's examples/ (one program + one instance each) in one context, one hnb_simulate per frame"}


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 600)
