"""A scene of many different small effects in one context: every single-entity effect of the reference's examples/
(bevy_hanabi_amd/reference_examples.py), COPIES of each program, all simulated by one hnb_simulate per frame.

This is the launch-bound regime (a few thousand particles per effect): what it measures is the fixed cost per program and
frame — host time inside simulate(), and wall time per frame with the device kept busy.
  python tools/scene_bench.py [copies] [frames]
"""
import ctypes as C
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


def run(copies=1, frames=600, device=0, quiet=False, set_module=None):
    """set_module: None = the context's default (a set module if the jit cache holds one: tools/warm_jit_cache.py), or a HNB_OPT_SET_MODULE value"""
    blobs = [bh.lower(e.asset) for name, entries in sorted(rx.catalog().items()) if all(x.parent is None for x in entries) for e in entries]
    t_pre = time.perf_counter()
    if set_module != 0:   # what an application does behind a loading screen (or on its build box): a cache hit where tools/warm_jit_cache.py ran
        bh.jit_precompile_set(blobs)
    t_pre = time.perf_counter() - t_pre
    ctx = bh.Context(device)
    if set_module is not None:
        ctx.set_option("set_module", set_module)
    players = []
    for c in range(copies):
        for name, entries in sorted(rx.catalog().items()):
            if any(e.parent is not None for e in entries):
                continue
            for e in entries:
                prog = ctx.create_program(bh.lower(e.asset))
                players.append({"e": e, "prog": prog, "fx": prog.create_effect(), "sp": bh.EffectSpawner(e.asset.spawner), "rng": bh.Pcg32(), "seed": 17 + len(players)})
    if not quiet:
        print(f"scene: {len(players)} effects / programs in one context")

    # The frame inputs (spawner ticks, property drives, transforms: what bevy_hanabi's tick_spawners + the example's systems produce) are
    # RECORDED first and replayed as bare C-ABI calls: the numbers below are the library's frame, not the Python that computes its inputs (26
    # effects x numpy conversions cost more than the frame: 0.05 ms). A native host pays nanoseconds per call.
    lib, keep = ctx._lib, []
    warm = 60

    def record(f):
        t = f * DT
        calls = [(lib.hnb_frame_begin, (ctx._h, C.byref(bh.SimParams(DT, t, DT, t, DT, t))))]
        for p in players:
            e, h = p["e"], p["fx"]._h
            for k, v in e.drive(f, t, p["sp"]).items():
                w = np.atleast_1d(np.asarray(v))
                w = np.ascontiguousarray(w.astype(np.float32).view(np.uint32) if w.dtype.kind == "f" else w.astype(np.uint32))
                keep.append(w)
                calls.append((lib.hnb_effect_set_property, (h, k.encode(), w.ctypes.data, len(w))))
            xf = e.transform(f, t) if callable(e.transform) else e.transform
            if xf is not None:
                xf = np.ascontiguousarray(np.asarray(xf, dtype=np.float32).reshape(12))
                keep.append(xf)
            p["seed"] = bh.next_prng_seed(p["seed"])
            calls.append((lib.hnb_effect_set_frame, (h, int(p["sp"].tick(DT, p["rng"])), p["seed"] & 0xFFFFFFFF, None if xf is None else xf.ctypes.data)))
        return calls

    script = [record(f) for f in range(warm + frames)]
    sim, h_ctx, perf = lib.hnb_simulate, ctx._h, time.perf_counter

    def frame(f):
        rc = 0
        for fn, a in script[f]:
            rc |= fn(*a)
        t0 = perf()
        rc |= sim(h_ctx)
        dt = perf() - t0
        assert rc == 0, lib.hnb_last_error().decode()
        return dt

    for f in range(warm):
        frame(f)
    ctx.synchronize()
    host = 0.0
    t0 = time.perf_counter()
    for f in range(warm, warm + frames):
        host += frame(f)
    ctx.synchronize()
    wall = time.perf_counter() - t0
    alive = sum(p["fx"].alive_count() for p in players)
    in_set = sum(1 for p in players if "set module (the program" in p["prog"].kernel_info())
    waited = [l for l in players[0]["prog"].kernel_info().split("\n") if l.startswith("hnb_simulate waited")]
    if not quiet and waited:
        print(waited[0])
    if not quiet:
        print(f"{frames} frames: {wall / frames * 1e3:.3f} ms per frame wall (recorded inputs replayed through the C ABI), {host / frames * 1e3:.3f} ms inside simulate(); "
              f"{host / frames / len(players) * 1e6:.1f} us of simulate() per effect and frame; {alive} particles alive at the end")
    ctx.close()
    return {"effects": len(players), "frames": frames, "ms_per_frame_wall": wall / frames * 1e3, "ms_per_frame_in_simulate": host / frames * 1e3,
            "us_of_simulate_per_effect": host / frames / len(players) * 1e6, "alive_at_end": alive, "programs_served_by_the_set_module": in_set, "set_module_precompile_s": round(t_pre, 3),
            "workload": "every single-entity effect of the reference's examples/ (one program + one instance each) in one context, one hnb_simulate per frame"}


if __name__ == "__main__":
    for mode in (0, None):
        r = run(int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 600, set_module=mode)
        print("set_module =", mode, "->", r["programs_served_by_the_set_module"], "programs served by the set module")
