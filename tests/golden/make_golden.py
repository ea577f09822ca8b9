"""Generates the committed fixtures of this directory from the oracle (run: python tests/golden/make_golden.py).

  oracle_states.json  sha256 of the full oracle state (counters, lists, attribute planes) of each
                      config after a fixed frame script: detects any drift of the oracle itself.
  states_small.npz    full final state arrays of small configs; the GPU tests compare the product
                      against these without calling the oracle.

The frame scripts only use explicit (spawn_count, seed) inputs and CpuValue::Single spawners, as
SURVEY.md §8c prescribes (the reference's own RNG crates are unpinned).
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import bevy_hanabi_amd as bh  # noqa: E402
from bevy_hanabi_amd import effects  # noqa: E402
from helpers import Frame, OracleRunner, frame_seed, translation  # noqa: E402


def scripts():
    """name -> (asset, frames)"""
    out = {}
    out["c1_single_particle_16"] = (effects.single_particle(16), [Frame(1 / 60, 16, 0)])
    cap = 5000
    fr = [Frame(1 / 60, cap, frame_seed(0))] + [Frame(1 / 60, 0, frame_seed(f), time=f / 60) for f in range(1, 60)]
    fr += [Frame(1 / 60, 1234, frame_seed(60), time=1.0)] + [Frame(1 / 60, 0, frame_seed(f), time=f / 60) for f in range(61, 70)]
    out["c2_firework_5000"] = (effects.firework_trails(cap), fr)
    cap = 6000
    fr = [Frame(1 / 60, cap, frame_seed(0))] + [Frame(1 / 60, 0, frame_seed(f), time=f / 60) for f in range(1, 50)]
    fr[20].props = {"repulsor_position": (0.1, 0.2, 0.0), "repulsor_accel": -25.0}
    out["c3_force_field_6000"] = (effects.force_field(cap), fr)
    cap = 4500
    asset = effects.instancing(cap, rate=cap / 0.25)
    sp = bh.EffectSpawner(asset.spawner)
    rng = bh.Pcg32()
    out["c4_instancing_4500"] = (asset, [Frame(1 / 60, sp.tick(1 / 60, rng), frame_seed(f), translation(10.0, -5.0, 0.5), time=f / 60) for f in range(40)])
    cap = 5000
    asset = effects.ribbon(cap)
    sp = bh.EffectSpawner(asset.spawner)
    fr = []
    for f in range(150):
        t = f / 60.0
        fr.append(Frame(1 / 60, sp.tick(1 / 60, rng), frame_seed(f), translation(np.sin(t), np.cos(t), 0.0), time=t))
    out["c5_ribbon_5000"] = (asset, fr)
    return out


def final_state(asset, frames):
    r = OracleRunner(asset)
    for fr in frames:
        r.step(fr)
    return r.state()


def state_hash(st):
    h = hashlib.sha256()
    h.update(json.dumps(st["counters"], sort_keys=True).encode())
    h.update(np.ascontiguousarray(st["alive"], dtype=np.uint32).tobytes())
    h.update(np.ascontiguousarray(st["dead"], dtype=np.uint32).tobytes())
    for name in sorted(st["attrs"]):
        h.update(name.encode())
        h.update(np.ascontiguousarray(st["attrs"][name], dtype=np.uint32).tobytes())
    return h.hexdigest()


def compute_all():
    return {name: state_hash(final_state(a, fr)) for name, (a, fr) in scripts().items()}


SMALL = ("c1_single_particle_16", "c2_firework_5000", "c3_force_field_6000")


def main():
    sc = scripts()
    json.dump(compute_all(), open(os.path.join(HERE, "oracle_states.json"), "w"), indent=1, sort_keys=True)
    arrays = {}
    for name in SMALL:
        st = final_state(*sc[name])
        arrays[f"{name}/alive"] = st["alive"]
        arrays[f"{name}/dead"] = st["dead"]
        arrays[f"{name}/counters"] = np.array([st["counters"][k] for k in sorted(st["counters"])], dtype=np.uint32)
        for an, v in st["attrs"].items():
            arrays[f"{name}/attr/{an}"] = v
    np.savez_compressed(os.path.join(HERE, "states_small.npz"), **arrays)
    print("wrote", os.path.join(HERE, "oracle_states.json"), os.path.join(HERE, "states_small.npz"))


if __name__ == "__main__":
    main()
