"""Generates tests/golden/example_digests.json from the oracle (run: python tests/golden/make_example_digests.py).

One sha256 per effect of the reference's examples (bevy_hanabi_amd/reference_examples.py) and checkpoint frame, over the full state
(counters, alive list, dead list, every stored attribute plane; NaNs of float attributes canonicalised, since WGSL leaves their bits
unspecified). The scripts are those of tests/test_reference_examples.py (spawner ticks with the host Pcg32 mirror, the examples' own
per-frame systems). CPU tests: the oracle still produces these digests (drift of the checker itself). GPU tests: the product produces
them WITHOUT the oracle in the loop.
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
from bevy_hanabi_amd import reference_examples as rx  # noqa: E402

CHECKPOINTS = (0, 29, 89, 149)
PATH = os.path.join(HERE, "example_digests.json")


def digest(state, asset):
    h = hashlib.sha256()
    h.update(json.dumps({k: int(v) for k, v in state["counters"].items()}, sort_keys=True).encode())
    h.update(np.ascontiguousarray(state["alive"], dtype=np.uint32).tobytes())
    h.update(np.ascontiguousarray(state["dead"], dtype=np.uint32).tobytes())
    for name in sorted(state["attrs"]):
        a = np.ascontiguousarray(state["attrs"][name], dtype=np.uint32).copy()
        attr = bh.Attribute.from_name(name)
        if attr is not None and attr.value_type.elem == bh.ScalarType.Float:
            nan = ((a & 0x7F800000) == 0x7F800000) & ((a & 0x007FFFFF) != 0)
            a[nan] = 0x7FC00000
        h.update(name.encode())
        h.update(a.tobytes())
    return h.hexdigest()


def play(make_runner, names=None):
    """{ "<example>[<i>]": {frame: digest} } for every single-entity example effect, through `make_runner(asset)`."""
    from test_reference_examples import Player   # the example scripts
    out = {}
    cat = rx.catalog()
    for name in sorted(cat):
        if names is not None and name not in names:
            continue
        if any(e.parent is not None for e in cat[name]):
            continue
        for index, entry in enumerate(cat[name]):
            run, player = make_runner(entry.asset), Player(entry, index)
            d = {}
            for f in range(CHECKPOINTS[-1] + 1):
                fr = player.frame(f)
                if fr is not None:
                    run.step(fr)
                if f in CHECKPOINTS:
                    d[str(f)] = digest(run.state(), entry.asset)
            out[f"{name}[{index}]"] = d
            if hasattr(run, "prog"):
                run.prog.destroy()
    return out


def main():
    from helpers import OracleRunner
    data = play(OracleRunner)
    with open(PATH, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
    print(f"wrote {len(data)} effects x {len(CHECKPOINTS)} checkpoints to {PATH}")


if __name__ == "__main__":
    main()
