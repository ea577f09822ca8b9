"""Every effect of the reference's examples/ (bevy_hanabi_amd/reference_examples.py), played frame by frame the way the example's
own systems drive it: spawner ticks on the host (EffectSpawner mirror + Pcg32), property updates, transforms, spawner resets,
visibility toggles, parent -> child spawn events.

CPU: the product's lowering, executed by the product's interpreter built for the host (tests/cpu_vm), against the oracle,
bit for bit after every frame; RON round trip and the C authoring ABI reproduce the same program. GPU: the same scripts through
the C ABI (HIP kernels) against the oracle, including the two-effect worms system.
"""
import numpy as np
import pytest

import bevy_hanabi_amd as bh
from bevy_hanabi_amd import reference_examples as rx
from helpers import (CpuVmRunner, EffectSpec, Frame, GpuRunner, GpuSystem, OracleRunner, OracleSystem, assert_same_state, assert_same_system_state,
                     frame_seed)

DT = 1.0 / 60.0
CATALOG = rx.catalog()
SINGLE = sorted(k for k, v in CATALOG.items() if all(e.parent is None for e in v))
FRAMES = {"lifetime": 420, "lightning": 200, "random": 400, "circle": 90, "visibility": 200, "spawn_on_command": 150, "activate": 420,
          "puffs": 240, "spawn": 220}


class Player:
    """Host side of one ParticleEffect entity: what bevy_hanabi's tick_spawners + the example's systems produce per frame."""

    def __init__(self, entry, index):
        self.entry, self.index = entry, index
        self.spawner = bh.EffectSpawner(entry.asset.spawner)
        self.rng = bh.Pcg32()

    def frame(self, f):
        """Frame (or None while SimulationCondition::WhenVisible freezes the invisible instance, spawn.rs:983-991)."""
        e, t = self.entry, f * DT
        props = e.drive(f, t, self.spawner)
        visible = e.visible(f, t)
        if not visible and e.asset.simulation_condition == bh.SimulationCondition.WhenVisible:
            self.pending = getattr(self, "pending", {})
            self.pending.update(props)
            return None
        props = {**getattr(self, "pending", {}), **props}
        self.pending = {}
        xf = e.transform(f, t) if callable(e.transform) else e.transform
        n = 0 if e.parent is not None else self.spawner.tick(DT, self.rng)
        return Frame(DT, n, frame_seed(f * 16 + self.index), None if xf is None else np.asarray(xf, dtype=np.float32), time=t,
                     props=props)


def play_single(name, make_runner, frames, check_every=1):
    """Each effect of a single-effect-per-entity example, oracle against `make_runner`."""
    stats = []
    for index, entry in enumerate(rx.catalog()[name]):   # a fresh catalog: some drives carry state (the ball of activate.rs)
        orc, run, player = OracleRunner(entry.asset), make_runner(entry.asset), Player(entry, index)
        spawned = peak = 0
        for f in range(frames):
            fr = player.frame(f)
            if fr is None:
                continue
            orc.step(fr)
            run.step(fr)
            spawned += fr.spawn
            if (f + 1) % check_every == 0 or f == frames - 1:
                ref = orc.state()
                assert_same_state(ref, run.state(), f"{name}[{index}] frame {f}")
                peak = max(peak, ref["counters"]["alive_count"])
        stats.append((spawned, peak, orc.state()["counters"]["alive_count"]))
    return stats


@pytest.mark.parametrize("name", SINGLE)
def test_example_lowering_matches_the_oracle(name):
    frames = FRAMES.get(name, 120)
    stats = play_single(name, CpuVmRunner, frames, check_every=1 if frames <= 150 else 7)
    for spawned, peak, _ in stats:
        assert spawned > 0 and peak > 0, f"{name}: the script never spawned anything"


def test_examples_exercise_what_they_are_about():
    """The scripts reach the behaviour each example demonstrates (not just 'something ran')."""
    # lifetime.rs: 12 s particles accumulate over bursts, 0.75 s particles are gone before the next burst
    s = play_single("lifetime", CpuVmRunner, 420, check_every=60)
    assert s[0][2] == s[0][0] and s[2][2] == 0 and 0 < s[1][2] <= 100
    # instancing.rs (alternate): no AGE initialiser, but ColorOverLifetimeModifier puts AGE into the layout (default 0): particles
    # live 5 s and 102 / s keep about 510 of the 512 slots busy
    s = play_single("instancing_alternate", CpuVmRunner, 400, check_every=50)
    assert 505 <= s[0][2] <= 512 and s[0][0] > 512
    # spawn_on_command.rs: nothing before the first reset, 100 per reset afterwards
    s = play_single("spawn_on_command", CpuVmRunner, 150, check_every=10)
    assert s[0][0] == 100 * len([f for f in range(150) if f % 23 == 5])
    # activate.rs: the spawner only runs while the ball is below y = 0
    p = Player(rx.catalog()["activate"][0], 0)
    counts = [p.frame(f).spawn for f in range(420)]
    assert sum(counts[:115]) == 0 and sum(counts[125:230]) > 50 and sum(counts[240:350]) == 0 and sum(counts[360:]) > 0   # up 2 s, down 2 s, ...
    # visibility.rs: the WhenVisible instance misses the frames it was hidden for, the Always instance does not
    s = play_single("visibility", CpuVmRunner, 200, check_every=20)
    assert s[0][0] == s[1][0] == 50
    # random.rs: CpuValue::Uniform counts and periods come from the host PRNG
    s = play_single("random", CpuVmRunner, 400, check_every=40)
    assert 2 <= s[0][0] <= 100 * 7


def test_lightning_bolt_is_a_pure_function_of_the_counter_and_the_seed():
    """lightning.rs:86-170: 40 particles per strike, RIBBON_ID = strike number, AGE offsets keep the ribbon order, the jitter
    vanishes at both ends of the bolt (weight 4 t (1 - t))."""
    entry = rx.catalog()["lightning"][0]
    orc, player = OracleRunner(entry.asset), Player(entry, 0)
    for f in range(10):
        fr = player.frame(f)
        orc.step(fr)
    st = orc.state()
    n = rx.LIGHTNING_PARTICLES_PER_BOLT
    assert st["counters"]["particle_counter"] == 2 * n   # emit on start + the first timer strike 0.1 s later
    pos = orc.fx.read_attr(bh.Attribute.POSITION.id)[: 2 * n]
    rid = orc.fx.read_attr(bh.Attribute.RIBBON_ID.id).view(np.uint32)[: 2 * n].ravel()
    assert set(rid[:n]) == {0} and set(rid[n:]) == {1}
    top = rx.LIGHTNING_GROUND_Y + rx.LIGHTNING_BOLT_LENGTH
    for b in range(2):
        y = pos[b * n:(b + 1) * n, 1]
        np.testing.assert_allclose(y, top - np.arange(n, dtype=np.float32) / np.float32(n - 1) * rx.LIGHTNING_BOLT_LENGTH, rtol=1e-6)
        assert pos[b * n, 0] == 0 and pos[b * n, 2] == 0 and abs(pos[(b + 1) * n - 1, 0]) < 1e-5
        assert np.abs(pos[b * n:(b + 1) * n, 0]).max() > 0.05
    assert not np.array_equal(pos[:n, 0], pos[n:, 0])   # a different wave_seed, a different shape


@pytest.mark.parametrize("name", sorted(CATALOG))
def test_example_assets_survive_ron_and_the_c_abi(name):
    """RON round trip (asset.rs serde derives) and re-lowering give the identical program blob."""
    for entry in CATALOG[name]:
        blob = bh.lower(entry.asset)
        back = bh.from_ron(bh.to_ron(entry.asset))
        assert bh.lower(back) == blob
        assert bh.serialize_asset(back) == bh.serialize_asset(entry.asset)


def play_system(name, frames, make_system=None):
    entries = rx.catalog()[name]
    specs = [EffectSpec(e.asset, parent=e.parent, channel=e.channel) for e in entries]
    orc, run = OracleSystem(specs), make_system(specs) if make_system else None
    players = [Player(e, i) for i, e in enumerate(entries)]
    for f in range(frames):
        frs = [p.frame(f) for p in players]
        orc.step(frs)
        if run:
            run.step(frs)
            assert_same_system_state(orc.state(), run.state(), f"{name} frame {f}")
    return orc.state()


def check_worms(state):
    heads, bodies = state
    assert heads["counters"]["alive_count"] >= 4
    # 5 bodies per alive head and frame, 1.5 s lifetime -> 450 per head in steady state
    assert bodies["counters"]["alive_count"] > 1000
    rid = bodies["attrs"]["ribbon_id"].ravel()[bodies["alive"]]
    assert set(np.unique(rid)) <= set(range(heads["counters"]["particle_counter"]))
    assert (np.diff(rid.astype(np.int64)) >= 0).all()   # the alive list is grouped by ribbon
    # every body inherited its head's colour: one colour per ribbon
    col = bodies["attrs"]["color"].ravel()[bodies["alive"]]
    for r in np.unique(rid):
        assert len(np.unique(col[rid == r])) == 1


def test_worms_system_in_the_oracle():
    """worms.rs on the oracle alone (the two-effect GPU run is test_gpu_worms_heads_and_bodies)."""
    check_worms(play_system("worms", 150))


# ---- GPU: the same scripts through the C ABI -------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ctx():
    c = bh.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", SINGLE)
def test_gpu_example_matches_the_oracle(name, ctx):
    frames = FRAMES.get(name, 120)
    made = []

    def make(asset):
        r = GpuRunner(asset, ctx=ctx)
        made.append(r)
        return r

    try:
        stats = play_single(name, make, frames, check_every=1 if frames <= 150 else 7)
        assert all(sp > 0 and peak > 0 for sp, peak, _ in stats)
    finally:
        for r in made:
            r.prog.destroy()


@pytest.mark.gpu
def test_gpu_worms_heads_and_bodies(ctx):
    """worms.rs: heads (rate 2) steer by sin(time) and emit 5 events per head and frame; bodies inherit position and colour and
    take RIBBON_ID from the head's U32_0, so every body ribbon is one head's trail, sorted by age."""
    sys_ = []

    def make(specs):
        s = GpuSystem(specs, ctx)
        sys_.append(s)
        return s

    try:
        st = play_system("worms", 150, make)
    finally:
        for s in sys_:
            s.destroy()
    check_worms(st)


# ---- committed digests of the example runs (tests/golden/example_digests.json, made by tests/golden/make_example_digests.py) ---------
def _load_digests():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "example_digests.json")) as f:
        return json.load(f)


def test_oracle_reproduces_the_committed_example_digests():
    """Drift of the checker itself: the oracle's full state of every example effect at frames 0, 29, 89 and 149."""
    from golden.make_example_digests import play
    assert play(OracleRunner) == _load_digests()


@pytest.mark.gpu
def test_gpu_reproduces_the_committed_example_digests(ctx):
    """The product against the committed digests, no oracle in the loop."""
    from golden.make_example_digests import play
    got = play(lambda asset: GpuRunner(asset, ctx=ctx))
    want = _load_digests()
    assert got.keys() == want.keys()
    bad = [(k, f) for k in want for f in want[k] if got[k][f] != want[k][f]]
    assert not bad, bad
