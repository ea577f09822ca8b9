"""Spawn nursery (hnb_kernels.hip.h "Spawn nursery", HNB_OPT_SPAWN_NURSERY): spawns into scattered free slots travel to the update as
32-byte records (nursery[slot], announced by alive byte 5) instead of plane-granular stores. Same state after every frame as the direct path
and as the oracle, bit for bit: sparse spawns into a churning effect, spawns of up to an eighth of the capacity into the dead list a die-off
left (several records per lane of the update, taken one after the other), re-bursts (stored directly: the per-frame rule), spawns into chunks
that keep their ages in a cohort word, several instances, ragged capacities.
"""
import re

import numpy as np
import pytest

import bevy_hanabi_amd as bh
from bevy_hanabi_amd import effects
from helpers import A, Frame, GpuRunner, OracleRunner, assert_same_state, frame_seed

pytestmark = pytest.mark.gpu


def _ctx(nursery, cohort=None):
    c = bh.Context(0)
    c.set_option("spawn_nursery", 1 if nursery else 0)
    if cohort is not None:
        c.set_option("age_cohort", cohort)
    return c


def _churn_asset(cap, life_lo=0.02, life_hi=0.2):
    w = bh.ExprWriter()
    color = (w.rand(bh.VectorType.VEC3F) * w.lit(0.9) + w.lit(0.1)).vec4_xyz_w(w.lit(1.0)).pack4x8unorm()
    mods = [bh.SetPositionSphereModifier(w.lit((0.0, 0.0, 0.0)).expr(), w.lit(1.0).expr(), bh.ShapeDimension.Volume),
            bh.SetVelocitySphereModifier(w.lit((0.0, 0.0, 0.0)).expr(), w.lit(2.0).expr()),
            bh.SetAttributeModifier(A.AGE, w.lit(0.0).expr()),
            bh.SetAttributeModifier(A.LIFETIME, w.lit(life_lo).uniform(w.lit(life_hi)).expr()),
            bh.SetAttributeModifier(A.COLOR, color.expr())]
    upd = [bh.LinearDragModifier(w.lit(2.0).expr()), bh.AccelModifier(w.lit((0.0, -5.0, 0.0)).expr())]
    asset = bh.EffectAsset(cap, bh.SpawnerSettings.once(1.0), w.finish())
    for m in mods:
        asset.init(m)
    for m in upd:
        asset.update(m)
    return asset


def _stats(prog):
    m = re.search(r"spawn nursery: (\d+) of (\d+) sampled records written, (\d+) records waiting", prog.kernel_info())
    return None if m is None else tuple(int(x) for x in m.groups())


@pytest.mark.parametrize("cap", [257, 4097, 20037])
@pytest.mark.parametrize("cohort", [0, 1])
def test_scattered_spawns_through_records_equal_the_direct_path_and_the_oracle(cap, cohort):
    """Random spawn requests into an effect whose particles die at random ages: the dead list is in death order within a few frames."""
    asset = _churn_asset(cap)
    on, off = _ctx(True, cohort), _ctx(False, cohort)
    g_on, g_off, o = GpuRunner(asset, ctx=on), GpuRunner(asset, ctx=off), OracleRunner(asset)
    assert _stats(g_on.prog) is not None and _stats(g_off.prog) is None
    rng = np.random.default_rng(cap + cohort)
    for f in range(64):
        r = rng.random()
        spawn = cap if f == 0 else (0 if r < 0.2 else (cap * 2 if r > 0.93 else int(rng.integers(1, max(2, cap // 3)))))
        fr = Frame(1 / 60, spawn, frame_seed(f + cap), time=f / 60)
        for x in (g_on, g_off, o):
            x.step(fr)
        if f % 7 == 6 or f == 63:
            ref = o.state()
            assert_same_state(ref, g_on.state(), f"nursery on, frame {f}")
            assert_same_state(ref, g_off.state(), f"nursery off, frame {f}")
    used, groups, waiting = _stats(g_on.prog)
    assert waiting == 0, "a bucket was left armed between frames"
    assert used > 0, "no spawn ever went through a record: the test does not test what it says"
    on.close()
    off.close()


def test_a_dense_reburst_into_the_dead_list_of_a_die_off():
    """Burst, die-off at random ages, then (a) a burst of the whole capacity into the dead list in death order - stored plane by plane: a frame that
    spawns more than an eighth of the slots does not use records - and (b) spawns of just under an eighth per frame, frame after frame: records,
    many lanes of the update owning two or more of them."""
    cap = 3 * 4096 + 1234
    asset = _churn_asset(cap, 0.05, 0.4)
    on = _ctx(True)
    g, o = GpuRunner(asset, ctx=on), OracleRunner(asset)
    frames = [Frame(1 / 60, cap, frame_seed(0))] + [Frame(1 / 60, 0, frame_seed(f), time=f / 60) for f in range(1, 30)]
    frames += [Frame(1 / 60, cap, frame_seed(30), time=0.5)] + [Frame(1 / 60, 0, frame_seed(f), time=f / 60) for f in range(31, 36)]
    frames += [Frame(1 / 60, 0, frame_seed(f), time=f / 60) for f in range(36, 60)]                       # ... and dies off again
    frames += [Frame(1 / 60, cap // 8, frame_seed(f), time=f / 60) for f in range(60, 72)]                # an eighth per frame into the second dead list
    for i, fr in enumerate(frames):
        g.step(fr)
        o.step(fr)
        if i in (0, 12, 29, 30, 31, 35, 59, 60, 61, 65, 71):
            assert_same_state(o.state(), g.state(), f"frame {i}")
        if i in (29, 59):
            assert g.fx.alive_count() == 0, "the die-off must be complete before the re-burst"
        if i == 35:
            assert _stats(g.prog)[0] == 0, "the re-burst must not have gone through records"
    used, sampled, waiting = _stats(g.prog)
    assert waiting == 0 and used > sampled // 4, (used, sampled)
    on.close()


def test_spawns_into_chunks_that_keep_their_ages_in_a_cohort_word():
    """The firework burst (one age for everybody: cohort state 1), a few casualties, then small spawns into the freed slots: state 2 with the
    fresh particles' ages coming from their records."""
    cap = 5 * 4096
    asset = effects.firework_trails(cap)
    on, off = _ctx(True), _ctx(False)
    g_on, g_off, o = GpuRunner(asset, ctx=on), GpuRunner(asset, ctx=off), OracleRunner(asset)
    frames = [Frame(1 / 60, cap, frame_seed(0))] + [Frame(1 / 60, 0, frame_seed(f), time=f / 60) for f in range(1, 50)]
    for f in range(50, 90):
        frames.append(Frame(1 / 60, 150 if f % 3 else 0, frame_seed(f), time=f / 60))
    for i, fr in enumerate(frames):
        for x in (g_on, g_off, o):
            x.step(fr)
        if i in (0, 49, 50, 51, 55, 70, 89):
            ref = o.state()
            assert_same_state(ref, g_on.state(), f"nursery on, frame {i}")
            assert_same_state(ref, g_off.state(), f"nursery off, frame {i}")
    used, groups, waiting = _stats(g_on.prog)
    assert waiting == 0 and used > 0
    on.close()
    off.close()


def test_several_instances_and_a_frozen_one():
    """Buckets are per instance; an instance that is not simulated spawns nothing and keeps its buckets empty."""
    cap = 6000
    asset = _churn_asset(cap)
    on = _ctx(True)
    prog = on.create_program(bh.lower(asset))
    fxs = [prog.create_effect() for _ in range(3)]
    orcs = [OracleRunner(asset) for _ in range(3)]
    rng = np.random.default_rng(7)
    for f in range(40):
        on.frame_begin(1 / 60, f / 60)
        for k, (fx, oc) in enumerate(zip(fxs, orcs)):
            frozen = k == 1 and 10 <= f < 20
            fx.set_simulated(not frozen)
            spawn = cap if f == 0 else int(rng.integers(0, 900))
            seed = frame_seed(f * 3 + k)
            fx.set_frame(spawn, seed)
            if not frozen:
                oc.step(Frame(1 / 60, spawn, seed, time=f / 60))
        on.simulate()
        if f in (9, 19, 20, 39):
            for k, (fx, oc) in enumerate(zip(fxs, orcs)):
                ref = oc.state()
                m = fx.metadata()
                assert m["alive_count"] == ref["counters"]["alive_count"], (f, k)
                np.testing.assert_array_equal(fx.alive_list(), ref["alive"], err_msg=f"frame {f} instance {k}")
                for a in (A.POSITION, A.VELOCITY, A.AGE, A.LIFETIME, A.COLOR):
                    np.testing.assert_array_equal(fx.read_attr(a.id).view(np.uint32), ref["attrs"][a.name], err_msg=f"frame {f} instance {k} {a.name}")
    used, groups, waiting = _stats(prog)
    assert waiting == 0 and used > 0
    on.close()
