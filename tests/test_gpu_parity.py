"""GPU parity (the product path: lowering -> C ABI -> HIP kernels) against the oracle.

Bit-exact on every attribute, counter and list: the arithmetic of both sides is pinned by
the hanabi-math definition, so the 1e-5 relative tolerance BASELINE.json allows on
position/velocity is met with zero difference.
"""
import numpy as np
import pytest

import bevy_hanabi_amd as bh
from bevy_hanabi_amd import effects
from helpers import A, Frame, GpuRunner, OracleRunner, assert_same_state, frame_seed, run_script, translation
from test_lowering_cpu import ZOO, burst_then_run

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = bh.Context(0)
    yield c
    c.close()


def test_native_library_is_the_one_running(ctx):
    import os
    maps = open("/proc/self/maps").read()
    assert "libhanabi_amd.so" in maps and os.path.dirname(bh.__file__) in maps


def test_c1_single_particle(ctx):
    asset = effects.single_particle(16)
    st = run_script(GpuRunner(asset, ctx=ctx), [Frame(1 / 60, 16, 0)], OracleRunner(asset))
    assert st["counters"]["alive_count"] == 16
    assert (st["attrs"]["position"].view(np.float32) == np.array([0.1, 0.2, 0.3], dtype=np.float32)).all()


def test_c2_firework_life_cycle(ctx):
    cap = 20000  # 5 chunks: exercises the cross-chunk look-back
    asset = effects.firework_trails(cap)
    frames = burst_then_run(cap, 80) + [Frame(1 / 60, 7777, frame_seed(100))] + [Frame(1 / 60, 0, frame_seed(101 + f)) for f in range(30)]
    st = run_script(GpuRunner(asset, ctx=ctx), frames, OracleRunner(asset), every=6)
    assert 0 < st["counters"]["alive_count"] <= 7777


def test_c3_force_field(ctx):
    cap = 30000
    asset = effects.force_field(cap)
    frames = burst_then_run(cap, 100)
    frames[40].props = {"repulsor_position": (0.1, 0.2, 0.0), "repulsor_accel": -25.0}
    run_script(GpuRunner(asset, ctx=ctx), frames, OracleRunner(asset), every=10)


def test_c4_instancing_batch_of_instances(ctx):
    """Several instances of one program: one init + one update launch for the whole batch."""
    cap, n_inst = 9000, 5
    asset = effects.instancing(cap, rate=cap / 0.25)
    blob = bh.lower(asset)
    prog = ctx.create_program(blob)
    fxs = [prog.create_effect() for _ in range(n_inst)]
    oracles = [OracleRunner(asset) for _ in range(n_inst)]
    spawners = [bh.EffectSpawner(asset.spawner) for _ in range(n_inst)]
    rng = bh.Pcg32()
    for f in range(40):
        ctx.frame_begin(1 / 60, f / 60)
        for i, (fx, orc, sp) in enumerate(zip(fxs, oracles, spawners)):
            n = sp.tick(1 / 60, rng) if (f + i) % 3 else 0
            xf = translation(10.0 * i, -5.0, 0.5 * i)
            seed = frame_seed(f * 16 + i)
            fx.set_frame(n, seed, xf)
            orc.step(Frame(1 / 60, n, seed, xf, time=f / 60))
        ctx.simulate()
    for fx, orc in zip(fxs, oracles):
        ref = orc.state()
        np.testing.assert_array_equal(ref["alive"], fx.alive_list())
        np.testing.assert_array_equal(ref["dead"], fx.dead_list())
        assert ref["counters"]["alive_count"] == fx.alive_count()
        for a in (A.POSITION, A.VELOCITY, A.AGE, A.LIFETIME):
            np.testing.assert_array_equal(ref["attrs"][a.name], fx.read_attr(a.id).view(np.uint32))
    prog.destroy()


def test_c5_ribbon_churn(ctx):
    cap = 12000
    asset = effects.ribbon(cap)
    sp = bh.EffectSpawner(asset.spawner)
    rng = bh.Pcg32()
    frames = []
    for f in range(200):
        t = f / 60.0
        frames.append(Frame(1 / 60, sp.tick(1 / 60, rng), frame_seed(f), translation(np.sin(t), np.cos(t), 0.0), time=t))
    st = run_script(GpuRunner(asset, ctx=ctx), frames, OracleRunner(asset), every=25)
    assert st["counters"]["dead_count"] > 0


@pytest.mark.parametrize("name", sorted(ZOO))
def test_zoo(ctx, name):
    asset = ZOO[name]()
    cap = asset.capacity
    xf = np.array([0.0, -1.0, 0.0, 4.0, 1.0, 0.0, 0.0, -2.0, 0.0, 0.0, 1.0, 0.5], dtype=np.float32)
    frames = [Frame(1 / 60, cap // 2, frame_seed(0), xf)]
    for f in range(1, 60):
        frames.append(Frame(1 / 60 if f % 7 else 1 / 30, (cap // 9) if f % 11 == 0 else 0, frame_seed(f), xf, time=f / 60.0))
    run_script(GpuRunner(asset, ctx=ctx), frames, OracleRunner(asset), every=10)


def test_reference_contract_vectors_through_the_abi(ctx):
    """shader_contract_tests.rs:1254-1486 (vfx_indirect) replayed end to end: capacities (200, 5),
    alive (130, 1) -> max_update (130, 1), max_spawn (70, 4), instance_count after an update that
    kills nothing (130, 1)."""
    for cap, alive in ((200, 130), (5, 1)):
        asset = effects.single_particle(cap)  # only POSITION/SIZE3: nothing dies (cf. shader_contract_tests.rs:888)
        r = GpuRunner(asset, ctx=ctx)
        r.step(Frame(1 / 60, alive, 3))
        r.step(Frame(1 / 60, 0, 4))
        m = r.fx.metadata()
        assert (m["max_update"], m["max_spawn"], m["instance_count"], m["alive_count"]) == (alive, cap - alive, alive, alive)
        assert m["dispatch_x"] == (alive + 63) // 64
        np.testing.assert_array_equal(r.fx.alive_list(), np.arange(alive, dtype=np.uint32))


def test_large_burst_properties(ctx):
    """BASELINE config size (16M): size-independent checks — every slot allocated exactly once,
    alive + dead == capacity, ages advance by dt, and a sampled window matches the oracle."""
    cap = 1 << 24
    asset = effects.firework_trails(cap)
    r = GpuRunner(asset, ctx=ctx)
    seeds = [frame_seed(f) for f in range(4)]
    for f in range(4):
        r.step(Frame(1 / 60, cap if f == 0 else 0, seeds[f]))
    m = r.fx.metadata()
    assert m["alive_count"] == cap and m["dead_count"] == 0 and m["fault"] == 0
    alive = r.fx.alive_list()
    assert np.array_equal(alive, np.arange(cap, dtype=np.uint32))
    age = r.fx.read_attr(A.AGE.id)[:, 0]
    expect = np.float32(0)
    for _ in range(4):
        expect = np.float32(expect + np.float32(1 / 60))
    assert (age == expect).all()
    # window check: slots [base, base+4096) of the big effect == a 4096-capacity oracle with slot_base
    base = 12345 * 1024
    orc = OracleRunner(effects.firework_trails(4096), slot_base=base)
    for f in range(4):
        orc.step(Frame(1 / 60, 4096 if f == 0 else 0, seeds[f]))
    ref = orc.state()
    for a in (A.POSITION, A.VELOCITY, A.LIFETIME, A.COLOR):
        np.testing.assert_array_equal(ref["attrs"][a.name], r.fx.read_attr(a.id).view(np.uint32)[base:base + 4096])
    r.fx.destroy()
    r.prog.destroy()
