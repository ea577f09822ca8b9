"""GPU spawn events: EmitSpawnEventModifier -> child init with InheritAttributeModifier /
parent_attr, one frame later. Oracle semantics are pinned by a hand-evaluated scenario (CPU test); the GPU
tests compare the product with the oracle bit-for-bit, incl. the real three-effect examples/firework.rs."""
import numpy as np
import pytest

import bevy_hanabi_amd as bh
import oracle
from bevy_hanabi_amd import effects
from helpers import A, EffectSpec, Frame, GpuSystem, OracleSystem, assert_same_system_state, frame_seed


def tiny_parent(capacity=8, n_always=2, n_die=3, lifetime=0.05):
    w = bh.ExprWriter()
    init = [bh.SetAttributeModifier(A.POSITION, w.lit((1.0, 2.0, 3.0)).expr()), bh.SetAttributeModifier(A.AGE, w.lit(0.0).expr()),
            bh.SetAttributeModifier(A.LIFETIME, w.lit(lifetime).expr()), bh.SetAttributeModifier(A.U32_0, w.attr(A.ID).expr())]
    always = bh.EmitSpawnEventModifier(bh.EventEmitCondition.Always, w.lit(bh.Value.u32(n_always)).expr(), 0)
    ondie = bh.EmitSpawnEventModifier(bh.EventEmitCondition.OnDie, w.lit(bh.Value.u32(n_die)).expr(), 1)
    a = bh.EffectAsset(capacity, bh.SpawnerSettings.once(4.0), w.finish())
    for m in init:
        a.init(m)
    return a.update(always).update(ondie)


def tiny_child(capacity=64):
    w = bh.ExprWriter()
    init = [bh.InheritAttributeModifier(A.POSITION), bh.SetAttributeModifier(A.U32_1, w.parent_attr(A.U32_0).expr()),
            bh.SetAttributeModifier(A.AGE, w.lit(0.0).expr()), bh.SetAttributeModifier(A.LIFETIME, w.lit(10.0).expr())]
    a = bh.EffectAsset(capacity, bh.SpawnerSettings(), w.finish())
    for m in init:
        a.init(m)
    return a


def tiny_system():
    return [EffectSpec(tiny_parent()), EffectSpec(tiny_child(), parent=0, channel=0, event_capacity=5), EffectSpec(tiny_child(), parent=0, channel=1, event_capacity=256)]


def tiny_frames(n):
    return [[Frame(1 / 60, 4 if f == 0 else 0, 7 + f, time=f / 60)] + [Frame(1 / 60, 0, 100 + f, time=f / 60), Frame(1 / 60, 0, 200 + f, time=f / 60)] for f in range(n)]


def test_oracle_event_semantics_hand_evaluated():
    """4 parents (lifetime 3 frames): every alive parent appends 2 events per frame on channel 0 (capacity 5:
    8 requested, 5 stored, event_count keeps counting: src/lib.rs:976-993); on death 3 events on channel 1.
    Children spawn one frame later (vfx_init.wgsl:123-129) and read the emitting particle (:166-171)."""
    s = OracleSystem(tiny_system())
    P, C0, C1 = s.fx
    fr = tiny_frames(5)
    s.step(fr[0])
    n, ev = P.events(0)
    assert n == 8 and ev[:5].tolist() == [0, 0, 1, 1, 2]
    assert P.events(1)[0] == 0 and C0.alive_count() == 0
    s.step(fr[1])
    assert C0.alive_count() == 5 and P.alive_count() == 4
    s.step(fr[2])  # the parents die: no `Always` events (is_alive is false), 4 x 3 `OnDie` events
    assert P.alive_count() == 0 and P.events(0)[0] == 0
    n, ev = P.events(1)
    assert n == 12 and ev[:12].tolist() == [0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3]
    assert C0.alive_count() == 10 and C1.alive_count() == 0
    s.step(fr[3])
    assert C1.alive_count() == 12 and C0.alive_count() == 10
    np.testing.assert_array_equal(C1.read_attr(A.U32_1.id)[:12, 0], [0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3])   # parent_attr(U32_0) = parent's ID
    assert (C1.read_attr(A.POSITION.id)[:12] == np.array([1, 2, 3], dtype=np.float32)).all()                # inherited, no emitter translation
    s.step(fr[4])
    assert P.events(1)[0] == 0 and C1.alive_count() == 12


def test_event_modifiers_lowering_and_errors():
    blob = bh.lower(tiny_parent())
    txt = bh.disassemble(blob)
    assert txt.count("M_EMIT_EVENTS") == 2 and "aux257" in txt  # channel 1 | OnDie
    assert "LDPARENT" in bh.disassemble(bh.lower(tiny_child()))
    bh.validate_program(blob)
    w = bh.ExprWriter()
    bad = bh.EmitSpawnEventModifier(bh.EventEmitCondition.Always, w.lit(1.0).expr(), 0)   # count must be u32
    pos = w.lit((0.0, 0.0, 0.0)).expr()
    a = bh.EffectAsset(8, bh.SpawnerSettings.once(1.0), w.finish()).init(bh.SetAttributeModifier(A.POSITION, pos)).update(bad)
    with pytest.raises(bh.ExprError):
        bh.lower(a)
    w = bh.ExprWriter()
    far = bh.EmitSpawnEventModifier(bh.EventEmitCondition.Always, w.lit(bh.Value.u32(1)).expr(), 9)
    pos = w.lit((0.0, 0.0, 0.0)).expr()
    a = bh.EffectAsset(8, bh.SpawnerSettings.once(1.0), w.finish()).init(bh.SetAttributeModifier(A.POSITION, pos)).update(far)
    with pytest.raises(bh.ShaderGenerateError):
        bh.lower(a)
    w = bh.ExprWriter()   # parent_attr in update: `parent_particle` does not exist there
    e = w.parent_attr(A.POSITION).expr()
    pos = w.lit((0.0, 0.0, 0.0)).expr()
    a = bh.EffectAsset(8, bh.SpawnerSettings.once(1.0), w.finish()).init(bh.SetAttributeModifier(A.POSITION, pos)).update(bh.SetAttributeModifier(A.POSITION, e))
    with pytest.raises(bh.ExprError):
        bh.lower(a)
    with pytest.raises(bh.PanicError):   # InheritAttributeModifier is init-only (attr.rs:173-186)
        bh.EffectAsset(8, bh.SpawnerSettings.once(1.0), bh.Module()).update(bh.InheritAttributeModifier(A.POSITION))


def firework_system(ev_caps=(256, 256), caps=(32, 1000, 10000), counts=(5, 1000)):
    return [EffectSpec(effects.firework_rocket(caps[0], counts[0], counts[1])),
            EffectSpec(effects.firework_sparkle_trail(caps[1]), parent=0, channel=0, event_capacity=ev_caps[0]),
            EffectSpec(effects.firework_trails_child(caps[2]), parent=0, channel=1, event_capacity=ev_caps[1])]


def firework_frames(n, rate_asset):
    sp = bh.EffectSpawner(rate_asset.spawner)
    rng = bh.Pcg32(3, 5)
    out = []
    for f in range(n):
        t = f / 60
        out.append([Frame(1 / 60, sp.tick(1 / 60, rng), frame_seed(f), time=t), Frame(1 / 60, 0, frame_seed(1000 + f), time=t), Frame(1 / 60, 0, frame_seed(2000 + f), time=t)])
    return out


def test_oracle_real_firework_runs_and_children_spawn():
    specs = firework_system()
    s = OracleSystem(specs)
    for fr in firework_frames(240, specs[0].asset):
        s.step(fr)
    st = s.state()
    assert st[0]["counters"]["particle_counter"] >= 4        # rockets were launched ...
    assert st[1]["counters"]["particle_counter"] > 100       # ... left sparkle trails (5 per rocket per frame, capped by 256 events)
    assert st[2]["counters"]["particle_counter"] >= 256      # ... and exploded (1000 requested, 256 stored per frame: event.rs:267)


# ---- GPU ---------------------------------------------------------------------------------------------------

@pytest.fixture(scope="module")
def ctx():
    c = bh.Context(0)
    yield c
    c.close()


@pytest.fixture(params=["jit", "interp"])
def kernels(request, monkeypatch):
    monkeypatch.setenv("HNB_JIT", "1" if request.param == "jit" else "0")
    return request.param


@pytest.mark.gpu
def test_gpu_tiny_event_system(ctx, kernels):
    specs = tiny_system()
    g, o = GpuSystem(specs, ctx), OracleSystem(specs)
    for f, fr in enumerate(tiny_frames(8)):
        g.step(fr)
        o.step(fr)
        assert_same_system_state(o.state(), g.state(), f"frame {f}")
    g.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("ev_caps", [(256, 256), (4096, 65536)])
def test_gpu_real_firework(ctx, kernels, ev_caps):
    """examples/firework.rs: rocket -> sparkle trail (Always, 5/frame) + explosion trails (OnDie, 1000), with the
    reference's 256-event buffers and with buffers large enough to hold every event."""
    specs = firework_system(ev_caps, caps=(32, 4000, 30000))
    g, o = GpuSystem(specs, ctx), OracleSystem(specs)
    assert "jit" in g.progs[0].kernel_info() if kernels == "jit" else True
    frames = firework_frames(260, specs[0].asset)
    for f, fr in enumerate(frames):
        g.step(fr)
        o.step(fr)
        if f % 20 == 19 or f == len(frames) - 1:
            assert_same_system_state(o.state(), g.state(), f"frame {f}")
    st = g.state()
    assert st[2]["counters"]["particle_counter"] >= 256
    g.destroy()


@pytest.mark.gpu
def test_gpu_many_chunk_parent_with_random_counts(ctx):
    """A 20,000-particle parent (5 chunks) whose particles emit a random 0..3 events each frame and 7 on death:
    exercises the cross-chunk prefix and the per-row scan of k_emit_events, with and without overflow."""
    w = bh.ExprWriter()
    init = [bh.SetPositionSphereModifier(w.lit((0.0, 0.0, 0.0)).expr(), w.lit(2.0).expr(), bh.ShapeDimension.Volume),
            bh.SetAttributeModifier(A.AGE, w.lit(0.0).expr()), bh.SetAttributeModifier(A.LIFETIME, w.lit(0.1).uniform(w.lit(0.4)).expr()),
            bh.SetAttributeModifier(A.F32_0, w.rand(bh.ValueType(bh.ScalarType.Float)).expr())]
    n_rand = (w.rand(bh.ValueType(bh.ScalarType.Float)) * w.lit(3.99)).cast(bh.ValueType(bh.ScalarType.Uint))
    upd = [bh.EmitSpawnEventModifier(bh.EventEmitCondition.Always, n_rand.expr(), 0),
           bh.EmitSpawnEventModifier(bh.EventEmitCondition.OnDie, w.lit(bh.Value.u32(7)).expr(), 1)]
    parent = bh.EffectAsset(20000, bh.SpawnerSettings.once(20000.0), w.finish())
    for m in init:
        parent.init(m)
    for m in upd:
        parent.update(m)

    def child(cap):
        w = bh.ExprWriter()
        mods = [bh.InheritAttributeModifier(A.POSITION), bh.SetAttributeModifier(A.F32_1, w.parent_attr(A.F32_0).expr()),
                bh.SetAttributeModifier(A.AGE, w.lit(0.0).expr()), bh.SetAttributeModifier(A.LIFETIME, w.lit(0.05).expr())]
        a = bh.EffectAsset(cap, bh.SpawnerSettings(), w.finish())
        for m in mods:
            a.init(m)
        return a

    specs = [EffectSpec(parent), EffectSpec(child(150000), parent=0, channel=0, event_capacity=1 << 17), EffectSpec(child(60000), parent=0, channel=1, event_capacity=3000)]
    g, o = GpuSystem(specs, ctx), OracleSystem(specs, omp=True)
    for f in range(30):
        fr = [Frame(1 / 60, 20000 if f in (0, 12) else 0, frame_seed(f), time=f / 60), Frame(1 / 60, 0, frame_seed(500 + f), time=f / 60), Frame(1 / 60, 0, frame_seed(900 + f), time=f / 60)]
        g.step(fr)
        o.step(fr)
        if f in (1, 2, 5, 8, 13, 14, 20, 29):
            assert_same_system_state(o.state(), g.state(), f"frame {f}")
    g.destroy()


@pytest.mark.gpu
def test_gpu_set_parent_validation(ctx):
    rocket = ctx.create_program(bh.lower(effects.firework_rocket())).create_effect()
    child = ctx.create_program(bh.lower(effects.firework_trails_child(1000))).create_effect()
    plain = ctx.create_program(bh.lower(effects.single_particle(16))).create_effect()
    with pytest.raises(bh.HanabiError):   # the parent emits on channels 0 and 1 only
        child.set_parent(rocket, 2)
    with pytest.raises(bh.HanabiError):   # single_particle emits nothing
        child.set_parent(plain, 0)
    with pytest.raises(bh.HanabiError):   # the parent layout lacks U32_0, which the child's init reads
        child.set_parent(ctx.create_program(bh.lower(tiny_parent_without_u32())).create_effect(), 0)
    ctx.frame_begin(1 / 60, 0.0)
    with pytest.raises(bh.HanabiError):   # reads its parent particle but has no parent
        ctx.simulate()
    child.set_parent(rocket, 1)
    ctx.simulate()


def tiny_parent_without_u32():
    w = bh.ExprWriter()
    pos = w.lit((0.0, 0.0, 0.0)).expr()
    emit = bh.EmitSpawnEventModifier(bh.EventEmitCondition.Always, w.lit(bh.Value.u32(1)).expr(), 0)
    return bh.EffectAsset(8, bh.SpawnerSettings.once(1.0), w.finish()).init(bh.SetAttributeModifier(A.POSITION, pos)).update(emit)


@pytest.mark.gpu
def test_gpu_real_firework_slot_order():
    """The three-effect firework with the lists kept in slot order (HNB_LIST_ORDER_SLOT): events are appended in
    list-row order, which now is the slot order, in the product and in the oracle alike."""
    c = bh.Context(0)
    c.set_list_order("slot")
    specs = firework_system((4096, 65536), caps=(32, 4000, 30000))
    g, o = GpuSystem(specs, c), OracleSystem(specs, slot_order=True)
    frames = firework_frames(200, specs[0].asset)
    for f, fr in enumerate(frames):
        g.step(fr)
        o.step(fr)
        if f % 25 == 24 or f == len(frames) - 1:
            assert_same_system_state(o.state(), g.state(), f"frame {f}")
    assert g.state()[2]["counters"]["particle_counter"] > 1000
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [1, 0])
def test_gpu_update_phase_on_two_streams_equals_one_stream_and_the_oracle(overlap):
    """HNB_OPT_OVERLAP_UPDATES: with one program at least four times heavier than the rest (here the 1M-slot trails next to rockets and
    sparkles), the light programs' update phase - update, event ordering, lists - runs on a second stream next to the heavy update and
    joins before hnb_simulate returns. Same state as with one stream, bit for bit, and as the oracle; a consumer enqueued right behind the
    frame (a host read here) sees the joined frame."""
    cap = 1 << 20
    rocket = effects.firework_rocket(8192, 5, 1000)
    rocket.spawner = bh.SpawnerSettings.rate(1000.0)
    specs = [EffectSpec(rocket), EffectSpec(effects.firework_sparkle_trail(1 << 16), parent=0, channel=0, event_capacity=1 << 14),
             EffectSpec(effects.firework_trails_child(cap), parent=0, channel=1, event_capacity=1 << 19)]
    ctx = bh.Context(0)
    ctx.set_option("overlap_updates", overlap)
    g, o = GpuSystem(specs, ctx), OracleSystem(specs, omp=True)
    sp, rng = bh.EffectSpawner(rocket.spawner), bh.Pcg32()
    dt = 0.25
    for f in range(12):
        fr = [Frame(dt, sp.tick(dt, rng), frame_seed(f), time=f * dt), Frame(dt, 0, frame_seed(1000 + f), time=f * dt), Frame(dt, 0, frame_seed(2000 + f), time=f * dt)]
        g.step(fr)
        o.step(fr)
        for fx, ofx in zip(g.fx, o.fx):      # (read straight behind the frame, every frame: the join must cover the side stream's counters)
            assert fx.metadata()["alive_count"] == ofx.alive_count(), f"frame {f}"
        if f in (7, 11):
            assert_same_system_state(o.state(), g.state(), f"overlap={overlap} frame {f}")
    assert g.fx[2].metadata()["particle_counter"] > cap // 4 and all(fx.metadata()["fault"] == 0 for fx in g.fx)
    g.destroy()
    ctx.close()


def six_channel_system():
    """One parent appending to SIX child channels (the reference loops over any number of event bindings, src/lib.rs:964-1002; this library's
    limit is HNB_MAX_EVENT_CHANNELS = 8): channels 0 / 2 / 4 every frame (1, 2, 3 events), channels 1 / 3 / 5 on death (4, 5, 6 events)."""
    w = bh.ExprWriter()
    init = [bh.SetAttributeModifier(A.POSITION, w.lit((1.0, 2.0, 3.0)).expr()), bh.SetAttributeModifier(A.AGE, w.lit(0.0).expr()),
            bh.SetAttributeModifier(A.LIFETIME, w.lit(0.05).expr()), bh.SetAttributeModifier(A.U32_0, w.attr(A.ID).expr())]
    emits = []
    for ch in range(6):
        cond = bh.EventEmitCondition.Always if ch % 2 == 0 else bh.EventEmitCondition.OnDie
        emits.append(bh.EmitSpawnEventModifier(cond, w.lit(bh.Value.u32(ch // 2 + 1 if ch % 2 == 0 else ch // 2 + 4)).expr(), ch))
    parent = bh.EffectAsset(16, bh.SpawnerSettings.once(5.0), w.finish())
    for m in init:
        parent.init(m)
    for m in emits:
        parent.update(m)
    return [EffectSpec(parent)] + [EffectSpec(tiny_child(256), parent=0, channel=ch, event_capacity=64) for ch in range(6)]


def six_channel_frames(n):
    return [[Frame(1 / 60, 5 if f in (0, 4) else 0, 7 + f, time=f / 60)] + [Frame(1 / 60, 0, 100 * (c + 1) + f, time=f / 60) for c in range(6)] for f in range(n)]


def test_oracle_six_event_channels():
    o = OracleSystem(six_channel_system())
    frames = six_channel_frames(8)
    for fr in frames:
        o.step(fr)
    st = o.state()
    # 5 parents live 3 frames (0.05 s at 1/60 s: alive after 1, 2, 3 ticks? age 3/60 = 0.05 is not < 0.05: they die in their third update), twice
    spawned = [s["counters"]["particle_counter"] for s in st]
    assert spawned[0] == 10
    assert spawned[1] > 0 and spawned[3] == 2 * spawned[1] and spawned[5] == 3 * spawned[1]      # Always channels: 1, 2, 3 events per alive parent and frame
    assert spawned[2] == 10 * 4 and spawned[4] == 10 * 5 and spawned[6] == 10 * 6                  # OnDie channels: 4, 5, 6 events per dying parent
    with pytest.raises(Exception):                                                                 # channel 8 is beyond the limit
        w = bh.ExprWriter()
        pos, emit = bh.SetAttributeModifier(A.POSITION, w.lit((0.0, 0.0, 0.0)).expr()), bh.EmitSpawnEventModifier(bh.EventEmitCondition.Always, w.lit(bh.Value.u32(1)).expr(), 8)
        bh.lower(bh.EffectAsset(4, bh.SpawnerSettings.once(1.0), w.finish()).init(pos).update(emit))


@pytest.mark.gpu
def test_gpu_six_event_channels(ctx, kernels):
    specs = six_channel_system()
    g, o = GpuSystem(specs, ctx), OracleSystem(specs)
    for f, fr in enumerate(six_channel_frames(10)):
        g.step(fr)
        o.step(fr)
        assert_same_system_state(o.state(), g.state(), f"six channels frame {f}")
    g.destroy()
