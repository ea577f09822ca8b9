"""Lowering + program interpreters vs the oracle, on the CPU (bit-exact).

The product's lowering (EffectAsset -> uniform/init/update streams) and its interpreters
(hnb_vm.h, compiled for the host by tests/cpu_vm) must reproduce the oracle's serial-order
restatement of the reference exactly: counters, alive/dead lists and every attribute bit.
"""
import numpy as np
import pytest

import bevy_hanabi_amd as bh
from bevy_hanabi_amd import effects
from helpers import A, CpuVmRunner, Frame, OracleRunner, frame_seed, run_script, translation

h = bh


def burst_then_run(capacity, n_frames, dt=1.0 / 60.0, xf=None, spawn=None):
    fr = [Frame(dt, capacity if spawn is None else spawn, frame_seed(0), xf)]
    fr += [Frame(dt, 0, frame_seed(f), xf, time=f * dt) for f in range(1, n_frames)]
    return fr


@pytest.mark.parametrize("force_generic", [False, True])
def test_c1_single_particle(force_generic):
    asset = effects.single_particle(16)
    sp = bh.EffectSpawner(asset.spawner)
    n = sp.tick(1.0 / 60.0, bh.Pcg32())
    assert n == 16  # 1000/60 = 16.67 -> 16
    st = run_script(CpuVmRunner(asset, force_generic=force_generic), [Frame(1 / 60, n, 0)], OracleRunner(asset))
    assert st["counters"]["alive_count"] == 16
    pos = st["attrs"]["position"].view(np.float32)
    assert (pos == np.array([0.1, 0.2, 0.3], dtype=np.float32)).all()
    assert (st["attrs"]["size3"].view(np.float32) == 10.0).all()


def test_c1_capacity_one_caps_spawn():
    asset = effects.single_particle(1)
    st = run_script(CpuVmRunner(asset), [Frame(1 / 60, 16, 0)], OracleRunner(asset))
    assert st["counters"]["alive_count"] == 1 and st["counters"]["max_spawn"] == 0


@pytest.mark.parametrize("force_generic", [False, True])
def test_c2_firework_full_life_cycle(force_generic):
    asset = effects.firework_trails(3000)
    r = CpuVmRunner(asset, force_generic=force_generic)
    assert r.streamable
    # burst, fly, die off (lifetime 0.8..1.2 s), respawn into recycled slots, run again
    frames = burst_then_run(3000, 80) + [Frame(1 / 60, 1234, frame_seed(100))] + [Frame(1 / 60, 0, frame_seed(101 + f)) for f in range(30)]
    st = run_script(r, frames, OracleRunner(asset), every=5)
    assert 0 < st["counters"]["alive_count"] <= 1234


@pytest.mark.parametrize("force_generic", [False, True])
def test_c3_force_field(force_generic):
    asset = effects.force_field(2500)
    r = CpuVmRunner(asset, force_generic=force_generic)
    assert r.streamable
    frames = burst_then_run(2500, 120)
    frames[40].props = {"repulsor_position": (0.1, 0.2, 0.0), "repulsor_accel": -25.0}
    frames[70].props = {"attraction_accel": 35.0, "sticky_factor": 3.0, "shell_half_thickness": 0.2, "max_attraction_speed": 7.0}
    run_script(r, frames, OracleRunner(asset), every=10)


def test_c4_instancing_steady_state_churn():
    cap = 4096
    asset = effects.instancing(cap, rate=cap / 0.5)  # lifetime is 12 s; spawn fast to fill, then churn via short run
    sp = bh.EffectSpawner(asset.spawner)
    rng = bh.Pcg32()
    xf = translation(10.0, -20.0, 30.0)
    frames = [Frame(1 / 60, sp.tick(1 / 60, rng), frame_seed(f), xf) for f in range(50)]
    st = run_script(CpuVmRunner(asset), frames, OracleRunner(asset), every=10)
    assert st["counters"]["alive_count"] == cap


def test_c5_ribbon_spawn_kill_churn():
    cap = 1024
    asset = effects.ribbon(cap)
    sp = bh.EffectSpawner(asset.spawner)
    rng = bh.Pcg32()
    frames = []
    for f in range(200):  # > 1.5 s: constant lifetime => continuous spawn/kill
        t = f / 60.0
        frames.append(Frame(1 / 60, sp.tick(1 / 60, rng), frame_seed(f), translation(np.sin(t), np.cos(t), 0.0), time=t))
    st = run_script(CpuVmRunner(asset), frames, OracleRunner(asset), every=20)
    assert 0 < st["counters"]["alive_count"] < cap
    assert st["counters"]["dead_count"] > 0


# ---- expression / modifier coverage -----------------------------------------------------------

def _expr_asset(build, capacity=257, attrs_update=None):
    w = h.ExprWriter()
    asset = h.EffectAsset(capacity, h.SpawnerSettings.once(float(capacity)), w.finish())
    return build(w)


def _mk(capacity, w, init, update=(), **kw):
    asset = h.EffectAsset(capacity, h.SpawnerSettings.once(float(capacity)), w.finish())
    for m in init:
        asset = asset.init(m)
    for m in update:
        asset = asset.update(m)
    for k, v in kw.items():
        setattr(asset, k, v)
    return asset


F = h.ValueType(h.ScalarType.Float)


def asset_unary_a():
    w = h.ExprWriter()
    x = w.rand(h.VectorType.VEC4F) * w.lit(8.0) - w.lit(4.0)
    init = [
        h.SetAttributeModifier(A.POSITION, w.lit((1.0, 2.0, 3.0)).expr()),
        h.SetAttributeModifier(A.F32X4_0, x.sin().expr()),
        h.SetAttributeModifier(A.F32X4_1, (x.cos() + x.tan() * w.lit(0.001)).expr()),
        h.SetAttributeModifier(A.F32X4_2, (x.exp() + x.exp2()).expr()),
        h.SetAttributeModifier(A.F32X4_3, (x.abs().log() + x.abs().log2() + x.abs().sqrt() + x.abs().inverse_sqrt()).expr()),
    ]
    return _mk(300, w, init)


def asset_unary_b():
    w = h.ExprWriter()
    u = w.rand(h.VectorType.VEC3F)
    init = [
        h.SetAttributeModifier(A.POSITION, u.expr()),
        h.SetAttributeModifier(A.F32X3_0, (u * w.lit(2.0) - w.lit(1.0)).asin().expr()),
        h.SetAttributeModifier(A.F32X3_1, (u * w.lit(2.0) - w.lit(1.0)).acos().expr()),
        h.SetAttributeModifier(A.F32X3_2, (u * w.lit(20.0) - w.lit(10.0)).atan().expr()),
        h.SetAttributeModifier(A.F32X3_3, ((u * w.lit(9.0)).floor() + (u * w.lit(9.0)).ceil() + (u * w.lit(9.0)).fract() + (u * w.lit(9.0)).round()).expr()),
    ]
    return _mk(300, w, init)


def asset_unary_c():
    w = h.ExprWriter()
    x = w.rand(h.VectorType.VEC4F) * w.lit(8.0) - w.lit(4.0)
    xc = x.cast(h.VectorType.VEC4F)   # `vec4<f32>(...)`: a component access needs a functional operand (`(l) - (r).x` is what an infix one prints)
    u = w.rand(h.VectorType.VEC3F)
    init = [
        h.SetAttributeModifier(A.POSITION, u.expr()),
        h.SetAttributeModifier(A.F32X2_0, (xc.x().vec2(xc.y())).saturate().expr()),
        h.SetAttributeModifier(A.F32X2_1, (xc.z().vec2(xc.w())).sign().expr()),
        h.SetAttributeModifier(A.F32_0, x.length().expr()),
        h.SetAttributeModifier(A.F32_1, u.normalized().dot(u).expr()),
        h.SetAttributeModifier(A.F32_2, u.distance(w.lit((0.5, 0.5, 0.5))).expr()),
        h.SetAttributeModifier(A.F32_3, xc.x().atan2(xc.y()).expr()),
        h.SetAttributeModifier(A.COLOR, x.saturate().pack4x8unorm().expr()),
        h.SetAttributeModifier(A.U32_0, (x * w.lit(0.3)).pack4x8snorm().expr()),
        h.SetAttributeModifier(A.HDR_COLOR, w.attr(A.COLOR).unpack4x8unorm().expr()),
    ]
    return _mk(300, w, init)


def asset_binary_a():
    w = h.ExprWriter()
    a = w.rand(h.VectorType.VEC3F) * w.lit(4.0) - w.lit(2.0)
    b = w.rand(h.VectorType.VEC3F) + w.lit(0.25)
    s = w.rand(F)
    init = [
        h.SetAttributeModifier(A.POSITION, (a / b).expr()),
        h.SetAttributeModifier(A.VELOCITY, (a % b).expr()),
        h.SetAttributeModifier(A.F32X3_0, a.max(b).min(w.lit((1.0, 1.0, 1.0))).expr()),
        h.SetAttributeModifier(A.F32X3_1, a.step(b).expr()),
        h.SetAttributeModifier(A.F32X3_2, a.cross(b).expr()),
    ]
    return _mk(300, w, init)


def asset_binary_b():
    w = h.ExprWriter()
    a = w.rand(h.VectorType.VEC3F) * w.lit(4.0) - w.lit(2.0)
    b = w.rand(h.VectorType.VEC3F) + w.lit(0.25)
    s = w.rand(F)
    init = [
        h.SetAttributeModifier(A.POSITION, a.mix(b, s).expr()),
        h.SetAttributeModifier(A.AXIS_X, a.mix(b, b.fract()).expr()),
        h.SetAttributeModifier(A.AXIS_Y, a.clamp(w.lit((-1.0, -0.5, 0.0)), w.lit((0.5, 1.0, 1.5))).expr()),
        h.SetAttributeModifier(A.AXIS_Z, a.smoothstep(w.lit((-1.0, -1.0, -1.0)), w.lit((1.0, 1.0, 1.0))).expr()),
        h.SetAttributeModifier(A.F32_0, (a.cast(h.VectorType.VEC3F).x() * s + b.cast(h.VectorType.VEC3F).y()).expr()),
        h.SetAttributeModifier(A.SIZE3, s.vec3(s * s, w.lit(2.0)).expr()),
    ]
    return _mk(300, w, init)


def asset_binary_c():
    w = h.ExprWriter()
    a = w.rand(h.VectorType.VEC3F) * w.lit(4.0) - w.lit(2.0)
    b = w.rand(h.VectorType.VEC3F) + w.lit(0.25)
    s = w.rand(F)
    init = [
        h.SetAttributeModifier(A.POSITION, b.expr()),
        h.SetAttributeModifier(A.F32X4_0, a.vec4_xyz_w(s).expr()),
        h.SetAttributeModifier(A.F32_1, (a.gt(b).any().cast(F) + a.lt(b).all().cast(F) * w.lit(2.0) + a.ge(b).any().cast(F) * w.lit(4.0) + a.le(b).all().cast(F) * w.lit(8.0)).expr()),
        h.SetAttributeModifier(A.F32_2, (a.cast(h.VectorType.VEC3F).x().cast(h.ValueType(h.ScalarType.Int)).cast(F) + (b.cast(h.VectorType.VEC3F).y() * w.lit(100.0)).cast(h.ValueType(h.ScalarType.Uint)).cast(F)).expr()),
        h.SetAttributeModifier(A.F32_3, w.lit(2.5).uniform(w.lit(7.5)).expr()),
        h.SetAttributeModifier(A.SIZE2, w.lit((1.0, 2.0)).normal(w.lit((0.5, 0.25))).expr()),
        h.SetAttributeModifier(A.ALPHA, w.lit(0.0).normal(w.lit(1.0)).expr()),
    ]
    return _mk(300, w, init)


def asset_integer_zoo():
    w = h.ExprWriter()
    I = h.ValueType(h.ScalarType.Int)
    U = h.ValueType(h.ScalarType.Uint)
    idx = w.attr(A.ID)
    cnt = w.attr(A.PARTICLE_COUNTER)
    i = (w.rand(F) * w.lit(2000.0) - w.lit(1000.0)).cast(I)
    init = [
        h.SetAttributeModifier(A.POSITION, idx.cast(F).vec3(cnt.cast(F), w.lit(0.0)).expr()),
        h.SetAttributeModifier(A.U32_0, (idx * w.lit(h.Value.u32(2654435761)) + cnt).expr()),
        h.SetAttributeModifier(A.U32_1, (idx / w.lit(h.Value.u32(7)) + idx % w.lit(h.Value.u32(5))).expr()),
        h.SetAttributeModifier(A.U32_2, idx.max(w.lit(h.Value.u32(100))).min(w.lit(h.Value.u32(200))).expr()),
        h.SetAttributeModifier(A.U32_3, (idx / w.lit(h.Value.u32(0))).expr()),
        h.SetAttributeModifier(A.SPRITE_INDEX, (i / w.lit(7) + i % w.lit(-3) + i.abs() + i.sign() * w.lit(1000)).expr()),
        h.SetAttributeModifier(A.RIBBON_ID, i.clamp(w.lit(-50), w.lit(50)).cast(U).expr()),
        h.SetAttributeModifier(A.F32_0, i.lt(w.lit(0)).cast(F).expr()),
        h.SetAttributeModifier(A.AGE, w.lit(0.0).expr()),
        h.SetAttributeModifier(A.LIFETIME, w.lit(1.0).expr()),
    ]
    return _mk(300, w, init)


def asset_shapes(dim, space):
    w = h.ExprWriter()
    init = [
        h.SetPositionCircleModifier(w.lit((1.0, 2.0, 3.0)).expr(), w.lit((0.0, 0.6, 0.8)).expr(), w.lit(2.5).expr(), dim),
        h.SetVelocityCircleModifier(w.lit((1.0, 2.0, 3.0)).expr(), w.lit((0.0, 0.6, 0.8)).expr(), (w.rand(F) + w.lit(1.0)).expr()),
        h.SetAttributeModifier(A.F32X3_0, w.attr(A.POSITION).expr()),
        h.SetAttributeModifier(A.F32X3_1, w.attr(A.VELOCITY).expr()),
        h.SetPositionSphereModifier(w.lit((0.0, -1.0, 0.5)).expr(), (w.rand(F) + w.lit(0.5)).expr(), dim),
        h.SetVelocitySphereModifier(w.lit((0.0, -1.0, 0.5)).expr(), w.lit(3.0).expr()),
        h.SetAttributeModifier(A.F32X3_2, w.attr(A.POSITION).expr()),
        h.SetAttributeModifier(A.F32X3_3, w.attr(A.VELOCITY).expr()),
        h.SetPositionCone3dModifier(w.lit(4.0).expr(), w.lit(2.0).expr(), w.lit(0.5).expr(), dim),
        h.SetVelocityTangentModifier(w.lit((0.0, 0.0, 0.0)).expr(), w.lit((0.0, 1.0, 0.0)).expr(), w.lit(1.5).expr()),
        h.SetAttributeModifier(A.AGE, w.lit(0.0).expr()),
        h.SetAttributeModifier(A.LIFETIME, w.lit(0.3).uniform(w.lit(0.9)).expr()),
    ]
    asset = _mk(500, w, init)
    asset.simulation_space = space
    return asset


def asset_update_generic():
    """Update stream with varying operands: must take the generic interpreter."""
    w = h.ExprWriter()
    accel_prop = w.add_property("my_accel", (0.0, -3.0, 0.0))
    init = [
        h.SetPositionSphereModifier(w.lit((0.0, 0.0, 0.0)).expr(), w.lit(2.0).expr(), h.ShapeDimension.Volume),
        h.SetVelocitySphereModifier(w.lit((0.0, 0.0, 0.0)).expr(), w.lit(1.0).expr()),
        h.SetAttributeModifier(A.AGE, w.lit(0.0).expr()),
        h.SetAttributeModifier(A.LIFETIME, w.lit(0.5).uniform(w.lit(1.5)).expr()),
        h.SetAttributeModifier(A.F32_0, w.rand(F).expr()),
    ]
    jitter = w.rand(h.VectorType.VEC3F) - w.lit(0.5)
    update = [
        h.AccelModifier((w.prop(accel_prop) + jitter * w.attr(A.F32_0)).expr()),
        h.RadialAccelModifier(w.attr(A.F32_0).vec3(w.lit(0.0), w.lit(0.0)).expr(), (w.time() * w.lit(0.1) + w.lit(1.0)).expr()),
        h.TangentAccelModifier((w.lit((0.0, 0.0, 0.0)) + w.lit((0.1, 0.0, 0.0))).expr(), w.lit((0.0, 1.0, 0.0)).expr(), w.attr(A.AGE).expr()),
        h.LinearDragModifier((w.attr(A.AGE) * w.lit(2.0)).expr()),
        h.KillSphereModifier(w.lit((0.0, 0.0, 0.0)).expr(), (w.attr(A.F32_0) * w.lit(30.0) + w.lit(4.0)).expr()),
        h.KillAabbModifier(w.attr(A.F32_0).vec3(w.lit(0.0), w.lit(0.0)).expr(), w.lit((0.05, 0.05, 0.05)).expr(), True),
        h.SetAttributeModifier(A.F32_0, (w.attr(A.F32_0) * w.lit(0.99)).expr()),
        h.SetAttributeModifier(A.COLOR, w.attr(A.VELOCITY).normalized().vec4_xyz_w(w.lit(1.0)).pack4x8unorm().expr()),
    ]
    return _mk(700, w, init, update)


def asset_update_streamable_all_macros():
    w = h.ExprWriter()
    p = w.add_property("wind", (0.5, 0.0, 0.25))
    k = w.add_property("k", 0.75)
    init = [
        h.SetPositionSphereModifier(w.lit((0.0, 0.0, 0.0)).expr(), w.lit(1.0).expr(), h.ShapeDimension.Volume),
        h.SetVelocitySphereModifier(w.lit((0.0, 0.0, 0.0)).expr(), w.lit(2.0).expr()),
        h.SetAttributeModifier(A.AGE, w.lit(0.0).expr()),
        h.SetAttributeModifier(A.LIFETIME, w.lit(0.4).uniform(w.lit(2.0)).expr()),
    ]
    update = [
        h.AccelModifier((w.prop(p) * w.time().sin()).expr()),
        h.AccelModifier(w.prop(k).expr()),  # scalar acceleration broadcasts
        h.RadialAccelModifier(w.lit((0.0, 1.0, 0.0)).expr(), w.prop(k).expr()),
        h.TangentAccelModifier(w.lit((0.0, 0.0, 0.0)).expr(), w.lit((0.0, 1.0, 0.0)).expr(), w.lit(3.0).expr()),
        h.LinearDragModifier(w.prop(k).expr()),
        h.ConformToSphereModifier(w.lit((0.2, 0.0, 0.0)).expr(), w.lit(1.5).expr(), w.lit(3.0).expr(), w.lit(6.0).expr(), w.lit(2.0).expr()),
        h.KillSphereModifier(w.lit((0.0, 0.0, 0.0)).expr(), w.lit(9.0).expr()),
        h.KillAabbModifier(w.lit((0.0, -0.7, 0.0)).expr(), w.lit((0.1, 0.1, 0.1)).expr(), True),
        h.SetAttributeModifier(A.LIFETIME, w.lit(1.9).expr()),
    ]
    asset = _mk(900, w, init, update)
    asset.motion_integration = h.MotionIntegration.PreUpdate
    return asset


def asset_ribbons_multi():
    """Five interleaved ribbons with random lifetimes: particles die in the middle of a ribbon, so the
    post-update sort by (RIBBON_ID, AGE) (vfx_sort*.wgsl) really reorders the alive list."""
    w = h.ExprWriter()
    U = h.ValueType(h.ScalarType.Uint)
    init = [
        h.SetPositionSphereModifier(w.lit((0.0, 0.0, 0.0)).expr(), w.lit(1.0).expr(), h.ShapeDimension.Surface),
        h.SetAttributeModifier(A.AGE, (w.rand(F) * w.lit(0.05)).expr()),
        h.SetAttributeModifier(A.LIFETIME, w.lit(0.1).uniform(w.lit(0.6)).expr()),
        h.SetAttributeModifier(A.RIBBON_ID, (w.rand(F) * w.lit(4.999)).cast(U).expr()),
        h.SetAttributeModifier(A.SIZE, w.lit(0.5).expr()),
    ]
    asset = _mk(6000, w, init, [])
    asset.motion_integration = h.MotionIntegration.None_
    return asset


def asset_ribbons_rewritten_keys():
    """The update program rewrites AGE and RIBBON_ID at random: last frame's order is worthless, the
    ribbon sort has to order the whole list every frame (the head check of hnb_sort.hip.h fails)."""
    w = h.ExprWriter()
    U = h.ValueType(h.ScalarType.Uint)
    init = [
        h.SetPositionSphereModifier(w.lit((0.0, 0.0, 0.0)).expr(), w.lit(1.0).expr(), h.ShapeDimension.Surface),
        h.SetAttributeModifier(A.AGE, w.lit(0.0).expr()),
        h.SetAttributeModifier(A.LIFETIME, w.lit(5.0).expr()),
        h.SetAttributeModifier(A.RIBBON_ID, w.lit(h.Value.u32(0)).expr()),
    ]
    update = [
        h.SetAttributeModifier(A.AGE, (w.rand(F) * w.lit(3.0)).expr()),
        h.SetAttributeModifier(A.RIBBON_ID, (w.rand(F) * w.lit(2.999)).cast(U).expr()),
    ]
    asset = _mk(9000, w, init, update)
    asset.motion_integration = h.MotionIntegration.None_
    return asset


ZOO = {
    "ribbons_multi": asset_ribbons_multi,
    "ribbons_rewritten_keys": asset_ribbons_rewritten_keys,
    "unary_a": asset_unary_a,
    "unary_b": asset_unary_b,
    "unary_c": asset_unary_c,
    "binary_a": asset_binary_a,
    "binary_b": asset_binary_b,
    "binary_c": asset_binary_c,
    "integer": asset_integer_zoo,
    "shapes_surface_global": lambda: asset_shapes(h.ShapeDimension.Surface, h.SimulationSpace.Global),
    "shapes_volume_local": lambda: asset_shapes(h.ShapeDimension.Volume, h.SimulationSpace.Local),
    "update_generic": asset_update_generic,
    "update_all_macros": asset_update_streamable_all_macros,
}


@pytest.mark.parametrize("name", sorted(ZOO))
def test_zoo_matches_oracle(name):
    asset = ZOO[name]()
    cap = asset.capacity
    xf = np.array([0.0, -1.0, 0.0, 4.0, 1.0, 0.0, 0.0, -2.0, 0.0, 0.0, 1.0, 0.5], dtype=np.float32)  # rotation about z + translation
    frames = [Frame(1 / 60, cap // 2, frame_seed(0), xf)]
    for f in range(1, 90):
        frames.append(Frame(1 / 60 if f % 7 else 1 / 30, (cap // 9) if f % 11 == 0 else 0, frame_seed(f), xf, time=f / 60.0))
    frames[30].props = {"k": 1.25} if name == "update_all_macros" else {}
    r = CpuVmRunner(asset)
    if name == "update_generic":
        assert not r.streamable
    if name == "update_all_macros":
        assert r.streamable
    run_script(r, frames, OracleRunner(asset), every=10)
    if r.streamable:  # the same stream through the generic interpreter must agree as well
        run_script(CpuVmRunner(asset, force_generic=True), frames, OracleRunner(asset), every=30)


def test_rand_expression_shared_between_statements_is_drawn_once():
    """modifier/mod.rs:1411-1432: a side-effect expression is hoisted to one `let varN`."""
    w = h.ExprWriter()
    r = w.rand(h.VectorType.VEC3F)
    asset = _mk(64, w, [h.SetAttributeModifier(A.POSITION, r.expr()), h.SetAttributeModifier(A.VELOCITY, r.expr()),
                        h.SetAttributeModifier(A.F32X3_0, (r + r).expr())])
    asset.motion_integration = h.MotionIntegration.None_
    st = run_script(CpuVmRunner(asset), [Frame(1 / 60, 64, 7)], OracleRunner(asset))
    np.testing.assert_array_equal(st["attrs"]["position"], st["attrs"]["velocity"])


def test_rand_inside_function_modifier_is_redrawn():
    """modifier/mod.rs:339-349: function-style modifiers evaluate with a fresh expression cache."""
    w = h.ExprWriter()
    r = w.rand(F)
    asset = _mk(64, w, [h.SetAttributeModifier(A.F32_0, r.expr()),
                        h.SetPositionSphereModifier(w.lit((0.0, 0.0, 0.0)).expr(), r.expr(), h.ShapeDimension.Surface),
                        h.SetAttributeModifier(A.F32_1, r.expr())])
    st = run_script(CpuVmRunner(asset), [Frame(1 / 60, 64, 9)], OracleRunner(asset))
    f0 = st["attrs"]["f32_0"].view(np.float32)[:, 0]
    radius = np.linalg.norm(st["attrs"]["position"].view(np.float32).astype(np.float64), axis=1)
    np.testing.assert_array_equal(st["attrs"]["f32_0"], st["attrs"]["f32_1"])  # main writer reuses var0
    assert np.abs(radius - f0).max() > 1e-3  # the sphere drew its own radius


def test_origin_expression_is_pasted_without_parentheses():
    """accel.rs:176: `position - {origin}` with an infix origin parses as (position - l) + r."""
    w = h.ExprWriter()
    origin = w.lit((1.0, 0.0, 0.0)) + w.lit((0.0, 2.0, 0.0))
    asset = _mk(32, w, [h.SetAttributeModifier(A.POSITION, w.lit((3.0, 1.0, 0.5)).expr()), h.SetAttributeModifier(A.VELOCITY, w.lit((0.0, 0.0, 0.0)).expr())],
                [h.RadialAccelModifier(origin.expr(), w.lit(60.0).expr())])
    asset.motion_integration = h.MotionIntegration.None_
    st = run_script(CpuVmRunner(asset), [Frame(1 / 60, 32, 1)], OracleRunner(asset))
    v = st["attrs"]["velocity"].view(np.float32)[0].astype(np.float64)
    expect = np.array([2.0, 3.0, 0.5]) / np.linalg.norm([2.0, 3.0, 0.5])  # (3,1,.5) - (1,0,0) + (0,2,0)
    np.testing.assert_allclose(v, expect, rtol=1e-6)


def test_slot_base_shifts_prng_and_id():
    asset = effects.firework_trails(512)
    whole = run_script(OracleRunner(effects.firework_trails(1024)), [Frame(1 / 60, 1024, 5)])
    lo = run_script(CpuVmRunner(asset, slot_base=0), [Frame(1 / 60, 512, 5)])
    hi = run_script(CpuVmRunner(asset, slot_base=512), [Frame(1 / 60, 512, 5)])
    for k in whole["attrs"]:
        np.testing.assert_array_equal(whole["attrs"][k][:512], lo["attrs"][k])
        np.testing.assert_array_equal(whole["attrs"][k][512:], hi["attrs"][k])
