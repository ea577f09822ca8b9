"""The reference's example assets, restated through the mirrored authoring API.

These are the configurations of BASELINE.json / SURVEY.md §8(d): synthetic scalings of
assets the reference ships (the reference never runs them at these sizes, and has no
CPU simulation path — SURVEY.md §0).
"""
from . import _hanabi_host as h

A = h.Attribute


def single_particle(capacity=16):
    """gpu_tests/single_particle.rs:37-45 (C1)."""
    module = h.Module()
    pos = module.lit((0.1, 0.2, 0.3))
    size = module.lit((10.0, 10.0, 10.0))
    asset = (h.EffectAsset(capacity, h.SpawnerSettings.rate(1000.0), module)
             .init(h.SetAttributeModifier(A.POSITION, pos))
             .init(h.SetAttributeModifier(A.SIZE3, size)))
    asset.name = "test_asset"
    return asset


def firework_trails(capacity=1 << 24, spawner=None):
    """The `trails` program of examples/firework.rs:184-251 made self-contained (C2).

    `spawner`: None = the burst `once(capacity)` of SURVEY.md §8(d) C2; bench.py's c2_mixed passes a rate spawner
    (capacity / mean lifetime per second) so that the same program runs in a spawn / age / die steady state.

    InheritAttributeModifier(POSITION) -> POSITION = lit(0,0,0); parent_attr(U32_0) colour ->
    the rocket's colour expression (firework.rs:64-66). Update: LinearDrag(4) then
    Accel((0,-16,0)) (firework.rs:239-240), PostUpdate Euler.
    """
    w = h.ExprWriter()
    init_pos = h.SetAttributeModifier(A.POSITION, w.lit((0.0, 0.0, 0.0)).expr())
    center = w.attr(A.POSITION)
    speed = w.lit(40.0).uniform(w.lit(60.0))
    direction = w.rand(h.VectorType.VEC3F).mul(w.lit(2.0)).sub(w.lit(1.0)).normalized()
    init_vel = h.SetAttributeModifier(A.VELOCITY, (center + direction * speed).expr())
    init_age = h.SetAttributeModifier(A.AGE, w.lit(0.0).expr())
    init_lifetime = h.SetAttributeModifier(A.LIFETIME, w.lit(0.8).uniform(w.lit(1.2)).expr())
    color = (w.rand(h.VectorType.VEC3F) * w.lit(0.9) + w.lit(0.1)).vec4_xyz_w(w.lit(1.0)).pack4x8unorm()
    init_color = h.SetAttributeModifier(A.COLOR, color.expr())
    # `Vec3::Y * -16.` yields (-0., -16., -0.)
    update_accel = h.AccelModifier(w.lit((-0.0, -16.0, -0.0)).expr())
    update_drag = h.LinearDragModifier(w.lit(4.0).expr())
    return (h.EffectAsset(capacity, spawner if spawner is not None else h.SpawnerSettings.once(float(capacity)), w.finish())
            .with_name("trail")
            .init(init_pos).init(init_vel).init(init_age).init(init_lifetime).init(init_color)
            .update(update_drag).update(update_accel)
            .render(h.ColorOverLifetimeModifier())
            .render(h.SizeOverLifetimeModifier())
            .render(h.OrientModifier(h.OrientMode.AlongVelocity)))


BALL_RADIUS = 0.05
ATTRACTOR_POS = (0.01, 0.0, 0.0)
REPULSOR_POS = (0.3, 0.5, 0.0)


def force_field(capacity=1 << 23, emit_on_start=True):
    """examples/force_field.rs:126-209 (C3): 2x ConformToSphere + KillAabb + KillSphere, 6 properties."""
    spawner = h.SpawnerSettings.once(float(capacity)).with_emit_on_start(emit_on_start)
    w = h.ExprWriter()
    init_age = h.SetAttributeModifier(A.AGE, w.lit(0.0).expr())
    init_lifetime = h.SetAttributeModifier(A.LIFETIME, w.lit(10.0).expr())
    allow_zone = h.KillAabbModifier(w.lit((0.0, 0.0, 0.0)).expr(), w.lit((3.0, 2.0, 3.0)).expr())
    radius = w.lit(0.6)
    deny_zone = h.KillSphereModifier(w.lit((-2.0, 1.0, 0.0)).expr(), (radius * radius).expr(), True)
    init_pos = h.SetPositionSphereModifier(w.lit((0.0, 0.0, 0.0)).expr(), w.lit(BALL_RADIUS).expr(), h.ShapeDimension.Surface)
    init_vel = h.SetVelocitySphereModifier(w.lit((0.0, 0.0, 0.0)).expr(),
                                           (w.rand(h.ValueType(h.ScalarType.Float)) * w.lit(0.2) + w.lit(0.1)).expr())
    repulsor_accel = w.add_property("repulsor_accel", -15.0)
    repulsor_position = w.add_property("repulsor_position", REPULSOR_POS)
    update_repulsor = h.ConformToSphereModifier(
        origin=w.prop(repulsor_position).expr(), radius=w.lit(BALL_RADIUS).expr(), influence_dist=w.lit(BALL_RADIUS * 10.0).expr(),
        attraction_accel=w.prop(repulsor_accel).expr(), max_attraction_speed=w.lit(10.0).expr())
    attraction_accel = w.add_property("attraction_accel", 20.0)
    max_attraction_speed = w.add_property("max_attraction_speed", 5.0)
    sticky_factor = w.add_property("sticky_factor", 2.0)
    shell_half_thickness = w.add_property("shell_half_thickness", 0.1)
    update_attractor = h.ConformToSphereModifier(
        origin=w.lit(ATTRACTOR_POS).expr(), radius=w.lit(BALL_RADIUS * 6.0).expr(), influence_dist=w.lit(BALL_RADIUS * 100.0).expr(),
        attraction_accel=w.prop(attraction_accel).expr(), max_attraction_speed=w.prop(max_attraction_speed).expr(),
        shell_half_thickness=w.prop(shell_half_thickness).expr(), sticky_factor=w.prop(sticky_factor).expr())
    return (h.EffectAsset(capacity, spawner, w.finish())
            .with_name("force_field")
            .init(init_pos).init(init_vel).init(init_age).init(init_lifetime)
            .update(update_attractor).update(update_repulsor).update(allow_zone).update(deny_zone)
            .render(h.SizeOverLifetimeModifier())
            .render(h.ColorOverLifetimeModifier()))


def instancing(capacity=65536, rate=None):
    """First asset of examples/instancing.rs:222-249 (C4): sphere volume r=1, speed 2, lifetime 12."""
    w = h.ExprWriter()
    init_age = h.SetAttributeModifier(A.AGE, w.lit(0.0).expr())
    init_lifetime = h.SetAttributeModifier(A.LIFETIME, w.lit(12.0).expr())
    init_pos = h.SetPositionSphereModifier(w.lit((0.0, 0.0, 0.0)).expr(), w.lit(1.0).expr(), h.ShapeDimension.Volume)
    init_vel = h.SetVelocitySphereModifier(w.lit((0.0, 0.0, 0.0)).expr(), w.lit(2.0).expr())
    spawner = h.SpawnerSettings.rate(float(capacity) / 12.0 if rate is None else float(rate))
    return (h.EffectAsset(capacity, spawner, w.finish())
            .with_name("instancing")
            .init(init_pos).init(init_vel).init(init_age).init(init_lifetime)
            .render(h.ColorOverLifetimeModifier()))


RIBBON_LIFETIME = 1.5


def ribbon(capacity=1 << 22, rate=None):
    """examples/ribbon.rs:120-178 (C5): MotionIntegration::None, Global space, RIBBON_ID."""
    w = h.ExprWriter()
    mods = [
        h.SetAttributeModifier(A.POSITION, w.lit((0.0, 0.0, 0.0)).expr()),
        h.SetAttributeModifier(A.AGE, w.lit(0.0).expr()),
        h.SetAttributeModifier(A.LIFETIME, w.lit(RIBBON_LIFETIME).expr()),
        h.SetAttributeModifier(A.SIZE, w.lit(0.5).expr()),
        h.SetAttributeModifier(A.RIBBON_ID, w.lit(h.Value.u32(0)).expr()),
    ]
    spawner = h.SpawnerSettings.rate(float(capacity) / RIBBON_LIFETIME if rate is None else float(rate))
    asset = (h.EffectAsset(capacity, spawner, w.finish())
             .with_name("ribbon")
             .with_motion_integration(h.MotionIntegration.None_)
             .with_simulation_space(h.SimulationSpace.Global))
    for m in mods:
        asset = asset.init(m)
    return asset.render(h.SizeOverLifetimeModifier()).render(h.ColorOverLifetimeModifier())


# ---- the real examples/firework.rs: three linked effects (GPU spawn events) ------------------------------

def firework_rocket(capacity=32, trail_count=5, explosion_count=1000):
    """create_rocket_effect (examples/firework.rs:36-131): rate(1..3)/s rockets; every alive rocket emits
    `trail_count` sparkle events per frame on channel 0 and `explosion_count` events on channel 1 when it dies."""
    w = h.ExprWriter()
    init_pos = h.SetPositionCircleModifier(w.lit((0.0, 0.0, 0.0)).expr(), w.lit((0.0, 1.0, 0.0)).expr(), w.lit(30.0).expr(), h.ShapeDimension.Volume)
    zero = w.lit(0.0)
    y = w.lit(140.0).uniform(w.lit(160.0))
    init_vel = h.SetAttributeModifier(A.VELOCITY, zero.vec3(y, zero).expr())
    init_age = h.SetAttributeModifier(A.AGE, w.lit(0.0).expr())
    rgb = w.rand(h.VectorType.VEC3F) * w.lit(0.9) + w.lit(0.1)
    init_trails_color = h.SetAttributeModifier(A.U32_0, rgb.vec4_xyz_w(w.lit(1.0)).pack4x8unorm().expr())
    init_lifetime = h.SetAttributeModifier(A.LIFETIME, w.lit(0.8).uniform(w.lit(1.2)).expr())
    update_accel = h.AccelModifier(w.lit((-0.0, -16.0, -0.0)).expr())
    update_drag = h.LinearDragModifier(w.lit(4.0).expr())
    spawn_trail = h.EmitSpawnEventModifier(h.EventEmitCondition.Always, w.lit(h.Value.u32(trail_count)).expr(), 0)
    spawn_on_die = h.EmitSpawnEventModifier(h.EventEmitCondition.OnDie, w.lit(h.Value.u32(explosion_count)).expr(), 1)
    spawner = h.SpawnerSettings.rate(h.CpuValue.Uniform(1.0, 3.0))
    return (h.EffectAsset(capacity, spawner, w.finish()).with_name("rocket")
            .init(init_pos).init(init_vel).init(init_age).init(init_lifetime).init(init_trails_color)
            .update(update_drag).update(update_accel).update(spawn_trail).update(spawn_on_die)
            .render(h.ColorOverLifetimeModifier()).render(h.SizeOverLifetimeModifier()))


def firework_sparkle_trail(capacity=1000):
    """create_sparkle_trail_effect (examples/firework.rs:135-183): child of the rocket on channel 0."""
    w = h.ExprWriter()
    init_pos = h.InheritAttributeModifier(A.POSITION)
    vel = (w.rand(h.VectorType.VEC3F) * w.lit(2.0) - w.lit(1.0)).normalized() * w.lit(1.0)
    init_vel = h.SetAttributeModifier(A.VELOCITY, vel.expr())
    init_age = h.SetAttributeModifier(A.AGE, w.lit(0.0).expr())
    init_lifetime = h.SetAttributeModifier(A.LIFETIME, w.lit(0.2).expr())
    update_accel = h.AccelModifier(w.lit((-0.0, -16.0, -0.0)).expr())
    update_drag = h.LinearDragModifier(w.lit(4.0).expr())
    return (h.EffectAsset(capacity, h.SpawnerSettings(), w.finish()).with_name("sparkle_trail")
            .init(init_pos).init(init_vel).init(init_age).init(init_lifetime)
            .update(update_drag).update(update_accel)
            .render(h.ColorOverLifetimeModifier()).render(h.SizeOverLifetimeModifier()))


def firework_trails_child(capacity=10000):
    """create_trails_effect (examples/firework.rs:187-251) verbatim: child of the rocket on channel 1,
    position inherited from the exploding rocket, colour from the rocket's U32_0."""
    w = h.ExprWriter()
    init_pos = h.InheritAttributeModifier(A.POSITION)
    init_color = h.SetAttributeModifier(A.COLOR, w.parent_attr(A.U32_0).expr())
    center = w.attr(A.POSITION)
    speed = w.lit(40.0).uniform(w.lit(60.0))
    direction = w.rand(h.VectorType.VEC3F).mul(w.lit(2.0)).sub(w.lit(1.0)).normalized()
    init_vel = h.SetAttributeModifier(A.VELOCITY, (center + direction * speed).expr())
    init_age = h.SetAttributeModifier(A.AGE, w.lit(0.0).expr())
    init_lifetime = h.SetAttributeModifier(A.LIFETIME, w.lit(0.8).uniform(w.lit(1.2)).expr())
    update_accel = h.AccelModifier(w.lit((-0.0, -16.0, -0.0)).expr())
    update_drag = h.LinearDragModifier(w.lit(4.0).expr())
    return (h.EffectAsset(capacity, h.SpawnerSettings(), w.finish()).with_name("trail")
            .init(init_pos).init(init_vel).init(init_age).init(init_lifetime).init(init_color)
            .update(update_drag).update(update_accel)
            .render(h.ColorOverLifetimeModifier()).render(h.SizeOverLifetimeModifier()).render(h.OrientModifier(h.OrientMode.AlongVelocity)))
