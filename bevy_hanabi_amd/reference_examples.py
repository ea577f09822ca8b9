"""The simulation side of every effect in the reference's `examples/`, restated through the mirrored authoring API.

`effects.py` holds the five SURVEY.md §8(d) bench configurations (firework, force_field, instancing, ribbon and the
single-particle GPU test). This module is the rest of `examples/*.rs`: each function builds the asset(s) one example
registers — spawner, simulation space and condition, motion integration, init and update modifiers, properties, and
the render modifiers only as far as they add attributes to the particle layout (gradients, textures, meshes and
cameras are outside the simulation path) — and `CATALOG` adds what the example's own systems do to the effect every
frame (property updates, transforms, spawner resets, parent/child links), so that a test or a host application can play
an example frame by frame. Every entry cites the file:line it restates.

Literal typing follows the reference: `writer.lit(5.)` is an f32 (an AbstractFloat in the emitted WGSL), `writer.lit(-3)`
an i32 (an AbstractInt: instancing.rs:274 passes it where an f32 acceleration is expected), `writer.lit(4u32)` a u32.
"""
import math

from . import _hanabi_host as h

A = h.Attribute
F = h.ScalarType.Float
TAU = 6.283185307179586
ZERO3 = (0.0, 0.0, 0.0)
X3, Y3, Z3 = (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)


def u32(v):
    return h.Value.u32(v)


def _scaled(v, k):
    """`Vec3::Y * -8.` of glam: every component is multiplied, so zeros take the sign of the product."""
    return tuple(c * k for c in v)


def _age_lifetime(w, lifetime, age=0.0):
    return (h.SetAttributeModifier(A.AGE, w.lit(age).expr()), h.SetAttributeModifier(A.LIFETIME, w.lit(lifetime).expr()))


def _build(asset, init=(), update=(), render=()):
    for m in init:
        asset = asset.init(m)
    for m in update:
        asset = asset.update(m)
    for m in render:
        asset = asset.render(m)
    return asset


# ---- single-effect examples ----------------------------------------------------------------------------------------------
def example_2d():
    """examples/2d.rs:52-92: rate(30), circle surface r=0.05 around Z, radial speed 0.1, lifetime 5."""
    w = h.ExprWriter()
    age, life = _age_lifetime(w, 5.0)
    pos = h.SetPositionCircleModifier(w.lit(ZERO3).expr(), w.lit(Z3).expr(), w.lit(0.05).expr(), h.ShapeDimension.Surface)
    vel = h.SetVelocityCircleModifier(w.lit(ZERO3).expr(), w.lit(Z3).expr(), w.lit(0.1).expr())
    return _build(h.EffectAsset(4096, h.SpawnerSettings.rate(30.0), w.finish()).with_name("2d"),
                  init=[pos, vel, age, life], render=[h.SizeOverLifetimeModifier(), h.ColorOverLifetimeModifier(), h.RoundModifier()])


def example_activate():
    """examples/activate.rs:91-135: rate(30) that starts inactive (the ball toggles it), buoyancy, KillAabb."""
    w = h.ExprWriter()
    age, life = _age_lifetime(w, 5.0)
    pos = h.SetPositionSphereModifier(w.lit(ZERO3).expr(), w.lit(0.05).expr(), h.ShapeDimension.Surface)
    vel = h.SetVelocitySphereModifier(w.lit(ZERO3).expr(), w.lit(0.1).expr())
    buoyancy = h.AccelModifier(w.lit(_scaled(Y3, 0.2)).expr())
    allow_zone = h.KillAabbModifier(w.lit(_scaled(Y3, -2.02)).expr(), w.lit((2.0, 2.0, 2.0)).expr())
    spawner = h.SpawnerSettings.rate(30.0).with_starts_active(False)
    return _build(h.EffectAsset(32768, spawner, w.finish()).with_name("activate"),
                  init=[pos, vel, age, life], update=[buoyancy, allow_zone], render=[h.SetSizeModifier(), h.ColorOverLifetimeModifier(), h.RoundModifier()])


def example_billboard():
    """examples/billboard.rs:70-140: rate(64), circle volume r=1 around Y, random speed, rotation in F32_0, random colour."""
    w = h.ExprWriter()
    age, life = _age_lifetime(w, 5.0)
    pos = h.SetPositionCircleModifier(w.lit(_scaled(Y3, 0.1)).expr(), w.lit(Y3).expr(), w.lit(1.0).expr(), h.ShapeDimension.Volume)
    vel = h.SetVelocityCircleModifier(w.lit(ZERO3).expr(), w.lit(Y3).expr(), (w.lit(0.5) + w.lit(0.2) * w.rand(F)).expr())
    color = h.SetAttributeModifier(A.COLOR, w.rand(h.VectorType.VEC4F).pack4x8unorm().expr())
    rotation = h.SetAttributeModifier(A.F32_0, (w.rand(F) * w.lit(TAU)).expr())
    return _build(h.EffectAsset(32768, h.SpawnerSettings.rate(64.0), w.finish()).with_name("billboard"),
                  init=[pos, vel, age, life, rotation, color],
                  render=[h.ParticleTextureModifier(), h.OrientModifier(h.OrientMode.FaceCameraPosition), h.SizeOverLifetimeModifier()])


CIRCLE_FRAME_COUNT = 64  # sprite_grid_size 8 x 8 (circle.rs:44, 53)


def example_circle():
    """examples/circle.rs:56-145: burst(32, 8 s), random initial age, animated SPRITE_INDEX computed in the update pass."""
    w = h.ExprWriter()
    age = h.SetAttributeModifier(A.AGE, w.rand(F).expr())
    life = h.SetAttributeModifier(A.LIFETIME, w.lit(5.0).expr())
    pos = h.SetPositionCircleModifier(w.lit(_scaled(Y3, 0.1)).expr(), w.lit(Y3).expr(), w.lit(0.4).expr(), h.ShapeDimension.Surface)
    vel = h.SetVelocityCircleModifier(w.lit(ZERO3).expr(), w.lit(Y3).expr(), (w.lit(1.0) + w.lit(0.5) * w.rand(F)).expr())
    sprite_index = (w.attr(A.AGE).mul(w.lit(0.1)).fract().mul(w.lit(2.0)).sub(w.lit(1.0)).abs()
                    .mul(w.lit(float(CIRCLE_FRAME_COUNT))).cast(h.ScalarType.Int).rem(w.lit(CIRCLE_FRAME_COUNT)))
    update_sprite = h.SetAttributeModifier(A.SPRITE_INDEX, sprite_index.expr())
    return _build(h.EffectAsset(32768, h.SpawnerSettings.burst(32.0, 8.0), w.finish()).with_name("circle"),
                  init=[pos, vel, age, life], update=[update_sprite],
                  render=[h.ParticleTextureModifier(), h.ParticleTextureModifier(), h.FlipbookModifier(), h.ColorOverLifetimeModifier(),
                          h.SizeOverLifetimeModifier()])


def example_expr():
    """examples/expr.rs:52-100 ("whirlwind"): the acceleration is an expression of the position and of time."""
    w = h.ExprWriter()
    age = h.SetAttributeModifier(A.AGE, w.lit(0.0).expr())
    life = h.SetAttributeModifier(A.LIFETIME, w.lit(2.5).uniform(w.lit(3.5)).expr())
    radial = (w.attr(A.POSITION) - w.lit(ZERO3)).normalized()
    vertical = w.lit(_scaled(Y3, 4.0))
    anim = w.time().sin() * w.lit(6.0) - w.lit(6.0)
    accel = h.AccelModifier((radial * anim + vertical).expr())
    pos = h.SetPositionCircleModifier(w.lit(ZERO3).expr(), w.lit(Y3).expr(), w.lit(4.0).expr(), h.ShapeDimension.Surface)
    vel = h.SetVelocityTangentModifier(w.lit(ZERO3).expr(), w.lit(Y3).expr(), w.lit(3.0).expr())
    return _build(h.EffectAsset(32768, h.SpawnerSettings.rate(500.0), w.finish()).with_name("whirlwind"),
                  init=[pos, age, life, vel], update=[accel],
                  render=[h.ColorOverLifetimeModifier(), h.SizeOverLifetimeModifier(), h.OrientModifier(h.OrientMode.AlongVelocity)])


def example_gradient():
    """examples/gradient.rs:54-92: rate(1000), sphere volume r=1, speed 2, lifetime 5."""
    w = h.ExprWriter()
    age, life = _age_lifetime(w, 5.0)
    pos = h.SetPositionSphereModifier(w.lit(ZERO3).expr(), w.lit(1.0).expr(), h.ShapeDimension.Volume)
    vel = h.SetVelocitySphereModifier(w.lit(ZERO3).expr(), w.lit(2.0).expr())
    return _build(h.EffectAsset(32768, h.SpawnerSettings.rate(1000.0), w.finish()).with_name("gradient"),
                  init=[pos, vel, age, life], render=[h.ParticleTextureModifier(), h.ColorOverLifetimeModifier()])


def example_init(shape):
    """examples/init.rs:35-58, 112-165: once(500) particles that never move (MotionIntegration::None, Local space); `shape` is
    'circle' (axis Z, r 5, Volume), 'sphere' (r 5, Volume) or 'cone' (height 10, base radius 1, top radius 4, Volume)."""
    w = h.ExprWriter()
    if shape == "circle":
        init = h.SetPositionCircleModifier(w.lit(ZERO3).expr(), w.lit(Z3).expr(), w.lit(5.0).expr(), h.ShapeDimension.Volume)
    elif shape == "sphere":
        init = h.SetPositionSphereModifier(w.lit(ZERO3).expr(), w.lit(5.0).expr(), h.ShapeDimension.Volume)
    elif shape == "cone":
        init = h.SetPositionCone3dModifier(w.lit(10.0).expr(), w.lit(1.0).expr(), w.lit(4.0).expr(), h.ShapeDimension.Volume)
    else:
        raise ValueError(shape)
    asset = (h.EffectAsset(32768, h.SpawnerSettings.once(500.0), w.finish()).with_name("SetPosition" + shape.capitalize() + "Modifier")
             .with_motion_integration(h.MotionIntegration.None_).with_simulation_space(h.SimulationSpace.Local))
    return _build(asset, init=[init], render=[h.OrientModifier(h.OrientMode.FaceCameraPosition), h.SetColorModifier(), h.SetSizeModifier()])


def example_instancing_alternate():
    """Second asset of examples/instancing.rs:255-290: Local space, AGE only through ColorOverLifetimeModifier's layout requirement
    (never initialised: the attribute default 0), tangent velocity, RadialAccelModifier whose acceleration is the i32 literal `writer.lit(-3)` (line 274)."""
    w = h.ExprWriter()
    life = h.SetAttributeModifier(A.LIFETIME, w.lit(5.0).expr())
    pos = h.SetPositionSphereModifier(w.lit(ZERO3).expr(), w.lit(7.0).expr(), h.ShapeDimension.Volume)
    vel = h.SetVelocityTangentModifier(w.lit(ZERO3).expr(), w.lit(Z3).expr(), w.lit(4.0).expr())
    radial = h.RadialAccelModifier(w.lit(ZERO3).expr(), w.lit(-3).expr())
    asset = (h.EffectAsset(512, h.SpawnerSettings.rate(102.0), w.finish()).with_simulation_space(h.SimulationSpace.Local)
             .with_name("alternate instancing"))
    return _build(asset, init=[pos, vel, life], update=[radial], render=[h.ParticleTextureModifier(), h.ColorOverLifetimeModifier()])


def example_lifetime(lifetime):
    """examples/lifetime.rs:57-195: three effects differing only by LIFETIME (12, 3, 0.75 s) against a 3 s burst(50) period."""
    w = h.ExprWriter()
    age, life = _age_lifetime(w, lifetime)
    pos = h.SetPositionSphereModifier(w.lit(ZERO3).expr(), w.lit(5.0).expr(), h.ShapeDimension.Volume)
    vel = h.SetVelocitySphereModifier(w.lit(ZERO3).expr(), w.lit(2.0).expr())
    return _build(h.EffectAsset(512, h.SpawnerSettings.burst(50.0, 3.0), w.finish()).with_name("emit:burst"),
                  init=[pos, vel, age, life], render=[h.ColorOverLifetimeModifier()])


def example_multicam():
    """examples/multicam.rs:44-90: rate(5), sphere surface r=2, speed 6, gravity -3."""
    w = h.ExprWriter()
    age, life = _age_lifetime(w, 5.0)
    accel = h.AccelModifier(w.lit(_scaled(Y3, -3.0)).expr())
    pos = h.SetPositionSphereModifier(w.lit(ZERO3).expr(), w.lit(2.0).expr(), h.ShapeDimension.Surface)
    vel = h.SetVelocitySphereModifier(w.lit(ZERO3).expr(), w.lit(6.0).expr())
    return _build(h.EffectAsset(32768, h.SpawnerSettings.rate(5.0), w.finish()).with_name("effect"),
                  init=[pos, vel, age, life], update=[accel], render=[h.ColorOverLifetimeModifier(), h.SizeOverLifetimeModifier(), h.OrientModifier(h.OrientMode.FaceCameraPosition)])


def example_ordering():
    """examples/ordering.rs:47-93 ("firework"): rate(128), random age and lifetime, drag 5 then gravity -8."""
    w = h.ExprWriter()
    age = h.SetAttributeModifier(A.AGE, w.lit(0.0).uniform(w.lit(0.2)).expr())
    life = h.SetAttributeModifier(A.LIFETIME, w.lit(2.0).uniform(w.lit(3.0)).expr())
    accel = h.AccelModifier(w.lit(_scaled(Y3, -8.0)).expr())
    drag = h.LinearDragModifier(w.lit(5.0).expr())
    pos = h.SetPositionSphereModifier(w.lit(ZERO3).expr(), w.lit(2.0).expr(), h.ShapeDimension.Volume)
    vel = h.SetVelocitySphereModifier(w.lit(ZERO3).expr(), (w.rand(F) * w.lit(20.0) + w.lit(60.0)).expr())
    return _build(h.EffectAsset(2048, h.SpawnerSettings.rate(128.0), w.finish()).with_name("firework"),
                  init=[pos, vel, age, life], update=[drag, accel], render=[h.ColorOverLifetimeModifier(), h.SizeOverLifetimeModifier()])


def example_portal():
    """examples/portal.rs:56-96: rate(5000) on a circle r=4, no initial velocity, drag 2 and a tangential acceleration of 30
    (`TangentAccelModifier::constant(&mut module, Vec3::ZERO, Vec3::Z, 30.)`, literals added to the finished module)."""
    w = h.ExprWriter()
    pos = h.SetPositionCircleModifier(w.lit(ZERO3).expr(), w.lit(Z3).expr(), w.lit(4.0).expr(), h.ShapeDimension.Surface)
    age = h.SetAttributeModifier(A.AGE, w.lit(0.0).expr())
    life = h.SetAttributeModifier(A.LIFETIME, w.lit(0.6).uniform(w.lit(1.3)).expr())
    drag = h.LinearDragModifier(w.lit(2.0).expr())
    module = w.finish()
    tangent = h.TangentAccelModifier_constant(module, ZERO3, Z3, 30.0)
    return _build(h.EffectAsset(16384, h.SpawnerSettings.rate(5000.0), module).with_name("portal"),
                  init=[pos, age, life], update=[drag, tangent],
                  render=[h.ColorOverLifetimeModifier(), h.SizeOverLifetimeModifier(), h.OrientModifier(h.OrientMode.AlongVelocity)])


def example_puffs():
    """examples/puffs.rs:114-190 ("cartoon explosion"): burst(16, 0.45 s); POSITION is set twice (circle, then jiggled in Y by a
    vec3 draw), the size lives in F32_0 and SIZE is recomputed from AGE in the update pass."""
    w = h.ExprWriter()
    xz = h.SetPositionCircleModifier(w.lit(ZERO3).expr(), w.lit(Z3).expr(), w.lit(1.0).expr(), h.ShapeDimension.Volume)
    y = h.SetAttributeModifier(A.POSITION, w.attr(A.POSITION).add(w.rand(h.VectorType.VEC3F) * w.lit((0.0, 1.0, 0.0))).expr())
    age, life = _age_lifetime(w, 3.0)
    size0 = h.SetAttributeModifier(A.F32_0, (w.rand(F) * w.lit(2.0) + w.lit(0.5)).expr())
    velocity = h.SetAttributeModifier(A.VELOCITY, w.lit((0.0, 0.0, -20.0)).expr())
    size = h.SetAttributeModifier(A.SIZE, w.attr(A.F32_0).mul(w.lit(1.0).sub(w.attr(A.AGE).mul(w.lit(0.75))).max(w.lit(0.0))).expr())
    return _build(h.EffectAsset(256, h.SpawnerSettings.burst(16.0, 0.45), w.finish()).with_name("cartoon explosion"),
                  init=[xz, y, age, life, size0, velocity], update=[size])


def example_random():
    """examples/random.rs:51-86: burst with a random count in [1, 100] and a random period in [1, 4] s (CpuValue::Uniform,
    sampled by the spawner on the host), gravity +5."""
    w = h.ExprWriter()
    age, life = _age_lifetime(w, 5.0)
    accel = h.AccelModifier(w.lit(_scaled(Y3, 5.0)).expr())
    pos = h.SetPositionSphereModifier(w.lit(ZERO3).expr(), w.lit(5.0).expr(), h.ShapeDimension.Volume)
    vel = h.SetVelocitySphereModifier(w.lit(ZERO3).expr(), w.lit(2.0).expr())
    spawner = h.SpawnerSettings.burst(h.CpuValue.Uniform(1.0, 100.0), h.CpuValue.Uniform(1.0, 4.0))
    return _build(h.EffectAsset(32768, spawner, w.finish()).with_name("emit:burst"),
                  init=[pos, vel, age, life], update=[accel], render=[h.ColorOverLifetimeModifier()])


def example_spawn(which):
    """examples/spawn.rs:91-267: 'rate' (cone, property-driven acceleration, rotated and translated emitter), 'once' (1000
    particles, no update modifier) and 'burst' (400 every 3 s, random SIZE, property-driven acceleration)."""
    w = h.ExprWriter()
    age, life = _age_lifetime(w, 5.0)
    if which == "rate":
        my_accel = w.add_property("my_accel", _scaled(Y3, 3.0))
        accel = h.AccelModifier(w.prop(my_accel).expr())
        pos = h.SetPositionCone3dModifier(w.lit(20.0).expr(), w.lit(0.0).expr(), w.lit(10.0).expr(), h.ShapeDimension.Volume)
        vel = h.SetVelocitySphereModifier(w.lit(ZERO3).expr(), w.lit(10.0).expr())
        return _build(h.EffectAsset(32768, h.SpawnerSettings.rate(500.0), w.finish()).with_name("emit:rate"),
                      init=[pos, vel, age, life], update=[accel], render=[h.ColorOverLifetimeModifier(), h.SizeOverLifetimeModifier()])
    pos = h.SetPositionSphereModifier(w.lit(ZERO3).expr(), w.lit(5.0).expr(), h.ShapeDimension.Volume)
    vel = h.SetVelocitySphereModifier(w.lit(ZERO3).expr(), w.lit(2.0).expr())
    if which == "once":
        return _build(h.EffectAsset(32768, h.SpawnerSettings.once(1000.0), w.finish()).with_name("emit:once"),
                      init=[pos, vel, age, life], render=[h.ColorOverLifetimeModifier()])
    if which != "burst":
        raise ValueError(which)
    size = h.SetAttributeModifier(A.SIZE, (w.rand(F) * w.lit(0.4) + w.lit(0.3)).expr())
    my_accel = w.add_property("my_accel", (0.0, -3.0, 0.0))
    accel = h.AccelModifier(w.prop(my_accel).expr())
    return _build(h.EffectAsset(32768, h.SpawnerSettings.burst(400.0, 3.0), w.finish()).with_name("emit:burst"),
                  init=[pos, vel, age, life, size], update=[accel], render=[h.ColorOverLifetimeModifier()])


SPAWN_ON_COMMAND_BALL_RADIUS = 0.05


def example_spawn_on_command():
    """examples/spawn_on_command.rs:81-127: once(100) that does not emit on start; the host resets the spawner at every
    collision after setting the `spawn_color` and `normal` properties the init expressions read."""
    spawner = h.SpawnerSettings.once(100.0).with_emit_on_start(False)
    w = h.ExprWriter()
    age, life = _age_lifetime(w, 1.5)
    drag = h.LinearDragModifier(w.lit(2.0).expr())
    spawn_color = w.add_property("spawn_color", u32(0xFFFFFFFF))
    color = h.SetAttributeModifier(A.COLOR, w.prop(spawn_color).expr())
    normal = w.prop(w.add_property("normal", ZERO3))
    pos = h.SetAttributeModifier(A.POSITION, (normal * w.lit(-SPAWN_ON_COMMAND_BALL_RADIUS) + w.lit(_scaled(Z3, 0.2))).expr())
    tangent = w.lit(Z3).cross(normal)
    spread = w.rand(F) * w.lit(2.0) - w.lit(1.0)
    speed = w.rand(F) * w.lit(0.2)
    vel = h.SetAttributeModifier(A.VELOCITY, ((normal + tangent * spread * w.lit(5.0)).normalized() * speed).expr())
    return _build(h.EffectAsset(32768, spawner, w.finish()).with_name("spawn_on_command"),
                  init=[pos, vel, age, life, color], update=[drag], render=[h.SetSizeModifier(), h.ScreenSpaceSizeModifier()])


def example_visibility(condition):
    """examples/visibility.rs:65-90: the same asset twice, SimulationCondition::WhenVisible and ::Always; the example toggles
    the entities' visibility every 1.5 s."""
    w = h.ExprWriter()
    velocity = h.SetAttributeModifier(A.VELOCITY, w.lit(_scaled(X3, 3.0)).expr())
    age, life = _age_lifetime(w, 15.0)
    pos = h.SetPositionSphereModifier(w.lit(ZERO3).expr(), w.lit(5.0).expr(), h.ShapeDimension.Volume)
    asset = h.EffectAsset(4096, h.SpawnerSettings.burst(50.0, 15.0), w.finish()).with_simulation_condition(condition)
    return _build(asset, init=[pos, velocity, age, life], render=[h.ColorOverLifetimeModifier()])


# ---- lightning.rs: a ribbon whose shape is integer hashing of PARTICLE_COUNTER --------------------------------------------
LIGHTNING_PARTICLES_PER_BOLT = 40
LIGHTNING_BOLT_LIFETIME = 0.3
LIGHTNING_BOLT_LENGTH = 30.0
LIGHTNING_MAX_SPREAD = 1.5
LIGHTNING_BURST_INTERVAL = 1.5
LIGHTNING_GROUND_Y = -15.0


def example_lightning_bolt():
    """create_bolt_effect (examples/lightning.rs:86-205): once(40) per strike, MotionIntegration::None, every attribute a pure
    function of PARTICLE_COUNTER and of the `wave_seed` property (u32 hashing, casts, step / mix / max)."""
    w = h.ExprWriter()
    U, Fl = h.ScalarType.Uint, h.ScalarType.Float
    wave_seed = w.add_property("wave_seed", 0.0)
    wave_seed_int = ((w.prop(wave_seed) + w.lit(100.0)) * w.lit(1000.0)).cast(U)
    particle_index = w.attr(A.PARTICLE_COUNTER) % w.lit(u32(LIGHTNING_PARTICLES_PER_BOLT))
    cell_size = w.lit(u32(4))
    cell_id = particle_index / cell_size

    def control_point(ident):
        def hash_(mult, modulus):
            return (ident * w.lit(u32(mult)) + wave_seed_int * w.lit(u32(67891))) % w.lit(u32(modulus))
        jitter = hash_(12345, 10111) % w.lit(u32(3))
        x_rnd = hash_(54321, 10111).cast(Fl) / w.lit(5055.5) - w.lit(1.0)
        z_rnd = hash_(98765, 10111).cast(Fl) / w.lit(5055.5) - w.lit(1.0)
        return jitter, x_rnd, z_rnd

    id0 = cell_id.max(w.lit(u32(1))) - w.lit(u32(1))
    id1 = cell_id
    id2 = cell_id + w.lit(u32(1))
    j0, xr0, zr0 = control_point(id0)
    j1, xr1, zr1 = control_point(id1)
    j2, xr2, zr2 = control_point(id2)
    p0 = id0 * cell_size + j0
    p1 = id1 * cell_size + j1
    p2 = id2 * cell_size + j2
    is_after_p1 = p1.cast(Fl).step(particle_index.cast(Fl))
    start_p = p0.cast(Fl).mix(p1.cast(Fl), is_after_p1)
    end_p = p1.cast(Fl).mix(p2.cast(Fl), is_after_p1)
    start_xr = xr0.mix(xr1, is_after_p1)
    end_xr = xr1.mix(xr2, is_after_p1)
    start_zr = zr0.mix(zr1, is_after_p1)
    end_zr = zr1.mix(zr2, is_after_p1)
    dist = (end_p - start_p).max(w.lit(1.0))
    progress = (particle_index.cast(Fl) - start_p) / dist
    x_jitter = start_xr.mix(end_xr, progress) * w.lit(LIGHTNING_MAX_SPREAD)
    z_jitter = start_zr.mix(end_zr, progress) * w.lit(LIGHTNING_MAX_SPREAD * 0.5)
    total_progress = particle_index.cast(Fl) / w.lit(float(LIGHTNING_PARTICLES_PER_BOLT - 1))
    y_top = w.lit(LIGHTNING_GROUND_Y + LIGHTNING_BOLT_LENGTH)
    y_pos = y_top - total_progress * w.lit(LIGHTNING_BOLT_LENGTH)
    weight = w.lit(4.0) * total_progress * (w.lit(1.0) - total_progress)
    position = (x_jitter * weight).vec3(y_pos, z_jitter * weight)
    init_age = particle_index.cast(Fl) * w.lit(0.0001)
    init_ribbon_id = w.attr(A.PARTICLE_COUNTER) / w.lit(u32(LIGHTNING_PARTICLES_PER_BOLT))
    init_size = w.lit(0.08) * (weight + w.lit(0.1))
    mods = [h.SetAttributeModifier(A.POSITION, position.expr()),
            h.SetAttributeModifier(A.AGE, init_age.expr()),
            h.SetAttributeModifier(A.LIFETIME, w.lit(LIGHTNING_BOLT_LIFETIME).expr()),
            h.SetAttributeModifier(A.RIBBON_ID, init_ribbon_id.expr()),
            h.SetAttributeModifier(A.SIZE, init_size.expr())]
    asset = (h.EffectAsset(1024, h.SpawnerSettings.once(float(LIGHTNING_PARTICLES_PER_BOLT)), w.finish()).with_name("lightning_bolt")
             .with_motion_integration(h.MotionIntegration.None_))
    return _build(asset, init=mods, render=[h.ColorOverLifetimeModifier()])


def example_lightning_impact():
    """create_impact_effect (examples/lightning.rs:207-248): once(80) sparks, velocity from cos/sin of a random angle, drag 2,
    gravity -40. No AGE initialisation (the attribute defaults to 0)."""
    w = h.ExprWriter()
    angle = w.rand(F) * w.lit(TAU)
    speed = w.lit(15.0) + w.rand(F) * w.lit(25.0)
    velocity = (angle.cos() * speed).vec3(w.lit(5.0) + w.rand(F) * w.lit(15.0), angle.sin() * speed)
    mods = [h.SetAttributeModifier(A.POSITION, w.lit(ZERO3).expr()),
            h.SetAttributeModifier(A.VELOCITY, velocity.expr()),
            h.SetAttributeModifier(A.LIFETIME, (w.lit(0.3) + w.rand(F) * w.lit(0.4)).expr()),
            h.SetAttributeModifier(A.SIZE, (w.lit(0.1) + w.rand(F) * w.lit(0.15)).expr())]
    drag = h.LinearDragModifier(w.lit(2.0).expr())
    gravity = h.AccelModifier(w.lit((0.0, -40.0, 0.0)).expr())
    return _build(h.EffectAsset(512, h.SpawnerSettings.once(80.0), w.finish()).with_name("impact_burst"),
                  init=mods, update=[drag, gravity], render=[h.ColorOverLifetimeModifier(), h.SizeOverLifetimeModifier()])


# ---- worms.rs: parent heads emitting GPU spawn events, child bodies as ribbons -------------------------------------------
def example_worms_head():
    """create_head_effect (examples/worms.rs:34-116): rate(2) heads steered by sin() of time, 5 spawn events per head and frame
    on channel 0; U32_0 carries PARTICLE_COUNTER as the ribbon id the bodies inherit."""
    w = h.ExprWriter()
    pos = h.SetAttributeModifier(A.POSITION, ((w.rand(h.VectorType.VEC3F) + w.lit((-0.5, -0.5, 0.0))) * w.lit((16.0, 16.0, 0.0))).expr())
    angle = h.SetAttributeModifier(A.F32_0, w.lit(0.0).normal(w.lit(1.0)).expr())
    color = h.SetAttributeModifier(A.COLOR, (w.rand(h.VectorType.VEC4F) * w.lit((1.0, 1.0, 1.0, 0.0)) + w.lit((0.0, 0.0, 0.0, 1.0))).pack4x8unorm().expr())
    age, life = _age_lifetime(w, 3.0)
    ribbon_id = h.SetAttributeModifier(A.U32_0, w.attr(A.PARTICLE_COUNTER).expr())
    steer = (w.lit((1.0, 1.0, 0.0)) * (w.attr(A.F32_0) + (w.time() * w.lit(5.0)).sin() * w.lit(1.0)) + w.lit((0.0, math.pi / 2.0, 0.0))).sin().mul(w.lit(5.0))
    velocity = h.SetAttributeModifier(A.VELOCITY, steer.expr())
    spawn_trail = h.EmitSpawnEventModifier(h.EventEmitCondition.Always, w.lit(u32(5)).expr(), 0)
    return _build(h.EffectAsset(100, h.SpawnerSettings.rate(2.0), w.finish()).with_name("worms_heads"),
                  init=[pos, angle, age, life, color, ribbon_id], update=[velocity, spawn_trail],
                  render=[h.SetSizeModifier(), h.ParticleTextureModifier()])


def example_worms_body():
    """create_body_effect (examples/worms.rs:118-160): child of the heads; position and colour inherited, RIBBON_ID from the
    parent's U32_0, MotionIntegration::None. Its own spawner (rate 0.5) is ignored for a child effect."""
    w = h.ExprWriter()
    mods = [h.InheritAttributeModifier(A.POSITION),
            h.SetAttributeModifier(A.RIBBON_ID, w.parent_attr(A.U32_0).expr()),
            h.SetAttributeModifier(A.AGE, w.lit(0.0).expr()),
            h.SetAttributeModifier(A.LIFETIME, w.lit(1.5).expr()),
            h.SetAttributeModifier(A.COLOR, w.parent_attr(A.COLOR).expr())]
    asset = (h.EffectAsset(5000, h.SpawnerSettings.rate(0.5), w.finish()).with_name("worms_bodies")
             .with_motion_integration(h.MotionIntegration.None_))
    return _build(asset, init=mods, render=[h.SetSizeModifier()])


# ---- what each example's systems do per frame ---------------------------------------------------------------------------
def rotation_z_translation(angle, t):
    """Row-major 3x4 of Transform::from_translation(t).with_rotation(Quat::from_rotation_z(angle))."""
    c, s = math.cos(angle), math.sin(angle)
    return [c, -s, 0.0, t[0], s, c, 0.0, t[1], 0.0, 0.0, 1.0, t[2]]


def rotation_y_translation(angle, t):
    c, s = math.cos(angle), math.sin(angle)
    return [c, 0.0, s, t[0], 0.0, 1.0, 0.0, t[1], -s, 0.0, c, t[2]]


def translation(t):
    return [1.0, 0.0, 0.0, t[0], 0.0, 1.0, 0.0, t[1], 0.0, 0.0, 1.0, t[2]]


class ExampleEffect:
    """One ParticleEffect entity of an example: `asset`, the entity's `transform`, its EffectParent (`parent` = index into
    the example's list, `channel` = index among the parent's children) and `drive(frame, time, spawner)`, which
    returns the properties the example's systems set that frame and may reset / (de)activate the host-side spawner."""

    def __init__(self, asset, transform=None, parent=None, channel=0, drive=None, visible=None):
        # `transform`: None (identity), a row-major 3x4 list, or a function of (frame, time) returning one
        self.asset, self.transform, self.parent, self.channel = asset, transform, parent, channel
        self.drive = drive or (lambda frame, time, spawner: {})
        self.visible = visible or (lambda frame, time: True)


def _spawn_rate_drive(frame, time, spawner):
    # update_accel (spawn.rs:269-277): my_accel = 10 * (cos, sin)(0.8 t)
    return {"my_accel": (math.cos(time * 0.8) * 10.0, math.sin(time * 0.8) * 10.0, 0.0)}


def _spawn_burst_drive(frame, time, spawner):
    # update_accel (spawn.rs:279-284)
    return {"my_accel": (math.cos(time * 0.3) * 10.0, math.sin(time * 0.3) * 10.0, 0.0)}


def _spawn_on_command_drive(frame, time, spawner):
    # update (spawn_on_command.rs:130-188): on a wall hit the properties are set and the spawner reset; here a hit
    # every 23 frames, cycling through the four walls, with a colour derived from the frame number
    if frame % 23 != 5:
        return {}
    nx, ny = [(1.0, 0.0), (0.0, -1.0), (-1.0, 0.0), (0.0, 1.0)][(frame // 23) % 4]
    spawner.reset()
    return {"spawn_color": 0xFF000000 | ((frame * 2654435761) & 0xFFFFFF), "normal": (nx, ny, 0.0)}


def _activate_drive():
    # update (activate.rs:161-185): a ball oscillating around y = 0 (acceleration 1 towards it, starting at y = 0 with
    # velocity 1) switches the spawner on while it is below; the effect is a child of the ball
    ball = {"y": 0.0, "vy": 1.0}

    def drive(frame, time, spawner, dt=1.0 / 60.0):
        ball["vy"] += (-1.0 if ball["y"] >= 0.0 else 1.0) * dt
        ball["y"] += ball["vy"] * dt
        spawner.active = ball["y"] < 0.0
        return {}

    def transform(frame, time):
        return translation((0.0, ball["y"], 0.0))

    return drive, transform


def _lightning_strikes(frame, dt=1.0 / 60.0):
    """update_lightning_timers (lightning.rs:257-280): a repeating timer of BURST_INTERVAL that starts 0.1 s before its end."""
    before = (LIGHTNING_BURST_INTERVAL - 0.1 + frame * dt) // LIGHTNING_BURST_INTERVAL
    after = (LIGHTNING_BURST_INTERVAL - 0.1 + (frame + 1) * dt) // LIGHTNING_BURST_INTERVAL
    return int(after) if after > before else 0


def _lightning_drive(frame, time, spawner):
    # a strike sets a new `wave_seed` in [-50, 50) (rand::random there; a fixed sequence here) and resets both spawners
    n = _lightning_strikes(frame)
    if not n:
        return {}
    spawner.reset()
    return {"wave_seed": float((n * 37) % 100) - 50.0}


def _lightning_impact_drive(frame, time, spawner):
    if _lightning_strikes(frame):
        spawner.reset()
    return {}


def _visibility_toggle(frame, time):
    # update (visibility.rs:135-150): toggled every 1.5 s, starting visible
    return int(time / 1.5) % 2 == 0


def catalog():
    """name -> list of ExampleEffect, one entry per example file (the five bench configurations live in effects.py)."""
    V = h.SimulationCondition
    _activate = _activate_drive()

    def spinning(t, speed=3.0):
        # rotate_effect (init.rs:162-167): the effect entity turns about Y under a translated parent
        return lambda frame, time: rotation_y_translation(time * 0.1 * speed * math.pi, t)

    return {
        "2d": [ExampleEffect(example_2d())],
        "activate": [ExampleEffect(example_activate(), transform=_activate[1], drive=_activate[0])],
        "billboard": [ExampleEffect(example_billboard())],
        "circle": [ExampleEffect(example_circle())],
        "expr": [ExampleEffect(example_expr())],
        "gradient": [ExampleEffect(example_gradient())],
        "init": [ExampleEffect(example_init("circle"), spinning((-20.0, 0.0, 0.0))), ExampleEffect(example_init("sphere"), spinning(ZERO3)),
                 ExampleEffect(example_init("cone"), spinning((20.0, 0.0, 0.0)))],
        "instancing_alternate": [ExampleEffect(example_instancing_alternate(), translation((3.0, -2.0, 0.0)))],
        "lifetime": [ExampleEffect(example_lifetime(12.0), translation((-50.0, 0.0, 0.0))), ExampleEffect(example_lifetime(3.0)),
                     ExampleEffect(example_lifetime(0.75), translation((50.0, 0.0, 0.0)))],
        "lightning": [ExampleEffect(example_lightning_bolt(), drive=_lightning_drive),
                      ExampleEffect(example_lightning_impact(), translation((0.0, LIGHTNING_GROUND_Y, 0.0)), drive=_lightning_impact_drive)],
        "multicam": [ExampleEffect(example_multicam())],
        "ordering": [ExampleEffect(example_ordering())],
        "portal": [ExampleEffect(example_portal())],
        "puffs": [ExampleEffect(example_puffs())],
        "random": [ExampleEffect(example_random())],
        "spawn": [ExampleEffect(example_spawn("rate"), rotation_z_translation(1.0, (-30.0, 0.0, 0.0)), drive=_spawn_rate_drive),
                  ExampleEffect(example_spawn("once")),
                  ExampleEffect(example_spawn("burst"), translation((30.0, 0.0, 0.0)), drive=_spawn_burst_drive)],
        "spawn_on_command": [ExampleEffect(example_spawn_on_command(), drive=_spawn_on_command_drive)],
        "visibility": [ExampleEffect(example_visibility(V.WhenVisible), translation((-30.0, -20.0, 0.0)), visible=_visibility_toggle),
                       ExampleEffect(example_visibility(V.Always), translation((-30.0, 20.0, 0.0)), visible=_visibility_toggle)],
        "worms": [ExampleEffect(example_worms_head()), ExampleEffect(example_worms_body(), parent=0, channel=0)],
    }
