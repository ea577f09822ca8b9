"""The reference's on-disk asset format: EffectAsset::serialize / deserialize (src/asset.rs:674-716).

The reference ships no `.effect` sample and no golden string; what pins the format is its source (serde derives, the
`EffectAsset` field list, ExprHandle "#<id>", Attribute by name, modifiers as { "type path": ( fields ) } — see
bevy_hanabi_amd/csrc/host/ron.cpp). The tests mirror the reference's own: the round trips of asset.rs:1303-1365 and
registry.rs:308-431, the attribute / handle forms of attributes.rs:2308-2321 and expr.rs:4831-4885, the deserializer's
missing-field / duplicate-field / unknown-modifier errors (asset.rs:812-948, registry.rs:158-165); plus: a hand-written file in
free RON style (comments, struct names, trailing-comma-free, exponents) loads; every asset of the suite round-trips to the
same program blob; and (GPU) a loaded file simulates what the oracle computes for the original asset."""
import ctypes as C

import numpy as np
import pytest

import bevy_hanabi_amd as bh
from bevy_hanabi_amd import build as hb
from bevy_hanabi_amd import effects

A = bh.Attribute


def reference_serde_asset():
    """The asset of the reference's `serde_asset` test (src/asset.rs:1305-1333)."""
    w = bh.ExprWriter()
    pos = w.lit((1.2, -3.45, 87.54485))
    x = w.lit(bh.Value.vec_b([False, True]))
    _ = x + pos
    mod_pos = bh.SetAttributeModifier(A.POSITION, pos.expr())
    module = w.finish()
    prop = module.add_property("my_prop", (1.2, -2.3, 55.32))
    p = module.prop(prop)
    module.abs(p)
    asset = bh.EffectAsset(4096, bh.SpawnerSettings.rate(30.0), module)
    asset.name = "Effect"
    asset.z_layer_2d = 1.5
    asset.simulation_space = bh.SimulationSpace.Local
    asset.simulation_condition = bh.SimulationCondition.Always
    asset.prng_seed = 4284
    asset.motion_integration = bh.MotionIntegration.PreUpdate
    return asset.init(mod_pos)


def test_round_trip_of_the_reference_serde_asset():
    effect = reference_serde_asset()
    s = bh.to_ron(effect)
    back = bh.from_ron(s)
    assert (back.name, back.capacity, back.z_layer_2d, back.prng_seed) == ("Effect", 4096, 1.5, 4284)
    assert back.simulation_space == bh.SimulationSpace.Local and back.simulation_condition == bh.SimulationCondition.Always
    assert back.motion_integration == bh.MotionIntegration.PreUpdate
    assert bh.serialize_asset(back) == bh.serialize_asset(effect)          # spawner, module (expressions + properties), modifiers: identical
    assert (len(back.init_modifiers), len(back.update_modifiers), len(back.render_modifiers)) == (1, 0, 0)
    assert bh.to_ron(back) == s
    # the forms the reference's tests pin
    assert 'value: "#1"' in s and 'attribute: "position"' in s              # expr.rs:4831-4885, attributes.rs:2308-2321
    assert '"bevy_hanabi::modifier::attr::SetAttributeModifier": (' in s    # registry.rs:345-349: the type names appear
    assert "Modifiers" not in s                                             # registry.rs:350-353: the wrapper does not
    assert "Literal(Vector(Vec3((1.2, -3.45, 87.54485))))" in s and "Literal(Vector(BVec2((false, true))))" in s
    assert 'Binary(op: Add, left: "#2", right: "#1")' in s and 'Unary(op: Abs, expr: "#4")' in s and "Property(1)" in s
    assert "count: Single(30.0)" in s and "period: Single(1.0)" in s and "cycle_count: 0" in s
    assert s.startswith("(\n  name: \"Effect\",\n  capacity: 4096,\n  spawner: (\n    count:")   # PrettyConfig: two-space indentor, \n


def test_round_trip_of_the_registry_tests():
    """registry.rs:308-391 (two modifiers of different types keep type, order and fields) and :413-447 (a whole asset)."""
    w = bh.ExprWriter()
    size, zero, one = w.lit(2.0).expr(), w.lit((0.0, 0.0, 0.0)).expr(), w.lit(1.0).expr()
    a = (bh.EffectAsset(16, bh.SpawnerSettings.once(1.0), w.finish())
         .init(bh.SetAttributeModifier(A.SIZE, size)).init(bh.SetPositionSphereModifier(zero, one, bh.ShapeDimension.Surface)))
    back = bh.from_ron(bh.to_ron(a))
    assert bh.serialize_asset(back) == bh.serialize_asset(a)
    w = bh.ExprWriter()
    accel = bh.AccelModifier(w.lit((1.0, 0.0, 0.0)).expr())
    asset = bh.EffectAsset(24, bh.SpawnerSettings.once(3.0), w.finish()).update(accel)
    back = bh.from_ron(bh.to_ron(asset))
    assert back.capacity == 24 and bh.serialize_asset(back) == bh.serialize_asset(asset)


def test_every_suite_asset_round_trips_to_the_same_program():
    from test_host_capi import suite_assets
    for name, asset in suite_assets().items():
        text = bh.to_ron(asset)
        back = bh.from_ron(text)
        assert bh.serialize_asset(back) == bh.serialize_asset(asset), name
        assert bh.lower(back) == bh.lower(asset), name
        assert [x.id for x in back.particle_layout()] == [x.id for x in asset.particle_layout()], name   # render modifiers keep their attributes


HAND_WRITTEN = """
// an .effect file written by hand: comments, an explicit struct name, no trailing commas, floats with exponents
EffectAsset(
    name: "hand", capacity: 2000,
    spawner: (count: Single(2e3), spawn_duration: Single(0), period: Single(0.0), cycle_count: 1, starts_active: true, emit_on_start: true),
    z_layer_2d: 0.0, simulation_space: Global, simulation_condition: WhenVisible, prng_seed: 7,
    init_modifiers: [
        {"bevy_hanabi::modifier::position::SetPositionSphereModifier": (center: "#1", radius: "#2", dimension: Volume)},
        {"bevy_hanabi::modifier::velocity::SetVelocitySphereModifier": (center: "#1", speed: "#3")},
        {"bevy_hanabi::modifier::attr::SetAttributeModifier": (attribute: "age", value: "#4")},
        {"bevy_hanabi::modifier::attr::SetAttributeModifier": (attribute: "lifetime", value: "#7")}
    ],
    update_modifiers: [
        {"bevy_hanabi::modifier::accel::AccelModifier": (accel: "#8")},
        {"bevy_hanabi::modifier::force::LinearDragModifier": (drag: "#9")} /* order matters: accel, then drag */
    ],
    render_modifiers: [ {"bevy_hanabi::modifier::output::ColorOverLifetimeModifier": (gradient: (keys: []), blend: Overwrite, mask: 15)} ],
    motion_integration: PostUpdate,
    module: (
        expressions: [
            Literal(Vector(Vec3((0.0, 0.0, 0.0)))), Literal(Scalar(Float(1.5))), Literal(Scalar(Float(2.5e0))), Literal(Scalar(Float(0))),
            Literal(Scalar(Float(0.5))), Literal(Scalar(Float(1.5))), Binary(op: UniformRand, left: "#5", right: "#6"),
            Literal(Vector(Vec3((0.0, -9.81, 0.0)))), Property(1)
        ],
        properties: [ (name: "drag", default_value: Scalar(Float(0.75))) ],
        texture_layout: (layout: [])
    ),
    alpha_mode: Blend, mesh: None
)
"""


def hand_written_equivalent():
    w = bh.ExprWriter()
    drag = w.add_property("drag", 0.75)
    c, r, s, z = w.lit((0.0, 0.0, 0.0)), w.lit(1.5), w.lit(2.5), w.lit(0.0)
    life = w.lit(0.5).uniform(w.lit(1.5))
    g = w.lit((0.0, -9.81, 0.0))
    d = w.prop(drag)
    return (bh.EffectAsset(2000, bh.SpawnerSettings.once(2000.0), w.finish())
            .init(bh.SetPositionSphereModifier(c.expr(), r.expr(), bh.ShapeDimension.Volume)).init(bh.SetVelocitySphereModifier(c.expr(), s.expr()))
            .init(bh.SetAttributeModifier(A.AGE, z.expr())).init(bh.SetAttributeModifier(A.LIFETIME, life.expr()))
            .update(bh.AccelModifier(g.expr())).update(bh.LinearDragModifier(d.expr())).render(bh.ColorOverLifetimeModifier()))


def test_hand_written_file_loads():
    asset = bh.from_ron(HAND_WRITTEN)
    assert asset.name == "hand" and asset.capacity == 2000 and asset.prng_seed == 7
    want = hand_written_equivalent()
    want.prng_seed = 7
    assert bh.lower(asset) == bh.lower(want)
    assert [p for p in asset.module().property_names] == ["drag"]


def test_deserializer_errors():
    good = bh.to_ron(reference_serde_asset())
    def broken(old, new, count=1):
        assert old in good
        return good.replace(old, new, count)
    cases = {
        "missing field": broken("  capacity: 4096,\n", ""),                                       # asset.rs:926-927
        "duplicate field": broken("  capacity: 4096,\n", "  capacity: 4096,\n  capacity: 12,\n"),   # asset.rs:818-822
        "unknown field": broken("  capacity: 4096,\n", "  capacity: 4096,\n  colour: 3,\n"),
        "Unknown attribute name": broken('attribute: "position"', 'attribute: "UNKNOWN"'),         # attributes.rs:2320-2321
        "no modifier registered": broken("modifier::attr::SetAttributeModifier", "modifier::attr::NoSuchModifier"),  # registry.rs:158-165
        "expected '#N'": broken('value: "#1"', 'value: "1"'),                                       # expr.rs:182-200
        "out of range": broken('value: "#1"', 'value: "#99"'),
        "unknown Expr": None,
        "line": good[: len(good) // 2],                                                               # truncated text: position reported
    }
    for what, text in cases.items():
        if text is None:
            continue
        with pytest.raises(Exception) as ei:
            bh.from_ron(text)
        assert any(k in str(ei.value) for k in (what, "ID", "handle")), (what, str(ei.value))
    for bad in ('""', "()", "(name: 3)", "[1, 2]", "(name: \"x\""):
        with pytest.raises(Exception):
            bh.from_ron(bad)


def test_nesting_limit_is_an_error_not_a_crash():
    """The `ron` crate refuses documents nested deeper than its recursion limit; here a buffer of 2,000,000 '(' used to overflow
    the parser's stack (SIGSEGV through hnb_asset_from_ron). Every bracket kind, through the C ABI (status code, no crash)."""
    lib = C.CDLL(hb.build_host_lib())
    lib.hnb_host_last_error.restype = C.c_char_p
    a = C.c_void_p()
    for opener in (b"(", b"[", b"{", b"Some("):
        text = opener * 2_000_000
        assert lib.hnb_asset_from_ron(text, len(text), C.byref(a)) == -2
        assert b"recursion limit" in lib.hnb_host_last_error()
    with pytest.raises(Exception, match="recursion limit"):
        bh.from_ron("(" * 200)
    # 100 levels are fine for the parser itself (the asset is then rejected for its content, not its depth)
    with pytest.raises(Exception) as ei:
        bh.from_ron("(" * 100 + ")" * 100)
    assert "recursion limit" not in str(ei.value)


def test_int_vector_components_are_range_checked():
    good = bh.to_ron(_ivec_asset())
    assert "IVec3((1, -2, 3))" in good
    assert bh.lower(bh.from_ron(good)) == bh.lower(_ivec_asset())
    for bad in ("3000000000", "-2147483649", "1.5", "NaN", "inf"):
        with pytest.raises(Exception, match="signed 32-bit"):
            bh.from_ron(good.replace("IVec3((1, -2, 3))", f"IVec3((1, {bad}, 3))"))


def _ivec_asset():
    w = bh.ExprWriter()
    pos = w.lit(bh.Value.vec_i([1, -2, 3])).cast(bh.VectorType.VEC3F).expr()
    asset = bh.EffectAsset(16, bh.SpawnerSettings.once(4.0), w.finish())
    asset.init(bh.SetAttributeModifier(bh.Attribute.POSITION, pos))
    return asset


def test_c_abi_round_trip():
    lib = C.CDLL(hb.build_host_lib())
    lib.hnb_host_last_error.restype = C.c_char_p
    lib.hnb_host_free.argtypes = [C.c_void_p]
    lib.hnb_asset_destroy.argtypes = [C.c_void_p]
    text = bh.to_ron(effects.firework_trails(1 << 24)).encode()
    a = C.c_void_p()
    assert lib.hnb_asset_from_ron(text, len(text), C.byref(a)) == 0, lib.hnb_host_last_error()
    out, size = C.c_void_p(), C.c_size_t()
    assert lib.hnb_lower(a, C.byref(out), C.byref(size)) == 0
    assert C.string_at(out, size.value) == bh.lower(effects.firework_trails(1 << 24))
    lib.hnb_host_free(out)
    t, n = C.c_char_p(), C.c_size_t()
    assert lib.hnb_asset_to_ron(a, C.byref(t), C.byref(n)) == 0 and C.string_at(t, n.value) == text
    lib.hnb_asset_destroy(a)
    assert lib.hnb_asset_from_ron(b"(name: 3)", 9, C.byref(a)) == -2 and b"RON" in lib.hnb_host_last_error()


@pytest.mark.gpu
def test_loaded_file_simulates_like_the_original_asset():
    """load -> lower -> simulate on the GPU; the oracle runs the ORIGINAL (never serialised) asset."""
    from helpers import Frame, GpuRunner, OracleRunner, assert_same_state, frame_seed
    loaded = bh.from_ron(HAND_WRITTEN)
    original = hand_written_equivalent()
    original.prng_seed = 7
    ctx = bh.Context(0)
    gpu, orc = GpuRunner(loaded, ctx=ctx), OracleRunner(original)
    for f in range(80):
        fr = Frame(1 / 60, 2000 if f == 0 else 0, frame_seed(f), time=f / 60, props={"drag": 0.75 if f < 30 else 2.0})
        gpu.step(fr)
        orc.step(fr)
    st = gpu.state()
    assert_same_state(orc.state(), st, "hand-written .effect file")
    assert 0 < st["counters"]["alive_count"] < 2000
    ctx.close()
