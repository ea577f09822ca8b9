// hanabi:: host library: data model, CPU spawner, asset serialisation.
// (The lowering lives in lowering.cpp.)
#include "hanabi.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <set>

namespace hanabi {

ExprHandle ExprHandle::parse(const std::string& s) {
    if (s.size() < 2 || s[0] != '#') throw std::invalid_argument("Invalid ID format (expected '#N')");
    // Rust's str::parse::<u32>: an optional '+', then decimal digits only, no overflow
    size_t i = 1;
    if (s[i] == '+') ++i;
    if (i == s.size()) throw std::invalid_argument("Failed to parse ID value");
    uint64_t v = 0;
    for (; i < s.size(); ++i) {
        if (s[i] < '0' || s[i] > '9') throw std::invalid_argument("Failed to parse ID value");
        v = v * 10 + (uint64_t)(s[i] - '0');
        if (v > 0xffffffffull) throw std::invalid_argument("Failed to parse ID value");
    }
    if (v == 0) throw std::invalid_argument("Invalid ID value");
    ExprHandle h;
    h.id = (uint32_t)v;
    return h;
}


std::string ValueType::to_string() const {
    static const char* names[] = {"bool", "f32", "i32", "u32"};
    const char* e = names[(int)elem];
    if (count == 1) return e;
    return "vec" + std::to_string((int)count) + "<" + e + ">";
}

// ---- attributes (src/attributes.rs:549-675) ------------------------------------------------
namespace {
struct AttrInfo { const char* name; ScalarType elem; uint8_t count; uint32_t def[4]; };
constexpr uint32_t F1 = 0x3f800000u;  // 1.0f
const AttrInfo kAttrs[HNB_ATTR_COUNT] = {
    {"id", ScalarType::Uint, 1, {0, 0, 0, 0}},
    {"particle_counter", ScalarType::Uint, 1, {0, 0, 0, 0}},
    {"position", ScalarType::Float, 3, {0, 0, 0, 0}},
    {"velocity", ScalarType::Float, 3, {0, 0, 0, 0}},
    {"age", ScalarType::Float, 1, {0, 0, 0, 0}},
    {"lifetime", ScalarType::Float, 1, {F1, 0, 0, 0}},
    {"color", ScalarType::Uint, 1, {0xffffffffu, 0, 0, 0}},
    {"hdr_color", ScalarType::Float, 4, {F1, F1, F1, F1}},
    {"alpha", ScalarType::Float, 1, {F1, 0, 0, 0}},
    {"size", ScalarType::Float, 1, {F1, 0, 0, 0}},
    {"size2", ScalarType::Float, 2, {F1, F1, 0, 0}},
    {"size3", ScalarType::Float, 3, {F1, F1, F1, 0}},
    {"prev", ScalarType::Uint, 1, {0xffffffffu, 0, 0, 0}},
    {"next", ScalarType::Uint, 1, {0xffffffffu, 0, 0, 0}},
    {"axis_x", ScalarType::Float, 3, {F1, 0, 0, 0}},
    {"axis_y", ScalarType::Float, 3, {0, F1, 0, 0}},
    {"axis_z", ScalarType::Float, 3, {0, 0, F1, 0}},
    {"sprite_index", ScalarType::Int, 1, {0, 0, 0, 0}},
    {"f32_0", ScalarType::Float, 1, {0, 0, 0, 0}}, {"f32_1", ScalarType::Float, 1, {0, 0, 0, 0}},
    {"f32_2", ScalarType::Float, 1, {0, 0, 0, 0}}, {"f32_3", ScalarType::Float, 1, {0, 0, 0, 0}},
    {"f32x2_0", ScalarType::Float, 2, {0, 0, 0, 0}}, {"f32x2_1", ScalarType::Float, 2, {0, 0, 0, 0}},
    {"f32x2_2", ScalarType::Float, 2, {0, 0, 0, 0}}, {"f32x2_3", ScalarType::Float, 2, {0, 0, 0, 0}},
    {"f32x3_0", ScalarType::Float, 3, {0, 0, 0, 0}}, {"f32x3_1", ScalarType::Float, 3, {0, 0, 0, 0}},
    {"f32x3_2", ScalarType::Float, 3, {0, 0, 0, 0}}, {"f32x3_3", ScalarType::Float, 3, {0, 0, 0, 0}},
    {"f32x4_0", ScalarType::Float, 4, {0, 0, 0, 0}}, {"f32x4_1", ScalarType::Float, 4, {0, 0, 0, 0}},
    {"f32x4_2", ScalarType::Float, 4, {0, 0, 0, 0}}, {"f32x4_3", ScalarType::Float, 4, {0, 0, 0, 0}},
    {"u32_0", ScalarType::Uint, 1, {0, 0, 0, 0}}, {"u32_1", ScalarType::Uint, 1, {0, 0, 0, 0}},
    {"u32_2", ScalarType::Uint, 1, {0, 0, 0, 0}}, {"u32_3", ScalarType::Uint, 1, {0, 0, 0, 0}},
    {"ribbon_id", ScalarType::Uint, 1, {0, 0, 0, 0}},
};
}  // namespace

const char* Attribute::name() const { return kAttrs[id].name; }
ValueType Attribute::value_type() const { return ValueType(kAttrs[id].elem, kAttrs[id].count); }
Value Attribute::default_value() const { return Value::from_bits(value_type(), kAttrs[id].def); }
bool Attribute::from_name(const std::string& name, Attribute* out) {
    for (int i = 0; i < HNB_ATTR_COUNT; ++i)
        if (name == kAttrs[i].name) { *out = Attribute((HnbAttr)i); return true; }
    return false;
}
const std::vector<Attribute>& Attribute::all() {
    static const std::vector<Attribute> v = [] {
        std::vector<Attribute> r;
        for (int i = 0; i < HNB_ATTR_COUNT; ++i) r.push_back(Attribute((HnbAttr)i));
        return r;
    }();
    return v;
}
#define HNB_DEF_ATTR(N) const Attribute Attribute::N{HNB_ATTR_##N};
HNB_DEF_ATTR(ID) HNB_DEF_ATTR(PARTICLE_COUNTER) HNB_DEF_ATTR(POSITION) HNB_DEF_ATTR(VELOCITY) HNB_DEF_ATTR(AGE) HNB_DEF_ATTR(LIFETIME)
HNB_DEF_ATTR(COLOR) HNB_DEF_ATTR(HDR_COLOR) HNB_DEF_ATTR(ALPHA) HNB_DEF_ATTR(SIZE) HNB_DEF_ATTR(SIZE2) HNB_DEF_ATTR(SIZE3) HNB_DEF_ATTR(PREV)
HNB_DEF_ATTR(NEXT) HNB_DEF_ATTR(AXIS_X) HNB_DEF_ATTR(AXIS_Y) HNB_DEF_ATTR(AXIS_Z) HNB_DEF_ATTR(SPRITE_INDEX) HNB_DEF_ATTR(F32_0)
HNB_DEF_ATTR(F32_1) HNB_DEF_ATTR(F32_2) HNB_DEF_ATTR(F32_3) HNB_DEF_ATTR(F32X2_0) HNB_DEF_ATTR(F32X2_1) HNB_DEF_ATTR(F32X2_2)
HNB_DEF_ATTR(F32X2_3) HNB_DEF_ATTR(F32X3_0) HNB_DEF_ATTR(F32X3_1) HNB_DEF_ATTR(F32X3_2) HNB_DEF_ATTR(F32X3_3) HNB_DEF_ATTR(F32X4_0)
HNB_DEF_ATTR(F32X4_1) HNB_DEF_ATTR(F32X4_2) HNB_DEF_ATTR(F32X4_3) HNB_DEF_ATTR(U32_0) HNB_DEF_ATTR(U32_1) HNB_DEF_ATTR(U32_2)
HNB_DEF_ATTR(U32_3) HNB_DEF_ATTR(RIBBON_ID)
#undef HNB_DEF_ATTR

// ---- expressions ---------------------------------------------------------------------------------
bool Expr::value_type(ValueType* out) const {
    switch (kind) {
        case Kind::BuiltIn:
            switch (builtin) {
                case BuiltInOperator::Rand: *out = rand_type; return true;
                case BuiltInOperator::IsAlive: *out = ValueType(ScalarType::Bool); return true;
                default: *out = ValueType(ScalarType::Float); return true;
            }
        case Kind::Literal: *out = literal.type; return true;
        case Kind::Attribute:
        case Kind::ParentAttribute: *out = attribute.value_type(); return true;
        case Kind::Cast: *out = rand_type; return true;
        case Kind::TextureSample: *out = VectorType::VEC4F; return true;
        default: return false;  // Property, Unary, Binary, Ternary: unknown to the reference
    }
}

PropertyHandle Module::add_property(const std::string& name, const Value& default_value) {
    for (const Property& p : properties_)
        if (p.name == name) throw PanicError("property '" + name + "' already exists in the module");
    properties_.push_back(Property{name, default_value});
    return PropertyHandle{(uint32_t)properties_.size()};
}
bool Module::get_property_by_name(const std::string& name, PropertyHandle* out) const {
    for (size_t i = 0; i < properties_.size(); ++i)
        if (properties_[i].name == name) { *out = PropertyHandle{(uint32_t)i + 1}; return true; }
    return false;
}
const Expr& Module::try_get(ExprHandle h) const {
    const Expr* e = get(h);
    if (!e)
        throw ExprError(ExprError::InvalidExprHandleError,
                        "Cannot find expression with handle #" + std::to_string(h.id) +
                            " in the current module. Check that the Module used to build the expression was the same used in "
                            "the EvalContext or the original EffectAsset.");
    return *e;
}
ExprHandle Module::cast(ExprHandle inner, ValueType target) {
    check(inner);
    // CastExpr::is_valid (expr.rs:1475-1497): scalar <- scalar only; vector <- {scalar, vector}
    ValueType it;
    if (expressions_[inner.index()].value_type(&it)) {
        if (target.is_scalar() && !it.is_scalar()) throw PanicError("invalid cast: cannot cast a vector to a scalar");
    }
    Expr e;
    e.kind = Expr::Kind::Cast;
    e.a = inner;
    e.rand_type = target;
    return add_expr(e);
}
bool Module::is_const(ExprHandle h) const {
    const Expr& e = try_get(h);
    switch (e.kind) {
        case Expr::Kind::Literal: return true;
        case Expr::Kind::Unary:
        case Expr::Kind::Cast: return is_const(e.a);
        case Expr::Kind::Binary: return is_const(e.a) && is_const(e.b);
        case Expr::Kind::Ternary: return is_const(e.a) && is_const(e.b) && is_const(e.c);
        default: return false;
    }
}
WriterExpr WriterExpr::cast(ValueType target) const { return WriterExpr(module_->cast(handle_, target), module_); }

// ---- modifiers ---------------------------------------------------------------------------------------
uint32_t Modifier::context() const {
    switch (kind) {
        case Kind::SetAttribute: case Kind::SetPositionCircle: case Kind::SetPositionSphere: case Kind::SetPositionCone3d:
        case Kind::SetVelocityCircle: case Kind::SetVelocitySphere: case Kind::SetVelocityTangent:
            return CONTEXT_INIT | CONTEXT_UPDATE;
        case Kind::InheritAttribute: return CONTEXT_INIT;
        case Kind::Render: return CONTEXT_RENDER;
        default: return CONTEXT_UPDATE;
    }
}
std::vector<Attribute> Modifier::attributes() const {
    switch (kind) {
        case Kind::SetAttribute: case Kind::InheritAttribute: return {attribute};
        case Kind::SetPositionCircle: case Kind::SetPositionSphere: case Kind::SetPositionCone3d: return {Attribute::POSITION};
        case Kind::SetVelocityCircle: case Kind::SetVelocitySphere: case Kind::SetVelocityTangent:
            return {Attribute::POSITION, Attribute::VELOCITY};
        case Kind::Accel: case Kind::LinearDrag: return {Attribute::VELOCITY};
        case Kind::RadialAccel: case Kind::TangentAccel: case Kind::ConformToSphere: return {Attribute::POSITION, Attribute::VELOCITY};
        case Kind::KillSphere: case Kind::KillAabb: return {Attribute::POSITION};
        case Kind::EmitSpawnEvent: return {};
        case Kind::Render: return render_attributes;
    }
    return {};
}

Modifier SetAttributeModifier(Attribute attribute, ExprHandle value) {
    // attr.rs:81-88
    if (attribute == Attribute::ID) throw PanicError("The particle's ID is a read-only pseudo-attribute, cannot be assigned.");
    if (attribute == Attribute::PARTICLE_COUNTER)
        throw PanicError("The PARTICLE_COUNTER attribute is a read-only pseudo-attribute, cannot be assigned.");
    Modifier m; m.kind = Modifier::Kind::SetAttribute; m.attribute = attribute; m.e[0] = value; return m;
}
Modifier InheritAttributeModifier(Attribute attribute) { Modifier m; m.kind = Modifier::Kind::InheritAttribute; m.attribute = attribute; return m; }
Modifier SetPositionCircleModifier(ExprHandle center, ExprHandle axis, ExprHandle radius, ShapeDimension dimension) {
    Modifier m; m.kind = Modifier::Kind::SetPositionCircle; m.e[0] = center; m.e[1] = axis; m.e[2] = radius; m.dimension = dimension; return m;
}
Modifier SetPositionSphereModifier(ExprHandle center, ExprHandle radius, ShapeDimension dimension) {
    Modifier m; m.kind = Modifier::Kind::SetPositionSphere; m.e[0] = center; m.e[1] = radius; m.dimension = dimension; return m;
}
Modifier SetPositionCone3dModifier(ExprHandle height, ExprHandle base_radius, ExprHandle top_radius, ShapeDimension dimension) {
    Modifier m; m.kind = Modifier::Kind::SetPositionCone3d; m.e[0] = height; m.e[1] = base_radius; m.e[2] = top_radius; m.dimension = dimension; return m;
}
Modifier SetVelocityCircleModifier(ExprHandle center, ExprHandle axis, ExprHandle speed) {
    Modifier m; m.kind = Modifier::Kind::SetVelocityCircle; m.e[0] = center; m.e[1] = axis; m.e[2] = speed; return m;
}
Modifier SetVelocitySphereModifier(ExprHandle center, ExprHandle speed) {
    Modifier m; m.kind = Modifier::Kind::SetVelocitySphere; m.e[0] = center; m.e[1] = speed; return m;
}
Modifier SetVelocityTangentModifier(ExprHandle origin, ExprHandle axis, ExprHandle speed) {
    Modifier m; m.kind = Modifier::Kind::SetVelocityTangent; m.e[0] = origin; m.e[1] = axis; m.e[2] = speed; return m;
}
Modifier AccelModifier(ExprHandle accel) { Modifier m; m.kind = Modifier::Kind::Accel; m.e[0] = accel; return m; }
Modifier RadialAccelModifier(ExprHandle origin, ExprHandle accel) {
    Modifier m; m.kind = Modifier::Kind::RadialAccel; m.e[0] = origin; m.e[1] = accel; return m;
}
Modifier TangentAccelModifier(ExprHandle origin, ExprHandle axis, ExprHandle accel) {
    Modifier m; m.kind = Modifier::Kind::TangentAccel; m.e[0] = origin; m.e[1] = axis; m.e[2] = accel; return m;
}
Modifier LinearDragModifier(ExprHandle drag) { Modifier m; m.kind = Modifier::Kind::LinearDrag; m.e[0] = drag; return m; }
static Value vec3_value(Vec3Lit v) {
    Value r;
    r.type = VectorType::VEC3F;
    r.set_f(0, v.x); r.set_f(1, v.y); r.set_f(2, v.z);
    return r;
}
Modifier AccelModifierConstant(Module& module, Vec3Lit acceleration) { return AccelModifier(module.lit(vec3_value(acceleration))); }
Modifier AccelModifierViaProperty(Module& module, PropertyHandle property) { return AccelModifier(module.prop(property)); }
Modifier RadialAccelModifierConstant(Module& module, Vec3Lit origin, float acceleration) {
    const ExprHandle o = module.lit(vec3_value(origin));  // field order of the struct literal: origin first (accel.rs:142-146)
    return RadialAccelModifier(o, module.lit(Value(acceleration)));
}
Modifier RadialAccelModifierViaProperty(Module& module, Vec3Lit origin, PropertyHandle property) {
    const ExprHandle o = module.lit(vec3_value(origin));
    return RadialAccelModifier(o, module.prop(property));
}
Modifier TangentAccelModifierConstant(Module& module, Vec3Lit origin, Vec3Lit axis, float acceleration) {
    const ExprHandle o = module.lit(vec3_value(origin));
    const ExprHandle a = module.lit(vec3_value(axis));
    return TangentAccelModifier(o, a, module.lit(Value(acceleration)));
}
Modifier TangentAccelModifierViaProperty(Module& module, Vec3Lit origin, Vec3Lit axis, PropertyHandle property) {
    const ExprHandle o = module.lit(vec3_value(origin));
    const ExprHandle a = module.lit(vec3_value(axis));
    return TangentAccelModifier(o, a, module.prop(property));
}
Modifier LinearDragModifierConstant(Module& module, float drag) { return LinearDragModifier(module.lit(Value(drag))); }
Modifier ConformToSphereModifier(ExprHandle origin, ExprHandle radius, ExprHandle influence_dist, ExprHandle attraction_accel,
                                 ExprHandle max_attraction_speed, ExprHandle shell_half_thickness, ExprHandle sticky_factor) {
    Modifier m;
    m.kind = Modifier::Kind::ConformToSphere;
    m.e[0] = origin; m.e[1] = radius; m.e[2] = influence_dist; m.e[3] = attraction_accel; m.e[4] = max_attraction_speed;
    m.e[5] = shell_half_thickness; m.e[6] = sticky_factor;
    m.has_shell = shell_half_thickness.valid();
    m.has_sticky = sticky_factor.valid();
    return m;
}
Modifier KillSphereModifier(ExprHandle center, ExprHandle sqr_radius, bool kill_inside) {
    Modifier m; m.kind = Modifier::Kind::KillSphere; m.e[0] = center; m.e[1] = sqr_radius; m.kill_inside = kill_inside; return m;
}
Modifier KillAabbModifier(ExprHandle center, ExprHandle half_size, bool kill_inside) {
    Modifier m; m.kind = Modifier::Kind::KillAabb; m.e[0] = center; m.e[1] = half_size; m.kill_inside = kill_inside; return m;
}
Modifier EmitSpawnEventModifier(EventEmitCondition condition, ExprHandle count, uint32_t child_index) {
    Modifier m; m.kind = Modifier::Kind::EmitSpawnEvent; m.condition = condition; m.e[0] = count; m.child_index = child_index; return m;
}
Modifier RenderModifier(const std::string& name, const std::vector<Attribute>& attributes) {
    Modifier m; m.kind = Modifier::Kind::Render; m.render_name = name; m.render_attributes = attributes; return m;
}
Modifier ColorOverLifetimeModifier() { return RenderModifier("ColorOverLifetimeModifier", {Attribute::AGE, Attribute::LIFETIME}); }
Modifier SizeOverLifetimeModifier() { return RenderModifier("SizeOverLifetimeModifier", {Attribute::AGE, Attribute::LIFETIME}); }
Modifier SetColorModifier() { return RenderModifier("SetColorModifier", {}); }
Modifier SetSizeModifier() { return RenderModifier("SetSizeModifier", {}); }
Modifier OrientModifier(OrientMode mode) {
    std::vector<Attribute> a;
    if (mode == OrientMode::FaceCameraPosition) a = {Attribute::POSITION};
    if (mode == OrientMode::AlongVelocity) a = {Attribute::POSITION, Attribute::VELOCITY};
    return RenderModifier("OrientModifier", a);
}
Modifier FlipbookModifier() { return RenderModifier("FlipbookModifier", {Attribute::SPRITE_INDEX}); }
Modifier ScreenSpaceSizeModifier() { return RenderModifier("ScreenSpaceSizeModifier", {Attribute::POSITION, Attribute::SIZE}); }
Modifier RoundModifier() { return RenderModifier("RoundModifier", {}); }
Modifier ParticleTextureModifier() { return RenderModifier("ParticleTextureModifier", {}); }

// ---- spawner (src/spawn.rs) ---------------------------------------------------------------------------
Pcg32::Pcg32(uint64_t seed_state, uint64_t stream) {
    inc = (stream << 1) | 1u;
    state = seed_state + inc;
    state = state * 6364136223846793005ull + inc;
}
uint32_t Pcg32::next_u32() {
    const uint64_t old = state;
    state = old * 6364136223846793005ull + inc;
    const uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    const uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
}

// ---- per-frame seed evolution (src/lib.rs:1813-1820; rand's StdRng, see hanabi.hpp) -----------------------------------
void chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int rounds, uint32_t out[16]) {
    uint32_t init[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u,   // "expand 32-byte k"
                         key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                         (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
    uint32_t s[16];
    std::memcpy(s, init, sizeof s);
    auto rotl = [](uint32_t x, int n) { return (x << n) | (x >> (32 - n)); };
    auto qr = [&](int a, int b, int c, int d) {
        s[a] += s[b]; s[d] = rotl(s[d] ^ s[a], 16);
        s[c] += s[d]; s[b] = rotl(s[b] ^ s[c], 12);
        s[a] += s[b]; s[d] = rotl(s[d] ^ s[a], 8);
        s[c] += s[d]; s[b] = rotl(s[b] ^ s[c], 7);
    };
    for (int r = 0; r < rounds / 2; ++r) {
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);   // column round
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);   // diagonal round
    }
    for (int i = 0; i < 16; ++i) out[i] = s[i] + init[i];
}
void seed_from_u64(uint64_t state, uint8_t out_seed[32]) {
    for (int chunk = 0; chunk < 8; ++chunk) {   // PCG32 (XSH-RR), state advanced BEFORE each output
        state = state * 6364136223846793005ull + 11634580027462260723ull;
        const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
        const uint32_t rot = (uint32_t)(state >> 59);
        const uint32_t x = (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
        for (int b = 0; b < 4; ++b) out_seed[chunk * 4 + b] = (uint8_t)(x >> (8 * b));   // to_le_bytes
    }
}
uint32_t next_prng_seed(uint32_t prng_seed) {
    uint8_t seed[32];
    seed_from_u64((uint64_t)prng_seed, seed);
    uint32_t key[8], out[16];
    for (int i = 0; i < 8; ++i) key[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) | ((uint32_t)seed[4 * i + 2] << 16) | ((uint32_t)seed[4 * i + 3] << 24);
    chacha_block(key, 0, 0, 12, out);
    return out[0];
}

// CpuValue::sample (spawn.rs:105-110). `Uniform::new_inclusive(a, b).sample(rng)` is the
// `rand` crate's UniformFloat: value in [0,1) from the top 23 bits, times a scale adjusted
// so the maximum never exceeds `b`. Third-party algorithm, restated; parity unpinned.
float CpuValue::sample(Pcg32& rng) const {
    if (!is_uniform) return a;
    const float max_rand = 1.0f - 1.1920929e-7f;  // (u32::MAX >> 9) as mantissa, minus 1
    float scale = (b - a) / max_rand;
    while (!(scale * max_rand + a <= b)) scale = std::nextafter(scale, -INFINITY);
    uint32_t bits = (rng.next_u32() >> 9) | 0x3f800000u;
    float v12;
    std::memcpy(&v12, &bits, 4);
    return (v12 - 1.0f) * scale + a;
}
std::array<float, 2> CpuValue::range() const {
    if (!is_uniform) return {a, a};
    return a <= b ? std::array<float, 2>{a, b} : std::array<float, 2>{b, a};
}

SpawnerSettings SpawnerSettings::try_make(CpuValue count, CpuValue spawn_duration, CpuValue period, uint32_t cycle_count) {
    const auto range = period.range();
    if (cycle_count != 1 && (range[0] < 0.0f || range[1] <= 0.0f))
        throw SpawnerSettingsError(SpawnerSettingsError::InvalidPeriod, range[0], range[1], "invalid period");
    if (!std::isfinite(range[0]) || !std::isfinite(range[1]))
        throw SpawnerSettingsError(SpawnerSettingsError::InfinitePeriod, range[0], range[1], "infinite period");
    return SpawnerSettings(count, spawn_duration, period, cycle_count);
}
SpawnerSettings SpawnerSettings::make(CpuValue count, CpuValue spawn_duration, CpuValue period, uint32_t cycle_count) {
    try {
        return try_make(count, spawn_duration, period, cycle_count);
    } catch (const SpawnerSettingsError& e) {
        if (e.kind == SpawnerSettingsError::InvalidPeriod) {
            if (e.min < 0.0f)
                throw PanicError("`period` must not generate negative numbers (period.min was " + std::to_string(e.min) + ", expected >= 0).");
            throw PanicError("`period` must be able to generate a positive number (period.max was " + std::to_string(e.max) + ", expected > 0).");
        }
        throw PanicError("`period` has an infinite bound. If upgrading from a previous version, use `cycle_count = 1` instead for a single-cycle burst.");
    }
}
void SpawnerSettings::set_period(CpuValue v) {
    const auto r = v.range();
    if (!std::isfinite(r[0]) || !std::isfinite(r[1]))
        throw PanicError("`period` has an infinite bound. If upgrading from a previous version, use `cycle_count = 1` instead for a single-cycle burst.");
    period_ = v;
}

EffectSpawner::EffectSpawner(const SpawnerSettings& s) : settings(s) {
    // spawn.rs:699-717
    completed_cycle_count_ = (s.emits_on_start() || s.is_forever()) ? 0u : s.cycle_count();
    active = s.starts_active();
}
void EffectSpawner::reset() {
    cycle_time_ = 0; completed_cycle_count_ = 0; sampled_spawn_duration_ = 0; sampled_period_ = 0; sampled_count_ = 0;
    spawn_count = 0; spawn_remainder_ = 0;
}
// EffectSpawner::tick (spawn.rs:838-921), statement for statement.
uint32_t EffectSpawner::tick(float dt, Pcg32& rng) {
    if (!active || (!settings.is_forever() && completed_cycle_count_ >= settings.cycle_count())) {
        spawn_count = 0;
        return 0;
    }
    for (;;) {
        if (sampled_period_ == 0.0f) {
            if (settings.is_once()) {
                sampled_spawn_duration_ = settings.spawn_duration().sample(rng);
                sampled_period_ = std::max(sampled_spawn_duration_, 1e-12f);
            } else {
                sampled_period_ = settings.period().sample(rng);
                if (!(sampled_period_ > 0.0f)) throw PanicError("assertion failed: self.sampled_period > 0.");
                const float d = settings.spawn_duration().sample(rng);
                sampled_spawn_duration_ = std::min(std::max(d, 0.0f), sampled_period_);
            }
            // spawn.rs:867: the duration is sampled again unconditionally (undoing the clamp above)
            sampled_spawn_duration_ = settings.spawn_duration().sample(rng);
            sampled_count_ = std::max(settings.count().sample(rng), 0.0f);
        }
        const float new_time = cycle_time_ + dt;
        if (cycle_time_ <= sampled_spawn_duration_) {
            if (sampled_spawn_duration_ < std::max(1e-5f, dt / 100.0f)) {
                spawn_remainder_ += sampled_count_;
            } else {
                float ratio = (std::min(new_time, sampled_spawn_duration_) - cycle_time_) / sampled_spawn_duration_;
                ratio = std::min(std::max(ratio, 0.0f), 1.0f);
                spawn_remainder_ += sampled_count_ * ratio;
            }
        }
        cycle_time_ = new_time;
        if (cycle_time_ >= sampled_period_) {
            dt = cycle_time_ - sampled_period_;
            cycle_time_ = 0.0f;
            completed_cycle_count_ += 1;
            sampled_period_ = 0.0f;
            if (!settings.is_forever() && completed_cycle_count_ >= settings.cycle_count()) break;
        } else {
            break;
        }
    }
    const float count = std::floor(spawn_remainder_);
    spawn_remainder_ -= count;
    // `count as u32`: saturating float -> int conversion
    spawn_count = count <= 0.0f ? 0u : (count >= 4294967296.0f ? 0xffffffffu : (uint32_t)count);
    return spawn_count;
}

// ---- asset -------------------------------------------------------------------------------------------------
EffectAsset& EffectAsset::init(const Modifier& m) {
    if (!(m.context() & CONTEXT_INIT)) throw PanicError("assertion failed: modifier.context().contains(ModifierContext::Init)");
    init_.push_back(m);
    return *this;
}
EffectAsset& EffectAsset::update(const Modifier& m) {
    if (!(m.context() & CONTEXT_UPDATE)) throw PanicError("assertion failed: modifier.context().contains(ModifierContext::Update)");
    update_.push_back(m);
    return *this;
}
EffectAsset& EffectAsset::render(const Modifier& m) {
    if (!(m.context() & CONTEXT_RENDER)) throw PanicError("assertion failed: modifier.context().contains(ModifierContext::Render)");
    render_.push_back(m);
    return *this;
}
EffectAsset& EffectAsset::add_modifier(uint32_t context, const Modifier& m) {
    if (context != CONTEXT_INIT && context != CONTEXT_UPDATE)
        throw PanicError("assertion failed: context == ModifierContext::Init || context == ModifierContext::Update");
    return context == CONTEXT_INIT ? init(m) : update(m);
}
std::vector<Attribute> EffectAsset::particle_layout() const {
    std::set<int> ids;
    for (const auto* list : {&init_, &update_, &render_})
        for (const Modifier& m : *list)
            for (Attribute a : m.attributes()) ids.insert((int)a.id);
    for (const Expr& e : module_.expressions())
        if (e.kind == Expr::Kind::Attribute) ids.insert((int)e.attribute.id);
    std::vector<Attribute> out;
    for (int i : ids) out.push_back(Attribute((HnbAttr)i));
    return out;
}

ParticleLayout EffectAsset::reference_particle_layout() const {
    ParticleLayout::Builder b = ParticleLayout::make();
    for (Attribute a : particle_layout())
        if (!a.is_pseudo()) b.append(a);
    return b.build();
}

// ---- ParticleLayout (attributes.rs:1516-1670) --------------------------------------------------------------
ParticleLayout ParticleLayout::Builder::build() const {
    // remove duplicates, sort by size (ties by name: the reference sorts by name first, then by size)
    std::vector<Attribute> v = attrs_;
    std::sort(v.begin(), v.end(), [](Attribute x, Attribute y) { return std::string(x.name()) < std::string(y.name()); });
    v.erase(std::unique(v.begin(), v.end()), v.end());
    std::stable_sort(v.begin(), v.end(), [](Attribute x, Attribute y) { return x.size() < y.size(); });
    ParticleLayout out;
    out.unpadded_len_ = (uint32_t)v.size();
    std::vector<Attribute> s1, s2, s3, s4;
    for (Attribute a : v) (a.size() >= 16 ? s4 : a.size() >= 12 ? s3 : a.size() >= 8 ? s2 : s1).push_back(a);
    uint32_t offset = 0, align = 4;
    auto push = [&](Attribute a, uint32_t bytes) { out.layout_.push_back(AttributeLayout{a, offset, false}); offset += bytes; };
    auto pad = [&](uint32_t at) { AttributeLayout p; p.attribute = Attribute::F32_0; p.offset = at; p.padding = true; out.layout_.push_back(p); };
    for (Attribute a : s4) push(a, 16);
    if (!s4.empty()) align = 16;
    if (!s3.empty()) align = 16; else if (!s2.empty()) align = std::max<uint32_t>(align, 8);
    const size_t pairs = std::min(s1.size(), s3.size());
    for (size_t i = 0; i < pairs; ++i) { push(s3[i], 12); push(s1[i], 4); }        // { vec3 + scalar }
    for (size_t i = 0; i + 1 < s2.size(); i += 2) { push(s2[i], 8); push(s2[i + 1], 8); }  // { vec2 + vec2 }
    for (size_t i = pairs; i < s3.size(); ++i) { push(s3[i], 12); pad(offset); offset += 4; }  // vec3 + padding field
    if (s2.size() % 2) push(s2.back(), 8);
    for (size_t i = pairs; i < s1.size(); ++i) push(s1[i], 4);
    for (uint32_t rem = (offset + align - 1) / align * align - offset; rem > 0; rem -= 4) { pad(offset); offset += 4; }
    out.align_ = align;
    return out;
}
ParticleLayout ParticleLayout::default_layout() {
    return make().append(Attribute::POSITION).append(Attribute::AGE).append(Attribute::VELOCITY).append(Attribute::LIFETIME).build();
}
uint32_t ParticleLayout::size() const {
    if (layout_.empty()) return 0;
    const AttributeLayout& last = layout_.back();
    return last.offset + (last.padding ? 4u : last.attribute.size());
}
bool ParticleLayout::contains(Attribute a) const {
    for (const AttributeLayout& e : layout_) if (!e.padding && e.attribute == a) return true;
    return false;
}
bool ParticleLayout::byte_offset(Attribute a, uint32_t* out) const {
    for (const AttributeLayout& e : layout_) if (!e.padding && e.attribute == a) { *out = e.offset; return true; }
    return false;
}
ParticleLayout ParticleLayout::merged_with(const std::vector<Attribute>& more) const {
    Builder b = make();
    for (const AttributeLayout& e : layout_) if (!e.padding) b.append(e.attribute);
    for (Attribute a : more) b.append(a);
    return b.build();
}

// ---- PropertyLayout (src/properties.rs:521-842) -------------------------------------------------------------------------
namespace {
uint32_t prop_size(const Property& p) { return 4u * p.default_value.type.count; }                       // f32 4, vec2 8, vec3 12, vec4 16
uint32_t prop_align(const Property& p) { return p.default_value.type.count == 1 ? 4u : p.default_value.type.count == 2 ? 8u : 16u; }  // WGSL
}  // namespace

PropertyLayout::PropertyLayout(const std::vector<Property>& properties) {
    // properties.rs:563-680. The reference sorts by size with an unstable sort, which keeps the input order of equal
    // sizes for the short lists it is used on (its tests pin that order); a stable sort states it.
    std::vector<const Property*> sorted;
    for (const Property& p : properties) sorted.push_back(&p);
    std::stable_sort(sorted.begin(), sorted.end(), [](const Property* a, const Property* b) { return prop_size(*a) < prop_size(*b); });
    auto first_at_least = [&](uint32_t size) {
        size_t i = 0;
        while (i < sorted.size() && prop_size(*sorted[i]) < size) ++i;
        return i;
    };
    uint32_t offset = 0;
    auto put = [&](const Property* p, uint32_t advance) { layout_.push_back(Entry{*p, offset}); offset += advance; };
    const size_t index4 = first_at_least(16), index3 = first_at_least(12), index2 = first_at_least(8);
    for (size_t i = index4; i < sorted.size(); ++i) put(sorted[i], 16);                 // vec4: already aligned
    size_t num1 = index2, num2 = index3 - index2, num3 = index4 - index3;
    const size_t pairs = std::min(num1, num3);
    for (size_t i = 0; i < pairs; ++i) { put(sorted[index3 + i], 12); put(sorted[i], 4); }  // {vec3 + scalar}
    const size_t i1 = pairs, i3 = index3 + pairs;
    num1 -= pairs; num3 -= pairs;
    for (size_t i = 0; i < num2 / 2; ++i)                                                  // {vec2 + vec2}
        for (size_t j = 0; j < 2; ++j) put(sorted[index2 + i * 2 + j], 8);
    const size_t i2 = index2 + (num2 / 2) * 2;
    num2 %= 2;
    if (num3 > num1) {  // scalars are used up: the remaining vec3 take 16 bytes each (WGSL alignment), then the odd vec2
        for (size_t i = 0; i < num3; ++i) put(sorted[i3 + i], 16);
        if (num2) put(sorted[i2], 0);
    } else {            // vec3 are used up: the odd vec2, then the remaining scalars
        if (num2) put(sorted[i2], 8);
        for (size_t i = 0; i < num1; ++i) put(sorted[i1 + i], 4);
    }
}
uint32_t PropertyLayout::cpu_size() const { return layout_.empty() ? 0u : layout_.back().offset + prop_size(layout_.back().property); }
uint32_t PropertyLayout::align() const {
    uint32_t a = 0;
    for (const Entry& e : layout_) a = std::max(a, prop_align(e.property));
    return a;
}
uint32_t PropertyLayout::min_binding_size() const {
    if (layout_.empty()) throw PanicError("Cannot compute min binding size for empty property layout.");
    const uint32_t a = align();
    return (cpu_size() + a - 1) / a * a;
}
bool PropertyLayout::contains(const std::string& name) const {
    for (const Entry& e : layout_) if (e.property.name == name) return true;
    return false;
}
bool PropertyLayout::offset(const std::string& name, uint32_t* out) const {
    for (const Entry& e : layout_) if (e.property.name == name) { *out = e.offset; return true; }
    return false;
}
std::string PropertyLayout::generate_property_struct_code() const {
    if (layout_.empty()) return "";
    std::string s = "struct Properties {\n";
    for (const Entry& e : layout_) s += "    " + e.property.name + ": " + e.property.default_value.type.to_string() + ",\n";
    return s + "}\n";
}
std::vector<uint8_t> PropertyLayout::serialize(const std::vector<Property>& values) const {
    std::vector<uint8_t> data(cpu_size(), 0);
    for (const Property& p : values) {
        uint32_t off;
        if (!offset(p.name, &off)) continue;
        std::memcpy(data.data() + off, p.default_value.bits, prop_size(p));
    }
    return data;
}

// ---- EffectProperties (src/properties.rs:200-453) -------------------------------------------------------------------------
namespace {
bool same_value(const Value& a, const Value& b) {
    if (a.type != b.type) return false;
    for (int i = 0; i < a.type.count; ++i) if (a.bits[i] != b.bits[i]) return false;
    return true;
}
}  // namespace
EffectProperties& EffectProperties::with_properties(const std::vector<std::pair<std::string, Value>>& properties) {
    for (const auto& kv : properties) {
        bool found = false;
        for (Instance& pi : properties_) {
            if (pi.def.name != kv.first) continue;
            if (pi.value.type != kv.second.type)
                throw PanicError("Trying to overwrite existing property '" + kv.first + "' with value of type " + kv.second.type.to_string() +
                                 ", but property has type " + pi.value.type.to_string());
            pi.value = kv.second;  // the definition keeps its original default (properties.rs:224-233)
            found = true;
            break;
        }
        if (!found) properties_.push_back(Instance{Property{kv.first, kv.second}, kv.second});
    }
    return *this;
}
bool EffectProperties::get_stored(const std::string& name, Value* out) const {
    for (const Instance& pi : properties_) if (pi.def.name == name) { *out = pi.value; return true; }
    return false;
}
void EffectProperties::set(const std::string& name, const Value& value) { (void)set_if_changed(name, value); }
bool EffectProperties::set_if_changed(const std::string& name, const Value& value) {
    for (Instance& pi : properties_) {
        if (pi.def.name != name) continue;
        if (pi.def.default_value.type != value.type)
            throw PanicError("Cannot assign value of type " + value.type.to_string() + " to property '" + name + "' of type " + pi.def.default_value.type.to_string());
        if (same_value(pi.value, value)) return false;
        pi.value = value;
        return true;
    }
    properties_.push_back(Instance{Property{name, value}, value});
    return true;
}
void EffectProperties::update(const std::vector<Property>& asset_properties) {
    // properties.rs:395-430: instances the asset does not declare are dropped, declared properties without an instance
    // are appended with their default value, the others keep their current value and position
    std::vector<Instance> kept, added;
    for (const Property& prop : asset_properties) {
        bool known = false;
        for (const Instance& pi : properties_) known = known || pi.def.name == prop.name;
        if (!known) added.push_back(Instance{prop, prop.default_value});
    }
    for (const Instance& pi : properties_) {
        bool declared = false;
        for (const Property& prop : asset_properties) declared = declared || prop.name == pi.def.name;
        if (declared) kept.push_back(pi);
    }
    properties_ = kept;
    properties_.insert(properties_.end(), added.begin(), added.end());
}
std::vector<uint8_t> EffectProperties::serialize(const PropertyLayout& layout) const {
    std::vector<Property> values;
    for (const Instance& pi : properties_) values.push_back(Property{pi.def.name, pi.value});
    return layout.serialize(values);
}

// `ToWgslString for f32` (src/lib.rs:264-269): format!("{:.6}") then parsed back by the WGSL
// front end as an abstract float converted to f32.
float round_literal_f32(float x) {
    if (!std::isfinite(x)) return x;
    char buf[400];
    std::snprintf(buf, sizeof buf, "%.6f", (double)x);
    return (float)std::strtod(buf, nullptr);
}

// ---- flat asset description ------------------------------------------------------------------------------------
namespace {
struct Writer {
    std::vector<uint8_t> b;
    void u32(uint32_t v) { const uint8_t* p = reinterpret_cast<const uint8_t*>(&v); b.insert(b.end(), p, p + 4); }
    void f32(float v) { uint32_t u; std::memcpy(&u, &v, 4); u32(u); }
    void bytes(const void* p, size_t n) { const uint8_t* q = static_cast<const uint8_t*>(p); b.insert(b.end(), q, q + n); }
};
void write_cpu_value(Writer& w, const CpuValue& v) { w.f32(v.a); w.f32(v.b); w.u32(v.is_uniform ? 1u : 0u); }
void write_modifier(Writer& w, const Modifier& m) {
    w.u32((uint32_t)m.kind);
    w.u32((uint32_t)m.attribute.id);
    for (int i = 0; i < 7; ++i) w.u32(m.e[i].id);
    w.u32((m.has_shell ? 1u : 0u) | (m.has_sticky ? 2u : 0u) | (m.kill_inside ? 4u : 0u));
    w.u32((uint32_t)m.dimension);
    w.u32((uint32_t)m.condition);
    w.u32(m.child_index);
    w.u32((uint32_t)std::min<size_t>(m.render_attributes.size(), 8));
    for (size_t i = 0; i < 8; ++i) w.u32(i < m.render_attributes.size() ? (uint32_t)m.render_attributes[i].id : 0u);
}
}  // namespace

// Layout (all little-endian u32 / f32), consumed by oracle/hanabi_oracle.c:
//   header[24], exprs[n][14], props[n][12 + 6], init mods, update mods, render mods ([22] each)
std::vector<uint8_t> serialize_asset(const EffectAsset& asset) {
    Writer w;
    const Module& m = asset.module();
    w.u32(0x31534148u);  // "HAS1"
    w.u32(1);
    w.u32(asset.capacity());
    w.u32((uint32_t)asset.simulation_space);
    w.u32((uint32_t)asset.simulation_condition);
    w.u32((uint32_t)asset.motion_integration);
    w.u32(asset.prng_seed);
    w.u32((uint32_t)m.expressions().size());
    w.u32((uint32_t)m.properties().size());
    w.u32((uint32_t)asset.init_modifiers().size());
    w.u32((uint32_t)asset.update_modifiers().size());
    w.u32((uint32_t)asset.render_modifiers().size());
    write_cpu_value(w, asset.spawner.count());
    write_cpu_value(w, asset.spawner.spawn_duration());
    write_cpu_value(w, asset.spawner.period());
    w.u32(asset.spawner.cycle_count());
    w.u32(asset.spawner.starts_active() ? 1u : 0u);
    w.u32(asset.spawner.emits_on_start() ? 1u : 0u);
    for (const Expr& e : m.expressions()) {
        w.u32((uint32_t)e.kind);
        uint32_t op = 0;
        ValueType vt;
        switch (e.kind) {
            case Expr::Kind::BuiltIn: op = (uint32_t)e.builtin; vt = e.rand_type; break;
            case Expr::Kind::Unary: op = (uint32_t)e.unary; break;
            case Expr::Kind::Binary: op = (uint32_t)e.binary; break;
            case Expr::Kind::Ternary: op = (uint32_t)e.ternary; break;
            case Expr::Kind::Cast: vt = e.rand_type; break;
            case Expr::Kind::Literal: vt = e.literal.type; break;
            default: break;
        }
        w.u32(op);
        w.u32(e.a.id); w.u32(e.b.id); w.u32(e.c.id);
        w.u32((uint32_t)vt.elem); w.u32(vt.count);
        for (int i = 0; i < 4; ++i) w.u32(e.kind == Expr::Kind::Literal ? e.literal.bits[i] : 0u);
        w.u32((uint32_t)e.attribute.id);
        w.u32(e.property.id);
        w.u32(0);
    }
    for (const Property& p : m.properties()) {
        char name[48] = {0};
        std::snprintf(name, sizeof name, "%s", p.name.c_str());
        w.bytes(name, sizeof name);
        w.u32((uint32_t)p.default_value.type.elem);
        w.u32(p.default_value.type.count);
        for (int i = 0; i < 4; ++i) w.u32(p.default_value.bits[i]);
    }
    for (const Modifier& md : asset.init_modifiers()) write_modifier(w, md);
    for (const Modifier& md : asset.update_modifiers()) write_modifier(w, md);
    for (const Modifier& md : asset.render_modifiers()) write_modifier(w, md);
    return w.b;
}

}  // namespace hanabi
