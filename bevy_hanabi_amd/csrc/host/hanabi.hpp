// hanabi:: — C++17 host-side mirror of bevy_hanabi's authoring API for the simulation path.
//
// The reference is a Rust crate; no Rust toolchain exists in this environment, so the
// host side above the C ABI (include/hanabi_amd.h) is C++. Names, field meaning, defaults,
// insertion-order semantics and failure modes follow the reference:
//   Attribute            src/attributes.rs:549-675,1338-1378
//   Value / ValueType    src/graph/mod.rs
//   Expr / Module / ExprWriter / WriterExpr   src/graph/expr.rs:337-778,910-995,2399-4128
//   modifiers            src/modifier/{attr,position,velocity,accel,force,kill}.rs
//   SpawnerSettings / EffectSpawner           src/spawn.rs:255-922
//   EffectAsset          src/asset.rs:272-646
// Rust panics become hanabi::PanicError, ExprError / ShaderGenerateError become the
// exceptions of the same name. Render-only modifiers are kept only for the attributes
// they add to the particle layout (src/modifier/output.rs); rendering is out of scope.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/hanabi_amd.h"

namespace hanabi {

// ---- errors -------------------------------------------------------------------------------
struct PanicError : std::logic_error { using std::logic_error::logic_error; };
struct ExprError : std::runtime_error {
    enum Kind { TypeError, SyntaxError, GraphEvalError, PropertyError, InvalidExprHandleError, InvalidModifierContext };
    Kind kind;
    ExprError(Kind k, const std::string& msg) : std::runtime_error(msg), kind(k) {}
};
struct ShaderGenerateError : std::runtime_error { using std::runtime_error::runtime_error; };
struct RonError : std::runtime_error { using std::runtime_error::runtime_error; };   // ron::Error of EffectAsset::deserialize
struct SpawnerSettingsError : std::runtime_error {
    enum Kind { InvalidPeriod, InfinitePeriod };
    Kind kind;
    float min = 0, max = 0;
    SpawnerSettingsError(Kind k, float mn, float mx, const std::string& msg) : std::runtime_error(msg), kind(k), min(mn), max(mx) {}
};

// ---- value types (src/graph/mod.rs) ----------------------------------------------------------
enum class ScalarType : uint8_t { Bool = HNB_BOOL, Float = HNB_F32, Int = HNB_I32, Uint = HNB_U32 };

struct ValueType {
    ScalarType elem = ScalarType::Float;
    uint8_t count = 1;  // 1 = scalar, 2..4 = vector
    constexpr ValueType() = default;
    constexpr ValueType(ScalarType e, uint8_t c = 1) : elem(e), count(c) {}
    bool is_scalar() const { return count == 1; }
    bool is_vector() const { return count > 1; }
    bool is_float() const { return elem == ScalarType::Float; }
    bool operator==(const ValueType& o) const { return elem == o.elem && count == o.count; }
    bool operator!=(const ValueType& o) const { return !(*this == o); }
    std::string to_string() const;
};
namespace VectorType {
constexpr ValueType VEC2B{ScalarType::Bool, 2}, VEC3B{ScalarType::Bool, 3}, VEC4B{ScalarType::Bool, 4};
constexpr ValueType VEC2F{ScalarType::Float, 2}, VEC3F{ScalarType::Float, 3}, VEC4F{ScalarType::Float, 4};
constexpr ValueType VEC2I{ScalarType::Int, 2}, VEC3I{ScalarType::Int, 3}, VEC4I{ScalarType::Int, 4};
constexpr ValueType VEC2U{ScalarType::Uint, 2}, VEC3U{ScalarType::Uint, 3}, VEC4U{ScalarType::Uint, 4};
}  // namespace VectorType

using Vec2 = std::array<float, 2>;
using Vec3 = std::array<float, 3>;
using Vec4 = std::array<float, 4>;

struct Value {
    ValueType type;
    uint32_t bits[4] = {0, 0, 0, 0};
    Value() = default;
    Value(float v) : type(ScalarType::Float) { set_f(0, v); }
    Value(double v) : Value((float)v) {}
    Value(int32_t v) : type(ScalarType::Int) { bits[0] = (uint32_t)v; }
    Value(uint32_t v) : type(ScalarType::Uint) { bits[0] = v; }
    Value(bool v) : type(ScalarType::Bool) { bits[0] = v ? 1u : 0u; }
    Value(Vec2 v) : type(VectorType::VEC2F) { for (int i = 0; i < 2; ++i) set_f(i, v[i]); }
    Value(Vec3 v) : type(VectorType::VEC3F) { for (int i = 0; i < 3; ++i) set_f(i, v[i]); }
    Value(Vec4 v) : type(VectorType::VEC4F) { for (int i = 0; i < 4; ++i) set_f(i, v[i]); }
    static Value from_bits(ValueType t, const uint32_t* b) { Value v; v.type = t; for (int i = 0; i < t.count; ++i) v.bits[i] = b[i]; return v; }
    static Value vec_i(std::initializer_list<int32_t> l) { Value v; v.type = ValueType(ScalarType::Int, (uint8_t)l.size()); int i = 0; for (int32_t x : l) v.bits[i++] = (uint32_t)x; return v; }
    static Value vec_u(std::initializer_list<uint32_t> l) { Value v; v.type = ValueType(ScalarType::Uint, (uint8_t)l.size()); int i = 0; for (uint32_t x : l) v.bits[i++] = x; return v; }
    static Value vec_b(std::initializer_list<bool> l) { Value v; v.type = ValueType(ScalarType::Bool, (uint8_t)l.size()); int i = 0; for (bool x : l) v.bits[i++] = x ? 1u : 0u; return v; }
    void set_f(int i, float f) { std::memcpy(&bits[i], &f, 4); }
    float get_f(int i) const { float f; std::memcpy(&f, &bits[i], 4); return f; }
    ValueType value_type() const { return type; }
    // Value::as_bytes (src/graph/mod.rs): the components, little-endian, no padding
    std::vector<uint8_t> as_bytes() const { std::vector<uint8_t> b(4u * type.count); std::memcpy(b.data(), bits, b.size()); return b; }
};

// ---- attributes ----------------------------------------------------------------------------------
struct Attribute {
    HnbAttr id = HNB_ATTR_POSITION;
    constexpr Attribute() = default;
    constexpr explicit Attribute(HnbAttr a) : id(a) {}
    bool operator==(const Attribute& o) const { return id == o.id; }
    bool operator!=(const Attribute& o) const { return id != o.id; }
    const char* name() const;
    ValueType value_type() const;
    Value default_value() const;
    uint32_t size() const { return value_type().count * 4u; }
    bool is_pseudo() const { return id == HNB_ATTR_ID || id == HNB_ATTR_PARTICLE_COUNTER; }
    static bool from_name(const std::string& name, Attribute* out);
    static const std::vector<Attribute>& all();

    static const Attribute ID, PARTICLE_COUNTER, POSITION, VELOCITY, AGE, LIFETIME, COLOR, HDR_COLOR, ALPHA, SIZE, SIZE2, SIZE3,
        PREV, NEXT, AXIS_X, AXIS_Y, AXIS_Z, SPRITE_INDEX, F32_0, F32_1, F32_2, F32_3, F32X2_0, F32X2_1, F32X2_2, F32X2_3,
        F32X3_0, F32X3_1, F32X3_2, F32X3_3, F32X4_0, F32X4_1, F32X4_2, F32X4_3, U32_0, U32_1, U32_2, U32_3, RIBBON_ID;
};

// ---- ParticleLayout: the reference's interleaved (AoS) particle struct (src/attributes.rs:1479-1890) --------
// This engine stores one packed plane per attribute instead, so the layout below drives nothing on the GPU; it is
// mirrored because it is part of the authoring API (EffectAsset::particle_layout) and because its sizes are the
// "bytes per particle" of the reference that DESIGN.md compares against.
struct AttributeLayout {
    Attribute attribute;
    uint32_t offset = 0;
    bool padding = false;  // one of the reference's PAD0..PAD4 filler fields
};
class ParticleLayout {
   public:
    class Builder {
       public:
        Builder& append(Attribute a) { attrs_.push_back(a); return *this; }
        // WGSL struct packing as ParticleLayoutBuilder::build (attributes.rs:1516-1670): duplicates dropped, vec4 first,
        // then {vec3 + scalar} pairs, {vec2 + vec2} pairs, padded vec3, the odd vec2, scalars; struct padded to its alignment.
        ParticleLayout build() const;
       private:
        std::vector<Attribute> attrs_;
    };
    static Builder make() { return Builder(); }                 // ParticleLayout::new()
    static ParticleLayout empty() { return ParticleLayout(); }
    static ParticleLayout default_layout();                     // { position, age, velocity, lifetime }
    bool is_empty() const { return unpadded_len_ == 0; }
    uint32_t len() const { return unpadded_len_; }              // attributes, padding fields not counted
    uint32_t size() const;                                      // bytes, incl. padding
    uint32_t align() const { return align_; }
    uint32_t min_binding_size() const { return (size() + 15u) / 16u * 16u; }  // the reference's particle stride
    bool contains(Attribute a) const;
    bool byte_offset(Attribute a, uint32_t* out) const;
    const std::vector<AttributeLayout>& entries() const { return layout_; }
    ParticleLayout merged_with(const std::vector<Attribute>& more) const;

   private:
    std::vector<AttributeLayout> layout_;
    uint32_t align_ = 4, unpadded_len_ = 0;
};

// ---- expressions -----------------------------------------------------------------------------------
struct ExprHandle {
    uint32_t id = 0;  // 1-based, serialised "#<id>" in the reference (expr.rs:132-213)
    uint32_t index() const { return id - 1; }
    bool valid() const { return id != 0; }
    bool operator==(const ExprHandle& o) const { return id == o.id; }
    // The serialised form of the reference, "#<id>" (expr.rs:159-166), and its inverse with the same rejections
    // (expr.rs:182-200): not "#N", N not a u32, N == 0. Throws std::invalid_argument.
    std::string to_string() const { return "#" + std::to_string(id); }
    static ExprHandle parse(const std::string& s);
};
struct PropertyHandle {
    uint32_t id = 0;
    uint32_t index() const { return id - 1; }
};

enum class BuiltInOperator : uint8_t { Time, DeltaTime, VirtualTime, VirtualDeltaTime, RealTime, RealDeltaTime, Rand, AlphaCutoff, IsAlive };
enum class UnaryOperator : uint8_t {
    Abs, Acos, Asin, Atan, All, Any, Ceil, Cos, Exp, Exp2, Floor, Fract, InvSqrt, Length, Log, Log2, Normalize, Pack4x8snorm,
    Pack4x8unorm, Round, Saturate, Sign, Sin, Sqrt, Tan, Unpack4x8snorm, Unpack4x8unorm, W, X, Y, Z
};
enum class BinaryOperator : uint8_t {
    Add, Atan2, Cross, Distance, Div, Dot, GreaterThan, GreaterThanOrEqual, LessThan, LessThanOrEqual, Max, Min, Mul, Remainder,
    Step, Sub, UniformRand, NormalRand, Vec2, Vec4XyzW
};
enum class TernaryOperator : uint8_t { Mix, Clamp, SmoothStep, Vec3 };

struct Expr {
    enum class Kind : uint8_t { BuiltIn, Literal, Property, Attribute, ParentAttribute, Unary, Binary, Ternary, Cast, TextureSample };
    Kind kind = Kind::Literal;
    BuiltInOperator builtin = BuiltInOperator::Time;
    ValueType rand_type;          // BuiltIn Rand(T); Cast target
    Value literal;                // Literal
    PropertyHandle property;      // Property
    Attribute attribute;          // Attribute / ParentAttribute
    UnaryOperator unary = UnaryOperator::Abs;
    BinaryOperator binary = BinaryOperator::Add;
    TernaryOperator ternary = TernaryOperator::Mix;
    ExprHandle a, b, c;           // operands (Unary: a; Binary: a,b; Ternary: a,b,c; Cast: a)

    // `Expr::value_type()` of the reference: only known for leaves and casts (expr.rs:1084-1098).
    bool value_type(ValueType* out) const;
    bool has_side_effect() const {
        return (kind == Kind::BuiltIn && builtin == BuiltInOperator::Rand) ||
               (kind == Kind::Binary && (binary == BinaryOperator::UniformRand || binary == BinaryOperator::NormalRand));
    }
};

struct Property {
    std::string name;
    Value default_value;
};

// The byte layout of an effect's property block in the reference (`PropertyLayout`, src/properties.rs:521-842): WGSL
// struct packing - vec4 first, {vec3 + scalar} pairs, {vec2 + vec2} pairs, then padded vec3 and the odd vec2, or the
// odd vec2 and the remaining scalars. A host that uploads property bytes the reference's way (EffectProperties::serialize,
// properties.rs:437-453) needs these offsets; this library itself sets properties by name (hnb_effect_set_property).
class PropertyLayout {
   public:
    struct Entry { Property property; uint32_t offset; };
    PropertyLayout() = default;                                  // PropertyLayout::empty()
    explicit PropertyLayout(const std::vector<Property>& properties);
    bool is_empty() const { return layout_.empty(); }
    uint32_t cpu_size() const;                                   // offset + size of the last entry, no tail padding
    uint32_t align() const;                                      // largest WGSL alignment of a member, 0 when empty
    uint32_t min_binding_size() const;                           // cpu_size rounded up to align; PanicError when empty
    bool contains(const std::string& name) const;
    bool offset(const std::string& name, uint32_t* out) const;
    const std::vector<Entry>& properties() const { return layout_; }
    std::string generate_property_struct_code() const;           // "" when empty (the reference returns None)
    // EffectProperties::serialize (properties.rs:437-453): the values, at their offsets, in cpu_size() bytes
    std::vector<uint8_t> serialize(const std::vector<Property>& values) const;

   private:
    std::vector<Entry> layout_;
};

// Per-instance property values (`EffectProperties`, src/properties.rs:200-453): an ordered list of (definition, value);
// a value may be set before the asset's properties are known, a definition's type never changes (PanicError, like the
// reference's assert), `update` reconciles the list with an asset's properties the way the component does when the
// asset (re)loads. The values reach the GPU through hnb_effect_set_property (by name) or, laid out the reference's
// way, through serialize().
class EffectProperties {
   public:
    struct Instance { Property def; Value value; };
    EffectProperties& with_properties(const std::vector<std::pair<std::string, Value>>& properties);
    const std::vector<Instance>& properties() const { return properties_; }
    bool get_stored(const std::string& name, Value* out) const;
    void set(const std::string& name, const Value& value);
    bool set_if_changed(const std::string& name, const Value& value);   // true when the list was modified
    void update(const std::vector<Property>& asset_properties);
    std::vector<uint8_t> serialize(const PropertyLayout& layout) const;

   private:
    std::vector<Instance> properties_;
};

class Module {
   public:
    ExprHandle add_expr(const Expr& e) { expressions_.push_back(e); return ExprHandle{(uint32_t)expressions_.size()}; }
    PropertyHandle add_property(const std::string& name, const Value& default_value);
    const Property* get_property(PropertyHandle h) const { return h.id >= 1 && h.index() < properties_.size() ? &properties_[h.index()] : nullptr; }
    bool get_property_by_name(const std::string& name, PropertyHandle* out) const;
    const std::vector<Property>& properties() const { return properties_; }
    const std::vector<Expr>& expressions() const { return expressions_; }
    const Expr* get(ExprHandle h) const { return h.valid() && h.index() < expressions_.size() ? &expressions_[h.index()] : nullptr; }
    const Expr& try_get(ExprHandle h) const;

    ExprHandle lit(const Value& v) { Expr e; e.kind = Expr::Kind::Literal; e.literal = v; return add_expr(e); }
    ExprHandle attr(Attribute a) { Expr e; e.kind = Expr::Kind::Attribute; e.attribute = a; return add_expr(e); }
    ExprHandle parent_attr(Attribute a) { Expr e; e.kind = Expr::Kind::ParentAttribute; e.attribute = a; return add_expr(e); }
    ExprHandle prop(PropertyHandle p) { Expr e; e.kind = Expr::Kind::Property; e.property = p; return add_expr(e); }
    ExprHandle builtin(BuiltInOperator op, ValueType rand_type = ValueType()) {
        Expr e; e.kind = Expr::Kind::BuiltIn; e.builtin = op; e.rand_type = rand_type; return add_expr(e);
    }
    ExprHandle unary(UnaryOperator op, ExprHandle inner) { check(inner); Expr e; e.kind = Expr::Kind::Unary; e.unary = op; e.a = inner; return add_expr(e); }
    ExprHandle binary(BinaryOperator op, ExprHandle l, ExprHandle r) {
        check(l); check(r); Expr e; e.kind = Expr::Kind::Binary; e.binary = op; e.a = l; e.b = r; return add_expr(e);
    }
    ExprHandle ternary(TernaryOperator op, ExprHandle x, ExprHandle y, ExprHandle z) {
        check(x); check(y); check(z); Expr e; e.kind = Expr::Kind::Ternary; e.ternary = op; e.a = x; e.b = y; e.c = z; return add_expr(e);
    }
    ExprHandle cast(ExprHandle inner, ValueType target);
    bool is_const(ExprHandle h) const;
    bool has_side_effect(ExprHandle h) const { return try_get(h).has_side_effect(); }

#define HNB_UN(fn, OP) ExprHandle fn(ExprHandle x) { return unary(UnaryOperator::OP, x); }
    HNB_UN(abs, Abs) HNB_UN(acos, Acos) HNB_UN(asin, Asin) HNB_UN(atan, Atan) HNB_UN(all, All) HNB_UN(any, Any) HNB_UN(ceil, Ceil)
    HNB_UN(cos, Cos) HNB_UN(exp, Exp) HNB_UN(exp2, Exp2) HNB_UN(floor, Floor) HNB_UN(fract, Fract) HNB_UN(inverse_sqrt, InvSqrt)
    HNB_UN(length, Length) HNB_UN(log, Log) HNB_UN(log2, Log2) HNB_UN(normalize, Normalize) HNB_UN(pack4x8snorm, Pack4x8snorm)
    HNB_UN(pack4x8unorm, Pack4x8unorm) HNB_UN(round, Round) HNB_UN(saturate, Saturate) HNB_UN(sign, Sign) HNB_UN(sin, Sin)
    HNB_UN(sqrt, Sqrt) HNB_UN(tan, Tan) HNB_UN(unpack4x8snorm, Unpack4x8snorm) HNB_UN(unpack4x8unorm, Unpack4x8unorm)
    HNB_UN(w, W) HNB_UN(x, X) HNB_UN(y, Y) HNB_UN(z, Z)
#undef HNB_UN
#define HNB_BIN(fn, OP) ExprHandle fn(ExprHandle l, ExprHandle r) { return binary(BinaryOperator::OP, l, r); }
    HNB_BIN(add, Add) HNB_BIN(atan2, Atan2) HNB_BIN(cross, Cross) HNB_BIN(distance, Distance) HNB_BIN(div, Div) HNB_BIN(dot, Dot)
    HNB_BIN(ge, GreaterThanOrEqual) HNB_BIN(gt, GreaterThan) HNB_BIN(le, LessThanOrEqual) HNB_BIN(lt, LessThan) HNB_BIN(max, Max)
    HNB_BIN(min, Min) HNB_BIN(mul, Mul) HNB_BIN(rem, Remainder) HNB_BIN(step, Step) HNB_BIN(sub, Sub) HNB_BIN(uniform, UniformRand)
    HNB_BIN(normal, NormalRand) HNB_BIN(vec2, Vec2) HNB_BIN(vec4_xyz_w, Vec4XyzW)
#undef HNB_BIN
    ExprHandle mix(ExprHandle a, ExprHandle b, ExprHandle t) { return ternary(TernaryOperator::Mix, a, b, t); }
    ExprHandle clamp(ExprHandle x, ExprHandle lo, ExprHandle hi) { return ternary(TernaryOperator::Clamp, x, lo, hi); }
    ExprHandle smoothstep(ExprHandle lo, ExprHandle hi, ExprHandle x) { return ternary(TernaryOperator::SmoothStep, lo, hi, x); }
    ExprHandle vec3(ExprHandle x, ExprHandle y, ExprHandle z) { return ternary(TernaryOperator::Vec3, x, y, z); }

   private:
    void check(ExprHandle h) const { if (!h.valid() || h.index() >= expressions_.size()) throw PanicError("expression handle out of range for this module"); }
    std::vector<Expr> expressions_;
    std::vector<Property> properties_;
};

class ExprWriter;
// Fluent expression builder (expr.rs:2643-4128).
class WriterExpr {
   public:
    ExprHandle expr() const { return handle_; }
#define HNB_WUN(fn, OP) WriterExpr fn() const { return un(UnaryOperator::OP); }
    HNB_WUN(abs, Abs) HNB_WUN(all, All) HNB_WUN(any, Any) HNB_WUN(acos, Acos) HNB_WUN(asin, Asin) HNB_WUN(atan, Atan) HNB_WUN(ceil, Ceil)
    HNB_WUN(cos, Cos) HNB_WUN(exp, Exp) HNB_WUN(exp2, Exp2) HNB_WUN(floor, Floor) HNB_WUN(fract, Fract) HNB_WUN(inverse_sqrt, InvSqrt)
    HNB_WUN(length, Length) HNB_WUN(log, Log) HNB_WUN(log2, Log2) HNB_WUN(normalized, Normalize) HNB_WUN(pack4x8snorm, Pack4x8snorm)
    HNB_WUN(pack4x8unorm, Pack4x8unorm) HNB_WUN(round, Round) HNB_WUN(sign, Sign) HNB_WUN(sin, Sin) HNB_WUN(sqrt, Sqrt)
    HNB_WUN(tan, Tan) HNB_WUN(unpack4x8snorm, Unpack4x8snorm) HNB_WUN(unpack4x8unorm, Unpack4x8unorm) HNB_WUN(saturate, Saturate)
    HNB_WUN(x, X) HNB_WUN(y, Y) HNB_WUN(z, Z) HNB_WUN(w, W)
#undef HNB_WUN
#define HNB_WBIN(fn, OP) WriterExpr fn(const WriterExpr& o) const { return bin(o, BinaryOperator::OP); }
    HNB_WBIN(add, Add) HNB_WBIN(atan2, Atan2) HNB_WBIN(cross, Cross) HNB_WBIN(dot, Dot) HNB_WBIN(distance, Distance) HNB_WBIN(div, Div)
    HNB_WBIN(ge, GreaterThanOrEqual) HNB_WBIN(gt, GreaterThan) HNB_WBIN(le, LessThanOrEqual) HNB_WBIN(lt, LessThan) HNB_WBIN(max, Max)
    HNB_WBIN(min, Min) HNB_WBIN(mul, Mul) HNB_WBIN(normal, NormalRand) HNB_WBIN(rem, Remainder) HNB_WBIN(sub, Sub)
    HNB_WBIN(uniform, UniformRand) HNB_WBIN(vec2, Vec2) HNB_WBIN(vec4_xyz_w, Vec4XyzW)
#undef HNB_WBIN
    // Note: order is step(edge, x) but x.step(edge) (expr.rs:3819-3822)
    WriterExpr step(const WriterExpr& edge) const { return edge.bin(*this, BinaryOperator::Step); }
    WriterExpr mix(const WriterExpr& other, const WriterExpr& fraction) const { return ter(other, fraction, TernaryOperator::Mix); }
    WriterExpr clamp(const WriterExpr& lo, const WriterExpr& hi) const { return ter(lo, hi, TernaryOperator::Clamp); }
    // Note: order is smoothstep(low, high, x) but x.smoothstep(low, high) (expr.rs:3985-3988)
    WriterExpr smoothstep(const WriterExpr& low, const WriterExpr& high) const { return low.ter(high, *this, TernaryOperator::SmoothStep); }
    WriterExpr vec3(const WriterExpr& y, const WriterExpr& z) const { return ter(y, z, TernaryOperator::Vec3); }
    WriterExpr cast(ValueType target) const;
    WriterExpr operator+(const WriterExpr& o) const { return add(o); }
    WriterExpr operator-(const WriterExpr& o) const { return sub(o); }
    WriterExpr operator*(const WriterExpr& o) const { return mul(o); }
    WriterExpr operator/(const WriterExpr& o) const { return div(o); }
    WriterExpr operator%(const WriterExpr& o) const { return rem(o); }

   private:
    friend class ExprWriter;
    WriterExpr(ExprHandle h, std::shared_ptr<Module> m) : handle_(h), module_(std::move(m)) {}
    WriterExpr un(UnaryOperator op) const { return WriterExpr(module_->unary(op, handle_), module_); }
    WriterExpr bin(const WriterExpr& o, BinaryOperator op) const {
        if (module_.get() != o.module_.get()) throw PanicError("expressions belong to different modules");
        return WriterExpr(module_->binary(op, handle_, o.handle_), module_);
    }
    WriterExpr ter(const WriterExpr& y, const WriterExpr& z, TernaryOperator op) const {
        if (module_.get() != y.module_.get() || module_.get() != z.module_.get()) throw PanicError("expressions belong to different modules");
        return WriterExpr(module_->ternary(op, handle_, y.handle_, z.handle_), module_);
    }
    ExprHandle handle_;
    std::shared_ptr<Module> module_;
};

class ExprWriter {
   public:
    ExprWriter() : module_(std::make_shared<Module>()) {}
    explicit ExprWriter(std::shared_ptr<Module> m) : module_(std::move(m)) {}
    PropertyHandle add_property(const std::string& name, const Value& default_value) { return module_->add_property(name, default_value); }
    WriterExpr push(const Expr& e) { return WriterExpr(module_->add_expr(e), module_); }
    WriterExpr lit(const Value& v) { return WriterExpr(module_->lit(v), module_); }
    WriterExpr attr(Attribute a) { return WriterExpr(module_->attr(a), module_); }
    WriterExpr parent_attr(Attribute a) { return WriterExpr(module_->parent_attr(a), module_); }
    WriterExpr prop(PropertyHandle h) { return WriterExpr(module_->prop(h), module_); }
    WriterExpr time() { return WriterExpr(module_->builtin(BuiltInOperator::Time), module_); }
    WriterExpr delta_time() { return WriterExpr(module_->builtin(BuiltInOperator::DeltaTime), module_); }
    WriterExpr rand(ValueType t) { return WriterExpr(module_->builtin(BuiltInOperator::Rand, t), module_); }
    WriterExpr alpha_cutoff() { return WriterExpr(module_->builtin(BuiltInOperator::AlphaCutoff), module_); }
    Module finish() const { return *module_; }
    std::shared_ptr<Module> module() const { return module_; }

   private:
    std::shared_ptr<Module> module_;
};

// ---- modifiers ---------------------------------------------------------------------------------------
enum ModifierContext : uint32_t { CONTEXT_INIT = 1, CONTEXT_UPDATE = 2, CONTEXT_RENDER = 4 };
enum class ShapeDimension : uint8_t { Surface, Volume };
enum class OrientMode : uint8_t { ParallelCameraDepthPlane, FaceCameraPosition, AlongVelocity };
enum class EventEmitCondition : uint8_t { Always, OnDie };

struct Modifier {
    enum class Kind : uint32_t {
        SetAttribute = 1, InheritAttribute, SetPositionCircle, SetPositionSphere, SetPositionCone3d, SetVelocityCircle,
        SetVelocitySphere, SetVelocityTangent, Accel, RadialAccel, TangentAccel, LinearDrag, ConformToSphere, KillSphere,
        KillAabb, EmitSpawnEvent, Render
    };
    Kind kind = Kind::SetAttribute;
    Attribute attribute;               // SetAttribute / InheritAttribute
    ExprHandle e[7];                   // expression fields, meaning per kind (see factory functions)
    bool has_shell = false, has_sticky = false;  // ConformToSphere optional fields
    ShapeDimension dimension = ShapeDimension::Surface;
    bool kill_inside = false;
    EventEmitCondition condition = EventEmitCondition::Always;
    uint32_t child_index = 0;
    std::vector<Attribute> render_attributes;  // Render
    std::string render_name;

    uint32_t context() const;
    std::vector<Attribute> attributes() const;
    // KillSphereModifier / KillAabbModifier::with_kill_inside (kill.rs:57-60, 137-140)
    Modifier with_kill_inside(bool inside) const { Modifier m = *this; m.kill_inside = inside; return m; }
};

// Factory functions named after the reference's modifier types.
Modifier SetAttributeModifier(Attribute attribute, ExprHandle value);
Modifier InheritAttributeModifier(Attribute attribute);
Modifier SetPositionCircleModifier(ExprHandle center, ExprHandle axis, ExprHandle radius, ShapeDimension dimension);
Modifier SetPositionSphereModifier(ExprHandle center, ExprHandle radius, ShapeDimension dimension);
Modifier SetPositionCone3dModifier(ExprHandle height, ExprHandle base_radius, ExprHandle top_radius, ShapeDimension dimension);
Modifier SetVelocityCircleModifier(ExprHandle center, ExprHandle axis, ExprHandle speed);
Modifier SetVelocitySphereModifier(ExprHandle center, ExprHandle speed);
Modifier SetVelocityTangentModifier(ExprHandle origin, ExprHandle axis, ExprHandle speed);
Modifier AccelModifier(ExprHandle accel);
Modifier RadialAccelModifier(ExprHandle origin, ExprHandle accel);
Modifier TangentAccelModifier(ExprHandle origin, ExprHandle axis, ExprHandle accel);
Modifier LinearDragModifier(ExprHandle drag);
Modifier ConformToSphereModifier(ExprHandle origin, ExprHandle radius, ExprHandle influence_dist, ExprHandle attraction_accel,
                                 ExprHandle max_attraction_speed, ExprHandle shell_half_thickness = ExprHandle{},
                                 ExprHandle sticky_factor = ExprHandle{});
Modifier KillSphereModifier(ExprHandle center, ExprHandle sqr_radius, bool kill_inside = false);
Modifier KillAabbModifier(ExprHandle center, ExprHandle half_size, bool kill_inside = false);
Modifier EmitSpawnEventModifier(EventEmitCondition condition, ExprHandle count, uint32_t child_index);
// The reference's helper constructors (`AccelModifier::constant(&mut module, v)`, `::via_property(&mut module, ..., property)`;
// accel.rs:52-64, 133-147, 245-266, force.rs:264-268): they add the literal / property expressions to the module themselves.
struct Vec3Lit { float x, y, z; };
Modifier AccelModifierConstant(Module& module, Vec3Lit acceleration);
Modifier AccelModifierViaProperty(Module& module, PropertyHandle property);
Modifier RadialAccelModifierConstant(Module& module, Vec3Lit origin, float acceleration);
Modifier RadialAccelModifierViaProperty(Module& module, Vec3Lit origin, PropertyHandle property);
Modifier TangentAccelModifierConstant(Module& module, Vec3Lit origin, Vec3Lit axis, float acceleration);
Modifier TangentAccelModifierViaProperty(Module& module, Vec3Lit origin, Vec3Lit axis, PropertyHandle property);
Modifier LinearDragModifierConstant(Module& module, float drag);
// Render-only modifiers: only their attribute requirements matter here (modifier/output.rs).
Modifier RenderModifier(const std::string& name, const std::vector<Attribute>& attributes);
Modifier ColorOverLifetimeModifier();
Modifier SizeOverLifetimeModifier();
Modifier SetColorModifier();
Modifier SetSizeModifier();
Modifier OrientModifier(OrientMode mode);
Modifier FlipbookModifier();
Modifier ScreenSpaceSizeModifier();
Modifier RoundModifier();
Modifier ParticleTextureModifier();

// ---- spawner (src/spawn.rs) -------------------------------------------------------------------------------
// PCG-XSH-RR 64/32 (rand_pcg::Pcg32, third-party; published algorithm, parity unpinned).
struct Pcg32 {
    uint64_t state = 0, inc = 0;
    Pcg32() : Pcg32(0xcafef00dd15ea5e5ull, 0xa02bdbf7bb3c0a7ull) {}
    Pcg32(uint64_t seed_state, uint64_t stream);
    uint32_t next_u32();
};

// Per-frame PRNG seed evolution of the reference (`compile_effects`, src/lib.rs:1813-1820): every frame an effect was not
// recompiled, `prng_seed = StdRng::seed_from_u64(prng_seed as u64).random::<u32>()`. Third-party code (rand 0.10, absent from
// /root/reference): restated from the published algorithms of the rand family - `SeedableRng::seed_from_u64` expands the u64
// with PCG32 (XSH-RR, multiplier 6364136223846793005, increment 11634580027462260723) into the 32-byte seed, `StdRng` is
// ChaCha with 12 rounds, key = seed, 64-bit block counter 0, stream 0, and `random::<u32>()` is the first output word.
// Pinned on the ChaCha vectors of RFC 7539 and eSTREAM and on rand's own `StdRng` value-stability test (tests/test_seed.py);
// whether rand 0.10 changed `StdRng` cannot be checked here, so the reference-exact sequence is "parity unpinned".
void chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int rounds, uint32_t out[16]);
void seed_from_u64(uint64_t state, uint8_t out_seed[32]);
uint32_t next_prng_seed(uint32_t prng_seed);

struct CpuValue {
    bool is_uniform = false;
    float a = 0.0f, b = 0.0f;
    CpuValue() = default;
    CpuValue(float single) : is_uniform(false), a(single), b(single) {}
    CpuValue(float lo, float hi) : is_uniform(true), a(lo), b(hi) {}
    static CpuValue Single(float v) { return CpuValue(v); }
    static CpuValue Uniform(float lo, float hi) { return CpuValue(lo, hi); }
    float sample(Pcg32& rng) const;
    std::array<float, 2> range() const;
};

class SpawnerSettings {
   public:
    SpawnerSettings() : SpawnerSettings(once(CpuValue(1.0f))) {}
    static SpawnerSettings make(CpuValue count, CpuValue spawn_duration, CpuValue period, uint32_t cycle_count);      // `new`
    static SpawnerSettings try_make(CpuValue count, CpuValue spawn_duration, CpuValue period, uint32_t cycle_count);  // `try_new`
    static SpawnerSettings once(CpuValue count) { return make(count, CpuValue(0.0f), CpuValue(0.0f), 1); }
    static SpawnerSettings rate(CpuValue rate) { return make(rate, CpuValue(1.0f), CpuValue(1.0f), 0); }
    static SpawnerSettings burst(CpuValue count, CpuValue period) { return make(count, CpuValue(0.0f), period, 0); }
    bool is_once() const { return cycle_count_ == 1; }
    bool is_forever() const { return cycle_count_ == 0; }
    SpawnerSettings with_emit_on_start(bool v) const { SpawnerSettings s = *this; s.emit_on_start_ = v; return s; }
    void set_emit_on_start(bool v) { emit_on_start_ = v; }
    bool emits_on_start() const { return emit_on_start_; }
    SpawnerSettings with_count(CpuValue v) const { SpawnerSettings s = *this; s.count_ = v; return s; }
    void set_count(CpuValue v) { count_ = v; }
    CpuValue count() const { return count_; }
    SpawnerSettings with_spawn_duration(CpuValue v) const { SpawnerSettings s = *this; s.spawn_duration_ = v; return s; }
    void set_spawn_duration(CpuValue v) { spawn_duration_ = v; }
    CpuValue spawn_duration() const { return spawn_duration_; }
    SpawnerSettings with_period(CpuValue v) const { SpawnerSettings s = *this; s.set_period(v); return s; }
    void set_period(CpuValue v);
    CpuValue period() const { return period_; }
    SpawnerSettings with_cycle_count(uint32_t v) const { SpawnerSettings s = *this; s.cycle_count_ = v; return s; }
    void set_cycle_count(uint32_t v) { cycle_count_ = v; }
    uint32_t cycle_count() const { return cycle_count_; }
    SpawnerSettings with_starts_active(bool v) const { SpawnerSettings s = *this; s.starts_active_ = v; return s; }
    void set_starts_active(bool v) { starts_active_ = v; }
    bool starts_active() const { return starts_active_; }

   private:
    SpawnerSettings(CpuValue c, CpuValue d, CpuValue p, uint32_t n) : count_(c), spawn_duration_(d), period_(p), cycle_count_(n) {}
    CpuValue count_, spawn_duration_, period_;
    uint32_t cycle_count_ = 1;
    bool starts_active_ = true, emit_on_start_ = true;
};

class EffectSpawner {
   public:
    SpawnerSettings settings;
    uint32_t spawn_count = 0;
    bool active = true;
    EffectSpawner() : EffectSpawner(SpawnerSettings()) {}
    explicit EffectSpawner(const SpawnerSettings& s);
    EffectSpawner with_active(bool a) const { EffectSpawner e = *this; e.active = a; return e; }
    float cycle_time() const { return cycle_time_; }
    float cycle_spawn_duration() const { return sampled_spawn_duration_; }
    float cycle_period() const { return settings.is_once() ? 0.0f : sampled_period_; }
    float cycle_ratio() const { return settings.is_once() ? 0.0f : cycle_time_ / sampled_period_; }
    float cycle_spawn_count() const { return sampled_count_; }
    uint32_t completed_cycle_count() const { return completed_cycle_count_; }
    bool has_completed() const { return !settings.is_forever() && completed_cycle_count_ >= settings.cycle_count(); }
    void reset();
    uint32_t tick(float dt, Pcg32& rng);

   private:
    float cycle_time_ = 0, sampled_spawn_duration_ = 0, sampled_period_ = 0, sampled_count_ = 0, spawn_remainder_ = 0;
    uint32_t completed_cycle_count_ = 0;
};

// ---- asset (src/asset.rs) -------------------------------------------------------------------------------------
enum class SimulationSpace : uint8_t { Global, Local };
// ---- WGSL text of the reference (wgsl.cpp) ---------------------------------------------------------------------
// `ToWgslString` (src/lib.rs:259-430, src/graph/mod.rs:287-296,1003-1024)
std::string to_wgsl_string(float x);
std::string to_wgsl_string(const Value& v);

// The evaluation context of the reference (`ShaderWriter`, src/modifier/mod.rs:204-367), restricted to what
// `Expr::eval` needs: the WGSL text of an expression, memoised per handle, with side-effect expressions hoisted
// into `let varN = ...;` statements appended to `main_code`. Inspection / test output only: the simulation runs
// the program produced by lower(), not this text.
class ShaderWriter {
   public:
    explicit ShaderWriter(uint32_t modifier_context, bool attribute_pointer = false) : context_(modifier_context), attribute_pointer_(attribute_pointer) {}
    ShaderWriter with_attribute_pointer() const { ShaderWriter w = *this; w.attribute_pointer_ = true; return w; }
    uint32_t modifier_context() const { return context_; }
    bool is_attribute_pointer() const { return attribute_pointer_; }
    std::string eval(const Module& module, ExprHandle handle);
    std::string make_local_var();
    void push_stmt(const std::string& stmt) { main_code += stmt; main_code += "\n"; }   // modifier/mod.rs:325-328
    // `make_fn` (modifier/mod.rs:330-362): the body is generated in a fresh writer whose particle is a pointer; its hoisted `let varN`
    // statements precede the body inside the function, which is appended to extra_code.
    template <class F>
    void make_fn(const std::string& func_name, const std::string& args, Module& module, F&& body_of) {
        ShaderWriter ctx = ShaderWriter(context_).with_attribute_pointer();
        const std::string body = body_of(module, ctx);
        extra_code += ctx.extra_code;
        extra_code += "fn " + func_name + "(" + args + ") {\n" + ctx.main_code + body + "}";
    }
    void set_emits_gpu_spawn_events(bool use_events);   // modifier/mod.rs:262-281
    int emits_gpu_spawn_events() const { return emits_events_; }   // -1: no modifier said
    std::string main_code;
    std::string extra_code;   // functions emitted at shader top level, called from main_code

   private:
    std::string hoist_if_side_effect(const std::string& code, bool side_effect);
    uint32_t context_;
    bool attribute_pointer_;
    uint32_t var_counter_ = 0;
    int emits_events_ = -1;
    std::map<uint32_t, std::string> expr_cache_;
};

enum class SimulationCondition : uint8_t { WhenVisible, Always };
enum class MotionIntegration : uint8_t { None, PreUpdate, PostUpdate };

class EffectAsset {
   public:
    std::string name;
    SpawnerSettings spawner;
    float z_layer_2d = 0.0f;
    SimulationSpace simulation_space = SimulationSpace::Global;
    SimulationCondition simulation_condition = SimulationCondition::WhenVisible;
    uint32_t prng_seed = 0;
    MotionIntegration motion_integration = MotionIntegration::PostUpdate;

    EffectAsset() = default;
    EffectAsset(uint32_t capacity, const SpawnerSettings& s, const Module& m) : spawner(s), capacity_(capacity), module_(m) {}
    uint32_t capacity() const { return capacity_; }
    const Module& module() const { return module_; }
    EffectAsset& with_name(const std::string& n) { name = n; return *this; }
    EffectAsset& with_simulation_condition(SimulationCondition c) { simulation_condition = c; return *this; }
    EffectAsset& with_simulation_space(SimulationSpace s) { simulation_space = s; return *this; }
    EffectAsset& with_motion_integration(MotionIntegration m) { motion_integration = m; return *this; }
    const std::vector<Property>& properties() const { return module_.properties(); }
    EffectAsset& init(const Modifier& m);
    EffectAsset& update(const Modifier& m);
    EffectAsset& render(const Modifier& m);
    EffectAsset& add_modifier(uint32_t context, const Modifier& m);
    const std::vector<Modifier>& init_modifiers() const { return init_; }
    const std::vector<Modifier>& update_modifiers() const { return update_; }
    const std::vector<Modifier>& render_modifiers() const { return render_; }
    // Union of the modifiers' attributes and of every Expr::Attribute in the module
    // (asset.rs:605-624), in ascending attribute-id order.
    std::vector<Attribute> particle_layout() const;
    // The same set as the reference's interleaved struct (sizes / offsets the reference's buffers would have).
    ParticleLayout reference_particle_layout() const;
    PropertyLayout property_layout() const { return PropertyLayout(module_.properties()); }  // asset.rs: EffectAsset::property_layout()

   private:
    uint32_t capacity_ = 0;
    Module module_;
    std::vector<Modifier> init_, update_, render_;
};

// The simulation side of `EffectShaderSources::generate` (src/lib.rs:1026-1302) as WGSL TEXT: what the reference pastes into
// vfx_init.wgsl ({{INIT_CODE}}, {{INIT_EXTRA}}, {{SIMULATION_SPACE_TRANSFORM_PARTICLE}}) and vfx_update.wgsl ({{AGE_CODE}}, {{REAP_CODE}},
// {{UPDATE_CODE}} incl. the Euler integration, {{UPDATE_EXTRA}}, {{WRITEBACK_CODE}}). This library does not execute it - lower() produces
// the program the HIP kernels run - but the text IS the reference's definition of the effect: tests/wgsl_eval interprets it with code that
// shares nothing with lowering.cpp or the oracle and compares the three. Function names carry a hash of the modifier's fields like the
// reference's (`calc_func_id`); the hash function itself (Rust's DefaultHasher) is not reproduced.
struct WgslSources {
    std::string init_code, init_extra, init_sim_space_transform;
    std::string age_code, reap_code, update_code, update_extra, writeback_code;
    bool consume_gpu_spawn_events = false, emit_gpu_spawn_events = false, read_parent_particle = false;
    std::vector<Attribute> attributes;   // the particle struct, in layout order
};
WgslSources generate_wgsl(const EffectAsset& asset, bool has_parent = false);
// `ToWgslString for CpuValue<f32>` (src/lib.rs:432-482): "1." / "(frand() * (2. - 1.) + 1.)"
std::string to_wgsl_string(const CpuValue& v);

// ---- lowering + serialisation ------------------------------------------------------------------------------------
// `ToWgslString for f32` (src/lib.rs:264-269): literals reach the GPU with 6 decimals.
float round_literal_f32(float x);
// EffectShaderSources::generate (src/lib.rs:805-1336) -> HnbProgram blob for hnb_program_create().
std::vector<uint8_t> lower(const EffectAsset& asset);
// Human-readable listing of a program blob (debugging / tests).
std::string disassemble(const std::vector<uint8_t>& blob);
// The reference's on-disk format (RON; EffectAsset::serialize / deserialize, src/asset.rs:674-716); see ron.cpp.
std::string to_ron(const EffectAsset& asset);
EffectAsset from_ron(const std::string& text);
// Flat authoring-level description of the asset (expressions, modifiers, settings): the
// input format of the CPU oracle under oracle/ and a first step towards an on-disk format.
std::vector<uint8_t> serialize_asset(const EffectAsset& asset);

}  // namespace hanabi
