// The reference's on-disk asset format: `EffectAsset::serialize` / `deserialize`
// (src/asset.rs:674-716), RON text produced by serde + bevy_reflect.
//
// No sample file and no golden string ships with the reference; what pins the format is its source:
//   * the field list and order of the `EffectAsset` struct                          src/asset.rs:716-750
//   * `ExprHandle` as the string "#<id>"                                            src/graph/expr.rs:132-213 (test :4831-4885)
//   * `Attribute` as its name                                                       src/attributes.rs:707-723 (test :2308-2321)
//   * modifiers as single-entry maps { "<type path>": ( fields ) } in a sequence    src/modifier/registry.rs:108-194 (test :308-391)
//   * the serde derives of Module / Expr / Value / Property / SpawnerSettings / CpuValue / ValueType
//     (src/graph/expr.rs:336-344,909-960,1268-1273,1324-1331,1399-1408,1443-1449,1741-1748; src/graph/mod.rs:90-100,438-443,
//     1192-1206,1479-1489; src/properties.rs:112-116; src/spawn.rs:80-84,217-253; src/attributes.rs:150-164,222-228,411-420)
//     under RON's rules: structs `(field: value, ...)` without a name, enum variants `Name`, `Name(value)` or
//     `Name(field: value)`, newtype structs transparent only where serde says so, Option as `Some(..)` / `None`,
//     glam vectors as tuples `(x, y, z)`, floats in Rust's `Display` form (shortest round-trip digits, never an exponent)
//     with `.0` appended to integral values.
// The reader is a general RON value parser (whitespace, comments, trailing commas, optional struct names, raw exponents in
// floats), so it does not depend on the writer's layout choices; the round-trip tests mirror the reference's
// (asset.rs:1303-1365, registry.rs:308-431). Render modifiers are written with their type path and no fields (the reference's
// deserializer applies the fields present onto a default instance) and read back as the attributes they add to the layout.
// Matrices and texture slots are outside the simulation path and rejected.
#include <cctype>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <sstream>

#include "hanabi.hpp"

namespace hanabi {

namespace {

// ---- generic RON value tree -------------------------------------------------------------------------------------------
struct Node {
    enum Kind { Struct, Seq, Map, String, Number, Ident } kind = Ident;
    std::string text;                                  // String: content; Number: literal text; Ident / Struct: name (may be empty)
    std::vector<std::pair<std::string, Node>> fields;  // Struct: named fields (key) or positional items (empty key)
    std::vector<Node> items;                           // Seq
    std::vector<std::pair<Node, Node>> entries;        // Map
    bool named_fields = false;

    const Node* field(const std::string& k) const {
        for (const auto& f : fields)
            if (f.first == k) return &f.second;
        return nullptr;
    }
};

struct Parser {
    const std::string& s;
    size_t i = 0;
    explicit Parser(const std::string& str) : s(str) {}
    [[noreturn]] void fail(const std::string& msg) const {
        size_t line = 1, col = 1;
        for (size_t k = 0; k < i && k < s.size(); ++k) { if (s[k] == '\n') { ++line; col = 1; } else ++col; }
        throw RonError("RON: " + msg + " at line " + std::to_string(line) + ", column " + std::to_string(col));
    }
    void ws() {
        for (;;) {
            while (i < s.size() && std::isspace((unsigned char)s[i])) ++i;
            if (i + 1 < s.size() && s[i] == '/' && s[i + 1] == '/') { while (i < s.size() && s[i] != '\n') ++i; continue; }
            if (i + 1 < s.size() && s[i] == '/' && s[i + 1] == '*') {
                int depth = 1; i += 2;
                while (i + 1 < s.size() && depth) { if (s[i] == '/' && s[i + 1] == '*') { ++depth; i += 2; } else if (s[i] == '*' && s[i + 1] == '/') { --depth; i += 2; } else ++i; }
                if (depth) fail("unterminated block comment");
                continue;
            }
            if (i < s.size() && s[i] == '#' && i + 1 < s.size() && s[i + 1] == '!') { while (i < s.size() && s[i] != '\n') ++i; continue; }  // #![enable(...)]
            break;
        }
    }
    bool eat(char c) { ws(); if (i < s.size() && s[i] == c) { ++i; return true; } return false; }
    void expect(char c) { if (!eat(c)) fail(std::string("expected '") + c + "'"); }
    static bool ident_start(char c) { return std::isalpha((unsigned char)c) || c == '_'; }
    static bool ident_char(char c) { return std::isalnum((unsigned char)c) || c == '_'; }

    std::string parse_string() {
        std::string out;
        ++i;  // opening quote
        while (i < s.size() && s[i] != '"') {
            char c = s[i++];
            if (c == '\\') {
                if (i >= s.size()) fail("unterminated escape");
                const char e = s[i++];
                switch (e) {
                    case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break; case '0': out += '\0'; break;
                    case '\\': out += '\\'; break; case '"': out += '"'; break; case '\'': out += '\''; break;
                    case 'u': {  // \u{XXXX}: ASCII only here
                        if (i >= s.size() || s[i] != '{') fail("bad unicode escape");
                        size_t e2 = s.find('}', i);
                        if (e2 == std::string::npos) fail("bad unicode escape");
                        const unsigned long cp = std::strtoul(s.substr(i + 1, e2 - i - 1).c_str(), nullptr, 16);
                        if (cp > 0x7f) fail("non-ASCII escape not supported");
                        out += (char)cp; i = e2 + 1;
                    } break;
                    default: fail("unknown escape");
                }
            } else out += c;
        }
        if (i >= s.size()) fail("unterminated string");
        ++i;
        return out;
    }

    // contents of ( ... ): named fields `a: v` or positional values
    void parse_parens(Node& n) {
        n.kind = Node::Struct;
        ws();
        bool first = true;
        while (!eat(')')) {
            if (!first) { /* separator consumed below */ }
            first = false;
            ws();
            // named field?
            size_t save = i;
            if (i < s.size() && ident_start(s[i])) {
                size_t j = i;
                while (j < s.size() && ident_char(s[j])) ++j;
                size_t k = j;
                while (k < s.size() && std::isspace((unsigned char)s[k])) ++k;
                if (k < s.size() && s[k] == ':' ) {
                    const std::string key = s.substr(i, j - i);
                    i = k + 1;
                    n.named_fields = true;
                    n.fields.emplace_back(key, parse_value());
                    if (!eat(',')) { expect(')'); break; }
                    continue;
                }
            }
            i = save;
            n.fields.emplace_back(std::string(), parse_value());
            if (!eat(',')) { expect(')'); break; }
        }
    }

    // Nesting limit: the `ron` crate refuses documents nested deeper than its recursion limit (128 by default) with an error;
    // without one, a buffer of a few hundred thousand '(' overflows the stack of this recursive-descent parser.
    static constexpr int kMaxDepth = 128;
    int depth = 0;
    struct DepthGuard {
        Parser& p;
        explicit DepthGuard(Parser& q) : p(q) { if (++p.depth > kMaxDepth) p.fail("exceeded the recursion limit (128 nested values)"); }
        ~DepthGuard() { --p.depth; }
    };

    Node parse_value() {
        DepthGuard guard(*this);
        ws();
        if (i >= s.size()) fail("unexpected end of input");
        Node n;
        const char c = s[i];
        if (c == '"') { n.kind = Node::String; n.text = parse_string(); return n; }
        if (c == '[') {
            ++i; n.kind = Node::Seq;
            while (!eat(']')) { n.items.push_back(parse_value()); if (!eat(',')) { expect(']'); break; } }
            return n;
        }
        if (c == '{') {
            ++i; n.kind = Node::Map;
            while (!eat('}')) {
                Node k = parse_value();
                expect(':');
                Node v = parse_value();
                n.entries.emplace_back(std::move(k), std::move(v));
                if (!eat(',')) { expect('}'); break; }
            }
            return n;
        }
        if (c == '(') { ++i; parse_parens(n); return n; }
        if (std::isdigit((unsigned char)c) || c == '-' || c == '+' || c == '.') {
            size_t j = i + 1;
            if ((c == '-' || c == '+') && j < s.size() && ident_start(s[j])) {  // -inf
                while (j < s.size() && ident_char(s[j])) ++j;
            } else {
                while (j < s.size() && (std::isalnum((unsigned char)s[j]) || s[j] == '.' || s[j] == '_' || ((s[j] == '-' || s[j] == '+') && (s[j - 1] == 'e' || s[j - 1] == 'E')))) ++j;
            }
            n.kind = Node::Number; n.text = s.substr(i, j - i); i = j;
            return n;
        }
        if (ident_start(c)) {
            size_t j = i;
            while (j < s.size() && ident_char(s[j])) ++j;
            n.text = s.substr(i, j - i);
            i = j;
            if (n.text == "inf" || n.text == "NaN") { n.kind = Node::Number; return n; }
            ws();
            if (i < s.size() && s[i] == '(') { ++i; parse_parens(n); return n; }   // Name( ... )
            n.kind = Node::Ident;
            return n;
        }
        fail(std::string("unexpected character '") + c + "'");
    }
};

[[noreturn]] void bad(const std::string& what) { throw RonError("EffectAsset RON: " + what); }

// ---- tree -> typed values -----------------------------------------------------------------------------------------------
double num(const Node& n, const char* what) {
    if (n.kind != Node::Number) bad(std::string(what) + ": expected a number");
    std::string t;
    for (char c : n.text) if (c != '_') t += c;
    if (t == "inf" || t == "+inf") return INFINITY;
    if (t == "-inf") return -INFINITY;
    if (t == "NaN") return NAN;
    char* end = nullptr;
    const double v = std::strtod(t.c_str(), &end);
    if (!end || *end) bad(std::string(what) + ": malformed number '" + n.text + "'");
    return v;
}
uint32_t u32_of(const Node& n, const char* what) {
    const double v = num(n, what);
    if (!(v >= 0.0 && v <= 4294967295.0) || v != std::floor(v)) bad(std::string(what) + ": expected an unsigned 32-bit integer");
    return (uint32_t)v;
}
int32_t i32_of(const Node& n, const char* what) {
    const double v = num(n, what);   // (NaN fails the range test)
    if (!(v >= -2147483648.0 && v <= 2147483647.0) || v != std::floor(v)) bad(std::string(what) + ": expected a signed 32-bit integer");
    return (int32_t)v;
}
bool bool_of(const Node& n, const char* what) {
    if (n.kind == Node::Ident && n.text == "true") return true;
    if (n.kind == Node::Ident && n.text == "false") return false;
    bad(std::string(what) + ": expected a bool");
}
const std::string& str_of(const Node& n, const char* what) {
    if (n.kind != Node::String) bad(std::string(what) + ": expected a string");
    return n.text;
}
// a newtype / tuple payload: Name(x) -> x
const Node& payload(const Node& n, size_t k, const char* what) {
    if (n.kind != Node::Struct || n.named_fields || n.fields.size() <= k) bad(std::string(what) + ": expected " + std::to_string(k + 1) + " positional value(s)");
    return n.fields[k].second;
}
const Node& req(const Node& n, const char* key, const char* what) {
    if (n.kind != Node::Struct) bad(std::string(what) + ": expected a struct");
    const Node* f = n.field(key);
    if (!f) bad(std::string(what) + ": missing field `" + key + "`");
    return *f;
}
const std::string& variant(const Node& n, const char* what) {
    if (n.kind != Node::Ident && n.kind != Node::Struct) bad(std::string(what) + ": expected an enum variant");
    if (n.text.empty()) bad(std::string(what) + ": expected a named enum variant");
    return n.text;
}

ExprHandle handle_of(const Node& n, const char* what) {
    try {
        return ExprHandle::parse(str_of(n, what));   // "#<id>" with the reference's rejections (expr.rs:182-200)
    } catch (const std::invalid_argument& e) {
        bad(std::string(what) + ": " + e.what());
    }
}
ExprHandle opt_handle_of(const Node& n, const char* what) {
    if (n.kind == Node::Ident && n.text == "None") return ExprHandle{};
    if (n.kind == Node::Struct && n.text == "Some") return handle_of(payload(n, 0, what), what);
    if (n.kind == Node::String) return handle_of(n, what);   // RON's implicit_some extension
    bad(std::string(what) + ": expected Some(\"#N\") or None");
}
Attribute attr_of(const Node& n, const char* what) {
    Attribute a;
    if (!Attribute::from_name(str_of(n, what), &a)) bad(std::string(what) + ": Unknown attribute name.");   // attributes.rs:711-715
    return a;
}

ScalarType scalar_type_of(const Node& n) {
    const std::string& v = variant(n, "ScalarType");
    if (v == "Bool") return ScalarType::Bool;
    if (v == "Float") return ScalarType::Float;
    if (v == "Int") return ScalarType::Int;
    if (v == "Uint") return ScalarType::Uint;
    bad("unknown ScalarType::" + v);
}
ValueType value_type_of(const Node& n) {
    const std::string& v = variant(n, "ValueType");
    if (v == "Scalar") return ValueType(scalar_type_of(payload(n, 0, "ValueType::Scalar")));
    if (v == "Vector") {
        const Node& vt = payload(n, 0, "ValueType::Vector");
        const uint32_t count = u32_of(req(vt, "count", "VectorType"), "VectorType::count");
        if (count < 2 || count > 4) bad("VectorType::count must be 2, 3 or 4");
        return ValueType(scalar_type_of(req(vt, "elem_type", "VectorType")), (uint8_t)count);
    }
    bad("ValueType::" + v + " is not supported on the simulation path");
}
Value value_of(const Node& n) {
    const std::string& v = variant(n, "Value");
    const Node& inner = payload(n, 0, "Value");
    if (v == "Scalar") {
        const std::string& st = variant(inner, "ScalarValue");
        const Node& x = payload(inner, 0, "ScalarValue");
        if (st == "Bool") return Value(bool_of(x, "ScalarValue::Bool"));
        if (st == "Float") return Value((float)num(x, "ScalarValue::Float"));
        if (st == "Int") return Value(i32_of(x, "ScalarValue::Int"));
        if (st == "Uint") return Value(u32_of(x, "ScalarValue::Uint"));
        bad("unknown ScalarValue::" + st);
    }
    if (v == "Vector") {
        const std::string& vt = variant(inner, "VectorValue");   // BVec2 .. Vec4 (VectorValueEnum, graph/mod.rs:1192-1206)
        if (vt.size() < 4) bad("unknown vector type " + vt);
        const char kind = vt[0];
        const int count = vt.back() - '0';
        const bool ok = (vt == std::string("Vec") + vt.back()) || ((kind == 'B' || kind == 'I' || kind == 'U') && vt.substr(1) == std::string("Vec") + vt.back());
        if (!ok || count < 2 || count > 4) bad("unknown vector type " + vt);
        const Node& tup = payload(inner, 0, "VectorValue");
        if (tup.kind != Node::Struct || tup.named_fields || (int)tup.fields.size() != count) bad(vt + ": expected " + std::to_string(count) + " components");
        Value out;
        const ScalarType st = vt[0] == 'V' ? ScalarType::Float : kind == 'B' ? ScalarType::Bool : kind == 'I' ? ScalarType::Int : ScalarType::Uint;
        out.type = ValueType(st, (uint8_t)count);
        for (int c = 0; c < count; ++c) {
            const Node& x = tup.fields[c].second;
            if (st == ScalarType::Float) out.set_f(c, (float)num(x, "vector component"));
            else if (st == ScalarType::Bool) out.bits[c] = bool_of(x, "vector component") ? 1u : 0u;
            else if (st == ScalarType::Uint) out.bits[c] = u32_of(x, "vector component");
            else out.bits[c] = (uint32_t)i32_of(x, "vector component");
        }
        return out;
    }
    bad("Value::" + v + " is not supported on the simulation path");
}

CpuValue cpu_value_of(const Node& n, const char* what) {
    const std::string& v = variant(n, what);
    if (v == "Single") return CpuValue::Single((float)num(payload(n, 0, what), what));
    if (v == "Uniform") {
        const Node& t = payload(n, 0, what);
        if (t.kind != Node::Struct || t.named_fields || t.fields.size() != 2) bad(std::string(what) + ": Uniform takes a pair");
        return CpuValue::Uniform((float)num(t.fields[0].second, what), (float)num(t.fields[1].second, what));
    }
    bad(std::string(what) + ": unknown CpuValue::" + v);
}

template <class E>
E enum_of(const Node& n, const char* what, std::initializer_list<std::pair<const char*, E>> table) {
    const std::string& v = variant(n, what);
    for (const auto& e : table)
        if (v == e.first) return e.second;
    bad(std::string("unknown ") + what + "::" + v);
}

// ---- modifiers -------------------------------------------------------------------------------------------------------------
struct ModType { const char* path; Modifier::Kind kind; };
const ModType kModTypes[] = {
    {"bevy_hanabi::modifier::attr::SetAttributeModifier", Modifier::Kind::SetAttribute},
    {"bevy_hanabi::modifier::attr::InheritAttributeModifier", Modifier::Kind::InheritAttribute},
    {"bevy_hanabi::modifier::position::SetPositionCircleModifier", Modifier::Kind::SetPositionCircle},
    {"bevy_hanabi::modifier::position::SetPositionSphereModifier", Modifier::Kind::SetPositionSphere},
    {"bevy_hanabi::modifier::position::SetPositionCone3dModifier", Modifier::Kind::SetPositionCone3d},
    {"bevy_hanabi::modifier::velocity::SetVelocityCircleModifier", Modifier::Kind::SetVelocityCircle},
    {"bevy_hanabi::modifier::velocity::SetVelocitySphereModifier", Modifier::Kind::SetVelocitySphere},
    {"bevy_hanabi::modifier::velocity::SetVelocityTangentModifier", Modifier::Kind::SetVelocityTangent},
    {"bevy_hanabi::modifier::accel::AccelModifier", Modifier::Kind::Accel},
    {"bevy_hanabi::modifier::accel::RadialAccelModifier", Modifier::Kind::RadialAccel},
    {"bevy_hanabi::modifier::accel::TangentAccelModifier", Modifier::Kind::TangentAccel},
    {"bevy_hanabi::modifier::force::LinearDragModifier", Modifier::Kind::LinearDrag},
    {"bevy_hanabi::modifier::force::ConformToSphereModifier", Modifier::Kind::ConformToSphere},
    {"bevy_hanabi::modifier::kill::KillSphereModifier", Modifier::Kind::KillSphere},
    {"bevy_hanabi::modifier::kill::KillAabbModifier", Modifier::Kind::KillAabb},
    {"bevy_hanabi::modifier::EmitSpawnEventModifier", Modifier::Kind::EmitSpawnEvent},
};
// field names of each modifier struct, in declaration order = the order of Modifier::e[]
const char* const* mod_fields(Modifier::Kind k, int* n) {
    static const char* const f_attr[] = {"value"};
    static const char* const f_circle[] = {"center", "axis", "radius"};
    static const char* const f_sphere[] = {"center", "radius"};
    static const char* const f_cone[] = {"height", "base_radius", "top_radius"};
    static const char* const f_vcircle[] = {"center", "axis", "speed"};
    static const char* const f_vsphere[] = {"center", "speed"};
    static const char* const f_vtangent[] = {"origin", "axis", "speed"};
    static const char* const f_accel[] = {"accel"};
    static const char* const f_radial[] = {"origin", "accel"};
    static const char* const f_tangent[] = {"origin", "axis", "accel"};
    static const char* const f_drag[] = {"drag"};
    static const char* const f_conform[] = {"origin", "radius", "influence_dist", "attraction_accel", "max_attraction_speed"};
    static const char* const f_ksphere[] = {"center", "sqr_radius"};
    static const char* const f_kaabb[] = {"center", "half_size"};
    static const char* const f_emit[] = {"count"};
    switch (k) {
        case Modifier::Kind::SetAttribute: *n = 1; return f_attr;
        case Modifier::Kind::SetPositionCircle: *n = 3; return f_circle;
        case Modifier::Kind::SetPositionSphere: *n = 2; return f_sphere;
        case Modifier::Kind::SetPositionCone3d: *n = 3; return f_cone;
        case Modifier::Kind::SetVelocityCircle: *n = 3; return f_vcircle;
        case Modifier::Kind::SetVelocitySphere: *n = 2; return f_vsphere;
        case Modifier::Kind::SetVelocityTangent: *n = 3; return f_vtangent;
        case Modifier::Kind::Accel: *n = 1; return f_accel;
        case Modifier::Kind::RadialAccel: *n = 2; return f_radial;
        case Modifier::Kind::TangentAccel: *n = 3; return f_tangent;
        case Modifier::Kind::LinearDrag: *n = 1; return f_drag;
        case Modifier::Kind::ConformToSphere: *n = 5; return f_conform;
        case Modifier::Kind::KillSphere: *n = 2; return f_ksphere;
        case Modifier::Kind::KillAabb: *n = 2; return f_kaabb;
        case Modifier::Kind::EmitSpawnEvent: *n = 1; return f_emit;
        default: *n = 0; return nullptr;
    }
}
bool has_dimension(Modifier::Kind k) {
    return k == Modifier::Kind::SetPositionCircle || k == Modifier::Kind::SetPositionSphere || k == Modifier::Kind::SetPositionCone3d;
}
bool has_kill_inside(Modifier::Kind k) { return k == Modifier::Kind::KillSphere || k == Modifier::Kind::KillAabb; }

// render-context modifiers: type path -> factory of the mirror (only their attribute requirements matter on this path)
Modifier render_modifier_of(const std::string& path, const Node& body) {
    const size_t k = path.rfind("::");
    const std::string name = k == std::string::npos ? path : path.substr(k + 2);
    if (name == "ColorOverLifetimeModifier") return ColorOverLifetimeModifier();
    if (name == "SizeOverLifetimeModifier") return SizeOverLifetimeModifier();
    if (name == "SetColorModifier") return SetColorModifier();
    if (name == "SetSizeModifier") return SetSizeModifier();
    if (name == "FlipbookModifier") return FlipbookModifier();
    if (name == "ScreenSpaceSizeModifier") return ScreenSpaceSizeModifier();
    if (name == "RoundModifier") return RoundModifier();
    if (name == "ParticleTextureModifier") return ParticleTextureModifier();
    if (name == "OrientModifier") {
        OrientMode mode = OrientMode::ParallelCameraDepthPlane;
        if (const Node* m = body.kind == Node::Struct ? body.field("mode") : nullptr)
            mode = enum_of<OrientMode>(*m, "OrientMode", {{"ParallelCameraDepthPlane", OrientMode::ParallelCameraDepthPlane}, {"FaceCameraPosition", OrientMode::FaceCameraPosition}, {"AlongVelocity", OrientMode::AlongVelocity}});
        return OrientModifier(mode);
    }
    bad("no modifier registered for type path '" + path + "'");   // registry.rs:158-165
}

std::vector<Modifier> modifiers_of(const Node& seq, uint32_t context, const char* what) {
    if (seq.kind != Node::Seq) bad(std::string(what) + ": expected a sequence of modifiers");
    std::vector<Modifier> out;
    for (const Node& entry : seq.items) {
        if (entry.kind != Node::Map || entry.entries.size() != 1) bad(std::string(what) + ": each modifier is a single-entry map { \"type path\": ( fields ) }");
        const std::string& path = str_of(entry.entries[0].first, what);
        const Node& body = entry.entries[0].second;
        if (context == CONTEXT_RENDER) { out.push_back(render_modifier_of(path, body)); continue; }
        const ModType* mt = nullptr;
        for (const ModType& t : kModTypes)
            if (path == t.path) mt = &t;
        if (!mt) bad("no modifier registered for type path '" + path + "'");
        if (body.kind != Node::Struct || (!body.named_fields && !body.fields.empty())) bad(path + ": expected a struct");
        Modifier m;
        m.kind = mt->kind;
        if (m.kind == Modifier::Kind::SetAttribute || m.kind == Modifier::Kind::InheritAttribute) m.attribute = attr_of(req(body, "attribute", path.c_str()), "attribute");
        int nf = 0;
        const char* const* names = mod_fields(m.kind, &nf);
        for (int f = 0; f < nf; ++f) m.e[f] = handle_of(req(body, names[f], path.c_str()), names[f]);
        if (m.kind == Modifier::Kind::ConformToSphere) {
            if (const Node* x = body.field("shell_half_thickness")) m.e[5] = opt_handle_of(*x, "shell_half_thickness");
            if (const Node* x = body.field("sticky_factor")) m.e[6] = opt_handle_of(*x, "sticky_factor");
            m.has_shell = m.e[5].valid();
            m.has_sticky = m.e[6].valid();
        }
        if (has_dimension(m.kind)) m.dimension = enum_of<ShapeDimension>(req(body, "dimension", path.c_str()), "ShapeDimension", {{"Surface", ShapeDimension::Surface}, {"Volume", ShapeDimension::Volume}});
        if (has_kill_inside(m.kind)) m.kill_inside = bool_of(req(body, "kill_inside", path.c_str()), "kill_inside");
        if (m.kind == Modifier::Kind::EmitSpawnEvent) {
            m.condition = enum_of<EventEmitCondition>(req(body, "condition", path.c_str()), "EventEmitCondition", {{"Always", EventEmitCondition::Always}, {"OnDie", EventEmitCondition::OnDie}});
            m.child_index = u32_of(req(body, "child_index", path.c_str()), "child_index");
        }
        out.push_back(m);
    }
    return out;
}

// ---- expressions -------------------------------------------------------------------------------------------------------------
const char* const kUnaryNames[] = {"Abs", "Acos", "Asin", "Atan", "All", "Any", "Ceil", "Cos", "Exp", "Exp2", "Floor", "Fract", "InvSqrt", "Length", "Log", "Log2",
                                   "Normalize", "Pack4x8snorm", "Pack4x8unorm", "Round", "Saturate", "Sign", "Sin", "Sqrt", "Tan", "Unpack4x8snorm", "Unpack4x8unorm",
                                   "W", "X", "Y", "Z"};
const char* const kBinaryNames[] = {"Add", "Atan2", "Cross", "Distance", "Div", "Dot", "GreaterThan", "GreaterThanOrEqual", "LessThan", "LessThanOrEqual", "Max", "Min",
                                    "Mul", "Remainder", "Step", "Sub", "UniformRand", "NormalRand", "Vec2", "Vec4XyzW"};
const char* const kTernaryNames[] = {"Mix", "Clamp", "SmoothStep", "Vec3"};
const char* const kBuiltInNames[] = {"Time", "DeltaTime", "VirtualTime", "VirtualDeltaTime", "RealTime", "RealDeltaTime", "Rand", "AlphaCutoff", "IsAlive"};

template <size_t N>
int index_of(const char* const (&names)[N], const std::string& v, const char* what) {
    for (size_t i = 0; i < N; ++i)
        if (v == names[i]) return (int)i;
    bad(std::string("unknown ") + what + "::" + v);
}

Expr expr_of(const Node& n) {
    const std::string& v = variant(n, "Expr");
    Expr e;
    if (v == "BuiltIn") {
        const Node& op = req(payload(n, 0, "Expr::BuiltIn"), "operator", "BuiltInExpr");
        e.kind = Expr::Kind::BuiltIn;
        e.builtin = (BuiltInOperator)index_of(kBuiltInNames, variant(op, "BuiltInOperator"), "BuiltInOperator");
        if (e.builtin == BuiltInOperator::Rand) e.rand_type = value_type_of(payload(op, 0, "BuiltInOperator::Rand"));
    } else if (v == "Literal") {
        e.kind = Expr::Kind::Literal;
        e.literal = value_of(payload(n, 0, "Expr::Literal"));   // LiteralExpr is #[serde(transparent)]
    } else if (v == "Property") {
        e.kind = Expr::Kind::Property;
        e.property = PropertyHandle{u32_of(payload(n, 0, "Expr::Property"), "PropertyHandle")};   // PropertyExpr and PropertyHandle are transparent
        if (e.property.id == 0) bad("PropertyHandle must be non-zero");
    } else if (v == "Attribute" || v == "ParentAttribute") {
        e.kind = v == "Attribute" ? Expr::Kind::Attribute : Expr::Kind::ParentAttribute;
        e.attribute = attr_of(req(payload(n, 0, "Expr::Attribute"), "attr", "AttributeExpr"), "attr");
    } else if (v == "Unary") {
        e.kind = Expr::Kind::Unary;
        e.unary = (UnaryOperator)index_of(kUnaryNames, variant(req(n, "op", "Expr::Unary"), "UnaryOperator"), "UnaryOperator");
        e.a = handle_of(req(n, "expr", "Expr::Unary"), "expr");
    } else if (v == "Binary") {
        e.kind = Expr::Kind::Binary;
        e.binary = (BinaryOperator)index_of(kBinaryNames, variant(req(n, "op", "Expr::Binary"), "BinaryOperator"), "BinaryOperator");
        e.a = handle_of(req(n, "left", "Expr::Binary"), "left");
        e.b = handle_of(req(n, "right", "Expr::Binary"), "right");
    } else if (v == "Ternary") {
        e.kind = Expr::Kind::Ternary;
        e.ternary = (TernaryOperator)index_of(kTernaryNames, variant(req(n, "op", "Expr::Ternary"), "TernaryOperator"), "TernaryOperator");
        e.a = handle_of(req(n, "first", "Expr::Ternary"), "first");
        e.b = handle_of(req(n, "second", "Expr::Ternary"), "second");
        e.c = handle_of(req(n, "third", "Expr::Ternary"), "third");
    } else if (v == "Cast") {
        const Node& c = payload(n, 0, "Expr::Cast");
        e.kind = Expr::Kind::Cast;
        e.a = handle_of(req(c, "inner", "CastExpr"), "inner");
        e.rand_type = value_type_of(req(c, "target", "CastExpr"));
    } else {
        bad("Expr::" + v + " is not supported on the simulation path");
    }
    return e;
}

// ---- writer ----------------------------------------------------------------------------------------------------------------
// f32 in Rust's `Display` form: the shortest decimal digits that round-trip, positional notation, `.0` for integral values.
std::string f32_text(float x) {
    if (std::isnan(x)) return "NaN";
    if (std::isinf(x)) return x < 0 ? "-inf" : "inf";
    char buf[64];
    int prec = 1;
    for (; prec <= 9; ++prec) {
        std::snprintf(buf, sizeof buf, "%.*e", prec - 1, (double)x);
        if ((float)std::strtod(buf, nullptr) == x) break;
    }
    // buf = d.ddddde[+-]XX -> digits and exponent
    std::string digits;
    int exp10 = 0;
    bool neg = false;
    {
        const char* p = buf;
        if (*p == '-') { neg = true; ++p; }
        for (; *p && *p != 'e'; ++p) if (*p != '.') digits += *p;
        exp10 = std::atoi(p + 1);
    }
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    std::string out = neg ? "-" : "";
    if (digits == "0") return out + "0.0";
    const int point = exp10 + 1;   // position of the decimal point relative to the start of `digits`
    if (point <= 0) out += "0." + std::string((size_t)-point, '0') + digits;
    else if ((size_t)point >= digits.size()) out += digits + std::string((size_t)point - digits.size(), '0') + ".0";
    else out += digits.substr(0, (size_t)point) + "." + digits.substr((size_t)point);
    return out;
}

struct Out {
    std::string s;
    int depth = 0;
    void nl() { s += "\n"; s.append((size_t)depth * 2, ' '); }
};
std::string quote(const std::string& v) {
    std::string o = "\"";
    for (char c : v) { if (c == '"' || c == '\\') { o += '\\'; o += c; } else if (c == '\n') o += "\\n"; else if (c == '\t') o += "\\t"; else if (c == '\r') o += "\\r"; else o += c; }
    return o + "\"";
}
std::string cpu_value_text(const CpuValue& v) {
    return v.is_uniform ? "Uniform((" + f32_text(v.a) + ", " + f32_text(v.b) + "))" : "Single(" + f32_text(v.a) + ")";
}
const char* scalar_type_name(ScalarType t) { return t == ScalarType::Bool ? "Bool" : t == ScalarType::Float ? "Float" : t == ScalarType::Int ? "Int" : "Uint"; }
std::string value_type_text(const ValueType& t) {
    if (t.count == 1) return std::string("Scalar(") + scalar_type_name(t.elem) + ")";
    return std::string("Vector((elem_type: ") + scalar_type_name(t.elem) + ", count: " + std::to_string((int)t.count) + "))";
}
std::string component_text(const Value& v, int c) {
    switch (v.type.elem) {
        case ScalarType::Float: return f32_text(v.get_f(c));
        case ScalarType::Bool: return v.bits[c] ? "true" : "false";
        case ScalarType::Int: return std::to_string((int32_t)v.bits[c]);
        default: return std::to_string(v.bits[c]);
    }
}
std::string value_text(const Value& v) {
    if (v.type.count == 1) return std::string("Scalar(") + scalar_type_name(v.type.elem) + "(" + component_text(v, 0) + "))";
    const char* prefix = v.type.elem == ScalarType::Float ? "" : v.type.elem == ScalarType::Bool ? "B" : v.type.elem == ScalarType::Int ? "I" : "U";
    std::string s = std::string("Vector(") + prefix + "Vec" + std::to_string((int)v.type.count) + "((";
    for (int c = 0; c < v.type.count; ++c) s += (c ? ", " : "") + component_text(v, c);
    return s + ")))";
}
std::string expr_text(const Expr& e) {
    switch (e.kind) {
        case Expr::Kind::BuiltIn:
            if (e.builtin == BuiltInOperator::Rand) return "BuiltIn((operator: Rand(" + value_type_text(e.rand_type) + ")))";
            return std::string("BuiltIn((operator: ") + kBuiltInNames[(int)e.builtin] + "))";
        case Expr::Kind::Literal: return "Literal(" + value_text(e.literal) + ")";
        case Expr::Kind::Property: return "Property(" + std::to_string(e.property.id) + ")";
        case Expr::Kind::Attribute: return "Attribute((attr: " + quote(e.attribute.name()) + "))";
        case Expr::Kind::ParentAttribute: return "ParentAttribute((attr: " + quote(e.attribute.name()) + "))";
        case Expr::Kind::Unary: return std::string("Unary(op: ") + kUnaryNames[(int)e.unary] + ", expr: " + quote(e.a.to_string()) + ")";
        case Expr::Kind::Binary: return std::string("Binary(op: ") + kBinaryNames[(int)e.binary] + ", left: " + quote(e.a.to_string()) + ", right: " + quote(e.b.to_string()) + ")";
        case Expr::Kind::Ternary:
            return std::string("Ternary(op: ") + kTernaryNames[(int)e.ternary] + ", first: " + quote(e.a.to_string()) + ", second: " + quote(e.b.to_string()) + ", third: " + quote(e.c.to_string()) + ")";
        case Expr::Kind::Cast: return "Cast((inner: " + quote(e.a.to_string()) + ", target: " + value_type_text(e.rand_type) + "))";
        default: throw RonError("EffectAsset RON: texture sample expressions are outside the simulation path");
    }
}

void write_modifiers(Out& o, const char* field, const std::vector<Modifier>& mods) {
    o.nl(); o.s += field; o.s += ": [";
    o.depth += 1;
    for (const Modifier& m : mods) {
        o.nl(); o.s += "{";
        o.depth += 1;
        o.nl();
        if (m.kind == Modifier::Kind::Render) {
            // type path of the render modifiers: bevy_hanabi::modifier::output::<Name>; no fields (applied onto a default instance)
            if (m.render_name == "OrientModifier") {   // the one render modifier whose fields change the particle layout (OrientMode, output.rs)
                const char* mode = m.render_attributes.size() >= 2 ? "AlongVelocity" : m.render_attributes.size() == 1 ? "FaceCameraPosition" : "ParallelCameraDepthPlane";
                o.s += quote("bevy_hanabi::modifier::output::OrientModifier") + ": (";
                o.depth += 1; o.nl(); o.s += std::string("mode: ") + mode + ","; o.depth -= 1;
                o.nl(); o.s += "),";
            } else {
                o.s += quote("bevy_hanabi::modifier::output::" + m.render_name) + ": (),";
            }
        } else {
            const ModType* mt = nullptr;
            for (const ModType& t : kModTypes)
                if (t.kind == m.kind) mt = &t;
            o.s += quote(mt->path) + ": (";
            o.depth += 1;
            if (m.kind == Modifier::Kind::SetAttribute || m.kind == Modifier::Kind::InheritAttribute) { o.nl(); o.s += "attribute: " + quote(m.attribute.name()) + ","; }
            if (m.kind == Modifier::Kind::EmitSpawnEvent) { o.nl(); o.s += std::string("condition: ") + (m.condition == EventEmitCondition::Always ? "Always" : "OnDie") + ","; }
            int nf = 0;
            const char* const* names = mod_fields(m.kind, &nf);
            for (int f = 0; f < nf; ++f) { o.nl(); o.s += std::string(names[f]) + ": " + quote(m.e[f].to_string()) + ","; }
            if (m.kind == Modifier::Kind::ConformToSphere) {
                o.nl(); o.s += "shell_half_thickness: " + (m.has_shell ? "Some(" + quote(m.e[5].to_string()) + ")" : std::string("None")) + ",";
                o.nl(); o.s += "sticky_factor: " + (m.has_sticky ? "Some(" + quote(m.e[6].to_string()) + ")" : std::string("None")) + ",";
            }
            if (has_dimension(m.kind)) { o.nl(); o.s += std::string("dimension: ") + (m.dimension == ShapeDimension::Surface ? "Surface" : "Volume") + ","; }
            if (has_kill_inside(m.kind)) { o.nl(); o.s += std::string("kill_inside: ") + (m.kill_inside ? "true" : "false") + ","; }
            if (m.kind == Modifier::Kind::EmitSpawnEvent) { o.nl(); o.s += "child_index: " + std::to_string(m.child_index) + ","; }
            o.depth -= 1;
            o.nl(); o.s += "),";
        }
        o.depth -= 1;
        o.nl(); o.s += "},";
    }
    o.depth -= 1;
    if (!mods.empty()) o.nl();
    o.s += "],";
}

}  // namespace

// EffectAsset::serialize (src/asset.rs:674-681): RON, pretty, two-space indentation.
std::string to_ron(const EffectAsset& asset) {
    Out o;
    o.s = "(";
    o.depth = 1;
    o.nl(); o.s += "name: " + quote(asset.name) + ",";
    o.nl(); o.s += "capacity: " + std::to_string(asset.capacity()) + ",";
    o.nl(); o.s += "spawner: (";
    o.depth += 1;
    o.nl(); o.s += "count: " + cpu_value_text(asset.spawner.count()) + ",";
    o.nl(); o.s += "spawn_duration: " + cpu_value_text(asset.spawner.spawn_duration()) + ",";
    o.nl(); o.s += "period: " + cpu_value_text(asset.spawner.period()) + ",";
    o.nl(); o.s += "cycle_count: " + std::to_string(asset.spawner.cycle_count()) + ",";
    o.nl(); o.s += std::string("starts_active: ") + (asset.spawner.starts_active() ? "true" : "false") + ",";
    o.nl(); o.s += std::string("emit_on_start: ") + (asset.spawner.emits_on_start() ? "true" : "false") + ",";
    o.depth -= 1;
    o.nl(); o.s += "),";
    o.nl(); o.s += "z_layer_2d: " + f32_text(asset.z_layer_2d) + ",";
    o.nl(); o.s += std::string("simulation_space: ") + (asset.simulation_space == SimulationSpace::Global ? "Global" : "Local") + ",";
    o.nl(); o.s += std::string("simulation_condition: ") + (asset.simulation_condition == SimulationCondition::WhenVisible ? "WhenVisible" : "Always") + ",";
    o.nl(); o.s += "prng_seed: " + std::to_string(asset.prng_seed) + ",";
    write_modifiers(o, "init_modifiers", asset.init_modifiers());
    write_modifiers(o, "update_modifiers", asset.update_modifiers());
    write_modifiers(o, "render_modifiers", asset.render_modifiers());
    o.nl(); o.s += std::string("motion_integration: ") + (asset.motion_integration == MotionIntegration::None ? "None" : asset.motion_integration == MotionIntegration::PreUpdate ? "PreUpdate" : "PostUpdate") + ",";
    o.nl(); o.s += "module: (";
    o.depth += 1;
    o.nl(); o.s += "expressions: [";
    o.depth += 1;
    for (const Expr& e : asset.module().expressions()) { o.nl(); o.s += expr_text(e) + ","; }
    o.depth -= 1;
    if (!asset.module().expressions().empty()) o.nl();
    o.s += "],";
    o.nl(); o.s += "properties: [";
    o.depth += 1;
    for (const Property& p : asset.module().properties()) {
        o.nl(); o.s += "(";
        o.depth += 1;
        o.nl(); o.s += "name: " + quote(p.name) + ",";
        o.nl(); o.s += "default_value: " + value_text(p.default_value) + ",";
        o.depth -= 1;
        o.nl(); o.s += "),";
    }
    o.depth -= 1;
    if (!asset.module().properties().empty()) o.nl();
    o.s += "],";
    o.nl(); o.s += "texture_layout: (";
    o.depth += 1;
    o.nl(); o.s += "layout: [],";
    o.depth -= 1;
    o.nl(); o.s += "),";
    o.depth -= 1;
    o.nl(); o.s += "),";
    o.nl(); o.s += "alpha_mode: Blend,";   // render state: not part of the simulation path, written at its default
    o.nl(); o.s += "mesh: None,";
    o.depth = 0;
    o.nl(); o.s += ")";
    return o.s;
}

// EffectAsset::deserialize (src/asset.rs:707-716). Every field of the struct must be present (missing_field errors of the
// reference's visitor, asset.rs:925-948), except the three modifier lists, which default to empty.
EffectAsset from_ron(const std::string& text) {
    Parser p(text);
    const Node root = p.parse_value();
    p.ws();
    if (p.i != text.size()) p.fail("trailing characters");
    if (root.kind != Node::Struct || !root.named_fields) bad("expected struct EffectAsset");
    if (!root.text.empty() && root.text != "EffectAsset") bad("expected struct EffectAsset, found " + root.text);
    for (size_t a = 0; a < root.fields.size(); ++a)
        for (size_t b = a + 1; b < root.fields.size(); ++b)
            if (root.fields[a].first == root.fields[b].first) bad("duplicate field `" + root.fields[a].first + "`");
    static const char* const known[] = {"name", "capacity", "spawner", "z_layer_2d", "simulation_space", "simulation_condition", "prng_seed", "init_modifiers",
                                        "update_modifiers", "render_modifiers", "motion_integration", "module", "alpha_mode", "mesh"};
    for (const auto& f : root.fields) {
        bool ok = false;
        for (const char* k : known) ok = ok || f.first == k;
        if (!ok) bad("unknown field `" + f.first + "`");
    }
    const Node& sp = req(root, "spawner", "EffectAsset");
    SpawnerSettings spawner = SpawnerSettings::make(cpu_value_of(req(sp, "count", "SpawnerSettings"), "count"), cpu_value_of(req(sp, "spawn_duration", "SpawnerSettings"), "spawn_duration"),
                                                    CpuValue(1.0f), 1);   // (period is assigned below: serde fills the fields without running `new`'s checks)
    {
        const CpuValue period = cpu_value_of(req(sp, "period", "SpawnerSettings"), "period");
        const uint32_t cycles = u32_of(req(sp, "cycle_count", "SpawnerSettings"), "cycle_count");
        // The reference deserialises the struct field by field; an asset that `SpawnerSettings::new` would have rejected can
        // only come from a hand-edited file. It is rejected here (the mirror keeps the type's invariant).
        spawner = SpawnerSettings::make(spawner.count(), spawner.spawn_duration(), period, cycles)
                      .with_starts_active(bool_of(req(sp, "starts_active", "SpawnerSettings"), "starts_active"))
                      .with_emit_on_start(bool_of(req(sp, "emit_on_start", "SpawnerSettings"), "emit_on_start"));
    }
    const Node& mod = req(root, "module", "EffectAsset");
    Module module;
    {
        const Node& props = req(mod, "properties", "Module");
        if (props.kind != Node::Seq) bad("Module::properties: expected a sequence");
        for (const Node& pn : props.items) module.add_property(str_of(req(pn, "name", "Property"), "Property::name"), value_of(req(pn, "default_value", "Property")));
        const Node& exprs = req(mod, "expressions", "Module");
        if (exprs.kind != Node::Seq) bad("Module::expressions: expected a sequence");
        for (const Node& en : exprs.items) module.add_expr(expr_of(en));
        const Node& tl = req(mod, "texture_layout", "Module");
        const Node& layout = req(tl, "layout", "TextureLayout");
        if (layout.kind != Node::Seq) bad("TextureLayout::layout: expected a sequence");
        // texture slots only matter to render modifiers: accepted and dropped
        // every handle an expression names must exist (the reference trusts the file and fails later, at shader generation)
        const uint32_t n = (uint32_t)module.expressions().size();
        for (const Expr& e : module.expressions())
            for (ExprHandle h : {e.a, e.b, e.c})
                if (h.valid() && h.id > n) bad("expression handle #" + std::to_string(h.id) + " is out of range (" + std::to_string(n) + " expressions)");
        for (const Expr& e : module.expressions())
            if (e.kind == Expr::Kind::Property && !module.get_property(e.property)) bad("property handle " + std::to_string(e.property.id) + " is out of range");
    }
    EffectAsset asset(u32_of(req(root, "capacity", "EffectAsset"), "capacity"), spawner, module);
    asset.name = str_of(req(root, "name", "EffectAsset"), "name");
    asset.z_layer_2d = (float)num(req(root, "z_layer_2d", "EffectAsset"), "z_layer_2d");
    asset.simulation_space = enum_of<SimulationSpace>(req(root, "simulation_space", "EffectAsset"), "SimulationSpace", {{"Global", SimulationSpace::Global}, {"Local", SimulationSpace::Local}});
    asset.simulation_condition = enum_of<SimulationCondition>(req(root, "simulation_condition", "EffectAsset"), "SimulationCondition",
                                                              {{"WhenVisible", SimulationCondition::WhenVisible}, {"Always", SimulationCondition::Always}});
    asset.prng_seed = u32_of(req(root, "prng_seed", "EffectAsset"), "prng_seed");
    asset.motion_integration = enum_of<MotionIntegration>(req(root, "motion_integration", "EffectAsset"), "MotionIntegration",
                                                          {{"None", MotionIntegration::None}, {"PreUpdate", MotionIntegration::PreUpdate}, {"PostUpdate", MotionIntegration::PostUpdate}});
    (void)req(root, "alpha_mode", "EffectAsset");   // render state: must be present like in the reference, not used on this path
    (void)req(root, "mesh", "EffectAsset");
    const uint32_t n_exprs = (uint32_t)asset.module().expressions().size();
    auto add = [&](const char* field, uint32_t context) {
        const Node* seq = root.field(field);
        if (!seq) return;   // unwrap_or_default (asset.rs:952-966)
        for (const Modifier& m : modifiers_of(*seq, context, field)) {
            for (const ExprHandle& h : m.e)
                if (h.valid() && h.id > n_exprs) bad(std::string(field) + ": expression handle #" + std::to_string(h.id) + " is out of range");
            if (context == CONTEXT_RENDER) asset.render(m);
            else asset.add_modifier(context, m);   // panics (PanicError) like the reference when the modifier does not support the context
        }
    };
    add("init_modifiers", CONTEXT_INIT);
    add("update_modifiers", CONTEXT_UPDATE);
    add("render_modifiers", CONTEXT_RENDER);
    return asset;
}

}  // namespace hanabi
