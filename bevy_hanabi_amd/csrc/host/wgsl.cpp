// The WGSL text the reference would emit for a literal or an expression.
//
// The reference defines the semantics of an effect by the WGSL its modifiers and expressions print
// (`ToWgslString`, src/lib.rs:259-430; `Expr::eval`, src/graph/expr.rs:1121-1258; the evaluation
// context `ShaderWriter`, src/modifier/mod.rs:204-367). This library does not run WGSL - lowering.cpp
// turns the same expression graph into a program for the HIP kernels - but the text is what the
// reference's own tests pin (expr.rs:4256-4680, lib.rs:1924-2029), so it is reproduced here: as the
// inspection output a maintainer can diff against the reference, and as the check that this Module
// mirror reads an expression graph the way the reference does (operand order, which operators are
// infix / functional / postfix, where parentheses go, literal rounding, rand hoisting into `let varN`).
#include <cmath>
#include <cstdio>
#include "hanabi.hpp"

namespace hanabi {

// `ToWgslString for f32` (lib.rs:264-269): "{:.6}", then trailing zeros trimmed ("1." / "0." / "-0.").
std::string to_wgsl_string(float x) {
    if (std::isnan(x)) return "NaN";  // Rust's Display for non-finite floats
    if (std::isinf(x)) return x < 0 ? "-inf" : "inf";
    char buf[400];
    std::snprintf(buf, sizeof buf, "%.6f", (double)x);
    std::string s = buf;
    while (!s.empty() && s.back() == '0') s.pop_back();
    return s;
}

namespace {
std::string scalar_text(ScalarType t, uint32_t bits) {
    switch (t) {
        case ScalarType::Bool: return bits ? "true" : "false";                       // lib.rs:311-319
        case ScalarType::Float: { float f; std::memcpy(&f, &bits, 4); return to_wgsl_string(f); }
        case ScalarType::Int: return std::to_string((int32_t)bits);                   // lib.rs:354-358
        case ScalarType::Uint: return std::to_string(bits) + "u";                     // lib.rs:393-397
    }
    return "";
}

const char* unary_name(UnaryOperator op) {  // expr.rs:2035-2071
    static const char* n[] = {"abs", "acos", "asin", "atan", "all", "any", "ceil", "cos", "exp", "exp2", "floor", "fract", "inverseSqrt",
                              "length", "log", "log2", "normalize", "pack4x8snorm", "pack4x8unorm", "round", "saturate", "sign", "sin", "sqrt",
                              "tan", "unpack4x8snorm", "unpack4x8unorm", "w", "x", "y", "z"};
    return n[(int)op];
}
bool unary_is_functional(UnaryOperator op) {  // expr.rs:2027-2032
    return !(op == UnaryOperator::X || op == UnaryOperator::Y || op == UnaryOperator::Z || op == UnaryOperator::W);
}
const char* binary_name(BinaryOperator op) {  // expr.rs:2272-2297
    static const char* n[] = {"+", "atan2", "cross", "distance", "/", "dot", ">", ">=", "<", "<=", "max", "min", "*", "%",
                              "step", "-", "rand_uniform", "rand_normal", "vec2", "vec4"};
    return n[(int)op];
}
bool binary_is_functional(BinaryOperator op) {  // expr.rs:2232-2255
    switch (op) {
        case BinaryOperator::Add: case BinaryOperator::Div: case BinaryOperator::GreaterThan: case BinaryOperator::GreaterThanOrEqual:
        case BinaryOperator::LessThan: case BinaryOperator::LessThanOrEqual: case BinaryOperator::Mul: case BinaryOperator::Remainder:
        case BinaryOperator::Sub: return false;
        default: return true;
    }
}
const char* ternary_name(TernaryOperator op) {  // expr.rs:2349-2357
    static const char* n[] = {"mix", "clamp", "smoothstep", "vec3"};
    return n[(int)op];
}
const char* builtin_name(const Expr& e) {  // expr.rs:1666-1705
    switch (e.builtin) {
        case BuiltInOperator::Time: return "time";
        case BuiltInOperator::DeltaTime: return "delta_time";
        case BuiltInOperator::VirtualTime: return "virtual_time";
        case BuiltInOperator::VirtualDeltaTime: return "virtual_delta_time";
        case BuiltInOperator::RealTime: return "real_time";
        case BuiltInOperator::RealDeltaTime: return "real_delta_time";
        case BuiltInOperator::AlphaCutoff: return "alpha_cutoff";
        case BuiltInOperator::IsAlive: return "is_alive";
        case BuiltInOperator::Rand: {
            static const char* n[4][4] = {{"brand", "brand2", "brand3", "brand4"}, {"frand", "frand2", "frand3", "frand4"},
                                          {"irand", "irand2", "irand3", "irand4"}, {"urand", "urand2", "urand3", "urand4"}};
            return n[(int)e.rand_type.elem][e.rand_type.count - 1];
        }
    }
    return "";
}
}  // namespace

// `ToWgslString for Value` (graph/mod.rs:287-296,1003-1024): scalars as above, vectors `vecN<T>(a,b,..)` without spaces.
std::string to_wgsl_string(const Value& v) {
    if (v.type.count == 1) return scalar_text(v.type.elem, v.bits[0]);
    std::string s = v.type.to_string() + "(";
    for (int i = 0; i < v.type.count; ++i) {
        if (i) s += ",";
        s += scalar_text(v.type.elem, v.bits[i]);
    }
    return s + ")";
}

std::string ShaderWriter::make_local_var() { return "var" + std::to_string(var_counter_++); }  // modifier/mod.rs:321-325

// check_side_effects_and_create_local_if_needed (expr.rs:1812-1824)
std::string ShaderWriter::hoist_if_side_effect(const std::string& code, bool side_effect) {
    if (!side_effect) return code;
    const std::string var = make_local_var();
    main_code += "let " + var + " = " + code + ";\n";
    return var;
}

// `EvalContext::eval` of ShaderWriter: memoised per handle so that a side effect is emitted once
// (modifier/mod.rs:309-319), then `Expr::eval` (expr.rs:1121-1258).
std::string ShaderWriter::eval(const Module& m, ExprHandle h) {
    auto it = expr_cache_.find(h.id);
    if (it != expr_cache_.end()) return it->second;
    const Expr& e = m.try_get(h);
    std::string out;
    switch (e.kind) {
        case Expr::Kind::BuiltIn: {  // expr.rs:1730-1738,1791-1797
            std::string s;
            if (e.builtin == BuiltInOperator::Rand) s = std::string(builtin_name(e)) + "()";
            else if (e.builtin == BuiltInOperator::IsAlive) s = "is_alive";
            else s = std::string("sim_params.") + builtin_name(e);
            out = hoist_if_side_effect(s, e.has_side_effect());
        } break;
        case Expr::Kind::Literal: out = to_wgsl_string(e.literal); break;
        case Expr::Kind::Property: {  // expr.rs:1424-1439, properties.rs:168-172
            const Property* p = m.get_property(e.property);
            if (!p) throw ExprError(ExprError::PropertyError, "Unknown property handle in evaluation module.");
            out = "properties[properties_array_index]." + p->name;
        } break;
        case Expr::Kind::Attribute:
        case Expr::Kind::ParentAttribute: {  // expr.rs:1352-1376
            const bool parent = e.kind == Expr::Kind::ParentAttribute;
            if (e.attribute == Attribute::ID) out = parent ? "parent_particle_index" : "particle_index";
            else if (e.attribute == Attribute::PARTICLE_COUNTER) out = "particle_counter";
            else {
                const std::string owner = parent ? "parent_particle" : "particle";
                out = (attribute_pointer_ ? "(*" + owner + ")." : owner + ".") + e.attribute.name();
            }
        } break;
        case Expr::Kind::Unary: {  // expr.rs:1133-1149
            const std::string x = eval(m, e.a);
            out = unary_is_functional(e.unary) ? std::string(unary_name(e.unary)) + "(" + x + ")" : x + "." + unary_name(e.unary);
        } break;
        case Expr::Kind::Binary: {  // expr.rs:1150-1218
            const std::string l = eval(m, e.a), r = eval(m, e.b);
            std::string body;
            if (!binary_is_functional(e.binary)) body = "(" + l + ") " + binary_name(e.binary) + " (" + r + ")";
            else if (e.binary == BinaryOperator::UniformRand || e.binary == BinaryOperator::NormalRand) {  // needs_type_suffix
                ValueType lt, rt;
                if (!m.try_get(e.a).value_type(&lt) || !m.try_get(e.b).value_type(&rt))
                    throw ExprError(ExprError::TypeError, "Can't determine the type of the operand");
                if (lt != rt) throw ExprError(ExprError::TypeError, "Mismatched types");
                if (!lt.is_float()) throw ExprError(ExprError::TypeError, "Unsupported type");
                static const char* suffix[] = {"f", "vec2", "vec3", "vec4"};
                body = std::string(binary_name(e.binary)) + "_" + suffix[lt.count - 1] + "(" + l + ", " + r + ")";
            } else body = std::string(binary_name(e.binary)) + "(" + l + ", " + r + ")";
            out = hoist_if_side_effect(body, e.has_side_effect());
        } break;
        case Expr::Kind::Ternary: {  // expr.rs:1219-1244
            const std::string a = eval(m, e.a), b = eval(m, e.b), c = eval(m, e.c);
            out = std::string(ternary_name(e.ternary)) + "(" + a + ", " + b + ", " + c + ")";
        } break;
        case Expr::Kind::Cast: out = e.rand_type.to_string() + "(" + eval(m, e.a) + ")"; break;  // expr.rs:1245-1250
        case Expr::Kind::TextureSample: throw ExprError(ExprError::GraphEvalError, "texture sampling is a render-only expression");
    }
    expr_cache_[h.id] = out;
    return out;
}

}  // namespace hanabi
