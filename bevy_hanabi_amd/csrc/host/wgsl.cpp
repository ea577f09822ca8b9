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
#include <cctype>
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


// ---- the simulation side of EffectShaderSources::generate ------------------------------------------------------------------------------
void ShaderWriter::set_emits_gpu_spawn_events(bool use_events) {   // modifier/mod.rs:262-281
    if (emits_events_ >= 0 && (emits_events_ != 0) != use_events) throw ExprError(ExprError::GraphEvalError, "Conflicting use of GPU spawn events.");
    emits_events_ = use_events ? 1 : 0;
}

// `ToWgslString for CpuValue<f32>` (lib.rs:432-482)
std::string to_wgsl_string(const CpuValue& v) {
    if (!v.is_uniform) return to_wgsl_string(v.a);
    return "(frand() * (" + to_wgsl_string(v.b) + " - " + to_wgsl_string(v.a) + ") + " + to_wgsl_string(v.a) + ")";
}

namespace {

// Function names: `<prefix>_<16 hex digits>` with a hash of the modifier's fields (the reference hashes the modifier struct with Rust's
// DefaultHasher, `calc_func_id`: same role - one function per distinct modifier -, different digits).
std::string func_name(const char* prefix, const Modifier& m) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&h](uint64_t v) { for (int i = 0; i < 8; ++i) { h ^= (v >> (8 * i)) & 0xffu; h *= 1099511628211ull; } };
    mix((uint64_t)m.kind);
    for (const ExprHandle& e : m.e) mix(e.id);
    mix((uint64_t)m.dimension); mix(m.kill_inside); mix(m.has_shell); mix(m.has_sticky);
    char buf[80];
    std::snprintf(buf, sizeof buf, "%s_%016llX", prefix, (unsigned long long)h);
    return buf;
}

std::string upper_of(const char* s) { std::string o; for (; *s; ++s) o += (char)std::toupper((unsigned char)*s); return o; }
const char* kPtrArg = "particle: ptr<function, Particle>";
const char* kXfPtrArgs = "transform: mat4x4<f32>, particle: ptr<function, Particle>";

// `Modifier::apply` of every simulation modifier: the statements it appends to main_code / extra_code (file:line per case).
void apply_modifier(const Modifier& mod, Module& m, ShaderWriter& w) {
    using K = Modifier::Kind;
    const std::string POS = Attribute::POSITION.name(), VEL = Attribute::VELOCITY.name();
    switch (mod.kind) {
        case K::SetAttribute: {   // attr.rs:92-115
            const Expr& ve = m.try_get(mod.e[0]);
            ValueType vt;
            if (ve.value_type(&vt) && vt != mod.attribute.value_type())
                throw ExprError(ExprError::TypeError, "Mismatching expression type in SetAttributeModifer: attribute '" + upper_of(mod.attribute.name()) + "' requires an expression producing a value of type " +
                                                          mod.attribute.value_type().to_string() + ", but a value of type " + vt.to_string() + " was produced instead");
            const std::string attr = w.eval(m, m.attr(mod.attribute));
            const std::string expr = w.eval(m, mod.e[0]);
            w.main_code += attr + " = " + expr + ";\n";
        } break;
        case K::InheritAttribute: {   // attr.rs:173-186
            const std::string attr = w.eval(m, m.attr(mod.attribute));
            w.main_code += attr + " = parent_particle." + mod.attribute.name() + ";\n";
        } break;
        case K::SetPositionCircle: {   // position.rs:52-108
            const std::string fn = func_name("set_position_circle", mod);
            w.make_fn(fn, kPtrArg, m, [&](Module& mm, ShaderWriter& ctx) {
                const std::string center = ctx.eval(mm, mod.e[0]), axis = ctx.eval(mm, mod.e[1]);
                const std::string radius = mod.dimension == ShapeDimension::Surface ? "let r = " + ctx.eval(mm, mod.e[2]) + ";" : "let r = sqrt(frand()) * (" + ctx.eval(mm, mod.e[2]) + ");";
                return "    // Circle center\n    let c = " + center + ";\n    // Circle basis\n    let n = " + axis + ";\n"
                       "    let sign = step(0.0, n.z) * 2.0 - 1.0;\n    let a = -1.0 / (sign + n.z);\n    let b = n.x * n.y * a;\n"
                       "    let tangent = vec3<f32>(1.0 + sign * n.x * n.x * a, sign * b, -sign * n.x);\n"
                       "    let bitangent = vec3<f32>(b, sign + n.y * n.y * a, -n.y);\n    // Circle radius\n    " + radius + "\n"
                       "    // Spawn random point on/in circle\n    let theta = frand() * tau;\n    let dir = tangent * cos(theta) + bitangent * sin(theta);\n"
                       "    (*particle)." + POS + " = c + r * dir;\n";
            });
            w.main_code += fn + "(&particle);\n";
        } break;
        case K::SetPositionSphere: {   // position.rs:152-210
            const std::string fn = func_name("set_position_sphere", mod);
            w.make_fn(fn, kPtrArg, m, [&](Module& mm, ShaderWriter& ctx) {
                const std::string center = ctx.eval(mm, mod.e[0]);
                const std::string radius = mod.dimension == ShapeDimension::Surface ? "let r = " + ctx.eval(mm, mod.e[1]) + ";" : "let r = pow(frand(), 1./3.) * (" + ctx.eval(mm, mod.e[1]) + ");";
                return "    // Sphere center\n    let c = " + center + ";\n\n    // Sphere radius\n    " + radius + "\n\n"
                       "    // Spawn randomly along the sphere surface using Archimedes's theorem\n    let theta = frand() * tau;\n    let z = frand() * 2. - 1.;\n"
                       "    let phi = acos(z);\n    let sinphi = sin(phi);\n    let x = sinphi * cos(theta);\n    let y = sinphi * sin(theta);\n"
                       "    let dir = vec3<f32>(x, y, z);\n    (*particle)." + POS + " = c + r * dir;\n";
            });
            w.main_code += fn + "(&particle);\n";
        } break;
        case K::SetPositionCone3d: {   // position.rs:267-324 (fields: height, base_radius, top_radius)
            const std::string fn = func_name("set_position_cone3d", mod);
            w.make_fn(fn, kXfPtrArgs, m, [&](Module& mm, ShaderWriter& ctx) {
                const std::string height = ctx.eval(mm, mod.e[0]), top = ctx.eval(mm, mod.e[2]), base = ctx.eval(mm, mod.e[1]);
                return "    // Truncated cone height\n    let h0 = " + height + ";\n    // Random height ratio\n    let alpha_h = pow(frand(), 1.0 / 3.0);\n"
                       "    // Random delta height from top\n    let h = h0 * alpha_h;\n    // Top radius\n    let rt = " + top + ";\n    // Bottom radius\n    let rb = " + base + ";\n"
                       "    // Radius at height h\n    let r0 = rb + (rt - rb) * alpha_h;\n    // Random delta radius\n    let alpha_r = sqrt(frand());\n"
                       "    // Random radius at height h\n    let r = r0 * alpha_r;\n    // Random base angle\n    let theta = frand() * tau;\n    let cost = cos(theta);\n    let sint = sin(theta);\n"
                       "    // Random position relative to truncated cone origin (not apex)\n    let x = r * cost;\n    let y = h;\n    let z = r * sint;\n    let p = vec3<f32>(x, y, z);\n"
                       "    let p2 = transform * vec4<f32>(p, 0.0);\n    (*particle)." + POS + " = p2.xyz;\n";
            });
            w.main_code += fn + "(transform, &particle);\n";
        } break;
        case K::SetVelocityCircle: {   // velocity.rs:45-80
            const std::string fn = func_name("set_velocity_circle", mod);
            w.make_fn(fn, kXfPtrArgs, m, [&](Module& mm, ShaderWriter& ctx) {
                const std::string center = ctx.eval(mm, mod.e[0]), axis = ctx.eval(mm, mod.e[1]), speed = ctx.eval(mm, mod.e[2]);
                return "    let delta = (*particle)." + POS + " - (" + center + ");\n    let radial = normalize(delta - dot(delta, " + axis + ") * (" + axis + "));\n"
                       "    let radial_vec4 = transform * vec4<f32>(radial.xyz, 0.0);\n    (*particle)." + VEL + " = radial_vec4.xyz * (" + speed + ");\n";
            });
            w.main_code += fn + "(transform, &particle);\n";
        } break;
        case K::SetVelocitySphere: {   // velocity.rs:124-139
            const std::string center = w.eval(m, mod.e[0]), speed = w.eval(m, mod.e[1]);
            w.main_code += "particle." + VEL + " = normalize(particle." + POS + " - (" + center + ")) * (" + speed + ");\n";
        } break;
        case K::SetVelocityTangent: {   // velocity.rs:188-223
            const std::string fn = func_name("set_velocity_tangent", mod);
            w.make_fn(fn, kXfPtrArgs, m, [&](Module& mm, ShaderWriter& ctx) {
                const std::string origin = ctx.eval(mm, mod.e[0]), axis = ctx.eval(mm, mod.e[1]), speed = ctx.eval(mm, mod.e[2]);
                return "    let radial = (*particle)." + POS + " - (" + origin + ");\n    let tangent = normalize(cross(" + axis + ", radial));\n"
                       "    let tangent_vec4 = transform * vec4<f32>(tangent.xyz, 0.0);\n    (*particle)." + VEL + " = tangent_vec4.xyz * (" + speed + ");\n";
            });
            w.main_code += fn + "(transform, &particle);\n";
        } break;
        case K::Accel: {   // accel.rs:79-86
            const std::string attr = w.eval(m, m.attr(Attribute::VELOCITY));
            const std::string expr = w.eval(m, mod.e[0]);
            w.main_code += attr + " += (" + expr + ") * sim_params.delta_time;";
        } break;
        case K::RadialAccel: {   // accel.rs:162-189 (the origin is pasted without parentheses)
            const std::string fn = func_name("radial_accel", mod);
            w.make_fn(fn, kPtrArg, m, [&](Module& mm, ShaderWriter& ctx) {
                const std::string origin = ctx.eval(mm, mod.e[0]), accel = ctx.eval(mm, mod.e[1]);
                return "let radial = normalize((*particle)." + POS + " - " + origin + ");\n            (*particle)." + VEL + " += radial * ((" + accel + ") * sim_params.delta_time);\n        ";
            });
            w.main_code += fn + "(&particle);\n";
        } break;
        case K::TangentAccel: {   // accel.rs:281-307 (evaluated in the CALLER's context: `particle.` appears inside a function - see tests)
            const std::string fn = func_name("tangent_accel", mod);
            const std::string origin = w.eval(m, mod.e[0]), axis = w.eval(m, mod.e[1]), accel = w.eval(m, mod.e[2]);
            w.extra_code += "fn " + fn + "(particle: ptr<function, Particle>) {\n    let radial = normalize((*particle)." + POS + " - " + origin + ");\n"
                            "    let tangent = normalize(cross(" + axis + ", radial));\n    (*particle)." + VEL + " += tangent * ((" + accel + ") * sim_params.delta_time);\n}\n";
            w.main_code += fn + "(&particle);\n";
        } break;
        case K::LinearDrag: {   // force.rs:284-297: built from Module operators
            const ExprHandle attr = m.attr(Attribute::VELOCITY);
            const ExprHandle dt = m.builtin(BuiltInOperator::DeltaTime);
            const ExprHandle drag_dt = m.mul(mod.e[0], dt);
            Value one; one.type = ValueType(ScalarType::Float, 1); one.set_f(0, 1.0f);
            Value zero; zero.type = ValueType(ScalarType::Float, 1); zero.set_f(0, 0.0f);
            const ExprHandle one_minus = m.sub(m.lit(one), drag_dt);
            const ExprHandle expr = m.max(m.lit(zero), one_minus);
            const std::string a = w.eval(m, attr), x = w.eval(m, expr);
            w.main_code += a + " *= " + x + ";";
        } break;
        case K::ConformToSphere: {   // force.rs:175-238 (fields: origin, radius, influence_dist, attraction_accel, max_attraction_speed, shell, sticky)
            const std::string fn = func_name("force_field", mod);
            w.make_fn(fn, kPtrArg, m, [&](Module& mm, ShaderWriter& ctx) {
                const std::string origin = ctx.eval(mm, mod.e[0]), radius = ctx.eval(mm, mod.e[1]), influence = ctx.eval(mm, mod.e[2]);
                const std::string shell = mod.has_shell ? ctx.eval(mm, mod.e[5]) : "0.1";
                const std::string max_speed = ctx.eval(mm, mod.e[4]), accel = ctx.eval(mm, mod.e[3]);
                const std::string sticky = mod.has_sticky ? ctx.eval(mm, mod.e[6]) : "2.0";
                const std::string P = "(*particle)." + POS, V = "(*particle)." + VEL;
                return "    // Sphere center\n    let c = " + origin + ";\n    // Sphere radius\n    let r = " + radius + ";\n"
                       "    // Distance and direction to origin (sphere center)\n    let rel_pos = c - " + P + ";\n    let origin_dist = length(rel_pos);\n    let origin_dir = normalize(rel_pos);\n"
                       "    // Signed distance to sphere surface, negative if inside sphere\n    let surface_dist = origin_dist - r;\n    // Influence distance\n    let influence_dist = " + influence + ";\n"
                       "    if (surface_dist > influence_dist) {\n        return;\n    }\n"
                       "    let cur_radial_speed = dot(" + V + ", origin_dir);\n    let shell_half_thickness = " + shell + ";\n"
                       "    let shell_factor = smoothstep(0., shell_half_thickness, abs(surface_dist));\n    let max_attraction_speed = " + max_speed + ";\n"
                       "    let max_radial_speed = sign(surface_dist) * shell_factor * max_attraction_speed;\n    // Delta radial speed to reach the ideal value\n    let delta_speed = max_radial_speed - cur_radial_speed;\n"
                       "    // Conforming delta speed from attraction acceleration\n    let attraction_accel = " + accel + ";\n    let sticky_accel = attraction_accel * " + sticky + ";\n"
                       "    let conforming_accel = mix(sticky_accel, attraction_accel, shell_factor);\n    let conforming_delta_speed = sim_params.delta_time * conforming_accel;\n"
                       "    // Final impulse clamped by the maximum acceleration speed\n    " + V + " += sign(delta_speed) * min(abs(delta_speed), conforming_delta_speed) * origin_dir;\n";
            });
            w.main_code += fn + "(&particle);\n";
        } break;
        case K::KillSphere: {   // kill.rs:76-96
            const ExprHandle diff = m.sub(m.attr(Attribute::POSITION), mod.e[0]);
            const ExprHandle sqr = m.dot(diff, diff);
            const ExprHandle cmp = mod.kill_inside ? m.lt(sqr, mod.e[1]) : m.gt(sqr, mod.e[1]);
            w.main_code += "if (" + w.eval(m, cmp) + ") {\n    is_alive = false;\n}\n";
        } break;
        case K::KillAabb: {   // kill.rs:156-181
            const ExprHandle dist = m.abs(m.sub(m.attr(Attribute::POSITION), mod.e[0]));
            const ExprHandle cmp = mod.kill_inside ? m.lt(dist, mod.e[1]) : m.gt(dist, mod.e[1]);
            const ExprHandle red = mod.kill_inside ? m.all(cmp) : m.any(cmp);
            w.main_code += "if (" + w.eval(m, red) + ") {\n    is_alive = false;\n}\n";
        } break;
        case K::EmitSpawnEvent: {   // modifier/mod.rs:671-715
            const std::string count_val = w.eval(m, mod.e[0]);
            const std::string count_var = w.make_local_var();
            w.push_stmt("let " + count_var + " = " + count_val + ";");
            const std::string call = "append_spawn_events_" + std::to_string(mod.child_index) + "((*effect_metadata).base_child_index, particle_index, " + count_var + "); }";
            w.main_code += (mod.condition == EventEmitCondition::Always ? "if (is_alive) { " : "if (was_alive && !is_alive) { ") + call;
            w.set_emits_gpu_spawn_events(true);
        } break;
        case K::Render: break;
    }
}

}  // namespace

WgslSources generate_wgsl(const EffectAsset& asset, bool has_parent) {
    WgslSources out;
    Module m = asset.module();   // `apply` adds expressions to (a clone of) the module (lib.rs:1021)
    out.attributes = asset.particle_layout();
    auto present = [&out](Attribute a) { for (const Attribute& x : out.attributes) if (x == a) return true; return false; };
    {   // init (lib.rs:1026-1058)
        ShaderWriter w(CONTEXT_INIT);
        for (const Modifier& mod : asset.init_modifiers()) apply_modifier(mod, m, w);
        out.init_code = w.main_code;
        out.init_extra = w.extra_code;
        if (asset.simulation_space == SimulationSpace::Global) {   // SimulationSpace::eval, lib.rs:518-531
            if (!present(Attribute::POSITION)) throw ExprError(ExprError::GraphEvalError, "Global-space simulation requires that the particles have a position attribute.");
            out.init_sim_space_transform = std::string("particle.") + Attribute::POSITION.name() + " += transform[3].xyz;";
        }
        out.consume_gpu_spawn_events = w.emits_gpu_spawn_events() == 1 || has_parent;
        for (const Expr& e : m.expressions()) if (e.kind == Expr::Kind::ParentAttribute) out.read_parent_particle = true;
        for (const Modifier& mod : asset.init_modifiers()) if (mod.kind == Modifier::Kind::InheritAttribute) out.read_parent_particle = true;
    }
    {   // update (lib.rs:1078-1131)
        ShaderWriter w(CONTEXT_UPDATE);
        for (const Modifier& mod : asset.update_modifiers()) apply_modifier(mod, m, w);
        out.update_code = w.main_code;
        out.update_extra = w.extra_code;
        out.emit_gpu_spawn_events = w.emits_gpu_spawn_events() == 1;
        if (asset.motion_integration != MotionIntegration::None && present(Attribute::POSITION) && present(Attribute::VELOCITY)) {
            const std::string code = std::string("\nparticle.") + Attribute::POSITION.name() + " += particle." + Attribute::VELOCITY.name() + " * sim_params.delta_time;\n";
            if (asset.motion_integration == MotionIntegration::PreUpdate) out.update_code.insert(0, code);
            else out.update_code += code;
        }
    }
    // aging / reaping (lib.rs:1223-1258)
    const bool has_age = present(Attribute::AGE), has_lifetime = present(Attribute::LIFETIME);
    const std::string AGE = Attribute::AGE.name(), LIFE = Attribute::LIFETIME.name();
    if (has_age) {
        if (has_lifetime) out.age_code += "\n    let was_alive = particle." + AGE + " < particle." + LIFE + ";";
        out.age_code += "\n    particle." + AGE + " = particle." + AGE + " + sim_params.delta_time;";
        if (has_lifetime) out.age_code += "\n    var is_alive = particle." + AGE + " < particle." + LIFE + ";";
    } else out.age_code = "\n    let was_alive = true;\n    var is_alive = true;";
    if (has_age && has_lifetime) out.reap_code = "is_alive = is_alive && (particle." + AGE + " < particle." + LIFE + ");";
    for (const Attribute& a : out.attributes)   // lib.rs:1266-1281
        if (!(a == Attribute::PREV) && !(a == Attribute::NEXT))
            out.writeback_code += std::string("    particle_buffer.particles[base_particle + particle_index].") + a.name() + " = particle." + a.name() + ";\n";
    return out;
}

}  // namespace hanabi
