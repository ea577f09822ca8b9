// Python binding of the hanabi:: host library (test / bench harness entry point).
// The binding contains no simulation code: it builds assets, lowers them to program
// blobs and serialises them; the GPU path is reached through the C ABI (ctypes).
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "hanabi.hpp"

namespace py = pybind11;
using namespace hanabi;

namespace {

Value value_from_py(const py::handle& o) {
    if (py::isinstance<Value>(o)) return o.cast<Value>();
    if (py::isinstance<py::bool_>(o)) return Value(o.cast<bool>());
    if (py::isinstance<py::float_>(o)) return Value((float)o.cast<double>());
    if (py::isinstance<py::int_>(o)) return Value((int32_t)o.cast<long long>());
    if (py::isinstance<py::sequence>(o) && !py::isinstance<py::str>(o)) {
        py::sequence s = py::reinterpret_borrow<py::sequence>(o);
        const size_t n = py::len(s);
        if (n < 2 || n > 4) throw py::value_error("vector literals need 2 to 4 components");
        Value v;
        v.type = ValueType(ScalarType::Float, (uint8_t)n);
        for (size_t i = 0; i < n; ++i) v.set_f((int)i, (float)s[i].cast<double>());
        return v;
    }
    throw py::type_error("cannot convert object to a hanabi Value");
}

py::object value_to_py(const Value& v) {
    auto comp = [&](int i) -> py::object {
        switch (v.type.elem) {
            case ScalarType::Float: return py::float_(v.get_f(i));
            case ScalarType::Int: return py::int_((int32_t)v.bits[i]);
            case ScalarType::Uint: return py::int_(v.bits[i]);
            default: return py::bool_(v.bits[i] != 0);
        }
    };
    if (v.type.count == 1) return comp(0);
    py::list l;
    for (int i = 0; i < v.type.count; ++i) l.append(comp(i));
    return py::tuple(l);
}

CpuValue cpu_value_from_py(const py::handle& o) {
    if (py::isinstance<CpuValue>(o)) return o.cast<CpuValue>();
    if (py::isinstance<py::sequence>(o)) {
        py::sequence s = py::reinterpret_borrow<py::sequence>(o);
        if (py::len(s) != 2) throw py::value_error("CpuValue range needs exactly two values");
        return CpuValue((float)s[0].cast<double>(), (float)s[1].cast<double>());
    }
    return CpuValue((float)o.cast<double>());
}

}  // namespace

PYBIND11_MODULE(_hanabi_host, m) {
    m.doc() = "hanabi:: host library (authoring API mirror + lowering) for the MI355X particle hot path";

    py::register_exception<PanicError>(m, "PanicError");
    py::register_exception<ShaderGenerateError>(m, "ShaderGenerateError");
    static py::exception<ExprError> ex_expr(m, "ExprError");
    static py::exception<SpawnerSettingsError> ex_spawner(m, "SpawnerSettingsError");
    py::register_exception_translator([](std::exception_ptr p) {
        try {
            if (p) std::rethrow_exception(p);
        } catch (const ExprError& e) {
            static const char* kinds[] = {"TypeError", "SyntaxError", "GraphEvalError", "PropertyError", "InvalidExprHandleError", "InvalidModifierContext"};
            PyErr_SetString(ex_expr.ptr(), (std::string(kinds[e.kind]) + ": " + e.what()).c_str());
        } catch (const SpawnerSettingsError& e) {
            PyErr_SetString(ex_spawner.ptr(), (std::string(e.kind == SpawnerSettingsError::InvalidPeriod ? "InvalidPeriod" : "InfinitePeriod") + " min=" +
                                               std::to_string(e.min) + " max=" + std::to_string(e.max)).c_str());
        }
    });

    py::enum_<ScalarType>(m, "ScalarType").value("Bool", ScalarType::Bool).value("Float", ScalarType::Float).value("Int", ScalarType::Int).value("Uint", ScalarType::Uint);
    py::class_<ValueType>(m, "ValueType")
        .def(py::init<ScalarType, uint8_t>(), py::arg("elem"), py::arg("count") = 1)
        .def_readonly("elem", &ValueType::elem)
        .def_readonly("count", &ValueType::count)
        .def("__eq__", [](const ValueType& a, const ValueType& b) { return a == b; })
        .def("__repr__", &ValueType::to_string);
    py::implicitly_convertible<ScalarType, ValueType>();
    auto vt = m.def_submodule("VectorType");
    vt.attr("VEC2B") = VectorType::VEC2B; vt.attr("VEC3B") = VectorType::VEC3B; vt.attr("VEC4B") = VectorType::VEC4B;
    vt.attr("VEC2F") = VectorType::VEC2F; vt.attr("VEC3F") = VectorType::VEC3F; vt.attr("VEC4F") = VectorType::VEC4F;
    vt.attr("VEC2I") = VectorType::VEC2I; vt.attr("VEC3I") = VectorType::VEC3I; vt.attr("VEC4I") = VectorType::VEC4I;
    vt.attr("VEC2U") = VectorType::VEC2U; vt.attr("VEC3U") = VectorType::VEC3U; vt.attr("VEC4U") = VectorType::VEC4U;

    py::class_<Value>(m, "Value")
        .def(py::init([](py::object o) { return value_from_py(o); }))
        .def_static("f32", [](double x) { return Value((float)x); })
        .def_static("i32", [](long long x) { return Value((int32_t)x); })
        .def_static("u32", [](unsigned long long x) { return Value((uint32_t)x); })
        .def_static("bool", [](bool x) { return Value(x); })
        .def_static("vec_i", [](std::vector<int32_t> v) { Value r; r.type = ValueType(ScalarType::Int, (uint8_t)v.size()); for (size_t i = 0; i < v.size() && i < 4; ++i) r.bits[i] = (uint32_t)v[i]; return r; })
        .def_static("vec_u", [](std::vector<uint32_t> v) { Value r; r.type = ValueType(ScalarType::Uint, (uint8_t)v.size()); for (size_t i = 0; i < v.size() && i < 4; ++i) r.bits[i] = v[i]; return r; })
        .def_static("vec_b", [](std::vector<bool> v) { Value r; r.type = ValueType(ScalarType::Bool, (uint8_t)v.size()); for (size_t i = 0; i < v.size() && i < 4; ++i) r.bits[i] = v[i] ? 1u : 0u; return r; })
        .def("as_bytes", [](const Value& v) { const std::vector<uint8_t> b = v.as_bytes(); return py::bytes(reinterpret_cast<const char*>(b.data()), b.size()); })
        .def_property_readonly("value_type", &Value::value_type)
        .def_property_readonly("bits", [](const Value& v) { return std::vector<uint32_t>(v.bits, v.bits + v.type.count); })
        .def("to_py", &value_to_py);

    py::class_<Attribute> attr(m, "Attribute");
    attr.def_property_readonly("name", &Attribute::name)
        .def_property_readonly("id", [](const Attribute& a) { return (int)a.id; })
        .def_property_readonly("value_type", &Attribute::value_type)
        .def_property_readonly("default_value", &Attribute::default_value)
        .def_property_readonly("size", &Attribute::size)
        .def("__eq__", [](const Attribute& a, const Attribute& b) { return a == b; })
        .def("__hash__", [](const Attribute& a) { return (int)a.id; })
        .def("__repr__", [](const Attribute& a) { return std::string("Attribute.") + a.name(); })
        .def_static("from_name", [](const std::string& n) -> py::object { Attribute a; if (Attribute::from_name(n, &a)) return py::cast(a); return py::none(); })
        .def_static("all", &Attribute::all);
    for (const Attribute& a : Attribute::all()) {
        std::string up(a.name());
        for (char& c : up) c = (char)std::toupper((unsigned char)c);
        attr.attr(up.c_str()) = a;
    }

    py::class_<ExprHandle>(m, "ExprHandle")
        .def_readonly("id", &ExprHandle::id)
        .def("__eq__", [](const ExprHandle& a, const ExprHandle& b) { return a == b; })
        .def("to_string", &ExprHandle::to_string)
        .def_static("parse", &ExprHandle::parse)
        .def("__repr__", [](const ExprHandle& h) { return h.to_string(); });
    py::class_<PropertyHandle>(m, "PropertyHandle").def_readonly("id", &PropertyHandle::id);

    py::enum_<BuiltInOperator>(m, "BuiltInOperator")
        .value("Time", BuiltInOperator::Time).value("DeltaTime", BuiltInOperator::DeltaTime).value("VirtualTime", BuiltInOperator::VirtualTime)
        .value("VirtualDeltaTime", BuiltInOperator::VirtualDeltaTime).value("RealTime", BuiltInOperator::RealTime)
        .value("RealDeltaTime", BuiltInOperator::RealDeltaTime).value("Rand", BuiltInOperator::Rand).value("AlphaCutoff", BuiltInOperator::AlphaCutoff)
        .value("IsAlive", BuiltInOperator::IsAlive);
    py::enum_<ShapeDimension>(m, "ShapeDimension").value("Surface", ShapeDimension::Surface).value("Volume", ShapeDimension::Volume);
    py::enum_<OrientMode>(m, "OrientMode").value("ParallelCameraDepthPlane", OrientMode::ParallelCameraDepthPlane)
        .value("FaceCameraPosition", OrientMode::FaceCameraPosition).value("AlongVelocity", OrientMode::AlongVelocity);
    py::enum_<EventEmitCondition>(m, "EventEmitCondition").value("Always", EventEmitCondition::Always).value("OnDie", EventEmitCondition::OnDie);
    py::enum_<SimulationSpace>(m, "SimulationSpace").value("Global", SimulationSpace::Global).value("Local", SimulationSpace::Local);
    py::enum_<SimulationCondition>(m, "SimulationCondition").value("WhenVisible", SimulationCondition::WhenVisible).value("Always", SimulationCondition::Always);
    py::enum_<MotionIntegration>(m, "MotionIntegration").value("None_", MotionIntegration::None).value("PreUpdate", MotionIntegration::PreUpdate)
        .value("PostUpdate", MotionIntegration::PostUpdate);
    m.attr("CONTEXT_INIT") = (uint32_t)CONTEXT_INIT;
    m.attr("CONTEXT_UPDATE") = (uint32_t)CONTEXT_UPDATE;
    m.attr("CONTEXT_RENDER") = (uint32_t)CONTEXT_RENDER;

    py::class_<Module, std::shared_ptr<Module>> mod(m, "Module");
    mod.def(py::init<>())
        .def("lit", [](Module& s, py::object v) { return s.lit(value_from_py(v)); })
        .def("attr", &Module::attr)
        .def("parent_attr", &Module::parent_attr)
        .def("prop", &Module::prop)
        .def("builtin", &Module::builtin, py::arg("op"), py::arg("rand_type") = ValueType())
        .def("cast", &Module::cast)
        .def("add_property", [](Module& s, const std::string& n, py::object v) { return s.add_property(n, value_from_py(v)); })
        .def("get_property_by_name", [](const Module& s, const std::string& n) -> py::object { PropertyHandle h; if (s.get_property_by_name(n, &h)) return py::cast(h); return py::none(); })
        .def("is_const", &Module::is_const)
        .def("has_side_effect", &Module::has_side_effect)
        .def_property_readonly("num_expressions", [](const Module& s) { return s.expressions().size(); })
        .def_property_readonly("property_names", [](const Module& s) { std::vector<std::string> r; for (auto& p : s.properties()) r.push_back(p.name); return r; })
        .def_property_readonly("property_defaults", [](const Module& s) { std::vector<std::pair<std::string, Value>> r; for (auto& p : s.properties()) r.emplace_back(p.name, p.default_value); return r; });
#define B_UN(fn) mod.def(#fn, &Module::fn);
    B_UN(abs) B_UN(acos) B_UN(asin) B_UN(atan) B_UN(all) B_UN(any) B_UN(ceil) B_UN(cos) B_UN(exp) B_UN(exp2) B_UN(floor) B_UN(fract)
    B_UN(inverse_sqrt) B_UN(length) B_UN(log) B_UN(log2) B_UN(normalize) B_UN(pack4x8snorm) B_UN(pack4x8unorm) B_UN(round) B_UN(saturate)
    B_UN(sign) B_UN(sin) B_UN(sqrt) B_UN(tan) B_UN(unpack4x8snorm) B_UN(unpack4x8unorm) B_UN(w) B_UN(x) B_UN(y) B_UN(z)
    B_UN(add) B_UN(atan2) B_UN(cross) B_UN(distance) B_UN(div) B_UN(dot) B_UN(ge) B_UN(gt) B_UN(le) B_UN(lt) B_UN(max) B_UN(min) B_UN(mul)
    B_UN(rem) B_UN(step) B_UN(sub) B_UN(uniform) B_UN(normal) B_UN(vec2) B_UN(vec4_xyz_w) B_UN(mix) B_UN(clamp) B_UN(smoothstep) B_UN(vec3)
#undef B_UN

    py::class_<WriterExpr> we(m, "WriterExpr");
    we.def("expr", &WriterExpr::expr);
#define B_W(fn) we.def(#fn, &WriterExpr::fn);
    B_W(abs) B_W(all) B_W(any) B_W(acos) B_W(asin) B_W(atan) B_W(ceil) B_W(cos) B_W(exp) B_W(exp2) B_W(floor) B_W(fract) B_W(inverse_sqrt)
    B_W(length) B_W(log) B_W(log2) B_W(normalized) B_W(pack4x8snorm) B_W(pack4x8unorm) B_W(round) B_W(sign) B_W(sin) B_W(sqrt) B_W(tan)
    B_W(unpack4x8snorm) B_W(unpack4x8unorm) B_W(saturate) B_W(x) B_W(y) B_W(z) B_W(w)
    B_W(add) B_W(atan2) B_W(cross) B_W(dot) B_W(distance) B_W(div) B_W(ge) B_W(gt) B_W(le) B_W(lt) B_W(max) B_W(min) B_W(mul) B_W(normal)
    B_W(rem) B_W(sub) B_W(uniform) B_W(vec2) B_W(vec4_xyz_w) B_W(step) B_W(mix) B_W(clamp) B_W(smoothstep) B_W(vec3) B_W(cast)
#undef B_W
    we.def("__add__", [](const WriterExpr& a, const WriterExpr& b) { return a + b; })
        .def("__sub__", [](const WriterExpr& a, const WriterExpr& b) { return a - b; })
        .def("__mul__", [](const WriterExpr& a, const WriterExpr& b) { return a * b; })
        .def("__truediv__", [](const WriterExpr& a, const WriterExpr& b) { return a / b; })
        .def("__mod__", [](const WriterExpr& a, const WriterExpr& b) { return a % b; });

    py::class_<ExprWriter>(m, "ExprWriter")
        .def(py::init<>())
        .def("add_property", [](ExprWriter& w, const std::string& n, py::object v) { return w.add_property(n, value_from_py(v)); })
        .def("lit", [](ExprWriter& w, py::object v) { return w.lit(value_from_py(v)); })
        .def("attr", &ExprWriter::attr)
        .def("parent_attr", &ExprWriter::parent_attr)
        .def("prop", &ExprWriter::prop)
        .def("time", &ExprWriter::time)
        .def("delta_time", &ExprWriter::delta_time)
        .def("rand", &ExprWriter::rand)
        .def("alpha_cutoff", &ExprWriter::alpha_cutoff)
        .def("finish", &ExprWriter::finish);

    m.def("to_wgsl_string", [](py::object v) { return to_wgsl_string(value_from_py(v)); });
    m.def("cpu_value_to_wgsl_string", [](const CpuValue& v) { return to_wgsl_string(v); });
    // the simulation side of EffectShaderSources::generate as WGSL text (host/wgsl.cpp): a dict of the template slots of vfx_init / vfx_update
    m.def("generate_wgsl", [](const EffectAsset& a, bool has_parent) {
        const WgslSources w = generate_wgsl(a, has_parent);
        py::dict d;
        d["init_code"] = w.init_code; d["init_extra"] = w.init_extra; d["init_sim_space_transform"] = w.init_sim_space_transform;
        d["age_code"] = w.age_code; d["reap_code"] = w.reap_code; d["update_code"] = w.update_code; d["update_extra"] = w.update_extra;
        d["writeback_code"] = w.writeback_code;
        d["consume_gpu_spawn_events"] = w.consume_gpu_spawn_events; d["emit_gpu_spawn_events"] = w.emit_gpu_spawn_events; d["read_parent_particle"] = w.read_parent_particle;
        py::list attrs;
        for (const Attribute& at : w.attributes) attrs.append(at);
        d["attributes"] = attrs;
        return d;
    }, py::arg("asset"), py::arg("has_parent") = false);
    py::class_<ShaderWriter>(m, "ShaderWriter")
        .def(py::init<uint32_t, bool>(), py::arg("modifier_context"), py::arg("attribute_pointer") = false)
        .def("with_attribute_pointer", &ShaderWriter::with_attribute_pointer)
        .def("eval", &ShaderWriter::eval)
        .def("make_local_var", &ShaderWriter::make_local_var)
        .def_property_readonly("is_attribute_pointer", &ShaderWriter::is_attribute_pointer)
        .def_readonly("main_code", &ShaderWriter::main_code);

    py::class_<Modifier>(m, "Modifier")
        .def_property_readonly("context", &Modifier::context)
        .def_property_readonly("attributes", &Modifier::attributes)
        .def_property_readonly("kind", [](const Modifier& md) { return (uint32_t)md.kind; })
        .def_property_readonly("kill_inside", [](const Modifier& md) { return md.kill_inside; })
        .def("with_kill_inside", &Modifier::with_kill_inside);
    m.def("SetAttributeModifier", &SetAttributeModifier, py::arg("attribute"), py::arg("value"));
    m.def("InheritAttributeModifier", &InheritAttributeModifier);
    m.def("SetPositionCircleModifier", &SetPositionCircleModifier, py::arg("center"), py::arg("axis"), py::arg("radius"), py::arg("dimension"));
    m.def("SetPositionSphereModifier", &SetPositionSphereModifier, py::arg("center"), py::arg("radius"), py::arg("dimension"));
    m.def("SetPositionCone3dModifier", &SetPositionCone3dModifier, py::arg("height"), py::arg("base_radius"), py::arg("top_radius"), py::arg("dimension"));
    m.def("SetVelocityCircleModifier", &SetVelocityCircleModifier, py::arg("center"), py::arg("axis"), py::arg("speed"));
    m.def("SetVelocitySphereModifier", &SetVelocitySphereModifier, py::arg("center"), py::arg("speed"));
    m.def("SetVelocityTangentModifier", &SetVelocityTangentModifier, py::arg("origin"), py::arg("axis"), py::arg("speed"));
    m.def("AccelModifier", &AccelModifier, py::arg("accel"));
    m.def("RadialAccelModifier", &RadialAccelModifier, py::arg("origin"), py::arg("accel"));
    m.def("TangentAccelModifier", &TangentAccelModifier, py::arg("origin"), py::arg("axis"), py::arg("accel"));
    m.def("LinearDragModifier", &LinearDragModifier, py::arg("drag"));
    {   // `XModifier::constant(&mut module, ...)` / `::via_property(...)` of the reference, as XModifier_constant / XModifier_via_property
        auto v3 = [](const py::sequence& s) {
            if (py::len(s) != 3) throw py::value_error("expected 3 components");
            return Vec3Lit{(float)s[0].cast<double>(), (float)s[1].cast<double>(), (float)s[2].cast<double>()};
        };
        m.def("AccelModifier_constant", [v3](Module& md, py::sequence a) { return AccelModifierConstant(md, v3(a)); });
        m.def("AccelModifier_via_property", &AccelModifierViaProperty);
        m.def("RadialAccelModifier_constant", [v3](Module& md, py::sequence o, double a) { return RadialAccelModifierConstant(md, v3(o), (float)a); });
        m.def("RadialAccelModifier_via_property", [v3](Module& md, py::sequence o, PropertyHandle p) { return RadialAccelModifierViaProperty(md, v3(o), p); });
        m.def("TangentAccelModifier_constant",
              [v3](Module& md, py::sequence o, py::sequence ax, double a) { return TangentAccelModifierConstant(md, v3(o), v3(ax), (float)a); });
        m.def("TangentAccelModifier_via_property",
              [v3](Module& md, py::sequence o, py::sequence ax, PropertyHandle p) { return TangentAccelModifierViaProperty(md, v3(o), v3(ax), p); });
        m.def("LinearDragModifier_constant", [](Module& md, double d) { return LinearDragModifierConstant(md, (float)d); });
    }
    m.def("ConformToSphereModifier",
          [](ExprHandle origin, ExprHandle radius, ExprHandle influence_dist, ExprHandle attraction_accel, ExprHandle max_attraction_speed,
             py::object shell_half_thickness, py::object sticky_factor) {
              return ConformToSphereModifier(origin, radius, influence_dist, attraction_accel, max_attraction_speed,
                                             shell_half_thickness.is_none() ? ExprHandle{} : shell_half_thickness.cast<ExprHandle>(),
                                             sticky_factor.is_none() ? ExprHandle{} : sticky_factor.cast<ExprHandle>());
          },
          py::arg("origin"), py::arg("radius"), py::arg("influence_dist"), py::arg("attraction_accel"), py::arg("max_attraction_speed"),
          py::arg("shell_half_thickness") = py::none(), py::arg("sticky_factor") = py::none());
    m.def("KillSphereModifier", &KillSphereModifier, py::arg("center"), py::arg("sqr_radius"), py::arg("kill_inside") = false);
    m.def("KillAabbModifier", &KillAabbModifier, py::arg("center"), py::arg("half_size"), py::arg("kill_inside") = false);
    m.def("EmitSpawnEventModifier", &EmitSpawnEventModifier, py::arg("condition"), py::arg("count"), py::arg("child_index"));
    m.def("RenderModifier", &RenderModifier);
    m.def("ColorOverLifetimeModifier", &ColorOverLifetimeModifier);
    m.def("SizeOverLifetimeModifier", &SizeOverLifetimeModifier);
    m.def("SetColorModifier", &SetColorModifier);
    m.def("SetSizeModifier", &SetSizeModifier);
    m.def("OrientModifier", &OrientModifier);
    m.def("FlipbookModifier", &FlipbookModifier);
    m.def("ScreenSpaceSizeModifier", &ScreenSpaceSizeModifier);
    m.def("RoundModifier", &RoundModifier);
    m.def("ParticleTextureModifier", &ParticleTextureModifier);

    py::class_<Pcg32>(m, "Pcg32").def(py::init<>()).def(py::init<uint64_t, uint64_t>()).def("next_u32", &Pcg32::next_u32);
    py::class_<CpuValue>(m, "CpuValue")
        .def(py::init([](py::object o) { return cpu_value_from_py(o); }))
        .def_static("Single", &CpuValue::Single)
        .def_static("Uniform", &CpuValue::Uniform)
        .def("sample", &CpuValue::sample)
        .def("range", &CpuValue::range)
        .def_readonly("is_uniform", &CpuValue::is_uniform);
    py::class_<SpawnerSettings>(m, "SpawnerSettings")
        .def(py::init<>())
        .def_static("new", [](py::object c, py::object d, py::object p, uint32_t n) { return SpawnerSettings::make(cpu_value_from_py(c), cpu_value_from_py(d), cpu_value_from_py(p), n); })
        .def_static("try_new", [](py::object c, py::object d, py::object p, uint32_t n) { return SpawnerSettings::try_make(cpu_value_from_py(c), cpu_value_from_py(d), cpu_value_from_py(p), n); })
        .def_static("once", [](py::object c) { return SpawnerSettings::once(cpu_value_from_py(c)); })
        .def_static("rate", [](py::object c) { return SpawnerSettings::rate(cpu_value_from_py(c)); })
        .def_static("burst", [](py::object c, py::object p) { return SpawnerSettings::burst(cpu_value_from_py(c), cpu_value_from_py(p)); })
        .def("is_once", &SpawnerSettings::is_once)
        .def("is_forever", &SpawnerSettings::is_forever)
        .def("with_emit_on_start", &SpawnerSettings::with_emit_on_start)
        .def("set_emit_on_start", &SpawnerSettings::set_emit_on_start)
        .def("emits_on_start", &SpawnerSettings::emits_on_start)
        .def("with_count", [](const SpawnerSettings& s, py::object v) { return s.with_count(cpu_value_from_py(v)); })
        .def("with_spawn_duration", [](const SpawnerSettings& s, py::object v) { return s.with_spawn_duration(cpu_value_from_py(v)); })
        .def("with_period", [](const SpawnerSettings& s, py::object v) { return s.with_period(cpu_value_from_py(v)); })
        .def("set_period", [](SpawnerSettings& s, py::object v) { s.set_period(cpu_value_from_py(v)); })
        .def("with_cycle_count", &SpawnerSettings::with_cycle_count)
        .def("with_starts_active", &SpawnerSettings::with_starts_active)
        .def("set_starts_active", &SpawnerSettings::set_starts_active)
        .def("starts_active", &SpawnerSettings::starts_active)
        .def("count", &SpawnerSettings::count)
        .def("spawn_duration", &SpawnerSettings::spawn_duration)
        .def("period", &SpawnerSettings::period)
        .def("cycle_count", &SpawnerSettings::cycle_count);
    py::class_<EffectSpawner>(m, "EffectSpawner")
        .def(py::init<const SpawnerSettings&>())
        .def_readwrite("settings", &EffectSpawner::settings)
        .def_readwrite("spawn_count", &EffectSpawner::spawn_count)
        .def_readwrite("active", &EffectSpawner::active)
        .def("with_active", &EffectSpawner::with_active)
        .def("cycle_time", &EffectSpawner::cycle_time)
        .def("cycle_spawn_duration", &EffectSpawner::cycle_spawn_duration)
        .def("cycle_period", &EffectSpawner::cycle_period)
        .def("cycle_ratio", &EffectSpawner::cycle_ratio)
        .def("cycle_spawn_count", &EffectSpawner::cycle_spawn_count)
        .def("completed_cycle_count", &EffectSpawner::completed_cycle_count)
        .def("has_completed", &EffectSpawner::has_completed)
        .def("reset", &EffectSpawner::reset)
        .def("tick", &EffectSpawner::tick);

    py::class_<PropertyLayout>(m, "PropertyLayout")
        .def(py::init<>())
        .def(py::init([](const std::vector<std::pair<std::string, py::object>>& props) {
            std::vector<Property> v;
            for (const auto& kv : props) v.push_back(Property{kv.first, value_from_py(kv.second)});
            return PropertyLayout(v);
        }))
        .def_static("empty", []() { return PropertyLayout(); })
        .def("is_empty", &PropertyLayout::is_empty)
        .def("cpu_size", &PropertyLayout::cpu_size)
        .def("align", &PropertyLayout::align)
        .def("min_binding_size", &PropertyLayout::min_binding_size)
        .def("contains", &PropertyLayout::contains)
        .def("offset", [](const PropertyLayout& l, const std::string& n) -> py::object { uint32_t o; if (l.offset(n, &o)) return py::int_(o); return py::none(); })
        .def("properties", [](const PropertyLayout& l) { std::vector<std::pair<uint32_t, std::string>> r; for (const auto& e : l.properties()) r.emplace_back(e.offset, e.property.name); return r; })
        .def("generate_property_struct_code", [](const PropertyLayout& l) -> py::object { if (l.is_empty()) return py::none(); return py::str(l.generate_property_struct_code()); })
        .def("serialize", [](const PropertyLayout& l, const std::vector<std::pair<std::string, py::object>>& props) {
            std::vector<Property> v;
            for (const auto& kv : props) v.push_back(Property{kv.first, value_from_py(kv.second)});
            const std::vector<uint8_t> d = l.serialize(v);
            return py::bytes(reinterpret_cast<const char*>(d.data()), d.size());
        });

    py::class_<EffectProperties>(m, "EffectProperties")
        .def(py::init<>())
        .def("with_properties", [](EffectProperties& ep, const std::vector<std::pair<std::string, py::object>>& props) {
            std::vector<std::pair<std::string, Value>> v;
            for (const auto& kv : props) v.emplace_back(kv.first, value_from_py(kv.second));
            ep.with_properties(v);
            return ep;
        })
        .def("properties", [](const EffectProperties& ep) {
            py::list out;
            for (const auto& pi : ep.properties()) out.append(py::make_tuple(pi.def.name, value_to_py(pi.def.default_value), value_to_py(pi.value)));
            return out;
        })
        .def("get_stored", [](const EffectProperties& ep, const std::string& n) -> py::object { Value v; if (ep.get_stored(n, &v)) return value_to_py(v); return py::none(); })
        .def("set", [](EffectProperties& ep, const std::string& n, py::object v) { ep.set(n, value_from_py(v)); })
        .def("set_if_changed", [](EffectProperties& ep, const std::string& n, py::object v) { return ep.set_if_changed(n, value_from_py(v)); })
        .def("update", [](EffectProperties& ep, const std::vector<std::pair<std::string, py::object>>& asset_props) {
            std::vector<Property> v;
            for (const auto& kv : asset_props) v.push_back(Property{kv.first, value_from_py(kv.second)});
            ep.update(v);
        })
        .def("serialize", [](const EffectProperties& ep, const PropertyLayout& layout) {
            const std::vector<uint8_t> d = ep.serialize(layout);
            return py::bytes(reinterpret_cast<const char*>(d.data()), d.size());
        })
        .def("layout", [](const EffectProperties& ep) { std::vector<Property> defs; for (const auto& pi : ep.properties()) defs.push_back(pi.def); return PropertyLayout(defs); });

    py::class_<ParticleLayout>(m, "ParticleLayout")
        .def_static("new", []() { return ParticleLayout::make(); })
        .def_static("empty", &ParticleLayout::empty)
        .def_static("default", &ParticleLayout::default_layout)
        .def("is_empty", &ParticleLayout::is_empty)
        .def("len", &ParticleLayout::len)
        .def("size", &ParticleLayout::size)
        .def("align", &ParticleLayout::align)
        .def("min_binding_size", &ParticleLayout::min_binding_size)
        .def("contains", &ParticleLayout::contains)
        .def("merged_with", &ParticleLayout::merged_with)
        .def("byte_offset", [](const ParticleLayout& l, Attribute a) -> py::object { uint32_t o; if (l.byte_offset(a, &o)) return py::int_(o); return py::none(); })
        .def("entries", [](const ParticleLayout& l) {
            py::list out;
            for (const AttributeLayout& e : l.entries()) out.append(py::make_tuple(e.padding ? std::string("pad") : std::string(e.attribute.name()), e.offset));
            return out;
        });
    py::class_<ParticleLayout::Builder>(m, "ParticleLayoutBuilder")
        .def("append", [](ParticleLayout::Builder& b, Attribute a) { b.append(a); return b; })
        .def("build", &ParticleLayout::Builder::build);
    py::class_<EffectAsset>(m, "EffectAsset")
        .def(py::init<uint32_t, const SpawnerSettings&, const Module&>(), py::arg("capacity"), py::arg("spawner"), py::arg("module"))
        .def_readwrite("name", &EffectAsset::name)
        .def_readwrite("spawner", &EffectAsset::spawner)
        .def_readwrite("simulation_space", &EffectAsset::simulation_space)
        .def_readwrite("simulation_condition", &EffectAsset::simulation_condition)
        .def_readwrite("prng_seed", &EffectAsset::prng_seed)
        .def_readwrite("motion_integration", &EffectAsset::motion_integration)
        .def_readwrite("z_layer_2d", &EffectAsset::z_layer_2d)
        .def_property_readonly("capacity", &EffectAsset::capacity)
        .def("module", &EffectAsset::module)
        .def("with_name", [](EffectAsset& a, const std::string& n) { return a.with_name(n); })
        .def("with_simulation_space", [](EffectAsset& a, SimulationSpace s) { return a.with_simulation_space(s); })
        .def("with_simulation_condition", [](EffectAsset& a, SimulationCondition s) { return a.with_simulation_condition(s); })
        .def("with_motion_integration", [](EffectAsset& a, MotionIntegration s) { return a.with_motion_integration(s); })
        .def("init", [](EffectAsset& a, const Modifier& md) { return a.init(md); })
        .def("update", [](EffectAsset& a, const Modifier& md) { return a.update(md); })
        .def("render", [](EffectAsset& a, const Modifier& md) { return a.render(md); })
        .def("add_modifier", [](EffectAsset& a, uint32_t ctx, const Modifier& md) { return a.add_modifier(ctx, md); })
        .def("particle_layout", &EffectAsset::particle_layout)
        .def("reference_particle_layout", &EffectAsset::reference_particle_layout)
        .def("property_layout", &EffectAsset::property_layout)
        .def_property_readonly("init_modifiers", &EffectAsset::init_modifiers)
        .def_property_readonly("update_modifiers", &EffectAsset::update_modifiers)
        .def_property_readonly("render_modifiers", &EffectAsset::render_modifiers);

    m.def("round_literal_f32", &round_literal_f32);
    m.def("lower", [](const EffectAsset& a) { auto b = lower(a); return py::bytes(reinterpret_cast<const char*>(b.data()), b.size()); });
    m.def("disassemble", [](py::bytes b) { std::string s = b; return disassemble(std::vector<uint8_t>(s.begin(), s.end())); });
    m.def("next_prng_seed", &next_prng_seed, "StdRng::seed_from_u64(seed).random::<u32>() (src/lib.rs:1813-1820)");
    m.def("seed_from_u64", [](uint64_t state) { uint8_t seed[32]; seed_from_u64(state, seed); return py::bytes(reinterpret_cast<const char*>(seed), 32); });
    m.def("chacha_block", [](py::bytes key, uint64_t counter, uint64_t stream, int rounds) {
        const std::string k = key;
        if (k.size() != 32) throw std::invalid_argument("key must be 32 bytes");
        uint32_t kw[8], out[16];
        std::memcpy(kw, k.data(), 32);
        chacha_block(kw, counter, stream, rounds, out);
        return py::bytes(reinterpret_cast<const char*>(out), 64);
    });
    m.def("to_ron", &to_ron, "EffectAsset::serialize (src/asset.rs:674-681): the reference's RON text format");
    m.def("from_ron", &from_ron, "EffectAsset::deserialize (src/asset.rs:707-716)");
    m.def("serialize_asset", [](const EffectAsset& a) { auto b = serialize_asset(a); return py::bytes(reinterpret_cast<const char*>(b.data()), b.size()); });
}
