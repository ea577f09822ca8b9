// C ABI over the host library (include/hanabi_amd_host.h): authoring + lowering for hosts that are not C++.
// Every entry point translates the exceptions the C++ mirror throws where the reference panics / returns Err
// (hanabi.hpp) into a status code + message; nothing here touches a GPU.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "../../../include/hanabi_amd_host.h"
#include "hanabi.hpp"

using namespace hanabi;

struct HnbModule { Module m; };
struct HnbAsset { EffectAsset a; };
struct HnbSpawner { EffectSpawner s; Pcg32 rng; };

namespace {

thread_local std::string g_error;

int fail(int code, const std::string& msg) { g_error = msg; return code; }

// Runs `f`, mapping the mirror's exception types to HnbStatus.
template <class F>
int guarded(F f) {
    try {
        return f();
    } catch (const ExprError& e) {
        return fail(HNB_ERR_EXPR, e.what());
    } catch (const ShaderGenerateError& e) {
        return fail(HNB_ERR_BAD_PROGRAM, e.what());
    } catch (const RonError& e) {
        return fail(HNB_ERR_BAD_PROGRAM, e.what());
    } catch (const PanicError& e) {
        return fail(HNB_ERR_INVALID_ARG, e.what());
    } catch (const SpawnerSettingsError& e) {
        return fail(HNB_ERR_INVALID_ARG, e.what());
    } catch (const std::bad_alloc&) {
        return fail(HNB_ERR_OUT_OF_MEMORY, "out of memory");
    } catch (const std::exception& e) {
        return fail(HNB_ERR_INVALID_ARG, e.what());
    }
}

bool to_value_type(uint32_t scalar_type, uint32_t count, ValueType* out) {
    if (scalar_type > HNB_U32 || count < 1 || count > 4) return false;
    *out = ValueType((ScalarType)scalar_type, (uint8_t)count);
    return true;
}
bool to_value(const HnbValue* v, Value* out) {
    ValueType t;
    if (!v || !to_value_type(v->scalar_type, v->count, &t)) return false;
    *out = Value::from_bits(t, v->bits);
    if (t.elem == ScalarType::Bool)
        for (uint32_t i = 0; i < v->count; ++i) out->bits[i] = v->bits[i] ? 1u : 0u;
    return true;
}
bool to_attr(uint32_t id, Attribute* out) {
    if (id >= HNB_ATTR_COUNT) return false;
    *out = Attribute((HnbAttr)id);
    return true;
}
CpuValue to_cpu(const HnbCpuValue& v) { return v.uniform ? CpuValue::Uniform(v.a, v.b) : CpuValue::Single(v.a); }
HnbCpuValue from_cpu(const CpuValue& v) { HnbCpuValue o; o.a = v.a; o.b = v.is_uniform ? v.b : v.a; o.uniform = v.is_uniform ? 1u : 0u; return o; }
void from_settings(const SpawnerSettings& s, HnbSpawnerSettings* out) {
    out->count = from_cpu(s.count());
    out->spawn_duration = from_cpu(s.spawn_duration());
    out->period = from_cpu(s.period());
    out->cycle_count = s.cycle_count();
    out->starts_active = s.starts_active() ? 1u : 0u;
    out->emit_on_start = s.emits_on_start() ? 1u : 0u;
}
SpawnerSettings to_settings(const HnbSpawnerSettings& s) {
    return SpawnerSettings::make(to_cpu(s.count), to_cpu(s.spawn_duration), to_cpu(s.period), s.cycle_count)
        .with_starts_active(s.starts_active != 0)
        .with_emit_on_start(s.emit_on_start != 0);
}

int copy_out(const std::vector<uint8_t>& bytes, void** out_blob, size_t* out_size) {
    void* p = std::malloc(bytes.empty() ? 1 : bytes.size());
    if (!p) return fail(HNB_ERR_OUT_OF_MEMORY, "out of memory");
    std::memcpy(p, bytes.data(), bytes.size());
    *out_blob = p;
    *out_size = bytes.size();
    return HNB_OK;
}

#define REQUIRE(cond, msg) do { if (!(cond)) return fail(HNB_ERR_INVALID_ARG, msg); } while (0)

}  // namespace

extern "C" {

const char* hnb_host_last_error(void) { return g_error.c_str(); }
void hnb_host_free(void* p) { std::free(p); }

// ---- Module ---------------------------------------------------------------------------------------------------------
int hnb_module_create(HnbModule** out_module) {
    REQUIRE(out_module, "out_module is NULL");
    return guarded([&] { *out_module = new HnbModule(); return HNB_OK; });
}
int hnb_module_destroy(HnbModule* module) { delete module; return HNB_OK; }

int hnb_module_lit(HnbModule* module, const HnbValue* value, HnbExprHandle* out) {
    REQUIRE(module && out, "NULL argument");
    Value v;
    REQUIRE(to_value(value, &v), "invalid value (scalar_type 0..3, count 1..4)");
    return guarded([&] { *out = module->m.lit(v).id; return HNB_OK; });
}
int hnb_module_attr(HnbModule* module, uint32_t attr, HnbExprHandle* out) {
    REQUIRE(module && out, "NULL argument");
    Attribute a;
    REQUIRE(to_attr(attr, &a), "unknown attribute");
    return guarded([&] { *out = module->m.attr(a).id; return HNB_OK; });
}
int hnb_module_parent_attr(HnbModule* module, uint32_t attr, HnbExprHandle* out) {
    REQUIRE(module && out, "NULL argument");
    Attribute a;
    REQUIRE(to_attr(attr, &a), "unknown attribute");
    return guarded([&] { *out = module->m.parent_attr(a).id; return HNB_OK; });
}
int hnb_module_add_property(HnbModule* module, const char* name, const HnbValue* default_value, uint32_t* out_property) {
    REQUIRE(module && name && out_property, "NULL argument");
    Value v;
    REQUIRE(to_value(default_value, &v), "invalid default value");
    return guarded([&] { *out_property = module->m.add_property(name, v).id; return HNB_OK; });
}
int hnb_module_prop(HnbModule* module, uint32_t property, HnbExprHandle* out) {
    REQUIRE(module && out, "NULL argument");
    REQUIRE(module->m.get_property(PropertyHandle{property}) != nullptr, "unknown property handle");
    return guarded([&] { *out = module->m.prop(PropertyHandle{property}).id; return HNB_OK; });
}
int hnb_module_builtin(HnbModule* module, uint32_t op, uint32_t rand_scalar_type, uint32_t rand_count, HnbExprHandle* out) {
    REQUIRE(module && out, "NULL argument");
    REQUIRE(op <= (uint32_t)BuiltInOperator::IsAlive, "unknown built-in operator");
    ValueType t;
    if (op == (uint32_t)BuiltInOperator::Rand) REQUIRE(to_value_type(rand_scalar_type, rand_count, &t), "invalid rand() value type");
    return guarded([&] { *out = module->m.builtin((BuiltInOperator)op, t).id; return HNB_OK; });
}
int hnb_module_unary(HnbModule* module, uint32_t op, HnbExprHandle a, HnbExprHandle* out) {
    REQUIRE(module && out, "NULL argument");
    REQUIRE(op <= (uint32_t)UnaryOperator::Z, "unknown unary operator");
    return guarded([&] { *out = module->m.unary((UnaryOperator)op, ExprHandle{a}).id; return HNB_OK; });
}
int hnb_module_binary(HnbModule* module, uint32_t op, HnbExprHandle a, HnbExprHandle b, HnbExprHandle* out) {
    REQUIRE(module && out, "NULL argument");
    REQUIRE(op <= (uint32_t)BinaryOperator::Vec4XyzW, "unknown binary operator");
    return guarded([&] { *out = module->m.binary((BinaryOperator)op, ExprHandle{a}, ExprHandle{b}).id; return HNB_OK; });
}
int hnb_module_ternary(HnbModule* module, uint32_t op, HnbExprHandle a, HnbExprHandle b, HnbExprHandle c, HnbExprHandle* out) {
    REQUIRE(module && out, "NULL argument");
    REQUIRE(op <= (uint32_t)TernaryOperator::Vec3, "unknown ternary operator");
    return guarded([&] { *out = module->m.ternary((TernaryOperator)op, ExprHandle{a}, ExprHandle{b}, ExprHandle{c}).id; return HNB_OK; });
}
int hnb_module_cast(HnbModule* module, HnbExprHandle a, uint32_t scalar_type, uint32_t count, HnbExprHandle* out) {
    REQUIRE(module && out, "NULL argument");
    ValueType t;
    REQUIRE(to_value_type(scalar_type, count, &t), "invalid cast target type");
    return guarded([&] { *out = module->m.cast(ExprHandle{a}, t).id; return HNB_OK; });
}
int hnb_module_num_expressions(const HnbModule* module, uint32_t* out) {
    REQUIRE(module && out, "NULL argument");
    *out = (uint32_t)module->m.expressions().size();
    return HNB_OK;
}

// ---- SpawnerSettings / EffectSpawner --------------------------------------------------------------------------------
int hnb_spawner_settings_new(HnbCpuValue count, HnbCpuValue spawn_duration, HnbCpuValue period, uint32_t cycle_count, HnbSpawnerSettings* out) {
    REQUIRE(out, "out is NULL");
    return guarded([&] { from_settings(SpawnerSettings::make(to_cpu(count), to_cpu(spawn_duration), to_cpu(period), cycle_count), out); return HNB_OK; });
}
int hnb_spawner_settings_once(HnbCpuValue count, HnbSpawnerSettings* out) {
    REQUIRE(out, "out is NULL");
    return guarded([&] { from_settings(SpawnerSettings::once(to_cpu(count)), out); return HNB_OK; });
}
int hnb_spawner_settings_rate(HnbCpuValue rate, HnbSpawnerSettings* out) {
    REQUIRE(out, "out is NULL");
    return guarded([&] { from_settings(SpawnerSettings::rate(to_cpu(rate)), out); return HNB_OK; });
}
int hnb_spawner_settings_burst(HnbCpuValue count, HnbCpuValue period, HnbSpawnerSettings* out) {
    REQUIRE(out, "out is NULL");
    return guarded([&] { from_settings(SpawnerSettings::burst(to_cpu(count), to_cpu(period)), out); return HNB_OK; });
}
int hnb_spawner_create(const HnbSpawnerSettings* settings, uint64_t rng_seed, uint64_t rng_stream, HnbSpawner** out) {
    REQUIRE(settings && out, "NULL argument");
    return guarded([&] {
        HnbSpawner* s = new HnbSpawner{EffectSpawner(to_settings(*settings)), Pcg32(rng_seed, rng_stream)};
        *out = s;
        return HNB_OK;
    });
}
int hnb_spawner_destroy(HnbSpawner* spawner) { delete spawner; return HNB_OK; }
int hnb_spawner_tick(HnbSpawner* spawner, float dt, uint32_t* out_spawn_count) {
    REQUIRE(spawner && out_spawn_count, "NULL argument");
    return guarded([&] { *out_spawn_count = spawner->s.tick(dt, spawner->rng); return HNB_OK; });
}
int hnb_spawner_reset(HnbSpawner* spawner) {
    REQUIRE(spawner, "spawner is NULL");
    spawner->s.reset();
    return HNB_OK;
}
int hnb_spawner_set_active(HnbSpawner* spawner, int active) {
    REQUIRE(spawner, "spawner is NULL");
    spawner->s.active = active != 0;
    return HNB_OK;
}

int hnb_next_prng_seed(uint32_t prng_seed, uint32_t* out_next) {
    REQUIRE(out_next, "out_next is NULL");
    *out_next = next_prng_seed(prng_seed);
    return HNB_OK;
}

// ---- EffectAsset ------------------------------------------------------------------------------------------------------
int hnb_asset_create(uint32_t capacity, const HnbSpawnerSettings* spawner, const HnbModule* module, HnbAsset** out_asset) {
    REQUIRE(spawner && module && out_asset, "NULL argument");
    return guarded([&] { *out_asset = new HnbAsset{EffectAsset(capacity, to_settings(*spawner), module->m)}; return HNB_OK; });
}
int hnb_asset_destroy(HnbAsset* asset) { delete asset; return HNB_OK; }
int hnb_asset_set_name(HnbAsset* asset, const char* name) {
    REQUIRE(asset && name, "NULL argument");
    return guarded([&] { asset->a.name = name; return HNB_OK; });
}
int hnb_asset_set_simulation_space(HnbAsset* asset, uint32_t space) {
    REQUIRE(asset, "asset is NULL");
    REQUIRE(space <= HNB_SPACE_LOCAL, "unknown simulation space");
    asset->a.simulation_space = (SimulationSpace)space;
    return HNB_OK;
}
int hnb_asset_set_simulation_condition(HnbAsset* asset, uint32_t condition) {
    REQUIRE(asset, "asset is NULL");
    REQUIRE(condition <= HNB_SIM_ALWAYS, "unknown simulation condition");
    asset->a.simulation_condition = (SimulationCondition)condition;
    return HNB_OK;
}
int hnb_asset_set_motion_integration(HnbAsset* asset, uint32_t integration) {
    REQUIRE(asset, "asset is NULL");
    REQUIRE(integration <= HNB_MOTION_POST_UPDATE, "unknown motion integration");
    asset->a.motion_integration = (MotionIntegration)integration;
    return HNB_OK;
}
int hnb_asset_set_prng_seed(HnbAsset* asset, uint32_t seed) {
    REQUIRE(asset, "asset is NULL");
    asset->a.prng_seed = seed;
    return HNB_OK;
}

int hnb_asset_add_modifier(HnbAsset* asset, uint32_t context, const HnbModifierDesc* d) {
    REQUIRE(asset && d, "NULL argument");
    REQUIRE(context == HNB_CONTEXT_INIT || context == HNB_CONTEXT_UPDATE || context == HNB_CONTEXT_RENDER, "context must be exactly one of INIT / UPDATE / RENDER");
    REQUIRE(d->kind >= HNB_MOD_SET_ATTRIBUTE && d->kind <= HNB_MOD_RENDER, "unknown modifier kind");
    REQUIRE(d->dimension <= 1u && d->condition <= 1u, "invalid dimension / condition");
    REQUIRE(d->n_render_attrs <= 8u, "too many render attributes");
    return guarded([&] {
        auto E = [&](int i) { return ExprHandle{d->e[i]}; };
        const ShapeDimension dim = (ShapeDimension)d->dimension;
        Attribute attr;
        if (d->kind == HNB_MOD_SET_ATTRIBUTE || d->kind == HNB_MOD_INHERIT_ATTRIBUTE)
            if (!to_attr(d->attribute, &attr)) return fail(HNB_ERR_INVALID_ARG, "unknown attribute");
        Modifier m;
        switch ((HnbModifierKind)d->kind) {
            case HNB_MOD_SET_ATTRIBUTE: m = SetAttributeModifier(attr, E(0)); break;
            case HNB_MOD_INHERIT_ATTRIBUTE: m = InheritAttributeModifier(attr); break;
            case HNB_MOD_SET_POSITION_CIRCLE: m = SetPositionCircleModifier(E(0), E(1), E(2), dim); break;
            case HNB_MOD_SET_POSITION_SPHERE: m = SetPositionSphereModifier(E(0), E(1), dim); break;
            case HNB_MOD_SET_POSITION_CONE3D: m = SetPositionCone3dModifier(E(0), E(1), E(2), dim); break;
            case HNB_MOD_SET_VELOCITY_CIRCLE: m = SetVelocityCircleModifier(E(0), E(1), E(2)); break;
            case HNB_MOD_SET_VELOCITY_SPHERE: m = SetVelocitySphereModifier(E(0), E(1)); break;
            case HNB_MOD_SET_VELOCITY_TANGENT: m = SetVelocityTangentModifier(E(0), E(1), E(2)); break;
            case HNB_MOD_ACCEL: m = AccelModifier(E(0)); break;
            case HNB_MOD_RADIAL_ACCEL: m = RadialAccelModifier(E(0), E(1)); break;
            case HNB_MOD_TANGENT_ACCEL: m = TangentAccelModifier(E(0), E(1), E(2)); break;
            case HNB_MOD_LINEAR_DRAG: m = LinearDragModifier(E(0)); break;
            case HNB_MOD_CONFORM_TO_SPHERE: m = ConformToSphereModifier(E(0), E(1), E(2), E(3), E(4), E(5), E(6)); break;
            case HNB_MOD_KILL_SPHERE: m = KillSphereModifier(E(0), E(1), d->kill_inside != 0); break;
            case HNB_MOD_KILL_AABB: m = KillAabbModifier(E(0), E(1), d->kill_inside != 0); break;
            case HNB_MOD_EMIT_SPAWN_EVENT: m = EmitSpawnEventModifier((EventEmitCondition)d->condition, E(0), d->child_index); break;
            case HNB_MOD_RENDER: {
                std::vector<Attribute> attrs;
                for (uint32_t i = 0; i < d->n_render_attrs; ++i) {
                    Attribute a;
                    if (!to_attr(d->render_attrs[i], &a)) return fail(HNB_ERR_INVALID_ARG, "unknown render attribute");
                    attrs.push_back(a);
                }
                m = RenderModifier("RenderModifier", attrs);
            } break;
        }
        // every expression a modifier names must exist in the asset's module (the reference stores handles unchecked and fails at
        // shader generation: ExprError::InvalidExprHandleError); reported here, where the mistake is made
        for (int i = 0; i < 7; ++i)
            if (d->e[i] != 0u && !asset->a.module().get(ExprHandle{d->e[i]}))
                return fail(HNB_ERR_EXPR, "modifier expression #" + std::to_string(d->e[i]) + " is not an expression of the asset's module");
        if (context == HNB_CONTEXT_RENDER) asset->a.render(m);   // EffectAsset::render (asset.rs:539-546); add_modifier itself only takes Init / Update
        else asset->a.add_modifier(context, m);
        return (int)HNB_OK;
    });
}

int hnb_asset_particle_layout(const HnbAsset* asset, uint32_t* out_attrs, uint32_t cap, uint32_t* out_count) {
    REQUIRE(asset && out_count, "NULL argument");
    return guarded([&] {
        const std::vector<Attribute> layout = asset->a.particle_layout();
        for (uint32_t i = 0; i < layout.size() && i < cap && out_attrs; ++i) out_attrs[i] = (uint32_t)layout[i].id;
        *out_count = (uint32_t)layout.size();
        return HNB_OK;
    });
}

int hnb_lower(const HnbAsset* asset, void** out_blob, size_t* out_size) {
    REQUIRE(asset && out_blob && out_size, "NULL argument");
    return guarded([&] { return copy_out(lower(asset->a), out_blob, out_size); });
}

int hnb_asset_to_ron(const HnbAsset* asset, char** out_text, size_t* out_size) {
    REQUIRE(asset && out_text && out_size, "NULL argument");
    return guarded([&] {
        const std::string t = to_ron(asset->a);
        char* p = static_cast<char*>(std::malloc(t.size() + 1));
        if (!p) return fail(HNB_ERR_OUT_OF_MEMORY, "out of memory");
        std::memcpy(p, t.c_str(), t.size() + 1);
        *out_text = p;
        *out_size = t.size();
        return (int)HNB_OK;
    });
}

int hnb_asset_wgsl(const HnbAsset* asset, int has_parent, char** out_text, size_t* out_size) {
    REQUIRE(asset && out_text && out_size, "NULL argument");
    return guarded([&] {
        const WgslSources w = generate_wgsl(asset->a, has_parent != 0);
        const std::string t = "// {{INIT_EXTRA}}\n" + w.init_extra + "\n// {{INIT_CODE}}\n" + w.init_code + "\n// {{SIMULATION_SPACE_TRANSFORM_PARTICLE}}\n" + w.init_sim_space_transform +
                              "\n// {{UPDATE_EXTRA}}\n" + w.update_extra + "\n// {{AGE_CODE}}" + w.age_code + "\n// {{REAP_CODE}}\n" + w.reap_code + "\n// {{UPDATE_CODE}}\n" + w.update_code +
                              "\n// {{WRITEBACK_CODE}}\n" + w.writeback_code;
        char* p = static_cast<char*>(std::malloc(t.size() + 1));
        if (!p) return fail(HNB_ERR_OUT_OF_MEMORY, "out of memory");
        std::memcpy(p, t.c_str(), t.size() + 1);
        *out_text = p;
        *out_size = t.size();
        return (int)HNB_OK;
    });
}

int hnb_asset_from_ron(const char* text, size_t size, HnbAsset** out_asset) {
    REQUIRE(text && out_asset, "NULL argument");
    return guarded([&] { *out_asset = new HnbAsset{from_ron(std::string(text, size))}; return HNB_OK; });
}

int hnb_asset_serialize(const HnbAsset* asset, void** out_blob, size_t* out_size) {
    REQUIRE(asset && out_blob && out_size, "NULL argument");
    return guarded([&] { return copy_out(serialize_asset(asset->a), out_blob, out_size); });
}

}  // extern "C"
