/* hanabi_amd_host.h — C ABI of the host side of the path: authoring (Module / EffectAsset) and lowering.
 *
 * include/hanabi_amd.h is the device boundary (programs in, simulated frames out). This header is what lets a host that
 * is NOT C++ — the reference's Rust crate first of all — produce the program blob hnb_program_create() takes, without
 * re-implementing the lowering: it exposes the C++ mirror of the authoring API (bevy_hanabi_amd/csrc/host/hanabi.hpp)
 * through plain pointers and integers, in libhanabi_host.so (no GPU, no HIP dependency).
 *
 *   reference                                                   here
 *   Module::{lit,attr,prop,builtin,unary,binary,ternary,cast}   src/graph/expr.rs:376-778      hnb_module_*
 *   Module::add_property                                        src/graph/expr.rs:376-412      hnb_module_add_property
 *   SpawnerSettings::{new,once,rate,burst}                      src/spawn.rs:293-474           hnb_spawner_settings_*
 *   EffectSpawner::{new,tick,reset}                             src/spawn.rs:699-717,814-921   hnb_spawner_*
 *   EffectAsset::{new,init,update,render,with_*}                src/asset.rs:391-546           hnb_asset_*
 *   Modifier impls (init / update contexts)                     src/modifier/{attr,..}.rs      HnbModifierDesc + hnb_asset_add_modifier
 *   EffectShaderSources::generate                               src/lib.rs:805-1336            hnb_lower
 *
 * A Rust binding walks its own `Module` (expressions are stored in a Vec, handles are 1-based indices: the same numbering
 * as HnbExprHandle) and its modifier lists once per asset, then calls hnb_lower; INTEGRATION.md shows the `extern "C"` block
 * and the walk. All functions return HNB_OK (0) or a negative HnbStatus; hnb_host_last_error() gives the text — Rust panics of
 * the reference (asset.rs:482,499; attr.rs:81-88; spawn.rs:299-311) come back as HNB_ERR_INVALID_ARG, ExprError as
 * HNB_ERR_EXPR, ShaderGenerateError as HNB_ERR_BAD_PROGRAM.
 */
#ifndef HANABI_AMD_HOST_H
#define HANABI_AMD_HOST_H

#include "hanabi_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

#define HNB_ERR_EXPR (-8) /* ExprError (src/graph/expr.rs:785-825): type error, invalid handle, property error ... */

typedef struct HnbModule HnbModule;   /* hanabi::Module: expression arena + property definitions */
typedef struct HnbAsset HnbAsset;     /* hanabi::EffectAsset */
typedef struct HnbSpawner HnbSpawner; /* hanabi::EffectSpawner + the Pcg32 it samples CpuValue::Uniform with */
typedef uint32_t HnbExprHandle;       /* ExprHandle: 1-based index into the module, 0 = none (src/graph/expr.rs:132-213) */

/* Operators, numbered in the reference's declaration order (src/graph/expr.rs:910-995, 1832-2390). */
typedef enum HnbBuiltIn { HNB_BI_TIME = 0, HNB_BI_DELTA_TIME, HNB_BI_VIRTUAL_TIME, HNB_BI_VIRTUAL_DELTA_TIME, HNB_BI_REAL_TIME,
                          HNB_BI_REAL_DELTA_TIME, HNB_BI_RAND, HNB_BI_ALPHA_CUTOFF, HNB_BI_IS_ALIVE } HnbBuiltIn;
typedef enum HnbUnaryOp {
    HNB_UN_ABS = 0, HNB_UN_ACOS, HNB_UN_ASIN, HNB_UN_ATAN, HNB_UN_ALL, HNB_UN_ANY, HNB_UN_CEIL, HNB_UN_COS, HNB_UN_EXP, HNB_UN_EXP2,
    HNB_UN_FLOOR, HNB_UN_FRACT, HNB_UN_INV_SQRT, HNB_UN_LENGTH, HNB_UN_LOG, HNB_UN_LOG2, HNB_UN_NORMALIZE, HNB_UN_PACK4X8SNORM,
    HNB_UN_PACK4X8UNORM, HNB_UN_ROUND, HNB_UN_SATURATE, HNB_UN_SIGN, HNB_UN_SIN, HNB_UN_SQRT, HNB_UN_TAN, HNB_UN_UNPACK4X8SNORM,
    HNB_UN_UNPACK4X8UNORM, HNB_UN_W, HNB_UN_X, HNB_UN_Y, HNB_UN_Z
} HnbUnaryOp;
typedef enum HnbBinaryOp {
    HNB_BIN_ADD = 0, HNB_BIN_ATAN2, HNB_BIN_CROSS, HNB_BIN_DISTANCE, HNB_BIN_DIV, HNB_BIN_DOT, HNB_BIN_GREATER_THAN,
    HNB_BIN_GREATER_THAN_OR_EQUAL, HNB_BIN_LESS_THAN, HNB_BIN_LESS_THAN_OR_EQUAL, HNB_BIN_MAX, HNB_BIN_MIN, HNB_BIN_MUL,
    HNB_BIN_REMAINDER, HNB_BIN_STEP, HNB_BIN_SUB, HNB_BIN_UNIFORM_RAND, HNB_BIN_NORMAL_RAND, HNB_BIN_VEC2, HNB_BIN_VEC4_XYZ_W
} HnbBinaryOp;
typedef enum HnbTernaryOp { HNB_TER_MIX = 0, HNB_TER_CLAMP, HNB_TER_SMOOTHSTEP, HNB_TER_VEC3 } HnbTernaryOp;

/* A typed value: `count` (1..4) components of HnbScalarType `scalar_type`, 32 bits each (f32 bit patterns for HNB_F32). */
typedef struct HnbValue {
    uint32_t scalar_type; /* HnbScalarType */
    uint32_t count;
    uint32_t bits[4];
} HnbValue;

const char* hnb_host_last_error(void);
/* Buffers returned by hnb_lower / hnb_asset_serialize are released with this. */
void hnb_host_free(void* p);

/* ---- Module (src/graph/expr.rs:337-778) ------------------------------------------------------------------------- */
int hnb_module_create(HnbModule** out_module);
int hnb_module_destroy(HnbModule* module);
int hnb_module_lit(HnbModule* module, const HnbValue* value, HnbExprHandle* out);
int hnb_module_attr(HnbModule* module, uint32_t attr /* HnbAttr */, HnbExprHandle* out);
int hnb_module_parent_attr(HnbModule* module, uint32_t attr, HnbExprHandle* out);
/* Module::add_property: *out_property is the 1-based PropertyHandle. */
int hnb_module_add_property(HnbModule* module, const char* name, const HnbValue* default_value, uint32_t* out_property);
int hnb_module_prop(HnbModule* module, uint32_t property, HnbExprHandle* out);
/* BuiltInOperator; for HNB_BI_RAND `rand_scalar_type` / `rand_count` give the value type of `rand(T)`. */
int hnb_module_builtin(HnbModule* module, uint32_t op /* HnbBuiltIn */, uint32_t rand_scalar_type, uint32_t rand_count, HnbExprHandle* out);
int hnb_module_unary(HnbModule* module, uint32_t op /* HnbUnaryOp */, HnbExprHandle a, HnbExprHandle* out);
int hnb_module_binary(HnbModule* module, uint32_t op /* HnbBinaryOp */, HnbExprHandle a, HnbExprHandle b, HnbExprHandle* out);
int hnb_module_ternary(HnbModule* module, uint32_t op /* HnbTernaryOp */, HnbExprHandle a, HnbExprHandle b, HnbExprHandle c, HnbExprHandle* out);
int hnb_module_cast(HnbModule* module, HnbExprHandle a, uint32_t scalar_type, uint32_t count, HnbExprHandle* out);
int hnb_module_num_expressions(const HnbModule* module, uint32_t* out);

/* ---- SpawnerSettings / EffectSpawner (src/spawn.rs:82-146, 255-618, 646-922) --------------------------------------- */
/* CpuValue<f32>: Single(a) when !uniform, Uniform((a, b)) otherwise. */
typedef struct HnbCpuValue { float a, b; uint32_t uniform; } HnbCpuValue;
typedef struct HnbSpawnerSettings {
    HnbCpuValue count, spawn_duration, period;
    uint32_t cycle_count;   /* 0 = forever */
    uint32_t starts_active; /* default 1 */
    uint32_t emit_on_start; /* default 1 */
} HnbSpawnerSettings;
/* SpawnerSettings::new: fails (HNB_ERR_INVALID_ARG, the reference panics) on a negative / non-positive / infinite period
 * unless cycle_count == 1. once / rate / burst as the reference's constructors. */
int hnb_spawner_settings_new(HnbCpuValue count, HnbCpuValue spawn_duration, HnbCpuValue period, uint32_t cycle_count, HnbSpawnerSettings* out);
int hnb_spawner_settings_once(HnbCpuValue count, HnbSpawnerSettings* out);
int hnb_spawner_settings_rate(HnbCpuValue rate, HnbSpawnerSettings* out);
int hnb_spawner_settings_burst(HnbCpuValue count, HnbCpuValue period, HnbSpawnerSettings* out);
/* EffectSpawner::new + a Pcg32 (rng_seed / rng_stream: rand_pcg's Pcg32::new(state, stream)); tick returns spawn_count. */
int hnb_spawner_create(const HnbSpawnerSettings* settings, uint64_t rng_seed, uint64_t rng_stream, HnbSpawner** out);
int hnb_spawner_destroy(HnbSpawner* spawner);
int hnb_spawner_tick(HnbSpawner* spawner, float dt, uint32_t* out_spawn_count);
int hnb_spawner_reset(HnbSpawner* spawner);
int hnb_spawner_set_active(HnbSpawner* spawner, int active);

/* Per-frame seed evolution of `compile_effects` (src/lib.rs:1813-1820): the seed handed to hnb_effect_set_frame in frame N+1 is
 * StdRng::seed_from_u64(seed_N as u64).random::<u32>() unless the effect was recompiled. rand's published algorithms (PCG32 seed
 * expansion, ChaCha12), restated and pinned on their known-answer vectors; see hanabi.hpp. */
int hnb_next_prng_seed(uint32_t prng_seed, uint32_t* out_next);

/* ---- EffectAsset (src/asset.rs:272-646) -------------------------------------------------------------------------- */
typedef enum HnbSimulationSpace { HNB_SPACE_GLOBAL = 0, HNB_SPACE_LOCAL = 1 } HnbSimulationSpace;
typedef enum HnbSimulationCondition { HNB_SIM_WHEN_VISIBLE = 0, HNB_SIM_ALWAYS = 1 } HnbSimulationCondition;
typedef enum HnbMotionIntegration { HNB_MOTION_NONE = 0, HNB_MOTION_PRE_UPDATE = 1, HNB_MOTION_POST_UPDATE = 2 } HnbMotionIntegration;
#define HNB_CONTEXT_INIT 1u
#define HNB_CONTEXT_UPDATE 2u
#define HNB_CONTEXT_RENDER 4u

/* One modifier. `e[]` are expression handles of the asset's module; their meaning per kind follows the reference's struct
 * fields in declaration order (0 = absent optional field):
 *   SET_ATTRIBUTE        attribute, e0 = value                                   src/modifier/attr.rs:43-115
 *   INHERIT_ATTRIBUTE    attribute                                               src/modifier/attr.rs:134-186
 *   SET_POSITION_CIRCLE  e0 center, e1 axis, e2 radius, dimension                src/modifier/position.rs:52-108
 *   SET_POSITION_SPHERE  e0 center, e1 radius, dimension                         src/modifier/position.rs:152-210
 *   SET_POSITION_CONE3D  e0 height, e1 base_radius, e2 top_radius, dimension     src/modifier/position.rs:267-324
 *   SET_VELOCITY_CIRCLE  e0 center, e1 axis, e2 speed                            src/modifier/velocity.rs:45-80
 *   SET_VELOCITY_SPHERE  e0 center, e1 speed                                     src/modifier/velocity.rs:124-139
 *   SET_VELOCITY_TANGENT e0 origin, e1 axis, e2 speed                            src/modifier/velocity.rs:188-223
 *   ACCEL                e0 accel                                                src/modifier/accel.rs:79-86
 *   RADIAL_ACCEL         e0 origin, e1 accel                                     src/modifier/accel.rs:162-189
 *   TANGENT_ACCEL        e0 origin, e1 axis, e2 accel                            src/modifier/accel.rs:281-307
 *   LINEAR_DRAG          e0 drag                                                 src/modifier/force.rs:284-297
 *   CONFORM_TO_SPHERE    e0 origin, e1 radius, e2 influence_dist, e3 attraction_accel, e4 max_attraction_speed,
 *                        e5 shell_half_thickness (optional), e6 sticky_factor (optional)   src/modifier/force.rs:70-238
 *   KILL_SPHERE          e0 center, e1 sqr_radius, kill_inside                   src/modifier/kill.rs:76-96
 *   KILL_AABB            e0 center, e1 half_size, kill_inside                    src/modifier/kill.rs:156-181
 *   EMIT_SPAWN_EVENT     condition, e0 count, child_index                        src/modifier/mod.rs:653-717
 *   RENDER               a render-context modifier: only the attributes it adds to the particle layout matter on this path
 *                        (src/modifier/output.rs `attributes()`), listed in render_attrs[0..n_render_attrs) */
typedef enum HnbModifierKind {
    HNB_MOD_SET_ATTRIBUTE = 1, HNB_MOD_INHERIT_ATTRIBUTE, HNB_MOD_SET_POSITION_CIRCLE, HNB_MOD_SET_POSITION_SPHERE, HNB_MOD_SET_POSITION_CONE3D,
    HNB_MOD_SET_VELOCITY_CIRCLE, HNB_MOD_SET_VELOCITY_SPHERE, HNB_MOD_SET_VELOCITY_TANGENT, HNB_MOD_ACCEL, HNB_MOD_RADIAL_ACCEL,
    HNB_MOD_TANGENT_ACCEL, HNB_MOD_LINEAR_DRAG, HNB_MOD_CONFORM_TO_SPHERE, HNB_MOD_KILL_SPHERE, HNB_MOD_KILL_AABB, HNB_MOD_EMIT_SPAWN_EVENT,
    HNB_MOD_RENDER
} HnbModifierKind;
typedef struct HnbModifierDesc {
    uint32_t kind;           /* HnbModifierKind */
    uint32_t attribute;      /* HnbAttr (SET_ATTRIBUTE / INHERIT_ATTRIBUTE) */
    HnbExprHandle e[7];
    uint32_t dimension;      /* ShapeDimension: 0 Surface, 1 Volume */
    uint32_t kill_inside;
    uint32_t condition;      /* EventEmitCondition: 0 Always, 1 OnDie */
    uint32_t child_index;
    uint32_t n_render_attrs;
    uint32_t render_attrs[8];
} HnbModifierDesc;

/* EffectAsset::new(capacity, spawner, module): the module is copied (the reference moves it into the asset). */
int hnb_asset_create(uint32_t capacity, const HnbSpawnerSettings* spawner, const HnbModule* module, HnbAsset** out_asset);
int hnb_asset_destroy(HnbAsset* asset);
int hnb_asset_set_name(HnbAsset* asset, const char* name);
int hnb_asset_set_simulation_space(HnbAsset* asset, uint32_t space /* HnbSimulationSpace, default Global */);
int hnb_asset_set_simulation_condition(HnbAsset* asset, uint32_t condition /* default WhenVisible */);
int hnb_asset_set_motion_integration(HnbAsset* asset, uint32_t integration /* default PostUpdate */);
int hnb_asset_set_prng_seed(HnbAsset* asset, uint32_t seed);
/* EffectAsset::init() / .update() / .render() by `context` (add_modifier, asset.rs:520-546): fails (the reference panics,
 * asset.rs:482,499) when the modifier does not support the context. */
int hnb_asset_add_modifier(HnbAsset* asset, uint32_t context, const HnbModifierDesc* modifier);
/* EffectAsset::particle_layout(): attribute ids in ascending order; *out_count receives the total even when it exceeds `cap`. */
int hnb_asset_particle_layout(const HnbAsset* asset, uint32_t* out_attrs, uint32_t cap, uint32_t* out_count);
/* EffectShaderSources::generate counterpart: the HnbProgram blob for hnb_program_create() / hnb_program_validate(). */
int hnb_lower(const HnbAsset* asset, void** out_blob, size_t* out_size);
/* The reference's on-disk asset format: RON text as EffectAsset::serialize writes and ::deserialize reads it
 * (src/asset.rs:674-716; `.effect` files, loader extension asset.rs:1128-1130). hnb_asset_to_ron returns a NUL-terminated
 * buffer (hnb_host_free); hnb_asset_from_ron fails with HNB_ERR_BAD_PROGRAM and the parser's message (line / column or the
 * offending field) on malformed text, unknown attributes, unknown modifier type paths, out-of-range expression handles. */
int hnb_asset_to_ron(const HnbAsset* asset, char** out_text, size_t* out_size);
int hnb_asset_from_ron(const char* text, size_t size, HnbAsset** out_asset);
/* The WGSL the reference would paste into vfx_init.wgsl / vfx_update.wgsl for this asset (EffectShaderSources::generate, src/lib.rs:
 * 1026-1302: every modifier's `apply`, aging / reaping, Euler integration, write-back) as one NUL-terminated text of `// {{SLOT}}` sections:
 * inspection output a maintainer can diff against the reference's trace log; tests/wgsl_eval executes it. Not used by the simulation. */
int hnb_asset_wgsl(const HnbAsset* asset, int has_parent, char** out_text, size_t* out_size);
/* Flat authoring-level description (expressions, properties, modifiers, settings): the CPU oracle's input format. */
int hnb_asset_serialize(const HnbAsset* asset, void** out_blob, size_t* out_size);

#ifdef __cplusplus
}
#endif
#endif /* HANABI_AMD_HOST_H */
