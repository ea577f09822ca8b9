/* TEST INFRASTRUCTURE — CPU oracle of bevy_hanabi's GPU simulation path.
 *
 * The reference has no CPU per-particle simulation: its semantics are the
 * WGSL it generates. This file restates those semantics in plain C, executing the kernel
 * threads SERIALLY in increasing global id (SURVEY.md §8c "canonical ordering"), by walking
 * the authoring-level asset (expression graph + modifier list, the flat blob produced by
 * hanabi::serialize_asset) directly — it shares no code with the product's lowering or VM.
 *
 *   PRNG                     src/render/vfx_common.wgsl:260-343
 *   init pass                src/render/vfx_init.wgsl:101-196
 *   indirect / prefix sum    src/render/vfx_indirect.wgsl:30-90, vfx_prefix_sum.wgsl:13-43
 *   update pass              src/render/vfx_update.wgsl:105-167
 *   statement order          src/lib.rs:1026-1133, 1223-1302
 *   expression evaluation    src/graph/expr.rs:1121-1258, 1812-1824
 *   modifier bodies          src/modifier/{attr,position,velocity,accel,force,kill}.rs
 *   literal formatting       src/lib.rs:264-269
 *   CPU spawner              src/spawn.rs:699-717, 814-921
 *
 * PARITY STATUS: integer control-plane behaviour is pinned by the reference's own golden
 * vectors (tests/test_oracle_golden.py). Float results of particle state are UNPINNED: no
 * reference test asserts any particle value, and WGSL builtin accuracy is implementation
 * defined; this oracle defines them through oracle_math.h.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this code.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "oracle_math.h"

#define N_ATTRS 39
enum { T_BOOL = 0, T_F32 = 1, T_I32 = 2, T_U32 = 3 };
enum { A_ID = 0, A_PARTICLE_COUNTER = 1, A_POSITION = 2, A_VELOCITY = 3, A_AGE = 4, A_LIFETIME = 5, A_PREV = 12, A_NEXT = 13, A_RIBBON_ID = 38 };

/* A value of the expression language. `abs` marks a WGSL abstract numeric (a scalar `5.` / `-3` literal of the emitted text, or a
 * constant expression of such literals): b[0] then holds the default concretisation (f32 / i32) and b[1..2] the f64 / i64 payload. */
enum { ABS_NONE = 0, ABS_INT = 1, ABS_FLOAT = 2 };
typedef struct { uint8_t elem, count, abs; uint32_t b[4]; } Val;

/* attribute table: name, elem type, component count (src/attributes.rs:549-675) */
static const struct { const char* name; uint8_t elem, count; } k_attr[N_ATTRS] = {
    {"id", T_U32, 1}, {"particle_counter", T_U32, 1}, {"position", T_F32, 3}, {"velocity", T_F32, 3}, {"age", T_F32, 1},
    {"lifetime", T_F32, 1}, {"color", T_U32, 1}, {"hdr_color", T_F32, 4}, {"alpha", T_F32, 1}, {"size", T_F32, 1}, {"size2", T_F32, 2},
    {"size3", T_F32, 3}, {"prev", T_U32, 1}, {"next", T_U32, 1}, {"axis_x", T_F32, 3}, {"axis_y", T_F32, 3}, {"axis_z", T_F32, 3},
    {"sprite_index", T_I32, 1}, {"f32_0", T_F32, 1}, {"f32_1", T_F32, 1}, {"f32_2", T_F32, 1}, {"f32_3", T_F32, 1},
    {"f32x2_0", T_F32, 2}, {"f32x2_1", T_F32, 2}, {"f32x2_2", T_F32, 2}, {"f32x2_3", T_F32, 2},
    {"f32x3_0", T_F32, 3}, {"f32x3_1", T_F32, 3}, {"f32x3_2", T_F32, 3}, {"f32x3_3", T_F32, 3},
    {"f32x4_0", T_F32, 4}, {"f32x4_1", T_F32, 4}, {"f32x4_2", T_F32, 4}, {"f32x4_3", T_F32, 4},
    {"u32_0", T_U32, 1}, {"u32_1", T_U32, 1}, {"u32_2", T_U32, 1}, {"u32_3", T_U32, 1}, {"ribbon_id", T_U32, 1}};

/* ---- asset blob (hanabi::serialize_asset) ------------------------------------------------ */
enum { EK_BUILTIN = 0, EK_LITERAL, EK_PROPERTY, EK_ATTRIBUTE, EK_PARENT_ATTRIBUTE, EK_UNARY, EK_BINARY, EK_TERNARY, EK_CAST, EK_TEXTURE };
enum { BI_TIME = 0, BI_DT, BI_VTIME, BI_VDT, BI_RTIME, BI_RDT, BI_RAND, BI_ALPHA_CUTOFF, BI_IS_ALIVE };
enum { U_ABS = 0, U_ACOS, U_ASIN, U_ATAN, U_ALL, U_ANY, U_CEIL, U_COS, U_EXP, U_EXP2, U_FLOOR, U_FRACT, U_INVSQRT, U_LENGTH, U_LOG, U_LOG2,
       U_NORMALIZE, U_PACK4X8SNORM, U_PACK4X8UNORM, U_ROUND, U_SATURATE, U_SIGN, U_SIN, U_SQRT, U_TAN, U_UNPACK4X8SNORM, U_UNPACK4X8UNORM,
       U_W, U_X, U_Y, U_Z };
enum { B_ADD = 0, B_ATAN2, B_CROSS, B_DISTANCE, B_DIV, B_DOT, B_GT, B_GE, B_LT, B_LE, B_MAX, B_MIN, B_MUL, B_REM, B_STEP, B_SUB,
       B_UNIFORM_RAND, B_NORMAL_RAND, B_VEC2, B_VEC4_XYZ_W };
enum { TR_MIX = 0, TR_CLAMP, TR_SMOOTHSTEP, TR_VEC3 };
enum { MK_SET_ATTRIBUTE = 1, MK_INHERIT_ATTRIBUTE, MK_SET_POSITION_CIRCLE, MK_SET_POSITION_SPHERE, MK_SET_POSITION_CONE3D,
       MK_SET_VELOCITY_CIRCLE, MK_SET_VELOCITY_SPHERE, MK_SET_VELOCITY_TANGENT, MK_ACCEL, MK_RADIAL_ACCEL, MK_TANGENT_ACCEL, MK_LINEAR_DRAG,
       MK_CONFORM_TO_SPHERE, MK_KILL_SPHERE, MK_KILL_AABB, MK_EMIT_SPAWN_EVENT, MK_RENDER };

typedef struct { uint32_t kind, op, a, b, c, vt_elem, vt_count, bits[4], attr, prop, pad; } ExprRec;      /* 14 words */
typedef struct { char name[48]; uint32_t elem, count, bits[4]; } PropRec;                                  /* 12 + 6 words */
typedef struct { uint32_t kind, attr, e[7], flags, dimension, condition, child_index, n_render, render_attrs[8]; } ModRec; /* 22 words */

typedef struct {
    uint32_t capacity, sim_space, sim_condition, motion_integration, prng_seed;
    uint32_t n_exprs, n_props, n_init, n_update, n_render;
    float count_a, count_b; uint32_t count_uniform;
    float dur_a, dur_b; uint32_t dur_uniform;
    float period_a, period_b; uint32_t period_uniform;
    uint32_t cycle_count, starts_active, emit_on_start;
    ExprRec* exprs;
    PropRec* props;
    ModRec *init, *update, *render;
    uint8_t in_layout[N_ATTRS];
    uint32_t prop_offset[64], prop_words;
    uint8_t* side_effect; /* per expression */
} Asset;

#define MAX_CHANNELS 8   /* == HNB_MAX_EVENT_CHANNELS of include/hanabi_amd.h */
typedef struct Effect_ {
    Asset* asset;
    uint32_t slot_base;
    void* plane[N_ATTRS];        /* packed SoA planes, capacity * count * 4 bytes */
    uint32_t* list[2];           /* alive ping/pong columns */
    uint32_t* dead;              /* dead-slot stack, row alive_count = top */
    uint32_t alive_count, max_update, max_spawn, write_index, particle_counter, instance_count, dead_count, spawned;
    uint32_t props[256];
    char error[256];
    int failed;
    /* GPU spawn events (src/render/event.rs, src/lib.rs:976-993, vfx_init.wgsl:123-129,166-171) */
    int order_by_slot;                      /* alternative canonical schedule: see hor_effect_set_list_order */
    struct Effect_* parent;                 /* EffectParent: this effect's init consumes the parent's events */
    uint32_t parent_channel;                /* the N-th child of a parent reads channel N */
    uint32_t ev_capacity[MAX_CHANNELS];     /* arrayLength(&event_buffer_N.spawn_events); 256 in the reference (event.rs:267) */
    uint32_t* ev_data[MAX_CHANNELS];        /* spawn_events[i].particle_index */
    uint32_t ev_count[MAX_CHANNELS];        /* GpuChildInfo::event_count (keeps growing past the capacity) */
} Effect;

static char g_err[512];
const char* hor_last_error(void) { return g_err; }

void hor_asset_free(Asset* a) {
    if (!a) return;
    free(a->exprs); free(a->props); free(a->init); free(a->update); free(a->render); free(a->side_effect);
    free(a);
}

Asset* hor_asset_parse(const uint8_t* blob, size_t size) {
    const size_t hdr_words = 24;
    if (size < hdr_words * 4) { snprintf(g_err, sizeof g_err, "asset blob too small"); return NULL; }
    const uint32_t* w = (const uint32_t*)blob;
    if (w[0] != 0x31534148u || w[1] != 1u) { snprintf(g_err, sizeof g_err, "bad asset blob magic/version"); return NULL; }
    Asset* a = (Asset*)calloc(1, sizeof(Asset));
    a->capacity = w[2]; a->sim_space = w[3]; a->sim_condition = w[4]; a->motion_integration = w[5]; a->prng_seed = w[6];
    a->n_exprs = w[7]; a->n_props = w[8]; a->n_init = w[9]; a->n_update = w[10]; a->n_render = w[11];
    memcpy(&a->count_a, &w[12], 4); memcpy(&a->count_b, &w[13], 4); a->count_uniform = w[14];
    memcpy(&a->dur_a, &w[15], 4); memcpy(&a->dur_b, &w[16], 4); a->dur_uniform = w[17];
    memcpy(&a->period_a, &w[18], 4); memcpy(&a->period_b, &w[19], 4); a->period_uniform = w[20];
    a->cycle_count = w[21]; a->starts_active = w[22]; a->emit_on_start = w[23];
    const size_t need = hdr_words * 4 + (size_t)a->n_exprs * sizeof(ExprRec) + (size_t)a->n_props * sizeof(PropRec) +
                        (size_t)(a->n_init + a->n_update + a->n_render) * sizeof(ModRec);
    if (need != size || a->n_props > 64) { snprintf(g_err, sizeof g_err, "asset blob size mismatch (%zu vs %zu)", need, size); free(a); return NULL; }
    const uint8_t* p = blob + hdr_words * 4;
    a->exprs = (ExprRec*)malloc(sizeof(ExprRec) * (a->n_exprs + 1)); memcpy(a->exprs, p, sizeof(ExprRec) * a->n_exprs); p += sizeof(ExprRec) * a->n_exprs;
    a->props = (PropRec*)malloc(sizeof(PropRec) * (a->n_props + 1)); memcpy(a->props, p, sizeof(PropRec) * a->n_props); p += sizeof(PropRec) * a->n_props;
    a->init = (ModRec*)malloc(sizeof(ModRec) * (a->n_init + 1)); memcpy(a->init, p, sizeof(ModRec) * a->n_init); p += sizeof(ModRec) * a->n_init;
    a->update = (ModRec*)malloc(sizeof(ModRec) * (a->n_update + 1)); memcpy(a->update, p, sizeof(ModRec) * a->n_update); p += sizeof(ModRec) * a->n_update;
    a->render = (ModRec*)malloc(sizeof(ModRec) * (a->n_render + 1)); memcpy(a->render, p, sizeof(ModRec) * a->n_render);
    /* particle layout: modifiers' attributes + every Expr::Attribute of the module (asset.rs:605-624) */
    for (int pass = 0; pass < 3; ++pass) {
        const ModRec* ms = pass == 0 ? a->init : pass == 1 ? a->update : a->render;
        const uint32_t n = pass == 0 ? a->n_init : pass == 1 ? a->n_update : a->n_render;
        for (uint32_t i = 0; i < n; ++i) {
            const ModRec* m = &ms[i];
            switch (m->kind) {
                case MK_SET_ATTRIBUTE: case MK_INHERIT_ATTRIBUTE: a->in_layout[m->attr] = 1; break;
                case MK_SET_POSITION_CIRCLE: case MK_SET_POSITION_SPHERE: case MK_SET_POSITION_CONE3D: case MK_KILL_SPHERE: case MK_KILL_AABB:
                    a->in_layout[A_POSITION] = 1; break;
                case MK_SET_VELOCITY_CIRCLE: case MK_SET_VELOCITY_SPHERE: case MK_SET_VELOCITY_TANGENT: case MK_RADIAL_ACCEL:
                case MK_TANGENT_ACCEL: case MK_CONFORM_TO_SPHERE:
                    a->in_layout[A_POSITION] = 1; a->in_layout[A_VELOCITY] = 1; break;
                case MK_ACCEL: case MK_LINEAR_DRAG: a->in_layout[A_VELOCITY] = 1; break;
                case MK_RENDER: for (uint32_t k = 0; k < m->n_render && k < 8; ++k) a->in_layout[m->render_attrs[k]] = 1; break;
                default: break;
            }
        }
    }
    for (uint32_t i = 0; i < a->n_exprs; ++i)
        if (a->exprs[i].kind == EK_ATTRIBUTE) a->in_layout[a->exprs[i].attr] = 1;
    a->in_layout[A_ID] = 0; a->in_layout[A_PARTICLE_COUNTER] = 0; /* pseudo attributes are never stored */
    for (uint32_t i = 0; i < a->n_props; ++i) { a->prop_offset[i] = a->prop_words; a->prop_words += a->props[i].count; }
    a->side_effect = (uint8_t*)calloc(a->n_exprs + 1, 1);
    for (uint32_t i = 0; i < a->n_exprs; ++i) {
        const ExprRec* e = &a->exprs[i];
        a->side_effect[i] = (e->kind == EK_BUILTIN && e->op == BI_RAND) || (e->kind == EK_BINARY && (e->op == B_UNIFORM_RAND || e->op == B_NORMAL_RAND));
    }
    return a;
}

/* ---- effect state ---------------------------------------------------------------------------- */
void hor_effect_free(Effect* fx) {
    if (!fx) return;
    for (int i = 0; i < N_ATTRS; ++i) free(fx->plane[i]);
    free(fx->list[0]); free(fx->list[1]); free(fx->dead);
    for (int i = 0; i < MAX_CHANNELS; ++i) free(fx->ev_data[i]);
    free(fx);
}
/* Slab initial state: dead_index[i] = i, alive_count = 0, max_spawn = capacity, write index 0
 * (effect_cache.rs:298-323, render/mod.rs:6048-6070). */
Effect* hor_effect_create(Asset* a, uint32_t slot_base) {
    Effect* fx = (Effect*)calloc(1, sizeof(Effect));
    fx->asset = a;
    fx->slot_base = slot_base;
    for (int i = 0; i < N_ATTRS; ++i)
        if (a->in_layout[i]) fx->plane[i] = calloc((size_t)a->capacity * k_attr[i].count, 4);
    fx->list[0] = (uint32_t*)calloc(a->capacity, 4);
    fx->list[1] = (uint32_t*)calloc(a->capacity, 4);
    fx->dead = (uint32_t*)malloc((size_t)a->capacity * 4);
    for (uint32_t i = 0; i < a->capacity; ++i) fx->dead[i] = i;
    fx->max_spawn = a->capacity;
    for (uint32_t i = 0; i < a->n_props; ++i)
        for (uint32_t c = 0; c < a->props[i].count; ++c) fx->props[a->prop_offset[i] + c] = a->props[i].bits[c];
    return fx;
}
int hor_effect_set_property(Effect* fx, const char* name, const uint32_t* words, uint32_t n) {
    for (uint32_t i = 0; i < fx->asset->n_props; ++i)
        if (strncmp(fx->asset->props[i].name, name, 48) == 0) {
            if (n != fx->asset->props[i].count) return -1;
            memcpy(&fx->props[fx->asset->prop_offset[i]], words, n * 4);
            return 0;
        }
    return -2;
}

/* ---- evaluation context ------------------------------------------------------------------------- */
typedef struct { Val v[N_ATTRS]; } Particle;
#define MAX_EXPRS 4096
typedef struct { Val* val; uint8_t* has; } Memo; /* hoisted `let varN` per ShaderWriter */
/* stack-allocated memo sized by the module's expression count */
#define MEMO_ON_STACK(name, n_exprs)                 \
    Val name##_val[(n_exprs) + 1];                   \
    uint8_t name##_has[(n_exprs) + 1];               \
    memset(name##_has, 0, (n_exprs) + 1);            \
    Memo name##_obj = {name##_val, name##_has};      \
    Memo* name = &name##_obj

typedef struct {
    const Effect* fx;
    Particle* p;
    uint32_t seed;            /* var<private> seed */
    uint32_t particle_index;  /* slot (+slot_base) */
    uint32_t particle_counter;
    int is_init;
    int is_alive;
    int was_alive;            /* AGE_CODE (src/lib.rs:1226-1233) */
    const Particle* parent;   /* parent_particle (READ_PARENT_PARTICLE), init of a child effect only */
    uint32_t ev[MAX_CHANNELS];/* events appended by this thread, per channel (all carry particle_index) */
    const float* sim;         /* [6] */
    const float* xf;          /* [12] row-major 3x4 */
    int failed;
    char* err;
} Ctx;

static void fail(Ctx* c, const char* msg) {
    if (!c->failed) { c->failed = 1; snprintf(c->err, 256, "%s", msg); }
}

static Val mk(uint8_t elem, uint8_t count) { Val v; memset(&v, 0, sizeof v); v.elem = elem; v.count = count; return v; }
static float vf(const Val* v, int i) { return u2f(v->b[i]); }
static void sf(Val* v, int i, float x) { v->b[i] = f2u(x); }
static Val mkf(float x) { Val v = mk(T_F32, 1); sf(&v, 0, x); return v; }

/* `ToWgslString for f32` (src/lib.rs:264-269): 6 decimals, then parsed back by the WGSL front end */
static float round_literal(float x) {
    if (!(x - x == 0.0f)) return x;
    char buf[400];
    snprintf(buf, sizeof buf, "%.6f", (double)x);
    return (float)strtod(buf, NULL);
}

/* ---- WGSL abstract numerics ----------------------------------------------------------------------------
 * `ToWgslString` writes a scalar f32 literal as `5.` / `0.1` and a scalar i32 literal as `-3` (src/lib.rs:264-269, 354-358): in WGSL
 * these are an AbstractFloat and an AbstractInt. The front end (naga's constant evaluator) evaluates expressions whose operands are
 * all abstract in f64 / i64, and converts an abstract operand to the type of the concrete one it meets (AbstractInt -> i32, u32, f32;
 * AbstractFloat -> f32); a `let` without a type concretises to i32 / f32. u32 literals (`3u`), booleans and every vector literal
 * (`vec3<f32>(...)`) are concrete. This is what lets examples/instancing.rs:274 pass `lit(-3)` as an acceleration. */
static int64_t abs_i(const Val* v) { int64_t i; memcpy(&i, &v->b[1], 8); return i; }
static double abs_d(const Val* v) { if (v->abs == ABS_INT) return (double)abs_i(v); double d; memcpy(&d, &v->b[1], 8); return d; }
static Val abs_int(int64_t i) { Val v = mk(T_I32, 1); v.abs = ABS_INT; v.b[0] = (uint32_t)i; memcpy(&v.b[1], &i, 8); return v; }
static Val abs_float(Ctx* c, double d) {
    if (!(d - d == 0.0)) fail(c, "abstract float constant is not finite");
    Val v = mk(T_F32, 1); v.abs = ABS_FLOAT; sf(&v, 0, (float)d); memcpy(&v.b[1], &d, 8); return v;
}
/* conversion of an abstract value to the concrete scalar type `elem` */
static Val conv(Ctx* c, Val v, uint8_t elem) {
    if (!v.abs) return v;
    Val o = mk(elem, 1);
    if (v.abs == ABS_FLOAT) {
        if (elem != T_F32) { fail(c, "an abstract float does not convert to an integer or boolean type"); return o; }
        const float f = (float)abs_d(&v);
        if (!(f - f == 0.0f)) fail(c, "abstract float constant out of range for f32");
        sf(&o, 0, f);
        return o;
    }
    const int64_t i = abs_i(&v);
    switch (elem) {
        case T_F32: sf(&o, 0, (float)i); break;
        case T_I32: if (i < INT32_MIN || i > INT32_MAX) fail(c, "abstract int constant out of range for i32"); o.b[0] = (uint32_t)(int32_t)i; break;
        case T_U32: if (i < 0 || i > (int64_t)UINT32_MAX) fail(c, "abstract int constant out of range for u32"); o.b[0] = (uint32_t)i; break;
        default: fail(c, "an abstract int does not convert to bool"); break;
    }
    return o;
}
/* default concretisation (`let x = 3;` is an i32, `let x = 3.;` an f32) */
static Val conc(Ctx* c, Val v) { return v.abs ? conv(c, v, v.abs == ABS_INT ? T_I32 : T_F32) : v; }
/* operands of one operator or constructor: abstract ones take the concrete one's element type; all abstract: f32 if any is a float, else i32 */
static void unify(Ctx* c, Val** v, int n) {
    int elem = -1, any_float = 0;
    for (int i = 0; i < n; ++i) { if (!v[i]->abs) { if (elem < 0) elem = v[i]->elem; } else if (v[i]->abs == ABS_FLOAT) any_float = 1; }
    if (elem < 0) elem = any_float ? T_F32 : T_I32;
    for (int i = 0; i < n; ++i) *v[i] = conv(c, *v[i], (uint8_t)elem);
}
/* constant evaluation of `l op r` with both operands abstract: i64 when both are ints, f64 otherwise */
static Val abs_arith(Ctx* c, uint32_t op, Val l, Val r);
static int abs_compare(uint32_t op, const Val* l, const Val* r);

/* PRNG (vfx_common.wgsl:278-343) */
static float frand(Ctx* c) { c->seed = pcg_hash(c->seed); return to_float01(pcg_hash(c->seed)); }
static Val frand_n(Ctx* c, int n) {
    Val v = mk(T_F32, (uint8_t)n);
    if (n == 1) { sf(&v, 0, frand(c)); return v; }
    if (n == 4) {
        uint32_t r0 = pcg_hash(c->seed), r1 = pcg_hash(r0), r2 = pcg_hash(r1);
        c->seed = r2;
        sf(&v, 0, to_float01(r0));
        sf(&v, 1, to_float01(((r0 & 0xff000000u) >> 8) | (r1 & 0x0000ffffu)));
        sf(&v, 2, to_float01(((r1 & 0xffff0000u) >> 8) | (r2 & 0x000000ffu)));
        sf(&v, 3, to_float01(r2 >> 8));
        return v;
    }
    for (int i = 0; i < n; ++i) { c->seed = pcg_hash(c->seed); sf(&v, i, to_float01(c->seed)); }
    return v;
}
#define TAU 6.283185307179586476925286766559f

static Val eval(Ctx* c, Memo* memo, uint32_t handle);

static int same_type(const Val* a, const Val* b) { return a->elem == b->elem && a->count == b->count; }

/* component-wise binary arithmetic with WGSL scalar/vector broadcasting */
static Val arith(Ctx* c, uint32_t op, Val l, Val r) {
    if (l.abs && r.abs) return abs_arith(c, op, l, r);
    if (l.abs) l = conv(c, l, r.elem); else if (r.abs) r = conv(c, r, l.elem);
    if (l.elem != r.elem || l.elem == T_BOOL || !(l.count == r.count || l.count == 1 || r.count == 1)) { fail(c, "type error in arithmetic"); return mkf(0); }
    const int n = l.count > r.count ? l.count : r.count;
    Val o = mk(l.elem, (uint8_t)n);
    for (int i = 0; i < n; ++i) {
        const uint32_t xb = l.b[l.count == 1 ? 0 : i], yb = r.b[r.count == 1 ? 0 : i];
        if (l.elem == T_F32) {
            const float x = u2f(xb), y = u2f(yb);
            float z;
            switch (op) {
                case B_ADD: z = x + y; break;
                case B_SUB: z = x - y; break;
                case B_MUL: z = x * y; break;
                case B_DIV: z = x / y; break;
                default: z = f_rem(x, y); break;
            }
            o.b[i] = f2u(z);
        } else if (l.elem == T_I32) {
            const int32_t x = (int32_t)xb, y = (int32_t)yb;
            switch (op) {
                case B_ADD: o.b[i] = xb + yb; break;
                case B_SUB: o.b[i] = xb - yb; break;
                case B_MUL: o.b[i] = xb * yb; break;
                case B_DIV: o.b[i] = (uint32_t)i_div(x, y); break;
                default: o.b[i] = (uint32_t)i_rem(x, y); break;
            }
        } else {
            switch (op) {
                case B_ADD: o.b[i] = xb + yb; break;
                case B_SUB: o.b[i] = xb - yb; break;
                case B_MUL: o.b[i] = xb * yb; break;
                case B_DIV: o.b[i] = u_div(xb, yb); break;
                default: o.b[i] = u_rem(xb, yb); break;
            }
        }
    }
    return o;
}
static Val abs_arith(Ctx* c, uint32_t op, Val l, Val r) {
    if (l.abs == ABS_INT && r.abs == ABS_INT) {
        const int64_t x = abs_i(&l), y = abs_i(&r);
        int64_t z = 0;
        int bad = 0;
        switch (op) {
            case B_ADD: bad = __builtin_add_overflow(x, y, &z); break;
            case B_SUB: bad = __builtin_sub_overflow(x, y, &z); break;
            case B_MUL: bad = __builtin_mul_overflow(x, y, &z); break;
            case B_DIV: if (y == 0 || (x == INT64_MIN && y == -1)) bad = 1; else z = x / y; break;
            default: if (y == 0 || (x == INT64_MIN && y == -1)) bad = 1; else z = x % y; break;
        }
        if (bad) { fail(c, "abstract int constant expression overflows or divides by zero"); return abs_int(0); }
        return abs_int(z);
    }
    const double x = abs_d(&l), y = abs_d(&r);
    switch (op) {
        case B_ADD: return abs_float(c, x + y);
        case B_SUB: return abs_float(c, x - y);
        case B_MUL: return abs_float(c, x * y);
        case B_DIV: return abs_float(c, x / y);
        default: return abs_float(c, fmod(x, y));
    }
}
static int abs_compare(uint32_t op, const Val* l, const Val* r) {
    if (l->abs == ABS_INT && r->abs == ABS_INT) { const int64_t x = abs_i(l), y = abs_i(r); return op == B_GT ? x > y : op == B_GE ? x >= y : op == B_LT ? x < y : x <= y; }
    const double x = abs_d(l), y = abs_d(r);
    return op == B_GT ? x > y : op == B_GE ? x >= y : op == B_LT ? x < y : x <= y;
}
static float dotf(const Val* a, const Val* b) {
    float s = vf(a, 0) * vf(b, 0);
    for (int i = 1; i < a->count; ++i) s = s + vf(a, i) * vf(b, i);
    return s;
}
/* normalize(v) = v * (1 / length(v)): the arithmetic definition of oracle_math.h (WGSL leaves normalize's accuracy to the implementation) */
static Val normalize_v(const Val* a) {
    const float inv = 1.0f / f_sqrt(dotf(a, a));
    Val o = mk(T_F32, a->count);
    for (int i = 0; i < a->count; ++i) sf(&o, i, vf(a, i) * inv);
    return o;
}
static Val cross_v(const Val* a, const Val* b) {
    Val o = mk(T_F32, 3);
    sf(&o, 0, vf(a, 1) * vf(b, 2) - vf(a, 2) * vf(b, 1));
    sf(&o, 1, vf(a, 2) * vf(b, 0) - vf(a, 0) * vf(b, 2));
    sf(&o, 2, vf(a, 0) * vf(b, 1) - vf(a, 1) * vf(b, 0));
    return o;
}
/* transform * vec4(v, 0.0) (vfx_init.wgsl:157-164) */
static Val xform_dir(const float* xf, const Val* v) {
    Val o = mk(T_F32, 3);
    for (int i = 0; i < 3; ++i)
        sf(&o, i, ((xf[4 * i] * vf(v, 0) + xf[4 * i + 1] * vf(v, 1)) + xf[4 * i + 2] * vf(v, 2)) + xf[4 * i + 3] * 0.0f);
    return o;
}

static int ref_value_type(const Asset* a, uint32_t handle, uint8_t* elem, uint8_t* count) {
    /* Expr::value_type() (expr.rs:1084-1098): known only for leaves and casts */
    const ExprRec* e = &a->exprs[handle - 1];
    switch (e->kind) {
        case EK_BUILTIN:
            if (e->op == BI_RAND) { *elem = (uint8_t)e->vt_elem; *count = (uint8_t)e->vt_count; }
            else if (e->op == BI_IS_ALIVE) { *elem = T_BOOL; *count = 1; }
            else { *elem = T_F32; *count = 1; }
            return 1;
        case EK_LITERAL: case EK_CAST: *elem = (uint8_t)e->vt_elem; *count = (uint8_t)e->vt_count; return 1;
        case EK_ATTRIBUTE: case EK_PARENT_ATTRIBUTE: *elem = k_attr[e->attr].elem; *count = k_attr[e->attr].count; return 1;
        default: return 0;
    }
}

static Val eval_node(Ctx* c, Memo* memo, const ExprRec* e) {
    const Asset* a = c->fx->asset;
    switch (e->kind) {
        case EK_LITERAL: {
            if (e->vt_count == 1 && e->vt_elem == T_I32) return abs_int((int32_t)e->bits[0]);
            if (e->vt_count == 1 && e->vt_elem == T_F32 && (u2f(e->bits[0]) - u2f(e->bits[0]) == 0.0f)) {
                char buf[400]; /* the f64 the front end reads back from the 6-decimal text */
                snprintf(buf, sizeof buf, "%.6f", (double)u2f(e->bits[0]));
                return abs_float(c, strtod(buf, NULL));
            }
            Val v = mk((uint8_t)e->vt_elem, (uint8_t)e->vt_count);
            for (int i = 0; i < v.count; ++i) v.b[i] = e->vt_elem == T_F32 ? f2u(round_literal(u2f(e->bits[i]))) : e->bits[i];
            return v;
        }
        case EK_PROPERTY: {
            if (e->prop == 0 || e->prop > a->n_props) { fail(c, "unknown property"); return mkf(0); }
            const PropRec* p = &a->props[e->prop - 1];
            Val v = mk((uint8_t)p->elem, (uint8_t)p->count);
            for (int i = 0; i < v.count; ++i) v.b[i] = c->fx->props[a->prop_offset[e->prop - 1] + i];
            return v;
        }
        case EK_BUILTIN:
            switch (e->op) {
                case BI_TIME: return mkf(c->sim[0]);
                case BI_DT: return mkf(c->sim[1]);
                case BI_VTIME: return mkf(c->sim[2]);
                case BI_VDT: return mkf(c->sim[3]);
                case BI_RTIME: return mkf(c->sim[4]);
                case BI_RDT: return mkf(c->sim[5]);
                case BI_RAND:
                    if (e->vt_elem != T_F32) { fail(c, "only float rand exists in WGSL"); return mkf(0); }
                    return frand_n(c, (int)e->vt_count);
                case BI_IS_ALIVE: {
                    if (c->is_init) { fail(c, "is_alive outside update"); return mkf(0); }
                    Val v = mk(T_BOOL, 1); v.b[0] = c->is_alive ? 1u : 0u; return v;
                }
                default: fail(c, "alpha_cutoff outside render"); return mkf(0);
            }
        case EK_ATTRIBUTE: {
            if (e->attr == A_ID) { Val v = mk(T_U32, 1); v.b[0] = c->particle_index; return v; }
            if (e->attr == A_PARTICLE_COUNTER) {
                if (!c->is_init) { fail(c, "particle_counter outside init"); return mkf(0); }
                Val v = mk(T_U32, 1); v.b[0] = c->particle_counter; return v;
            }
            return c->p->v[e->attr];
        }
        case EK_PARENT_ATTRIBUTE: { /* `parent_particle.<name>` */
            if (!c->is_init || !c->parent) { fail(c, "parent_particle is only defined in the init shader of an effect with a parent"); return mkf(0); }
            if (e->attr == A_ID || e->attr == A_PARTICLE_COUNTER) { fail(c, "pseudo-attribute of the parent particle"); return mkf(0); }
            return c->parent->v[e->attr];
        }
        case EK_UNARY: {
            if (e->op >= U_W && e->op <= U_Z) {
                /* `{expr}.x` after an infix `({l}) op ({r})` prints `(l) op (r).x` (expr.rs:1145-1149, 1164-1176: no parentheses around the
                 * operation): a shader that does not compile, or one that takes the component of the right operand only. Not given a meaning. */
                const ExprRec* in = &c->fx->asset->exprs[e->a - 1];
                if (in->kind == EK_BINARY && !c->fx->asset->side_effect[e->a - 1] &&
                    (in->op == B_ADD || in->op == B_DIV || in->op == B_GT || in->op == B_GE || in->op == B_LT || in->op == B_LE || in->op == B_MUL || in->op == B_REM || in->op == B_SUB)) {
                    fail(c, "component access on an infix expression: the reference prints `(l) op (r).x`");
                    return mkf(0);
                }
            }
            Val x = eval(c, memo, e->a);
            /* abs() / sign() keep an integer an integer; every other builtin of the list only exists for floats
             * (AbstractInt -> AbstractFloat -> f32), or fails below on a scalar */
            if (x.abs) x = (e->op == U_ABS || e->op == U_SIGN || e->op == U_ALL || e->op == U_ANY || (e->op >= U_W && e->op <= U_Z) ||
                            e->op == U_UNPACK4X8SNORM || e->op == U_UNPACK4X8UNORM) ? conc(c, x) : conv(c, x, T_F32);
            if (e->op >= U_W && e->op <= U_Z) {
                const int idx = e->op == U_X ? 0 : e->op == U_Y ? 1 : e->op == U_Z ? 2 : 3;
                if (x.count < 2 || idx >= x.count) { fail(c, "bad component access"); return mkf(0); }
                Val v = mk(x.elem, 1); v.b[0] = x.b[idx]; return v;
            }
            switch (e->op) {
                case U_ALL: case U_ANY: {
                    if (x.elem != T_BOOL) { fail(c, "all/any of non-bool"); return mkf(0); }
                    uint32_t r = e->op == U_ALL ? 1u : 0u;
                    for (int i = 0; i < x.count; ++i) r = e->op == U_ALL ? (r & (x.b[i] != 0)) : (r | (x.b[i] != 0));
                    Val v = mk(T_BOOL, 1); v.b[0] = r; return v;
                }
                case U_LENGTH: if (x.elem != T_F32) { fail(c, "length of non-float"); return mkf(0); } return mkf(f_sqrt(dotf(&x, &x)));
                case U_NORMALIZE: if (x.elem != T_F32 || x.count < 2) { fail(c, "normalize of non-vector"); return mkf(0); } return normalize_v(&x);
                case U_PACK4X8SNORM: case U_PACK4X8UNORM: {
                    if (x.elem != T_F32 || x.count != 4) { fail(c, "pack4x8 of non-vec4"); return mkf(0); }
                    Val v = mk(T_U32, 1);
                    for (int i = 0; i < 4; ++i) v.b[0] |= (e->op == U_PACK4X8UNORM ? pack_unorm8(vf(&x, i)) : pack_snorm8(vf(&x, i))) << (8 * i);
                    return v;
                }
                case U_UNPACK4X8SNORM: case U_UNPACK4X8UNORM: {
                    if (x.elem != T_U32 || x.count != 1) { fail(c, "unpack4x8 of non-u32"); return mkf(0); }
                    Val v = mk(T_F32, 4);
                    for (int i = 0; i < 4; ++i) sf(&v, i, e->op == U_UNPACK4X8UNORM ? unpack_unorm8(x.b[0] >> (8 * i)) : unpack_snorm8(x.b[0] >> (8 * i)));
                    return v;
                }
                default: break;
            }
            Val o = mk(x.elem, x.count);
            for (int i = 0; i < x.count; ++i) {
                if (e->op == U_ABS && x.elem == T_I32) { const int32_t t = (int32_t)x.b[i]; o.b[i] = t < 0 ? 0u - x.b[i] : x.b[i]; continue; }
                if (e->op == U_ABS && x.elem == T_U32) { o.b[i] = x.b[i]; continue; }
                if (e->op == U_SIGN && x.elem == T_I32) { const int32_t t = (int32_t)x.b[i]; o.b[i] = (uint32_t)(t > 0 ? 1 : (t < 0 ? -1 : 0)); continue; }
                if (x.elem != T_F32) { fail(c, "float builtin applied to non-float"); return mkf(0); }
                const float t = vf(&x, i);
                float z;
                switch (e->op) {
                    case U_ABS: z = f_abs(t); break;
                    case U_ACOS: z = f_acos(t); break;
                    case U_ASIN: z = f_asin(t); break;
                    case U_ATAN: z = f_atan(t); break;
                    case U_CEIL: z = f_ceil(t); break;
                    case U_COS: z = f_cos(t); break;
                    case U_EXP: z = f_exp(t); break;
                    case U_EXP2: z = f_exp2(t); break;
                    case U_FLOOR: z = f_floor(t); break;
                    case U_FRACT: z = f_fract(t); break;
                    case U_INVSQRT: z = f_inv_sqrt(t); break;
                    case U_LOG: z = f_log(t); break;
                    case U_LOG2: z = f_log2(t); break;
                    case U_ROUND: z = f_round_even(t); break;
                    case U_SATURATE: z = f_saturate(t); break;
                    case U_SIGN: z = f_sign(t); break;
                    case U_SIN: z = f_sin(t); break;
                    case U_SQRT: z = f_sqrt(t); break;
                    default: z = f_tan(t); break;
                }
                sf(&o, i, z);
            }
            return o;
        }
        case EK_BINARY: {
            if (e->op == B_UNIFORM_RAND || e->op == B_NORMAL_RAND) {
                /* operands first (left, right), then the draw: rand_uniform_T(a, b) (expr.rs:1149-1190) */
                Val l = conc(c, eval(c, memo, e->a));
                Val r = conc(c, eval(c, memo, e->b));
                uint8_t le, lc, re, rc;
                if (!ref_value_type(a, e->a, &le, &lc) || !ref_value_type(a, e->b, &re, &rc)) { fail(c, "Can't determine the type of the operand"); return mkf(0); }
                if (le != re || lc != rc) { fail(c, "Mismatched types"); return mkf(0); }
                if (le != T_F32) { fail(c, "Unsupported type"); return mkf(0); }
                Val o = mk(T_F32, lc);
                if (e->op == B_UNIFORM_RAND) {
                    Val rnd = frand_n(c, lc);
                    for (int i = 0; i < lc; ++i) sf(&o, i, vf(&l, i) + vf(&rnd, i) * (vf(&r, i) - vf(&l, i)));
                } else {
                    const float u = frand(c);
                    Val v = frand_n(c, lc);
                    const float rr = f_sqrt(-2.0f * f_log(u));
                    for (int i = 0; i < lc; ++i) sf(&o, i, vf(&l, i) + vf(&r, i) * rr * f_cos(TAU * vf(&v, i)));
                }
                return o;
            }
            Val l = eval(c, memo, e->a);
            Val r = eval(c, memo, e->b);
            if (l.abs || r.abs) {
                const int both = l.abs && r.abs;
                switch (e->op) {
                    case B_ADD: case B_SUB: case B_MUL: case B_DIV: case B_REM: break; /* arith() */
                    case B_GT: case B_GE: case B_LT: case B_LE:
                        if (both) { Val o = mk(T_BOOL, 1); o.b[0] = abs_compare(e->op, &l, &r) ? 1u : 0u; return o; }
                        { Val* lr[2] = {&l, &r}; unify(c, lr, 2); }
                        break;
                    case B_MAX: case B_MIN:
                        if (both) {
                            if (l.abs == ABS_INT && r.abs == ABS_INT) { const int64_t x = abs_i(&l), y = abs_i(&r); return abs_int(e->op == B_MAX ? (x < y ? y : x) : (y < x ? y : x)); }
                            const double x = abs_d(&l), y = abs_d(&r);
                            return abs_float(c, e->op == B_MAX ? (x < y ? y : x) : (y < x ? y : x));
                        }
                        { Val* lr[2] = {&l, &r}; unify(c, lr, 2); }
                        break;
                    case B_VEC2: case B_VEC4_XYZ_W: { Val* lr[2] = {&l, &r}; unify(c, lr, 2); } break;
                    default: l = conv(c, l, T_F32); r = conv(c, r, T_F32); break; /* step, atan2, cross, dot, distance: float only */
                }
            }
            switch (e->op) {
                case B_ADD: case B_SUB: case B_MUL: case B_DIV: case B_REM: return arith(c, e->op, l, r);
                case B_GT: case B_GE: case B_LT: case B_LE: {
                    if (!same_type(&l, &r) || l.elem == T_BOOL) { fail(c, "type error in comparison"); return mkf(0); }
                    Val o = mk(T_BOOL, l.count);
                    for (int i = 0; i < l.count; ++i) {
                        int t;
                        if (l.elem == T_F32) { const float x = vf(&l, i), y = vf(&r, i); t = e->op == B_GT ? x > y : e->op == B_GE ? x >= y : e->op == B_LT ? x < y : x <= y; }
                        else if (l.elem == T_I32) { const int32_t x = (int32_t)l.b[i], y = (int32_t)r.b[i]; t = e->op == B_GT ? x > y : e->op == B_GE ? x >= y : e->op == B_LT ? x < y : x <= y; }
                        else { const uint32_t x = l.b[i], y = r.b[i]; t = e->op == B_GT ? x > y : e->op == B_GE ? x >= y : e->op == B_LT ? x < y : x <= y; }
                        o.b[i] = t ? 1u : 0u;
                    }
                    return o;
                }
                case B_MAX: case B_MIN: {
                    if (!same_type(&l, &r) || l.elem == T_BOOL) { fail(c, "type error in min/max"); return mkf(0); }
                    Val o = mk(l.elem, l.count);
                    for (int i = 0; i < l.count; ++i) {
                        if (l.elem == T_F32) sf(&o, i, e->op == B_MAX ? f_max(vf(&l, i), vf(&r, i)) : f_min(vf(&l, i), vf(&r, i)));
                        else if (l.elem == T_I32) { const int32_t x = (int32_t)l.b[i], y = (int32_t)r.b[i]; o.b[i] = (uint32_t)(e->op == B_MAX ? (x < y ? y : x) : (y < x ? y : x)); }
                        else { const uint32_t x = l.b[i], y = r.b[i]; o.b[i] = e->op == B_MAX ? (x < y ? y : x) : (y < x ? y : x); }
                    }
                    return o;
                }
                case B_STEP: case B_ATAN2: {
                    if (!same_type(&l, &r) || l.elem != T_F32) { fail(c, "type error in step/atan2"); return mkf(0); }
                    Val o = mk(T_F32, l.count);
                    for (int i = 0; i < l.count; ++i) sf(&o, i, e->op == B_STEP ? f_step(vf(&l, i), vf(&r, i)) : f_atan2(vf(&l, i), vf(&r, i)));
                    return o;
                }
                case B_CROSS:
                    if (l.elem != T_F32 || l.count != 3 || !same_type(&l, &r)) { fail(c, "cross of non-vec3"); return mkf(0); }
                    return cross_v(&l, &r);
                case B_DOT:
                    if (l.elem != T_F32 || l.count < 2 || !same_type(&l, &r)) { fail(c, "dot of non-vector"); return mkf(0); }
                    return mkf(dotf(&l, &r));
                case B_DISTANCE: {
                    if (l.elem != T_F32 || !same_type(&l, &r)) { fail(c, "distance type error"); return mkf(0); }
                    const float t0 = vf(&l, 0) - vf(&r, 0);
                    float s = t0 * t0;
                    for (int i = 1; i < l.count; ++i) { const float t = vf(&l, i) - vf(&r, i); s = s + t * t; }
                    return mkf(f_sqrt(s));
                }
                case B_VEC2: {
                    if (l.count != 1 || r.count != 1 || l.elem != r.elem) { fail(c, "vec2 type error"); return mkf(0); }
                    Val o = mk(l.elem, 2); o.b[0] = l.b[0]; o.b[1] = r.b[0]; return o;
                }
                case B_VEC4_XYZ_W: {
                    if (l.count != 3 || r.count != 1 || l.elem != r.elem) { fail(c, "vec4(xyz,w) type error"); return mkf(0); }
                    Val o = mk(l.elem, 4); o.b[0] = l.b[0]; o.b[1] = l.b[1]; o.b[2] = l.b[2]; o.b[3] = r.b[0]; return o;
                }
                default: fail(c, "unknown binary operator"); return mkf(0);
            }
        }
        case EK_TERNARY: {
            Val x = eval(c, memo, e->a);
            Val y = eval(c, memo, e->b);
            Val z = eval(c, memo, e->c);
            if (x.abs || y.abs || z.abs) {
                if (e->op == TR_MIX || e->op == TR_SMOOTHSTEP) { x = conv(c, x, T_F32); y = conv(c, y, T_F32); z = conv(c, z, T_F32); }
                else { Val* xyz[3] = {&x, &y, &z}; unify(c, xyz, 3); } /* clamp, vec3 */
            }
            switch (e->op) {
                case TR_MIX: {
                    if (!same_type(&x, &y) || x.elem != T_F32 || !(same_type(&x, &z) || (z.elem == T_F32 && z.count == 1))) { fail(c, "mix type error"); return mkf(0); }
                    Val o = mk(T_F32, x.count);
                    for (int i = 0; i < x.count; ++i) sf(&o, i, f_mix(vf(&x, i), vf(&y, i), vf(&z, z.count == 1 ? 0 : i)));
                    return o;
                }
                case TR_CLAMP: {
                    if (!same_type(&x, &y) || !same_type(&x, &z) || x.elem == T_BOOL) { fail(c, "clamp type error"); return mkf(0); }
                    Val o = mk(x.elem, x.count);
                    for (int i = 0; i < x.count; ++i) {
                        if (x.elem == T_F32) sf(&o, i, f_clamp(vf(&x, i), vf(&y, i), vf(&z, i)));
                        else if (x.elem == T_I32) { int32_t t = (int32_t)x.b[i]; const int32_t lo = (int32_t)y.b[i], hi = (int32_t)z.b[i]; t = t < lo ? lo : t; t = hi < t ? hi : t; o.b[i] = (uint32_t)t; }
                        else { uint32_t t = x.b[i]; t = t < y.b[i] ? y.b[i] : t; t = z.b[i] < t ? z.b[i] : t; o.b[i] = t; }
                    }
                    return o;
                }
                case TR_SMOOTHSTEP: {
                    if (!same_type(&x, &y) || !same_type(&x, &z) || x.elem != T_F32) { fail(c, "smoothstep type error"); return mkf(0); }
                    Val o = mk(T_F32, x.count);
                    for (int i = 0; i < x.count; ++i) sf(&o, i, f_smoothstep(vf(&x, i), vf(&y, i), vf(&z, i)));
                    return o;
                }
                default: {
                    if (x.count != 1 || !same_type(&x, &y) || !same_type(&x, &z)) { fail(c, "vec3 type error"); return mkf(0); }
                    Val o = mk(x.elem, 3); o.b[0] = x.b[0]; o.b[1] = y.b[0]; o.b[2] = z.b[0]; return o;
                }
            }
        }
        case EK_CAST: {
            Val x = conc(c, eval(c, memo, e->a));
            const uint8_t te = (uint8_t)e->vt_elem, tc = (uint8_t)e->vt_count;
            if ((tc == 1 && x.count != 1) || (tc > 1 && x.count > 1 && x.count != tc)) { fail(c, "invalid cast"); return mkf(0); }
            Val o = mk(te, tc);
            for (int i = 0; i < tc; ++i) {
                const uint32_t s = x.b[x.count == 1 ? 0 : i];
                uint32_t r = s;
                if (x.elem != te) {
                    if (te == T_F32) r = x.elem == T_I32 ? f2u((float)(int32_t)s) : x.elem == T_U32 ? f2u((float)s) : f2u(s ? 1.0f : 0.0f);
                    else if (te == T_BOOL) r = x.elem == T_F32 ? (u2f(s) != 0.0f) : (s != 0u);
                    else if (x.elem == T_F32) r = te == T_I32 ? (uint32_t)f_to_i32(u2f(s)) : f_to_u32(u2f(s));
                }
                o.b[i] = r;
            }
            return o;
        }
        default: fail(c, "unsupported expression kind in the simulation path"); return mkf(0);
    }
}

/* EvalContext::eval (modifier/mod.rs:309-319): side-effect expressions are evaluated once per
 * writer and reused; everything else is a string that is re-evaluated wherever it is pasted. */
static Val eval(Ctx* c, Memo* memo, uint32_t handle) {
    const Asset* a = c->fx->asset;
    if (handle == 0 || handle > a->n_exprs || handle > MAX_EXPRS) { fail(c, "invalid expression handle"); return mkf(0); }
    if (a->side_effect[handle - 1] && memo->has[handle - 1]) return memo->val[handle - 1];
    Val v = eval_node(c, memo, &a->exprs[handle - 1]);
    if (a->side_effect[handle - 1]) { memo->val[handle - 1] = v; memo->has[handle - 1] = 1; }
    return v;
}

/* A modifier parameter whose text is pasted into an expression of the template (`... * ({speed})`): an abstract value takes the type
 * the expression asks for. want_let(): the template binds it first (`let r = {radius};`), which concretises it to i32 / f32. */
static Val want(Ctx* c, Val v, uint8_t elem, uint8_t count, const char* what) {
    if (v.abs && count == 1) v = conv(c, v, elem);
    v = conc(c, v);
    if (v.elem != elem || v.count != count) { fail(c, what); return mk(elem, count); }
    return v;
}

static Val want_let(Ctx* c, Val v, uint8_t elem, uint8_t count, const char* what) { return want(c, conc(c, v), elem, count, what); }

/* One modifier's emitted statement(s), executed for the current particle. */
static void apply_modifier(Ctx* c, Memo* main, const ModRec* m) {
    Particle* p = c->p;
    const float dt = c->sim[1];
    switch (m->kind) {
        case MK_SET_ATTRIBUTE: { /* attr.rs:92-115: particle.A = <expr>; */
            uint8_t ke, kc; /* the Rust-side check on expressions of known type comes first (attr.rs:97-107) */
            if (ref_value_type(c->fx->asset, m->e[0], &ke, &kc) && (ke != k_attr[m->attr].elem || kc != k_attr[m->attr].count)) { fail(c, "SetAttributeModifier type mismatch"); break; }
            Val v = eval(c, main, m->e[0]);
            p->v[m->attr] = want(c, v, k_attr[m->attr].elem, k_attr[m->attr].count, "SetAttributeModifier type mismatch");
        } break;
        case MK_SET_POSITION_CIRCLE: { /* position.rs:52-108 (function writer) */
            MEMO_ON_STACK(fn, c->fx->asset->n_exprs);
            Val ce = want(c, eval(c, fn, m->e[0]), T_F32, 3, "circle center");
            Val n = want(c, eval(c, fn, m->e[1]), T_F32, 3, "circle axis");
            /* Surface: `let r = {radius};`, Volume: `let r = sqrt(frand()) * ({radius});` (position.rs:68-78) */
            Val radius = (m->dimension == 1 ? want : want_let)(c, eval(c, fn, m->e[2]), T_F32, 1, "circle radius");
            const float sign = f_step(0.0f, vf(&n, 2)) * 2.0f - 1.0f;
            const float a = -1.0f / (sign + vf(&n, 2));
            const float b = vf(&n, 0) * vf(&n, 1) * a;
            const float tangent[3] = {1.0f + sign * vf(&n, 0) * vf(&n, 0) * a, sign * b, -sign * vf(&n, 0)};
            const float bitangent[3] = {b, sign + vf(&n, 1) * vf(&n, 1) * a, -vf(&n, 1)};
            const float r = m->dimension == 1 ? f_sqrt(frand(c)) * (vf(&radius, 0)) : vf(&radius, 0);
            const float theta = frand(c) * TAU;
            const float ct = f_cos(theta), st = f_sin(theta);
            for (int i = 0; i < 3; ++i) { const float dir = tangent[i] * ct + bitangent[i] * st; sf(&p->v[A_POSITION], i, vf(&ce, i) + r * dir); }
        } break;
        case MK_SET_POSITION_SPHERE: { /* position.rs:152-210 */
            MEMO_ON_STACK(fn, c->fx->asset->n_exprs);
            Val ce = want(c, eval(c, fn, m->e[0]), T_F32, 3, "sphere center");
            Val radius = (m->dimension == 1 ? want : want_let)(c, eval(c, fn, m->e[1]), T_F32, 1, "sphere radius"); /* position.rs:167-181 */
            const float r = m->dimension == 1 ? f_pow(frand(c), (float)(1. / 3.)) * (vf(&radius, 0)) : vf(&radius, 0);
            const float theta = frand(c) * TAU;
            const float z = frand(c) * 2.f - 1.f;
            const float phi = f_acos(z);
            const float sinphi = f_sin(phi);
            const float x = sinphi * f_cos(theta);
            const float y = sinphi * f_sin(theta);
            const float dir[3] = {x, y, z};
            for (int i = 0; i < 3; ++i) sf(&p->v[A_POSITION], i, vf(&ce, i) + r * dir[i]);
        } break;
        case MK_SET_POSITION_CONE3D: { /* position.rs:267-324; e = {height, base_radius, top_radius} */
            if (!c->is_init) { fail(c, "transform is not defined in the update shader"); break; }
            MEMO_ON_STACK(fn, c->fx->asset->n_exprs);
            Val h0 = want_let(c, eval(c, fn, m->e[0]), T_F32, 1, "cone height");
            Val rt = want_let(c, eval(c, fn, m->e[2]), T_F32, 1, "cone top radius");
            Val rb = want_let(c, eval(c, fn, m->e[1]), T_F32, 1, "cone base radius");
            const float alpha_h = f_pow(frand(c), (float)(1.0 / 3.0));
            const float h = vf(&h0, 0) * alpha_h;
            const float r0 = vf(&rb, 0) + (vf(&rt, 0) - vf(&rb, 0)) * alpha_h;
            const float alpha_r = f_sqrt(frand(c));
            const float r = r0 * alpha_r;
            const float theta = frand(c) * TAU;
            const float cost = f_cos(theta), sint = f_sin(theta);
            Val pv = mk(T_F32, 3);
            sf(&pv, 0, r * cost); sf(&pv, 1, h); sf(&pv, 2, r * sint);
            p->v[A_POSITION] = xform_dir(c->xf, &pv);
        } break;
        case MK_SET_VELOCITY_CIRCLE: { /* velocity.rs:45-80 */
            if (!c->is_init) { fail(c, "transform is not defined in the update shader"); break; }
            MEMO_ON_STACK(fn, c->fx->asset->n_exprs);
            Val ce = want(c, eval(c, fn, m->e[0]), T_F32, 3, "center");
            Val ax = want(c, eval(c, fn, m->e[1]), T_F32, 3, "axis");
            Val sp = want(c, eval(c, fn, m->e[2]), T_F32, 1, "speed");
            Val delta = arith(c, B_SUB, p->v[A_POSITION], ce);
            const float d = dotf(&delta, &ax);
            Val t = mk(T_F32, 3);
            for (int i = 0; i < 3; ++i) sf(&t, i, vf(&delta, i) - d * vf(&ax, i));
            Val radial = normalize_v(&t);
            Val rv = xform_dir(c->xf, &radial);
            for (int i = 0; i < 3; ++i) sf(&p->v[A_VELOCITY], i, vf(&rv, i) * (vf(&sp, 0)));
        } break;
        case MK_SET_VELOCITY_SPHERE: { /* velocity.rs:124-139 (main writer) */
            Val ce = want(c, eval(c, main, m->e[0]), T_F32, 3, "center");
            Val sp = want(c, eval(c, main, m->e[1]), T_F32, 1, "speed");
            Val d = arith(c, B_SUB, p->v[A_POSITION], ce);
            Val n = normalize_v(&d);
            for (int i = 0; i < 3; ++i) sf(&p->v[A_VELOCITY], i, vf(&n, i) * (vf(&sp, 0)));
        } break;
        case MK_SET_VELOCITY_TANGENT: { /* velocity.rs:188-223 */
            if (!c->is_init) { fail(c, "transform is not defined in the update shader"); break; }
            MEMO_ON_STACK(fn, c->fx->asset->n_exprs);
            Val o = want(c, eval(c, fn, m->e[0]), T_F32, 3, "origin");
            Val ax = want(c, eval(c, fn, m->e[1]), T_F32, 3, "axis");
            Val sp = want(c, eval(c, fn, m->e[2]), T_F32, 1, "speed");
            Val radial = arith(c, B_SUB, p->v[A_POSITION], o);
            Val cr = cross_v(&ax, &radial);
            Val tangent = normalize_v(&cr);
            Val tv = xform_dir(c->xf, &tangent);
            for (int i = 0; i < 3; ++i) sf(&p->v[A_VELOCITY], i, vf(&tv, i) * (vf(&sp, 0)));
        } break;
        case MK_ACCEL: { /* accel.rs:79-86: velocity += (accel) * delta_time; */
            Val a = eval(c, main, m->e[0]);
            Val t = arith(c, B_MUL, a, mkf(dt));
            if (t.elem != T_F32 || !(t.count == 3 || t.count == 1)) { fail(c, "acceleration type"); break; }
            p->v[A_VELOCITY] = arith(c, B_ADD, p->v[A_VELOCITY], t);
        } break;
        case MK_RADIAL_ACCEL:
        case MK_TANGENT_ACCEL: {
            const int radial = m->kind == MK_RADIAL_ACCEL;
            MEMO_ON_STACK(fn, c->fx->asset->n_exprs);
            Memo* w = radial ? fn : main; /* RadialAccel: make_fn writer; TangentAccel: main writer (accel.rs:281-307) */
            /* `position - <origin>`: the origin string is pasted WITHOUT parentheses (accel.rs:176, 291),
             * so an infix `(l) + (r)` / `(l) - (r)` origin parses as `(position - l) +/- r`. */
            const ExprRec* oe = &c->fx->asset->exprs[m->e[0] - 1];
            Val d;
            if (oe->kind == EK_BINARY && (oe->op == B_ADD || oe->op == B_SUB)) {
                Val l = eval(c, w, oe->a);
                Val r = eval(c, w, oe->b);
                d = arith(c, oe->op, arith(c, B_SUB, p->v[A_POSITION], l), r);
            } else {
                Val o = eval(c, w, m->e[0]);
                d = arith(c, B_SUB, p->v[A_POSITION], o);
            }
            d = want(c, d, T_F32, 3, "position - origin");
            Val dir = normalize_v(&d);
            if (!radial) {
                Val ax = want(c, eval(c, w, m->e[1]), T_F32, 3, "axis");
                Val cr = cross_v(&ax, &dir);
                dir = normalize_v(&cr);
            }
            Val acc = want(c, eval(c, w, m->e[radial ? 1 : 2]), T_F32, 1, "acceleration");
            const float s = (vf(&acc, 0)) * dt;
            for (int i = 0; i < 3; ++i) sf(&p->v[A_VELOCITY], i, vf(&p->v[A_VELOCITY], i) + vf(&dir, i) * s);
        } break;
        case MK_LINEAR_DRAG: { /* force.rs:284-297: velocity *= max(0., (1.) - ((drag) * (delta_time))); */
            Val drag = eval(c, main, m->e[0]);
            Val t = arith(c, B_SUB, mkf(1.0f), arith(c, B_MUL, drag, mkf(dt)));
            if (t.elem != T_F32 || t.count != 1) { fail(c, "drag type"); break; }
            const float f = f_max(0.0f, vf(&t, 0));
            for (int i = 0; i < 3; ++i) sf(&p->v[A_VELOCITY], i, vf(&p->v[A_VELOCITY], i) * f);
        } break;
        case MK_CONFORM_TO_SPHERE: { /* force.rs:175-238; e = {origin, radius, influence_dist, attraction_accel, max_attraction_speed, shell, sticky} */
            MEMO_ON_STACK(fn, c->fx->asset->n_exprs);
            Val ce = want(c, eval(c, fn, m->e[0]), T_F32, 3, "origin");
            Val radius = want_let(c, eval(c, fn, m->e[1]), T_F32, 1, "radius");
            Val infl = want_let(c, eval(c, fn, m->e[2]), T_F32, 1, "influence_dist");
            const float shell_half_thickness = (m->flags & 1u) ? vf((Val[]){want_let(c, eval(c, fn, m->e[5]), T_F32, 1, "shell")}, 0) : 0.1f;
            Val maxs = want_let(c, eval(c, fn, m->e[4]), T_F32, 1, "max_attraction_speed");
            Val acc = want_let(c, eval(c, fn, m->e[3]), T_F32, 1, "attraction_accel");
            const float sticky = (m->flags & 2u) ? vf((Val[]){want(c, eval(c, fn, m->e[6]), T_F32, 1, "sticky")}, 0) : 2.0f;
            Val rel_pos = arith(c, B_SUB, ce, p->v[A_POSITION]);
            const float origin_dist = f_sqrt(dotf(&rel_pos, &rel_pos));
            Val origin_dir = normalize_v(&rel_pos);
            const float surface_dist = origin_dist - vf(&radius, 0);
            if (surface_dist > vf(&infl, 0)) break;
            const float cur_radial_speed = dotf(&p->v[A_VELOCITY], &origin_dir);
            const float shell_factor = f_smoothstep(0.f, shell_half_thickness, f_abs(surface_dist));
            const float max_radial_speed = f_sign(surface_dist) * shell_factor * vf(&maxs, 0);
            const float delta_speed = max_radial_speed - cur_radial_speed;
            const float attraction_accel = vf(&acc, 0);
            const float sticky_accel = attraction_accel * sticky;
            const float conforming_accel = f_mix(sticky_accel, attraction_accel, shell_factor);
            const float conforming_delta_speed = dt * conforming_accel;
            const float k = f_sign(delta_speed) * f_min(f_abs(delta_speed), conforming_delta_speed);
            for (int i = 0; i < 3; ++i) sf(&p->v[A_VELOCITY], i, vf(&p->v[A_VELOCITY], i) + k * vf(&origin_dir, i));
        } break;
        case MK_KILL_SPHERE: { /* kill.rs:76-96 */
            Val ce = want(c, eval(c, main, m->e[0]), T_F32, 3, "center");
            Val r2 = want(c, eval(c, main, m->e[1]), T_F32, 1, "sqr_radius");
            Val diff = arith(c, B_SUB, p->v[A_POSITION], ce);
            const float sqr_dist = dotf(&diff, &diff);
            if ((m->flags & 4u) ? (sqr_dist < vf(&r2, 0)) : (sqr_dist > vf(&r2, 0))) c->is_alive = 0;
        } break;
        case MK_KILL_AABB: { /* kill.rs:156-181 */
            Val ce = want(c, eval(c, main, m->e[0]), T_F32, 3, "center");
            Val hs = want(c, eval(c, main, m->e[1]), T_F32, 3, "half_size");
            Val diff = arith(c, B_SUB, p->v[A_POSITION], ce);
            int all_in = 1, any_out = 0;
            for (int i = 0; i < 3; ++i) {
                const float d = f_abs(vf(&diff, i));
                all_in = all_in && (d < vf(&hs, i));
                any_out = any_out || (d > vf(&hs, i));
            }
            if ((m->flags & 4u) ? all_in : any_out) c->is_alive = 0;
        } break;
        case MK_INHERIT_ATTRIBUTE: /* attr.rs:173-186: particle.A = parent_particle.A; */
            if (!c->is_init || !c->parent) { fail(c, "InheritAttributeModifier needs a parent effect"); break; }
            p->v[m->attr] = c->parent->v[m->attr];
            break;
        case MK_EMIT_SPAWN_EVENT: { /* modifier/mod.rs:669-695 */
            Val cnt = conc(c, eval(c, main, m->e[0])); /* `let count = <expr>;` evaluated unconditionally */
            if (cnt.elem != T_U32 || cnt.count != 1) { fail(c, "EmitSpawnEventModifier count must be u32"); break; }
            if (m->child_index >= MAX_CHANNELS) { fail(c, "event channel out of range"); break; }
            const int fire = m->condition == 1 ? (c->was_alive && !c->is_alive) : c->is_alive;  /* OnDie : Always */
            if (fire) c->ev[m->child_index] += cnt.b[0];
        } break;
        case MK_RENDER: break;
        default: fail(c, "unknown modifier"); break;
    }
}

static void load_particle(const Effect* fx, uint32_t slot, Particle* p) {
    for (int a = 0; a < N_ATTRS; ++a) {
        p->v[a] = mk(k_attr[a].elem, k_attr[a].count);
        if (fx->plane[a]) memcpy(p->v[a].b, (const uint32_t*)fx->plane[a] + (size_t)slot * k_attr[a].count, k_attr[a].count * 4u);
    }
}
static void store_particle(Effect* fx, uint32_t slot, const Particle* p, int skip_prev_next) {
    for (int a = 0; a < N_ATTRS; ++a) {
        if (!fx->plane[a]) continue;
        if (skip_prev_next && (a == A_PREV || a == A_NEXT)) continue; /* lib.rs:1266-1281 */
        memcpy((uint32_t*)fx->plane[a] + (size_t)slot * k_attr[a].count, p->v[a].b, k_attr[a].count * 4u);
    }
}

static const float k_identity[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};

static int check_layout(Effect* fx) {
    Asset* a = fx->asset;
    if (!a->in_layout[A_POSITION]) { snprintf(fx->error, sizeof fx->error, "missing POSITION attribute"); return -1; }
    if (a->in_layout[A_AGE] && !a->in_layout[A_LIFETIME]) {
        snprintf(fx->error, sizeof fx->error, "AGE without LIFETIME: reference update shader does not compile");
        return -1;
    }
    return 0;
}

/* EffectParent: `child`'s init pass consumes the spawn events `parent` appends to `channel`
 * (the N-th child of a parent reads channel N). event_capacity = arrayLength(&event_buffer.spawn_events);
 * the reference hard-codes 256 (src/render/event.rs:267). */
int hor_effect_set_parent(Effect* child, Effect* parent, uint32_t channel, uint32_t event_capacity) {
    if (!child || !parent || channel >= MAX_CHANNELS || event_capacity == 0) return -1;
    child->parent = parent;
    child->parent_channel = channel;
    free(parent->ev_data[channel]);
    parent->ev_data[channel] = (uint32_t*)calloc(event_capacity, 4);
    parent->ev_capacity[channel] = event_capacity;
    parent->ev_count[channel] = 0;
    return 0;
}
uint32_t hor_effect_event_count(const Effect* fx, uint32_t channel) { return channel < MAX_CHANNELS ? fx->ev_count[channel] : 0; }
void hor_effect_read_events(const Effect* fx, uint32_t channel, uint32_t* dst) {
    if (channel >= MAX_CHANNELS || !fx->ev_data[channel]) return;
    const uint32_t n = fx->ev_count[channel] < fx->ev_capacity[channel] ? fx->ev_count[channel] : fx->ev_capacity[channel];
    memcpy(dst, fx->ev_data[channel], (size_t)n * 4);
}

/* ---- init pass (vfx_init.wgsl:101-196), threads executed serially in increasing id ----
 * CPU-spawned: `spawn_count` threads survive the caps. With a parent (CONSUME_GPU_SPAWN_EVENTS): one thread per
 * spawn event the parent appended during ITS previous update; `spawn_count` is ignored. Reading an event past the
 * buffer capacity is out of bounds in the reference (event_count keeps growing past it): clamped here. */
int hor_effect_init_pass(Effect* fx, const float* sim, uint32_t spawn_count, uint32_t seed, const float* xf) {
    Asset* a = fx->asset;
    if (!xf) xf = k_identity;
    if (check_layout(fx)) return -1;
    int failed = 0;
    const Effect* par = fx->parent;
    const uint32_t* events = NULL;
    if (par) {
        const uint32_t ch = fx->parent_channel;
        spawn_count = par->ev_count[ch] < par->ev_capacity[ch] ? par->ev_count[ch] : par->ev_capacity[ch];
        events = par->ev_data[ch];
    }
    const uint32_t alive0 = fx->alive_count;
    const uint32_t max_spawn = fx->max_spawn;                 /* constant during the pass */
    const uint32_t wi = fx->write_index;
    const uint32_t n_spawn = spawn_count < max_spawn ? spawn_count : max_spawn;
#pragma omp parallel for schedule(static) reduction(| : failed)
    for (uint32_t i = 0; i < n_spawn; ++i) {
        const uint32_t alive_index = alive0 + i;              /* atomicAdd(alive_count, 1) under serial order */
        const uint32_t slot = fx->dead[alive_index];          /* slab offset 0 */
        Particle p;
        for (int k = 0; k < N_ATTRS; ++k) p.v[k] = mk(k_attr[k].elem, k_attr[k].count); /* var particle = Particle(); */
        char err[256];
        Ctx c;
        memset(&c, 0, sizeof c);
        c.fx = fx; c.p = &p; c.particle_index = slot + fx->slot_base; c.particle_counter = fx->particle_counter + i;
        c.is_init = 1; c.is_alive = 1; c.was_alive = 1; c.sim = sim; c.xf = xf; c.err = err;
        c.seed = pcg_hash(c.particle_index ^ seed);
        Particle parent_particle;
        if (par) {                                            /* vfx_init.wgsl:166-171 */
            load_particle(par, events[i], &parent_particle);
            c.parent = &parent_particle;
        }
        MEMO_ON_STACK(main, a->n_exprs);
        for (uint32_t k = 0; k < a->n_init; ++k) apply_modifier(&c, main, &a->init[k]);
        if (fx->plane[A_PREV]) p.v[A_PREV].b[0] = 0xffffffffu;
        if (fx->plane[A_NEXT]) p.v[A_NEXT].b[0] = 0xffffffffu;
        if (a->sim_space == 0 && !par) /* Global, CPU-spawned only: particle.position += transform[3].xyz; (vfx_init.wgsl:183-189) */
            for (int k = 0; k < 3; ++k) sf(&p.v[A_POSITION], k, vf(&p.v[A_POSITION], k) + xf[4 * k + 3]);
        fx->list[wi][alive_index] = slot;
        store_particle(fx, slot, &p, 0);
        if (c.failed) {
            failed |= 1;
#pragma omp critical
            snprintf(fx->error, sizeof fx->error, "%s", err);
        }
    }
    fx->alive_count = alive0 + n_spawn;
    fx->particle_counter += n_spawn;
    fx->spawned = n_spawn;
    fx->failed |= failed;
    return failed ? -1 : 0;
}

/* ---- ribbon sort (vfx_sort_fill.wgsl, vfx_sort.wgsl, vfx_sort_copy.wgsl; scheduled after the update pass for
 * effects whose layout has RIBBON_ID, src/render/mod.rs:4599-4618,7372-7612) ----
 * fill: pairs[i] = {key = RIBBON_ID, key2 = AGE bits, value = particle_index} in alive-list order of the column the
 * update just wrote (atomicAdd order == thread order under serial execution); sort: insertion sort that moves an
 * element only past STRICTLY greater ones, i.e. a stable ascending sort by (key, key2); copy: values back into the
 * same column. Restated as a stable merge sort (same result, O(n log n)). Without AGE the reference reads key2 out
 * of bounds (sort_key2_offset = u32::MAX, mod.rs:6042-6046): taken as 0 here. */
typedef struct { uint32_t key, key2, value; } SortPair;
static int pair_greater(const SortPair* x, const SortPair* y) { return x->key > y->key || (x->key == y->key && x->key2 > y->key2); }
static void sort_ribbons(Effect* fx) {
    const uint32_t n = fx->alive_count;
    if (n < 2) return;
    uint32_t* col = fx->list[fx->write_index];
    SortPair* a = (SortPair*)malloc((size_t)n * sizeof(SortPair));
    SortPair* b = (SortPair*)malloc((size_t)n * sizeof(SortPair));
    const uint32_t* rid = (const uint32_t*)fx->plane[A_RIBBON_ID];
    const uint32_t* age = (const uint32_t*)fx->plane[A_AGE];
    for (uint32_t i = 0; i < n; ++i) { a[i].value = col[i]; a[i].key = rid[col[i]]; a[i].key2 = age ? age[col[i]] : 0u; }
    for (uint32_t w = 1; w < n; w *= 2) {
        for (uint32_t lo = 0; lo < n; lo += 2 * w) {
            const uint32_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            uint32_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) b[k++] = pair_greater(&a[i], &a[j]) ? a[j++] : a[i++];  /* ties keep the left element first */
            while (i < mid) b[k++] = a[i++];
            while (j < hi) b[k++] = a[j++];
        }
        SortPair* t = a; a = b; b = t;
    }
    for (uint32_t i = 0; i < n; ++i) col[i] = a[i].value;
    free(a); free(b);
}

/* The reference's alive-list order is whatever its atomics produce (vfx_update.wgsl:161-165): any permutation of
 * the survivors is a legal outcome. The default canonical schedule here is "threads in increasing global id" (stable
 * compaction). order_by_slot selects a second deterministic schedule in which the update pass
 * leaves the survivors in increasing SLOT order: equally legal, and the one the product offers for steady-state
 * effects because it keeps memory accesses coalesced (HNB_LIST_ORDER_SLOT). The dead-slot stack is left in increasing
 * slot order as well (any order of the casualties is a legal outcome of the reference's atomics), so that spawns fill the
 * lowest free slots. Threads of a pass still run in list-row order. */
void hor_effect_set_list_order(Effect* fx, int by_slot) { fx->order_by_slot = by_slot; }
static int cmp_u32(const void* x, const void* y) { const uint32_t a = *(const uint32_t*)x, b = *(const uint32_t*)y; return a < b ? -1 : a > b; }

/* ---- indirect (vfx_indirect.wgsl:38-85) + update (vfx_update.wgsl:105-167) ---- */
int hor_effect_update_pass(Effect* fx, const float* sim, uint32_t seed, const float* xf) {
    Asset* a = fx->asset;
    if (!xf) xf = k_identity;
    if (check_layout(fx)) return -1;
    const int has_age = a->in_layout[A_AGE];
    int failed = 0;

    /* vfx_indirect.wgsl:38-46: the events of the previous frame have been consumed by the children's init */
    for (int ch = 0; ch < MAX_CHANNELS; ++ch) fx->ev_count[ch] = 0;
    /* vfx_indirect.wgsl:57-85 */
    fx->instance_count = 0;
    fx->max_update = fx->alive_count;
    fx->max_spawn = a->capacity - fx->alive_count;
    fx->write_index = 1u - fx->write_index;

    const uint32_t n = fx->max_update;
    const uint32_t write_index = fx->write_index, read_index = 1u - write_index;
    uint8_t* alive_flag = (uint8_t*)malloc(n ? n : 1);
    uint32_t* ev = (uint32_t*)calloc((size_t)(n ? n : 1) * MAX_CHANNELS, 4);
    const int euler = a->motion_integration != 0 && a->in_layout[A_POSITION] && a->in_layout[A_VELOCITY];
#pragma omp parallel for schedule(static) reduction(| : failed)
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t slot = fx->list[read_index][i];
        Particle p;
        load_particle(fx, slot, &p);
        char err[256];
        Ctx c;
        memset(&c, 0, sizeof c);
        c.fx = fx; c.p = &p; c.particle_index = slot + fx->slot_base; c.is_alive = 1; c.was_alive = 1; c.sim = sim; c.xf = xf; c.err = err;
        c.seed = pcg_hash(c.particle_index ^ seed);
        /* AGE_CODE / REAP_CODE (lib.rs:1223-1258) */
        if (has_age) {
            c.was_alive = vf(&p.v[A_AGE], 0) < vf(&p.v[A_LIFETIME], 0);
            sf(&p.v[A_AGE], 0, vf(&p.v[A_AGE], 0) + sim[1]);
            c.is_alive = vf(&p.v[A_AGE], 0) < vf(&p.v[A_LIFETIME], 0);
            c.is_alive = c.is_alive && (vf(&p.v[A_AGE], 0) < vf(&p.v[A_LIFETIME], 0));
        }
        if (euler && a->motion_integration == 1) /* PreUpdate */
            for (int k = 0; k < 3; ++k) sf(&p.v[A_POSITION], k, vf(&p.v[A_POSITION], k) + vf(&p.v[A_VELOCITY], k) * sim[1]);
        MEMO_ON_STACK(main, a->n_exprs);
        for (uint32_t k = 0; k < a->n_update; ++k) apply_modifier(&c, main, &a->update[k]);
        if (euler && a->motion_integration == 2) /* PostUpdate */
            for (int k = 0; k < 3; ++k) sf(&p.v[A_POSITION], k, vf(&p.v[A_POSITION], k) + vf(&p.v[A_VELOCITY], k) * sim[1]);
        store_particle(fx, slot, &p, 1);
        alive_flag[i] = (uint8_t)c.is_alive;
        for (int ch = 0; ch < MAX_CHANNELS; ++ch) ev[(size_t)i * MAX_CHANNELS + ch] = c.ev[ch];
        if (c.failed) {
            failed |= 1;
#pragma omp critical
            snprintf(fx->error, sizeof fx->error, "%s", err);
        }
    }
    /* list rebuild and event append, serial thread order (vfx_update.wgsl:148-166, src/lib.rs:976-993) */
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t slot = fx->list[read_index][i];
        for (int ch = 0; ch < MAX_CHANNELS; ++ch) {
            const uint32_t count = ev[(size_t)i * MAX_CHANNELS + ch];
            if (count == 0u) continue;
            const uint32_t capacity = fx->ev_capacity[ch];                  /* no consumer bound: capacity 0, nothing stored */
            const uint32_t prev = fx->ev_count[ch];                         /* atomicAdd(event_count, count) */
            fx->ev_count[ch] = prev + count;
            const uint32_t base = prev < capacity ? prev : capacity;
            const uint32_t capped = count < capacity - base ? count : capacity - base;
            for (uint32_t k = 0; k < capped; ++k) fx->ev_data[ch][base + k] = slot;   /* particle_index (slab-relative) */
        }
        if (!alive_flag[i]) {
            const uint32_t alive_index = --fx->alive_count;   /* atomicSub(alive_count, 1) - 1 */
            fx->dead[alive_index] = slot;
            fx->max_spawn += 1;
        } else {
            fx->list[write_index][fx->instance_count++] = slot;
        }
    }
    free(alive_flag);
    free(ev);
    fx->dead_count = n - fx->alive_count;
    fx->failed |= failed;
    if (fx->order_by_slot && !a->in_layout[A_RIBBON_ID] && !failed) {   /* see hor_effect_set_list_order; ribbon effects are re-sorted below anyway */
        uint32_t* col = fx->list[fx->write_index];
        qsort(col, fx->alive_count, 4, cmp_u32);
        /* ... and the free slots in increasing order on the dead rows [alive_count, capacity): spawn i of the next
         * init pass pops dead[alive_count + i], i.e. the lowest free slots first */
        if (fx->spawned || fx->dead_count) qsort(fx->dead + fx->alive_count, a->capacity - fx->alive_count, 4, cmp_u32);
    }
    if (a->in_layout[A_RIBBON_ID] && !failed) sort_ribbons(fx);
    return failed ? -1 : 0;
}

/* One frame of a stand-alone effect: init -> indirect -> update. Systems with parent/child links run
 * every effect's init pass (parents first), then every effect's update pass (src/render/mod.rs:6975-7370). */
int hor_effect_step(Effect* fx, const float* sim, uint32_t spawn_count, uint32_t seed, const float* xf) {
    if (hor_effect_init_pass(fx, sim, spawn_count, seed, xf)) return -1;
    return hor_effect_update_pass(fx, sim, seed, xf);
}

/* ---- readback ---------------------------------------------------------------------------------- */
uint32_t hor_effect_alive_count(const Effect* fx) { return fx->alive_count; }
void hor_effect_counters(const Effect* fx, uint32_t* out8) {
    out8[0] = fx->asset->capacity; out8[1] = fx->alive_count; out8[2] = fx->max_update; out8[3] = fx->max_spawn;
    out8[4] = fx->write_index; out8[5] = fx->particle_counter; out8[6] = fx->instance_count; out8[7] = fx->dead_count;
}
const char* hor_effect_error(const Effect* fx) { return fx->error; }
int hor_effect_attr_components(const Effect* fx, uint32_t attr) { return (attr < N_ATTRS && fx->plane[attr]) ? k_attr[attr].count : 0; }
int hor_effect_read_attr(const Effect* fx, uint32_t attr, void* dst) {
    if (attr >= N_ATTRS || !fx->plane[attr]) return -1;
    memcpy(dst, fx->plane[attr], (size_t)fx->asset->capacity * k_attr[attr].count * 4u);
    return 0;
}
/* a host write of a whole attribute plane (the counterpart of hnb_effect_write_attr: tests change particle state behind the product's bookkeeping) */
int hor_effect_write_attr(Effect* fx, uint32_t attr, const void* src) {
    if (attr >= N_ATTRS || !fx->plane[attr]) return -1;
    memcpy(fx->plane[attr], src, (size_t)fx->asset->capacity * k_attr[attr].count * 4u);
    return 0;
}
/* alive list as the NEXT frame reads it (column written by the last update) */
void hor_effect_read_alive_list(const Effect* fx, uint32_t* dst) { memcpy(dst, fx->list[fx->write_index], (size_t)fx->alive_count * 4u); }
/* free slots, from the top of the stack (row alive_count) */
void hor_effect_read_dead_list(const Effect* fx, uint32_t* dst) { memcpy(dst, fx->dead + fx->alive_count, (size_t)(fx->asset->capacity - fx->alive_count) * 4u); }
int hor_omp_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- control-plane kernels restated for the reference's golden vectors --------------------------- */
/* vfx_indirect.wgsl:30-90, one "thread" per effect. meta rows: {capacity, alive_count, max_update, max_spawn, indirect_write_index} */
void hor_k2_indirect(uint32_t num_effects, uint32_t* meta, uint32_t meta_stride, uint32_t* prefix_sum, uint32_t* instance_count, uint32_t* render_pong) {
    for (uint32_t i = 0; i < num_effects; ++i) {
        uint32_t* em = meta + (size_t)i * meta_stride;
        instance_count[i] = 0u;
        const uint32_t capacity = em[0], alive_count = em[1];
        prefix_sum[i] = alive_count;
        em[2] = alive_count;
        em[3] = capacity - alive_count;
        const uint32_t pong = 1u - em[4];
        em[4] = pong;
        render_pong[i] = pong;
    }
}
/* vfx_prefix_sum.wgsl:13-43, one "thread" per batch: exclusive scan + dispatch args */
void hor_k3_prefix_sum(uint32_t num_batches, const uint32_t* batch_offset, const uint32_t* batch_count, uint32_t* prefix_sum,
                       uint32_t* total_update_count, uint32_t* dispatch_x) {
    for (uint32_t b = 0; b < num_batches; ++b) {
        uint32_t sum = 0;
        for (uint32_t i = batch_offset[b]; i < batch_offset[b] + batch_count[b]; ++i) {
            const uint32_t count = prefix_sum[i];
            prefix_sum[i] = sum;
            sum += count;
        }
        total_update_count[b] = sum;
        dispatch_x[b] = (sum + 63u) >> 6;
    }
}
/* find_location_from_particle (vfx_update.wgsl:51-72); out = {effect_index, base_particle, update_index} */
void hor_find_location(const uint32_t* prefix_sum, uint32_t prefix_sum_offset, uint32_t prefix_sum_count, uint32_t update_particle_index, uint32_t* out3) {
    uint32_t lo = prefix_sum_offset, hi = lo + prefix_sum_count;
    int num_iter = 0;
    while (lo < hi) {
        const uint32_t mid = (hi + lo) >> 1;
        const uint32_t base_particle = prefix_sum[mid];
        if (update_particle_index >= base_particle) lo = mid + 1u;
        else hi = mid;
        if (++num_iter >= 100) { out3[0] = out3[1] = out3[2] = 0xDEADBEEFu; return; }
    }
    const uint32_t base_particle = prefix_sum[lo - 1u];
    out3[0] = lo - 1u - prefix_sum_offset;
    out3[1] = base_particle;
    out3[2] = update_particle_index - base_particle;
}

/* ---- EffectSpawner (spawn.rs:699-717, 814-921), CpuValue::Single only ------------------------------ */
typedef struct {
    float count, spawn_duration, period;
    uint32_t cycle_count;
    int active;
    float cycle_time, sampled_spawn_duration, sampled_period, sampled_count, spawn_remainder;
    uint32_t completed_cycle_count, spawn_count;
} Spawner;
void hor_spawner_init(Spawner* s, float count, float spawn_duration, float period, uint32_t cycle_count, int starts_active, int emit_on_start) {
    memset(s, 0, sizeof *s);
    s->count = count; s->spawn_duration = spawn_duration; s->period = period; s->cycle_count = cycle_count;
    s->completed_cycle_count = (emit_on_start || cycle_count == 0) ? 0u : cycle_count;
    s->active = starts_active;
}
void hor_spawner_reset(Spawner* s) {
    s->cycle_time = 0; s->completed_cycle_count = 0; s->sampled_spawn_duration = 0; s->sampled_period = 0; s->sampled_count = 0;
    s->spawn_count = 0; s->spawn_remainder = 0;
}
uint32_t hor_spawner_tick(Spawner* s, float dt) {
    const int forever = s->cycle_count == 0, once = s->cycle_count == 1;
    if (!s->active || (!forever && s->completed_cycle_count >= s->cycle_count)) { s->spawn_count = 0; return 0; }
    for (;;) {
        if (s->sampled_period == 0.0f) {
            if (once) {
                s->sampled_spawn_duration = s->spawn_duration;
                s->sampled_period = fmaxf(s->sampled_spawn_duration, 1e-12f);
            } else {
                s->sampled_period = s->period;
                s->sampled_spawn_duration = fminf(fmaxf(s->spawn_duration, 0.0f), s->sampled_period);
            }
            s->sampled_spawn_duration = s->spawn_duration; /* spawn.rs:867 */
            s->sampled_count = fmaxf(s->count, 0.0f);
        }
        const float new_time = s->cycle_time + dt;
        if (s->cycle_time <= s->sampled_spawn_duration) {
            if (s->sampled_spawn_duration < fmaxf(1e-5f, dt / 100.0f)) s->spawn_remainder += s->sampled_count;
            else {
                float ratio = (fminf(new_time, s->sampled_spawn_duration) - s->cycle_time) / s->sampled_spawn_duration;
                ratio = fminf(fmaxf(ratio, 0.0f), 1.0f);
                s->spawn_remainder += s->sampled_count * ratio;
            }
        }
        s->cycle_time = new_time;
        if (s->cycle_time >= s->sampled_period) {
            dt = s->cycle_time - s->sampled_period;
            s->cycle_time = 0.0f;
            s->completed_cycle_count += 1;
            s->sampled_period = 0.0f;
            if (!forever && s->completed_cycle_count >= s->cycle_count) break;
        } else break;
    }
    const float count = floorf(s->spawn_remainder);
    s->spawn_remainder -= count;
    s->spawn_count = count <= 0.0f ? 0u : (count >= 4294967296.0f ? 0xffffffffu : (uint32_t)count);
    return s->spawn_count;
}

/* PCG / rand KATs */
uint32_t hor_pcg_hash(uint32_t x) { return pcg_hash(x); }
float hor_to_float01(uint32_t u) { return to_float01(u); }
void hor_frand_kat(uint32_t seed_in, uint32_t* state_out, float* out4, int which) {
    Effect fake; memset(&fake, 0, sizeof fake);
    char err[256];
    Ctx c;
    memset(&c, 0, sizeof c);
    c.fx = &fake; c.seed = seed_in; c.is_init = 1; c.is_alive = 1; c.err = err;
    if (which == 1) { out4[0] = frand(&c); }
    else { Val v = frand_n(&c, which); for (int i = 0; i < which; ++i) out4[i] = vf(&v, i); }
    *state_out = c.seed;
}
float hor_round_literal(float x) { return round_literal(x); }
float hor_math1(int fn, float x) {
    switch (fn) {
        case 0: return f_sin(x); case 1: return f_cos(x); case 2: return f_tan(x); case 3: return f_exp(x); case 4: return f_log(x);
        case 5: return f_log2(x); case 6: return f_atan(x); case 7: return f_asin(x); case 8: return f_acos(x); case 9: return f_exp2(x);
        case 10: return f_sqrt(x); case 11: return f_inv_sqrt(x); default: return x;
    }
}
/* the one binary64 kernel left (sin / cos of |x| > 65536) BEFORE its rounding to binary32 (tests/test_math.py measures it against libm) */
double hor_math1d(int fn, double x) {
    double s, c;
    d_sincos(x, &s, &c);
    return fn == 0 ? s : (fn == 1 ? c : x);
}
float hor_math2(int fn, float x, float y) { return fn == 0 ? f_pow(x, y) : (fn == 1 ? f_atan2(x, y) : f_rem(x, y)); }
