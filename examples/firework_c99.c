/* The `trails` effect of the reference's examples/firework.rs (lines 187-251) in plain C99, through the two C ABIs only:
 * include/hanabi_amd_host.h (authoring + lowering, libhanabi_host.so) and include/hanabi_amd.h (simulation, libhanabi_amd.so).
 * This is what a binding from another language does — the Rust `extern "C"` block of INTEGRATION.md mirrors these calls.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/firework_c99.c -Lbevy_hanabi_amd -lhanabi_host -lhanabi_amd \
 *       -Wl,-rpath,$PWD/bevy_hanabi_amd -o examples/firework_c99
 *
 * Usage: firework_c99 lower <capacity> <out.blob>            write the lowered program (needs no GPU)
 *        firework_c99 run <capacity> <frames> [<out.f32>]    simulate; prints "frame alive_count", optionally dumps the
 *                                                            POSITION plane (capacity x 3 floats) after the last frame
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hanabi_amd_host.h"

#define HOST(call)                                                                                   \
    do {                                                                                             \
        const int rc_ = (call);                                                                      \
        if (rc_ != HNB_OK) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, hnb_host_last_error()); return 1; } \
    } while (0)
#define DEV(call)                                                                                    \
    do {                                                                                             \
        const int rc_ = (call);                                                                      \
        if (rc_ != HNB_OK) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, hnb_last_error()); return 1; } \
    } while (0)

static HnbValue f32(float x) { HnbValue v; memset(&v, 0, sizeof v); v.scalar_type = HNB_F32; v.count = 1; memcpy(&v.bits[0], &x, 4); return v; }
static HnbValue vec3(float x, float y, float z) {
    HnbValue v; memset(&v, 0, sizeof v); v.scalar_type = HNB_F32; v.count = 3;
    memcpy(&v.bits[0], &x, 4); memcpy(&v.bits[1], &y, 4); memcpy(&v.bits[2], &z, 4);
    return v;
}
static HnbModifierDesc set_attribute(uint32_t attr, HnbExprHandle value) {
    HnbModifierDesc d; memset(&d, 0, sizeof d); d.kind = HNB_MOD_SET_ATTRIBUTE; d.attribute = attr; d.e[0] = value; return d;
}
static HnbModifierDesc one_expr(uint32_t kind, HnbExprHandle e0) { HnbModifierDesc d; memset(&d, 0, sizeof d); d.kind = kind; d.e[0] = e0; return d; }

/* create_trails_effect (examples/firework.rs:187-251), made self-contained as SURVEY.md section 8d C2 does. */
static int build_trails(uint32_t capacity, HnbAsset** out_asset) {
    HnbModule* m = NULL;
    HnbExprHandle zero3, center, lo, hi, speed, r3, two, one, dir, vel, age0, life_lo, life_hi, life, r3b, c09, c01, w1, color, drag, accel, t;
    HnbValue v;
    HnbModifierDesc d;
    HnbSpawnerSettings once;
    HnbCpuValue count;
    HOST(hnb_module_create(&m));
    v = vec3(0.f, 0.f, 0.f); HOST(hnb_module_lit(m, &v, &zero3));                         /* position = the origin (InheritAttribute in the original) */
    HOST(hnb_module_attr(m, HNB_ATTR_POSITION, &center));
    v = f32(40.f); HOST(hnb_module_lit(m, &v, &lo));
    v = f32(60.f); HOST(hnb_module_lit(m, &v, &hi));
    HOST(hnb_module_binary(m, HNB_BIN_UNIFORM_RAND, lo, hi, &speed));                      /* writer.lit(40.).uniform(writer.lit(60.)) */
    HOST(hnb_module_builtin(m, HNB_BI_RAND, HNB_F32, 3, &r3));                             /* writer.rand(VectorType::VEC3F) */
    v = f32(2.f); HOST(hnb_module_lit(m, &v, &two));
    HOST(hnb_module_binary(m, HNB_BIN_MUL, r3, two, &t));
    v = f32(1.f); HOST(hnb_module_lit(m, &v, &one));
    HOST(hnb_module_binary(m, HNB_BIN_SUB, t, one, &t));
    HOST(hnb_module_unary(m, HNB_UN_NORMALIZE, t, &dir));                                  /* (rand * 2 - 1).normalized() */
    HOST(hnb_module_binary(m, HNB_BIN_MUL, dir, speed, &t));
    HOST(hnb_module_binary(m, HNB_BIN_ADD, center, t, &vel));                              /* center + dir * speed */
    v = f32(0.f); HOST(hnb_module_lit(m, &v, &age0));
    v = f32(0.8f); HOST(hnb_module_lit(m, &v, &life_lo));
    v = f32(1.2f); HOST(hnb_module_lit(m, &v, &life_hi));
    HOST(hnb_module_binary(m, HNB_BIN_UNIFORM_RAND, life_lo, life_hi, &life));
    HOST(hnb_module_builtin(m, HNB_BI_RAND, HNB_F32, 3, &r3b));                            /* firework.rs:64-66: pack4x8unorm(vec4(rand3 * 0.9 + 0.1, 1)) */
    v = f32(0.9f); HOST(hnb_module_lit(m, &v, &c09));
    HOST(hnb_module_binary(m, HNB_BIN_MUL, r3b, c09, &t));
    v = f32(0.1f); HOST(hnb_module_lit(m, &v, &c01));
    HOST(hnb_module_binary(m, HNB_BIN_ADD, t, c01, &t));
    v = f32(1.f); HOST(hnb_module_lit(m, &v, &w1));
    HOST(hnb_module_binary(m, HNB_BIN_VEC4_XYZ_W, t, w1, &t));
    HOST(hnb_module_unary(m, HNB_UN_PACK4X8UNORM, t, &color));
    v = vec3(-0.f, -16.f, -0.f); HOST(hnb_module_lit(m, &v, &accel));                     /* Vec3::Y * -16. */
    v = f32(4.f); HOST(hnb_module_lit(m, &v, &drag));

    count.a = count.b = (float)capacity; count.uniform = 0;
    HOST(hnb_spawner_settings_once(count, &once));
    HOST(hnb_asset_create(capacity, &once, m, out_asset));
    HOST(hnb_module_destroy(m));                                                           /* the asset holds its own copy */
    HOST(hnb_asset_set_name(*out_asset, "trail"));
    d = set_attribute(HNB_ATTR_POSITION, zero3); HOST(hnb_asset_add_modifier(*out_asset, HNB_CONTEXT_INIT, &d));
    d = set_attribute(HNB_ATTR_VELOCITY, vel); HOST(hnb_asset_add_modifier(*out_asset, HNB_CONTEXT_INIT, &d));
    d = set_attribute(HNB_ATTR_AGE, age0); HOST(hnb_asset_add_modifier(*out_asset, HNB_CONTEXT_INIT, &d));
    d = set_attribute(HNB_ATTR_LIFETIME, life); HOST(hnb_asset_add_modifier(*out_asset, HNB_CONTEXT_INIT, &d));
    d = set_attribute(HNB_ATTR_COLOR, color); HOST(hnb_asset_add_modifier(*out_asset, HNB_CONTEXT_INIT, &d));
    d = one_expr(HNB_MOD_LINEAR_DRAG, drag); HOST(hnb_asset_add_modifier(*out_asset, HNB_CONTEXT_UPDATE, &d));   /* firework.rs:239-240: drag, then gravity */
    d = one_expr(HNB_MOD_ACCEL, accel); HOST(hnb_asset_add_modifier(*out_asset, HNB_CONTEXT_UPDATE, &d));
    /* render modifiers only add attributes to the layout (ColorOverLifetime / SizeOverLifetime: AGE, LIFETIME; Orient AlongVelocity: POSITION, VELOCITY) */
    memset(&d, 0, sizeof d); d.kind = HNB_MOD_RENDER; d.n_render_attrs = 2; d.render_attrs[0] = HNB_ATTR_AGE; d.render_attrs[1] = HNB_ATTR_LIFETIME;
    HOST(hnb_asset_add_modifier(*out_asset, HNB_CONTEXT_RENDER, &d));
    HOST(hnb_asset_add_modifier(*out_asset, HNB_CONTEXT_RENDER, &d));
    d.render_attrs[0] = HNB_ATTR_POSITION; d.render_attrs[1] = HNB_ATTR_VELOCITY;
    HOST(hnb_asset_add_modifier(*out_asset, HNB_CONTEXT_RENDER, &d));
    return 0;
}

int main(int argc, char** argv) {
    HnbAsset* asset = NULL;
    void* blob = NULL;
    size_t blob_size = 0;
    uint32_t capacity;
    if (argc < 3) { fprintf(stderr, "usage: %s lower <capacity> <out.blob> | run <capacity> <frames> [<positions.f32>]\n", argv[0]); return 2; }
    capacity = (uint32_t)strtoul(argv[2], NULL, 10);
    if (build_trails(capacity, &asset)) return 1;
    HOST(hnb_lower(asset, &blob, &blob_size));                       /* EffectShaderSources::generate (src/lib.rs:805-1336) */
    DEV(hnb_program_validate(blob, blob_size));
    if (strcmp(argv[1], "lower") == 0) {
        FILE* f = fopen(argc > 3 ? argv[3] : "firework.blob", "wb");
        if (!f || fwrite(blob, 1, blob_size, f) != blob_size) { fprintf(stderr, "cannot write the blob\n"); return 1; }
        fclose(f);
        printf("%zu\n", blob_size);
    } else {
        const int frames = argc > 3 ? atoi(argv[3]) : 90;
        const float dt = 1.0f / 60.0f;
        HnbContext* ctx = NULL;
        HnbProgram* prog = NULL;
        HnbEffect* fx = NULL;
        HnbSpawner* spawner = NULL;
        HnbSpawnerSettings once;
        HnbCpuValue count;
        int f;
        count.a = count.b = (float)capacity; count.uniform = 0;
        HOST(hnb_spawner_settings_once(count, &once));
        HOST(hnb_spawner_create(&once, 0xcafef00dd15ea5e5ull, 0xa02bdbf7bb3c0a7ull, &spawner));
        DEV(hnb_ctx_create(0, &ctx));                                 /* fails loudly without a GPU: there is no CPU path */
        DEV(hnb_program_create(ctx, blob, blob_size, &prog));
        DEV(hnb_effect_create(prog, 0, &fx));
        for (f = 0; f < frames; ++f) {
            HnbSimParams sim;
            uint32_t spawn = 0, alive = 0;
            sim.delta_time = sim.virtual_delta_time = sim.real_delta_time = dt;
            sim.time = sim.virtual_time = sim.real_time = (float)f * dt;
            DEV(hnb_frame_begin(ctx, &sim));
            HOST(hnb_spawner_tick(spawner, dt, &spawn));              /* EffectSpawner::tick (src/spawn.rs:838-921) */
            DEV(hnb_effect_set_frame(fx, spawn, 0x9e3779b9u * (uint32_t)(f + 1), NULL));
            DEV(hnb_simulate(ctx));
            DEV(hnb_effect_alive_count(fx, &alive));
            printf("%d %u\n", f, alive);
        }
        if (argc > 4) {
            const size_t bytes = (size_t)capacity * 12;
            float* pos = (float*)malloc(bytes);
            FILE* out = fopen(argv[4], "wb");
            if (!pos || !out) { fprintf(stderr, "cannot dump positions\n"); return 1; }
            DEV(hnb_effect_read_attr(fx, HNB_ATTR_POSITION, pos, bytes));
            fwrite(pos, 1, bytes, out);
            fclose(out);
            free(pos);
        }
        HOST(hnb_spawner_destroy(spawner));
        DEV(hnb_program_destroy(prog));
        DEV(hnb_ctx_destroy(ctx));
    }
    hnb_host_free(blob);
    HOST(hnb_asset_destroy(asset));
    return 0;
}
