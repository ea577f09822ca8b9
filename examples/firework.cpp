// The `trails` effect of the reference's examples/firework.rs (lines 187-251), written against the C++ mirror of the
// authoring API (bevy_hanabi_amd/csrc/host/hanabi.hpp) and run through the C ABI (include/hanabi_amd.h) - what a host
// application links: no Python anywhere. Build (see tests/test_examples.py):
//   g++ -std=c++17 -O2 -Iinclude -Ibevy_hanabi_amd/csrc/host examples/firework.cpp bevy_hanabi_amd/csrc/host/{hanabi,lowering,wgsl}.cpp \
//       -Lbevy_hanabi_amd -lhanabi_amd -Wl,-rpath,$PWD/bevy_hanabi_amd -o examples/firework
// Usage: firework [capacity] [frames]     prints "frame alive_count" per frame.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "hanabi.hpp"
#include "hanabi_amd.h"

using namespace hanabi;

static EffectAsset create_trails_effect(uint32_t capacity) {
    ExprWriter writer;
    // firework.rs:192-193 inherits the position of the exploding rocket; standalone: the origin
    const Modifier init_pos = SetAttributeModifier(Attribute::POSITION, writer.lit(Vec3{0.f, 0.f, 0.f}).expr());
    // firework.rs:197-204: velocity = center + normalize(rand3 * 2 - 1) * uniform(40, 60)
    const WriterExpr center = writer.attr(Attribute::POSITION);
    const WriterExpr speed = writer.lit(40.f).uniform(writer.lit(60.f));
    const WriterExpr dir = (writer.rand(VectorType::VEC3F) * writer.lit(2.f) - writer.lit(1.f)).normalized();
    const Modifier init_vel = SetAttributeModifier(Attribute::VELOCITY, (center + dir * speed).expr());
    const Modifier init_age = SetAttributeModifier(Attribute::AGE, writer.lit(0.f).expr());
    const Modifier init_lifetime = SetAttributeModifier(Attribute::LIFETIME, writer.lit(0.8f).uniform(writer.lit(1.2f)).expr());
    // firework.rs:64-66: colour = pack4x8unorm(vec4(rand3 * 0.9 + 0.1, 1))
    const WriterExpr color = (writer.rand(VectorType::VEC3F) * writer.lit(0.9f) + writer.lit(0.1f)).vec4_xyz_w(writer.lit(1.f)).pack4x8unorm();
    const Modifier init_color = SetAttributeModifier(Attribute::COLOR, color.expr());
    // firework.rs:239-240: drag 4, then gravity (Vec3::Y * -16. is (-0, -16, -0))
    const Modifier update_drag = LinearDragModifier(writer.lit(4.f).expr());
    const Modifier update_accel = AccelModifier(writer.lit(Vec3{-0.f, -16.f, -0.f}).expr());

    EffectAsset asset(capacity, SpawnerSettings::once((float)capacity), writer.finish());
    asset.name = "trail";
    asset.init(init_pos).init(init_vel).init(init_age).init(init_lifetime).init(init_color).update(update_drag).update(update_accel);
    return asset;
}

#define CHECK(call)                                                                      \
    do {                                                                                 \
        const int rc_ = (call);                                                          \
        if (rc_ != HNB_OK) { fprintf(stderr, "%s failed: %s\n", #call, hnb_last_error()); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    const uint32_t capacity = argc > 1 ? (uint32_t)atol(argv[1]) : 100000u;
    const int frames = argc > 2 ? atoi(argv[2]) : 90;
    const EffectAsset asset = create_trails_effect(capacity);
    const std::vector<uint8_t> program = lower(asset);  // the reference compiles WGSL at this point (lib.rs:805-1336)

    HnbContext* ctx = nullptr;
    HnbProgram* prog = nullptr;
    HnbEffect* fx = nullptr;
    CHECK(hnb_ctx_create(0, &ctx));
    CHECK(hnb_program_create(ctx, program.data(), program.size(), &prog));
    CHECK(hnb_effect_create(prog, /*slot_base=*/0, &fx));

    EffectSpawner spawner(asset.spawner);
    Pcg32 rng;
    const float dt = 1.0f / 60.0f;
    for (int f = 0; f < frames; ++f) {
        HnbSimParams sim;
        sim.delta_time = sim.virtual_delta_time = sim.real_delta_time = dt;
        sim.time = sim.virtual_time = sim.real_time = (float)f * dt;
        CHECK(hnb_frame_begin(ctx, &sim));
        const uint32_t spawn = spawner.tick(dt, rng);                       // EffectSpawner::tick (spawn.rs:838-921)
        CHECK(hnb_effect_set_frame(fx, spawn, 0x9e3779b9u * (uint32_t)(f + 1), nullptr));
        CHECK(hnb_simulate(ctx));
        uint32_t alive = 0;
        CHECK(hnb_effect_alive_count(fx, &alive));
        printf("%d %u\n", f, alive);
    }
    CHECK(hnb_program_destroy(prog));
    CHECK(hnb_ctx_destroy(ctx));
    return 0;
}
