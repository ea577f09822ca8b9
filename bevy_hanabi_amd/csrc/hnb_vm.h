// Particle program interpreters (host + device).
//
// Three consumers share this file:
//   * the host evaluates the UNIFORM stream once per instance per frame into the
//     parameter block of U registers (uniform_run);
//   * the generic kernels run the INIT stream and non-streamable UPDATE streams with
//     one particle per lane and a 32-register V file (vm_run);
//   * the streaming update kernel runs macro-op-only UPDATE streams on named registers
//     for 4 particles per lane (fast_run).
//
// Semantics follow the WGSL the reference would emit for the same modifiers:
//   expressions      src/graph/expr.rs:1121-1258 (operand order: left, then right)
//   PRNG             src/render/vfx_common.wgsl:260-343
//   modifier bodies  src/modifier/{accel,force,kill,position,velocity}.rs (cited per function)
// Arithmetic is defined by hnb_math.h.
#pragma once
#include "hnb_math.h"
#ifdef __HIPCC_RTC__
#include "hanabi_amd.h"
#else
#include "../../include/hanabi_amd.h"
#endif

namespace hnb {

// One instruction (two 32-bit words, see include/hanabi_amd.h).
struct __attribute__((aligned(8))) Ins { uint32_t x, y; };

#ifdef HNB_VM_BOUNDS_CHECK
// Host test builds only (tests/validator_fuzz): every register access is checked, so that a program the
// validator accepted but that addresses outside a file is reported instead of being undefined behaviour.
extern "C" void hnb_vm_oob(uint32_t index, uint32_t size);
template <uint32_t N>
struct checked_file_t {
    uint32_t v[N];
    uint32_t& operator[](uint32_t i) { if (i >= N) hnb_vm_oob(i, N); return v[i < N ? i : 0]; }
    const uint32_t& operator[](uint32_t i) const { if (i >= N) hnb_vm_oob(i, N); return v[i < N ? i : 0]; }
};
typedef checked_file_t<HNB_VM_MAX_REGS> vreg_file_t;
typedef checked_file_t<HNB_VM_MAX_REGS_WIDE> vreg_file_wide_t;
typedef checked_file_t<HNB_VM_MAX_UREGS> UFile;
#else
// V register file of one particle: a single LLVM vector so that dynamic (wave-uniform)
// register numbers lower to VGPR-indexed moves instead of scratch memory.
typedef uint32_t vreg_file_t __attribute__((vector_size(HNB_VM_MAX_REGS * 4)));
// Wide file for programs with init_regs / update_regs > HNB_VM_MAX_REGS (hanabi_amd.h).
// An array, not a vector type: constant indices (specialised kernels) scalarise it just the same, and the
// interpreter's dynamic indices address ONE scratch object instead of a vector temporary per site.
struct vreg_file_wide_t {
    uint32_t v[HNB_VM_MAX_REGS_WIDE];
    HNB_HD_MEMBER uint32_t& operator[](uint32_t i) { return v[i]; }
    HNB_HD_MEMBER const uint32_t& operator[](uint32_t i) const { return v[i]; }
};
// U register file on the host.
struct UFile {
    uint32_t v[HNB_VM_MAX_UREGS];
    HNB_HD_MEMBER uint32_t& operator[](uint32_t i) { return v[i]; }
    HNB_HD_MEMBER const uint32_t& operator[](uint32_t i) const { return v[i]; }
};
#endif

// Wave-uniform inputs of one effect instance for one frame (pointers into uniform memory:
// on the device every access through them is a scalar load).
struct VmUniforms {
    const uint32_t* u;   // parameter block: U registers produced by the uniform stream
    const float* xf;     // [12] emitter transform, row-major 3x4 (GpuSpawnerParams::transform)
};
constexpr uint32_t kNoPlane = 0xffffffffu;

// Attribute table entry as the kernels see it, and the per-particle window onto the SoA planes
// used by HNB_OP_LDA / HNB_OP_STA (non-pinned attributes are memory operands).
// A byte offset inside an instance slab, kept in units of 256 bytes (every section of a slab is 256-byte aligned): 32 bits reach 1 TiB, so a
// slab is not limited to 4 GiB - EffectAsset::capacity is a u32 (src/asset.rs:391-415), and a 100M-particle firework needs 6.6 GB. Converts
// to size_t wherever it meets a pointer (`base + off`).
struct soff_t {
    uint32_t v;
    HNB_HD_MEMBER operator size_t() const { return (size_t)v << 8; }
};
HNB_HD soff_t soff_of(uint64_t bytes) { soff_t o; o.v = (uint32_t)(bytes >> 8); return o; }
struct AttrDesc {
    soff_t plane_off;     // offset of the attribute plane from the instance slab base
    uint8_t ncomp;        // 32-bit components per particle (packed, vec3 = 12 B)
    uint8_t reg;          // first V register (pinned attributes) or HNB_REG_NONE
    uint8_t upd_flags;    // HNB_ATTR_UPD_*
    uint8_t pad;
};
struct VmAttrIO {
    char* slab;              // instance slab base
    const AttrDesc* attrs;   // attribute table (uniform memory)
    uint32_t slot;           // this particle's slot
    // parent particle that triggered this spawn (init of an effect with a parent, vfx_init.wgsl:166-171)
    const char* parent_slab = nullptr;          // parent instance slab base
    const uint32_t* parent_planes = nullptr;    // [HNB_ATTR_COUNT] plane offsets of the parent layout in 256-byte units (soff_t::v; kNoPlane = absent)
    uint32_t parent_slot = 0;
};
HNB_HD uint32_t* vm_attr_ptr(const VmAttrIO& io, uint32_t idx) {
    return reinterpret_cast<uint32_t*>(io.slab + io.attrs[idx].plane_off) + (size_t)io.slot * io.attrs[idx].ncomp;
}

template <class FILE_T>
struct VmState {
    FILE_T r;
    uint32_t seed;
    uint32_t pindex;    // particle_index (+slot_base): Attribute::ID
    uint32_t pcounter;  // particle_counter
    bool alive;
    bool was_alive = true;                       // AGE_CODE: age < lifetime before ageing (src/lib.rs:1226-1233)
    bool gpu_spawned = false;                    // init driven by GPU spawn events: no emitter translation (vfx_init.wgsl:183-189)
    uint32_t ev[HNB_MAX_EVENT_CHANNELS] = {};    // spawn events this particle appends per child channel
};

#define HNB_TAU 6.283185307179586476925286766559f

// ---- PRNG (vfx_common.wgsl:278-343) --------------------------------------------------
HNB_HD float vm_frand(uint32_t& seed) {
    seed = pcg_hash(seed);
    return to_float01(pcg_hash(seed));
}
struct Rand4 { float v0, v1, v2, v3; };
HNB_HD Rand4 vm_frand_n(uint32_t& seed, uint32_t n) {
    Rand4 o = {0.0f, 0.0f, 0.0f, 0.0f};
    if (n == 1) { o.v0 = vm_frand(seed); return o; }
    if (n == 4) {
        uint32_t r0 = pcg_hash(seed), r1 = pcg_hash(r0), r2 = pcg_hash(r1);
        seed = r2;
        o.v0 = to_float01(r0);
        o.v1 = to_float01(((r0 & 0xff000000u) >> 8) | (r1 & 0x0000ffffu));
        o.v2 = to_float01(((r1 & 0xffff0000u) >> 8) | (r2 & 0x000000ffu));
        o.v3 = to_float01(r2 >> 8);
        return o;
    }
    seed = pcg_hash(seed); o.v0 = to_float01(seed);
    seed = pcg_hash(seed); o.v1 = to_float01(seed);
    if (n == 3) { seed = pcg_hash(seed); o.v2 = to_float01(seed); }
    return o;
}
HNB_HD float rand4_get(const Rand4& r, uint32_t k) { return k == 0 ? r.v0 : (k == 1 ? r.v1 : (k == 2 ? r.v2 : r.v3)); }

// ---- small vector helpers (definition of WGSL dot / normalize / cross here) -------------
struct V3 { float x, y, z; };
HNB_HD float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
HNB_HD V3 sub3(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
// normalize(v) = v * (1 / length(v)) (hnb_math.h): one IEEE division instead of one per component
HNB_HD V3 normalize3(V3 a) {
    const float inv = 1.0f / f_sqrt(dot3(a, a));
    return V3{a.x * inv, a.y * inv, a.z * inv};
}
HNB_HD V3 cross3(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// transform * vec4(v, 0) (mat4x4 built from the 3x4 rows, vfx_init.wgsl:157-164)
HNB_HD V3 xform_dir(const float* xf, V3 v) {
    return V3{((xf[0] * v.x + xf[1] * v.y) + xf[2] * v.z) + xf[3] * 0.0f,
              ((xf[4] * v.x + xf[5] * v.y) + xf[6] * v.z) + xf[7] * 0.0f,
              ((xf[8] * v.x + xf[9] * v.y) + xf[10] * v.z) + xf[11] * 0.0f};
}

// ---- macro-op semantics on scalars ------------------------------------------------------
// AGE_CODE + REAP_CODE (src/lib.rs:1223-1258): age = age + dt; is_alive = age < lifetime.
HNB_HD void mac_age_tick(float& age, float lifetime, float dt, bool has_lifetime, bool& alive) {
    age = age + dt;
    if (has_lifetime) alive = age < lifetime;
}
// Euler integration (src/lib.rs:1106-1120): position += velocity * delta_time
HNB_HD void mac_euler(V3& pos, V3 vel, float dt) { pos = V3{pos.x + vel.x * dt, pos.y + vel.y * dt, pos.z + vel.z * dt}; }
// LinearDragModifier (force.rs:284-297): velocity *= <factor>
HNB_HD void mac_vel_scale(V3& vel, float s) { vel = V3{vel.x * s, vel.y * s, vel.z * s}; }
// AccelModifier (accel.rs:79-86): velocity += (<accel>) * delta_time, operand already multiplied
HNB_HD void mac_vel_add(V3& vel, V3 a) { vel = V3{vel.x + a.x, vel.y + a.y, vel.z + a.z}; }
// RadialAccelModifier (accel.rs:174-182)
HNB_HD void mac_radial_accel(V3 pos, V3& vel, V3 origin, float s) {
    const V3 radial = normalize3(sub3(pos, origin));
    vel = V3{vel.x + radial.x * s, vel.y + radial.y * s, vel.z + radial.z * s};
}
// TangentAccelModifier (accel.rs:289-299)
HNB_HD void mac_tangent_accel(V3 pos, V3& vel, V3 origin, V3 axis, float s) {
    const V3 radial = normalize3(sub3(pos, origin));
    const V3 tangent = normalize3(cross3(axis, radial));
    vel = V3{vel.x + tangent.x * s, vel.y + tangent.y * s, vel.z + tangent.z * s};
}
// SetVelocitySphereModifier (velocity.rs:131-137)
HNB_HD V3 mac_vel_sphere(V3 pos, V3 center, float speed) {
    const V3 n = normalize3(sub3(pos, center));
    return V3{n.x * speed, n.y * speed, n.z * speed};
}
// KillSphereModifier (kill.rs:76-96)
HNB_HD void mac_kill_sphere(V3 pos, V3 center, float sqr_radius, bool kill_inside, bool& alive) {
    const V3 d = sub3(pos, center);
    const float sqr = dot3(d, d);
    const bool kill = kill_inside ? (sqr < sqr_radius) : (sqr > sqr_radius);
    alive = alive && !kill;
}
// KillAabbModifier (kill.rs:156-181)
HNB_HD void mac_kill_aabb(V3 pos, V3 center, V3 half, bool kill_inside, bool& alive) {
    const float dx = f_abs(pos.x - center.x), dy = f_abs(pos.y - center.y), dz = f_abs(pos.z - center.z);
    const bool kill = kill_inside ? (dx < half.x && dy < half.y && dz < half.z) : (dx > half.x || dy > half.y || dz > half.z);
    alive = alive && !kill;
}
// ConformToSphereModifier body (force.rs:196-230)
struct ConformParams {
    V3 c;
    float radius, influence_dist, shell_half_thickness, max_attraction_speed, attraction_accel, sticky_factor;
};
HNB_HD void mac_conform_sphere(V3 pos, V3& vel, const ConformParams& q, float dt) {
    const V3 rel_pos = sub3(q.c, pos);
    const float origin_dist = f_sqrt(dot3(rel_pos, rel_pos));
    const V3 origin_dir = normalize3(rel_pos);
    const float surface_dist = origin_dist - q.radius;
    // `if (surface_dist > influence_dist) { return; }` expressed as a select
    const bool out_of_range = surface_dist > q.influence_dist;
    const float cur_radial_speed = dot3(vel, origin_dir);
    const float shell_factor = f_smoothstep(0.0f, q.shell_half_thickness, f_abs(surface_dist));
    const float max_radial_speed = f_sign(surface_dist) * shell_factor * q.max_attraction_speed;
    const float delta_speed = max_radial_speed - cur_radial_speed;
    const float sticky_accel = q.attraction_accel * q.sticky_factor;
    const float conforming_accel = f_mix(sticky_accel, q.attraction_accel, shell_factor);
    const float conforming_delta_speed = dt * conforming_accel;
    const float k = f_sign(delta_speed) * f_min(f_abs(delta_speed), conforming_delta_speed);
    if (!out_of_range) vel = V3{vel.x + k * origin_dir.x, vel.y + k * origin_dir.y, vel.z + k * origin_dir.z};
}
// SetPositionCircleModifier body (position.rs:71-99)
HNB_HD V3 mac_pos_circle(uint32_t& seed, V3 c, V3 n, float radius, bool volume) {
    const float sign = f_step(0.0f, n.z) * 2.0f - 1.0f;
    const float a = -1.0f / (sign + n.z);
    const float b = n.x * n.y * a;
    const V3 tangent = V3{1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x};
    const V3 bitangent = V3{b, sign + n.y * n.y * a, -n.y};
    const float r = volume ? f_sqrt(vm_frand(seed)) * radius : radius;
    const float theta = vm_frand(seed) * HNB_TAU;
    const float ct = f_cos(theta), st = f_sin(theta);
    const V3 dir = V3{tangent.x * ct + bitangent.x * st, tangent.y * ct + bitangent.y * st, tangent.z * ct + bitangent.z * st};
    return V3{c.x + r * dir.x, c.y + r * dir.y, c.z + r * dir.z};
}
// SetPositionSphereModifier body (position.rs:168-201)
HNB_HD V3 mac_pos_sphere(uint32_t& seed, V3 c, float radius, bool volume) {
    const float r = volume ? f_pow(vm_frand(seed), (float)(1.0 / 3.0)) * radius : radius;
    const float theta = vm_frand(seed) * HNB_TAU;
    const float z = vm_frand(seed) * 2.0f - 1.0f;
    const float phi = f_acos(z);
    const float sinphi = f_sin(phi);
    const float x = sinphi * f_cos(theta);
    const float y = sinphi * f_sin(theta);
    return V3{c.x + r * x, c.y + r * y, c.z + r * z};
}
// SetPositionCone3dModifier body (position.rs:283-315)
HNB_HD V3 mac_pos_cone3d(uint32_t& seed, float h0, float rt, float rb, const float* xf) {
    const float alpha_h = f_pow(vm_frand(seed), (float)(1.0 / 3.0));
    const float h = h0 * alpha_h;
    const float r0 = rb + (rt - rb) * alpha_h;
    const float alpha_r = f_sqrt(vm_frand(seed));
    const float r = r0 * alpha_r;
    const float theta = vm_frand(seed) * HNB_TAU;
    const float cost = f_cos(theta), sint = f_sin(theta);
    return xform_dir(xf, V3{r * cost, h, r * sint});
}
// SetVelocityCircleModifier body (velocity.rs:57-70)
HNB_HD V3 mac_vel_circle(V3 pos, V3 center, V3 axis, float speed, const float* xf) {
    const V3 delta = sub3(pos, center);
    const float d = dot3(delta, axis);
    const V3 radial = normalize3(V3{delta.x - d * axis.x, delta.y - d * axis.y, delta.z - d * axis.z});
    const V3 rv = xform_dir(xf, radial);
    return V3{rv.x * speed, rv.y * speed, rv.z * speed};
}
// SetVelocityTangentModifier body (velocity.rs:200-213)
HNB_HD V3 mac_vel_tangent(V3 pos, V3 origin, V3 axis, float speed, const float* xf) {
    const V3 radial = sub3(pos, origin);
    const V3 tangent = normalize3(cross3(axis, radial));
    const V3 tv = xform_dir(xf, tangent);
    return V3{tv.x * speed, tv.y * speed, tv.z * speed};
}

// ---- scalar op tables ------------------------------------------------------------------
template <bool HEAVY> HNB_HD float vm_f_unary(uint32_t op, float x) {
    switch (op) {
        case HNB_OP_FABS: return f_abs(x);
        case HNB_OP_FCEIL: return f_ceil(x);
        case HNB_OP_FFLOOR: return f_floor(x);
        case HNB_OP_FROUND: return f_round_even(x);
        case HNB_OP_FFRACT: return f_fract(x);
        case HNB_OP_FSQRT: return f_sqrt(x);
        case HNB_OP_FRSQ: return f_inv_sqrt(x);
        case HNB_OP_FSIGN: return f_sign(x);
        case HNB_OP_FSAT: return f_saturate(x);
        default: break;
    }
    if constexpr (HEAVY) {
        switch (op) {
            case HNB_OP_FSIN: return f_sin(x);
            case HNB_OP_FCOS: return f_cos(x);
            case HNB_OP_FTAN: return f_tan(x);
            case HNB_OP_FASIN: return f_asin(x);
            case HNB_OP_FACOS: return f_acos(x);
            case HNB_OP_FATAN: return f_atan(x);
            case HNB_OP_FEXP: return f_exp(x);
            case HNB_OP_FEXP2: return f_exp2(x);
            case HNB_OP_FLOG: return f_log(x);
            default: return f_log2(x);
        }
    }
    return 0.0f;
}
template <bool HEAVY> HNB_HD float vm_f_binary(uint32_t op, float x, float y) {
    switch (op) {
        case HNB_OP_FADD: return x + y;
        case HNB_OP_FSUB: return x - y;
        case HNB_OP_FMUL: return x * y;
        case HNB_OP_FDIV: return x / y;
        case HNB_OP_FREM: return f_rem(x, y);
        case HNB_OP_FMIN: return f_min(x, y);
        case HNB_OP_FMAX: return f_max(x, y);
        case HNB_OP_FSTEP: return f_step(x, y);
        default: break;
    }
    if constexpr (HEAVY) {
        if (op == HNB_OP_FATAN2) return f_atan2(x, y);
        return f_pow(x, y);
    }
    return 0.0f;
}
HNB_HD float vm_f_ternary(uint32_t op, float x, float y, float z) {
    switch (op) {
        case HNB_OP_FMIX: return f_mix(x, y, z);
        case HNB_OP_FCLAMP: return f_clamp(x, y, z);
        default: return f_smoothstep(x, y, z);
    }
}
HNB_HD uint32_t vm_f_compare(uint32_t op, float x, float y) {
    switch (op) {
        case HNB_OP_FLT: return x < y;
        case HNB_OP_FLE: return x <= y;
        case HNB_OP_FGT: return x > y;
        default: return x >= y;
    }
}
HNB_HD uint32_t vm_int_op(uint32_t op, uint32_t x, uint32_t y, uint32_t z) {
    const int32_t ix = (int32_t)x, iy = (int32_t)y, iz = (int32_t)z;
    switch (op) {
        case HNB_OP_IADD: return x + y;
        case HNB_OP_ISUB: return x - y;
        case HNB_OP_IMUL: return x * y;
        case HNB_OP_IDIV: return (uint32_t)i_div(ix, iy);
        case HNB_OP_IREM: return (uint32_t)i_rem(ix, iy);
        case HNB_OP_IMIN: return (uint32_t)((iy < ix) ? iy : ix);
        case HNB_OP_IMAX: return (uint32_t)((ix < iy) ? iy : ix);
        case HNB_OP_IABS: return (ix < 0) ? (0u - x) : x;
        case HNB_OP_ISIGN: return (uint32_t)((ix > 0) ? 1 : ((ix < 0) ? -1 : 0));
        case HNB_OP_ICLAMP: { const int32_t t = (ix < iy) ? iy : ix; return (uint32_t)((iz < t) ? iz : t); }
        case HNB_OP_ILT: return ix < iy;
        case HNB_OP_ILE: return ix <= iy;
        case HNB_OP_IGT: return ix > iy;
        case HNB_OP_IGE: return ix >= iy;
        case HNB_OP_UDIV: return u_div(x, y);
        case HNB_OP_UREM: return u_rem(x, y);
        case HNB_OP_UMIN: return (y < x) ? y : x;
        case HNB_OP_UMAX: return (x < y) ? y : x;
        case HNB_OP_UCLAMP: { const uint32_t t = (x < y) ? y : x; return (z < t) ? z : t; }
        case HNB_OP_ULT: return x < y;
        case HNB_OP_ULE: return x <= y;
        case HNB_OP_UGT: return x > y;
        default: return x >= y;
    }
}
HNB_HD uint32_t vm_convert(uint32_t op, uint32_t x) {
    switch (op) {
        case HNB_OP_F2I: return (uint32_t)f_to_i32(u2f(x));
        case HNB_OP_F2U: return f_to_u32(u2f(x));
        case HNB_OP_I2F: return f2u((float)(int32_t)x);
        case HNB_OP_U2F: return f2u((float)x);
        case HNB_OP_B2F: return f2u(x ? 1.0f : 0.0f);
        case HNB_OP_F2B: return u2f(x) != 0.0f;
        default: return x != 0u;  // I2B
    }
}
template <bool HEAVY>
HNB_HD uint32_t vm_elementwise(uint32_t op, uint32_t x, uint32_t y, uint32_t z) {
    if (op >= HNB_OP_FADD && op <= HNB_OP_FPOW) return f2u(vm_f_binary<HEAVY>(op, u2f(x), u2f(y)));
    if (op >= HNB_OP_FABS && op <= HNB_OP_FLOG2) return f2u(vm_f_unary<HEAVY>(op, u2f(x)));
    if (op >= HNB_OP_FMIX && op <= HNB_OP_FSMOOTH) return f2u(vm_f_ternary(op, u2f(x), u2f(y), u2f(z)));
    if (op >= HNB_OP_FLT && op <= HNB_OP_FGE) return vm_f_compare(op, u2f(x), u2f(y));
    if (op >= HNB_OP_IADD && op <= HNB_OP_UGE) return vm_int_op(op, x, y, z);
    if (op >= HNB_OP_F2I && op <= HNB_OP_I2B) return vm_convert(op, x);
    return x;  // MOV
}
HNB_HD bool vm_op_is_elementwise(uint32_t op) { return op >= HNB_OP_MOV && op <= HNB_OP_I2B; }
HNB_HD bool vm_op_is_heavy(uint32_t op) {
    return (op >= HNB_OP_FSIN && op <= HNB_OP_FLOG2) || op == HNB_OP_FATAN2 || op == HNB_OP_FPOW ||
           (op >= HNB_OP_PACK4UNORM && op <= HNB_OP_RANDN) || (op >= HNB_OP_M_POS_CIRCLE && op <= HNB_OP_M_VEL_TANGENT);
}
// Ops the streaming update kernel implements (with U operands only).
HNB_HD bool vm_op_is_streamable(uint32_t op) { return op >= HNB_OP_M_AGE_TICK && op <= HNB_OP_M_KILL_AABB; }

struct Out4 { uint32_t v0, v1, v2, v3; };
HNB_HD uint32_t out4_get(const Out4& o, uint32_t k) { return k == 0 ? o.v0 : (k == 1 ? o.v1 : (k == 2 ? o.v2 : o.v3)); }

// ---- generic interpreter (one particle) ------------------------------------------------
// USTREAM=true: the uniform stream on the host (operand bytes index the U file directly).
// Otherwise operand bit 7 selects the parameter block U.u[] (scalar load) over the V file.
//
// Code-generation contract (measured, hipcc 7.2 / gfx950): the V file stays in VGPRs
// without whole-file copies only if the loop has exactly ONE dynamically indexed write
// site; indexed reads and compile-time-indexed writes (pinned registers) are free. Hence
// the shape: every instruction produces up to four scalar results, one common loop stores.
template <bool USTREAM, class ST>
HNB_HD uint32_t vm_rd(const ST& S, const VmUniforms& U, uint32_t operand) {
    if constexpr (USTREAM) return S.r[operand];
    else return (operand & HNB_OPERAND_DECODED_U) ? U.u[operand & 0xffu] : S.r[operand];
}
template <bool USTREAM, class ST> HNB_HD float vm_rdf(const ST& S, const VmUniforms& U, uint32_t operand) {
    return u2f(vm_rd<USTREAM>(S, U, operand));
}
template <bool USTREAM, class ST> HNB_HD V3 vm_rd3(const ST& S, const VmUniforms& U, uint32_t operand) {
    return V3{vm_rdf<USTREAM>(S, U, operand), vm_rdf<USTREAM>(S, U, operand + 1), vm_rdf<USTREAM>(S, U, operand + 2)};
}
template <class ST> HNB_HD V3 vm_pin3(const ST& S, uint32_t reg) { return V3{u2f(S.r[reg]), u2f(S.r[reg + 1]), u2f(S.r[reg + 2])}; }

// One instruction. With a compile-time constant `ins` (the specialised kernels built at program
// creation, hnb_jit.h) every decode, switch and register index below folds away.
template <bool HEAVY, bool USTREAM, class ST>
HNB_HD void vm_exec(const Ins ins, ST& S, const VmUniforms& U, const uint32_t* props, const float* sim, const VmAttrIO& io) {
    const uint32_t op = ins.x & 0xffu, d = (ins.x >> 8) & 0xffu, w = ((ins.y >> 8) & 3u) + 1u;
    // varying streams: operand bytes -> decoded operands (bit 8 = U register); uniform stream: plain U indices
    const uint32_t a = USTREAM ? (ins.x >> 16) & 0xffu : HNB_OPERAND_DECODE((ins.x >> 16) & 0xffu, ins.y >> 13);
    const uint32_t b = USTREAM ? ins.x >> 24 : HNB_OPERAND_DECODE(ins.x >> 24, ins.y >> 14);
    const uint32_t c = USTREAM ? ins.y & 0xffu : HNB_OPERAND_DECODE(ins.y & 0xffu, ins.y >> 15);
    const uint32_t sa = (ins.y >> 10) & 1u ? 0u : 1u, sb = (ins.y >> 11) & 1u ? 0u : 1u, sc = (ins.y >> 12) & 1u ? 0u : 1u;
    const uint32_t aux = ins.y >> 16;
    const bool elementwise = vm_op_is_elementwise(op);
    uint32_t nout = w;  // registers written by the common store loop
    Out4 o = Out4{0u, 0u, 0u, 0u};
    // results of macro ops that rewrite pinned registers (stored with static indices)
    V3 npos = V3{0, 0, 0}, nvel = V3{0, 0, 0};
    float nage = 0.0f;
    bool wpos = false, wvel = false, wage = false;

    if (!elementwise) {
        nout = 0;
        switch (op) {
            case HNB_OP_LOADK:
                if constexpr (USTREAM) { nout = 1; o.v0 = ins.y; }
                break;
            case HNB_OP_LDB:
                if constexpr (USTREAM) { nout = 1; o.v0 = f2u(sim[a]); }
                break;
            case HNB_OP_LDP:
                if constexpr (USTREAM) {
                    nout = (a & 3u) + 1u;
                    o.v0 = props[ins.y];
                    o.v1 = nout > 1 ? props[ins.y + 1] : 0u;
                    o.v2 = nout > 2 ? props[ins.y + 2] : 0u;
                    o.v3 = nout > 3 ? props[ins.y + 3] : 0u;
                }
                break;
            case HNB_OP_LDA:
                if constexpr (!USTREAM) {
                    const uint32_t* p = vm_attr_ptr(io, aux);
                    nout = w;
                    o.v0 = p[0];
                    o.v1 = w > 1 ? p[1] : 0u;
                    o.v2 = w > 2 ? p[2] : 0u;
                    o.v3 = w > 3 ? p[3] : 0u;
                }
                break;
            case HNB_OP_STA:
                if constexpr (!USTREAM) {
                    uint32_t* p = vm_attr_ptr(io, aux);
                    for (uint32_t k = 0; k < w; ++k) p[k] = vm_rd<false>(S, U, a + k * sa);
                }
                break;
            case HNB_OP_LDPARENT:
                if constexpr (!USTREAM) {
                    nout = w;
                    const uint32_t* p = reinterpret_cast<const uint32_t*>(io.parent_slab + ((size_t)io.parent_planes[aux] << 8)) + (size_t)io.parent_slot * w;
                    o.v0 = p[0];
                    o.v1 = w > 1 ? p[1] : 0u;
                    o.v2 = w > 2 ? p[2] : 0u;
                    o.v3 = w > 3 ? p[3] : 0u;
                }
                break;
            case HNB_OP_M_EMIT_EVENTS:
                if constexpr (!USTREAM) {
                    const uint32_t cnt = vm_rd<false>(S, U, a);
                    const bool fire = (aux & 0x100u) ? (S.was_alive && !S.alive) : S.alive;
                    const uint32_t ch = aux & 0xffu;
#pragma unroll
                    for (uint32_t k = 0; k < HNB_MAX_EVENT_CHANNELS; ++k)
                        if (fire && k == ch) S.ev[k] += cnt;
                }
                break;
            case HNB_OP_LDID: nout = 1; o.v0 = S.pindex; break;
            case HNB_OP_LDPC: nout = 1; o.v0 = S.pcounter; break;
            case HNB_OP_LDALIVE: nout = 1; o.v0 = S.alive ? 1u : 0u; break;
            case HNB_OP_ALL: {
                nout = 1;
                uint32_t v = 1u;
                for (uint32_t k = 0; k < w; ++k) v &= (vm_rd<USTREAM>(S, U, a + k) != 0u) ? 1u : 0u;
                o.v0 = v;
            } break;
            case HNB_OP_ANY: {
                nout = 1;
                uint32_t v = 0u;
                for (uint32_t k = 0; k < w; ++k) v |= (vm_rd<USTREAM>(S, U, a + k) != 0u) ? 1u : 0u;
                o.v0 = v;
            } break;
            case HNB_OP_DOT: {
                nout = 1;
                float s = vm_rdf<USTREAM>(S, U, a) * vm_rdf<USTREAM>(S, U, b);
                for (uint32_t k = 1; k < w; ++k) s = s + vm_rdf<USTREAM>(S, U, a + k) * vm_rdf<USTREAM>(S, U, b + k);
                o.v0 = f2u(s);
            } break;
            case HNB_OP_LENGTH: {
                nout = 1;
                const float x0 = vm_rdf<USTREAM>(S, U, a);
                float s = x0 * x0;
                for (uint32_t k = 1; k < w; ++k) { const float x = vm_rdf<USTREAM>(S, U, a + k); s = s + x * x; }
                o.v0 = f2u(f_sqrt(s));
            } break;
            case HNB_OP_DISTANCE: {
                nout = 1;
                const float t0 = vm_rdf<USTREAM>(S, U, a) - vm_rdf<USTREAM>(S, U, b);
                float s = t0 * t0;
                for (uint32_t k = 1; k < w; ++k) {
                    const float t = vm_rdf<USTREAM>(S, U, a + k) - vm_rdf<USTREAM>(S, U, b + k);
                    s = s + t * t;
                }
                o.v0 = f2u(f_sqrt(s));
            } break;
            case HNB_OP_NORMALIZE: {
                nout = w;
                const float x0 = vm_rdf<USTREAM>(S, U, a), x1 = w > 1 ? vm_rdf<USTREAM>(S, U, a + 1) : 0.0f;
                const float x2 = w > 2 ? vm_rdf<USTREAM>(S, U, a + 2) : 0.0f, x3 = w > 3 ? vm_rdf<USTREAM>(S, U, a + 3) : 0.0f;
                float s = x0 * x0;
                if (w > 1) s = s + x1 * x1;
                if (w > 2) s = s + x2 * x2;
                if (w > 3) s = s + x3 * x3;
                const float inv = 1.0f / f_sqrt(s);
                o = Out4{f2u(x0 * inv), f2u(x1 * inv), f2u(x2 * inv), f2u(x3 * inv)};
            } break;
            case HNB_OP_CROSS: {
                nout = 3;
                const V3 r = cross3(vm_rd3<USTREAM>(S, U, a), vm_rd3<USTREAM>(S, U, b));
                o = Out4{f2u(r.x), f2u(r.y), f2u(r.z), 0u};
            } break;
            case HNB_OP_ALIVE_SET: S.alive = vm_rd<USTREAM>(S, U, a) != 0u; break;
            case HNB_OP_ALIVE_AND: S.alive = S.alive && (vm_rd<USTREAM>(S, U, a) != 0u); break;
            case HNB_OP_KILL_IF: S.alive = S.alive && (vm_rd<USTREAM>(S, U, a) == 0u); break;
            default:
                if constexpr (!USTREAM) {
                    const V3 pos = vm_pin3(S, HNB_REG_POSITION), vel = vm_pin3(S, HNB_REG_VELOCITY);
                    switch (op) {
                        case HNB_OP_M_AGE_TICK: {
                            float age = u2f(S.r[HNB_REG_AGE]);
                            if (aux & 1u) S.was_alive = age < u2f(S.r[HNB_REG_LIFETIME]);
                            mac_age_tick(age, u2f(S.r[HNB_REG_LIFETIME]), vm_rdf<false>(S, U, a), (aux & 1u) != 0u, S.alive);
                            nage = age; wage = true;
                        } break;
                        case HNB_OP_M_EULER: npos = pos; mac_euler(npos, vel, vm_rdf<false>(S, U, a)); wpos = true; break;
                        case HNB_OP_M_VEL_SCALE: nvel = vel; mac_vel_scale(nvel, vm_rdf<false>(S, U, a)); wvel = true; break;
                        case HNB_OP_M_VEL_ADD: nvel = vel; mac_vel_add(nvel, vm_rd3<false>(S, U, a)); wvel = true; break;
                        case HNB_OP_M_PIN_SET:  // dst is a pinned register: route through the pinned write-back
                            if (d == HNB_REG_POSITION) { npos = vm_rd3<false>(S, U, a); wpos = true; }
                            else if (d == HNB_REG_VELOCITY) { nvel = vm_rd3<false>(S, U, a); wvel = true; }
                            else if (d == HNB_REG_AGE) { nage = vm_rdf<false>(S, U, a); wage = true; }
                            else { nout = 1; o.v0 = vm_rd<false>(S, U, a); }  // LIFETIME: d == 7 via the common store
                            break;
                        case HNB_OP_M_RADIAL_ACCEL:
                            nvel = vel; mac_radial_accel(pos, nvel, vm_rd3<false>(S, U, a), vm_rdf<false>(S, U, b)); wvel = true;
                            break;
                        case HNB_OP_M_TANGENT_ACCEL:
                            nvel = vel;
                            mac_tangent_accel(pos, nvel, vm_rd3<false>(S, U, a), vm_rd3<false>(S, U, b), vm_rdf<false>(S, U, c));
                            wvel = true;
                            break;
                        case HNB_OP_M_CONFORM_SPHERE: {
                            ConformParams q;
                            q.c = vm_rd3<false>(S, U, a);
                            q.radius = vm_rdf<false>(S, U, a + 3); q.influence_dist = vm_rdf<false>(S, U, a + 4);
                            q.shell_half_thickness = vm_rdf<false>(S, U, a + 5); q.max_attraction_speed = vm_rdf<false>(S, U, a + 6);
                            q.attraction_accel = vm_rdf<false>(S, U, a + 7); q.sticky_factor = vm_rdf<false>(S, U, a + 8);
                            nvel = vel; mac_conform_sphere(pos, nvel, q, vm_rdf<false>(S, U, b)); wvel = true;
                        } break;
                        case HNB_OP_M_KILL_SPHERE:
                            mac_kill_sphere(pos, vm_rd3<false>(S, U, a), vm_rdf<false>(S, U, b), (aux & 1u) != 0u, S.alive);
                            break;
                        case HNB_OP_M_KILL_AABB:
                            mac_kill_aabb(pos, vm_rd3<false>(S, U, a), vm_rd3<false>(S, U, b), (aux & 1u) != 0u, S.alive);
                            break;
                        case HNB_OP_M_VEL_SPHERE:
                            nvel = mac_vel_sphere(pos, vm_rd3<false>(S, U, a), vm_rdf<false>(S, U, b)); wvel = true;
                            break;
                        case HNB_OP_M_ADD_XLATE:  // only CPU-spawned particles get the emitter translation (vfx_init.wgsl:183-189)
                            if (!S.gpu_spawned) { npos = V3{pos.x + U.xf[3], pos.y + U.xf[7], pos.z + U.xf[11]}; wpos = true; }
                            break;
                        default: break;
                    }
                }
                if constexpr (HEAVY && !USTREAM) {
                    const V3 pos = vm_pin3(S, HNB_REG_POSITION);
                    switch (op) {
                        case HNB_OP_FRAND: {
                            nout = w;
                            const Rand4 v = vm_frand_n(S.seed, w);
                            o = Out4{f2u(v.v0), f2u(v.v1), f2u(v.v2), f2u(v.v3)};
                        } break;
                        case HNB_OP_RANDU: {  // a + frandN() * (b - a)
                            nout = w;
                            const Rand4 v = vm_frand_n(S.seed, w);
                            uint32_t t[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                            for (uint32_t k = 0; k < 4; ++k)
                                if (k < w) {
                                    const float lo = vm_rdf<false>(S, U, a + k * sa), hi = vm_rdf<false>(S, U, b + k * sb);
                                    t[k] = f2u(lo + rand4_get(v, k) * (hi - lo));
                                }
                            o = Out4{t[0], t[1], t[2], t[3]};
                        } break;
                        case HNB_OP_RANDN: {  // mean + std_dev * r * cos(tau * v), r = sqrt(-2 log u)
                            nout = w;
                            const float u = vm_frand(S.seed);
                            const Rand4 v = vm_frand_n(S.seed, w);
                            const float rr = f_sqrt(-2.0f * f_log(u));
                            uint32_t t[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                            for (uint32_t k = 0; k < 4; ++k)
                                if (k < w) {
                                    const float mean = vm_rdf<false>(S, U, a + k * sa), sd = vm_rdf<false>(S, U, b + k * sb);
                                    t[k] = f2u(mean + sd * rr * f_cos(HNB_TAU * rand4_get(v, k)));
                                }
                            o = Out4{t[0], t[1], t[2], t[3]};
                        } break;
                        case HNB_OP_M_POS_CIRCLE:
                            npos = mac_pos_circle(S.seed, vm_rd3<false>(S, U, a), vm_rd3<false>(S, U, a + 3), vm_rdf<false>(S, U, a + 6),
                                                  (aux & 1u) != 0u);
                            wpos = true;
                            break;
                        case HNB_OP_M_POS_SPHERE:
                            npos = mac_pos_sphere(S.seed, vm_rd3<false>(S, U, a), vm_rdf<false>(S, U, a + 3), (aux & 1u) != 0u);
                            wpos = true;
                            break;
                        case HNB_OP_M_POS_CONE3D:
                            npos = mac_pos_cone3d(S.seed, vm_rdf<false>(S, U, a), vm_rdf<false>(S, U, a + 1), vm_rdf<false>(S, U, a + 2), U.xf);
                            wpos = true;
                            break;
                        case HNB_OP_M_VEL_CIRCLE:
                            nvel = mac_vel_circle(pos, vm_rd3<false>(S, U, a), vm_rd3<false>(S, U, a + 3), vm_rdf<false>(S, U, a + 6), U.xf);
                            wvel = true;
                            break;
                        case HNB_OP_M_VEL_TANGENT:
                            nvel = mac_vel_tangent(pos, vm_rd3<false>(S, U, a), vm_rd3<false>(S, U, a + 3), vm_rdf<false>(S, U, a + 6), U.xf);
                            wvel = true;
                            break;
                        default: break;
                    }
                }
                if constexpr (HEAVY) {
                    switch (op) {
                        case HNB_OP_PACK4UNORM:
                            nout = 1;
                            o.v0 = pack_unorm8(vm_rdf<USTREAM>(S, U, a)) | (pack_unorm8(vm_rdf<USTREAM>(S, U, a + 1)) << 8) |
                                   (pack_unorm8(vm_rdf<USTREAM>(S, U, a + 2)) << 16) | (pack_unorm8(vm_rdf<USTREAM>(S, U, a + 3)) << 24);
                            break;
                        case HNB_OP_PACK4SNORM:
                            nout = 1;
                            o.v0 = pack_snorm8(vm_rdf<USTREAM>(S, U, a)) | (pack_snorm8(vm_rdf<USTREAM>(S, U, a + 1)) << 8) |
                                   (pack_snorm8(vm_rdf<USTREAM>(S, U, a + 2)) << 16) | (pack_snorm8(vm_rdf<USTREAM>(S, U, a + 3)) << 24);
                            break;
                        case HNB_OP_UNPACK4UNORM: {
                            nout = 4;
                            const uint32_t v = vm_rd<USTREAM>(S, U, a);
                            o = Out4{f2u(unpack_unorm8(v)), f2u(unpack_unorm8(v >> 8)), f2u(unpack_unorm8(v >> 16)), f2u(unpack_unorm8(v >> 24))};
                        } break;
                        case HNB_OP_UNPACK4SNORM: {
                            nout = 4;
                            const uint32_t v = vm_rd<USTREAM>(S, U, a);
                            o = Out4{f2u(unpack_snorm8(v)), f2u(unpack_snorm8(v >> 8)), f2u(unpack_snorm8(v >> 16)), f2u(unpack_snorm8(v >> 24))};
                        } break;
                        default: break;
                    }
                }
                break;
        }
    }

    // pinned write-back: compile-time register numbers
    if constexpr (!USTREAM) {
        if (wpos) { S.r[HNB_REG_POSITION] = f2u(npos.x); S.r[HNB_REG_POSITION + 1] = f2u(npos.y); S.r[HNB_REG_POSITION + 2] = f2u(npos.z); }
        if (wvel) { S.r[HNB_REG_VELOCITY] = f2u(nvel.x); S.r[HNB_REG_VELOCITY + 1] = f2u(nvel.y); S.r[HNB_REG_VELOCITY + 2] = f2u(nvel.z); }
        if (wage) S.r[HNB_REG_AGE] = f2u(nage);
    }
    // The one dynamically indexed store site.
    for (uint32_t k = 0; k < nout; ++k) {
        uint32_t out;
        if (elementwise)
            out = vm_elementwise<HEAVY>(op, vm_rd<USTREAM>(S, U, a + k * sa), vm_rd<USTREAM>(S, U, b + k * sb),
                                        vm_rd<USTREAM>(S, U, c + k * sc));
        else out = out4_get(o, k);
        S.r[d + k] = out;
    }
}

template <bool HEAVY, bool USTREAM, class ST>
HNB_HD void vm_run(const Ins* __restrict__ code, uint32_t n_ins, ST& S, const VmUniforms& U, const uint32_t* props,
                   const float* sim, const VmAttrIO& io) {
    for (uint32_t pc = 0; pc < n_ins; ++pc) vm_exec<HEAVY, USTREAM>(code[pc], S, U, props, sim, io);
}

// Host: evaluate the uniform stream of one instance into its parameter block.
static inline void uniform_run(const Ins* code, uint32_t n_ins, const uint32_t* props, const float* sim, uint32_t* out_u,
                               uint32_t n_uregs) {
    VmState<UFile> S;
    for (uint32_t i = 0; i < HNB_VM_MAX_UREGS; ++i) S.r.v[i] = 0u;
    S.seed = 0; S.pindex = 0; S.pcounter = 0; S.alive = true;
    VmUniforms U;
    U.u = nullptr; U.xf = nullptr;
    VmAttrIO io;
    io.slab = nullptr; io.attrs = nullptr; io.slot = 0;
    vm_run<true, true>(code, n_ins, S, U, props, sim, io);
    for (uint32_t i = 0; i < n_uregs; ++i) out_u[i] = S.r.v[i];
}

// ---- streaming interpreter: macro ops on named registers, P particles per lane -----------
template <int P>
struct Pinned {
    V3 pos[P], vel[P];
    float age[P], lifetime[P];
    bool alive[P];
};
HNB_HD float uf(const VmUniforms& U, uint32_t operand) { return u2f(U.u[operand & 0xffu]); }
HNB_HD V3 uf3(const VmUniforms& U, uint32_t operand) { return V3{uf(U, operand), uf(U, operand + 1), uf(U, operand + 2)}; }

// Ops of the "lean" streaming variant; the rest (normalize / cross / smoothstep heavy) only exist
// in the FULL variant so that the common programs keep a small register footprint.
HNB_HD bool vm_op_is_lean(uint32_t op) {
    return op == HNB_OP_M_AGE_TICK || op == HNB_OP_M_EULER || op == HNB_OP_M_VEL_SCALE || op == HNB_OP_M_VEL_ADD || op == HNB_OP_M_PIN_SET ||
           op == HNB_OP_M_KILL_SPHERE || op == HNB_OP_M_KILL_AABB;
}

template <int P, bool FULL>
HNB_HD void fast_run(const Ins* __restrict__ code, uint32_t n_ins, Pinned<P>& X, const VmUniforms& U) {
    for (uint32_t pc = 0; pc < n_ins; ++pc) {
        const Ins ins = code[pc];
        const uint32_t op = ins.x & 0xffu, d = (ins.x >> 8) & 0xffu, aux = ins.y >> 16;
        const uint32_t a = HNB_OPERAND_DECODE((ins.x >> 16) & 0xffu, ins.y >> 13), b = HNB_OPERAND_DECODE(ins.x >> 24, ins.y >> 14);
        const uint32_t c = HNB_OPERAND_DECODE(ins.y & 0xffu, ins.y >> 15);
        switch (op) {
            case HNB_OP_M_AGE_TICK: {
                const float dt = uf(U, a);
#pragma unroll
                for (int p = 0; p < P; ++p) mac_age_tick(X.age[p], X.lifetime[p], dt, (aux & 1u) != 0u, X.alive[p]);
            } break;
            case HNB_OP_M_EULER: {
                const float dt = uf(U, a);
#pragma unroll
                for (int p = 0; p < P; ++p) mac_euler(X.pos[p], X.vel[p], dt);
            } break;
            case HNB_OP_M_VEL_SCALE: {
                const float s = uf(U, a);
#pragma unroll
                for (int p = 0; p < P; ++p) mac_vel_scale(X.vel[p], s);
            } break;
            case HNB_OP_M_VEL_ADD: {
                const V3 v = uf3(U, a);
#pragma unroll
                for (int p = 0; p < P; ++p) mac_vel_add(X.vel[p], v);
            } break;
            case HNB_OP_M_PIN_SET: {
                if (d == HNB_REG_POSITION || d == HNB_REG_VELOCITY) {
                    const V3 v = uf3(U, a);
#pragma unroll
                    for (int p = 0; p < P; ++p) { if (d == HNB_REG_POSITION) X.pos[p] = v; else X.vel[p] = v; }
                } else {
                    const float s = uf(U, a);
#pragma unroll
                    for (int p = 0; p < P; ++p) { if (d == HNB_REG_AGE) X.age[p] = s; else X.lifetime[p] = s; }
                }
            } break;
            case HNB_OP_M_RADIAL_ACCEL:
                if constexpr (FULL) {
                    const V3 origin = uf3(U, a);
                    const float s = uf(U, b);
#pragma unroll
                    for (int p = 0; p < P; ++p) mac_radial_accel(X.pos[p], X.vel[p], origin, s);
                }
                break;
            case HNB_OP_M_TANGENT_ACCEL:
                if constexpr (FULL) {
                    const V3 origin = uf3(U, a), axis = uf3(U, b);
                    const float s = uf(U, c);
#pragma unroll
                    for (int p = 0; p < P; ++p) mac_tangent_accel(X.pos[p], X.vel[p], origin, axis, s);
                }
                break;
            case HNB_OP_M_CONFORM_SPHERE:
                if constexpr (FULL) {
                    ConformParams q;
                    q.c = uf3(U, a);
                    q.radius = uf(U, a + 3); q.influence_dist = uf(U, a + 4); q.shell_half_thickness = uf(U, a + 5);
                    q.max_attraction_speed = uf(U, a + 6); q.attraction_accel = uf(U, a + 7); q.sticky_factor = uf(U, a + 8);
                    const float dt = uf(U, b);
#pragma unroll
                    for (int p = 0; p < P; ++p) mac_conform_sphere(X.pos[p], X.vel[p], q, dt);
                }
                break;
            case HNB_OP_M_KILL_SPHERE: {
                const V3 center = uf3(U, a);
                const float r2 = uf(U, b);
#pragma unroll
                for (int p = 0; p < P; ++p) mac_kill_sphere(X.pos[p], center, r2, (aux & 1u) != 0u, X.alive[p]);
            } break;
            case HNB_OP_M_KILL_AABB: {
                const V3 center = uf3(U, a), half = uf3(U, b);
#pragma unroll
                for (int p = 0; p < P; ++p) mac_kill_aabb(X.pos[p], center, half, (aux & 1u) != 0u, X.alive[p]);
            } break;
            default: break;
        }
    }
}


// ---- statically specialised op sequences ------------------------------------------------------------
// The hottest update programs are short, fixed sequences of macro ops. Compiling a sequence as a
// template pack gives straight-line code (no per-op scalar branch, no phi webs over the named
// registers); operands still come from the parameter block at run time, so one instantiation
// serves every effect whose update stream has the same opcode sequence.
template <uint32_t OP, int P>
HNB_HD void apply_static(const Ins ins, Pinned<P>& X, const VmUniforms& U) {
    const uint32_t a = HNB_OPERAND_DECODE((ins.x >> 16) & 0xffu, ins.y >> 13), b = HNB_OPERAND_DECODE(ins.x >> 24, ins.y >> 14);
    const uint32_t c = HNB_OPERAND_DECODE(ins.y & 0xffu, ins.y >> 15), aux = ins.y >> 16;
    if constexpr (OP == HNB_OP_M_AGE_TICK) {
        const float dt = uf(U, a);
        const bool has_lifetime = (aux & 1u) != 0u;
#pragma unroll
        for (int p = 0; p < P; ++p) mac_age_tick(X.age[p], X.lifetime[p], dt, has_lifetime, X.alive[p]);
    } else if constexpr (OP == HNB_OP_M_EULER) {
        const float dt = uf(U, a);
#pragma unroll
        for (int p = 0; p < P; ++p) mac_euler(X.pos[p], X.vel[p], dt);
    } else if constexpr (OP == HNB_OP_M_VEL_SCALE) {
        const float s = uf(U, a);
#pragma unroll
        for (int p = 0; p < P; ++p) mac_vel_scale(X.vel[p], s);
    } else if constexpr (OP == HNB_OP_M_VEL_ADD) {
        const V3 v = uf3(U, a);
#pragma unroll
        for (int p = 0; p < P; ++p) mac_vel_add(X.vel[p], v);
    } else if constexpr (OP == HNB_OP_M_PIN_SET) {
        const uint32_t d = (ins.x >> 8) & 0xffu;  // wave-uniform: the destination register is an instruction field
        if (d == HNB_REG_POSITION || d == HNB_REG_VELOCITY) {
            const V3 v = uf3(U, a);
#pragma unroll
            for (int p = 0; p < P; ++p) { if (d == HNB_REG_POSITION) X.pos[p] = v; else X.vel[p] = v; }
        } else {
            const float s = uf(U, a);
#pragma unroll
            for (int p = 0; p < P; ++p) { if (d == HNB_REG_AGE) X.age[p] = s; else X.lifetime[p] = s; }
        }
    } else if constexpr (OP == HNB_OP_M_RADIAL_ACCEL) {
        const V3 origin = uf3(U, a);
        const float s = uf(U, b);
#pragma unroll
        for (int p = 0; p < P; ++p) mac_radial_accel(X.pos[p], X.vel[p], origin, s);
    } else if constexpr (OP == HNB_OP_M_TANGENT_ACCEL) {
        const V3 origin = uf3(U, a), axis = uf3(U, b);
        const float s = uf(U, c);
#pragma unroll
        for (int p = 0; p < P; ++p) mac_tangent_accel(X.pos[p], X.vel[p], origin, axis, s);
    } else if constexpr (OP == HNB_OP_M_CONFORM_SPHERE) {
        ConformParams q;
        q.c = uf3(U, a);
        q.radius = uf(U, a + 3); q.influence_dist = uf(U, a + 4); q.shell_half_thickness = uf(U, a + 5);
        q.max_attraction_speed = uf(U, a + 6); q.attraction_accel = uf(U, a + 7); q.sticky_factor = uf(U, a + 8);
        const float dt = uf(U, b);
#pragma unroll
        for (int p = 0; p < P; ++p) mac_conform_sphere(X.pos[p], X.vel[p], q, dt);
    } else if constexpr (OP == HNB_OP_M_KILL_SPHERE) {
        const V3 center = uf3(U, a);
        const float r2 = uf(U, b);
        const bool inside = (aux & 1u) != 0u;
#pragma unroll
        for (int p = 0; p < P; ++p) mac_kill_sphere(X.pos[p], center, r2, inside, X.alive[p]);
    } else if constexpr (OP == HNB_OP_M_KILL_AABB) {
        const V3 center = uf3(U, a), half = uf3(U, b);
        const bool inside = (aux & 1u) != 0u;
#pragma unroll
        for (int p = 0; p < P; ++p) mac_kill_aabb(X.pos[p], center, half, inside, X.alive[p]);
    } else {
        static_assert(OP == HNB_OP_M_AGE_TICK, "apply_static: opcode is not a streamable macro op");
    }
}

// The same ops on a FLAT view of the position / velocity planes: 12 consecutive-in-the-plane floats per plane that do not belong to whole
// particles (three 16-byte words of a lane, see k_update_slots_stream's flat path). Float k of a plane is component k mod 3 of particle
// k / 3, and these four ops are component-wise - the same IEEE operations in the same order on every float as the per-particle form,
// with the per-component operand picked by `rot` (component of the lane's first float) + the float's position.
HNB_HD float rot3(const V3 v, uint32_t c) { return c == 0u ? v.x : (c == 1u ? v.y : v.z); }
template <uint32_t OP>
HNB_HD void apply_static_flat(const Ins ins, float (&pos)[3][4], float (&vel)[3][4], uint32_t rot, const VmUniforms& U) {
    const uint32_t a = HNB_OPERAND_DECODE((ins.x >> 16) & 0xffu, ins.y >> 13);
    if constexpr (OP == HNB_OP_M_EULER) {
        const float dt = uf(U, a);
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) pos[j][e] = pos[j][e] + vel[j][e] * dt;
    } else if constexpr (OP == HNB_OP_M_VEL_SCALE) {
        const float s = uf(U, a);
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) vel[j][e] = vel[j][e] * s;
    } else if constexpr (OP == HNB_OP_M_VEL_ADD) {
        const V3 v = uf3(U, a);
        const float vr[3] = {rot3(v, rot % 3u), rot3(v, (rot + 1u) % 3u), rot3(v, (rot + 2u) % 3u)};
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) vel[j][e] = vel[j][e] + vr[(j + e) % 3];
    } else {
        static_assert(OP == HNB_OP_M_AGE_TICK, "apply_static_flat: not a component-wise op");  // the age lives in the chunk's cohort word
    }
}

// Interpreted program (any streamable sequence).
struct ProgInterp {
    static constexpr bool kFlat = false;
    static constexpr bool kAgeOnly = false;
    HNB_HD_MEMBER static void run_flat(const Ins* __restrict__, float (&)[3][4], float (&)[3][4], uint32_t, const VmUniforms&) {}
    static constexpr uint32_t kLen = 0;
    template <int P> HNB_HD_MEMBER static void run(const Ins* __restrict__ code, uint32_t n_ins, Pinned<P>& X, const VmUniforms& U) {
        fast_run<P, true>(code, n_ins, X, U);
    }
};
// Empty update stream (no AGE, no velocity, no update modifier): only the lists are maintained.
struct ProgNone {
    static constexpr bool kFlat = false;
    static constexpr bool kAgeOnly = false;
    HNB_HD_MEMBER static void run_flat(const Ins* __restrict__, float (&)[3][4], float (&)[3][4], uint32_t, const VmUniforms&) {}
    static constexpr uint32_t kLen = 0;
    template <int P> HNB_HD_MEMBER static void run(const Ins* __restrict__, uint32_t, Pinned<P>&, const VmUniforms&) {}
    static bool matches(const Ins*, uint32_t n_ins) { return n_ins == 0; }
};
// Fixed opcode sequence.
template <uint32_t... OPS>
struct ProgStatic {
    static constexpr uint32_t kLen = sizeof...(OPS);
    // every op component-wise (and an AGE_TICK among them: the flat path needs the chunk's ages in a cohort word)
    static constexpr bool kFlat = ((OPS == HNB_OP_M_AGE_TICK || OPS == HNB_OP_M_EULER || OPS == HNB_OP_M_VEL_SCALE || OPS == HNB_OP_M_VEL_ADD) && ...) &&
                                  ((OPS == HNB_OP_M_AGE_TICK) || ...);
    // nothing but AGE_TICK (ribbon.rs: MotionIntegration::None): the update reads and writes ONE scalar plane - see the age prefetch of
    // update_stream_chunk
    static constexpr bool kAgeOnly = sizeof...(OPS) > 0 && ((OPS == HNB_OP_M_AGE_TICK) && ...);
    HNB_HD_MEMBER static void run_flat(const Ins* __restrict__ code, float (&pos)[3][4], float (&vel)[3][4], uint32_t rot, const VmUniforms& U) {
        if constexpr (kFlat) {
            uint32_t i = 0;
            ((apply_static_flat<OPS>(code[i++], pos, vel, rot, U)), ...);
        }
    }
    template <int P> HNB_HD_MEMBER static void run(const Ins* __restrict__ code, uint32_t, Pinned<P>& X, const VmUniforms& U) {
        uint32_t i = 0;
        ((apply_static<OPS, P>(code[i++], X, U)), ...);
    }
    static bool matches(const Ins* code, uint32_t n_ins) {
        const uint32_t ops[] = {OPS...};
        if (n_ins != kLen) return false;
        for (uint32_t i = 0; i < kLen; ++i)
            if ((code[i].x & 0xffu) != ops[i]) return false;
        return true;
    }
};

}  // namespace hnb
