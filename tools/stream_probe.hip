// Ablation probe for the streaming update kernel (development tool, not part of the product).
// Builds the same data layout as the runtime for one 16M-particle effect (all alive, identity
// alive list), then times k_update_slots_stream<ProgDragAccel> with parts switched off, next to plain
// copy kernels moving the same number of bytes. Usage: ./stream_probe [capacity] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../bevy_hanabi_amd/csrc/hnb_kernels.hip.h"
using namespace hnb;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define OP_(x) (uint32_t)HNB_OP_M_##x
typedef ProgStatic<OP_(AGE_TICK), OP_(VEL_SCALE), OP_(VEL_ADD), OP_(EULER)> ProgDragAccel;

__global__ void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n16; i += stride) dst[i] = src[i];
}
__global__ void k_read16(const uint4* __restrict__ src, uint32_t* out, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n16; i += stride) { uint4 v = src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) out[0] = acc;
}
// same traffic shape as the update kernel, hand-written without any of its bookkeeping
__global__ void __launch_bounds__(256) k_ideal(const uint32_t* __restrict__ list, float* pos, float* vel, float* age, const float* __restrict__ life,
                                               uint32_t* __restrict__ list_out, uint32_t n, float dt, float drag, float ay) {
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;  // quad index
    if (q * 4 >= n) return;
    const uint4 idx = reinterpret_cast<const uint4*>(list)[q];
    float4* pp = reinterpret_cast<float4*>(pos) + (size_t)q * 3;
    float4* vp = reinterpret_cast<float4*>(vel) + (size_t)q * 3;
    float4 p0 = pp[0], p1 = pp[1], p2 = pp[2], v0 = vp[0], v1 = vp[1], v2 = vp[2];
    float4 a = reinterpret_cast<float4*>(age)[q];
    const float4 l = reinterpret_cast<const float4*>(life)[q];
    float* pf = &p0.x; float* vf = &v0.x; (void)pf; (void)vf;
    float P[12] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w};
    float V[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
#pragma unroll
    for (int i = 0; i < 12; ++i) { V[i] = V[i] * drag; if (i % 3 == 1) V[i] = V[i] + ay; P[i] = P[i] + V[i] * dt; }
    a.x += dt; a.y += dt; a.z += dt; a.w += dt;
    pp[0] = make_float4(P[0], P[1], P[2], P[3]); pp[1] = make_float4(P[4], P[5], P[6], P[7]); pp[2] = make_float4(P[8], P[9], P[10], P[11]);
    vp[0] = make_float4(V[0], V[1], V[2], V[3]); vp[1] = make_float4(V[4], V[5], V[6], V[7]); vp[2] = make_float4(V[8], V[9], V[10], V[11]);
    reinterpret_cast<float4*>(age)[q] = a;
    uint4 o = idx;
    if (!(a.x < l.x)) o.x = 0xffffffffu;
    reinterpret_cast<uint4*>(list_out)[q] = o;
}

// the same update on component-planar storage (x[], y[], z[] per vec3 attribute): every access is one
// aligned 16-byte load per lane, 1 KiB contiguous per wave instruction
__global__ void __launch_bounds__(256) k_ideal_planar(const uint32_t* __restrict__ list, float* pos, float* vel, float* age, const float* __restrict__ life,
                                                      uint32_t* __restrict__ list_out, uint32_t n, float dt, float drag, float ay) {
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (q * 4 >= n) return;
    const uint4 idx = reinterpret_cast<const uint4*>(list)[q];
    const size_t plane = (size_t)n / 4;  // float4s per component plane
    float4* pp = reinterpret_cast<float4*>(pos);
    float4* vp = reinterpret_cast<float4*>(vel);
    float4 px = pp[q], py = pp[plane + q], pz = pp[2 * plane + q];
    float4 vx = vp[q], vy = vp[plane + q], vz = vp[2 * plane + q];
    float4 a = reinterpret_cast<float4*>(age)[q];
    const float4 l = reinterpret_cast<const float4*>(life)[q];
#define UPD(P, V, ADD) V.x = V.x * drag + ADD; V.y = V.y * drag + ADD; V.z = V.z * drag + ADD; V.w = V.w * drag + ADD; \
    P.x += V.x * dt; P.y += V.y * dt; P.z += V.z * dt; P.w += V.w * dt;
    UPD(px, vx, 0.0f) UPD(py, vy, ay) UPD(pz, vz, 0.0f)
#undef UPD
    a.x += dt; a.y += dt; a.z += dt; a.w += dt;
    pp[q] = px; pp[plane + q] = py; pp[2 * plane + q] = pz;
    vp[q] = vx; vp[plane + q] = vy; vp[2 * plane + q] = vz;
    reinterpret_cast<float4*>(age)[q] = a;
    uint4 o = idx;
    if (!(a.x < l.x)) o.x = 0xffffffffu;
    reinterpret_cast<uint4*>(list_out)[q] = o;
}

template <class F> float time_ms(int iters, F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main(int argc, char** argv) {
    const uint32_t cap = argc > 1 ? (uint32_t)atol(argv[1]) : (1u << 24);
    const int iters = argc > 2 ? atoi(argv[2]) : 20;
    const int blocks_per_cu = argc > 3 ? atoi(argv[3]) : 8;
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    SlotArgs sa{};
    sa.capacity = cap; sa.n_uregs = 8; sa.chunks_per_inst = (cap + kChunk - 1) / kChunk; sa.n_inst = 1;
    size_t off = 0;
    uint32_t alive_off[2], dead_off;
    alive_off[0] = off; off += al((size_t)cap * 4);
    alive_off[1] = off; off += al((size_t)cap * 4);
    dead_off = off; off += al((size_t)cap * 4);
    sa.plane_off[0] = off; off += al((size_t)cap * 12);
    sa.plane_off[1] = off; off += al((size_t)cap * 12);
    sa.plane_off[2] = off; off += al((size_t)cap * 4);
    sa.plane_off[3] = off; off += al((size_t)cap * 4);
    sa.alive_flag_off = off; off += al((size_t)cap);
    sa.lmin_off = off; off += al((size_t)sa.chunks_per_inst * 8);  // lifetime bounds + "completely alive" flags (left at 0: the probe measures the plain path)
    sa.flags = 0xf | (0x7 << 4);
    char* slab; CK(hipMalloc(&slab, off));
    CK(hipMemset(slab + sa.lmin_off, 0, (size_t)sa.chunks_per_inst * 8));
    std::vector<uint32_t> ident(cap); for (uint32_t i = 0; i < cap; ++i) ident[i] = i;
    CK(hipMemcpy(slab + alive_off[0], ident.data(), (size_t)cap * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(slab + alive_off[1], ident.data(), (size_t)cap * 4, hipMemcpyHostToDevice));
    CK(hipMemset(slab + sa.plane_off[0], 0, (size_t)cap * 32));
    CK(hipMemset(slab + sa.alive_flag_off, 1, (size_t)cap));
    std::vector<float> life(cap, 1e30f);
    CK(hipMemcpy(slab + sa.plane_off[3], life.data(), (size_t)cap * 4, hipMemcpyHostToDevice));
    // program: AGE_TICK u0 ; VEL_SCALE u1 ; VEL_ADD u2..4 ; EULER u0
    auto enc = [](uint32_t op, uint32_t a, uint32_t aux) { Ins i; i.x = op | (a << 16) | (a << 24); i.y = a | (7u << 10) | (aux << 16); return i; };
    Ins code[4] = {enc(OP_(AGE_TICK), 0x80, 1), enc(OP_(VEL_SCALE), 0x81, 0), enc(OP_(VEL_ADD), 0x82, 0), enc(OP_(EULER), 0x80, 0)};
    Ins* dcode; CK(hipMalloc(&dcode, sizeof code)); CK(hipMemcpy(dcode, code, sizeof code, hipMemcpyHostToDevice));
    sa.update_code = dcode; sa.update_len = 4;
    float ub[8] = {1.0f / 60, 0.93f, 0.0f, -0.26f, 0.0f, 0, 0, 0};
    uint32_t* dub; CK(hipMalloc(&dub, sizeof ub)); CK(hipMemcpy(dub, ub, sizeof ub, hipMemcpyHostToDevice));
    DevFrameInst fi{}; fi.xf[0] = fi.xf[5] = fi.xf[10] = 1.0f;
    DevFrameInst* dfi; CK(hipMalloc(&dfi, sizeof fi)); CK(hipMemcpy(dfi, &fi, sizeof fi, hipMemcpyHostToDevice));
    DevMeta m{}; m.alive_count = cap;
    DevMeta* dmeta; CK(hipMalloc(&dmeta, 2 * sizeof m)); CK(hipMemcpy(dmeta, &m, sizeof m, hipMemcpyHostToDevice)); CK(hipMemcpy(dmeta + 1, &m, sizeof m, hipMemcpyHostToDevice));
    uint64_t base_addr = (uint64_t)slab; uint64_t* dbase; CK(hipMalloc(&dbase, 8)); CK(hipMemcpy(dbase, &base_addr, 8, hipMemcpyHostToDevice));
    CompactBufs cb;
    CK(hipMalloc(&cb.counts, (size_t)sa.chunks_per_inst * 4)); CK(hipMemset(cb.counts, 0, (size_t)sa.chunks_per_inst * 4));
    CK(hipMalloc(&cb.deaths, 2 * 4 * 4)); CK(hipMemset(cb.deaths, 0, 2 * 4 * 4));
    cb.table_cap = 4; cb.parity = 0; cb.ev_totals = nullptr; cb.xcd_remap = 0;
    uint32_t* ticket = cb.deaths;
    CompactArgs ca{};
    ca.capacity = cap; ca.chunks_per_inst = sa.chunks_per_inst; ca.alive_off[0] = alive_off[0]; ca.alive_off[1] = alive_off[1]; ca.dead_off = dead_off;
    ca.alive_flag_off = sa.alive_flag_off; ca.slot_order = 0;
    const double bytes = (double)cap * 68.0;
    const uint32_t grid = sa.chunks_per_inst;
    (void)blocks_per_cu;
#define RUN(NAME, PROBE, WAVES, COMPACT)                                                                                         \
    {                                                                                                                            \
        float ms = time_ms(iters, [&] {                                                                                          \
            k_update_slots_stream<ProgDragAccel, WAVES, PROBE><<<grid, kBlock>>>(sa, dbase, dfi, dub, cb);                       \
            if (COMPACT) { k_count_rows<<<grid, kBlock>>>(ca, dbase, dmeta, dfi, cb); k_compact<<<grid, kBlock>>>(ca, dbase, dmeta, dmeta + 1, dfi, cb); }                                       \
        });                                                                                                                      \
        printf("%-44s %8.3f ms  %7.1f GB/s (68 B/particle)\n", NAME, ms, bytes / ms / 1e6);                                       \
    }
    RUN("k_update_slots (budget 8)", 0, 8, 0)
    RUN("k_update_slots (budget 6)", 0, 6, 0)
    RUN("k_update_slots + k_count_rows + k_compact (6)", 0, 6, 1)
    RUN("k_update_slots (budget 5)", 0, 5, 0)
    RUN("k_update_slots (budget 4)", 0, 4, 0)
    RUN("no stores", 4, 6, 0)
    RUN("no program", 8, 6, 0)
    {
        float ms = time_ms(iters, [&] { k_compact<<<grid, kBlock>>>(ca, dbase, dmeta, dmeta + 1, dfi, cb); });
        printf("%-44s %8.3f ms\n", "k_compact alone (no deaths)", ms);
    }
    {
        float ms = time_ms(iters, [&] { k_ideal<<<cap / 4 / 256, 256>>>((uint32_t*)(slab + alive_off[0]), (float*)(slab + sa.plane_off[0]), (float*)(slab + sa.plane_off[1]), (float*)(slab + sa.plane_off[2]), (float*)(slab + sa.plane_off[3]), (uint32_t*)(slab + alive_off[1]), cap, 1.f / 60, 0.93f, -0.26f); });
        printf("%-44s %8.3f ms  %7.1f GB/s\n", "hand-written ideal (no compaction)", ms, bytes / ms / 1e6);
        ms = time_ms(iters, [&] { k_ideal_planar<<<cap / 4 / 256, 256>>>((uint32_t*)(slab + alive_off[0]), (float*)(slab + sa.plane_off[0]), (float*)(slab + sa.plane_off[1]), (float*)(slab + sa.plane_off[2]), (float*)(slab + sa.plane_off[3]), (uint32_t*)(slab + alive_off[1]), cap, 1.f / 60, 0.93f, -0.26f); });
        printf("%-44s %8.3f ms  %7.1f GB/s\n", "ideal, component-planar vec3 storage", ms, bytes / ms / 1e6);
    }
    {
        const size_t n16 = (size_t)cap * 32 / 16;  // 32 B/particle read + 32 B/particle written ~ 64 B/particle
        float ms = time_ms(iters, [&] { k_copy16<<<2048, 256>>>((const uint4*)(slab + sa.plane_off[0]), (uint4*)(slab + alive_off[0]), n16 / 2); });
        printf("%-44s %8.3f ms  %7.1f GB/s (r+w)\n", "float4 copy 268 MB -> 268 MB, 2048 blocks", ms, (double)(n16 / 2) * 32 / ms / 1e6);
        ms = time_ms(iters, [&] { k_read16<<<2048, 256>>>((const uint4*)(slab + sa.plane_off[0]), ticket, n16); });
        printf("%-44s %8.3f ms  %7.1f GB/s (r)\n", "float4 read 537 MB", ms, (double)n16 * 16 / ms / 1e6);
    }
    return 0;
}
