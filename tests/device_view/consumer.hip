// A device-side consumer of the simulation's output, written against include/hanabi_amd.h ALONE (tests/test_device_view.py): what a renderer's
// vertex stage does with the reference's buffers (vfx_render.wgsl: particle_buffer[indirect_buffer[..]]) - gather an attribute by alive-list
// row - through HnbDeviceView. No host copy of particle state, no synchronisation until the test reads `out`.
#include <hip/hip_runtime.h>

#include "hanabi_amd.h"

__global__ void k_gather_rows(HnbDeviceView v, uint32_t attr_index, uint32_t* __restrict__ out, uint32_t* __restrict__ out_count) {
    const HnbDeviceMeta m = *v.meta;                                   // device-resident counters: no host knows alive_count here
    const uint32_t* list = v.alive_list[m.list_column & 1u];
    const uint32_t head = m.list_column >> 1;                          // 0 except for effects whose list is kept as a ring (HNB_OPT_RING_LISTS)
    const HnbDeviceAttr a = v.attrs[attr_index];
    const uint32_t nc = a.ncomp;
    if (blockIdx.x == 0 && threadIdx.x == 0) *out_count = m.instance_count;
    for (uint32_t row = blockIdx.x * blockDim.x + threadIdx.x; row < m.alive_count; row += gridDim.x * blockDim.x) {
        const uint32_t slot = list[(head + row) % v.capacity];
        const uint32_t* src = reinterpret_cast<const uint32_t*>(static_cast<const char*>(a.plane) + (size_t)slot * a.stride_bytes);
        for (uint32_t c = 0; c < nc; ++c) out[(size_t)row * nc + c] = src[c];
    }
}

// The same over EVERY instance of a program through HnbProgramView (one launch for the batch, as the reference binds one buffer per batch:
// src/render/batch.rs:348-386): blockIdx.y = instance; out[(k * capacity + row) * ncomp ..], out_count[k].
__global__ void k_gather_rows_program(HnbProgramView v, uint32_t attr_index, uint32_t* __restrict__ out, uint32_t* __restrict__ out_count) {
    const uint32_t k = blockIdx.y;
    const HnbDeviceMeta m = v.meta[k];
    const char* base = reinterpret_cast<const char*>(v.slabs[k]);
    const uint32_t* list = reinterpret_cast<const uint32_t*>(base + v.alive_list_off[m.list_column & 1u]);
    const uint32_t head = m.list_column >> 1;
    const HnbProgramAttr a = v.attrs[attr_index];
    const uint32_t nc = a.ncomp;
    if (blockIdx.x == 0 && threadIdx.x == 0) out_count[k] = m.instance_count;
    for (uint32_t row = blockIdx.x * blockDim.x + threadIdx.x; row < m.alive_count; row += gridDim.x * blockDim.x) {
        const uint32_t slot = list[(head + row) % v.capacity];
        const uint32_t* src = reinterpret_cast<const uint32_t*>(base + a.plane_off + (size_t)slot * a.stride_bytes);
        for (uint32_t c = 0; c < nc; ++c) out[((size_t)k * v.capacity + row) * nc + c] = src[c];
    }
}
extern "C" int consumer_gather_program(const HnbProgramView* view, uint32_t attr, void* out, void* out_count) {
    uint32_t idx = 0xffffffffu;
    for (uint32_t i = 0; i < view->n_attrs; ++i) if (view->attrs[i].attr == attr) idx = i;
    if (idx == 0xffffffffu || view->n_instances == 0u) return -1;
    if (hipSetDevice(view->device) != hipSuccess) return -2;
    k_gather_rows_program<<<dim3(64, view->n_instances), 256, 0, static_cast<hipStream_t>(view->stream)>>>(*view, idx, static_cast<uint32_t*>(out), static_cast<uint32_t*>(out_count));
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// What a renderer's vertex stage reads per drawn particle (vfx_render.wgsl + ColorOverLifetime / SizeOverLifetime, src/modifier/output.rs:310-312):
// position, age and lifetime by alive-list row; out[row] = (position, age / lifetime). bench.py's c2_view / c2_interop_view rows run it behind
// every frame: the end-to-end cost of the reference's own firework asset.
__global__ void k_render_like(HnbDeviceView v, uint32_t i_pos, uint32_t i_age, uint32_t i_life, float4* __restrict__ out) {
    const HnbDeviceMeta m = *v.meta;
    const uint32_t* list = v.alive_list[m.list_column & 1u];
    const uint32_t head = m.list_column >> 1;
    const float* pos = static_cast<const float*>(v.attrs[i_pos].plane);
    const float* age = static_cast<const float*>(v.attrs[i_age].plane);
    const float* life = static_cast<const float*>(v.attrs[i_life].plane);
    for (uint32_t row = blockIdx.x * blockDim.x + threadIdx.x; row < m.alive_count; row += gridDim.x * blockDim.x) {
        const uint32_t slot = list[(head + row) % v.capacity];
        out[row] = make_float4(pos[3 * (size_t)slot], pos[3 * (size_t)slot + 1], pos[3 * (size_t)slot + 2], age[slot] / life[slot]);
    }
}
extern "C" int consumer_render_like(const HnbDeviceView* view, void* out) {
    uint32_t ip = 0xffffffffu, ia = 0xffffffffu, il = 0xffffffffu;
    for (uint32_t i = 0; i < view->n_attrs; ++i) {
        if (view->attrs[i].attr == HNB_ATTR_POSITION) ip = i;
        if (view->attrs[i].attr == HNB_ATTR_AGE) ia = i;
        if (view->attrs[i].attr == HNB_ATTR_LIFETIME) il = i;
    }
    if (ip == 0xffffffffu || ia == 0xffffffffu || il == 0xffffffffu) return -1;
    if (hipSetDevice(view->device) != hipSuccess) return -2;
    const uint32_t grid = (view->capacity + 1023u) / 1024u < 65536u ? (view->capacity + 1023u) / 1024u : 65536u;
    k_render_like<<<grid ? grid : 1u, 256, 0, static_cast<hipStream_t>(view->stream)>>>(*view, ip, ia, il, static_cast<float4*>(out));
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// Enqueues the gather on the VIEW's stream and returns without waiting; `out` / `out_count` are device buffers.
extern "C" int consumer_gather(const HnbDeviceView* view, uint32_t attr, void* out, void* out_count) {
    uint32_t idx = 0xffffffffu;
    for (uint32_t i = 0; i < view->n_attrs; ++i) if (view->attrs[i].attr == attr) idx = i;
    if (idx == 0xffffffffu) return -1;
    if (hipSetDevice(view->device) != hipSuccess) return -2;
    k_gather_rows<<<1024, 256, 0, static_cast<hipStream_t>(view->stream)>>>(*view, idx, static_cast<uint32_t*>(out), static_cast<uint32_t*>(out_count));
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
