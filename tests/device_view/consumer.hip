// A device-side consumer of the simulation's output, written against include/hanabi_amd.h ALONE (tests/test_device_view.py): what a renderer's
// vertex stage does with the reference's buffers (vfx_render.wgsl: particle_buffer[indirect_buffer[..]]) - gather an attribute by alive-list
// row - through HnbDeviceView. No host copy of particle state, no synchronisation until the test reads `out`.
#include <hip/hip_runtime.h>

#include "hanabi_amd.h"

__global__ void k_gather_rows(HnbDeviceView v, uint32_t attr_index, uint32_t* __restrict__ out, uint32_t* __restrict__ out_count) {
    const HnbDeviceMeta m = *v.meta;                                   // device-resident counters: no host knows alive_count here
    const uint32_t* list = v.alive_list[m.list_column & 1u];
    const HnbDeviceAttr a = v.attrs[attr_index];
    const uint32_t nc = a.ncomp;
    if (blockIdx.x == 0 && threadIdx.x == 0) *out_count = m.instance_count;
    for (uint32_t row = blockIdx.x * blockDim.x + threadIdx.x; row < m.alive_count; row += gridDim.x * blockDim.x) {
        const uint32_t slot = list[row];
        const uint32_t* src = reinterpret_cast<const uint32_t*>(static_cast<const char*>(a.plane) + (size_t)slot * a.stride_bytes);
        for (uint32_t c = 0; c < nc; ++c) out[(size_t)row * nc + c] = src[c];
    }
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
