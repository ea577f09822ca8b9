// Device-side data contracts shared by the kernels and the runtime.
// Counterparts of the reference's GPU structs:
//   GpuSpawnerParams   src/render/mod.rs:381-449   -> DevFrameInst (+ parameter block), re-uploaded per frame
//   GpuEffectMetadata  src/render/mod.rs:566-622   -> DevMeta (double-buffered, device-resident)
//   GpuIndirectIndex   src/render/mod.rs:139-146   -> three separate u32 arrays (ping, pong, dead)
//   Particle AoS       src/attributes.rs:1516-1670 -> one packed plane per attribute (SoA)
#pragma once
#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif
#include "hnb_vm.h"

namespace hnb {

constexpr uint32_t kMaxAttrs = 40;
constexpr uint32_t kBlock = 256;      // threads per workgroup (4 waves)
#ifndef HNB_CHUNK
#define HNB_CHUNK 4096
#endif
constexpr uint32_t kChunk = HNB_CHUNK;  // alive-list rows per claimed chunk = unit of the cross-chunk scan
constexpr uint32_t kInitBlock = 256;  // init: one particle per thread

typedef AttrDesc DevAttr;

struct DevProgram {
    uint32_t capacity;
    uint32_t n_attrs;          // stored attributes (all written by init)
    uint32_t n_uregs;          // parameter-block words per instance
    uint32_t chunks_per_inst;  // ceil(capacity / kChunk)
    uint32_t alive_off[2];     // byte offsets of the ping/pong alive lists
    uint32_t dead_off;         // byte offset of the dead list
    uint32_t init_len, update_len;
    uint32_t n_inst;
    DevAttr attrs[kMaxAttrs];
    const Ins* init_code;
    const Ins* update_code;
};

// Per instance, per frame (uploaded with one memcpy per program per frame).
struct DevFrameInst {
    uint32_t spawn_count;       // GpuSpawnerParams::spawn
    uint32_t seed;              // GpuSpawnerParams::seed
    uint32_t slot_base;         // particle index offset for PRNG / ID (capacity-slab sharding)
    uint32_t init_block_start;  // first init workgroup of this instance (CPU prefix sum, batch.rs:348-386)
    float xf[12];               // row-major 3x4 transform
};
static_assert(sizeof(DevFrameInst) == 64, "DevFrameInst layout");

// Device-resident per-instance counters; frame f reads [f&1] and writes [(f+1)&1].
struct DevMeta {
    uint32_t alive_count;
    uint32_t particle_counter;
    uint32_t write_index;     // list column that currently holds the alive list (flips only when particles died)
    uint32_t max_update;
    uint32_t dead_count;
    uint32_t spawned;
    uint32_t ref_write_index; // EffectMetadata::indirect_write_index as the reference would report it (flips every frame)
    uint32_t instance_count;
};
static_assert(sizeof(DevMeta) == 32, "DevMeta layout");


}  // namespace hnb
