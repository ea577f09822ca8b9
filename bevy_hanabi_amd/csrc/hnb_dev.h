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
#ifndef HNB_INIT_ROUNDS
#define HNB_INIT_ROUNDS 4
#endif
constexpr uint32_t kInitRounds = HNB_INIT_ROUNDS;   // ... and at most this many groups of kInitBlock consecutive spawns per workgroup and pass (k_init)

typedef AttrDesc DevAttr;

struct DevProgram {
    uint32_t capacity;
    uint32_t n_attrs;          // stored attributes (all written by init)
    uint32_t n_uregs;          // parameter-block words per instance
    uint32_t chunks_per_inst;  // ceil(capacity / kChunk)
    soff_t alive_off[2];       // offsets of the ping/pong alive lists
    soff_t dead_off;           // offset of the dead list
    uint32_t init_len, update_len;
    uint32_t n_inst;
    // GPU spawn events this program's update appends (EmitSpawnEventModifier): per-row staging planes in the slab
    soff_t alive_flag_off;     // u8[capacity]: 0 free, 1 alive (3: alive, spawned this frame into an age-cohort chunk): drives the slot-major update
    // List maintenance of frames with casualties (k_count_rows / k_compact): died_bits = one bit per SLOT, "died in this frame's update",
    // rewritten completely by every update launch that is followed by the list kernels (64-bit words, bit s & 63 of word s >> 6);
    // row_mask = one bit per alive-list ROW, "survives", written by k_count_rows for k_compact. Both [chunks_per_inst * kChunk / 8] bytes.
    soff_t died_bits_off, row_mask_off;
    // Row-chunk death horizons (k_init / k_count_rows / k_compact, "Death horizons" in hnb_kernels.hip.h): [double clock][pad to 256 B]
    // [u64 D[2][chunks_per_inst]][u32 BF[2][chunks_per_inst]]; horizon: 1 = the program is eligible and the arrays are maintained;
    // hz_parity: which half is current; frame_no: frames this program ran (birth frames of the rows).
    soff_t horizon_off;
    uint32_t horizon, hz_parity, frame_no;
    uint32_t n_event_channels;
    soff_t ev_cnt_off[HNB_MAX_EVENT_CHANNELS];       // u32[capacity] per channel: events appended by the particle in that slot
    // Lifetime culling (k_update_slots_stream): f32[chunks_per_inst], a lower bound of the LIFETIME of every alive particle
    // of the 4096-slot chunk, 0 = unknown. k_init zeroes the entry of a chunk it spawns into.
    soff_t lmin_off;
    uint32_t cull_lifetime;    // 1: the streaming update keeps the bounds up to date and uses them
    uint32_t age_cohort;       // 1: chunks whose alive particles share one AGE keep it in a word (hnb_kernels.hip.h "Age cohorts")
    uint32_t stream_hint;      // this frame: list traffic of k_init carries the nontemporal hint (hnb_kernels.hip.h "cache policy of streamed data")
    uint32_t ring;             // this frame is a RING frame (hnb_kernels.hip.h "Ring lists"): k_init writes its spawns in FRONT of the list's head
    DevAttr attrs[kMaxAttrs];
    const Ins* init_code;
    const Ins* update_code;
};

// Per instance, per frame (uploaded with one memcpy per program per frame).
struct DevFrameInst {
    // The first 32 bytes are what every init and list kernel reads before anything else (requested_spawn, the `frozen` test): ONE 32-byte scalar load.
    // (Until round 6 spawn_count / ev_in / ev_parity / skip lay at offsets 0 / 80 / 152 / 156 and were read one behind the other's test: three dependent
    // round trips at the head of kernels that last five; the compiler sinks separately hoisted loads back behind the branches.)
    uint32_t spawn_count;       // GpuSpawnerParams::spawn
    uint32_t seed;              // GpuSpawnerParams::seed
    uint32_t slot_base;         // particle index offset for PRNG / ID (capacity-slab sharding)
    uint32_t init_block_start;  // first init workgroup of this instance (CPU prefix sum, batch.rs:348-386)
    uint64_t ev_in;             // child: DevEventBuffer the init pass consumes (GPU spawn events: GpuSpawnerParams::parent_slab_offset, GpuChildInfo; src/render/event.rs:200-214)
    uint32_t ev_parity;         // context frame parity: events are appended to count[ev_parity], consumed from count[ev_parity ^ 1]
    uint32_t skip;              // 1: the instance is not simulated this frame (SimulationCondition::WhenVisible and not visible): state frozen
    float xf[12];               // row-major 3x4 transform
    uint64_t parent_base;       // child: slab of the parent instance, 0 = CPU-spawned effect
    uint64_t parent_planes;     // child: u32[HNB_ATTR_COUNT] plane offsets of the parent layout in 256-byte units (kNoPlane = absent)
    uint64_t ev_out[HNB_MAX_EVENT_CHANNELS];  // parent: DevEventBuffer per child channel, 0 = nobody listens
};
static_assert(sizeof(DevFrameInst) == 96 + 8 * HNB_MAX_EVENT_CHANNELS && sizeof(DevFrameInst) % 32 == 0, "DevFrameInst layout: rows are 32-byte aligned (the parameter blocks are 256-byte aligned)");

// Spawn events of one (parent instance, channel). `count` keeps growing past the capacity like the
// reference's GpuChildInfo::event_count (src/lib.rs:976-993); it is double-buffered by frame parity so
// that every kernel of frame N+1 sees the number of events frame N appended, whatever the launch order.
struct DevEventBuffer {
    uint32_t count[2];
    uint32_t capacity;      // arrayLength(&event_buffer.spawn_events)
    uint32_t pad;
    uint32_t data[1];       // [capacity] spawn_events[i].particle_index (slot in the parent slab)
};

// Device-resident per-instance counters; frame f reads [f&1] and writes [(f+1)&1].
struct DevMeta {
    uint32_t alive_count;
    uint32_t particle_counter;
    uint32_t write_index;     // bit 0: the list column that currently holds the alive list (flips only when particles died);
                              // bits 1..31: list_head - row r of the list is column[(list_head + r) % capacity]. 0 except for ribbon effects whose
                              // list is kept as a RING (hnb_kernels.hip.h "Ring lists"): list_of() / ring_row() below
    uint32_t max_update;
    uint32_t dead_count;
    uint32_t spawned;
    uint32_t ref_write_index; // EffectMetadata::indirect_write_index as the reference would report it (flips every frame)
    uint32_t instance_count;
};
static_assert(sizeof(DevMeta) == 32, "DevMeta layout");
HNB_HD uint32_t list_column(uint32_t write_index) { return write_index & 1u; }
HNB_HD uint32_t list_head(uint32_t write_index) { return write_index >> 1; }
// physical index of list row `head + r` (head < capacity, r < capacity: one conditional subtraction, no division)
HNB_HD uint32_t ring_row(uint32_t head, uint32_t r, uint32_t capacity) { const uint32_t i = head + r; return (i >= capacity || i < head) ? i - capacity : i; }


}  // namespace hnb
