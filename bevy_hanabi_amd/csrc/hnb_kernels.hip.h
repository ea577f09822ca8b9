// HIP kernels of the particle hot path for gfx950 (MI355X, CDNA4).
//
//   k_init    replaces vfx_init.wgsl:101-196      (spawn: pop dead slot, run INIT program, append to alive list)
//   k_update  replaces vfx_indirect.wgsl:30-90 + vfx_prefix_sum.wgsl:13-43 + vfx_update.wgsl:105-167
//             (age/reap/modifiers/Euler, kill, and alive/dead list rebuild)
//
// Design notes (MI355X-first):
//  * SoA: one packed plane per attribute. In the streaming kernel a lane owns 4 consecutive
//    alive-list entries, so on the dense path every access is a 16-byte dwordx4.
//  * Uniform sub-expressions never reach the GPU as code: the host evaluates them into a
//    per-instance parameter block that the kernels read with scalar loads.
//  * The reference rebuilds the alive list with 1-3 global atomics per particle
//    (vfx_update.wgsl:148-166). Here each 4096-particle chunk compacts survivors and
//    casualties in LDS (wave prefix scan via cross-lane shuffles + 4-wave LDS combine), then a
//    single-pass decoupled look-back across chunks gives the global offsets; the lists
//    are written coalesced and in serial (stable) order. No per-particle atomics.
//  * HIP has no indirect dispatch: chunks are handed out by a ticket counter (forward
//    progress under any dispatch order) and sized from device-resident counters, so no
//    readback and no vfx_indirect / vfx_prefix_sum launches are needed.
#pragma once
#include <hip/hip_runtime.h>
#include "hnb_dev.h"

namespace hnb {

struct u2_t { uint32_t x, y; };
struct u3_t { uint32_t x, y, z; };
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
// Streaming accesses use the default cache policy: `nontemporal` hints measured 13 % SLOWER on
// MI355X for this kernel (profiles/r01_variants.md).
#ifdef HNB_NONTEMPORAL
#define HNB_NT_LOAD(p) __builtin_nontemporal_load(p)
#define HNB_NT_STORE(v, p) __builtin_nontemporal_store(v, p)
#else
#define HNB_NT_LOAD(p) (*(p))
#define HNB_NT_STORE(v, p) (*(p) = (v))
#endif

// ---- reset: dead_index[i] = i (effect_cache.rs:298-323) ------------------------------------
__global__ void k_reset_lists(uint32_t* __restrict__ dead, uint32_t* __restrict__ alive, uint32_t capacity) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < capacity) { dead[i] = i; alive[i] = 0u; }
}

// ---- V-file attribute access (generic kernels, one particle per lane) -----------------------
__device__ __forceinline__ void vfile_store_attr(const vreg_file_t& r, uint32_t ncomp, uint32_t reg, char* plane, uint32_t slot) {
    switch (ncomp) {
        case 1: reinterpret_cast<uint32_t*>(plane)[slot] = r[reg]; break;
        case 2: reinterpret_cast<u2_t*>(plane)[slot] = u2_t{r[reg], r[reg + 1]}; break;
        case 3: reinterpret_cast<u3_t*>(plane)[slot] = u3_t{r[reg], r[reg + 1], r[reg + 2]}; break;
        default: reinterpret_cast<uint4*>(plane)[slot] = make_uint4(r[reg], r[reg + 1], r[reg + 2], r[reg + 3]); break;
    }
}
// Returns the loaded components; the caller writes them at ONE indexed store site.
__device__ __forceinline__ Out4 vfile_load_attr(uint32_t ncomp, const char* plane, uint32_t slot) {
    Out4 o = Out4{0u, 0u, 0u, 0u};
    switch (ncomp) {
        case 1: o.v0 = reinterpret_cast<const uint32_t*>(plane)[slot]; break;
        case 2: { const u2_t t = reinterpret_cast<const u2_t*>(plane)[slot]; o.v0 = t.x; o.v1 = t.y; } break;
        case 3: { const u3_t t = reinterpret_cast<const u3_t*>(plane)[slot]; o.v0 = t.x; o.v1 = t.y; o.v2 = t.z; } break;
        default: { const uint4 t = reinterpret_cast<const uint4*>(plane)[slot]; o.v0 = t.x; o.v1 = t.y; o.v2 = t.z; o.v3 = t.w; } break;
    }
    return o;
}

// ---- init -----------------------------------------------------------------------------------
// One thread per spawned particle. Thread i of instance k (serial order == thread order):
//   slot = dead[alive0 + i]; seed = pcg_hash(slot ^ spawner.seed); run INIT; alive[w][alive0+i] = slot.
// vfx_init.wgsl:141-143 uses atomicAdd(alive_count): under serial execution thread i gets
// alive0 + i, which is what is computed here without atomics. Counters are advanced by k_update.
__global__ void __launch_bounds__(kInitBlock)
k_init(const DevProgram prog, const uint64_t* __restrict__ inst_base, const DevMeta* __restrict__ meta_in,
       const DevFrameInst* __restrict__ fi, const uint32_t* __restrict__ ublocks) {
    // Find the instance owning this workgroup: binary search over the CPU prefix sum of
    // init workgroups (find_location_from_particle, vfx_init.wgsl:51-72, at workgroup granularity).
    const uint32_t blk = blockIdx.x;
    uint32_t lo = 0, hi = prog.n_inst;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (blk >= fi[mid].init_block_start) lo = mid + 1; else hi = mid;
    }
    const uint32_t k = lo - 1;
    const uint32_t i = (blk - fi[k].init_block_start) * kInitBlock + threadIdx.x;

    const uint32_t alive0 = meta_in[k].alive_count;
    const uint32_t max_spawn = prog.capacity - alive0;
    const uint32_t spawn = fi[k].spawn_count;
    const uint32_t n_spawn = spawn < max_spawn ? spawn : max_spawn;
    if (i >= n_spawn) return;

    char* base = reinterpret_cast<char*>(inst_base[k]);
    const uint32_t* dead = reinterpret_cast<const uint32_t*>(base + prog.dead_off);
    uint32_t* alive = reinterpret_cast<uint32_t*>(base + prog.alive_off[0]);  // single list, compacted in place by k_update

    const uint32_t slot = dead[alive0 + i];

    VmUniforms U;
    U.u = ublocks + (size_t)k * prog.n_uregs;
    U.xf = fi[k].xf;
    VmState<vreg_file_t> S;
    S.r = vreg_file_t{};  // var particle = Particle();  (vfx_init.wgsl:174)
    S.pindex = slot + fi[k].slot_base;
    S.seed = pcg_hash(S.pindex ^ fi[k].seed);
    S.pcounter = meta_in[k].particle_counter + i;
    S.alive = true;

    VmAttrIO io;
    io.slab = base; io.attrs = prog.attrs; io.slot = slot;
    // var particle = Particle(): attributes the INIT program never assigns are stored as zero
    for (uint32_t a = 0; a < prog.n_attrs; ++a) {
        if (prog.attrs[a].reg != HNB_REG_NONE) continue;
        uint32_t* p = vm_attr_ptr(io, a);
        for (uint32_t c = 0; c < prog.attrs[a].ncomp; ++c) p[c] = 0u;
    }

    vm_run<true, false>(prog.init_code, prog.init_len, S, U, nullptr, nullptr, io);

    alive[alive0 + i] = slot;
    for (uint32_t a = 0; a < prog.n_attrs; ++a)
        if (prog.attrs[a].reg != HNB_REG_NONE)
            vfile_store_attr(S.r, prog.attrs[a].ncomp, prog.attrs[a].reg, base + prog.attrs[a].plane_off, slot);
}

// ---- streaming-kernel pinned attribute access (P = 4) -------------------------------------------
template <int P>
__device__ __forceinline__ void pin_load3(V3 (&dst)[P], const char* plane, const uint32_t (&slot)[P], const bool (&valid)[P], bool dense) {
    if constexpr (P == 4) {
        if (dense) {
            const u4v* src = reinterpret_cast<const u4v*>(plane) + (size_t)(slot[0] >> 2) * 3;
            const u4v q0 = HNB_NT_LOAD(src), q1 = HNB_NT_LOAD(src + 1), q2 = HNB_NT_LOAD(src + 2);
            dst[0] = V3{u2f(q0.x), u2f(q0.y), u2f(q0.z)};
            dst[1] = V3{u2f(q0.w), u2f(q1.x), u2f(q1.y)};
            dst[2] = V3{u2f(q1.z), u2f(q1.w), u2f(q2.x)};
            dst[3] = V3{u2f(q2.y), u2f(q2.z), u2f(q2.w)};
            return;
        }
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
        dst[p] = V3{0.0f, 0.0f, 0.0f};
        if (valid[p]) {
            const u3_t t = reinterpret_cast<const u3_t*>(plane)[slot[p]];
            dst[p] = V3{u2f(t.x), u2f(t.y), u2f(t.z)};
        }
    }
}
template <int P>
__device__ __forceinline__ void pin_store3(const V3 (&src)[P], char* plane, const uint32_t (&slot)[P], const bool (&valid)[P], bool dense) {
    if constexpr (P == 4) {
        if (dense) {
            u4v* dst = reinterpret_cast<u4v*>(plane) + (size_t)(slot[0] >> 2) * 3;
            HNB_NT_STORE((u4v{f2u(src[0].x), f2u(src[0].y), f2u(src[0].z), f2u(src[1].x)}), dst);
            HNB_NT_STORE((u4v{f2u(src[1].y), f2u(src[1].z), f2u(src[2].x), f2u(src[2].y)}), dst + 1);
            HNB_NT_STORE((u4v{f2u(src[2].z), f2u(src[3].x), f2u(src[3].y), f2u(src[3].z)}), dst + 2);
            return;
        }
    }
#pragma unroll
    for (int p = 0; p < P; ++p)
        if (valid[p]) reinterpret_cast<u3_t*>(plane)[slot[p]] = u3_t{f2u(src[p].x), f2u(src[p].y), f2u(src[p].z)};
}
template <int P>
__device__ __forceinline__ void pin_load1(float (&dst)[P], const char* plane, const uint32_t (&slot)[P], const bool (&valid)[P], bool dense) {
    if constexpr (P == 4) {
        if (dense) {
            const u4v q = HNB_NT_LOAD(reinterpret_cast<const u4v*>(plane) + (slot[0] >> 2));
            dst[0] = u2f(q.x); dst[1] = u2f(q.y); dst[2] = u2f(q.z); dst[3] = u2f(q.w);
            return;
        }
    }
#pragma unroll
    for (int p = 0; p < P; ++p) dst[p] = valid[p] ? reinterpret_cast<const float*>(plane)[slot[p]] : 0.0f;
}
template <int P>
__device__ __forceinline__ void pin_store1(const float (&src)[P], char* plane, const uint32_t (&slot)[P], const bool (&valid)[P], bool dense) {
    if constexpr (P == 4) {
        if (dense) {
            HNB_NT_STORE((u4v{f2u(src[0]), f2u(src[1]), f2u(src[2]), f2u(src[3])}), reinterpret_cast<u4v*>(plane) + (slot[0] >> 2));
            return;
        }
    }
#pragma unroll
    for (int p = 0; p < P; ++p)
        if (valid[p]) reinterpret_cast<float*>(plane)[slot[p]] = src[p];
}

// ---- update + kill + compaction ----------------------------------------------------------------
// Cross-chunk machinery shared by the two update kernels.
//
//  * Persistent workgroups claim chunks from a ticket counter; the next ticket is requested
//    before the current chunk is processed, so its latency is hidden. Tickets are handed out in
//    order: every chunk with a smaller id has been claimed by a running workgroup, which makes
//    all waits below deadlock-free under any dispatch order (no residency assumption).
//  * The alive list is compacted IN PLACE (stable): chunk c writes rows [E, E+A) with E <= its first
//    row, and only learns E after every earlier chunk has published, i.e. finished reading its rows.
//    When nothing before or inside the chunk died, the rows are already in place and nothing is written.
//  * E comes from a two-level decoupled look-back: chunks publish aggregates; the last chunk to
//    arrive in a group of 64 publishes the group aggregate and resolves the group prefix from the
//    (few) earlier groups. A chunk then needs one 64-wide read of its group's aggregates plus the
//    previous group's prefix, instead of walking thousands of simultaneously-finishing chunks.
__device__ __forceinline__ uint64_t pack_status(uint32_t epoch, uint64_t state, uint32_t value) {
    return ((uint64_t)epoch << 34) | (state << 32) | value;
}
constexpr uint32_t kGroup = 64;  // chunks per look-back group

struct ScanBufs {
    uint64_t* chunk_status;   // [n_inst * chunks_per_inst]
    uint64_t* group_status;   // [n_inst * groups_per_inst]
    uint32_t* arrive;         // [2][n_inst * groups_per_inst], frame parity double-buffered
    uint32_t* ticket;         // [0..1] ticket per parity, [2] watchdog word
    uint32_t groups_per_inst;
    uint32_t n_groups_total;
    uint32_t parity;
    uint32_t epoch;
};

struct ChunkCtx {
    uint32_t k, j;          // instance, chunk within instance
    uint32_t n;             // max_update of the instance
    uint32_t n_spawn;
    uint32_t start;         // first alive-list row of this chunk
    DevMeta m;
    char* base;
    uint32_t* alive;        // the instance's alive list (compacted in place)
    uint32_t* dead;
};

__device__ __forceinline__ uint32_t claim_ticket(const ScanBufs& sb) {
    return __hip_atomic_fetch_add(&sb.ticket[sb.parity], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Whoever draws ticket 0 prepares the counters of the NEXT frame (other parity): nothing of this
// launch or the previous one touches them.
__device__ __forceinline__ void reset_next_frame(const ScanBufs& sb) {
    uint32_t* a = sb.arrive + (size_t)(sb.parity ^ 1u) * sb.n_groups_total;
    for (uint32_t i = threadIdx.x; i < sb.n_groups_total; i += kBlock) a[i] = 0u;
    if (threadIdx.x == 0) __hip_atomic_store(&sb.ticket[sb.parity ^ 1u], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Decode a chunk id; false when the chunk has no rows (nothing to do, nobody waits on it).
template <class ARGS>
__device__ __forceinline__ bool chunk_setup(ChunkCtx& c, uint32_t chunk, const ARGS& args, const uint64_t* inst_base, const DevMeta* meta_in,
                                            DevMeta* meta_out, const DevFrameInst* fi) {
    c.k = chunk / args.chunks_per_inst;
    c.j = chunk - c.k * args.chunks_per_inst;
    // vfx_indirect.wgsl:57-85 folded in: max_update = alive_count after init.
    c.m = meta_in[c.k];
    const uint32_t spawn = fi[c.k].spawn_count;
    const uint32_t max_spawn = args.capacity - c.m.alive_count;
    c.n_spawn = spawn < max_spawn ? spawn : max_spawn;
    c.n = c.m.alive_count + c.n_spawn;
    c.start = c.j * kChunk;
    if (c.n == 0) {
        if (c.j == 0 && threadIdx.x == 0) {
            DevMeta o = c.m;
            o.write_index = c.m.write_index ^ 1u; o.max_update = 0; o.dead_count = 0; o.spawned = 0; o.instance_count = 0;
            meta_out[c.k] = o;
        }
        return false;
    }
    if (c.start >= c.n) return false;
    c.base = reinterpret_cast<char*>(inst_base[c.k]);
    c.alive = reinterpret_cast<uint32_t*>(c.base + args.alive_off[0]);
    c.dead = reinterpret_cast<uint32_t*>(c.base + args.dead_off);
    return true;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (uint32_t off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// Spin until the 64-bit word carries this frame's epoch (bounded: the watchdog reports a fault
// instead of hanging the GPU). Wave-uniform exit.
__device__ __forceinline__ uint64_t wait_word(const uint64_t* w, bool active, uint32_t epoch, uint32_t& fault) {
    uint64_t s = pack_status(epoch, kStatePrefix, 0u);
    uint32_t spins = 0;
    for (;;) {
        if (active) s = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool ready = (uint32_t)(s >> 34) == epoch;
        if (__all(ready)) break;
        if (++spins > (1u << 22)) { fault = 1u; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    return s;
}

// Publish this chunk's survivor count (wave 0). The last chunk of a group to arrive also publishes
// the group aggregate and resolves the group's prefix from the earlier groups. Nothing here waits on
// a chunk that has not been claimed, and processing a claimed chunk never waits at all.
__device__ __forceinline__ void publish_chunk(const ChunkCtx& c, const ScanBufs& sb, uint32_t chunks_per_inst, uint32_t local_alive, uint32_t& fault) {
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t* cst = sb.chunk_status + (size_t)c.k * chunks_per_inst;
    uint64_t* gst = sb.group_status + (size_t)c.k * sb.groups_per_inst;
    const uint32_t j = c.j, g = j / kGroup;
    const uint32_t n_chunks = (c.n + kChunk - 1) / kChunk;  // chunks of this instance that have rows
    const uint32_t g_count = (n_chunks - g * kGroup) < kGroup ? (n_chunks - g * kGroup) : kGroup;
    uint32_t old = 0;
    if (lane == 0) {
        __hip_atomic_store(&cst[j], pack_status(sb.epoch, kStateAggregate, local_alive), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = __hip_atomic_fetch_add(&sb.arrive[(size_t)sb.parity * sb.n_groups_total + (size_t)c.k * sb.groups_per_inst + g], 1u, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
    }
    old = __shfl(old, 0, 64);
    if (old + 1u != g_count) return;  // not the last arriver (election only: the data itself is polled)
    const uint64_t s = wait_word(&cst[g * kGroup + lane], lane < g_count, sb.epoch, fault);
    const uint32_t group_sum = wave_sum(lane < g_count ? (uint32_t)s : 0u);
    if (g == 0) {
        if (lane == 0) __hip_atomic_store(&gst[0], pack_status(sb.epoch, kStatePrefix, group_sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (lane == 0) __hip_atomic_store(&gst[g], pack_status(sb.epoch, kStateAggregate, group_sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // decoupled look-back over earlier GROUPS (lanes past the first group carry a virtual prefix 0)
    uint32_t group_excl = 0;
    int hi = (int)g - 1;
    while (hi >= 0 && !fault) {
        const int idx = hi - (int)lane;
        const uint64_t q = wait_word(&gst[idx >= 0 ? idx : 0], idx >= 0, sb.epoch, fault);
        const bool is_prefix = idx < 0 || ((q >> 32) & 3u) == kStatePrefix;
        const uint32_t val = idx >= 0 ? (uint32_t)q : 0u;
        const uint64_t pmask = __ballot(is_prefix);
        if (pmask) {
            const uint32_t first = (uint32_t)__builtin_ctzll(pmask);
            group_excl += wave_sum(lane <= first ? val : 0u);
            break;
        }
        group_excl += wave_sum(val);
        hi -= 64;
    }
    if (lane == 0) __hip_atomic_store(&gst[g], pack_status(sb.epoch, kStatePrefix, group_excl + group_sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Exclusive prefix of a published chunk (wave 0): survivors of all earlier chunks of the instance.
__device__ __forceinline__ uint32_t resolve_chunk(const ChunkCtx& c, const ScanBufs& sb, uint32_t chunks_per_inst, uint32_t& fault) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t* cst = sb.chunk_status + (size_t)c.k * chunks_per_inst;
    const uint64_t* gst = sb.group_status + (size_t)c.k * sb.groups_per_inst;
    const uint32_t g = c.j / kGroup, jg = c.j - g * kGroup;
    const uint64_t s = wait_word(&cst[g * kGroup + lane], lane < jg, sb.epoch, fault);
    const uint32_t intra = wave_sum(lane < jg ? (uint32_t)s : 0u);
    uint32_t group_excl = 0;
    if (g > 0) {  // inclusive prefix of the previous group
        uint64_t q;
        uint32_t spins = 0;
        for (;;) {
            q = __hip_atomic_load(&gst[g - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((uint32_t)(q >> 34) == sb.epoch && ((q >> 32) & 3u) == kStatePrefix) break;
            if (++spins > (1u << 22)) { fault = 1u; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        group_excl = (uint32_t)q;
    }
    return group_excl + intra;
}

// Staged chunk: survivors (front) and casualties (back) of NSEG LDS segments of seg_rows rows each;
// s_cnt[w] = survivors | casualties << 16 of segment w.
template <int NSEG>
__device__ __forceinline__ void chunk_publish(const ChunkCtx& c, const ScanBufs& sb, uint32_t chunks_per_inst, const uint32_t* s_cnt, uint32_t* s_bcast) {
    if ((threadIdx.x >> 6) != 0) return;
    uint32_t local_alive = 0;
#pragma unroll
    for (int w = 0; w < NSEG; ++w) local_alive += s_cnt[w] & 0xffffu;
    uint32_t fault = 0;
    publish_chunk(c, sb, chunks_per_inst, local_alive, fault);
    if ((threadIdx.x & 63u) == 0 && fault) atomicOr(&sb.ticket[2], 1u);  // watchdog word, reported by hnb_effect_metadata
}

template <int NSEG>
__device__ __forceinline__ void chunk_commit(const ChunkCtx& c, const ScanBufs& sb, uint32_t chunks_per_inst, DevMeta* meta_out, const uint32_t* s_list,
                                             uint32_t seg_rows, const uint32_t* s_cnt, uint32_t* s_bcast) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t a[NSEG], d[NSEG];
    uint32_t local_alive = 0, local_dead = 0;
#pragma unroll
    for (int w = 0; w < NSEG; ++w) { a[w] = s_cnt[w] & 0xffffu; d[w] = s_cnt[w] >> 16; local_alive += a[w]; local_dead += d[w]; }
    if (wave == 0) {
        uint32_t fault = 0;
        const uint32_t excl = resolve_chunk(c, sb, chunks_per_inst, fault);
        if (lane == 0) {
            s_bcast[0] = excl; s_bcast[1] = fault;
            if (fault) atomicOr(&sb.ticket[2], 1u);
        }
    }
    __syncthreads();
    const uint32_t excl_prefix = s_bcast[0];
    // Survivors in stable (serial) order (vfx_update.wgsl:161-165). Rows already in place are not rewritten.
    if (!(excl_prefix == c.start && local_dead == 0u)) {
        uint32_t abase = excl_prefix;
#pragma unroll
        for (int w = 0; w < NSEG; ++w) {
            const uint32_t* sg = s_list + w * seg_rows;
            for (uint32_t i = tid; i < a[w]; i += kBlock) c.alive[abase + i] = sg[i];
            abase += a[w];
        }
    }
    // The d-th casualty in serial order lands on dead row n-1-d (vfx_update.wgsl:150-151).
    if (local_dead) {
        uint32_t dbase = c.start - excl_prefix;
#pragma unroll
        for (int w = 0; w < NSEG; ++w) {
            const uint32_t* sg = s_list + w * seg_rows;
            for (uint32_t i = tid; i < d[w]; i += kBlock) c.dead[c.n - 1u - (dbase + i)] = sg[seg_rows - 1u - i];
            dbase += d[w];
        }
    }
    if (tid == 0 && c.start + kChunk >= c.n) {
        const uint32_t survivors = excl_prefix + local_alive;
        DevMeta o;
        o.alive_count = survivors;
        o.particle_counter = c.m.particle_counter + c.n_spawn;
        o.write_index = c.m.write_index ^ 1u;
        o.max_update = c.n;
        o.dead_count = c.n - survivors;
        o.spawned = c.n_spawn;
        o.fault = c.m.fault | s_bcast[1];
        o.instance_count = survivors;
        meta_out[c.k] = o;
    }
}

// Persistent-workgroup driver, software-pipelined at chunk granularity:
//   PROCESS(chunk, buf) -> bool : stage the chunk into LDS buffer `buf` and publish its aggregate
//   COMMIT(chunk, buf)          : resolve the prefix and write the lists of a chunk staged earlier
// The commit of chunk i runs after chunk i+1 has been processed, so the look-back words it needs
// were published a whole chunk-time ago, and the next ticket (requested before processing) has
// long arrived: neither latency is exposed except once at the tail.
#define HNB_PERSISTENT_LOOP(sb, total_chunks, s_bcast, PROCESS, COMMIT)                          \
    {                                                                                            \
        uint32_t t_next_ = 0, prev_ = 0xffffffffu, buf_ = 0;                                     \
        if (threadIdx.x == 0) s_bcast[2] = claim_ticket(sb);                                     \
        __syncthreads();                                                                         \
        uint32_t chunk_ = s_bcast[2];                                                            \
        if (chunk_ == 0) reset_next_frame(sb);                                                   \
        while (chunk_ < (total_chunks)) {                                                        \
            if (threadIdx.x == 0) t_next_ = claim_ticket(sb);                                    \
            const bool staged_ = PROCESS(chunk_, buf_);                                          \
            if (prev_ != 0xffffffffu) { COMMIT(prev_, buf_ ^ 1u); }                              \
            if (staged_) { prev_ = chunk_; buf_ ^= 1u; } else { prev_ = 0xffffffffu; }           \
            __syncthreads();                                                                     \
            if (threadIdx.x == 0) s_bcast[2] = t_next_;                                          \
            __syncthreads();                                                                     \
            chunk_ = s_bcast[2];                                                                 \
        }                                                                                        \
        if (prev_ != 0xffffffffu) { COMMIT(prev_, buf_ ^ 1u); }                                  \
    }

// ---- generic update kernel: any update stream, V register file, one particle per lane ----------
__device__ __forceinline__ bool generic_process(uint32_t chunk, const DevProgram& prog, const uint64_t* inst_base, const DevMeta* meta_in,
                                                DevMeta* meta_out, const DevFrameInst* fi, const uint32_t* ublocks, const ScanBufs& sb,
                                                uint32_t* s_list, uint32_t* s_cnt, uint32_t* s_wave, uint32_t* s_bcast) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    ChunkCtx c;
    if (!chunk_setup(c, chunk, prog, inst_base, meta_in, meta_out, fi)) return false;
    const uint32_t seed_k = fi[c.k].seed, slot_base = fi[c.k].slot_base;
    VmUniforms U;
    U.u = ublocks + (size_t)c.k * prog.n_uregs;
    U.xf = fi[c.k].xf;
    uint32_t local_alive = 0, local_dead = 0;
    for (uint32_t sub = 0; sub < kChunk / kBlock; ++sub) {
        const uint32_t li = c.start + sub * kBlock + tid;
        if (c.start + sub * kBlock >= c.n) break;
        const bool valid = li < c.n;
        const uint32_t slot = valid ? c.alive[li] : 0u;
        VmState<vreg_file_t> S;
        S.r = vreg_file_t{};
        for (uint32_t a = 0; a < prog.n_attrs; ++a) {
            const DevAttr at = prog.attrs[a];
            if (!(at.upd_flags & HNB_ATTR_UPD_LOAD) || at.reg == HNB_REG_NONE) continue;
            Out4 o = Out4{0u, 0u, 0u, 0u};
            if (valid) o = vfile_load_attr(at.ncomp, c.base + at.plane_off, slot);
            for (uint32_t cc = 0; cc < at.ncomp; ++cc) S.r[at.reg + cc] = out4_get(o, cc);  // single indexed store site
        }
        S.pindex = slot + slot_base;
        S.seed = pcg_hash(S.pindex ^ seed_k);  // vfx_update.wgsl:138
        S.pcounter = 0u;
        S.alive = true;
        VmAttrIO io;
        io.slab = c.base; io.attrs = prog.attrs; io.slot = slot;
        if (valid) vm_run<true, false>(prog.update_code, prog.update_len, S, U, nullptr, nullptr, io);
        if (valid) {
            for (uint32_t a = 0; a < prog.n_attrs; ++a) {
                const DevAttr at = prog.attrs[a];
                if ((at.upd_flags & HNB_ATTR_UPD_STORE) && at.reg != HNB_REG_NONE) vfile_store_attr(S.r, at.ncomp, at.reg, c.base + at.plane_off, slot);
            }
        }
        // chunk-local stable compaction in LDS
        const uint32_t x = (valid && S.alive ? 1u : 0u) | ((valid && !S.alive ? 1u : 0u) << 16);
        uint32_t incl = x;
#pragma unroll
        for (uint32_t off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(incl, off, 64);
            if (lane >= off) incl += y;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < kBlock / 64; ++w) {
            const uint32_t t = s_wave[w];
            if (w < wave) wbase += t;
            total += t;
        }
        const uint32_t excl = wbase + incl - x;
        if (valid) {
            if (S.alive) s_list[local_alive + (excl & 0xffffu)] = slot;
            else s_list[kChunk - 1u - (local_dead + (excl >> 16))] = slot;
        }
        local_alive += total & 0xffffu;
        local_dead += total >> 16;
        __syncthreads();
    }
    if (tid == 0) s_cnt[0] = local_alive | (local_dead << 16);
    __syncthreads();
    chunk_publish<1>(c, sb, prog.chunks_per_inst, s_cnt, s_bcast);
    return true;
}

__global__ void __launch_bounds__(kBlock)
k_update_generic(const DevProgram prog, const uint64_t* __restrict__ inst_base, const DevMeta* __restrict__ meta_in,
                 DevMeta* __restrict__ meta_out, const DevFrameInst* __restrict__ fi, const uint32_t* __restrict__ ublocks, const ScanBufs sb) {
    // 65536-row chunks would not fit twice: the generic kernel stages 4096 rows per buffer like the streaming one
    __shared__ uint32_t s_list[2][kChunk];
    __shared__ uint32_t s_cnt[2][1];
    __shared__ uint32_t s_wave[kBlock / 64];
    __shared__ uint32_t s_bcast[3];
    const uint32_t total = prog.n_inst * prog.chunks_per_inst;
#define PROCESS_(ch, buf) generic_process(ch, prog, inst_base, meta_in, meta_out, fi, ublocks, sb, s_list[buf], s_cnt[buf], s_wave, s_bcast)
#define COMMIT_(ch, buf)                                                                                     \
    {                                                                                                        \
        ChunkCtx pc_;                                                                                        \
        chunk_setup(pc_, ch, prog, inst_base, meta_in, meta_out, fi);                                        \
        chunk_commit<1>(pc_, sb, prog.chunks_per_inst, meta_out, s_list[buf], kChunk, s_cnt[buf], s_bcast);  \
    }
    HNB_PERSISTENT_LOOP(sb, total, s_bcast, PROCESS_, COMMIT_)
#undef PROCESS_
#undef COMMIT_
}

// ---- streaming update kernel ---------------------------------------------------------------------
// Macro-op update streams with U operands, named registers, 4 particles per lane.
//
// Work decomposition (all choices measured on MI355X):
//  * a workgroup owns a 4096-row chunk of the alive list; each of its 4 WAVES owns a private,
//    contiguous 1024-row quarter and walks it in 4 steps of 256 rows (64 lanes x 4 rows, so the
//    dense path moves 16 B per lane per access);
//  * survivors / casualties are ranked inside the wave with ballots (no shuffles, no LDS) and go
//    straight into the wave's own LDS segment, so the loop contains NO workgroup barrier and
//    keeps only 4 slot indices live: registers stay low enough for 8 waves per SIMD;
//  * the workgroup synchronises once, combines the 4 wave totals, resolves the cross-chunk prefix
//    and writes the lists coalesced (or not at all when nothing moved).
struct StreamArgs {
    uint32_t capacity, n_uregs, chunks_per_inst, n_inst;
    uint32_t alive_off[2], dead_off, update_len;
    uint32_t plane_off[4];   // position, velocity, age, lifetime
    uint32_t flags;          // bit i: load pinned attr i; bit 4+i: store pinned attr i
    const Ins* update_code;
};

#ifndef HNB_STREAM_WAVES
#define HNB_STREAM_WAVES 8   // waves per SIMD the lean streaming kernel is register-budgeted for
#endif
#ifndef HNB_STREAM_WAVES_FULL
#define HNB_STREAM_WAVES_FULL 5
#endif
constexpr uint32_t kWaveRows = kChunk / (kBlock / 64);  // 1024 rows per wave
constexpr uint32_t kStepRows = 64 * 4;                  // 256 rows per wave step

// PROBE (tools/stream_probe.hip only; 0 in the product): ablation bits used to attribute the kernel's
// time: 2 = skip prefix resolution + list writes, 4 = skip stores, 8 = skip the program,
// 16 = skip the alive-list read (assume identity).
template <class PROG, int PROBE>
__device__ __forceinline__ bool stream_process(uint32_t chunk, const StreamArgs& args, const uint64_t* inst_base, const DevMeta* meta_in,
                                               DevMeta* meta_out, const DevFrameInst* fi, const uint32_t* ublocks, const ScanBufs& sb,
                                               uint32_t* s_list, uint32_t* s_wave, uint32_t* s_bcast) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    ChunkCtx c;
    if (!chunk_setup(c, chunk, args, inst_base, meta_in, meta_out, fi)) return false;
    const uint32_t n = c.n;
    VmUniforms U;
    U.u = ublocks + (size_t)c.k * args.n_uregs;
    U.xf = fi[c.k].xf;
    char* p_pos = c.base + args.plane_off[0];
    char* p_vel = c.base + args.plane_off[1];
    char* p_age = c.base + args.plane_off[2];
    char* p_life = c.base + args.plane_off[3];
    const uint32_t fl = args.flags;
    const uint32_t* alive_rd = c.alive;

    // ---- the wave's private quarter ------------------------------------------------------------------
    const uint32_t wstart = c.start + wave * kWaveRows;
    uint32_t* seg = s_list + wave * kWaveRows;
    uint32_t wa = 0, wd = 0;  // wave-uniform survivor / casualty counts of this quarter
    const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll 1
    for (uint32_t step = 0; step < kWaveRows / kStepRows; ++step) {
        const uint32_t sbase = wstart + step * kStepRows;
        if (sbase >= n) break;
        const uint32_t li = sbase + lane * 4u;
        uint32_t slot[4];
        bool valid[4];
        if constexpr (PROBE & 16) {
#pragma unroll
            for (int p = 0; p < 4; ++p) { valid[p] = li + p < n; slot[p] = li + p; }
        } else if (li + 4u <= n) {
            const uint4 q = *reinterpret_cast<const uint4*>(alive_rd + li);
            slot[0] = q.x; slot[1] = q.y; slot[2] = q.z; slot[3] = q.w;
#pragma unroll
            for (int p = 0; p < 4; ++p) valid[p] = true;
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                valid[p] = li + p < n;
                slot[p] = valid[p] ? alive_rd[li + p] : 0u;
            }
        }
        const bool quad = valid[3] && ((slot[0] & 3u) == 0u) && slot[1] == slot[0] + 1u && slot[2] == slot[0] + 2u && slot[3] == slot[0] + 3u;
        const bool dense = __all(quad);  // wave-uniform: all 64 lanes own an aligned run of 4 slots

        Pinned<4> X;
#pragma unroll
        for (int p = 0; p < 4; ++p) { X.pos[p] = V3{0, 0, 0}; X.vel[p] = V3{0, 0, 0}; X.age[p] = 0.0f; X.lifetime[p] = 0.0f; X.alive[p] = true; }
        if (fl & 1u) pin_load3<4>(X.pos, p_pos, slot, valid, dense);
        if (fl & 2u) pin_load3<4>(X.vel, p_vel, slot, valid, dense);
        if (fl & 4u) pin_load1<4>(X.age, p_age, slot, valid, dense);
        if (fl & 8u) pin_load1<4>(X.lifetime, p_life, slot, valid, dense);

        if constexpr (!(PROBE & 8)) PROG::template run<4>(args.update_code, args.update_len, X, U);

        if constexpr (!(PROBE & 4)) {
            if (fl & 16u) pin_store3<4>(X.pos, p_pos, slot, valid, dense);
            if (fl & 32u) pin_store3<4>(X.vel, p_vel, slot, valid, dense);
            if (fl & 64u) pin_store1<4>(X.age, p_age, slot, valid, dense);
            if (fl & 128u) pin_store1<4>(X.lifetime, p_life, slot, valid, dense);
        } else {
            float acc = 0.0f;  // keep the loads alive
#pragma unroll
            for (int p = 0; p < 4; ++p) acc += X.pos[p].x + X.pos[p].y + X.pos[p].z + X.vel[p].x + X.vel[p].y + X.vel[p].z + X.age[p] + X.lifetime[p];
            if (acc == 123.456f) X.alive[0] = false;
        }

        // wave-local stable ranks from ballots: rows are lane-major (lane l owns rows 4l..4l+3)
        uint32_t before_a = 0, before_v = 0, tot_a = 0, tot_v = 0;
        bool al[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            al[p] = valid[p] && X.alive[p];
            const uint64_t ma = __ballot(al[p]), mv = __ballot(valid[p]);
            before_a += (uint32_t)__popcll(ma & below); tot_a += (uint32_t)__popcll(ma);
            before_v += (uint32_t)__popcll(mv & below); tot_v += (uint32_t)__popcll(mv);
        }
        uint32_t ra = wa + before_a;                 // survivors of this quarter before my first row
        uint32_t rd = wd + (before_v - before_a);    // casualties before my first row
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (!valid[p]) continue;
            if (al[p]) seg[ra++] = slot[p];
            else seg[kWaveRows - 1u - (rd++)] = slot[p];
        }
        wa += tot_a;
        wd += tot_v - tot_a;
    }
    if (lane == 0) s_wave[wave] = wa | (wd << 16);
    __syncthreads();
    if constexpr (PROBE & 2) {
        if (s_wave[0] == 0xffffffffu) meta_out[c.k].fault = 1;
        return false;
    }
    chunk_publish<kBlock / 64>(c, sb, args.chunks_per_inst, s_wave, s_bcast);
    return true;
}

template <class PROG, int WAVES, int PROBE = 0>
__global__ void __launch_bounds__(kBlock, WAVES)
k_update_stream(const StreamArgs args, const uint64_t* __restrict__ inst_base, const DevMeta* __restrict__ meta_in,
                DevMeta* __restrict__ meta_out, const DevFrameInst* __restrict__ fi, const uint32_t* __restrict__ ublocks, const ScanBufs sb) {
    __shared__ uint32_t s_list[2][kChunk];           // two staged chunks: commit of one overlaps processing of the next
    __shared__ uint32_t s_wave[2][kBlock / 64];
    __shared__ uint32_t s_bcast[3];
    const uint32_t total = args.n_inst * args.chunks_per_inst;
#define PROCESS_(ch, buf) stream_process<PROG, PROBE>(ch, args, inst_base, meta_in, meta_out, fi, ublocks, sb, s_list[buf], s_wave[buf], s_bcast)
#define COMMIT_(ch, buf)                                                                                                      \
    {                                                                                                                         \
        ChunkCtx pc_;                                                                                                         \
        chunk_setup(pc_, ch, args, inst_base, meta_in, meta_out, fi);                                                         \
        chunk_commit<kBlock / 64>(pc_, sb, args.chunks_per_inst, meta_out, s_list[buf], kWaveRows, s_wave[buf], s_bcast);     \
    }
    if constexpr (PROBE & 1) {  // no ticket: one chunk per workgroup, id = blockIdx
        if (PROCESS_(blockIdx.x, 0)) { COMMIT_(blockIdx.x, 0) }
    } else {
        HNB_PERSISTENT_LOOP(sb, total, s_bcast, PROCESS_, COMMIT_)
    }
#undef PROCESS_
#undef COMMIT_
}

}  // namespace hnb
