// HIP kernels of the particle hot path for gfx950 (MI355X, CDNA4).
//
//   k_init                  replaces vfx_init.wgsl:101-196   (spawn: pop dead slot, run INIT program, append to alive list)
//   k_update_slots_stream   replaces vfx_update.wgsl:105-167 for update programs made of macro ops on POSITION / VELOCITY /
//   k_update_slots_generic  AGE / LIFETIME (stream) or any program (generic): age / reap / modifiers / Euler / kill test
//   k_count_rows, k_compact the alive / dead list rebuild of vfx_update.wgsl:148-166 and the counter rotation of
//                           vfx_indirect.wgsl:30-90 + vfx_prefix_sum.wgsl:13-43, only where particles died
//   k_emit_count/_events    append_spawn_events_N (src/lib.rs:976-993) in serial order
//
// Design notes (MI355X-first):
//  * SoA: one packed plane per attribute. The update walks the SLOTS, driven by one alive byte per slot (0 free, 1 alive):
//    a lane owns 4 consecutive slots, so every attribute access is a 16-byte dwordx4 whatever the alive list looks like
//    after hours of spawn / kill churn; the update never touches the lists. What it leaves for them is one bit per slot,
//    "died in this frame" (2 MiB for 16.7M slots: it stays in every XCD's L2).
//  * Uniform sub-expressions never reach the GPU as code: the host evaluates them into a per-instance parameter block
//    that the kernels read with scalar loads.
//  * The reference rebuilds the alive list with 1-3 global atomics per particle (vfx_update.wgsl:148-166). Here the lists
//    change only in frames with casualties, and then in two light passes over the ROWS: k_count_rows reads every row's
//    slot, looks its died bit up (the only gather of the path) and leaves one survivor bit per ROW and one survivor count per
//    4096-row chunk - it does not move a row; k_compact takes the exclusive prefix of the earlier chunks' counts, ranks the
//    rows of its chunk from the row bits with ballot-free popcounts and moves survivors / casualties straight to their final
//    rows in serial (stable) order. One non-returning atomic per workgroup with casualties, none per particle.
//  * No workgroup ever waits for another (no tickets, no look-back spin): nothing here can hang the GPU. HIP has no
//    indirect dispatch: grids are sized on the host for the worst case the host knows (capacity, event-buffer size) and
//    surplus workgroups exit after reading the device-resident counters, so there is no readback on the frame path.
#pragma once
#ifndef __HIPCC_RTC__  // hiprtc (hnb_jit) provides the runtime declarations itself
#include <hip/hip_runtime.h>
#endif
#include "hnb_dev.h"

namespace hnb {

// A pointer that was loaded from a table (slab bases, event buffers) is a generic ("flat") pointer to the compiler: every
// access through it becomes a flat_load / flat_store, which the hardware resolves against the LDS and scratch apertures
// first. The slabs are global memory: saying so through the address space turns them into global_load / global_store
// with an SGPR base and a 32-bit VGPR offset.
template <class T>
__device__ __forceinline__ T* global_ptr(uint64_t addr) {
    typedef __attribute__((address_space(1))) T global_t;
    return (T*)(global_t*)addr;
}

struct u2_t { uint32_t x, y; };
struct u3_t { uint32_t x, y, z; };
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
// Streaming accesses use the default cache policy: `nontemporal` hints measured 13 % SLOWER on
// MI355X for this kernel (profiles/r01b_summary.md). Round 2, with the alternating walk, separately (two A/B rounds of bench.py on one
// box): nontemporal LOADS only 0.203 vs 0.159 ms on C2 and 0.292 vs 0.223 on C4 (they forgo the Infinity-Cache hits the walk is built
// for), nontemporal STORES only: no difference (0.159 / 0.160 vs 0.159 / 0.161).
#if defined(HNB_NONTEMPORAL) || defined(HNB_NONTEMPORAL_LOADS)
#define HNB_NT_LOAD(p) __builtin_nontemporal_load(p)
#else
#define HNB_NT_LOAD(p) (*(p))
#endif
#if defined(HNB_NONTEMPORAL) || defined(HNB_NONTEMPORAL_STORES)
#define HNB_NT_STORE(v, p) __builtin_nontemporal_store(v, p)
#else
#define HNB_NT_STORE(v, p) (*(p) = (v))
#endif

// ---- cache policy of streamed data ----------------------------------------------------------------------------------------------
// In a spawn / die steady state of a LARGE effect the update's time is set by how much of its planes the 256 MiB Infinity Cache still holds
// when the next frame's (reversed) walk starts, and between two update launches the init, k_count_rows and k_compact move a quarter of a
// gigabyte of their own. With `stream_hint` (decided per program and frame on the host: plan::use_streaming_hints - the frame touches more
// than the cache holds) the list accesses of those kernels and the update's loads of its READ-ONLY planes (LIFETIME, alive bytes) carry the
// nontemporal hint. Same-box A/B, two rounds each (profiles/r04f_ab_lnt.log, r04g_ab_nt2.log, r04i_ab_nt3.log): c2_mixed 0.382 -> 0.352 ms,
// c2_dieoff 0.270 -> 0.252, c2_events 0.461 -> 0.430; C3 / c2_interop unchanged. A SMALL effect is served by the caches the hint gives up:
// C5 (4.19M particles) 0.0402 -> 0.0451 ms with the hint, hence the size criterion. The hint on the init's scattered plane stores measured
// nothing and is not made.
template <class T> __device__ __forceinline__ T ld_hint(const T* p, bool nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
template <class T> __device__ __forceinline__ void st_hint(T v, T* p, bool nt) { if (nt) __builtin_nontemporal_store(v, p); else *p = v; }

// A workgroup barrier that orders LDS traffic only: __syncthreads() is also a release fence for the wave's global stores (s_waitcnt vmcnt(0): the wave sits
// until every plane store it issued is acknowledged), which the epilogues that hand four words per wave to thread 0 through LDS do not need (k_init's
// horizon merge has used the same since round 3).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- wave reductions over DPP (r6) ------------------------------------------------------------------------------------------------------------------
// The epilogues below reduce three to six per-lane values over the wave. As butterflies of __shfl_xor they were chains of dependent ds_bpermute
// round trips through the LDS crossbar - 6 per value, 18 to 36 in a row per wave - and in the kernels of a small frame, where every workgroup of the
// launch is resident at once and the kernel lasts as long as its slowest wave, 2.7 of the age kernel's 10.5 us (profiles/r06ab_age_kernel_cuts.log).
// Here: two quad permutes, two row rotations, two row broadcasts - VALU operands, no LDS - and the total read from lane 63 into a scalar register
// (wave-uniform). Integer sums and min / max of non-NaN floats do not depend on the order. Every lane of the wave must be active.
#define HNB_DPP_(v, ident, ctrl, rows) ((uint32_t)__builtin_amdgcn_update_dpp((int)(ident), (int)(v), ctrl, rows, 0xf, false))
// (the moved value is taken ONCE per step, with every lane active: an operator that names it twice - a conditional - would otherwise repeat the DPP move
// under its own branch, where the lanes that lost the comparison are switched off and cannot be read)
#define HNB_WAVE_STEP_(v, ident, OP, ctrl, rows) { const uint32_t t_ = HNB_DPP_(v, ident, ctrl, rows); v = OP(v, t_); }
#define HNB_WAVE_REDUCE_(v, ident, OP)                                                          \
    HNB_WAVE_STEP_(v, ident, OP, 0xB1, 0xf)  /* quad_perm [1,0,3,2] */                          \
    HNB_WAVE_STEP_(v, ident, OP, 0x4E, 0xf)  /* quad_perm [2,3,0,1] */                          \
    HNB_WAVE_STEP_(v, ident, OP, 0x124, 0xf) /* row_ror:4 */                                    \
    HNB_WAVE_STEP_(v, ident, OP, 0x128, 0xf) /* row_ror:8: every lane holds its row's total */  \
    HNB_WAVE_STEP_(v, ident, OP, 0x142, 0xa) /* row_bcast:15 into rows 1 and 3 */               \
    HNB_WAVE_STEP_(v, ident, OP, 0x143, 0xc) /* row_bcast:31 into rows 2 and 3: lane 63 holds the wave's */
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#define HNB_OP_(a, b) ((a) + (b))
    HNB_WAVE_REDUCE_(v, 0u, HNB_OP_)
#undef HNB_OP_
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#define HNB_OP_(a, b) ((b) < (a) ? (b) : (a))
    HNB_WAVE_REDUCE_(v, 0xffffffffu, HNB_OP_)
#undef HNB_OP_
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#define HNB_OP_(a, b) ((b) > (a) ? (b) : (a))
    HNB_WAVE_REDUCE_(v, 0u, HNB_OP_)
#undef HNB_OP_
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ float wave_min_f32(float x) {   // (no NaN among the operands: fminf's choice would depend on the order)
    uint32_t v = f2u(x);
#define HNB_OP_(a, b) f2u(fminf(u2f(a), u2f(b)))
    HNB_WAVE_REDUCE_(v, 0x7f800000u, HNB_OP_)
#undef HNB_OP_
    return u2f((uint32_t)__builtin_amdgcn_readlane((int)v, 63));
}


// ---- reset: dead_index[i] = i (effect_cache.rs:298-323) ------------------------------------
#ifndef HNB_JIT_TU
__global__ void k_reset_lists(uint32_t* __restrict__ dead, uint32_t* __restrict__ alive0, uint32_t* __restrict__ alive1, uint32_t capacity) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < capacity) { dead[i] = i; alive0[i] = 0u; alive1[i] = 0u; }
}
#endif

// ---- placement probe ------------------------------------------------------------------------------
// Streams through a block the way the slot-major update does (several planes read and written with 16-byte
// accesses, far apart from each other). Used once per large block at allocation: see alloc_slab_block().
#ifndef HNB_JIT_TU
struct ProbeArgs { uint32_t n_planes; uint32_t stride16[8]; uint64_t off[8]; uint32_t write_mask; uint32_t salt; uint64_t n_quads; };
__global__ void __launch_bounds__(256) k_probe_placement(char* __restrict__ base, const ProbeArgs a) {
    const uint64_t q = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (q >= a.n_quads) return;
    uint32_t acc = 0;
#pragma unroll
    for (uint32_t pl = 0; pl < 8; ++pl) {
        if (pl >= a.n_planes) break;
        uint4* ptr = reinterpret_cast<uint4*>(base + a.off[pl]) + q * a.stride16[pl];
        for (uint32_t i = 0; i < a.stride16[pl]; ++i) {
            uint4 v = ptr[i];
            acc += v.x + v.y + v.z + v.w;
            v.x ^= a.salt;  // salt is 0 at run time: the block keeps its content, the compiler keeps the store
            if (a.write_mask >> pl & 1u) ptr[i] = v;
        }
    }
    if (acc == 0x9e3779b9u && a.salt) *reinterpret_cast<uint32_t*>(base) = acc;  // keeps the read-only planes' loads
}
#endif

// ---- V-file attribute access (generic kernels, one particle per lane) -----------------------
template <class FILE_T>
__device__ __forceinline__ void vfile_store_attr(const FILE_T& r, uint32_t ncomp, uint32_t reg, char* plane, uint32_t slot) {
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

// Particles the init pass of instance k wants to spawn this frame: the CPU spawner's count, or — for an
// effect with a parent — the number of spawn events its parent appended during the PREVIOUS frame
// (vfx_init.wgsl:123-129; events past the buffer capacity were never stored).
// The compiler sinks a scalar load to its first use - behind the branches in front of it, where it becomes one more DEPENDENT round trip. An empty asm
// that names the value pins the load (and the wait for it, shared with everything requested beside it) where the source put it. Uniform values only.
__device__ __forceinline__ void pin_scalar(uint32_t v) { asm volatile("" ::"s"(v)); }
__device__ __forceinline__ uint32_t requested_spawn(const DevFrameInst& f) {
    // (r6: the row's first 32 bytes - spawn_count .. skip - in one scalar load: hnb_dev.h)
    const uint4 h0 = reinterpret_cast<const uint4*>(&f)[0], h1 = reinterpret_cast<const uint4*>(&f)[1];
    const uint32_t spawn_count = h0.x, parity = h1.z, skip = h1.w;
    const uint64_t ev_in = (uint64_t)h1.x | ((uint64_t)h1.y << 32);
    pin_scalar(spawn_count);
    if (skip) return 0u;
    if (ev_in == 0ull) return spawn_count;
    const DevEventBuffer* ev = global_ptr<const DevEventBuffer>(ev_in);
    const uint2 cnt = *reinterpret_cast<const uint2*>(ev->count);
    const uint32_t n = (parity ^ 1u) ? cnt.y : cnt.x, cap = ev->capacity;
    return n < cap ? n : cap;
}

// ---- code policies -----------------------------------------------------------------------------
// How the generic kernels run a program. InterpCode interprets the bytecode (always available, any
// program). The kernels specialised at program creation (hnb_jit.h) supply a policy with the same
// interface whose bodies are straight-line code generated from the same bytecode: every decode,
// switch and register index folds at compile time.
struct InterpCode {
    using file_t = vreg_file_t;
    template <class ST>
    static __device__ __forceinline__ void run_init(const DevProgram& p, ST& S, const VmUniforms& U, const VmAttrIO& io) {
        vm_run<true, false>(p.init_code, p.init_len, S, U, nullptr, nullptr, io);
    }
    template <class ST>
    static __device__ __forceinline__ void run_update(const DevProgram& p, ST& S, const VmUniforms& U, const VmAttrIO& io) {
        vm_run<true, false>(p.update_code, p.update_len, S, U, nullptr, nullptr, io);
    }
    // var particle = Particle(): attributes the INIT program never assigns are stored as zero
    static __device__ __forceinline__ void zero_unassigned(const DevProgram& p, const VmAttrIO& io) {
        for (uint32_t a = 0; a < p.n_attrs; ++a) {
            if (p.attrs[a].reg != HNB_REG_NONE) continue;
            uint32_t* q = vm_attr_ptr(io, a);
            for (uint32_t c = 0; c < p.attrs[a].ncomp; ++c) q[c] = 0u;
        }
    }
    template <class ST>
    static __device__ __forceinline__ void store_init(const DevProgram& p, const ST& S, char* base, uint32_t slot) {
        for (uint32_t a = 0; a < p.n_attrs; ++a)
            if (p.attrs[a].reg != HNB_REG_NONE) vfile_store_attr(S.r, p.attrs[a].ncomp, p.attrs[a].reg, base + p.attrs[a].plane_off, slot);
    }
    template <class ST>
    static __device__ __forceinline__ void load_update(const DevProgram& p, ST& S, const char* base, uint32_t slot, bool valid) {
        for (uint32_t a = 0; a < p.n_attrs; ++a) {
            const DevAttr at = p.attrs[a];
            if (!(at.upd_flags & HNB_ATTR_UPD_LOAD) || at.reg == HNB_REG_NONE) continue;
            Out4 o = Out4{0u, 0u, 0u, 0u};
            if (valid) o = vfile_load_attr(at.ncomp, base + at.plane_off, slot);
            for (uint32_t cc = 0; cc < at.ncomp; ++cc) S.r[at.reg + cc] = out4_get(o, cc);  // single indexed store site
        }
    }
    template <class ST>
    static __device__ __forceinline__ void store_update(const DevProgram& p, const ST& S, char* base, uint32_t slot) {
        for (uint32_t a = 0; a < p.n_attrs; ++a) {
            const DevAttr at = p.attrs[a];
            if ((at.upd_flags & HNB_ATTR_UPD_STORE) && at.reg != HNB_REG_NONE) vfile_store_attr(S.r, at.ncomp, at.reg, base + at.plane_off, slot);
        }
    }
};
// The same interpreter over the wide V file (programs above HNB_VM_MAX_REGS registers). 128 dynamically
// indexed registers do not stay in VGPRs: this is the slow-but-correct path for HNB_JIT=0; the specialised
// kernels of such programs index the file with constants and pay nothing.
struct InterpCodeWide : InterpCode {
    using file_t = vreg_file_wide_t;
};

// ---- death horizons ----------------------------------------------------------------------------------------------------------
// The one gather of the list path - k_count_rows looking up the died bit of every row's slot - is spent mostly on rows that cannot have
// died: the alive list is in BIRTH order (stable compaction, spawns appended), and in an effect whose particles only die of old age
// (streamable update, one AGE_TICK up front, no kill modifier: DevProgram::horizon) a row chunk of young particles has no casualty for most
// of its life. Per instance the slab keeps a CLOCK, the sum of the ticks so far (binary64, advanced once per frame by the update kernel by
// max(tick, 0) (1 + 2^-16)), and per 4096-ROW chunk a lower bound D of the clock value at which one of its rows can die first, plus the
// frame BF its oldest row was born in:
//   k_init       a spawn with 0 <= age0 <= 0.74 lifetime gets D = clock + (lifetime - age0)(1 - 2^-10), else D = clock (no claim); the
//                workgroup takes the minimum per row chunk of the rows it appends (of the f32 term, in LDS: the clock is common) and merges
//                it with an atomic min (u64: non-negative doubles order like their bits), BF likewise;
//   k_count_rows a chunk with clock < D and frame - BF <= 4096 is not gathered: every row survives (mask all ones, count = rows);
//   k_compact    rows move to lower rows: a source chunk's (D, BF) is merged into the one or two target chunks its survivors land in.
// Why this is safe: f32 ages are accumulated with one rounding of 2^-24 relative per frame, so after N <= 4097 frames
// age <= (age0 + sum of ticks)(1 + 2.5e-4); sum of ticks <= clock now - clock at birth < (lifetime - age0)(1 - 2^-10) gives
// age < lifetime - 2^-10 (lifetime - age0) + 2.5e-4 lifetime <= lifetime for age0 <= 0.74 lifetime: the program's own `age < lifetime` holds,
// the particle is alive after this frame. Non-finite ticks switch the use off for the frame (CompactArgs::hz_use); host writes reset the
// arrays; a violated claim would make k_compact's survivor count disagree with the update's casualty count: HnbEffectMetadata::fault.
struct HorizonView {   // (no pointer ARRAYS indexed by the parity: the compiler keeps such a struct in LDS - 40 B x 256 threads - and the
                       // launch of a small kernel with 10 KiB of LDS per workgroup took 25 us longer, profiles/r03j_kernels.log)
    double* clock;
    unsigned long long* d0;
    uint32_t* bf0;
    uint32_t chunks;
    __device__ __forceinline__ unsigned long long* D(uint32_t parity) const { return d0 + (size_t)parity * chunks; }
    __device__ __forceinline__ uint32_t* BF(uint32_t parity) const { return bf0 + (size_t)parity * chunks; }
};
__device__ __forceinline__ HorizonView horizon_view(char* base, soff_t off, uint32_t chunks) {
    HorizonView h;
    char* p = base + off;
    h.clock = reinterpret_cast<double*>(p);
    h.d0 = reinterpret_cast<unsigned long long*>(p + 256);
    h.bf0 = reinterpret_cast<uint32_t*>(h.d0 + 2 * (size_t)chunks);
    h.chunks = chunks;
    return h;
}
constexpr unsigned long long kHorizonNever = 0x7f7f7f7f7f7f7f7full;   // 1.4e306, above every clock: no row, nobody can die (one repeated byte: the host sets it with a memset)
constexpr uint32_t kHorizonFrames = 4096u;

// ---- init -----------------------------------------------------------------------------------
// One thread per spawned particle. Thread i of instance k (serial order == thread order):
//   slot = dead[alive0 + i]; seed = pcg_hash(slot ^ spawner.seed); run INIT; alive[w][alive0+i] = slot.
// vfx_init.wgsl:141-143 uses atomicAdd(alive_count): under serial execution thread i gets
// alive0 + i, which is what is computed here without atomics. Counters are advanced by k_compact.
template <class CODE>
__device__ __forceinline__ void init_workgroup(const DevProgram& prog, const uint64_t* __restrict__ inst_base, const DevMeta* __restrict__ meta_in,
                                               const DevFrameInst* __restrict__ fi, const uint32_t* __restrict__ ublocks,
                                               const uint32_t blk, const uint32_t grid) {   // workgroup blk of the program's `grid` (k_init, k_init_jobs)
    // Find the instance owning this workgroup: binary search over the CPU prefix sum of init workgroups
    // (find_location_from_particle, vfx_init.wgsl:51-72, at workgroup granularity). The prefix sums sit in a
    // packed array behind the parameter blocks, so the first steps of the search hit the same cached words in
    // every workgroup (searching the 128-byte DevFrameInst rows cost ~10 dependent cache misses per workgroup).
    // (r6: a program with ONE instance - every effect of a scene, C5 - owns the whole grid: no search, no row to look the first workgroup up in. The search
    // and the look-up were two dependent round trips in front of everything else of a kernel that lasts five)
    uint32_t k = 0u, first_block = 0u, n_blocks = grid;
    if (prog.n_inst != 1u) {
        const uint32_t* init_start = ublocks + (size_t)prog.n_inst * prog.n_uregs;
        uint32_t lo = 0, hi = prog.n_inst;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (blk >= init_start[mid]) lo = mid + 1; else hi = mid;
        }
        k = lo - 1;
        // The instance's workgroups stride over its spawns: exactly one round for CPU spawners; for effects with a
        // parent the event count is only known on the device, so the grid is capped and loops (no indirect dispatch in HIP).
        first_block = fi[k].init_block_start;
        n_blocks = (k + 1u < prog.n_inst ? fi[k + 1u].init_block_start : grid) - first_block;
    }
    const uint64_t slab = inst_base[k];   // (requested with the counters and the frame's row, in front of requested_spawn's tests)
    const uint32_t alive0 = meta_in[k].alive_count;
    const uint32_t max_spawn = prog.capacity - alive0;
    const uint32_t spawn = requested_spawn(fi[k]);
    const uint32_t n_spawn = spawn < max_spawn ? spawn : max_spawn;

    char* base = global_ptr<char>(slab);
    const uint32_t* dead = reinterpret_cast<const uint32_t*>(base + prog.dead_off);
    uint32_t* alive = reinterpret_cast<uint32_t*>(base + prog.alive_off[list_column(meta_in[k].write_index)]);  // the column holding the list
    // ... as a ring ("Ring lists"): row r lives at (head + r) % capacity. Appended spawns are rows alive0 + i; in a RING frame they are rows -n_spawn + i:
    // in front of everything, where (RIBBON_ID, AGE) order wants them, and k_compact only moves the head
    const uint32_t list_first = ring_row(list_head(meta_in[k].write_index), prog.ring ? prog.capacity - n_spawn : alive0, prog.capacity);
    VmUniforms U;
    U.u = ublocks + (size_t)k * prog.n_uregs;
    U.xf = fi[k].xf;

    const HorizonView hz = horizon_view(base, prog.horizon_off, prog.chunks_per_inst);
    const double clock_now = prog.horizon ? *hz.clock : 0.0;
    __shared__ uint32_t s_hz[2][2];
    uint32_t pass_no = 0;
    if (prog.horizon) { if (threadIdx.x < 2u) s_hz[0][threadIdx.x] = 0xffffffffu; __syncthreads(); }
    // A workgroup takes `rounds` consecutive groups of 256 spawns per pass (as many as the grid the host chose leaves to each workgroup, at
    // most 4: hnb_simulate launches a quarter of the workgroups when the frame spawns a million particles or more - the horizon merge
    // below costs two global atomics per PASS, and a short init program is bound by its per-workgroup preamble) and strides over the rest. Trip counts are uniform over the
    // workgroup (the merge has a barrier).
    const uint32_t per_round = n_blocks * kInitBlock;
    uint32_t rounds = 1u;                                               // ceil(n_spawn / per_round), at most kInitRounds (no division: this is per workgroup)
#pragma unroll
    for (uint32_t r = 1; r < kInitRounds; ++r) rounds += (uint64_t)n_spawn > (uint64_t)r * per_round ? 1u : 0u;
    const uint32_t pass_rows = rounds * kInitBlock;
    const uint32_t pass_stride = n_blocks * pass_rows;                  // (< n_spawn + per_round: no overflow; the loop ends before p0 could wrap)
    for (uint32_t p0 = (blk - first_block) * pass_rows; p0 < n_spawn; p0 = n_spawn - p0 <= pass_stride ? n_spawn : p0 + pass_stride) {
    uint32_t* sw = s_hz[pass_no & 1u];
    const uint32_t rc0 = (alive0 + p0) / kChunk;                        // (workgroup-uniform: the chunk of the pass's first row; its <= 1024 rows lie in it or the next)
    if (prog.horizon && threadIdx.x < 2u) s_hz[(pass_no & 1u) ^ 1u][threadIdx.x] = 0xffffffffu;   // see below
    for (uint32_t i0 = p0; i0 < p0 + pass_rows && i0 < n_spawn; i0 += kInitBlock) {
        const uint32_t i = i0 + threadIdx.x;
        uint32_t r_bits = 0xffffffffu;   // (an idle lane)
        if (i < n_spawn) {
        const uint32_t slot = ld_hint(dead + (alive0 + i), prog.stream_hint != 0u);
        VmState<typename CODE::file_t> S;
        S.r = typename CODE::file_t{};  // var particle = Particle();  (vfx_init.wgsl:174)
        S.pindex = slot + fi[k].slot_base;
        S.seed = pcg_hash(S.pindex ^ fi[k].seed);
        S.pcounter = meta_in[k].particle_counter + i;
        S.alive = true;

        VmAttrIO io;
        io.slab = base; io.attrs = prog.attrs; io.slot = slot;
        if (fi[k].ev_in != 0ull) {  // GPU-spawned: fetch the parent particle that emitted event i (vfx_init.wgsl:166-171)
            S.gpu_spawned = true;
            io.parent_slab = global_ptr<const char>(fi[k].parent_base);
            io.parent_planes = global_ptr<const uint32_t>(fi[k].parent_planes);
            io.parent_slot = global_ptr<const DevEventBuffer>(fi[k].ev_in)->data[i];
        }
        CODE::zero_unassigned(prog, io);
        CODE::run_init(prog, S, U, io);
        st_hint(slot, alive + ring_row(list_first, i, prog.capacity), prog.stream_hint != 0u);
        uint8_t alive_byte = 1u;  // the update walks the slots through these bytes
        if (prog.age_cohort) {  // a chunk that keeps its particles' common age in one word: this slot's age is in the plane (state 2, byte 3)
            uint32_t* astate = reinterpret_cast<uint32_t*>(base + prog.lmin_off) + 2u * prog.chunks_per_inst;
            const uint32_t st = astate[slot / kChunk];
            if (st == 1u || st == 2u) { if (st != 2u) astate[slot / kChunk] = 2u; alive_byte = 3u; }   // (0, 4: the plane holds every age of the chunk)
        }
        reinterpret_cast<uint8_t*>(base + prog.alive_flag_off)[slot] = alive_byte;
        if (prog.cull_lifetime) reinterpret_cast<float*>(base + prog.lmin_off)[slot / kChunk] = 0.0f;  // the chunk's lifetime bound is unknown again
        // (Round 5, built twice and removed: a "spawn nursery" - scattered spawns hand POSITION / VELOCITY / AGE to the same frame's update as ONE
        // 32-byte record instead of three partial-sector plane stores, the update substitutes them before its first tick. Bit-equal to this on the
        // whole GPU suite in both forms - records in a bucket per 256-slot group (a returning atomic per spawn, an LDS map + ds_bpermute routing in
        // the update), and records at nursery[slot] announced by the alive byte (no atomics) - and a loss in both: c2_mixed init 0.049 -> 0.039 /
        // 0.045 ms, update 0.205 -> 0.232 / 0.252 ms, frame 0.317 -> 0.333 / 0.363 ms on one box; c2_interop's update 0.1625 -> 0.1748 with the
        // substitution compiled in and never used (registers: 5 waves instead of 6). The init under churn is not bound by its write amplification:
        // 35 % fewer store sectors bought 9 %. profiles/r05_nursery/.)
        CODE::store_init(prog, S, base, slot);
        if (prog.horizon) {   // how much clock can pass before this particle may die? (see "death horizons"; f32, rounded towards less)
            const float age0 = u2f(S.r[HNB_REG_AGE]), life = u2f(S.r[HNB_REG_LIFETIME]);
            float r0 = 0.0f;
            if (age0 >= 0.0f && life > 0.0f && age0 <= 0.74f * life) r0 = (life - age0) * 0.998046875f;   // (1 - 2^-9): <= (life - age0)(1 - 2^-10) after both roundings
            r_bits = f2u(r0);
        }
        }
        // the lanes take the minimum of their f32 terms per row chunk in LDS (ds_min_u32: the LDS pipe is idle in this kernel, the VALU is what
        // bounds a burst; non-negative floats order like their bits; the clock is common to the launch: min D = clock + min r0)
        if (prog.horizon && r_bits != 0xffffffffu) atomicMin(&sw[(alive0 + i) / kChunk != rc0 ? 1 : 0], r_bits);
    }
    if (prog.horizon) {
        // ... and threads 0 and 1 merge the pass's two results into the chunks' words with global atomics. What bounds this is the number of
        // global atomics (they execute beyond the XCD's L2): one pair per WAVE (no barrier at all) cost a 16.7M burst 0.07 ms, one pair per 256
        // spawns 0.02-0.03 ms (profiles/r03l_ab.log, r03n_ab.log). Fire and forget: reading the words first, to skip atomics that cannot
        // lower them, made the workgroup wait for the loads - 3x the time of a burst (profiles/r03m_ab.log).
        // s_hz[pass & 1] is this pass's pair; threads 0 / 1 reset the other pair at the start of the pass (they read it themselves, after the
        // previous pass's barrier; the other waves touch it only after this one): one barrier per pass, and one that waits for the wave's LDS
        // operations only (__syncthreads() is also a release fence for the particle stores above - s_waitcnt vmcnt(0): the wave would sit
        // until its ~20 scattered stores per round are acknowledged).
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (threadIdx.x < 2u) {
            const uint32_t m = sw[threadIdx.x], rc = rc0 + threadIdx.x;
            if (m != 0xffffffffu && rc < prog.chunks_per_inst) {
                atomicMin(&hz.D(prog.hz_parity)[rc], d2u(clock_now + (double)u2f(m)));
                atomicMin(&hz.BF(prog.hz_parity)[rc], prog.frame_no);
            }
        }
        pass_no += 1u;
    }
    }
}
template <class CODE>
__global__ void __launch_bounds__(kInitBlock)
k_init(const DevProgram prog, const uint64_t* __restrict__ inst_base, const DevMeta* __restrict__ meta_in,
       const DevFrameInst* __restrict__ fi, const uint32_t* __restrict__ ublocks) {
    init_workgroup<CODE>(prog, inst_base, meta_in, fi, ublocks, blockIdx.x, gridDim.x);
}
// ---- slot-major init: large spawns ---------------------------------------------------------------------------------------------------------
// k_init is ROW-major: thread i takes slot dead[alive0 + i]. From a fresh slab that is slot i and the plane stores of a wave are contiguous; after
// a die-off the dead stack holds the slots in the order they were killed - per frame of the die-off a descending run with a stride of ~25 slots -
// and every lane of a wave writes a different line of every plane: a 16.7M re-burst took 2.06 ms where the first burst took 0.20
// (profiles/r05_nursery/r05b_reburst.log; SpawnerSettings::burst(count, period), src/spawn.rs:472, pop order vfx_init.wgsl:141-143).
// But WHAT a spawn writes into its slot does not depend on its rank: the PRNG is seeded by the slot (vfx_init.wgsl:145-146), the spawner's inputs
// are uniform - unless the init program reads PARTICLE_COUNTER (HNB_OP_LDPC) or a parent particle. Only the alive LIST depends on the rank, and
// that is a copy of the popped segment of the dead stack, dead[alive0 .. alive0 + n) -> rows alive0 .. alive0 + n. So for a frame that spawns a
// large share of the capacity (host: HnbProgram plan, >= 1/8 of the program's slots; any share is CORRECT) the init walks the SLOTS like the
// update does, a workgroup per quarter chunk of 1024 slots, lane l of a step owning slot 256 s + l: every plane store of a wave is contiguous
// whatever order the dead stack is in. Which slots spawn:
//   * the spawn fills every free slot (n_spawn == capacity - alive_count: every burst of `capacity` particles): the slots whose alive byte is 0;
//   * otherwise k_spawn_mark first writes alive byte 2 into the popped slots (one scattered byte per spawn instead of six scattered plane
//     stores), launched whenever the host cannot prove the first case (spawn request < capacity); it decides per instance from the device counters.
// The same workgroup appends rows [1024 w, 1024 w + 1024) of the spawn to the list (coalesced copy). Serial-order semantics are untouched: the
// state after the frame is the row-major kernel's, bit for bit (tests: re-bursts after die-offs, partial re-fills, several instances).
// Death horizons: a slot-major workgroup does not know the lifetimes of the ROWS it appends: it makes no claim for their row chunks (D = the
// clock: "may die now"; a burst's row chunks all hold a particle near the minimum lifetime anyway).
constexpr uint32_t kSlotInitWg = 1024u;   // slots (and spawn rows) per workgroup
struct SlotInitCtx {
    uint32_t k, w;            // instance, workgroup within the instance
    uint32_t alive0, n_spawn; // rows the list starts the frame with, spawns of the frame (capped)
    bool full;                // the spawn takes every free slot
    char* base;
};
__device__ __forceinline__ void slot_init_setup(SlotInitCtx& c, const DevProgram& prog, const uint64_t* __restrict__ inst_base, const DevMeta* __restrict__ meta_in,
                                                const DevFrameInst* __restrict__ fi, const uint32_t wg) {
    const uint32_t per_inst = prog.chunks_per_inst * (kChunk / kSlotInitWg);
    c.k = wg / per_inst;
    c.w = wg - c.k * per_inst;
    const uint64_t slab = inst_base[c.k];
    c.alive0 = meta_in[c.k].alive_count;
    const uint32_t max_spawn = prog.capacity - c.alive0;
    const uint32_t spawn = requested_spawn(fi[c.k]);
    c.n_spawn = spawn < max_spawn ? spawn : max_spawn;
    c.full = c.n_spawn == max_spawn;
    c.base = global_ptr<char>(slab);
}
#ifndef HNB_JIT_TU
// alive byte 2 = "spawns in this frame" for the slots a PARTIAL re-fill pops (k_init_slots turns every one of them into 1 / 3 in the same frame)
__global__ void __launch_bounds__(kBlock)
k_spawn_mark(const DevProgram prog, const uint64_t* __restrict__ inst_base, const DevMeta* __restrict__ meta_in, const DevFrameInst* __restrict__ fi) {
    SlotInitCtx c;
    slot_init_setup(c, prog, inst_base, meta_in, fi, blockIdx.x);
    if (c.full || c.w * kSlotInitWg >= c.n_spawn) return;
    const uint32_t* dead = reinterpret_cast<const uint32_t*>(c.base + prog.dead_off);
    uint8_t* flags = reinterpret_cast<uint8_t*>(c.base + prog.alive_flag_off);
    uint32_t slot[kSlotInitWg / kBlock];
#pragma unroll
    for (uint32_t q = 0; q < kSlotInitWg / kBlock; ++q) {
        const uint32_t r = c.w * kSlotInitWg + q * kBlock + threadIdx.x;
        slot[q] = r < c.n_spawn ? dead[c.alive0 + r] : 0xffffffffu;
    }
#pragma unroll
    for (uint32_t q = 0; q < kSlotInitWg / kBlock; ++q)
        if (slot[q] != 0xffffffffu) flags[slot[q]] = 2u;
}
#endif
template <class CODE>
__device__ __forceinline__ void init_slots_workgroup(const DevProgram& prog, const uint64_t* __restrict__ inst_base, const DevMeta* __restrict__ meta_in,
                                                     const DevFrameInst* __restrict__ fi, const uint32_t* __restrict__ ublocks, const uint32_t wg) {
    SlotInitCtx c;
    slot_init_setup(c, prog, inst_base, meta_in, fi, wg);
    if (c.n_spawn == 0u) return;
    const uint32_t tid = threadIdx.x, k = c.k;
    char* base = c.base;
    constexpr uint32_t kSubs = kSlotInitWg / kBlock;
    // (1) the list: rows alive0 + r <- dead[alive0 + r] for this workgroup's share of the spawn, requested first, stored last
    const uint32_t* dead = reinterpret_cast<const uint32_t*>(base + prog.dead_off);
    uint32_t* alive = reinterpret_cast<uint32_t*>(base + prog.alive_off[list_column(meta_in[k].write_index)]);
    const uint32_t list_first = ring_row(list_head(meta_in[k].write_index), c.alive0, prog.capacity);   // (programs that keep a ring never take this path: head 0)
    const bool nt = prog.stream_hint != 0u;
    uint32_t row_slot[kSubs];
#pragma unroll
    for (uint32_t q = 0; q < kSubs; ++q) {
        const uint32_t r = c.w * kSlotInitWg + q * kBlock + tid;
        row_slot[q] = r < c.n_spawn ? ld_hint(dead + (c.alive0 + r), nt) : 0xffffffffu;
    }
    // (2) the slots of this quarter chunk
    uint8_t* flags = reinterpret_cast<uint8_t*>(base + prog.alive_flag_off);
    const uint32_t slot0 = c.w * kSlotInitWg, j = slot0 / kChunk;
    uint32_t byte[kSubs];
#pragma unroll
    for (uint32_t q = 0; q < kSubs; ++q) {
        const uint32_t slot = slot0 + q * kBlock + tid;
        byte[q] = slot < prog.capacity ? (uint32_t)flags[slot] : 1u;
    }
    uint8_t alive_byte = 1u;
    uint32_t* astate = reinterpret_cast<uint32_t*>(base + prog.lmin_off) + 2u * prog.chunks_per_inst;
    uint32_t st = 0u;
    if (prog.age_cohort) {   // as in k_init: a chunk that keeps its particles' common age in one word - a spawn's age is in the plane (state 2, byte 3)
        st = slot0 < prog.capacity ? astate[j] : 0u;
        if (st == 1u || st == 2u) alive_byte = 3u;
    }
    VmUniforms U;
    U.u = ublocks + (size_t)k * prog.n_uregs;
    U.xf = fi[k].xf;
    const uint32_t want = c.full ? 0u : 2u;
    bool any_spawn = false;
#pragma unroll 1
    for (uint32_t q = 0; q < kSubs; ++q) {
        const uint32_t b = q == 0u ? byte[0] : q == 1u ? byte[1] : q == 2u ? byte[2] : byte[3];   // (selects: no dynamically indexed array)
        const bool here = b == want;
        if (!__any(here)) continue;
        any_spawn = true;
        if (here) {
            const uint32_t slot = slot0 + q * kBlock + tid;
            VmState<typename CODE::file_t> S;
            S.r = typename CODE::file_t{};  // var particle = Particle();  (vfx_init.wgsl:174)
            S.pindex = slot + fi[k].slot_base;
            S.seed = pcg_hash(S.pindex ^ fi[k].seed);
            S.pcounter = 0u;                // (programs that read PARTICLE_COUNTER are not eligible: the rank is not known here)
            S.alive = true;
            VmAttrIO io;
            io.slab = base; io.attrs = prog.attrs; io.slot = slot;
            CODE::zero_unassigned(prog, io);
            CODE::run_init(prog, S, U, io);
            flags[slot] = alive_byte;
            CODE::store_init(prog, S, base, slot);
        }
    }
    static_assert(kSubs == 4u, "the byte select above is written for four steps");
    if (any_spawn && (tid & 63u) == 0u) {   // (same values from every wave and every sibling workgroup of the chunk)
        if (prog.cull_lifetime) reinterpret_cast<float*>(base + prog.lmin_off)[j] = 0.0f;   // the chunk's lifetime bound is unknown again
        if (alive_byte == 3u && st != 2u) astate[j] = 2u;
    }
#pragma unroll
    for (uint32_t q = 0; q < kSubs; ++q) {
        const uint32_t r = c.w * kSlotInitWg + q * kBlock + tid;
        if (row_slot[q] != 0xffffffffu) st_hint(row_slot[q], alive + ring_row(list_first, r, prog.capacity), nt);
    }
    if (prog.horizon && tid < 2u && c.w * kSlotInitWg < c.n_spawn) {   // no claim for the row chunks this workgroup appended to (at most two)
        const HorizonView hz = horizon_view(base, prog.horizon_off, prog.chunks_per_inst);
        const uint32_t r_first = c.alive0 + c.w * kSlotInitWg;
        const uint32_t r_last = c.alive0 + ((c.w + 1u) * kSlotInitWg < c.n_spawn ? (c.w + 1u) * kSlotInitWg : c.n_spawn) - 1u;
        const uint32_t rc = tid == 0u ? r_first / kChunk : r_last / kChunk;
        if ((tid == 0u || rc != r_first / kChunk) && rc < prog.chunks_per_inst) {
            atomicMin(&hz.D(prog.hz_parity)[rc], d2u(*hz.clock));
            atomicMin(&hz.BF(prog.hz_parity)[rc], prog.frame_no);
        }
    }
}
template <class CODE>
__global__ void __launch_bounds__(kBlock)
k_init_slots(const DevProgram prog, const uint64_t* __restrict__ inst_base, const DevMeta* __restrict__ meta_in,
             const DevFrameInst* __restrict__ fi, const uint32_t* __restrict__ ublocks) {
    init_slots_workgroup<CODE>(prog, inst_base, meta_in, fi, ublocks, blockIdx.x);
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
    if constexpr (P == 4) {
        // A quad with free slots: its 48 bytes are read again (they were loaded a moment ago: an L2 / MALL hit), the
        // alive slots' words replaced, and the whole quad stored with 16-byte stores. Partial 12-byte stores leave
        // sectors half-written, which the memory side completes with a read-modify-write (measured in the firework
        // die-off: a frame with free slots took 1.24x a frame with none before, 1.19x now). Nothing else writes a
        // free slot during the update. (Round 2: a build that skipped the re-read - the free slots' words left wrong - was only
        // 9 % faster in those frames, 0.205 vs 0.227 ms: keeping the loaded words in 24 more registers is not worth the occupancy.)
        u4v* dst = reinterpret_cast<u4v*>(plane) + (size_t)(slot[0] >> 2) * 3;
        u4v q0 = dst[0], q1 = dst[1], q2 = dst[2];
        if (valid[0]) { q0.x = f2u(src[0].x); q0.y = f2u(src[0].y); q0.z = f2u(src[0].z); }
        if (valid[1]) { q0.w = f2u(src[1].x); q1.x = f2u(src[1].y); q1.y = f2u(src[1].z); }
        if (valid[2]) { q1.z = f2u(src[2].x); q1.w = f2u(src[2].y); q2.x = f2u(src[2].z); }
        if (valid[3]) { q2.y = f2u(src[3].x); q2.z = f2u(src[3].y); q2.w = f2u(src[3].z); }
        dst[0] = q0; dst[1] = q1; dst[2] = q2;
        return;
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
__device__ __forceinline__ void pin_store1(const float (&src)[P], char* plane, const uint32_t (&slot)[P], const bool (&valid)[P], bool dense, bool nt = false) {
    if constexpr (P == 4) {
        if (dense && nt) {
            __builtin_nontemporal_store((u4v{f2u(src[0]), f2u(src[1]), f2u(src[2]), f2u(src[3])}), reinterpret_cast<u4v*>(plane) + (slot[0] >> 2));
            return;
        }
        if (dense) {
            HNB_NT_STORE((u4v{f2u(src[0]), f2u(src[1]), f2u(src[2]), f2u(src[3])}), reinterpret_cast<u4v*>(plane) + (slot[0] >> 2));
            return;
        }
    }
    if constexpr (P == 4) {  // as in pin_store3: read the quad again, blend, store 16 bytes
        u4v* dst = reinterpret_cast<u4v*>(plane) + (slot[0] >> 2);
        u4v q = *dst;
        if (valid[0]) q.x = f2u(src[0]);
        if (valid[1]) q.y = f2u(src[1]);
        if (valid[2]) q.z = f2u(src[2]);
        if (valid[3]) q.w = f2u(src[3]);
        *dst = q;
        return;
    }
#pragma unroll
    for (int p = 0; p < P; ++p)
        if (valid[p]) reinterpret_cast<float*>(plane)[slot[p]] = src[p];
}

// ---- vec3 planes through a wave-private LDS transpose ---------------------------------------------
// A lane of the per-particle path owns 4 consecutive slots = 48 contiguous bytes of a vec3 plane. Accessing them directly makes every
// 16-byte access of a wave a set of 64 words at a 48-byte lane stride: 24 cache lines touched for 1 KiB of payload, and the update runs at
// 5.0 TB/s where contiguous accesses reach 6 (see the flat path in k_update_slots_stream). Here the wave takes the step's 192 words
// (256 slots x 12 B = 3 KiB) with three contiguous 1 KiB accesses (lane l: words l, 64 + l, 128 + l), parks them in its own 3 KiB of LDS
// and reads its three words 3l .. 3l + 2 back (a 48-byte lane stride is conflict-free on 64 four-byte banks: 16 lanes x 4 dwords cover every
// bank once). The store goes the other way - and since the staging buffer still holds the plane as loaded, a lane writes only the words of
// its ALIVE slots into it: free slots get their own bytes back, every line is stored in full, and the read-blend-store of partially alive
// quads (pin_store3) is not needed. LDS instructions of one wave execute in order: no barrier.
__device__ __forceinline__ void xpose_load3(V3 (&dst)[4], const char* plane, uint32_t first_slot, u4v* lds, uint32_t lane) {
    const u4v* src = reinterpret_cast<const u4v*>(plane) + (size_t)(first_slot >> 2) * 3;
    const u4v a = src[lane], b = src[64u + lane], c = src[128u + lane];
    lds[lane] = a; lds[64u + lane] = b; lds[128u + lane] = c;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const u4v q0 = lds[3u * lane], q1 = lds[3u * lane + 1u], q2 = lds[3u * lane + 2u];
    dst[0] = V3{u2f(q0.x), u2f(q0.y), u2f(q0.z)};
    dst[1] = V3{u2f(q0.w), u2f(q1.x), u2f(q1.y)};
    dst[2] = V3{u2f(q1.z), u2f(q1.w), u2f(q2.x)};
    dst[3] = V3{u2f(q2.y), u2f(q2.z), u2f(q2.w)};
}
// `lds` must still hold what xpose_load3 put there for this plane and step.
__device__ __forceinline__ void xpose_store3(const V3 (&src)[4], char* plane, uint32_t first_slot, u4v* lds, uint32_t lane, const bool (&valid)[4], bool full, bool nt = false) {
    if (full) {
        lds[3u * lane] = u4v{f2u(src[0].x), f2u(src[0].y), f2u(src[0].z), f2u(src[1].x)};
        lds[3u * lane + 1u] = u4v{f2u(src[1].y), f2u(src[1].z), f2u(src[2].x), f2u(src[2].y)};
        lds[3u * lane + 2u] = u4v{f2u(src[2].z), f2u(src[3].x), f2u(src[3].y), f2u(src[3].z)};
    } else {
        uint32_t* w = reinterpret_cast<uint32_t*>(lds) + 12u * lane;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (valid[p]) { w[3 * p] = f2u(src[p].x); w[3 * p + 1] = f2u(src[p].y); w[3 * p + 2] = f2u(src[p].z); }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    u4v* dstp = reinterpret_cast<u4v*>(plane) + (size_t)(first_slot >> 2) * 3;
    const u4v a = lds[lane], b = lds[64u + lane], c = lds[128u + lane];
    if (nt) { __builtin_nontemporal_store(a, dstp + lane); __builtin_nontemporal_store(b, dstp + 64u + lane); __builtin_nontemporal_store(c, dstp + 128u + lane); }
    else { dstp[lane] = a; dstp[64u + lane] = b; dstp[128u + lane] = c; }
}

// ---- update + kill + compaction ----------------------------------------------------------------
// Launches per program per frame, none of which waits for another workgroup (measured alternative: a
// single-pass decoupled look-back with ticketed persistent workgroups was 10-13 % slower on this short kernel
// because of its scheduling tail and spin-waits):
//   k_update_slots_*  the UPDATE program over the SLOTS (see "Slot-major update" below);
//   k_count_rows      only for instances that lost particles: one survivor bit per row of the alive list (from the died bit
//                     of the row's slot) and the survivor count of every 4096-row chunk;
//   k_compact         per chunk: if the instance had no casualty the list is final and the workgroup only
//                     rotates the counters (vfx_indirect.wgsl:57-85); otherwise exclusive prefix of the earlier
//                     chunks' survivor counts, survivors move to the other list column, casualties are pushed on
//                     the dead list in serial order.
constexpr uint32_t kWaveRows = kChunk / (kBlock / 64);  // 1024 slots (or list rows) per wave
constexpr uint32_t kStepRows = 64 * 4;                  // 256 per wave step: 4 per lane
struct ChunkCtx {
    uint32_t k, j;          // instance, chunk within instance
    uint32_t n;             // max_update of the instance
    uint32_t n_spawn;
    uint32_t start;         // first alive-list row of this chunk
    DevMeta m;
    char* base;
};

struct CompactBufs {
    uint32_t* counts;   // [n_inst * chunks_per_inst] survivors per chunk (this frame)
    uint32_t* deaths;   // [2][table_cap] casualties per instance, frame-parity double-buffered
    uint32_t table_cap;
    uint32_t parity;
    uint32_t* ev_totals;  // [n_inst * chunks_per_inst][HNB_MAX_EVENT_CHANNELS] spawn events per chunk (emitting programs)
    uint32_t xcd_remap;   // workgroup -> chunk mapping (chunk_of_workgroup): bit 0 XCD-aware (batches of instances), bit 1 descending order (every other frame)
};

// Workgroup -> chunk. The hardware deals workgroups to the 8 XCDs round-robin (workgroup b runs on XCD b mod 8,
// each XCD with its own L2). In a batch of instances, mapping b straight to chunk b pins chunk j of EVERY instance
// to XCD j mod 8 whenever an instance has a multiple of 8 chunks (measured, churn: 1024 x 65,536 slots 1.84 ms vs
// 1024 x 65,792 slots 0.93 ms); with bit 0 of `mode` (CompactBufs::xcd_remap) every XCD walks its own contiguous
// eighth of the chunks instead (0.99 ms). A single large instance keeps the straight mapping, which measured 2 %
// faster there.
// Bit 1 of `mode`: walk the chunks in DESCENDING order. hnb_simulate sets it in every other frame, because the 256 MiB
// Infinity Cache sits on the memory side and keeps what was written last: a frame that starts where the previous
// one ended finds the most recently written quarter of a 16.7M-particle effect's planes still on the die instead of in
// HBM (workgroups are dispatched in increasing blockIdx order). Measured with the firework update over
// 16,777,216 particles (tools/layout_probe.hip, profiles/r02d_layout_probe.log): 0.204 ms ascending every frame, 0.204 ms
// descending every frame, 0.171 ms alternating. Any bijection is correct: a workgroup handles the chunk it computes here.
// (b, total): the workgroup's index and the number of workgroups of ITS program - the launch's own, or a program's share of a
// launch that serves several programs (k_count_rows_multi / k_compact_multi).
__device__ __forceinline__ uint32_t chunk_of_workgroup(uint32_t mode, uint32_t b, uint32_t total) {
    uint32_t c = b;
    if (mode & 1u) {
        const uint32_t xcd = b & 7u, local = b >> 3;
        const uint32_t q = total >> 3, r = total & 7u;
        c = xcd * q + (xcd < r ? xcd : r) + local;
    }
    return (mode & 2u) ? total - 1u - c : c;
}
__device__ __forceinline__ uint32_t chunk_of_workgroup(uint32_t mode) { return chunk_of_workgroup(mode, blockIdx.x, gridDim.x); }

// The job of this workgroup: the last one whose first workgroup is <= blockIdx.x (they ascend, the first job starts at or below every workgroup
// that looks here). Every lane looks at one job: one round trip per 64 jobs. (Until round 4: a binary search - log2(n) DEPENDENT scalar loads
// from a table the frame's upload left in HBM, 4-5 us in front of each of a small scene's four shared launches.)
template <class JOB>
__device__ __forceinline__ const JOB& job_of_workgroup_t(const JOB* __restrict__ jobs, uint32_t n_jobs) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t idx = 0;
    for (uint32_t b = 0; b < n_jobs; b += 64u) {
        const uint32_t i = b + lane;
        const uint32_t c = (uint32_t)__popcll(__ballot(i < n_jobs && jobs[i].first_wg <= blockIdx.x));   // (wave-uniform)
        idx += c;
        if (c < 64u) break;
    }
    return jobs[__builtin_amdgcn_readfirstlane(idx - 1u)];
}

// DevMeta travels as two 16-byte words (a struct copy through pointers that may alias became a memcpy through a private array in the
// job-table kernels, which the compiler kept in LDS: 3 KiB per workgroup)
__device__ __forceinline__ DevMeta load_meta(const DevMeta* p) {
    const uint4 a = reinterpret_cast<const uint4*>(p)[0], b = reinterpret_cast<const uint4*>(p)[1];
    DevMeta m;
    m.alive_count = a.x; m.particle_counter = a.y; m.write_index = a.z; m.max_update = a.w;
    m.dead_count = b.x; m.spawned = b.y; m.ref_write_index = b.z; m.instance_count = b.w;
    return m;
}
__device__ __forceinline__ void store_meta(DevMeta* p, const DevMeta& m) {
    reinterpret_cast<uint4*>(p)[0] = make_uint4(m.alive_count, m.particle_counter, m.write_index, m.max_update);
    reinterpret_cast<uint4*>(p)[1] = make_uint4(m.dead_count, m.spawned, m.ref_write_index, m.instance_count);
}

// Decode a chunk id; false when the chunk has no rows.
template <class ARGS>
__device__ __forceinline__ bool chunk_setup(ChunkCtx& c, uint32_t chunk, const ARGS& args, const uint64_t* inst_base, const DevMeta* meta_in,
                                            const DevFrameInst* fi) {
    c.k = chunk / args.chunks_per_inst;
    c.j = chunk - c.k * args.chunks_per_inst;
    // (the instance's slab address and counters are requested in front of requested_spawn's tests: one round trip for the whole row)
    const uint64_t slab = inst_base[c.k];
    // vfx_indirect.wgsl:57-85 folded in: max_update = alive_count after init.
    c.m = load_meta(meta_in + c.k);
    const uint32_t frozen = fi[c.k].skip;
    const uint32_t spawn = requested_spawn(fi[c.k]);
    const uint32_t max_spawn = args.capacity - c.m.alive_count;
    c.n_spawn = spawn < max_spawn ? spawn : max_spawn;
    c.n = frozen ? 0u : c.m.alive_count + c.n_spawn;  // a frozen instance has nothing to update
    c.start = c.j * kChunk;
    c.base = global_ptr<char>(slab);
    return c.start < c.n;
}

// ---- k_compact -----------------------------------------------------------------------------------
template <class ARGS>
__device__ __forceinline__ void compact_chunk(const ARGS& args, const uint64_t* inst_base, const DevMeta* meta_in, DevMeta* meta_out,
                                              const DevFrameInst* fi, const CompactBufs& cb, uint32_t wg, uint32_t wg_total) {
    __shared__ uint32_t s_red[kBlock / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t chunk = chunk_of_workgroup(cb.xcd_remap, wg, wg_total);
    ChunkCtx c;
    // (the instance's casualty counter: requested with its row, not behind the `frozen` test)
    const uint32_t total_dead = cb.deaths[(size_t)cb.parity * cb.table_cap + chunk / args.chunks_per_inst];
    const bool has_rows = chunk_setup(c, chunk, args, inst_base, meta_in, fi);
    pin_scalar(total_dead);
    const HorizonView hz = horizon_view(c.base, args.horizon_off, args.chunks_per_inst);
    if (fi[c.k].skip) {  // not simulated this frame: counters carry over unchanged
        DevMeta keep = c.m;
        if (args.force_rewrite && list_head(c.m.write_index) != 0u) {
            // ("Ring lists") ... but a frozen instance whose list stands behind a head is moved to the other column, linear, like everybody else's in this
            // frame: the sort kernels that may follow walk every instance of the program and know linear lists only. Same rows, same order.
            const uint32_t hd = list_head(c.m.write_index), n_rows = c.m.alive_count;
            const uint32_t* from = reinterpret_cast<const uint32_t*>(c.base + (list_column(c.m.write_index) ? args.alive_off[1] : args.alive_off[0]));
            uint32_t* to = reinterpret_cast<uint32_t*>(c.base + (list_column(c.m.write_index) ? args.alive_off[0] : args.alive_off[1]));
            for (uint32_t r = c.j * kChunk + tid; r < n_rows && r < (c.j + 1u) * kChunk; r += kBlock) to[r] = from[ring_row(hd, r, args.capacity)];
            keep.write_index = list_column(c.m.write_index) ^ 1u;
        }
        if (c.j == 0 && tid == 0) { store_meta(meta_out + c.k, keep); cb.deaths[(size_t)(cb.parity ^ 1u) * cb.table_cap + c.k] = 0u; }
        if (args.hz && tid == 0) { hz.D(args.hz_parity ^ 1u)[c.j] = hz.D(args.hz_parity)[c.j]; hz.BF(args.hz_parity ^ 1u)[c.j] = hz.BF(args.hz_parity)[c.j]; }   // the rows stand: so do their horizons
        return;
    }
    if (c.j == 0 && tid == 0) cb.deaths[(size_t)(cb.parity ^ 1u) * cb.table_cap + c.k] = 0u;  // next frame's counter
    const bool last = c.n == 0 ? c.j == 0 : (c.start < c.n && c.start + kChunk >= c.n);
    const bool rotate = args.rotate_front != 0u && c.n_spawn != 0u;   // (uniform per instance; never set together with slot_order)
    const uint32_t head = list_head(c.m.write_index);
    // ---- Ring lists -----------------------------------------------------------------------------------------------------------------------------
    // A single-ribbon trail (C5: ribbon.rs) keeps its list in (RIBBON_ID, AGE) order = youngest first: every frame a few spawns go in FRONT and,
    // where all particles live equally long, the casualties are the LAST rows (CompactArgs::rotate_front / suffix_dead: host-proven). Rotating
    // the spawns in by rewriting the list moved every row of it every frame (4.19M rows = 33.8 MB to place 46 k: 11 us of C5's 37). As a ring the
    // list stays where it is: k_init has written the spawns in front of the head (DevProgram::ring), the head moves back by n_spawn, the count
    // drops by the casualties, and only the casualties' rows are READ (to push their slots on the dead list, and to check the host's proof against
    // their died bits). DevMeta::write_index carries the head in bits 1..31 (HnbDeviceMeta::list_column: readers use
    // column[(head + row) % capacity]); every other path reads rows through ring_row() and writes its result linear (head 0) - a frame that
    // cannot be a ring frame while a head is set is told to rewrite (CompactArgs::force_rewrite), so the sort kernels only ever see head 0.
    if (args.ring && !fi[c.k].skip) {
        if (!has_rows && !(c.n == 0u && c.j == 0u)) return;
        const uint32_t alive0 = c.m.alive_count;                                   // rows the frame started with (logical rows of the OLD head)
        if (total_dead > alive0 && tid == 0u && args.fault) *args.fault = 1u;
        const uint32_t dead_n = total_dead < alive0 ? total_dead : alive0;
        const uint32_t first_dead = alive0 - dead_n;                                // old rows [first_dead, alive0): the oldest = the frame's casualties
        const uint32_t* src = reinterpret_cast<const uint32_t*>(c.base + (list_column(c.m.write_index) ? args.alive_off[1] : args.alive_off[0]));
        uint32_t* dead = reinterpret_cast<uint32_t*>(c.base + args.dead_off);
        const uint32_t* died = reinterpret_cast<const uint32_t*>(c.base + args.died_bits_off);
        // (the casualties dealt over ALL the instance's workgroups, one per thread and pass: a workgroup walking 4096 of them alone was a chain of 16
        // dependent round trips per thread - 0.0413 ms per C5 frame against 0.0384 for the rewrite it replaces, profiles/r05h_ab_ring.log)
        const uint32_t wg_rows = c.n == 0u ? 1u : (c.n + kChunk - 1u) / kChunk;
        for (uint32_t d = c.j * kBlock + tid; d < dead_n; d += wg_rows * kBlock) {
            const uint32_t slot = src[ring_row(head, first_dead + d, args.capacity)];
            dead[c.n - 1u - d] = slot;                                             // the d-th casualty in serial order lands on dead row n-1-d (vfx_update.wgsl:150-151)
            if (((died[slot >> 5] >> (slot & 31u)) & 1u) == 0u && args.fault) *args.fault = 1u;   // the host's proof, checked (as the suffix path does)
        }
        if (last && tid == 0) {
            const uint32_t survivors = c.n - dead_n;
            DevMeta o = c.m;
            o.alive_count = survivors;
            o.particle_counter = c.m.particle_counter + c.n_spawn;
            // the spawns are rows 0 .. n_spawn - 1 now (an emptied list needs no head: 0, so that no later path has to look through one)
            o.write_index = ((survivors == 0u ? 0u : ring_row(head, args.capacity - c.n_spawn, args.capacity)) << 1) | list_column(c.m.write_index);
            o.ref_write_index = c.m.ref_write_index ^ 1u;
            o.max_update = c.n; o.dead_count = dead_n; o.spawned = c.n_spawn; o.instance_count = survivors;
            store_meta(meta_out + c.k, o);
        }
        return;
    }
    if (total_dead == 0u && !rotate && !(args.force_rewrite && head != 0u)) {
        if (args.hz && tid == 0) { hz.D(args.hz_parity ^ 1u)[c.j] = hz.D(args.hz_parity)[c.j]; hz.BF(args.hz_parity ^ 1u)[c.j] = hz.BF(args.hz_parity)[c.j]; }   // no row moved
        if (last && tid == 0) {
            DevMeta o = c.m;
            o.alive_count = c.n;
            o.particle_counter = c.m.particle_counter + c.n_spawn;
            o.ref_write_index = c.m.ref_write_index ^ 1u;
            o.max_update = c.n; o.dead_count = 0; o.spawned = c.n_spawn; o.instance_count = c.n;
            store_meta(meta_out + c.k, o);
        }
        return;
    }
    if (args.slot_order) {  // the lists are rebuilt from the alive bytes: only the counters move
        if (c.j == 0 && tid == 0) {
            const uint32_t survivors = c.n - total_dead;
            DevMeta o = c.m;
            o.alive_count = survivors;
            o.particle_counter = c.m.particle_counter + c.n_spawn;
            o.ref_write_index = c.m.ref_write_index ^ 1u;
            o.max_update = c.n; o.dead_count = total_dead; o.spawned = c.n_spawn; o.instance_count = survivors;
            store_meta(meta_out + c.k, o);
        }
        return;
    }
    if (c.n == 0u) {
        // An EMPTY list that still stands behind a head (a trail that died out through ring frames) in a frame that rewrites (force_rewrite): no
        // chunk has rows, so nobody reaches the store at the end - the counters of two frames ago and the head would stay in meta_out while the
        // host forgets that a head may be set (HnbProgram::ring_live), and the next linear append / sort would read the list through the wrong rows
        // (ADVICE r5). Nothing to move: the head goes, the counters rotate.
        if (c.j == 0u && tid == 0u) {
            DevMeta o = c.m;
            o.alive_count = 0u;
            o.write_index = list_column(c.m.write_index);
            o.ref_write_index = c.m.ref_write_index ^ 1u;
            o.max_update = 0u; o.dead_count = 0u; o.spawned = 0u; o.instance_count = 0u;
            store_meta(meta_out + c.k, o);
        }
        return;
    }
    if (!has_rows) return;
    // exclusive prefix of the survivor counts of the earlier chunks of this instance
    // (an instance without a casualty gets here only to be rotated: k_count_rows recorded nothing for it, every row survives)
    const uint32_t* cnt = cb.counts + (size_t)c.k * args.chunks_per_inst;
    const uint32_t rows = (c.n - c.start) < kChunk ? (c.n - c.start) : kChunk;
    // Everything this workgroup reads is requested BEFORE the prefix over the earlier chunks' counts is waited for: its 4096 rows (16 per
    // lane) and the chunk's row-mask words. (The prefix ends in a barrier; issued behind it, the rows were a further dependent round trip
    // in a kernel that is a chain of them: metadata -> counts -> mask -> rows -> stores.)
    constexpr uint32_t kSteps = kWaveRows / 64u;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(c.base + ((c.m.write_index & 1u) ? args.alive_off[1] : args.alive_off[0]));
    const uint32_t row0 = ring_row(head, c.start, args.capacity);   // (head 0 everywhere but behind ring frames: rows are read through the ring, written linear)
    uint32_t v[kSteps];
    const bool nt = args.stream_hint != 0u;   // (uniform: one scalar branch around each unrolled run of accesses)
    if (nt) {
#pragma unroll
        for (uint32_t q = 0; q < kSteps; ++q) { const uint32_t i = wave * kWaveRows + q * 64u + lane; v[q] = i < rows ? __builtin_nontemporal_load(src + ring_row(row0, i, args.capacity)) : 0u; }
    } else {
#pragma unroll
        for (uint32_t q = 0; q < kSteps; ++q) { const uint32_t i = wave * kWaveRows + q * 64u + lane; v[q] = i < rows ? src[ring_row(row0, i, args.capacity)] : 0u; }
    }
    unsigned long long word;   // (bit r of word i: row 64 i + r survives; nothing died in the instance: every row that exists survives)
    const bool suffix = args.suffix_dead != 0u && total_dead != 0u;
    const uint32_t alive0 = c.m.alive_count;                                  // rows the frame started with; this frame's spawns follow
    const uint32_t first_dead = alive0 >= total_dead ? alive0 - total_dead : 0u;   // suffix: rows [first_dead, alive0) are the casualties
    if (suffix) {
        auto below_mask = [](long long x) { return x <= 0 ? 0ull : (x >= 64 ? ~0ull : ((1ull << x) - 1ull)); };
        const long long b0 = (long long)c.start + 64ll * lane;               // first row of this lane's word
        word = below_mask((long long)c.n - b0) & (below_mask((long long)first_dead - b0) | ~below_mask((long long)alive0 - b0));
    } else if (total_dead != 0u) word = reinterpret_cast<const unsigned long long*>(c.base + args.row_mask_off)[(size_t)c.j * (kChunk / 64u) + lane];
    else word = rows >= (lane + 1u) * 64u ? ~0ull : (rows > lane * 64u ? ((1ull << (rows - lane * 64u)) - 1ull) : 0ull);
    uint32_t excl = c.start;
    if (suffix) {   // survivors in front of row c.start, in closed form
        excl = c.start <= first_dead ? c.start : (c.start < alive0 ? first_dead : c.start - total_dead);
        if (total_dead > alive0 && tid == 0u && args.fault) *args.fault = 1u;
    } else if (total_dead != 0u) {
        // (Round 4, also tried: count and compact in ONE kernel - a workgroup keeps its rows and masks in registers, publishes its survivor count as
        // a tagged 8-byte word and polls the words of its group of 64 chunks and one total per earlier group. Bit-equal on the whole GPU suite and
        // SLOWER: c2_mixed 0.332 vs 0.316 ms, c2_events 0.410 vs 0.393, c2_dieoff 0.253 vs 0.248 (profiles/r04u_ab_fused_lists.log): under the
        // kernel's own streaming load a hand-off between workgroups costs 3-5 us per hop (the poll queues behind the CU's own loads), two hops
        // per workgroup, against 1.7 us for the kernel boundary it replaces and a second read of the rows that mostly hits the Infinity Cache.)
        // (Round 4: a two-level prefix - group sums accumulated by k_count_rows with one fire-and-forget atomic per chunk, 508 bytes of
        // counts per workgroup here - made THIS kernel 2 us faster and k_count_rows 16 us slower at 4096 chunks: 64 atomics to one word from
        // 8 XCDs serialise beyond the L2, profiles/r04b_kernel_durations.json. The counts are read 16 bytes per lane instead.)
        uint32_t part = 0;
        const uint32_t head = (uint32_t)((16u - ((size_t)cnt & 15u)) & 15u) / 4u;   // counts in front of the first 16-byte boundary
        for (uint32_t i = tid; i < (head < c.j ? head : c.j); i += kBlock) part += cnt[i];
        if (c.j > head) {
            const uint4* c4 = reinterpret_cast<const uint4*>(cnt + head);
            const uint32_t n4 = (c.j - head) / 4u;
            for (uint32_t i = tid; i < n4; i += kBlock) { const uint4 v = c4[i]; part += (v.x + v.y) + (v.z + v.w); }
            for (uint32_t i = head + n4 * 4u + tid; i < c.j; i += kBlock) part += cnt[i];
        }
        part = wave_sum_u32(part);
        if (lane == 0) s_red[wave] = part;
        __syncthreads();
        excl = 0;
#pragma unroll
        for (uint32_t w = 0; w < kBlock / 64; ++w) excl += s_red[w];
    }
    // The chunk's 64 row-mask words, one per lane, and the survivors in front of each word: every wave ranks its own rows from them, no LDS
    // and no barrier.
    const uint32_t wcount = (uint32_t)__popcll(word);
    uint32_t wincl = wcount;
#pragma unroll
    for (uint32_t off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(wincl, off, 64); if (lane >= off) wincl += y; }
    const uint32_t a = __shfl(wincl, 63, 64);   // survivors of this chunk ( == cnt[c.j] where k_count_rows ran)
    if (args.hz && tid == 0u && a != 0u) {      // this chunk's survivors land on rows [excl, excl + a): one target chunk or two inherit its horizon
        const unsigned long long d = hz.D(args.hz_parity)[c.j];
        const uint32_t bf = hz.BF(args.hz_parity)[c.j];
        const uint32_t t0 = excl / kChunk, t1 = (excl + a - 1u) / kChunk;
        atomicMin(&hz.D(args.hz_parity ^ 1u)[t0], d); atomicMin(&hz.BF(args.hz_parity ^ 1u)[t0], bf);
        if (t1 != t0) { atomicMin(&hz.D(args.hz_parity ^ 1u)[t1], d); atomicMin(&hz.BF(args.hz_parity ^ 1u)[t1], bf); }
    }
    const uint32_t wexcl = wincl - wcount;
    uint32_t* out = reinterpret_cast<uint32_t*>(c.base + ((c.m.write_index & 1u) ? args.alive_off[0] : args.alive_off[1]));   // (selects, not a dynamic index: the job-table variant kept the offsets in LDS otherwise)
    // survivor g of the instance goes to row g - or, rotated: the last n_spawn survivors are this frame's spawns (k_init appended them, none of
    // them dies in its first frame: a premise of the proof) and go first, everything older follows
    const uint32_t tail = rotate ? c.n_spawn : 0u;
    const uint32_t head_n = (c.n - total_dead) - tail;
    uint32_t* dead = reinterpret_cast<uint32_t*>(c.base + args.dead_off);
    const uint32_t dead_before = c.start - excl;
    // survivors keep their (stable, serial) order (vfx_update.wgsl:161-165). A wave owns 1024 consecutive rows, 64 per step, lane l row
    // 64 step + l: loads and stores of a step are contiguous.
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (uint32_t q = 0; q < kSteps; ++q) {
        const uint32_t wi = wave * kSteps + q;                       // (wave-uniform)
        const uint32_t i = wi * 64u + lane;                          // row within the chunk
        const unsigned long long m = __shfl(word, wi, 64);
        const uint32_t r = __shfl(wexcl, wi, 64) + (uint32_t)__popcll(m & below);   // survivors of the chunk in front of this row
        if ((m >> lane) & 1ull) {
            const uint32_t g = excl + r;
            st_hint(v[q], out + (g >= head_n ? g - head_n : g + tail), nt);
        } else if (i < rows) {
            // the d-th casualty in serial order lands on dead row n-1-d (vfx_update.wgsl:150-151); its alive byte is already 0
            st_hint(v[q], dead + (c.n - 1u - (dead_before + (i - r))), nt);
            if (suffix) {   // the host's proof, checked: this row's particle must be one of the frame's casualties
                const uint32_t bits = reinterpret_cast<const uint32_t*>(c.base + args.died_bits_off)[v[q] >> 5];
                if (((bits >> (v[q] & 31u)) & 1u) == 0u && args.fault) *args.fault = 1u;
            }
        }
    }
    if (last && tid == 0) {
        const uint32_t survivors = excl + a;
        if (args.hz && args.fault && survivors != c.n - total_dead) *args.fault = 1u;   // a horizon claimed a chunk free of casualties that was not
        DevMeta o = c.m;
        o.alive_count = survivors;
        o.particle_counter = c.m.particle_counter + c.n_spawn;
        o.write_index = list_column(c.m.write_index) ^ 1u;  // the list now lives in the other column, linear (head 0)
        o.ref_write_index = c.m.ref_write_index ^ 1u;
        o.max_update = c.n;
        o.dead_count = c.n - survivors;
        o.spawned = c.n_spawn;
        o.instance_count = survivors;
        store_meta(meta_out + c.k, o);
    }
}

struct CompactArgs {
    uint32_t capacity, chunks_per_inst;
    soff_t alive_off[2], dead_off;
    soff_t alive_flag_off;     // u8[capacity]: 0 free, 1 alive
    soff_t died_bits_off, row_mask_off;     // DevProgram: one bit per slot "died in this frame's update"; one bit per list row "survives"
    soff_t horizon_off;        // death horizons (DevProgram): hz = maintained (eligible program), hz_use = this frame's ticks are finite: may skip
    uint32_t hz, hz_use, hz_parity, frame_no;
    uint32_t* fault;           // HnbEffectMetadata::fault
    uint32_t suffix_dead;      // ribbon programs, host-proven (HnbProgram::frame_suffix): the list is in age order and every particle has the same
                               // lifetime, so the frame's casualties are rows [alive_count - deaths, alive_count) - the oldest of the rows the frame
                               // started with; this frame's spawns behind them survive. k_count_rows does not run; k_compact checks the died bit of
                               // every row it treats as a casualty (they are `deaths` distinct slots: all set <=> the sets are equal) or raises `fault`
    uint32_t slot_order;       // HNB_LIST_ORDER_SLOT: k_order_write rebuilds the lists from the alive bytes
    uint32_t stream_hint;      // 1: list rows are read and written with the nontemporal hint ("cache policy of streamed data")
    uint32_t rotate_front;     // ribbon programs, host-proven (HnbProgram::sort_front_*): this frame's spawns sort in front of every older
                               // particle, so the survivors are written [spawns | older ones] and the list needs no sort afterwards
    uint32_t ring;             // "Ring lists": rotate_front (or no spawn) AND suffix_dead hold: nothing is rewritten, the head moves (DevProgram::ring for k_init)
    uint32_t force_rewrite;    // a list with a head may be standing in a frame that is not a ring frame: rewrite it linear whatever died
};
#ifndef HNB_JIT_TU
__global__ void __launch_bounds__(kBlock)
k_compact(const CompactArgs args, const uint64_t* __restrict__ inst_base, const DevMeta* __restrict__ meta_in, DevMeta* __restrict__ meta_out,
          const DevFrameInst* __restrict__ fi, const CompactBufs cb) {
    compact_chunk(args, inst_base, meta_in, meta_out, fi, cb, blockIdx.x, gridDim.x);
}
#endif

// ---- slot-ordered alive lists (HNB_LIST_ORDER_SLOT) -----------------------------------------------------
// The reference's list order is whatever its atomics produce; the default here is the serial-thread order
// (stable compaction, spawns appended). In steady spawn/kill churn that order decorrelates from the slot order
// and every attribute access becomes a random 12-byte gather (measured: 0.7 TB/s algorithmic). In this mode the
// list is rebuilt in increasing slot order after every frame from one alive byte per slot (set by k_init,
// cleared by k_compact for the casualties), so the update streams through memory again, and the free slots are
// listed in increasing order as well, so spawns fill the lowest free slots. Per 4096-slot chunk: k_order_count
// counts the flags, k_order_write takes the cross-chunk prefix and enumerates the set and the clear slots.
#ifndef HNB_JIT_TU
__device__ __forceinline__ bool order_unchanged(const ChunkCtx& c, const CompactBufs& cb, const DevFrameInst* fi) {
    // no casualty and no spawn this frame: the list is already in slot order
    return fi[c.k].skip || (cb.deaths[(size_t)cb.parity * cb.table_cap + c.k] == 0u && c.n_spawn == 0u);
}
__global__ void __launch_bounds__(kBlock)
k_order_count(const CompactArgs args, const uint64_t* __restrict__ inst_base, const DevMeta* __restrict__ meta_in,
              const DevFrameInst* __restrict__ fi, const CompactBufs cb) {
    __shared__ uint32_t s_red[kBlock / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    ChunkCtx c;
    const uint32_t chunk = chunk_of_workgroup(cb.xcd_remap);
    chunk_setup(c, chunk, args, inst_base, meta_in, fi);
    if (order_unchanged(c, cb, fi)) return;
    const uint32_t first = c.j * kChunk;
    const uint32_t* flags4 = reinterpret_cast<const uint32_t*>(c.base + args.alive_flag_off + first);  // 4 slots per word
    uint32_t cnt = 0;
    for (uint32_t w = tid; w < kChunk / 4u; w += kBlock)
        if (first + w * 4u < args.capacity) cnt += (uint32_t)__popc(flags4[w]);  // flags are 0 / 1 bytes; planes are padded to 256 B
#pragma unroll
    for (uint32_t off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (lane == 0) s_red[wave] = cnt;
    __syncthreads();
    if (tid == 0) cb.counts[chunk] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
}
__global__ void __launch_bounds__(kBlock)
k_order_write(const CompactArgs args, const uint64_t* __restrict__ inst_base, const DevMeta* __restrict__ meta_in, const DevMeta* __restrict__ meta_out,
              const DevFrameInst* __restrict__ fi, const CompactBufs cb) {
    __shared__ uint32_t s_red[kBlock / 64];
    __shared__ uint32_t s_alive[kChunk], s_free[kChunk];  // the chunk's alive / free slots, ascending: written out coalesced
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    ChunkCtx c;
    chunk_setup(c, chunk_of_workgroup(cb.xcd_remap), args, inst_base, meta_in, fi);
    if (order_unchanged(c, cb, fi)) return;
    const uint32_t* cnt = cb.counts + (size_t)c.k * args.chunks_per_inst;
    uint32_t part = 0;
    for (uint32_t i = tid; i < c.j; i += kBlock) part += cnt[i];
#pragma unroll
    for (uint32_t off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    if (lane == 0) s_red[wave] = part;
    __syncthreads();
    const uint32_t excl = s_red[0] + s_red[1] + s_red[2] + s_red[3];  // alive slots in the earlier chunks of this instance
    __syncthreads();
    // 16 consecutive slots per thread
    constexpr uint32_t kPer = kChunk / kBlock;
    const uint32_t chunk_first = c.j * kChunk;
    const uint32_t first = chunk_first + tid * kPer;
    const uint8_t* flags = reinterpret_cast<const uint8_t*>(c.base + args.alive_flag_off);
    uint32_t mask = 0;
    if (first < args.capacity) {
        const uint4 f = *reinterpret_cast<const uint4*>(flags + first);  // 16 alive bytes (planes are padded to 256 B)
        const uint32_t words[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q)
#pragma unroll
            for (uint32_t b = 0; b < 4; ++b)
                if ((words[q] >> (8u * b)) & 1u) mask |= 1u << (q * 4u + b);
    }
    const uint32_t local = (uint32_t)__popc(mask);
    uint32_t incl = local;
#pragma unroll
    for (uint32_t off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(incl, off, 64); if (lane >= off) incl += y; }
    if (lane == 63) s_red[wave] = incl;
    __syncthreads();
    uint32_t pos = incl - local;  // alive slots of this chunk before this thread's
    for (uint32_t w = 0; w < wave; ++w) pos += s_red[w];
    const uint32_t chunk_alive = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    const uint32_t chunk_slots = chunk_first < args.capacity ? ((args.capacity - chunk_first) < kChunk ? (args.capacity - chunk_first) : kChunk) : 0u;
    uint32_t fpos = tid * kPer - pos;  // free slots of this chunk before this thread's
    for (uint32_t b = 0; b < kPer; ++b) {
        const uint32_t slot = first + b;
        if (slot >= args.capacity) break;
        if (mask & (1u << b)) s_alive[pos++] = slot;
        else s_free[fpos++] = slot;
    }
    __syncthreads();
    // alive slots go to the list, free slots to the dead rows [alive_total, capacity), both ascending: the next
    // init pass pops dead[alive_total + i], so spawns fill the lowest free slots
    uint32_t* list = reinterpret_cast<uint32_t*>(c.base + args.alive_off[meta_out[c.k].write_index & 1u]) + excl;
    for (uint32_t i = tid; i < chunk_alive; i += kBlock) list[i] = s_alive[i];
    const uint32_t alive_total = meta_out[c.k].alive_count;
    uint32_t* dead = reinterpret_cast<uint32_t*>(c.base + args.dead_off) + alive_total + (chunk_first - excl);  // chunk_first - excl = free slots before this chunk
    for (uint32_t i = tid; i < chunk_slots - chunk_alive; i += kBlock) dead[i] = s_free[i];
}
#endif

// ---- k_emit_events: order the staged spawn events (src/lib.rs:976-993 under serial thread order) ----
// Event e of the frame is the e-th (row, repeat) pair in alive-list order; it is stored iff e < capacity.
// One workgroup per (chunk, split) and channel loop: cross-chunk exclusive prefix of the chunk totals, then a workgroup scan over the
// chunk's rows (16 consecutive rows per thread) gives every row the index of its first event. Rows with FEW events (EventEmitCondition::
// Always: a handful per particle and frame) are written by their own thread; rows with MANY (OnDie: a rocket explodes into 1000 trail
// particles) are queued in LDS and written by the whole workgroup, 256 contiguous events per round, the heavy rows dealt by row index to
// the gridDim.y workgroups of the chunk. (One thread writing its row's 1000 events alone: 0.25 ms per frame for 280 explosions; a
// binary search per event over the row offsets in LDS: 0.28 ms, a chain of dependent LDS reads; profiles/r03d, r03e.)
#ifndef HNB_JIT_TU
constexpr uint32_t kEmitHeavy = 32u;   // events of one row from which the workgroup writes them together
__global__ void __launch_bounds__(kBlock)
k_emit_events(const DevProgram prog, const uint64_t* __restrict__ inst_base, const DevMeta* __restrict__ meta_in,
              const DevFrameInst* __restrict__ fi, const CompactBufs cb, unsigned long long* __restrict__ ev_host, const uint32_t frame_tag) {
    __shared__ uint32_t s_red[kBlock / 64];
    __shared__ uint32_t s_scan[kBlock / 64];
    __shared__ uint32_t s_heavy_n;
    __shared__ uint32_t s_heavy[kChunk][3];   // (slot, first event, events) of the chunk's heavy rows
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t chunk = chunk_of_workgroup(cb.xcd_remap);
    const uint32_t split = blockIdx.y, n_split = gridDim.y;
    ChunkCtx c;
    const bool has_rows = chunk_setup(c, chunk, prog, inst_base, meta_in, fi);
    const bool last = c.n == 0 ? c.j == 0 : (c.start < c.n && c.start + kChunk >= c.n);
    const uint32_t rows = has_rows ? ((c.n - c.start) < kChunk ? (c.n - c.start) : kChunk) : 0u;
    constexpr uint32_t kPer = kChunk / kBlock;  // rows per thread
    for (uint32_t ch = 0; ch < prog.n_event_channels; ++ch) {
        DevEventBuffer* ev = global_ptr<DevEventBuffer>(fi[c.k].ev_out[ch]);
        if (!ev) continue;  // nobody listens on this channel
        const uint32_t* tot = cb.ev_totals + (size_t)c.k * prog.chunks_per_inst * HNB_MAX_EVENT_CHANNELS + ch;
        uint32_t part = 0;
        for (uint32_t i = tid; i < c.j; i += kBlock) part += tot[(size_t)i * HNB_MAX_EVENT_CHANNELS];
#pragma unroll
        for (uint32_t off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        __syncthreads();
        if (lane == 0) s_red[wave] = part;
        if (tid == 0) s_heavy_n = 0u;
        __syncthreads();
        uint32_t excl = 0;
#pragma unroll
        for (uint32_t w = 0; w < kBlock / 64; ++w) excl += s_red[w];
        const uint32_t mine = has_rows ? tot[(size_t)c.j * HNB_MAX_EVENT_CHANNELS] : 0u;
        if (last && tid == 0 && split == 0u) {
            ev->count[fi[c.k].ev_parity] = excl + mine;  // GpuChildInfo::event_count of this frame
            // ... and a copy for the host ({frame, count} in host-mapped memory, no read-back): when it has arrived by the time the next
            // frame is enqueued, the child's init grid is sized for the events that exist instead of for the buffer's capacity
            if (ev_host) *reinterpret_cast<volatile unsigned long long*>(ev_host + (size_t)c.k * HNB_MAX_EVENT_CHANNELS + ch) = ((unsigned long long)(excl + mine) << 32) | frame_tag;
        }
        const uint32_t capacity = ev->capacity;
        if (mine == 0u || excl >= capacity) continue;   // (uniform: every thread of the workgroup takes the same way)
        const uint32_t* cnt = reinterpret_cast<const uint32_t*>(c.base + prog.ev_cnt_off[ch]);                 // per slot
        const uint32_t* slots = reinterpret_cast<const uint32_t*>(c.base + prog.alive_off[list_column(c.m.write_index)]) + c.start;  // rows as the update saw them (programs that emit events never keep a ring: head 0)
        uint32_t slot[kPer], n_ev[kPer], local = 0;
#pragma unroll
        for (uint32_t r = 0; r < kPer; ++r) {
            const uint32_t row = tid * kPer + r;
            slot[r] = row < rows ? slots[row] : 0u;
            n_ev[r] = row < rows ? cnt[slot[r]] : 0u;
            local += n_ev[r];
        }
        // workgroup exclusive scan of the per-thread sums
        uint32_t incl = local;
#pragma unroll
        for (uint32_t off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(incl, off, 64); if (lane >= off) incl += y; }
        if (lane == 63) s_scan[wave] = incl;
        __syncthreads();
        uint32_t pos = excl + incl - local;
        for (uint32_t w = 0; w < wave; ++w) pos += s_scan[w];
#pragma unroll
        for (uint32_t r = 0; r < kPer; ++r) {
            if (n_ev[r] >= kEmitHeavy) {          // queued for the whole workgroup (any order: every entry carries its own range) - of the
                if ((tid * kPer + r) % n_split == split) {   // split that owns the ROW (the queue order differs from workgroup to workgroup)
                    const uint32_t h = atomicAdd(&s_heavy_n, 1u);
                    s_heavy[h][0] = slot[r]; s_heavy[h][1] = pos; s_heavy[h][2] = n_ev[r];
                }
            } else if (split == 0u) {             // a few events: this thread writes them (an event past the buffer's capacity is dropped)
                const uint32_t end = pos + n_ev[r] < capacity ? pos + n_ev[r] : capacity;
                for (uint32_t e = pos; e < end; ++e) ev->data[e] = slot[r];
            }
            pos += n_ev[r];
        }
        __syncthreads();
        const uint32_t n_heavy = s_heavy_n;
        for (uint32_t h = 0; h < n_heavy; ++h) {
            const uint32_t sl = s_heavy[h][0], first = s_heavy[h][1];
            const uint32_t end = first + s_heavy[h][2] < capacity ? first + s_heavy[h][2] : capacity;
            for (uint32_t e = first + tid; e < end; e += kBlock) ev->data[e] = sl;
        }
        __syncthreads();
    }
}
#endif

// ---- register budgets of the streaming kernels ------------------------------------------------------
#ifndef HNB_STREAM_WAVES
// Waves per SIMD the lean streaming kernels are register-budgeted for. Measured on MI355X (16M firework, the
// row-major predecessor of k_update_slots_stream): 8 (<= 64 VGPRs) 0.266 ms, 6 (<= 80 VGPRs) 0.230 ms: a tight
// budget serialises the loads of a 48-byte-per-lane plane that should all be in flight, and lines are fetched twice.
#define HNB_STREAM_WAVES 6
#endif
#ifndef HNB_STREAM_WAVES_FULL
#define HNB_STREAM_WAVES_FULL 5
#endif

// =====================================================================================================
// Slot-major update
// =====================================================================================================
// What a particle becomes in a frame does not depend on where it sits in the alive list (the PRNG is seeded
// by the slot, vfx_update.wgsl:138); only the LISTS depend on the row order. So the update walks the SLOTS,
// driven by one alive byte per slot (0 free, 1 alive, 2 died in this frame), and never touches the alive
// list: every access is a 16-byte load / store of 4 consecutive slots whatever the list looks like after hours
// of spawn / kill churn. The row order only matters to a second, much lighter pass over the rows of the
// instances that lost particles (k_count_rows: 4 bytes of list + one bit gather per row), which feeds the same
// chunk-local / cross-chunk compaction as before, and to k_emit_events.
template <bool B> struct BoolTag { static constexpr bool value = B; };
struct SlotArgs {
    uint32_t capacity, n_uregs, chunks_per_inst, n_inst;
    soff_t alive_flag_off;
    uint32_t update_len;
    soff_t plane_off[4];     // position, velocity, age, lifetime
    uint32_t flags;          // bit i: load pinned attr i; bit 4+i: store pinned attr i
    soff_t died_bits_off;    // DevProgram::died_bits_off; written (all of it) iff write_died: the frame's list kernels read it
    uint32_t write_died;
    uint32_t cull_lifetime;  // 1: lifetime culling (below); lmin_off = f32[chunks_per_inst] in the slab, dt_operand = operand a of the AGE_TICK
    soff_t lmin_off;
    uint32_t dt_operand;
    uint32_t age_cohort;     // 1: chunks whose alive particles all have the same AGE keep it in one word (below)
    soff_t horizon_off;      // death horizons: the instance's clock is advanced here (horizon != 0)
    uint32_t horizon;
    uint32_t frame_phase;    // frames this program ran, mod 16: staggers the re-check of chunks known to hold mixed ages (cohort state 4)
    const Ins* update_code;
    // "No particle can die before ..." (below): safe_words = u32[2][safe_stride] float bits, the frame's minimum remaining life
    // per chunk, double-buffered by safe_parity; safe_host = host-mapped {frame tag, bound bits} the host reads without any
    // synchronisation; publish_tag = index of the previous frame of this program. Null pointers: the program is not eligible.
    uint32_t* safe_words;
    unsigned long long* safe_host;
    uint32_t safe_parity, publish_tag, safe_stride;
    // skip_lists: the host PROVED that this frame has no spawn and no casualty (see hnb_simulate), so k_count_rows / k_compact
    // are not launched and this kernel rotates the counters itself (vfx_indirect.wgsl:57-85), one thread per instance.
    uint32_t skip_lists;
    const DevMeta* meta_in;
    DevMeta* meta_out;
    uint32_t* fault;         // set to 1 if a particle dies in a frame whose lists were skipped (never, unless the proof is wrong)
    uint32_t transpose;      // 1: vec3 planes of the per-particle path go through the wave's LDS transpose (xpose_load3 / xpose_store3)
    uint32_t stream_hint;    // 1: read-only planes (LIFETIME, alive bytes) are loaded with the nontemporal hint ("cache policy of streamed data")
    uint32_t store_hint;     // 1: the per-particle path stores its planes with the nontemporal hint (update_stream_chunk)
    uint32_t quarters;       // 1, or 4 (r6, the merged launches of small programs without cohorts): FOUR workgroups per 4096-slot chunk, a wave per 256-slot step. A program
                             // of a few chunks was a few workgroups whose waves walked four steps one after the other - four dependent load / run / store rounds, the
                             // longest link of a small scene's frame. The per-chunk state a quarter cannot know alone is left alone (the "completely alive" flag is
                             // only ever cleared, the lifetime bound only ever used), the no-death bound of the chunk is the minimum of four (atomicMin)
    uint32_t age_current;    // 1 (with age_cohort; HNB_AGE_COHORT_AUTO for an asset whose render modifiers read AGE): a chunk that keeps its common age in the
                             // value word ALSO writes it to the plane for its alive slots - write-only, 4 of the 8 bytes the cohort saves - so the AGE plane
                             // is current after every frame without a second pass over it (until round 6: a k_materialise_age launch behind every update)
};

// The deaths of a frame are known on the device only after its update ran, and HIP has no indirect dispatch: the list
// kernels of a frame are launched by a host that runs several frames ahead. For the effects whose particles can only die
// of old age (the streamable stacks without kill modifiers: every burst effect between its burst and its die-off) the
// device tells the host how long that cannot happen: every update computes R = min over the alive particles of
// (lifetime - age) - 1e-5 * lifetime (with the chunk's lifetime bound Lm where the lifetimes were not loaded: a lower
// bound of the same expression), the NEXT frame's kernel publishes {frame, R} to host-mapped memory, and the host skips the
// list kernels of a frame F as long as the ticks accumulated since the published frame stay below R (at most 64 frames
// ahead, no spawn or host write in between). age accumulates one rounding error of 2^-24 relative per frame: 64 frames
// stay inside the 1e-5 * lifetime margin. In such a frame this kernel is the ONLY launch of the program.

// Lifetime culling. In a streamable update the LIFETIME plane is read for one thing: `is_alive = age < lifetime` right
// after `age += dt` (src/lib.rs:1223-1258). Per 4096-slot chunk the slab keeps Lm, a lower bound of the lifetime of
// every alive particle of the chunk (the exact minimum when it was last recomputed; deaths can only raise the true
// minimum, a spawn resets the bound to "unknown"). A wave step whose alive particles all satisfy age + dt < Lm cannot
// lose a particle - age + dt < Lm <= lifetime, the very comparison the program makes, no rounding involved - so it
// does not load the lifetimes at all: 4 of the 61.7 bytes per particle of the firework update. Steps that cannot
// prove it load the plane and test exactly as before; when every step of a chunk loaded it, the chunk's bound is
// recomputed. A burst effect runs culled for most of its particles' lives; an effect that spawns into every chunk every
// frame recomputes every frame and costs what it did before.

// Age cohorts. A burst spawns its particles in one frame with one initial AGE, and AGE_TICK adds the same dt to all of them:
// the alive particles of a chunk then share ONE age, bit for bit, for the rest of their lives, and reading and writing 4 + 4
// bytes of it per particle per frame (8 of the 56 the firework update moves) carries no information. Per chunk the slab keeps
// a state word and a value word (behind the lifetime bounds and the "completely alive" flags):
//   state 0  the AGE plane holds every age (the default);
//   state 1  every alive particle of the chunk has age == value; the plane is stale for the alive slots (dead slots keep the age
//            they died with: the kernel stores the age of a particle in the frame it dies, as the reference's write-back does);
//   state 2  state 1 + this frame's spawns, whose ages ARE in the plane and whose alive byte is 3 (k_init sets both; within a frame only);
//   state 4  state 0, and the last check found survivors of DIFFERENT ages: the check (min / max over the survivors' age bits in every
//            step) is skipped in fifteen frames of sixteen, and the chunk runs the per-particle path without any cohort bookkeeping.
// The update reads the state, takes the value instead of the plane where it may (2: per slot, by the alive byte), and after the
// program checks whether the survivors' new ages are all equal (min == max of their bit patterns): then it writes the value
// and skips the plane, else it stores the ages (which materialises them) and returns to state 0. Host reads of the AGE plane
// materialise first (k_materialise_age), host writes reset the states. Eligible programs: the lifetime-culling ones (the
// stream starts with its only AGE_TICK, nothing else writes AGE) without ribbons (the sort reads the plane); HNB_OPT_AGE_COHORT = OFF switches them off.

// The died bits of one wave step (256 slots, lane l owns slots 4 l .. 4 l + 3 and brings their four bits in `nib`) as 8 dwords of the
// linear bit array: dword d holds lanes 8 d .. 8 d + 7. An OR over every group of 8 lanes in three DPP steps (quad_perm [1,0,3,2],
// quad_perm [2,3,0,1], row_half_mirror), then the first lane of each group stores: 32 contiguous bytes per step.
__device__ __forceinline__ void store_died_bits(uint32_t* __restrict__ bits, uint32_t step_first, uint32_t nib, uint32_t lane) {
    int v = (int)(nib << (4u * (lane & 7u)));
    v |= __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);
    if ((lane & 7u) == 0u) bits[(step_first >> 5) + (lane >> 3)] = (uint32_t)v;
}

// The workgroup's LDS of update_stream_chunk, ONE object however many instantiations of the template a kernel holds (a function-local
// __shared__ array of a template is one allocation per instantiation: k_update_jobs with its two paid 48 KiB, a set module -
// hnb_jit.h, one instantiation per program of the scene - would not fit at all).
struct StreamLds {
    uint32_t died[kBlock / 64];
    float lmin[kBlock / 64];
    uint32_t alive[kBlock / 64];
    float rem[kBlock / 64];
    uint32_t amin[kBlock / 64], amax[kBlock / 64];
    u4v xp[2][kBlock / 64][kStepRows * 3u / 4u];   // position / velocity staging of each wave's step (xpose_load3): 24 KiB per workgroup
};
__device__ __forceinline__ StreamLds& stream_lds() {
    __shared__ StreamLds s;
    return s;
}

// PROBE (tools/stream_probe.hip only; 0 in the product): 4 = skip stores, 8 = skip the program.
// COHORT: compile the age-cohort paths in (programs that are eligible: SlotArgs::age_cohort); false leaves the kernel as it was.
template <class PROG, int PROBE, bool COHORT>
__device__ __forceinline__ void update_stream_chunk(const SlotArgs& args, const uint64_t* __restrict__ inst_base, const DevFrameInst* __restrict__ fi,
                                                    const uint32_t* __restrict__ ublocks, const CompactBufs& cb,
                                                    const uint32_t wg, const uint32_t wg_total) {   // workgroup wg of the program's wg_total (k_update_slots_stream, k_update_stream_jobs)
    StreamLds& lds = stream_lds();
    uint32_t (&s_died)[kBlock / 64] = lds.died;
    float (&s_lmin)[kBlock / 64] = lds.lmin;
    uint32_t (&s_alive)[kBlock / 64] = lds.alive;
    float (&s_rem)[kBlock / 64] = lds.rem;
    uint32_t (&s_amin)[kBlock / 64] = lds.amin;
    uint32_t (&s_amax)[kBlock / 64] = lds.amax;
    u4v (&s_xp)[2][kBlock / 64][kStepRows * 3u / 4u] = lds.xp;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const bool split = !COHORT && args.quarters == 4u;   // (SlotArgs::quarters; wave-uniform)
    const uint32_t chunk = split ? chunk_of_workgroup(cb.xcd_remap, wg >> 2, wg_total >> 2) : chunk_of_workgroup(cb.xcd_remap, wg, wg_total);
    const uint32_t quarter = split ? (wg & 3u) : 0u;
    const uint32_t k = chunk / args.chunks_per_inst, j = chunk - k * args.chunks_per_inst;
    const uint32_t wave0 = j * kChunk + (split ? (quarter * (kBlock / 64u) + wave) * kStepRows : wave * kWaveRows);   // the wave's first slot
    const uint32_t n_steps = split ? 1u : kWaveRows / kStepRows;
    const uint32_t n_words = split ? wg_total >> 2 : wg_total;   // chunks of the program
    // (r6: the instance's row - frozen or not, its slab - requested here, in front of the publisher and the counter rotation, not behind their branches:
    // every dependent round trip at the head of a workgroup is a microsecond of a small frame)
    const uint32_t frozen = fi[k].skip;
    const uint64_t slab = inst_base[k];
    // Plane stores of a program whose WRITTEN planes exceed the Infinity Cache by half (SlotArgs::store_hint, plan::use_store_hints): nontemporal.
    // Same-box A/B (profiles/r04o_ab_walk.log, r04p_ab_walk2.log): c2_mixed 0.327 -> 0.316 ms, c2_events 0.456 -> 0.438; C3 - 8.4M particles,
    // 235 MB of written planes: they FIT the cache and the next frame's walk finds them there - 0.097 -> 0.120 ms with the hint, hence the size rule.
    const bool store_nt = args.store_hint != 0u;
#ifndef HNB_PUBLISH_FIRST_WG
#define HNB_PUBLISH_FIRST_WG 1
#endif
    // (r6: the FIRST workgroup of the launch publishes, not chunk 0: in a frame that walks downwards chunk 0 is the last workgroup dispatched, and its
    // store to host memory behind two barriers in front of its own chunk was the tail of the kernel)
    if (args.safe_words && (HNB_PUBLISH_FIRST_WG ? wg == 0u : chunk == 0u)) {  // publish the previous frame's bound: its kernel has completed, every chunk's word is final
        uint32_t* prev = args.safe_words + (size_t)(args.safe_parity ^ 1u) * args.safe_stride;
        uint32_t m = 0x7f800000u;
        // (each word goes back to +inf once read: the frame after next may merge the four quarters of a chunk into it with atomicMin - SlotArgs::quarters)
        for (uint32_t i = tid; i < n_words; i += kBlock) { const uint32_t v = prev[i]; prev[i] = 0x7f800000u; m = v < m ? v : m; }
        m = wave_min_u32(m);
        if (lane == 0) s_alive[wave] = m;
        __syncthreads();
        if (tid == 0u) {
            for (uint32_t w = 1; w < kBlock / 64; ++w) m = s_alive[w] < m ? s_alive[w] : m;
            *reinterpret_cast<volatile unsigned long long*>(args.safe_host) = ((unsigned long long)m << 32) | (unsigned long long)args.publish_tag;
        }
        __syncthreads();
    }
    if (args.skip_lists && j == 0u && quarter == 0u && tid == 0u) {  // counter rotation of a frame without spawn and casualty (k_compact's zero-casualty path)
        DevMeta o = args.meta_in[k];
        if (!frozen) {
            o.ref_write_index ^= 1u;
            o.max_update = o.alive_count; o.dead_count = 0u; o.spawned = 0u; o.instance_count = o.alive_count;
        }
        args.meta_out[k] = o;
        cb.deaths[(size_t)(cb.parity ^ 1u) * cb.table_cap + k] = 0u;
    }
    if (frozen) return;  // frozen instance
    char* base = global_ptr<char>(slab);
    VmUniforms U;
    U.u = ublocks + (size_t)k * args.n_uregs;
    U.xf = fi[k].xf;
    float rem_min = __builtin_inff();  // min over this lane's particles that stay alive of (lifetime - age) - 1e-5 * lifetime
    char* p_pos = base + args.plane_off[0];
    char* p_vel = base + args.plane_off[1];
    char* p_age = base + args.plane_off[2];
    char* p_life = base + args.plane_off[3];
    uint32_t* flags4 = reinterpret_cast<uint32_t*>(base + args.alive_flag_off);  // 4 alive bytes per word
    const uint32_t fl = args.flags;
    uint32_t died_total = 0;  // wave-uniform
    // Chunks known to be completely alive skip the alive bytes as well (u32[chunks_per_inst] after the lifetime bounds:
    // 1 = every slot of the chunk holds a live particle). Only this kernel maintains the flag: it sets it after counting
    // 4096 alive slots and no casualty, and clears it when a particle of the chunk dies; spawns only ever go to chunks
    // with free slots, whose flag is already clear.
    uint32_t* cfull = reinterpret_cast<uint32_t*>(base + args.lmin_off) + args.chunks_per_inst;
    const bool chunk_full = cfull[j] == 1u;
    uint32_t* astate = cfull + args.chunks_per_inst;      // age cohorts: state and value per chunk
    uint32_t* aval = astate + args.chunks_per_inst;
    // (state 4: "the chunk holds particles of different ages" - found by the check below. Such a chunk is not checked again in every frame -
    // in a spawn / die steady state it never becomes uniform, and the bookkeeping of the check is a third of the per-particle path's
    // instructions - but in one frame of sixteen, staggered over the chunks; everywhere else state 4 is state 0: the plane holds the ages)
    const uint32_t ast_raw = COHORT ? astate[j] : 0u;  // wave-uniform
    const bool mixed = COHORT && ast_raw == 4u && ((args.frame_phase + j) & 15u) != 0u;
    const uint32_t ast = ast_raw == 4u ? 0u : ast_raw;
    const float A = COHORT ? u2f(aval[j]) : 0.0f;
    uint32_t amin = 0xffffffffu, amax = 0u;               // bit patterns of the ages of the particles that stay alive
    uint32_t lane_alive = 0;
    const bool cull = args.cull_lifetime != 0u;
    float* lmin = reinterpret_cast<float*>(base + args.lmin_off);
    const float Lm = cull ? lmin[j] : 0.0f;           // 0 (or anything not > 0): unknown, every step loads the lifetimes
    const float dt_tick = cull ? uf(U, args.dt_operand) : 0.0f;
    if (args.horizon && j == 0u && quarter == 0u && tid == 0u) {       // the instance's clock: once per simulated frame (frozen instances returned above)
        double* clk = horizon_view(base, args.horizon_off, args.chunks_per_inst).clock;
        *clk = *clk + (double)(dt_tick > 0.0f ? dt_tick : 0.0f) * (1.0 + 0x1p-16);
    }
    float wave_min = __builtin_inff();                // minimum lifetime of the particles that stay alive (steps that loaded them)
    bool loaded_all = true;                           // wave-uniform: every step with alive slots loaded the lifetimes
    // ---- the flat path: a completely alive chunk of a component-wise program (PROG::kFlat: AGE_TICK, VEL_SCALE, VEL_ADD, EULER only) whose
    // ages live in the cohort word and whose particles provably all survive this frame (one comparison: the chunk's age + tick against its
    // smallest lifetime) has nothing per particle left to decide. Its position / velocity planes are then streamed as plain float arrays: every
    // access of a wave covers 1 KiB contiguous (lane l takes 16-byte word l), instead of three 16-byte words per lane at a 48-byte lane
    // stride, where each instruction of a wave touches 24 cache lines for 1 KiB of payload. Float k of a plane is component k mod 3, and the ops
    // are component-wise: same IEEE operations in the same order on every float (apply_static_flat). Measured on the bare access patterns
    // (tools/flat_probe.hip, 16.7M particles, alternating walk): 0.138-0.142 ms with the quads, 0.119-0.122 ms flat.
    bool flat = false;
    if constexpr (PROG::kFlat && COHORT && PROBE == 0) {
        flat = chunk_full && ast == 1u && cull && Lm > 0.0f && (A + dt_tick < Lm) && (fl & 3u) == 3u && !(fl & 128u);
    }
    if (flat) {
        if constexpr (PROG::kFlat && COHORT && PROBE == 0) {
            const float A2 = A + dt_tick;   // mac_age_tick's arithmetic; A2 < Lm <= every lifetime: every particle of the chunk stays alive
            u4v* pw = reinterpret_cast<u4v*>(p_pos + (size_t)j * (kChunk * 12u)) + wave * (kWaveRows * 3u / 4u);
            u4v* vw = reinterpret_cast<u4v*>(p_vel + (size_t)j * (kChunk * 12u)) + wave * (kWaveRows * 3u / 4u);
            const uint32_t rot = lane % 3u;   // component of this lane's first float in every word it takes: (word index) mod 3 = (3 step + w + lane) mod 3, w added below
            if (args.age_current && (fl & 64u)) {   // the plane kept current: the wave's 1024 ages, 16 bytes per lane and step, issued in front of the loads below
                // (behind the loop, and / or with the nontemporal hint: no difference on c2 or c4, two rounds on one box - profiles/r06d_ab_age_store.log)
                u4v* aw = reinterpret_cast<u4v*>(p_age + (size_t)j * (kChunk * 4u)) + wave * (kWaveRows / 4u);
                const uint32_t a2 = f2u(A2);
#pragma unroll
                for (uint32_t step = 0; step < kWaveRows / kStepRows; ++step) aw[step * 64u + lane] = u4v{a2, a2, a2, a2};
            }
#pragma unroll 2   // (all four steps unrolled: 0.1313 instead of 0.1298 ms, three A/B rounds on one box)
            for (uint32_t step = 0; step < kWaveRows / kStepRows; ++step) {
                float P[3][4], V[3][4];
#pragma unroll
                for (uint32_t w = 0; w < 3; ++w) {
                    const u4v a = pw[(step * 3u + w) * 64u + lane], b = vw[(step * 3u + w) * 64u + lane];
                    P[w][0] = u2f(a.x); P[w][1] = u2f(a.y); P[w][2] = u2f(a.z); P[w][3] = u2f(a.w);
                    V[w][0] = u2f(b.x); V[w][1] = u2f(b.y); V[w][2] = u2f(b.z); V[w][3] = u2f(b.w);
                }
                PROG::run_flat(args.update_code, P, V, rot, U);
#pragma unroll
                for (uint32_t w = 0; w < 3; ++w) {
                    if (fl & 16u) pw[(step * 3u + w) * 64u + lane] = u4v{f2u(P[w][0]), f2u(P[w][1]), f2u(P[w][2]), f2u(P[w][3])};
                    if (fl & 32u) vw[(step * 3u + w) * 64u + lane] = u4v{f2u(V[w][0]), f2u(V[w][1]), f2u(V[w][2]), f2u(V[w][3])};
                }
            }
            if (args.write_died && lane < kWaveRows / 32u)                // nobody died here (the frame's list kernels read every word)
                reinterpret_cast<uint32_t*>(base + args.died_bits_off)[((j * kChunk + wave * kWaveRows) >> 5) + lane] = 0u;
            amin = amax = f2u(A2);                                         // the survivors' common age
            loaded_all = false;                                            // no lifetime was loaded: the chunk's bound stands
            if (args.safe_words) rem_min = (f2u(A2) >> 31) ? 0.0f : (Lm - A2) - 1.0e-5f * Lm;   // as the per-particle form computes it from X.lifetime = Lm, X.age = A2 (a negative age: no claim, see there)
        }
    }
    // ---- the per-particle path, one wave step (256 slots, 4 per lane). COH: with the age-cohort bookkeeping (a chunk that is known to hold
    // mixed ages - state 4 - runs without it: see `mixed` above)
    auto step_body = [&](auto coh_tag, const uint32_t step, const uint32_t f4, const u4v* pre_age = nullptr) {
        constexpr bool COH = decltype(coh_tag)::value;
        const uint32_t s0 = wave0 + step * kStepRows + lane * 4u;  // first of this lane's 4 slots
        bool was[4], fresh[4];  // fresh: spawned this frame into a chunk that kept its ages in the value word (alive byte 3, state 2 only)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint32_t byte = (f4 >> (8 * p)) & 0xffu;
            fresh[p] = COH && byte == 3u;
            was[p] = byte == 1u || fresh[p];
            lane_alive += was[p] ? 1u : 0u;
        }
        const bool any = was[0] || was[1] || was[2] || was[3];
        if (!__any(any)) {
            if (args.write_died) store_died_bits(reinterpret_cast<uint32_t*>(base + args.died_bits_off), s0 - lane * 4u, 0u, lane);
            return;
        }
        const bool full = was[0] && was[1] && was[2] && was[3];
        uint32_t slot[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) slot[p] = s0 + p;
        bool lanes_on[4] = {any, any, any, any};  // loads: the whole quad whenever one of its slots is alive
        Pinned<4> X;
#pragma unroll
        for (int p = 0; p < 4; ++p) { X.pos[p] = V3{0, 0, 0}; X.vel[p] = V3{0, 0, 0}; X.age[p] = 0.0f; X.lifetime[p] = 0.0f; X.alive[p] = true; }
        bool need_life = true;  // wave-uniform
        float age_was[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // the AGE quad as loaded: free slots of a partially alive quad get their own bytes back without a second read
        // (wave-uniform) the step's 256 slots lie inside the planes: its vec3 planes go through the LDS transpose, every lane taking part
        const uint32_t step_first = s0 - lane * 4u;
        // (... where at least half of the quads hold a live particle: in the last frames of a die-off the direct path touches fewer lines)
        const bool xp = args.transpose != 0u && PROBE == 0 && step_first + kStepRows <= args.capacity && __popcll(__ballot(any)) >= 32;
        if (xp) {
            if (fl & 1u) xpose_load3(X.pos, p_pos, step_first, s_xp[0][wave], lane);
            if (fl & 2u) xpose_load3(X.vel, p_vel, step_first, s_xp[1][wave], lane);
        }
        if (any) {
            if (!xp) {
                if (fl & 1u) pin_load3<4>(X.pos, p_pos, slot, lanes_on, true);
                if (fl & 2u) pin_load3<4>(X.vel, p_vel, slot, lanes_on, true);
            }
            if (fl & 4u) {
                if (!COH || ast != 1u) {
                    if (pre_age) { X.age[0] = u2f(pre_age->x); X.age[1] = u2f(pre_age->y); X.age[2] = u2f(pre_age->z); X.age[3] = u2f(pre_age->w); }   // (requested with the alive bytes: age-only programs)
                    else pin_load1<4>(X.age, p_age, slot, lanes_on, true);
                    for (int p = 0; p < 4; ++p) age_was[p] = X.age[p];
                }
                if (COH && ast != 0u) {   // (selects, no divergence: ast is uniform, fresh[] only ever set in state 2)
#pragma unroll
                    for (int p = 0; p < 4; ++p) X.age[p] = fresh[p] ? X.age[p] : A;
                }
            }
        }
        // can this step lose a particle? `age + dt` is the AGE_TICK's own arithmetic. (A chunk known to hold mixed ages does not ask: some
        // particle of a step is always near its end there, and the question makes the lifetime loads wait for the ages - a third dependent
        // memory round trip per step.)
        if (cull && Lm > 0.0f && (COH || !mixed)) {
            bool may_die = false;
#pragma unroll
            for (int p = 0; p < 4; ++p) may_die = may_die || (was[p] && !(X.age[p] + dt_tick < Lm));
            need_life = __any(may_die);
        }
        if (any && (fl & 8u)) {
            if (need_life) {
                if (args.stream_hint) {   // (a read-only plane: see "cache policy of streamed data")
                    const u4v ql = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(p_life) + (slot[0] >> 2));
                    X.lifetime[0] = u2f(ql.x); X.lifetime[1] = u2f(ql.y); X.lifetime[2] = u2f(ql.z); X.lifetime[3] = u2f(ql.w);
                } else pin_load1<4>(X.lifetime, p_life, slot, lanes_on, true);
            }
            else {
#pragma unroll
                for (int p = 0; p < 4; ++p) X.lifetime[p] = Lm;  // age + dt < Lm holds for every alive slot of the step
            }
        }
        if (!need_life) loaded_all = false;
        if constexpr (!(PROBE & 8)) PROG::template run<4>(args.update_code, args.update_len, X, U);
        if constexpr (!(PROBE & 4)) {
            // (a plane that is stored without having been loaded has no staged copy: the direct path)
            const bool xp_pos = xp && (fl & 17u) == 17u, xp_vel = xp && (fl & 34u) == 34u;
            if (xp_pos) xpose_store3(X.pos, p_pos, step_first, s_xp[0][wave], lane, was, full, store_nt);
            if (xp_vel) xpose_store3(X.vel, p_vel, step_first, s_xp[1][wave], lane, was, full, store_nt);
            if (any) {  // a full quad is stored with 16-byte stores; otherwise only the alive slots are written
                if ((fl & 16u) && !xp_pos) pin_store3<4>(X.pos, p_pos, slot, was, full);
                if ((fl & 32u) && !xp_vel) pin_store3<4>(X.vel, p_vel, slot, was, full);
                if (fl & 64u) {
                    const bool plane_loaded = !COH || ast != 1u;   // (state 1: the ages live in the value word - stored only where the plane is kept current)
                    if (plane_loaded || args.age_current) {
                        if (plane_loaded && (fl & 4u) && !full) {   // loaded above: blend in registers, one 16-byte store
                            float q[4];
#pragma unroll
                            for (int p = 0; p < 4; ++p) q[p] = was[p] ? X.age[p] : age_was[p];
                            pin_store1<4>(q, p_age, slot, was, true, store_nt);
                        } else pin_store1<4>(X.age, p_age, slot, was, full, store_nt);
                    }
                }
                if (fl & 128u) pin_store1<4>(X.lifetime, p_life, slot, was, full);
            }
        } else {
            float acc = 0.0f;  // keep the loads alive
#pragma unroll
            for (int p = 0; p < 4; ++p) acc += X.pos[p].x + X.pos[p].y + X.pos[p].z + X.vel[p].x + X.vel[p].y + X.vel[p].z + X.age[p] + X.lifetime[p];
            if (acc == 123.456f) X.alive[0] = false;
        }
        if (cull && need_life) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (was[p] && X.alive[p]) wave_min = fminf(wave_min, X.lifetime[p]);
        }
        if (args.safe_words) {  // X.lifetime holds the lifetimes, or Lm where they were not loaded (a lower bound of each of them)
#pragma unroll
            for (int p = 0; p < 4; ++p)
                // (r6: a particle whose age carries the sign bit - a negative tick, a negative initial age, -0 - makes NO claim: the bound is 0 and the frames
                // that follow run their list kernels. A list-free frame may then also skip the ribbon sort - enqueue_ribbon_sort: keys are age BITS, and a
                // uniform tick keeps their order only while no age crosses zero)
                if (was[p] && X.alive[p]) rem_min = fminf(rem_min, (f2u(X.age[p]) >> 31) ? 0.0f : (X.lifetime[p] - X.age[p]) - 1.0e-5f * X.lifetime[p]);
        }
        if constexpr (COH) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {   // min / max of the survivors' age bits (selects)
                const bool stays = was[p] && X.alive[p];
                const uint32_t b = f2u(X.age[p]);
                amin = (stays && b < amin) ? b : amin;
                amax = (stays && b > amax) ? b : amax;
            }
            if constexpr (!(PROBE & 4)) {
                // a chunk in state 1 keeps its ages in the value word; a particle that dies now leaves its last age in the plane
                // (rare: one wave-uniform vote per step keeps the stores out of the way)
                const bool dies = (was[0] && !X.alive[0]) || (was[1] && !X.alive[1]) || (was[2] && !X.alive[2]) || (was[3] && !X.alive[3]);
                if (ast == 1u && (fl & 64u) && !args.age_current && __any(dies)) {   // (age_current: stored above with everybody's)
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        if (was[p] && !X.alive[p]) reinterpret_cast<float*>(p_age)[slot[p]] = X.age[p];
                }
            }
        }
        uint32_t nf = f4, nib = 0u;
        uint32_t died_here = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const bool died = was[p] && !X.alive[p];
            if (died) { nf &= ~(0xffu << (8 * p)); nib |= 1u << p; }   // the slot is free from now on; the lists learn it from the died bit
            else if (COH && fresh[p]) nf = (nf & ~(0xffu << (8 * p))) | (1u << (8 * p));   // an ordinary alive slot from now on
            died_here += (uint32_t)__popcll(__ballot(died));
        }
        if (nf != f4) flags4[s0 >> 2] = nf;
        if (args.write_died) store_died_bits(reinterpret_cast<uint32_t*>(base + args.died_bits_off), step_first, nib, lane);
        died_total += died_here;
    };
    if (!flat) {
        // the alive bytes of the wave's four steps, requested together (a step's own load would wait behind the previous step's stores to the
        // same plane: two dependent round trips per step)
        // (the cohort instantiations, budgeted for 4 waves, and the component-wise programs have the registers for it: -1 % on the churn
        // frames; the others - C3's force field at 5 waves - lost 2.5 % to it and load each step's word where it is used, profiles/r03s_ab.log)
        auto f4_of = [&](const uint32_t step) {
            const uint32_t s0 = wave0 + step * kStepRows + lane * 4u;
            return chunk_full ? 0x01010101u : (s0 < args.capacity ? ld_hint(flags4 + (s0 >> 2), args.stream_hint != 0u) : 0u);  // the plane is padded: slots past the capacity read 0
        };
        if constexpr (COHORT || PROG::kFlat) {
            uint32_t f4s[kWaveRows / kStepRows];
#pragma unroll
            for (uint32_t step = 0; step < kWaveRows / kStepRows; ++step) f4s[step] = step < n_steps ? f4_of(step) : 0u;
            // Age-only programs (PROG::kAgeOnly: ribbon.rs - a trail that does not move): the step is two dependent round trips for 8 bytes per
            // particle - alive bytes, then the ages - four times in a row, and a 4M-particle effect is one round of workgroups: the kernel is
            // those latencies. The ages of all four steps are requested together with the alive bytes (16 registers, only in these instantiations).
            if constexpr (PROG::kAgeOnly && !COHORT) {
                if ((fl & 4u) && !(fl & 3u)) {
                    u4v ages[kWaveRows / kStepRows];
#pragma unroll
                    for (uint32_t step = 0; step < kWaveRows / kStepRows; ++step) {
                        const uint32_t s0 = wave0 + step * kStepRows + lane * 4u;
                        ages[step] = (step < n_steps && s0 < args.capacity) ? reinterpret_cast<const u4v*>(p_age)[s0 >> 2] : u4v{0u, 0u, 0u, 0u};   // (planes are padded to 256 B: a quad never straddles the end)
                    }
#pragma unroll
                    for (uint32_t step = 0; step < kWaveRows / kStepRows; ++step) if (step < n_steps) step_body(BoolTag<false>{}, step, f4s[step], &ages[step]);
                    goto steps_done;
                }
            }
            if (mixed) {
#pragma unroll
                for (uint32_t step = 0; step < kWaveRows / kStepRows; ++step) if (step < n_steps) step_body(BoolTag<false>{}, step, f4s[step]);
            } else {
#pragma unroll
                for (uint32_t step = 0; step < kWaveRows / kStepRows; ++step) if (step < n_steps) step_body(BoolTag<COHORT>{}, step, f4s[step]);
            }
        } else {
#pragma unroll
            for (uint32_t step = 0; step < kWaveRows / kStepRows; ++step) if (step < n_steps) step_body(BoolTag<false>{}, step, f4_of(step));
        }
    }
steps_done:
    if (cull) wave_min = wave_min_f32(wave_min);
    if (!chunk_full) lane_alive = wave_sum_u32(lane_alive);
    if (lane == 0) s_alive[wave] = lane_alive;
    // (a negative value marks a wave that skipped a load; a real negative minimum reads the same: the bound then simply
    // stays unknown, which is always correct)
    if (args.safe_words) {
        rem_min = wave_min_f32(rem_min);
        if (lane == 0) s_rem[wave] = rem_min;
    }
    if constexpr (COHORT) {
        amin = wave_min_u32(amin); amax = wave_max_u32(amax);
        if (lane == 0) { s_amin[wave] = amin; s_amax[wave] = amax; }
    }
    if (lane == 0) { s_died[wave] = died_total; s_lmin[wave] = loaded_all ? wave_min : -1.0f; }
    lds_barrier();
    if (tid == 0) {
        if (COHORT && !mixed) {   // do the survivors share one age? (amin > amax: there are none)
            uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
            for (uint32_t w = 0; w < kBlock / 64; ++w) { lo = s_amin[w] < lo ? s_amin[w] : lo; hi = s_amax[w] > hi ? s_amax[w] : hi; }
            // a chunk that was NOT in state 1 stored its ages in this launch; one that was, did not: it may only stay in state 1
            // (same value for all survivors by construction) or empty out (lo > hi -> 0: nothing alive, the plane is right for the dead).
            // Survivors of different ages: state 4, the plane holds them and the check takes a rest
            const uint32_t next = (lo == hi) ? 1u : (lo > hi ? 0u : 4u);
            if (next == 1u) aval[j] = lo;
            if (next != ast_raw) astate[j] = next;
        }
        uint32_t d = 0;
#pragma unroll
        for (uint32_t w = 0; w < kBlock / 64; ++w) d += s_died[w];
        if (d) atomicAdd(&cb.deaths[(size_t)cb.parity * cb.table_cap + k], d);
        if (d && args.skip_lists) *args.fault = 1u;  // the host's proof was wrong: report it (HnbEffectMetadata::fault)
        if (args.safe_words) {  // one word per chunk and frame parity, plain store (same-address atomics from 8 XCDs cost ~0.1 us EACH)
            float r = fminf(fminf(s_rem[0], s_rem[1]), fminf(s_rem[2], s_rem[3]));
            r = r > 0.0f ? r : 0.0f;  // non-negative floats order like their bit patterns; +inf: no live particle in the chunk
            uint32_t* word = args.safe_words + (size_t)args.safe_parity * args.safe_stride + chunk;
            if (split) atomicMin(word, f2u(r));   // the chunk's four quarters (the word was +inf: the publisher of the frame before left it so)
            else *word = f2u(r);
        }
        if (chunk_full) { if (d) cfull[j] = 0u; }
        else if (!split && d == 0u && s_alive[0] + s_alive[1] + s_alive[2] + s_alive[3] == kChunk) cfull[j] = 1u;   // (a quarter cannot know)
        if (cull && !split) {  // every step of the chunk loaded the lifetimes: the exact minimum over the particles still alive
            float m = __builtin_inff();
            bool all = true;
#pragma unroll
            for (uint32_t w = 0; w < kBlock / 64; ++w) { all = all && !(s_lmin[w] < 0.0f); m = fminf(m, s_lmin[w]); }
            if (all) lmin[j] = m < 3.0e38f ? m : 3.0e38f;  // an empty chunk: any finite bound; a spawn resets it
        }
    }
}

template <class PROG, int WAVES, int PROBE = 0, bool COHORT = false>
__global__ void __launch_bounds__(kBlock, WAVES)
k_update_slots_stream(const SlotArgs args, const uint64_t* __restrict__ inst_base, const DevFrameInst* __restrict__ fi,
                      const uint32_t* __restrict__ ublocks, const CompactBufs cb) {
    update_stream_chunk<PROG, PROBE, COHORT>(args, inst_base, fi, ublocks, cb, blockIdx.x, gridDim.x);
}

#ifndef HNB_JIT_TU
// ---- k_update_slots_stream_age (r6): the update that is ONE AGE_TICK, as a kernel of its own ------------------------------------------------------------
// ribbon.rs, lightning.rs: trails that do not move. `age += dt; alive = age < lifetime` over one scalar plane is 9 bytes per particle, and a 4M-particle
// trail (C5) is ONE round of 1024 workgroups: the kernel's duration is the life of a wave, and inside k_update_slots_stream<ProgAge> that was a chain of
// dependent round trips in front of the first plane load (kernel arguments fetched in four places, the instance row, the chunk's flags, the tick) in a
// 30 KB instantiation whose position / velocity paths (runtime flags) are dead weight - 17-19 us for 37 MB where a plain kernel moves as much in 5
// (profiles/r06n_c5_counters.log, r06o_dispatch_probe.log). Same protocol, same results bit for bit as update_stream_chunk<ProgAge> under lifetime culling
// without cohorts (launch_stream_age decides; the parity gate's plain replay - culling off - keeps running the general kernel beside it), but: everything
// a wave can need is requested in ONE round behind the instance row (alive bytes - also of chunks flagged completely alive -, the ages of its four steps,
// the chunk's bound and flag, the tick), the lifetimes only in steps where somebody may die, counters per lane, one reduction per wave.
#ifndef HNB_AGEK_CUT
#define HNB_AGEK_CUT 0   // (tools/r06ab.sh: timing-only builds that cut pieces out of the kernel; 0 in the product)
#endif
__global__ void __launch_bounds__(kBlock)
k_update_slots_stream_age(const SlotArgs args, const uint64_t* __restrict__ inst_base, const DevFrameInst* __restrict__ fi,
                          const uint32_t* __restrict__ ublocks, const CompactBufs cb) {
    __shared__ uint32_t s_died[kBlock / 64], s_alive[kBlock / 64];
    __shared__ float s_lmin[kBlock / 64], s_rem[kBlock / 64];
    constexpr uint32_t kSteps = kWaveRows / kStepRows;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t wg = blockIdx.x, wg_total = gridDim.x;
    const uint32_t chunk = chunk_of_workgroup(cb.xcd_remap, wg, wg_total);
    const uint32_t k = chunk / args.chunks_per_inst, j = chunk - k * args.chunks_per_inst;
    // round 1: the instance row
    const uint32_t frozen = fi[k].skip;
    char* base = global_ptr<char>(inst_base[k]);
    const float dt_tick = u2f(ublocks[(size_t)k * args.n_uregs + (args.dt_operand & 0xffu)]);   // uf(U, dt_operand)
    // round 2: everything the chunk can need
    char* p_age = base + args.plane_off[2];
    const char* p_life = base + args.plane_off[3];
    uint32_t* flags4 = reinterpret_cast<uint32_t*>(base + args.alive_flag_off);
    float* lmin = reinterpret_cast<float*>(base + args.lmin_off);
    uint32_t* cfull = reinterpret_cast<uint32_t*>(lmin) + args.chunks_per_inst;
    const uint32_t wave0 = j * kChunk + wave * kWaveRows;
    uint32_t f4s[kSteps];
    u4v ages[kSteps];
#pragma unroll
    for (uint32_t step = 0; step < kSteps; ++step) {
        const uint32_t s0 = wave0 + step * kStepRows + lane * 4u;
        const bool in = s0 < args.capacity;   // (planes are padded to 256 B: a quad never straddles the end; slots past the capacity read 0)
        f4s[step] = in ? ld_hint(flags4 + (s0 >> 2), args.stream_hint != 0u) : 0u;
        ages[step] = in ? reinterpret_cast<const u4v*>(p_age)[s0 >> 2] : u4v{0u, 0u, 0u, 0u};
    }
    const uint32_t cf = cfull[j];
    const float Lm = lmin[j];                 // 0 (or anything not > 0): unknown, every step loads the lifetimes
    if (!(HNB_AGEK_CUT & 1) && args.safe_words && wg == 0u) {        // publish the previous frame's bound: its kernel has completed, every chunk's word is final
        // (the FIRST workgroup of the launch, not chunk 0: in a frame that walks downwards chunk 0 is the last one dispatched, and the store to host memory
        // behind two barriers was the tail of the kernel)
        uint32_t* prev = args.safe_words + (size_t)(args.safe_parity ^ 1u) * args.safe_stride;
        uint32_t m = 0x7f800000u;
        for (uint32_t i = tid; i < wg_total; i += kBlock) { const uint32_t v = prev[i]; prev[i] = 0x7f800000u; m = v < m ? v : m; }   // (back to +inf: update_stream_chunk's publisher)
        m = wave_min_u32(m);
        if (lane == 0) s_alive[wave] = m;
        __syncthreads();
        if (tid == 0u) {
            for (uint32_t w = 1; w < kBlock / 64; ++w) m = s_alive[w] < m ? s_alive[w] : m;
            *reinterpret_cast<volatile unsigned long long*>(args.safe_host) = ((unsigned long long)m << 32) | (unsigned long long)args.publish_tag;
        }
        __syncthreads();
    }
    if (args.skip_lists && j == 0u && tid == 0u) {  // counter rotation of a frame without spawn and casualty (k_compact's zero-casualty path)
        DevMeta o = args.meta_in[k];
        if (!frozen) {
            o.ref_write_index ^= 1u;
            o.max_update = o.alive_count; o.dead_count = 0u; o.spawned = 0u; o.instance_count = o.alive_count;
        }
        args.meta_out[k] = o;
        cb.deaths[(size_t)(cb.parity ^ 1u) * cb.table_cap + k] = 0u;
    }
    if (frozen) return;
    if (!(HNB_AGEK_CUT & 2) && args.horizon && j == 0u && tid == 0u) {       // the instance's clock: once per simulated frame
        double* clk = horizon_view(base, args.horizon_off, args.chunks_per_inst).clock;
        *clk = *clk + (double)(dt_tick > 0.0f ? dt_tick : 0.0f) * (1.0 + 0x1p-16);
    }
    const bool chunk_full = cf == 1u;
    const bool bound_known = Lm > 0.0f;
    uint32_t* died_bits = reinterpret_cast<uint32_t*>(base + args.died_bits_off);
    float an[kSteps][4];
    uint32_t need_mask = 0u;   // wave-uniform: steps in which somebody may die (or the bound is unknown): they load the lifetimes
#pragma unroll
    for (uint32_t step = 0; step < kSteps; ++step) {
        if (chunk_full) f4s[step] = 0x01010101u;   // (what the flag promises; the bytes were requested anyway: they cost no round trip of their own)
        const uint32_t w = f4s[step];
        const uint32_t a4[4] = {ages[step].x, ages[step].y, ages[step].z, ages[step].w};
        bool may = false;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            an[step][p] = u2f(a4[p]) + dt_tick;                                  // mac_age_tick's arithmetic
            const bool was = ((w >> (8 * p)) & 0xffu) == 1u;
            may = may || (was && !(bound_known && an[step][p] < Lm));
        }
        if (!(HNB_AGEK_CUT & 4) && __any(may)) need_mask |= 1u << step;
    }
    // The lifetimes: for a wave with a step in which somebody may die, ALL of its steps' quads in one round, no lane-dependent branch around the loads (the
    // first version loaded per step and lane under `if`, and the compiler waited for each load inside its branch: four dependent round trips in exactly the
    // workgroups that are the kernel's tail - the dozen chunks the frame's spawns went to and the dozen its casualties come from)
    u4v life[kSteps];
#pragma unroll
    for (uint32_t step = 0; step < kSteps; ++step) life[step] = u4v{0u, 0u, 0u, 0u};
    if (need_mask != 0u) {   // wave-uniform
#pragma unroll
        for (uint32_t step = 0; step < kSteps; ++step) {
            const uint32_t s0 = wave0 + step * kStepRows + lane * 4u;
            const u4v* pl = reinterpret_cast<const u4v*>(p_life) + (s0 < args.capacity ? (s0 >> 2) : 0u);   // (a lane past the end reads quad 0: its value is not used)
            life[step] = args.stream_hint ? __builtin_nontemporal_load(pl) : *pl;
        }
    }
    uint32_t lane_died = 0u, lane_alive = 0u;
    float wave_min = __builtin_inff();   // minimum lifetime of the particles that stay alive (steps that loaded them)
    float rem_min = __builtin_inff();    // min over this lane's particles that stay alive of (lifetime - age) - 1e-5 * lifetime
    float an_max = -__builtin_inff();    // ... of the steps that did not load the lifetimes: the largest new age (see below)
    uint32_t an_sign = 0u;               // ... and the OR of their new ages' bits (bit 31: somebody's age is negative or -0)
    bool loaded_all = true;              // wave-uniform: every step with alive slots loaded the lifetimes
    // Three kinds of step (wave-uniform), because four waves share a SIMD here and the 640 VALU instructions per wave of the first version were 4 us of the
    // kernel's 12 (the bare access pattern: 5, tools/probes/age_stream_probe.hip): (A) nobody can die and every slot is alive - a trail in its steady state -:
    // one add per slot; (B) nobody can die: the selects; (C) somebody may: the lifetimes, the death bookkeeping.
#pragma unroll
    for (uint32_t step = 0; step < kSteps; ++step) {
        const uint32_t w = f4s[step];
        const uint32_t step_first = wave0 + step * kStepRows;
        const bool need_s = ((need_mask >> step) & 1u) != 0u;   // wave-uniform
        if (!need_s) {
            // nobody dies: age + dt < Lm <= lifetime for every live slot. The no-death bound's term (Lm - an) - 1e-5 Lm is a non-increasing function of an
            // (two correctly rounded steps, each monotonic): its minimum over the slots is its value at the largest an - one max per slot here, the term once.
            const uint32_t s0 = step_first + lane * 4u;
            if (!chunk_full) lane_alive += (uint32_t)__popc(w & 0x01010101u);
            if (args.write_died && (lane & 7u) == 0u) died_bits[(step_first >> 5) + (lane >> 3)] = 0u;   // store_died_bits of no casualty
            if (__all(w == 0x01010101u)) {   // (A)
                an_max = fmaxf(an_max, fmaxf(fmaxf(an[step][0], an[step][1]), fmaxf(an[step][2], an[step][3])));
                an_sign |= (f2u(an[step][0]) | f2u(an[step][1])) | (f2u(an[step][2]) | f2u(an[step][3]));
                __builtin_nontemporal_store((u4v{f2u(an[step][0]), f2u(an[step][1]), f2u(an[step][2]), f2u(an[step][3])}), reinterpret_cast<u4v*>(p_age) + (s0 >> 2));
                loaded_all = false;
                continue;
            }
            if (!__any(w != 0u)) continue;
            loaded_all = false;
            const uint32_t a4[4] = {ages[step].x, ages[step].y, ages[step].z, ages[step].w};   // (B)
            uint32_t q[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const bool was = ((w >> (8 * p)) & 0xffu) == 1u;
                q[p] = was ? f2u(an[step][p]) : a4[p];        // free slots get their own bytes back: one 16-byte store
                an_max = fmaxf(an_max, was ? an[step][p] : -__builtin_inff());
                an_sign |= was ? f2u(an[step][p]) : 0u;
            }
            if (w != 0u) __builtin_nontemporal_store((u4v{q[0], q[1], q[2], q[3]}), reinterpret_cast<u4v*>(p_age) + (s0 >> 2));
            continue;
        }
        const uint32_t s0 = step_first + lane * 4u;
        const bool any = w != 0u;                        // (bytes are 0 or 1 here: no cohorts, and no mark outlives the init pass)
        if (!chunk_full) lane_alive += (uint32_t)__popc(w & 0x01010101u);
        if (!__any(any)) {
            if (args.write_died) store_died_bits(died_bits, step_first, 0u, lane);
            continue;
        }
        constexpr bool need = true;   // (C)
        const uint32_t a4[4] = {ages[step].x, ages[step].y, ages[step].z, ages[step].w};
        const uint32_t l4[4] = {life[step].x, life[step].y, life[step].z, life[step].w};
        uint32_t q[4], nib = 0u, nf = w;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const bool was = ((w >> (8 * p)) & 0xffu) == 1u;
            const float lf = need ? u2f(l4[p]) : Lm;
            const bool stays = was && an[step][p] < lf;   // (without the lifetimes: age + dt < Lm <= lifetime holds for every live slot of the step)
            q[p] = was ? f2u(an[step][p]) : a4[p];        // free slots get their own bytes back: one 16-byte store
            if (was && !stays) { nib |= 1u << p; nf &= ~(0xffu << (8 * p)); }
            if (stays) {
                if (need) wave_min = fminf(wave_min, lf);
                if (args.safe_words) rem_min = fminf(rem_min, (f2u(an[step][p]) >> 31) ? 0.0f : (lf - an[step][p]) - 1.0e-5f * lf);
            }
        }
        // (always nontemporal: the plain store cost the C5 frame 0.0323 against 0.0309 ms in three rounds on one box, profiles/r06r_ab_lean_nt.log)
        if (any) __builtin_nontemporal_store((u4v{q[0], q[1], q[2], q[3]}), reinterpret_cast<u4v*>(p_age) + (s0 >> 2));
        if (need) {
            lane_died += (uint32_t)__popc(nib);
            if (nf != w) flags4[s0 >> 2] = nf;            // the slot is free from now on; the lists learn it from the died bit
        }
        if (args.write_died) store_died_bits(died_bits, step_first, nib, lane);
    }
    if (HNB_AGEK_CUT & 16) return;
    lane_died = need_mask != 0u ? wave_sum_u32(lane_died) : 0u;   // (wave-uniform from here on)
    wave_min = need_mask != 0u ? wave_min_f32(wave_min) : __builtin_inff();
    if (!chunk_full) lane_alive = wave_sum_u32(lane_alive);
    if (args.safe_words) {
        if (an_max > -__builtin_inff()) rem_min = fminf(rem_min, (an_sign >> 31) ? 0.0f : (Lm - an_max) - 1.0e-5f * Lm);   // the steps without lifetimes: lf = Lm there; an age with the sign bit: no claim (update_stream_chunk)
        rem_min = wave_min_f32(rem_min);
    }
    // (a negative value marks a wave that skipped a load; a real negative minimum reads the same: the bound then simply stays unknown, which is always correct)
    if (lane == 0) { s_died[wave] = lane_died; s_alive[wave] = lane_alive; s_rem[wave] = rem_min; s_lmin[wave] = loaded_all ? wave_min : -1.0f; }
    lds_barrier();
    if (tid == 0) {
        uint32_t d = 0;
#pragma unroll
        for (uint32_t w = 0; w < kBlock / 64; ++w) d += s_died[w];
        if (d) atomicAdd(&cb.deaths[(size_t)cb.parity * cb.table_cap + k], d);
        if (HNB_AGEK_CUT & 8) return;
        if (d && args.skip_lists) *args.fault = 1u;  // the host's proof was wrong: report it (HnbEffectMetadata::fault)
        if (args.safe_words) {  // one word per chunk and frame parity, plain store
            float r = fminf(fminf(s_rem[0], s_rem[1]), fminf(s_rem[2], s_rem[3]));
            r = r > 0.0f ? r : 0.0f;  // non-negative floats order like their bit patterns; +inf: no live particle in the chunk
            args.safe_words[(size_t)args.safe_parity * args.safe_stride + chunk] = f2u(r);
        }
        if (chunk_full) { if (d) cfull[j] = 0u; }
        else if (d == 0u && s_alive[0] + s_alive[1] + s_alive[2] + s_alive[3] == kChunk) cfull[j] = 1u;
        float m = __builtin_inff();   // every step of the chunk loaded the lifetimes: the exact minimum over the particles still alive
        bool all = true;
#pragma unroll
        for (uint32_t w = 0; w < kBlock / 64; ++w) { all = all && !(s_lmin[w] < 0.0f); m = fminf(m, s_lmin[w]); }
        if (all) lmin[j] = m < 3.0e38f ? m : 3.0e38f;  // an empty chunk: any finite bound; a spawn resets it
    }
}
#endif

// Any update program on the V register file: one slot per lane, same protocol.
template <class CODE>
__device__ __forceinline__ void update_generic_chunk(const DevProgram& prog, const uint64_t* __restrict__ inst_base, const DevFrameInst* __restrict__ fi,
                                                     const uint32_t* __restrict__ ublocks, const CompactBufs& cb, const uint32_t write_died,
                                                     const uint32_t wg, const uint32_t wg_total,
                                                     const uint32_t sub_begin = 0u, const uint32_t sub_end = kChunk / kBlock) {
    // (sub_begin, sub_end: the 256-slot groups of the chunk this workgroup takes - all 16 in the program's own launch, one in k_update_jobs,
    // where the latency of a single small chunk is the frame)
    __shared__ uint32_t s_died[kBlock / 64], s_alive[kBlock / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t chunk = chunk_of_workgroup(cb.xcd_remap, wg, wg_total);
    const uint32_t k = chunk / prog.chunks_per_inst, j = chunk - k * prog.chunks_per_inst;
    if (fi[k].skip) return;
    char* base = global_ptr<char>(inst_base[k]);
    uint8_t* flags = reinterpret_cast<uint8_t*>(base + prog.alive_flag_off);
    const uint32_t seed_k = fi[k].seed, slot_base = fi[k].slot_base;
    VmUniforms U;
    U.u = ublocks + (size_t)k * prog.n_uregs;
    U.xf = fi[k].xf;
    uint32_t died_total = 0, alive_total = 0;  // wave-uniform
    // completely alive chunks skip the alive bytes (same flag and rules as in k_update_slots_stream)
    uint32_t* cfull = reinterpret_cast<uint32_t*>(base + prog.lmin_off) + prog.chunks_per_inst;
    const bool chunk_full = cfull[j] == 1u;
    for (uint32_t sub = sub_begin; sub < sub_end; ++sub) {
        const uint32_t slot = j * kChunk + sub * kBlock + tid;
        const bool valid = chunk_full || (slot < prog.capacity && flags[slot] == 1u);
        alive_total += (uint32_t)__popcll(__ballot(valid));
        // one died bit per slot for the list kernels: a wave's 64 consecutive slots are one word of the array (SlotArgs::died_bits_off)
        unsigned long long* died_word = reinterpret_cast<unsigned long long*>(base + prog.died_bits_off) + ((slot - lane) >> 6);
        if (!__any(valid)) {
            if (write_died && lane == 0u) *died_word = 0ull;
            continue;
        }
        VmState<typename CODE::file_t> S;
        S.r = typename CODE::file_t{};
        CODE::load_update(prog, S, base, slot, valid);
        S.pindex = slot + slot_base;
        S.seed = pcg_hash(S.pindex ^ seed_k);  // vfx_update.wgsl:138
        S.pcounter = 0u;
        S.alive = true;
        VmAttrIO io;
        io.slab = base; io.attrs = prog.attrs; io.slot = slot;
        if (valid) {
            CODE::run_update(prog, S, U, io);
            CODE::store_update(prog, S, base, slot);
            if (prog.n_event_channels) {  // this slot's spawn events; k_emit_count / k_emit_events order them by list row
#pragma unroll
                for (uint32_t ch = 0; ch < HNB_MAX_EVENT_CHANNELS; ++ch)
                    if (ch < prog.n_event_channels) reinterpret_cast<uint32_t*>(base + prog.ev_cnt_off[ch])[slot] = S.ev[ch];
            }
            if (!S.alive) flags[slot] = 0u;   // free from now on; the lists learn it from the died bit
        }
        const unsigned long long dm = __ballot(valid && !S.alive);
        if (write_died && lane == 0u) *died_word = dm;
        died_total += (uint32_t)__popcll(dm);
    }
    if (lane == 0) { s_died[wave] = died_total; s_alive[wave] = alive_total; }
    __syncthreads();
    if (tid == 0) {
        uint32_t d = 0, a = 0;
#pragma unroll
        for (uint32_t w = 0; w < kBlock / 64; ++w) { d += s_died[w]; a += s_alive[w]; }
        if (d) atomicAdd(&cb.deaths[(size_t)cb.parity * cb.table_cap + k], d);
        if (chunk_full) { if (d) cfull[j] = 0u; }
        else if (d == 0u && a == kChunk) cfull[j] = 1u;   // (never reached by a workgroup that saw only a part of the chunk: the flag is an optimisation)
    }
}

// split != 0 (r6): one workgroup per 256 slots instead of per 4096-slot chunk (the grid is 16 x the chunks), as the merged launches do it. A program of a few
// chunks is a few workgroups walking 16 groups of 256 slots one after the other - 16 dependent load / run / store rounds: the rocket effect of firework.rs,
// 8 chunks, took 33-70 us per update. The host splits where the whole grid still fits the GPU at once (hanabi_amd.hip: kGenericSplitMaxChunks).
template <class CODE>
__global__ void __launch_bounds__(kBlock)
k_update_slots_generic(const DevProgram prog, const uint64_t* __restrict__ inst_base, const DevFrameInst* __restrict__ fi,
                       const uint32_t* __restrict__ ublocks, const CompactBufs cb, const uint32_t write_died, const uint32_t split) {
    if (split) {
        const uint32_t w = blockIdx.x / (kChunk / kBlock), sub = blockIdx.x % (kChunk / kBlock);
        update_generic_chunk<CODE>(prog, inst_base, fi, ublocks, cb, write_died, w, gridDim.x / (kChunk / kBlock), sub, sub + 1u);
    } else update_generic_chunk<CODE>(prog, inst_base, fi, ublocks, cb, write_died, blockIdx.x, gridDim.x);
}

#ifndef HNB_JIT_TU
// Writes the common age of every cohort chunk (state 1) into the AGE plane for its alive slots: before the host reads the plane, for a
// device-side consumer that asks (hnb_effect_materialise), and at the end of every frame for programs whose render modifiers read AGE
// (HNB_AGE_COHORT_AUTO). One workgroup per chunk of every instance in inst_base[0 .. gridDim.x / chunks_per_inst).
__global__ void __launch_bounds__(kBlock)
k_materialise_age(const uint64_t* __restrict__ inst_base, uint32_t capacity, uint32_t chunks_per_inst, soff_t lmin_off, soff_t age_plane_off, soff_t alive_flag_off) {
    const uint32_t k = blockIdx.x / chunks_per_inst, j = blockIdx.x - k * chunks_per_inst;
    char* base = global_ptr<char>(inst_base[k]);
    const uint32_t* astate = reinterpret_cast<const uint32_t*>(base + lmin_off) + 2u * chunks_per_inst;
    const uint32_t* aval = astate + chunks_per_inst;
    if (astate[j] != 1u) return;
    const uint32_t v = aval[j];
    // a lane owns quads of 4 consecutive slots: 16-byte stores; a completely alive chunk (the flag the update maintains behind the lifetime bounds)
    // needs no alive bytes at all. (Round 5: one byte load and one 4-byte store per slot took 34 us for 16.7M particles, a quarter of the update
    // it follows in a host that materialises every frame.) The planes are padded to 256 B: a quad never straddles the end of the plane.
    const bool chunk_full = (reinterpret_cast<const uint32_t*>(base + lmin_off) + chunks_per_inst)[j] == 1u;
    const uint32_t* flags4 = reinterpret_cast<const uint32_t*>(base + alive_flag_off);
    u4v* age4 = reinterpret_cast<u4v*>(base + age_plane_off);
    for (uint32_t q = threadIdx.x; q < kChunk / 4u; q += kBlock) {
        const uint32_t quad = j * (kChunk / 4u) + q, slot = quad * 4u;
        if (slot >= capacity) break;
        const uint32_t f4 = chunk_full ? 0x01010101u : flags4[quad];
        if (f4 == 0x01010101u) { age4[quad] = u4v{v, v, v, v}; continue; }
        if (f4 == 0u) continue;
        uint32_t* a = reinterpret_cast<uint32_t*>(age4 + quad);
#pragma unroll
        for (uint32_t p = 0; p < 4u; ++p)
            if (((f4 >> (8u * p)) & 0xffu) == 1u) a[p] = v;
    }
}

// Row-major list maintenance for the instances that lost particles this frame, first pass: which rows survive? Every row's slot is
// looked up in the died bits the update left (one bit per slot: the table of a 16.7M-slot effect is 2 MiB and stays in the L2 of every
// XCD, where the alive BYTES it replaces - 16 MiB, a quarter of them L2 hits - cost this kernel 0.74 GB of fabric reads and 0.23 ms in
// steady spawn / kill churn, profiles/r03a_*). Output: one survivor bit per row (ballots: lane l of a step owns row 64 step + l, so a
// ballot IS the mask word) and the chunk's survivor count for k_compact's cross-chunk prefix. No row is moved here.
__device__ __forceinline__ void count_rows_chunk(const CompactArgs& args, const uint64_t* __restrict__ inst_base, const DevMeta* __restrict__ meta_in,
                                                 const DevFrameInst* __restrict__ fi, const CompactBufs& cb, uint32_t wg, uint32_t wg_total) {
    __shared__ uint32_t s_wave[kBlock / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t chunk = chunk_of_workgroup(cb.xcd_remap, wg, wg_total);
    ChunkCtx c;
    const uint32_t total_dead = cb.deaths[(size_t)cb.parity * cb.table_cap + chunk / args.chunks_per_inst];   // (requested with the instance's row)
    const bool has_rows = chunk_setup(c, chunk, args, inst_base, meta_in, fi);
    pin_scalar(total_dead);
    const HorizonView hz = horizon_view(c.base, args.horizon_off, args.chunks_per_inst);
    if (args.hz && tid == 0u) { hz.D(args.hz_parity ^ 1u)[c.j] = kHorizonNever; hz.BF(args.hz_parity ^ 1u)[c.j] = 0xffffffffu; }   // k_compact merges into these
    if (!has_rows) return;
    if (args.suffix_dead) return;   // (the job-table launch: this program's casualties are known to be its list's last rows, CompactArgs)
    if (total_dead == 0u) return;  // nothing died in this instance: the list stands
    if (args.hz_use) {   // can a row of this chunk have died in this frame? (see "death horizons")
        const double clock = *hz.clock;
        if (clock < u2d(hz.D(args.hz_parity)[c.j]) && args.frame_no - hz.BF(args.hz_parity)[c.j] <= kHorizonFrames) {
            const uint32_t rows_ = (c.n - c.start) < kChunk ? (c.n - c.start) : kChunk;
            if (tid < kChunk / 64u) {
                const unsigned long long m = rows_ >= (tid + 1u) * 64u ? ~0ull : (rows_ > tid * 64u ? ((1ull << (rows_ - tid * 64u)) - 1ull) : 0ull);
                (reinterpret_cast<unsigned long long*>(c.base + args.row_mask_off) + (size_t)c.j * (kChunk / 64u))[tid] = m;
            }
            if (tid == 0u) cb.counts[chunk] = rows_;
            return;
        }
    }
    const uint32_t* list = reinterpret_cast<const uint32_t*>(c.base + ((c.m.write_index & 1u) ? args.alive_off[1] : args.alive_off[0]));
    const uint32_t row0 = ring_row(list_head(c.m.write_index), c.start, args.capacity);   // ("Ring lists": the rows may stand behind a head)
    const uint32_t* died = reinterpret_cast<const uint32_t*>(c.base + args.died_bits_off);
    unsigned long long* rmask = reinterpret_cast<unsigned long long*>(c.base + args.row_mask_off) + (size_t)c.j * (kChunk / 64u);
    const uint32_t rows = (c.n - c.start) < kChunk ? (c.n - c.start) : kChunk;
    // All 16 rows of a lane are requested before anything is used, then all 16 bit words: the kernel is a chain of two dependent
    // memory accesses per row, so its speed is the number of them in flight.
    constexpr uint32_t kSteps = kWaveRows / 64u;
    uint32_t slot[kSteps], bits[kSteps];
#pragma unroll
    for (uint32_t s = 0; s < kSteps; ++s) {
        const uint32_t i = wave * kWaveRows + s * 64u + lane;
        slot[s] = i < rows ? ld_hint(list + ring_row(row0, i, args.capacity), args.stream_hint != 0u) : 0xffffffffu;
    }
#pragma unroll
    for (uint32_t s = 0; s < kSteps; ++s) bits[s] = slot[s] != 0xffffffffu ? died[slot[s] >> 5] : 0xffffffffu;   // (nontemporal loads: 1.8x slower; agent-scope atomic loads: the same, profiles/r03c_count_load.log)
    uint32_t wa = 0;
#pragma unroll
    for (uint32_t s = 0; s < kSteps; ++s) {
        const unsigned long long m = __ballot(((bits[s] >> (slot[s] & 31u)) & 1u) == 0u);   // (rows past the end: "died", they are nobody's)
        if (lane == 0u) rmask[wave * kSteps + s] = m;
        wa += (uint32_t)__popcll(m);
    }
    if (lane == 0u) s_wave[wave] = wa;
    __syncthreads();
    if (tid == 0u) {
        uint32_t t = 0;
#pragma unroll
        for (uint32_t w = 0; w < kBlock / 64; ++w) t += s_wave[w];
        cb.counts[chunk] = t;
    }
}
__global__ void __launch_bounds__(kBlock)
k_count_rows(const CompactArgs args, const uint64_t* __restrict__ inst_base, const DevMeta* __restrict__ meta_in,
             const DevFrameInst* __restrict__ fi, const CompactBufs cb) {
    count_rows_chunk(args, inst_base, meta_in, fi, cb, blockIdx.x, gridDim.x);
}

// The same two kernels for SEVERAL programs in one launch each. k_count_rows and k_compact are the same code for every program, only their
// arguments differ, and in a scene of many small effects the frame is bound by the number of launches the host can issue (26 example effects:
// 4 launches per effect, 12 us of host time per effect and frame). hnb_simulate therefore collects the programs that need their lists this
// frame into a job table (it travels with the frame's parameter upload) and serves them with two launches after all update kernels.
struct ListsJob {
    CompactArgs args;
    CompactBufs cb;
    const uint64_t* inst_base;
    const DevMeta* meta_in;
    DevMeta* meta_out;
    const DevFrameInst* fi;
    uint32_t first_wg, n_wg;   // this program's workgroups in the launch
    uint32_t pad[2];
};
__device__ __forceinline__ const ListsJob& job_of_workgroup(const ListsJob* __restrict__ jobs, uint32_t n_jobs) { return job_of_workgroup_t(jobs, n_jobs); }
__global__ void __launch_bounds__(kBlock)
k_count_rows_multi(const ListsJob* __restrict__ jobs, uint32_t n_jobs) {
    const ListsJob& jb = job_of_workgroup(jobs, n_jobs);
    if (jb.args.slot_order) return;   // its lists are rebuilt from the alive bytes by k_order_*
    count_rows_chunk(jb.args, jb.inst_base, jb.meta_in, jb.fi, jb.cb, blockIdx.x - jb.first_wg, jb.n_wg);
}
__global__ void __launch_bounds__(kBlock)
k_compact_multi(const ListsJob* __restrict__ jobs, uint32_t n_jobs) {
    const ListsJob& jb = job_of_workgroup(jobs, n_jobs);
    compact_chunk(jb.args, jb.inst_base, jb.meta_in, jb.meta_out, jb.fi, jb.cb, blockIdx.x - jb.first_wg, jb.n_wg);
}

// ---- a scene of many small effects: init and update of SEVERAL programs in one launch each ------------------------------------------------
// A game scene is dozens of different effects of a few thousand particles; each program's own (specialised) k_init and update kernel is a
// launch of a few microseconds of work, and the frame is bound by the number of launches the host can issue (26 example effects: 0.30 ms per
// frame, the device idle half of the time, profiles/r02u_scene.md). For programs that are small this frame hnb_simulate therefore fills job
// tables (they travel with the frame's parameter upload, like ListsJob) and serves ALL of them with one k_init_jobs launch and one
// k_update_jobs launch - the INTERPRETER instantiations, the only code that fits every program: 3-9x the instructions per particle of
// the specialised kernels, and irrelevant at these sizes. Same code paths as HNB_JIT=0, same results bit for bit (tests/test_scene_merge.py).
//
// SET MODULES (round 4; hnb_jit.h make_set_source): the interpreters' latency is the floor of such a scene - k_init_jobs 35 us, k_update_jobs
// 28 us for microseconds of work, every op a cold instruction-cache miss behind the previous one. For the SET of small programs a context
// holds, hiprtc compiles one module with two kernels of the same signatures - hnb_set_init / hnb_set_update - whose bodies are a switch over
// the job's `set_case` into the SPECIALISED instantiation of each program (the code its own launch would run). Same job tables, same
// launches; a launch is served by the set kernels when every job of it has a case in the loaded module, by the interpreters otherwise.
#endif  // HNB_JIT_TU (the job tables are shared with the set modules)
constexpr uint32_t kNoSetCase = 0xffffffffu;
struct ProgJob {             // k_init_jobs (first_wg / n_wg count init workgroups), k_update_jobs / k_update_generic_wide_jobs (... groups of 256 slots)
    DevProgram prog;
    const uint64_t* inst_base; const DevMeta* meta_in; const DevFrameInst* fi; const uint32_t* ublocks;
    CompactBufs cb;
    uint32_t write_died;
    uint32_t first_wg, n_wg;
    uint32_t set_case;       // the program's case in the context's set module (kNoSetCase: none)
};
struct StreamJob {           // k_update_jobs (first_wg / n_wg count chunks)
    SlotArgs args;
    const uint64_t* inst_base; const DevFrameInst* fi; const uint32_t* ublocks;
    CompactBufs cb;
    uint32_t first_wg, n_wg;
    uint32_t set_case, pad;
};
constexpr uint32_t kGenericSubs = kChunk / kBlock;   // k_update_jobs: one workgroup per 256 slots of a program on the V register file
#ifndef HNB_JIT_TU
template <class CODE>
__global__ void __launch_bounds__(kInitBlock)
k_init_jobs(const ProgJob* __restrict__ jobs, uint32_t n_jobs) {
    const ProgJob& jb = job_of_workgroup_t(jobs, n_jobs);
    init_workgroup<CODE>(jb.prog, jb.inst_base, jb.meta_in, jb.fi, jb.ublocks, blockIdx.x - jb.first_wg, jb.n_wg);
}
// One launch for every small program's update: workgroups [0, b0) serve the streaming jobs without age cohorts, [b0, b1) those with, the rest
// the programs on the V register file - there one workgroup per 256 slots (the latency of one 4096-slot chunk walked by a single workgroup
// WAS the frame: 0.13 ms for the 26-effect scene, profiles/r03u_scene_kernel_stats.csv). first_wg counts over the whole grid.
__global__ void __launch_bounds__(kBlock)
k_update_jobs(const StreamJob* __restrict__ sj0, uint32_t n0, const StreamJob* __restrict__ sj1, uint32_t n1, const ProgJob* __restrict__ pj, uint32_t np,
              uint32_t b0, uint32_t b1) {
    if (blockIdx.x < b0) {
        const StreamJob& jb = job_of_workgroup_t(sj0, n0);
        update_stream_chunk<ProgInterp, 0, false>(jb.args, jb.inst_base, jb.fi, jb.ublocks, jb.cb, blockIdx.x - jb.first_wg, jb.n_wg);
    } else if (blockIdx.x < b1) {
        const StreamJob& jb = job_of_workgroup_t(sj1, n1);
        update_stream_chunk<ProgInterp, 0, true>(jb.args, jb.inst_base, jb.fi, jb.ublocks, jb.cb, blockIdx.x - jb.first_wg, jb.n_wg);
    } else {
        const ProgJob& jb = job_of_workgroup_t(pj, np);
        const uint32_t w = blockIdx.x - jb.first_wg, sub = w % kGenericSubs;
        update_generic_chunk<InterpCode>(jb.prog, jb.inst_base, jb.fi, jb.ublocks, jb.cb, jb.write_died, w / kGenericSubs, jb.n_wg / kGenericSubs, sub, sub + 1u);
    }
}
// (programs on the wide register file: their own launch - the kernel needs scratch)
__global__ void __launch_bounds__(kBlock)
k_update_generic_wide_jobs(const ProgJob* __restrict__ jobs, uint32_t n_jobs) {
    const ProgJob& jb = job_of_workgroup_t(jobs, n_jobs);
    const uint32_t w = blockIdx.x - jb.first_wg, sub = w % kGenericSubs;
    update_generic_chunk<InterpCodeWide>(jb.prog, jb.inst_base, jb.fi, jb.ublocks, jb.cb, jb.write_died, w / kGenericSubs, jb.n_wg / kGenericSubs, sub, sub + 1u);
}

// Per 4096-row chunk of the alive list (as the update saw it): spawn events per channel, for the cross-chunk
// prefix of k_emit_events. Event counts are staged per SLOT by k_update_slots_generic.
__global__ void __launch_bounds__(kBlock)
k_emit_count(const DevProgram prog, const uint64_t* __restrict__ inst_base, const DevMeta* __restrict__ meta_in,
             const DevFrameInst* __restrict__ fi, const CompactBufs cb) {
    __shared__ uint32_t s_red[kBlock / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t chunk = chunk_of_workgroup(cb.xcd_remap);
    ChunkCtx c;
    if (!chunk_setup(c, chunk, prog, inst_base, meta_in, fi)) return;
    const uint32_t* list = reinterpret_cast<const uint32_t*>(c.base + prog.alive_off[list_column(c.m.write_index)]);   // (programs that emit events never keep a ring: head 0)
    const uint32_t rows = (c.n - c.start) < kChunk ? (c.n - c.start) : kChunk;
    // a thread's 16 rows are requested together, then their 16 counts per channel (row by row it was a chain of 32 dependent accesses per
    // channel: 20 us for the one chunk of the firework's rocket effect, profiles/r03zz_kernel_stats.csv)
    constexpr uint32_t kPer = kChunk / kBlock;
    uint32_t slot[kPer];
#pragma unroll
    for (uint32_t q = 0; q < kPer; ++q) { const uint32_t r = q * kBlock + tid; slot[q] = r < rows ? list[c.start + r] : 0xffffffffu; }
    for (uint32_t ch = 0; ch < prog.n_event_channels; ++ch) {
        const uint32_t* cnt = reinterpret_cast<const uint32_t*>(c.base + prog.ev_cnt_off[ch]);
        uint32_t g[kPer];
#pragma unroll
        for (uint32_t q = 0; q < kPer; ++q) g[q] = slot[q] != 0xffffffffu ? cnt[slot[q]] : 0u;
        uint32_t v = 0;
#pragma unroll
        for (uint32_t q = 0; q < kPer; ++q) v += g[q];
#pragma unroll
        for (uint32_t off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        __syncthreads();
        if (lane == 0) s_red[wave] = v;
        __syncthreads();
        if (tid == 0) cb.ev_totals[(size_t)chunk * HNB_MAX_EVENT_CHANNELS + ch] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    }
}
#endif

}  // namespace hnb
