// C-ABI runtime of the MI355X particle hot path (see include/hanabi_amd.h).
//
// Replaces the render-world resource management the hot path depends on
// (EffectCache slabs src/render/effect_cache.rs:232-356, metadata init src/render/mod.rs:6048-6070,
// per-frame spawner upload src/render/mod.rs:4437-4445,4679-4687) and the `simulate` dispatch
// sequence (src/render/mod.rs:6942-7613) with: one hipMalloc'd SoA slab per effect instance,
// device-resident double-buffered counters, one H2D copy + at most two kernel launches per
// program per frame. There is no CPU fallback: without a HIP device every entry point that
// needs one fails with HNB_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "hnb_kernels.hip.h"
#include "hnb_jit.h"
#include "hnb_sort.hip.h"
#include "hnb_comm.h"
#include "hnb_plan.h"

using namespace hnb;

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(e_ == hipErrorOutOfMemory ? HNB_ERR_OUT_OF_MEMORY : HNB_ERR_HIP, "%s failed: %s", #expr, \
                        hipGetErrorString(e_));                                                  \
    } while (0)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }


struct TimingPair { hipEvent_t a, b; const HnbProgram* prog; };
constexpr uint32_t kFrameRing = 4;
constexpr uint32_t kSceneMaxChunks = 16;       // a program is "small" this frame: <= 65,536 slots over all of its instances ...
constexpr uint32_t kSceneMaxInitBlocks = 64;   // ... and <= 16,384 spawns (hnb_simulate: merged launches)
#ifndef HNB_GENERIC_SPLIT_MAX
#define HNB_GENERIC_SPLIT_MAX 128
#endif
constexpr uint32_t kGenericSplitMaxChunks = HNB_GENERIC_SPLIT_MAX;  // a generic-update program of up to this many chunks (524,288 slots) launches one workgroup per 256 slots: 2048 workgroups, one round on 256 CUs
#ifndef HNB_STREAM_QUARTERS
#define HNB_STREAM_QUARTERS 1
#endif
constexpr uint32_t kSceneMaxCodeLen = 64;      // ... and a pass of at most this many instructions: the merged launches INTERPRET, and one long program (the
                                               // lightning bolt's 510-instruction init: 43 us interpreted, 5 us specialised) would set the latency of all

}  // namespace

// ---- streaming kernel instantiations ------------------------------------------------------------------
// Statically specialised op sequences for the common modifier stacks (the reference's example
// effects), plus the interpreter for every other streamable sequence.
typedef void (*StreamLaunchFn)(uint32_t grid, hipStream_t stream, const SlotArgs& sa, const uint64_t* inst_base, const DevFrameInst* fi,
                               const uint32_t* ublocks, const CompactBufs& cb);
#ifndef HNB_STREAM_WAVES_COHORT
#define HNB_STREAM_WAVES_COHORT 4   // 97 VGPRs, no scratch (a 5-wave budget: 96 + 8 bytes of scratch; A/B on one box, profiles/r03f_waves_ab.log: c2 the same, c2_mixed 0.215 vs 0.228 ms)
#endif
template <class PROG, int WAVES>
void launch_stream(uint32_t grid, hipStream_t stream, const SlotArgs& sa, const uint64_t* inst_base, const DevFrameInst* fi,
                   const uint32_t* ublocks, const CompactBufs& cb) {
    // (the age-cohort paths need more registers: budgeted for 6 waves (80 VGPRs) the per-particle path of the firework kernel spilled 48 bytes
    // per lane to scratch; see HNB_STREAM_WAVES_COHORT)
    if (sa.age_cohort) k_update_slots_stream<PROG, (WAVES > HNB_STREAM_WAVES_COHORT ? HNB_STREAM_WAVES_COHORT : WAVES), 0, true><<<grid, kBlock, 0, stream>>>(sa, inst_base, fi, ublocks, cb);
    else k_update_slots_stream<PROG, WAVES, 0, false><<<grid, kBlock, 0, stream>>>(sa, inst_base, fi, ublocks, cb);
}
#define OP_(x) (uint32_t)HNB_OP_M_##x
typedef ProgStatic<OP_(AGE_TICK)> ProgAge;                                                        // ribbon.rs
#ifndef HNB_AGE_KERNEL
#define HNB_AGE_KERNEL 1
#endif
// The update that is one AGE_TICK has a kernel of its own (k_update_slots_stream_age) for the case it is written for - lifetime culling, no cohorts, AGE
// loaded and stored, LIFETIME loaded -; every other combination (culling off: the parity gate's plain replay) runs the general instantiation.
void launch_stream_age(uint32_t grid, hipStream_t stream, const SlotArgs& sa, const uint64_t* inst_base, const DevFrameInst* fi,
                       const uint32_t* ublocks, const CompactBufs& cb) {
    if (HNB_AGE_KERNEL && !sa.age_cohort && sa.cull_lifetime && (sa.flags & 0xffu) == (4u | 8u | 64u))
        k_update_slots_stream_age<<<grid, kBlock, 0, stream>>>(sa, inst_base, fi, ublocks, cb);
    else launch_stream<ProgAge, 4>(grid, stream, sa, inst_base, fi, ublocks, cb);
}
typedef ProgStatic<OP_(AGE_TICK), OP_(EULER)> ProgAgeEuler;                                       // instancing.rs
typedef ProgStatic<OP_(AGE_TICK), OP_(VEL_ADD), OP_(EULER)> ProgAccel;                            // AccelModifier stacks
typedef ProgStatic<OP_(AGE_TICK), OP_(VEL_SCALE), OP_(EULER)> ProgDrag;
typedef ProgStatic<OP_(AGE_TICK), OP_(VEL_SCALE), OP_(VEL_ADD), OP_(EULER)> ProgDragAccel;        // firework.rs
typedef ProgStatic<OP_(AGE_TICK), OP_(VEL_ADD), OP_(VEL_SCALE), OP_(EULER)> ProgAccelDrag;
typedef ProgStatic<OP_(AGE_TICK), OP_(VEL_ADD), OP_(KILL_AABB), OP_(EULER)> ProgAccelKillAabb;    // activate.rs / spawn_on_command.rs
typedef ProgStatic<OP_(AGE_TICK), OP_(RADIAL_ACCEL), OP_(EULER)> ProgRadial;                      // instancing.rs (2nd asset)
typedef ProgStatic<OP_(AGE_TICK), OP_(CONFORM_SPHERE), OP_(CONFORM_SPHERE), OP_(KILL_AABB), OP_(KILL_SPHERE), OP_(EULER)> ProgForceField;  // force_field.rs
#undef OP_
#ifndef HNB_STREAM_WAVES_INTERP
#define HNB_STREAM_WAVES_INTERP 5
#endif

void select_stream_kernel(const Ins* code, uint32_t n, StreamLaunchFn* fn, const char** name) {
#define TRY_(PROG, WAVES) if (PROG::matches(code, n)) { *fn = &launch_stream<PROG, WAVES>; *name = #PROG; return; }
    TRY_(ProgNone, HNB_STREAM_WAVES)
    if (ProgAge::matches(code, n)) { *fn = &launch_stream_age; *name = "ProgAge"; return; }   // (its general instantiation: four waves per SIMD - the age prefetch of update_stream_chunk takes 16 more registers, and one scalar plane never needs more)
    TRY_(ProgAgeEuler, HNB_STREAM_WAVES)
    TRY_(ProgAccel, HNB_STREAM_WAVES)
    TRY_(ProgDrag, HNB_STREAM_WAVES)
    TRY_(ProgDragAccel, HNB_STREAM_WAVES)
    TRY_(ProgAccelDrag, HNB_STREAM_WAVES)
    TRY_(ProgAccelKillAabb, HNB_STREAM_WAVES)
    TRY_(ProgRadial, HNB_STREAM_WAVES)
    TRY_(ProgForceField, HNB_STREAM_WAVES_FULL)
#undef TRY_
    *fn = &launch_stream<ProgInterp, HNB_STREAM_WAVES_INTERP>;
    *name = "ProgInterp";
}

// What hnb_program_create fixes in a program (hnb_ctx_set_option before the program is created; hnb_jit_precompile uses the defaults)
struct ProgramOptions {
    uint32_t age_cohort = HNB_AGE_COHORT_AUTO;   // HNB_OPT_AGE_COHORT
    bool cull_lifetime = true;                   // HNB_OPT_CULL_LIFETIME
    bool horizon = true;                         // HNB_OPT_HORIZON
};

// A set module being compiled beside the frames (HNB_SET_MODULE_BACKGROUND). The job owns copies of everything the generated source is made from: the
// programs it was started for may be destroyed while hiprtc runs. `done` is the hand-over: the worker writes `res` / `ok`, then sets it (release);
// hnb_simulate reads it (acquire), joins and loads the module on its own thread (the device context is current there).
struct SetBuildJob {
    struct Member { std::vector<HnbAttrEntry> attrs; std::vector<Ins> init, update; bool streams = false, cohort = false; };
    std::vector<Member> members;
    jit::SetResult res;
    bool ok = false;
    std::atomic<bool> done{false};
    std::thread worker;
};

// Specialisation of single programs beside the frames (HNB_OPT_JIT_ASYNC): hnb_program_create returns with the ahead-of-time / interpreter kernels
// and queues the compilation; one worker thread per context compiles job after job; hnb_simulate installs what is ready into the programs that still
// exist (identified by a serial number: a program may be destroyed while its job waits or runs). Jobs own a copy of the program blob.
struct JitJob {
    uint64_t program_serial = 0;
    std::vector<uint8_t> blob;
    HnbProgramHeader hdr{};
    std::vector<HnbAttrEntry> attrs;
    bool streams = false, aot_static = false;
    ProgramOptions opt;
    jit::Result res;
    bool ok = false;
};
struct JitWorker {
    std::thread thread;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::unique_ptr<JitJob>> queue, done;
    bool stop = false;
    uint32_t in_flight = 0;   // queued or compiling
};

struct HnbContext {
    int device = 0;
    hipStream_t stream = nullptr;       // the stream hnb_simulate enqueues on: own_stream or the caller's (hnb_ctx_set_stream)
    hipStream_t own_stream = nullptr;   // created with the context, lives as long as it does
    hipStream_t upload_stream = nullptr;  // per-frame parameter uploads, overlapped with the previous frame's kernels
    hipStream_t side_stream = nullptr;    // update phase of the LIGHT programs of a frame, next to the heavy one's on `stream` (enqueue_update_passes)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool overlap_updates = true;          // HNB_OPT_OVERLAP_UPDATES
    bool stream_hints = true;             // HNB_OPT_STREAM_HINTS
    bool break_proof = false;             // HNB_OPT_TEST_BREAK_PROOF (a test hook: see stage_program_frame)
    // The per-frame parameters of EVERY program (instance rows, uniform blocks, init block starts) are staged in one pinned buffer and go to the
    // device with one copy per frame: with a copy per program, a scene of 26 small effects spent a quarter of its frame in 26 serialised
    // 3.5 us copy kernels and the host waiting for them (profiles/r02u_scene.md). A ring of slots: the host fills slot f % kFrameRing while the
    // device still reads the slots of the frames before.
    void* h_stage[kFrameRing] = {};
    void* d_stage[kFrameRing] = {};
    size_t stage_bytes = 0;
    // How a slot reaches the device (r6, HNB_OPT_DIRECT_UPLOAD): until round 6 hipMemcpyAsync on the upload stream and the HOST waiting for it - 10 us of every
    // frame's submission, and the submission of a small frame (C5, a scene of small effects) is as long as its kernels: such a frame was bound by both at
    // once, which is why neither a shorter kernel nor any of rounds 5 / 6's upload experiments (copy on the simulation stream, the stream waiting for the
    // copy's event, kernels reading the pinned block, a copy kernel) moved it. Now, where the device's memory is host-visible (a large PCIe BAR), the slots
    // are FINE-GRAINED device memory and the host writes the block into them itself - the way the runtime places kernel arguments: posted writes,
    // ordered in front of the launch's doorbell by a store fence - no copy, no wait, no extra launch. tools/probes/upload_probe.hip: a frame of three
    // dependent launches 20.8 us with copy + wait, 17.3 with an event wait on the stream, 13.8 reading pinned memory, 11.1 this way (profiles/r06w_upload_probe.log).
    bool large_bar = false;               // hipDeviceProp_t::isLargeBar
    bool direct_upload = true;            // HNB_OPT_DIRECT_UPLOAD
    bool stage_direct = false;            // the slots that exist now are host-written fine-grained device memory
    bool direct_failed = false;           // the allocation or its self-test failed once: copies from then on
    uint64_t direct_frames = 0, copied_frames = 0;   // statistics (hnb_program_kernel_info)
    uint64_t ring_waits = 0, ring_wait_ns = 0;       // frames in which hnb_simulate found the device still busy with the frame kFrameRing before (it is device-bound then), and for how long it waited
    hipEvent_t stage_done[kFrameRing] = {};  // recorded on the simulation stream after the frame that used the slot
    uint32_t num_cus = 256;
    std::vector<HnbProgram*> programs;
    HnbSimParams sim{};
    uint32_t frame = 0;         // simulated frames: parity double-buffers the spawn-event counters
    // Options (hnb_ctx_set_option; none of them is read from the environment). The first four are fixed in a program when it is created:
    uint32_t list_order = HNB_LIST_ORDER_SPAWN;
    ProgramOptions popt;        // age cohorts, lifetime culling, death horizons
    // ... the others are scheduling choices that apply from the next hnb_simulate on:
    bool skip_lists = true;     // skip the list kernels of frames the device's no-death bound covers (HNB_OPT_SKIP_LISTS)
    bool alternate = true;      // walk the chunks in alternating directions from frame to frame (HNB_OPT_ALTERNATE)
    bool suffix_proof = true;   // ribbon programs: "the casualties are the last rows of the sorted list" replaces k_count_rows where the host can prove it (HNB_OPT_SUFFIX_PROOF)
    uint32_t slot_init = 1;     // frames that spawn a large share of a program's slots run their init slot-major (k_init_slots; HNB_OPT_SLOT_INIT)
    bool ring_lists = true;     // ... and where in addition the spawns sort in front, the list is kept as a ring: nothing is rewritten (HNB_OPT_RING_LISTS)
    bool scene_merge = true;    // small programs share their init / update launches (ProgJob / StreamJob; HNB_OPT_SCENE_MERGE)
    bool transpose = true;      // vec3 planes of the per-particle update path through the wave's LDS transpose (HNB_OPT_TRANSPOSE)
    uint32_t timing = 0;        // 0 = off, n = time every n-th simulated frame
    uint32_t timing_tick = 0;
    std::vector<TimingPair> t_update, t_init, t_compact;
    std::vector<hipEvent_t> event_pool;  // timing events are recycled, never created on the frame path once the pool is warm
    uint32_t comm_refs = 0;     // HnbComm objects that hold this context: it cannot be destroyed before them
    // The set module of the context's small programs (hnb_jit.h, hnb_kernels.hip.h "SET MODULES"): HNB_OPT_SET_MODULE
    uint32_t set_mode = HNB_SET_MODULE_CACHED;
    struct SetModule {
        hipModule_t module = nullptr;
        hipFunction_t init = nullptr, update = nullptr;
        jit::SetPlan plan;
        uint32_t gen = 0;           // bumped with every module loaded: HnbProgram::set_gen / set_case are valid for one generation
    } set;
    bool jit_async = false;               // HNB_OPT_JIT_ASYNC
    std::unique_ptr<JitWorker> jit_worker;
    uint64_t next_program_serial = 1;
    std::shared_ptr<SetBuildJob> set_job;   // HNB_SET_MODULE_BACKGROUND: the compilation in flight (joined when its result is taken, or with the context)
    plan::SetLookupState set_lookup;   // the population the last lookup / build was made for, and the one the previous merged frame had (hnb_plan.h)
    uint32_t set_frames = 0;        // statistics: frames with a launch served by the set kernels
    uint32_t set_failed_builds = 0; // statistics: background compilations of a set module that failed (each population is tried once)
    std::string set_log;            // why the last build failed
};

struct HnbProgram {
    HnbContext* ctx = nullptr;
    HnbProgramHeader hdr{};
    std::vector<HnbAttrEntry> attrs;
    std::vector<HnbPropEntry> props;
    DevProgram dev{};
    Ins* d_code = nullptr;
    size_t slab_bytes = 0;
    bool wide_file = false;       // init_regs / update_regs above HNB_VM_MAX_REGS: generic kernels use the wide V file
    uint32_t cull_dt_operand = 0; // lifetime culling: decoded operand a of the update stream's AGE_TICK (dev.cull_lifetime)
    bool update_streams = false;  // update stream runs on the streaming kernel (macro ops, U operands)
    uint32_t age_cohort_mode = HNB_AGE_COHORT_AUTO;   // HNB_OPT_AGE_COHORT as it stood when the program was created (fixed in the program from then on)
    bool auto_materialise = false;  // HNB_AGE_COHORT_AUTO, the render modifiers read AGE and the effect is large: cohorts, and the update keeps the plane current (SlotArgs::age_current)
    StreamLaunchFn stream_launch = nullptr;  // specialised (or interpreted) streaming kernel for this update stream
    const char* stream_kernel_name = "";
    // kernels specialised for this program at creation (hnb_jit.h); null = the ahead-of-time kernels run
    hipModule_t jit_module = nullptr;
    hipFunction_t jit_init = nullptr, jit_update = nullptr, jit_init_slots = nullptr;
    bool slot_init_eligible = false;   // the init reads neither PARTICLE_COUNTER nor a parent particle, no ribbons: large spawns may run slot-major (plan::plan_slot_init)
    uint32_t slot_init_frames = 0;     // statistics
    uint32_t sort_skipped_frames = 0;  // statistics: list-free frames of a ribbon program that did not sort (enqueue_ribbon_sort)
    bool tick_sign_seen = false;       // sticky: some frame's AGE tick of some instance was negative, -0 or NaN (stage_program_frame; enqueue_ribbon_sort)
    uint64_t serial = 0;            // identity across destruction (HNB_OPT_JIT_ASYNC: a finished compilation looks its program up by it)
    bool jit_pending = false;
    std::string kernel_info, jit_log;
    // what a set module generates this program's cases from (narrow register file only; empty: never a member), and its case in the loaded module
    std::vector<Ins> h_init, h_update;
    std::string set_sig;
    uint64_t set_sig_hash = 0;
    uint32_t set_gen = 0, set_case = kNoSetCase, set_frames = 0;
    std::vector<Ins> uniform_code;  // evaluated on the host per instance per frame
    std::vector<HnbEffect*> effects;
    // device tables, sized for `table_cap` instances
    uint32_t table_cap = 0;
    uint64_t* d_inst_base = nullptr;
    DevMeta* d_meta[2] = {nullptr, nullptr};
    // Effect slabs are carved out of large blocks (one hipMalloc per ~1 GiB instead of one per instance):
    // thousands of instances stay within a few large mappings, and creating an instance does not hit the driver.
    struct SlabBlock { char* base = nullptr; uint32_t n_slots = 0; };
    std::vector<SlabBlock> slab_blocks;
    std::vector<void*> free_slabs;
    size_t slab_stride = 0;
    bool slot_order = false;              // HNB_LIST_ORDER_SLOT: lists rebuilt in slot order every frame
    bool has_ribbons = false;             // layout has RIBBON_ID: the alive list is sorted after every update (hnb_sort.hip.h)
    SortArgs sort{};                      // slab offsets of the sort scratch
    uint32_t* d_plane_by_attr = nullptr;  // [HNB_ATTR_COUNT] plane offsets by attribute id (children read parent particles through it)
    std::vector<uint32_t> parent_attrs;   // attribute ids the init stream reads from the parent particle
    uint32_t* d_ev_totals = nullptr;      // [table_cap * chunks_per_inst][HNB_MAX_EVENT_CHANNELS] (emitting programs)
    unsigned long long* h_ev_counts = nullptr;  // host-mapped [table_cap][HNB_MAX_EVENT_CHANNELS] {frame, event count} written by k_emit_events (emitting programs)
    uint32_t level = 0;                   // dependency level: parents are simulated before their children
    uint32_t* d_counts = nullptr;  // per chunk: survivors this frame
    uint32_t* d_deaths = nullptr;  // [2][table_cap]: casualties per instance, frame-parity double-buffered
    // ---- frame planning (hnb_plan.h): static facts, the history each proof carries, and the plan of the frame being enqueued ----
    plan::SkipFacts skip_facts;             // list-free frames: eligible = streamable, lifetime-culled, no kill modifier, no spawn events in or out
    plan::SkipHistory skip_hist;            // (skip_hist.dirty: something outside the frame inputs changed the particles since the last frame)
    plan::RibbonFacts ribbon_facts;         // what program creation established about AGE / RIBBON_ID / LIFETIME of a ribbon effect
    plan::RibbonHistory ribbon_hist;        // sticky violations, the one RIBBON_ID / lifetime value seen so far, the smallest tick
    plan::FramePlan plan;                   // decided once per hnb_simulate, read by every launch of the frame
    const char* d_frame_cur = nullptr;  // this frame's parameter block of the program inside the context's staging slot (HnbContext::d_stage)
    uint32_t ring = 0;          // slot of the next frame
    size_t frame_bytes = 0;
    uint32_t parity = 0;
    uint32_t frames_run = 0;    // frames this program was simulated in: the chunk walk alternates its direction with it
    uint32_t merged_frames = 0;             // statistics
    uint32_t unmerged_frames = 0;           // statistics: frames in which the program stayed out of the shared launches because the loaded set module does not know it
    bool horizon_eligible = false;          // streamable, lifetime-culled, no kill modifier, no ribbons: row-chunk death horizons are maintained (hnb_kernels.hip.h)
    uint32_t hz_parity = 0;                 // which half of the horizon arrays is current (flips in frames whose list kernels ran)
    uint32_t hz_frames = 0;                 // statistics: frames in which the horizons were in use
    uint32_t* d_safe = nullptr;             // u32[2][table_cap * chunks_per_inst] device words (allocated with the tables)
    unsigned long long* h_safe = nullptr;   // host-mapped {tag, bound bits}: what prove_skip_lists reads (no read-back)
    uint32_t* d_fault = nullptr;            // set by the kernel if a particle died in a frame whose lists were skipped
    uint32_t skipped_frames = 0;            // statistics: frames whose list kernels were skipped
    uint32_t sort_parity = 0;               // frames in which the sort ran (its state double buffer)
    uint32_t suffix_frames = 0;             // statistics
    uint32_t sort_rotated_frames = 0;       // statistics: frames whose ribbon sort was a rotation
    bool ring_live = false;                 // an instance's list may stand behind a non-zero head (a ring frame ran since the last rewrite)
    uint32_t ring_frames = 0;               // statistics: frames in which the list was kept as a ring (no row rewritten)
};

struct EventChannel {
    DevEventBuffer* buf = nullptr;  // device
    uint32_t capacity = 0;
    HnbEffect* child = nullptr;
};

struct HnbEffect {
    HnbProgram* prog = nullptr;
    uint32_t index = 0;
    void* slab = nullptr;
    bool simulated = true;          // false: frozen (SimulationCondition::WhenVisible while invisible, src/render/mod.rs:4347-4356)
    HnbEffect* parent = nullptr;    // EffectParent: init consumes the parent's spawn events
    uint32_t parent_channel = 0;
    EventChannel channels[HNB_MAX_EVENT_CHANNELS];  // as a parent
    uint32_t slot_base = 0;
    uint32_t spawn_count = 0;
    uint32_t seed = 0;
    float xf[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    std::vector<uint32_t> props;
};

static_assert(HNB_COMM_ID_BYTES == comm::kNcclUniqueIdBytes, "HNB_COMM_ID_BYTES is sizeof(ncclUniqueId)");
struct HnbComm {
    std::vector<HnbContext*> ctxs;               // the local contexts (one in rank mode)
    std::vector<comm::ncclComm_t> comms;         // one RCCL communicator per local context; empty: reduced through the host
    uint32_t n_ranks = 1, rank = 0;
    uint32_t scratch_cap = 0;                    // effects the device scratch below holds
    std::vector<uint64_t*> d_rows;               // per local context: DevMeta row addresses of the effects to report
    std::vector<unsigned long long*> d_counts, d_totals;
};

namespace {

hipEvent_t take_event(HnbContext* ctx) {
    if (!ctx->event_pool.empty()) { hipEvent_t e = ctx->event_pool.back(); ctx->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}
void recycle_timing_events(HnbContext* ctx) {
    for (auto& t : ctx->t_update) { ctx->event_pool.push_back(t.a); ctx->event_pool.push_back(t.b); }
    for (auto& t : ctx->t_init) { ctx->event_pool.push_back(t.a); ctx->event_pool.push_back(t.b); }
    for (auto& t : ctx->t_compact) ctx->event_pool.push_back(t.b);  // (t.a is the update's end event)
    ctx->t_compact.clear();
    ctx->t_update.clear();
    ctx->t_init.clear();
}

// Components of every HnbAttr (src/attributes.rs:549-675): the streaming kernels and the sort address the planes of
// POSITION / VELOCITY / AGE / LIFETIME / RIBBON_ID by these sizes, so the attribute table must agree with them.
constexpr uint8_t kAttrComponents[HNB_ATTR_COUNT] = {
    1, 1,              // ID, PARTICLE_COUNTER (pseudo, never stored)
    3, 3, 1, 1,        // POSITION, VELOCITY, AGE, LIFETIME
    1, 4, 1,           // COLOR, HDR_COLOR, ALPHA
    1, 2, 3,           // SIZE, SIZE2, SIZE3
    1, 1,              // PREV, NEXT
    3, 3, 3,           // AXIS_X, AXIS_Y, AXIS_Z
    1,                 // SPRITE_INDEX
    1, 1, 1, 1,        // F32_0..3
    2, 2, 2, 2,        // F32X2_0..3
    3, 3, 3, 3,        // F32X3_0..3
    4, 4, 4, 4,        // F32X4_0..3
    1, 1, 1, 1,        // U32_0..3
    1,                 // RIBBON_ID
};

// Operand span check: V operands must stay inside the V file, U operands inside the
// parameter block.
bool operand_ok(uint32_t operand, uint32_t span, uint32_t nregs, uint32_t n_uregs, bool ustream) {
    if (ustream) return operand + span <= n_uregs;
    if (operand & HNB_OPERAND_DECODED_U) return (operand & 0xffu) + span <= n_uregs;
    return operand + span <= nregs;
}

int validate_stream(const uint8_t* blob_base, const uint8_t* code, uint32_t len, uint32_t nregs, const HnbProgramHeader& h, int which) {
    const bool ustream = which == 0;
    const char* sname = which == 0 ? "uniform" : which == 1 ? "init" : "update";
    const uint32_t file_regs = ustream ? h.n_uregs : nregs;
    for (uint32_t i = 0; i < len; ++i) {
        uint32_t w[2];
        memcpy(w, code + (size_t)i * 8, 8);
        const uint32_t op = w[0] & 0xff, d = (w[0] >> 8) & 0xff, wd = ((w[1] >> 8) & 3u) + 1u;
        // same decode as vm_exec
        const uint32_t a = ustream ? (w[0] >> 16) & 0xff : HNB_OPERAND_DECODE((w[0] >> 16) & 0xff, w[1] >> 13);
        const uint32_t b = ustream ? w[0] >> 24 : HNB_OPERAND_DECODE(w[0] >> 24, w[1] >> 14);
        const uint32_t c = ustream ? w[1] & 0xff : HNB_OPERAND_DECODE(w[1] & 0xff, w[1] >> 15);
        const bool ba = (w[1] >> 10) & 1u, bb = (w[1] >> 11) & 1u, bc = (w[1] >> 12) & 1u;
#define BAD(msg) return fail(HNB_ERR_BAD_PROGRAM, "%s stream, instruction %u (op %u): %s", sname, i, op, msg)
        if (op >= HNB_OP_COUNT || op == HNB_OP_NOP) BAD("invalid opcode");
        if (op == HNB_OP_LOADK || op == HNB_OP_LDB || op == HNB_OP_LDP) {
            if (!ustream) BAD("uniform-only opcode in a per-particle stream");
            const uint32_t span = op == HNB_OP_LDP ? (a & 3u) + 1u : 1u;
            if (d + span > file_regs) BAD("destination out of range");
            if (op == HNB_OP_LDB && a > 5) BAD("bad sim-params field");
            if (op == HNB_OP_LDP && (uint64_t)w[1] + span > h.prop_words) BAD("property offset out of range");
            continue;
        }
        if (ustream && !(vm_op_is_elementwise(op) || (op >= HNB_OP_ALL && op <= HNB_OP_UNPACK4SNORM)))
            BAD("per-particle opcode in the uniform stream");
        const bool is_init = which == 1;
        if (!ustream && (d & HNB_OPERAND_U)) BAD("destination must be a V register");
        auto OK = [&](uint32_t operand, uint32_t span) { return operand_ok(operand, span, nregs, h.n_uregs, ustream); };
        if (vm_op_is_elementwise(op)) {
            const bool two = (op >= HNB_OP_FADD && op <= HNB_OP_FPOW) || (op >= HNB_OP_FLT && op <= HNB_OP_FGE) ||
                             (op >= HNB_OP_FMIX && op <= HNB_OP_FSMOOTH) || (op >= HNB_OP_IADD && op <= HNB_OP_UGE);
            const bool three = (op >= HNB_OP_FMIX && op <= HNB_OP_FSMOOTH) || op == HNB_OP_ICLAMP || op == HNB_OP_UCLAMP;
            if (d + wd > file_regs) BAD("destination out of range");
            if (!OK(a, ba ? 1 : wd)) BAD("operand a out of range");
            // b and c are read unconditionally by the interpreter: they must always be addressable
            if (!OK(b, (two && !bb) ? wd : 1)) BAD("operand b out of range");
            if (!OK(c, (three && !bc) ? wd : 1)) BAD("operand c out of range");
            continue;
        }
        switch (op) {
            case HNB_OP_LDID: case HNB_OP_LDPC: case HNB_OP_LDALIVE:
                if (d >= file_regs) BAD("destination out of range");
                break;
            case HNB_OP_LDA: case HNB_OP_STA: {
                const uint32_t ai = w[1] >> 16;
                if (ai >= h.n_attrs) BAD("attribute index out of range");
                HnbAttrEntry ae;
                memcpy(&ae, blob_base + h.attrs_off + ai * sizeof ae, sizeof ae);
                if (ae.reg != HNB_REG_NONE || ae.ncomp != wd) BAD("LDA/STA must address a whole non-pinned attribute");
                if (op == HNB_OP_LDA) { if (d + wd > file_regs) BAD("destination out of range"); }
                else if (!OK(a, ba ? 1 : wd)) BAD("operand out of range");
            } break;
            case HNB_OP_ALL: case HNB_OP_ANY: case HNB_OP_LENGTH:
                if (d >= file_regs || !OK(a, wd)) BAD("operand out of range");
                break;
            case HNB_OP_DOT: case HNB_OP_DISTANCE:
                if (d >= file_regs || !OK(a, wd) || !OK(b, wd)) BAD("operand out of range");
                break;
            case HNB_OP_NORMALIZE:
                if (d + wd > file_regs || !OK(a, wd)) BAD("operand out of range");
                break;
            case HNB_OP_CROSS:
                if (d + 3 > file_regs || !OK(a, 3) || !OK(b, 3)) BAD("operand out of range");
                break;
            case HNB_OP_PACK4UNORM: case HNB_OP_PACK4SNORM:
                if (d >= file_regs || !OK(a, 4)) BAD("operand out of range");
                break;
            case HNB_OP_UNPACK4UNORM: case HNB_OP_UNPACK4SNORM:
                if (d + 4 > file_regs || !OK(a, 1)) BAD("operand out of range");
                break;
            case HNB_OP_FRAND:
                if (d + wd > file_regs) BAD("destination out of range");
                break;
            case HNB_OP_RANDU: case HNB_OP_RANDN:
                if (d + wd > file_regs || !OK(a, ba ? 1 : wd) || !OK(b, bb ? 1 : wd)) BAD("operand out of range");
                break;
            case HNB_OP_ALIVE_SET: case HNB_OP_ALIVE_AND: case HNB_OP_KILL_IF:
            case HNB_OP_M_AGE_TICK: case HNB_OP_M_EULER: case HNB_OP_M_VEL_SCALE:
                if (!OK(a, 1)) BAD("operand out of range");
                break;
            case HNB_OP_M_VEL_ADD:
                if (!OK(a, 3)) BAD("operand out of range");
                break;
            case HNB_OP_M_PIN_SET:
                if (!(d == HNB_REG_POSITION || d == HNB_REG_VELOCITY || d == HNB_REG_AGE || d == HNB_REG_LIFETIME)) BAD("M_PIN_SET needs a pinned destination");
                if (!OK(a, (d == HNB_REG_POSITION || d == HNB_REG_VELOCITY) ? 3 : 1)) BAD("operand out of range");
                break;
            case HNB_OP_M_RADIAL_ACCEL: case HNB_OP_M_KILL_SPHERE: case HNB_OP_M_VEL_SPHERE:
                if (!OK(a, 3) || !OK(b, 1)) BAD("operand out of range");
                break;
            case HNB_OP_M_TANGENT_ACCEL:
                if (!OK(a, 3) || !OK(b, 3) || !OK(c, 1)) BAD("operand out of range");
                break;
            case HNB_OP_M_CONFORM_SPHERE:
                if (!OK(a, 9) || !OK(b, 1)) BAD("operand out of range");
                break;
            case HNB_OP_M_KILL_AABB:
                if (!OK(a, 3) || !OK(b, 3)) BAD("operand out of range");
                break;
            case HNB_OP_M_POS_CIRCLE: case HNB_OP_M_VEL_CIRCLE: case HNB_OP_M_VEL_TANGENT:
                if (!OK(a, 7)) BAD("operand out of range");
                break;
            case HNB_OP_M_POS_SPHERE:
                if (!OK(a, 4)) BAD("operand out of range");
                break;
            case HNB_OP_M_POS_CONE3D:
                if (!OK(a, 3)) BAD("operand out of range");
                break;
            case HNB_OP_M_ADD_XLATE: break;
            case HNB_OP_LDPARENT:
                if (!is_init) BAD("LDPARENT outside the init stream");
                if (d + wd > file_regs) BAD("destination out of range");
                {
                    const uint32_t id = w[1] >> 16;
                    if (id >= HNB_ATTR_COUNT || id < HNB_ATTR_POSITION) BAD("parent attribute id out of range");
                    if (wd != kAttrComponents[id]) BAD("LDPARENT must read a whole attribute");
                    bool listed = false;  // hnb_effect_set_parent checks the parent's layout against this list
                    for (uint32_t k = 0; k < h.parent_n_attrs && !listed; ++k) {
                        uint32_t pid;
                        memcpy(&pid, blob_base + h.parent_attrs_off + k * 4, 4);
                        listed = pid == id;
                    }
                    if (!listed) BAD("LDPARENT of an attribute that is not in the program's parent attribute list");
                }
                break;
            case HNB_OP_M_EMIT_EVENTS:
                if (is_init) BAD("M_EMIT_EVENTS outside the update stream");
                if (!OK(a, 1)) BAD("operand out of range");
                if (((w[1] >> 16) & 0xffu) >= h.n_event_channels) BAD("event channel out of range");
                break;
            default: BAD("unhandled opcode");
        }
#undef BAD
    }
    return HNB_OK;
}

// Slab layout of one effect instance: [alive list column 0][column 1][dead list][attribute planes...][alive byte per slot]
// [per-chunk lifetime bounds + "completely alive" flags][spawn-event staging][ribbon-sort scratch], 256-byte aligned sections.
// The alive list moves to the other column only in frames where particles died (k_compact). Every section offset is a
// u32 in the device structs: the layout is computed in 64 bits and rejected as a whole when it does not fit (offsets grow
// monotonically, so the total bounds every one of them).
bool layout_slab(const HnbProgramHeader& h, const HnbAttrEntry* attrs, DevProgram& d, SortArgs& so, size_t* out_bytes) {
    d.capacity = h.capacity;
    d.n_attrs = h.n_attrs;
    d.n_uregs = h.n_uregs;
    d.chunks_per_inst = (h.capacity + kChunk - 1) / kChunk;
    d.init_len = h.init_len;
    d.update_len = h.update_len;
    uint64_t off = 0;
    const uint64_t list_bytes = align_up((size_t)h.capacity * 4, 256);
    auto place = [&](uint64_t bytes) { const uint64_t at = off; off += bytes; return at; };
    const uint64_t a0 = place(list_bytes), a1 = place(list_bytes), dd = place(list_bytes);
    uint64_t plane[kMaxAttrs];
    for (uint32_t i = 0; i < h.n_attrs; ++i) plane[i] = place(align_up((size_t)h.capacity * attrs[i].ncomp * 4, 256));
    const uint64_t flag_off = place(align_up((size_t)h.capacity, 256));             // alive byte per slot, zeroed with the attribute planes
    const uint64_t lmin_off = place(align_up((size_t)d.chunks_per_inst * 16, 256));  // four u32 / f32 arrays per chunk: lifetime bound (0 = unknown), "completely alive" flag, age-cohort state, age-cohort value; zeroed too
    const uint64_t bit_bytes = align_up((size_t)d.chunks_per_inst * (kChunk / 8), 256);
    const uint64_t died_off = place(bit_bytes), rmask_off = place(bit_bytes);     // one bit per slot "died this frame", one bit per list row "survives" (k_count_rows / k_compact)
    const uint64_t hz_off = place(align_up(256 + (size_t)d.chunks_per_inst * 24, 256));   // death horizons: clock, D[2][chunks], BF[2][chunks] (zero = "may die")
    uint64_t ev_off[HNB_MAX_EVENT_CHANNELS] = {};
    for (uint32_t c = 0; c < h.n_event_channels; ++c) ev_off[c] = place(list_bytes);  // per-slot staging of spawn events (k_update_slots_generic -> k_emit_count / k_emit_events)
    uint64_t key_off[2] = {}, val_off[2] = {}, hist_off = 0, gsum_off = 0, bits_off = 0;
    const bool ribbons = (h.flags & HNB_PROG_HAS_RIBBONS) != 0;
    const uint32_t sort_chunks = (h.capacity + kSortTile - 1) / kSortTile;
    if (ribbons) {  // radix-sort scratch: 64-bit keys and values ping-pong, per-chunk digit histograms, key OR/AND
        for (int i = 0; i < 2; ++i) key_off[i] = place(align_up((size_t)h.capacity * 8, 256));
        for (int i = 0; i < 2; ++i) val_off[i] = place(list_bytes);
        hist_off = place(align_up((size_t)256 * sort_chunks * 4, 256));
        gsum_off = place(align_up((size_t)8 * ((sort_chunks + kSortGroup - 1) / kSortGroup) * 256 * 4, 256));
        bits_off = place(256);
    }
    if (off > ((uint64_t)0xffffffffu << 8)) return false;   // (offsets are kept in 256-byte units: 1 TiB)
    d.alive_off[0] = soff_of(a0); d.alive_off[1] = soff_of(a1); d.dead_off = soff_of(dd);
    for (uint32_t i = 0; i < h.n_attrs; ++i) {
        d.attrs[i].plane_off = soff_of(plane[i]);
        d.attrs[i].ncomp = attrs[i].ncomp;
        d.attrs[i].reg = attrs[i].reg;
        d.attrs[i].upd_flags = attrs[i].update_flags;
    }
    d.alive_flag_off = soff_of(flag_off);
    d.lmin_off = soff_of(lmin_off);
    d.died_bits_off = soff_of(died_off); d.row_mask_off = soff_of(rmask_off);
    d.horizon_off = soff_of(hz_off);
    d.n_event_channels = h.n_event_channels;
    for (uint32_t c = 0; c < h.n_event_channels; ++c) d.ev_cnt_off[c] = soff_of(ev_off[c]);
    if (ribbons) {
        so.capacity = h.capacity; so.chunks_per_inst = sort_chunks;
        so.alive_off[0] = d.alive_off[0]; so.alive_off[1] = d.alive_off[1];
        for (int i = 0; i < 2; ++i) { so.key_off[i] = soff_of(key_off[i]); so.val_off[i] = soff_of(val_off[i]); }
        so.hist_off = soff_of(hist_off); so.gsum_off = soff_of(gsum_off); so.bits_off = soff_of(bits_off);
        so.rid_plane.v = so.age_plane.v = kNoPlane;
        for (uint32_t i = 0; i < h.n_attrs; ++i) {
            if (attrs[i].attr == HNB_ATTR_RIBBON_ID) so.rid_plane = d.attrs[i].plane_off;
            if (attrs[i].attr == HNB_ATTR_AGE) so.age_plane = d.attrs[i].plane_off;
        }
    }
    *out_bytes = (size_t)off;
    return true;
}

int validate_blob(const void* blob, size_t size, HnbProgramHeader* out_hdr) {
    if (!blob || size < sizeof(HnbProgramHeader)) return fail(HNB_ERR_BAD_PROGRAM, "program blob too small");
    HnbProgramHeader h;
    memcpy(&h, blob, sizeof h);
    if (h.magic != HNB_PROGRAM_MAGIC) return fail(HNB_ERR_BAD_PROGRAM, "bad program magic 0x%08x", h.magic);
    if (h.version != HNB_PROGRAM_VERSION) return fail(HNB_ERR_BAD_PROGRAM, "unsupported program version %u", h.version);
    if (h.total_size != size) return fail(HNB_ERR_BAD_PROGRAM, "program size mismatch (%u vs %zu)", h.total_size, size);
    if (h.capacity == 0) return fail(HNB_ERR_BAD_PROGRAM, "capacity must be > 0");
    if (h.n_attrs == 0 || h.n_attrs > kMaxAttrs) return fail(HNB_ERR_BAD_PROGRAM, "invalid attribute count %u", h.n_attrs);
    auto in_range = [&](uint64_t off, uint64_t bytes) { return off <= size && bytes <= size - off; };
    if (!in_range(h.attrs_off, (uint64_t)h.n_attrs * sizeof(HnbAttrEntry)) ||
        !in_range(h.props_off, (uint64_t)h.n_props * sizeof(HnbPropEntry)) ||
        !in_range(h.uniform_off, (uint64_t)h.uniform_len * 8) || !in_range(h.init_off, (uint64_t)h.init_len * 8) ||
        !in_range(h.update_off, (uint64_t)h.update_len * 8) || !in_range(h.parent_attrs_off, (uint64_t)h.parent_n_attrs * 4))
        return fail(HNB_ERR_BAD_PROGRAM, "program section out of bounds");
    if (h.prop_words > 4ull * h.n_props) return fail(HNB_ERR_BAD_PROGRAM, "%u property words for %u properties", h.prop_words, h.n_props);
    if (h.n_event_channels > HNB_MAX_EVENT_CHANNELS)
        return fail(HNB_ERR_BAD_PROGRAM, "program appends to %u event channels, limit %u", h.n_event_channels, HNB_MAX_EVENT_CHANNELS);
    if (h.parent_n_attrs > HNB_ATTR_COUNT) return fail(HNB_ERR_BAD_PROGRAM, "invalid parent attribute count %u", h.parent_n_attrs);
    static_assert(HNB_ATTR_COUNT <= 64, "HnbProgramHeader::render_reads_* is a 64-bit mask");
    if ((((uint64_t)h.render_reads_hi << 32) | h.render_reads_lo) >> HNB_ATTR_COUNT) return fail(HNB_ERR_BAD_PROGRAM, "render_reads mask names attributes that do not exist");
    if ((h.uniform_off & 7) || (h.init_off & 7) || (h.update_off & 7)) return fail(HNB_ERR_BAD_PROGRAM, "code sections must be 8-byte aligned");
    if (h.init_regs > HNB_VM_MAX_REGS_WIDE || h.update_regs > HNB_VM_MAX_REGS_WIDE)
        return fail(HNB_ERR_BAD_PROGRAM, "program needs %u V registers, the VM has %u", std::max(h.init_regs, h.update_regs),
                    HNB_VM_MAX_REGS_WIDE);
    if (h.init_regs < HNB_REG_FIRST_FREE || h.update_regs < HNB_REG_FIRST_FREE)
        return fail(HNB_ERR_BAD_PROGRAM, "register counts must cover the pinned registers");
    if (h.n_uregs > HNB_VM_MAX_UREGS) return fail(HNB_ERR_BAD_PROGRAM, "program needs %u U registers, limit %u", h.n_uregs, HNB_VM_MAX_UREGS);
    const uint8_t* p = static_cast<const uint8_t*>(blob);
    bool has_position = false, has_age = false, has_ribbon_id = false;
    uint64_t seen = 0;
    for (uint32_t i = 0; i < h.n_attrs; ++i) {
        HnbAttrEntry a;
        memcpy(&a, p + h.attrs_off + i * sizeof a, sizeof a);
        if (a.attr >= HNB_ATTR_COUNT || a.attr < HNB_ATTR_POSITION) return fail(HNB_ERR_BAD_PROGRAM, "attribute %u is not storable", a.attr);
        if (a.ncomp != kAttrComponents[a.attr]) return fail(HNB_ERR_BAD_PROGRAM, "attribute %u has %u components, not %u", a.attr, a.ncomp, kAttrComponents[a.attr]);
        if (seen >> a.attr & 1u) return fail(HNB_ERR_BAD_PROGRAM, "attribute %u appears twice in the layout", a.attr);
        seen |= 1ull << a.attr;
        if (a.attr == HNB_ATTR_POSITION) has_position = true;
        if (a.attr == HNB_ATTR_AGE) has_age = true;
        if (a.attr == HNB_ATTR_RIBBON_ID) has_ribbon_id = true;
        const bool pinned = a.attr == HNB_ATTR_POSITION || a.attr == HNB_ATTR_VELOCITY || a.attr == HNB_ATTR_AGE ||
                            a.attr == HNB_ATTR_LIFETIME;
        const uint32_t want = a.attr == HNB_ATTR_POSITION ? HNB_REG_POSITION
                              : a.attr == HNB_ATTR_VELOCITY ? HNB_REG_VELOCITY
                              : a.attr == HNB_ATTR_AGE ? HNB_REG_AGE : HNB_REG_LIFETIME;
        if (pinned && a.reg != want) return fail(HNB_ERR_BAD_PROGRAM, "pinned attribute %u in register %u", a.attr, a.reg);
        if (!pinned && a.reg != HNB_REG_NONE) return fail(HNB_ERR_BAD_PROGRAM, "attribute %u must be a memory operand", a.attr);
    }
    // The POSITION attribute is mandatory (src/lib.rs:838-845).
    if (!has_position) return fail(HNB_ERR_BAD_PROGRAM, "the particle layout is missing the POSITION attribute");
    // Ribbons: the flag and the RIBBON_ID attribute go together, and AGE is mandatory (src/lib.rs:846-856).
    if (((h.flags & HNB_PROG_HAS_RIBBONS) != 0) != has_ribbon_id)
        return fail(HNB_ERR_BAD_PROGRAM, "HNB_PROG_HAS_RIBBONS must be set exactly when the layout has RIBBON_ID");
    if (has_ribbon_id && !has_age) return fail(HNB_ERR_BAD_PROGRAM, "a layout with RIBBON_ID needs the AGE attribute");
    if (((h.flags & HNB_PROG_READS_PARENT) != 0) != (h.parent_n_attrs != 0))
        return fail(HNB_ERR_BAD_PROGRAM, "HNB_PROG_READS_PARENT must be set exactly when parent attributes are listed");
    if (((h.flags & HNB_PROG_EMITS_EVENTS) != 0) != (h.n_event_channels != 0))
        return fail(HNB_ERR_BAD_PROGRAM, "HNB_PROG_EMITS_EVENTS must be set exactly when the program has event channels");
    for (uint32_t i = 0; i < h.parent_n_attrs; ++i) {
        uint32_t id;
        memcpy(&id, p + h.parent_attrs_off + i * 4, 4);
        if (id >= HNB_ATTR_COUNT || id < HNB_ATTR_POSITION) return fail(HNB_ERR_BAD_PROGRAM, "parent attribute %u is not storable", id);
    }
    for (uint32_t i = 0; i < h.n_props; ++i) {
        HnbPropEntry pe;
        memcpy(&pe, p + h.props_off + i * sizeof pe, sizeof pe);
        if (pe.ncomp < 1 || pe.ncomp > 4 || (uint64_t)pe.word_offset + pe.ncomp > h.prop_words) return fail(HNB_ERR_BAD_PROGRAM, "bad property entry %u", i);
        if (!memchr(pe.name, 0, sizeof pe.name)) return fail(HNB_ERR_BAD_PROGRAM, "unterminated property name %u", i);
    }
    int rc = validate_stream(p, p + h.uniform_off, h.uniform_len, 0, h, 0);
    if (rc != HNB_OK) return rc;
    rc = validate_stream(p, p + h.init_off, h.init_len, h.init_regs, h, 1);
    if (rc != HNB_OK) return rc;
    rc = validate_stream(p, p + h.update_off, h.update_len, h.update_regs, h, 2);
    if (rc != HNB_OK) return rc;
    {   // the instance slab is addressed with 32-bit section offsets in 256-byte units (1 TiB)
        std::vector<HnbAttrEntry> at(h.n_attrs);
        memcpy(at.data(), p + h.attrs_off, (size_t)h.n_attrs * sizeof(HnbAttrEntry));
        DevProgram d{};
        SortArgs so{};
        size_t bytes = 0;
        if (!layout_slab(h, at.data(), d, so, &bytes)) return fail(HNB_ERR_BAD_PROGRAM, "effect slab exceeds 1 TiB (capacity %u)", h.capacity);
    }
    if (out_hdr) *out_hdr = h;
    return HNB_OK;
}

void free_tables(HnbProgram* p) {
    hipFree(p->d_inst_base); p->d_inst_base = nullptr;
    for (int i = 0; i < 2; ++i) { hipFree(p->d_meta[i]); p->d_meta[i] = nullptr; }
    hipFree(p->d_counts); p->d_counts = nullptr;
    hipFree(p->d_deaths); p->d_deaths = nullptr;
    hipFree(p->d_ev_totals); p->d_ev_totals = nullptr;
    if (p->h_ev_counts) hipHostFree(p->h_ev_counts);
    p->h_ev_counts = nullptr;
    hipFree(p->d_safe); p->d_safe = nullptr;
    p->table_cap = 0;
}

size_t frame_bytes_for(const HnbProgram* p, uint32_t n) {
    return (size_t)n * sizeof(DevFrameInst) + (size_t)n * p->dev.n_uregs * 4 + (size_t)n * 4 + 16;  // + packed init_block_start[]
}

// Grow the per-program device tables to hold `need` instances, preserving state.
int ensure_tables(HnbProgram* p, uint32_t need) {
    if (need <= p->table_cap) return HNB_OK;
    HnbContext* ctx = p->ctx;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const uint32_t cap = std::max<uint32_t>(need, std::max<uint32_t>(4, p->table_cap * 2));
    uint64_t* nb = nullptr;
    DevMeta* nm[2] = {nullptr, nullptr};
    uint32_t* ns = nullptr;
    uint32_t* nd = nullptr;
    HIP_TRY(hipMalloc(&nb, (size_t)cap * 8));
    HIP_TRY(hipMemset(nb, 0, (size_t)cap * 8));
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipMalloc(&nm[i], (size_t)cap * sizeof(DevMeta)));
        HIP_TRY(hipMemset(nm[i], 0, (size_t)cap * sizeof(DevMeta)));
    }
    const size_t n_counts = (size_t)cap * p->dev.chunks_per_inst;
    HIP_TRY(hipMalloc(&ns, n_counts * 4));
    HIP_TRY(hipMemset(ns, 0, n_counts * 4));
    // casualty counters are zero between frames (k_compact re-arms them), so a fresh table is valid
    HIP_TRY(hipMalloc(&nd, (size_t)2 * cap * 4));
    HIP_TRY(hipMemset(nd, 0, (size_t)2 * cap * 4));
    if (p->table_cap) {
        HIP_TRY(hipMemcpy(nb, p->d_inst_base, (size_t)p->table_cap * 8, hipMemcpyDeviceToDevice));
        for (int i = 0; i < 2; ++i)
            HIP_TRY(hipMemcpy(nm[i], p->d_meta[i], (size_t)p->table_cap * sizeof(DevMeta), hipMemcpyDeviceToDevice));
    }
    hipFree(p->d_inst_base);
    hipFree(p->d_meta[0]);
    hipFree(p->d_meta[1]);
    hipFree(p->d_counts);
    hipFree(p->d_deaths);
    if (p->dev.n_event_channels) {
        hipFree(p->d_ev_totals);
        p->d_ev_totals = nullptr;
        HIP_TRY(hipMalloc(&p->d_ev_totals, n_counts * HNB_MAX_EVENT_CHANNELS * 4));
        HIP_TRY(hipMemset(p->d_ev_totals, 0, n_counts * HNB_MAX_EVENT_CHANNELS * 4));
        if (p->h_ev_counts) hipHostFree(p->h_ev_counts);
        p->h_ev_counts = nullptr;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&p->h_ev_counts), (size_t)cap * HNB_MAX_EVENT_CHANNELS * 8, hipHostMallocDefault));
        for (size_t i = 0; i < (size_t)cap * HNB_MAX_EVENT_CHANNELS; ++i) p->h_ev_counts[i] = 0xffffffffull;  // frame tag: none
    }
    p->d_inst_base = nb;
    p->d_meta[0] = nm[0];
    p->d_meta[1] = nm[1];
    p->d_counts = ns;
    p->d_deaths = nd;
    {   // no-death bounds per chunk and frame parity: start at +inf; growing the tables restarts them (effect creation marks the program dirty)
        hipFree(p->d_safe);
        p->d_safe = nullptr;
        HIP_TRY(hipMalloc(&p->d_safe, 2 * n_counts * 4));
        std::vector<uint32_t> inf(2 * n_counts, 0x7f800000u);
        HIP_TRY(hipMemcpy(p->d_safe, inf.data(), inf.size() * 4, hipMemcpyHostToDevice));
        p->skip_hist.dirty = true;
    }
    p->frame_bytes = frame_bytes_for(p, cap);
    p->table_cap = cap;
    return HNB_OK;
}

int find_attr(const HnbProgram* p, uint32_t attr) {
    for (size_t i = 0; i < p->attrs.size(); ++i)
        if (p->attrs[i].attr == attr) return (int)i;
    return -1;
}

int read_meta(HnbEffect* fx, DevMeta* out) {
    HnbProgram* p = fx->prog;
    HIP_TRY(hipSetDevice(p->ctx->device));   // (a host with one context per GPU: the calling thread's current device may be another one)
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    HIP_TRY(hipMemcpy(out, p->d_meta[p->parity] + fx->index, sizeof(DevMeta), hipMemcpyDeviceToHost));
    return HNB_OK;
}

// Streaming eligibility of an update stream: only macro ops on pinned registers, every operand in
// the parameter block, and no non-pinned attribute touched by the update program.
bool update_is_streamable(const uint8_t* b, const HnbProgramHeader& h, const HnbAttrEntry* attrs) {
    bool streams = true;
    for (uint32_t i = 0; i < h.update_len; ++i) {
        uint32_t w[2];
        memcpy(w, b + h.update_off + (size_t)i * 8, 8);
        const uint32_t op = w[0] & 0xffu, a = (w[0] >> 16) & 0xffu, bb = w[0] >> 24, c = w[1] & 0xffu;
        bool ok = vm_op_is_streamable(op) && (a & HNB_OPERAND_U);
        if (ok && (op == HNB_OP_M_RADIAL_ACCEL || op == HNB_OP_M_TANGENT_ACCEL || op == HNB_OP_M_CONFORM_SPHERE ||
                   op == HNB_OP_M_KILL_SPHERE || op == HNB_OP_M_KILL_AABB))
            ok = (bb & HNB_OPERAND_U) != 0;
        if (ok && op == HNB_OP_M_TANGENT_ACCEL) ok = (c & HNB_OPERAND_U) != 0;
        if (!ok) streams = false;
    }
    for (uint32_t i = 0; i < h.n_attrs; ++i)
        if (attrs[i].update_flags && attrs[i].reg == HNB_REG_NONE) streams = false;
    return streams;
}

// Lifetime culling (hnb_kernels.hip.h): the streaming update reads LIFETIME only for the AGE_TICK's `age < lifetime`.
// Eligible: the stream starts with the only AGE_TICK, which tests the lifetime; nothing else writes AGE or LIFETIME. The age
// cohorts rest on the same structure (nothing but that one AGE_TICK changes AGE) and on nobody else reading the AGE plane on
// the device (no ribbon sort).
bool cull_eligible(const uint8_t* b, const HnbProgramHeader& h, const HnbAttrEntry* attrs, bool streams, const ProgramOptions& opt, uint32_t* dt_operand) {
    if (!streams || h.update_len == 0 || !opt.cull_lifetime) return false;
    const Ins* uc = reinterpret_cast<const Ins*>(b + h.update_off);
    bool ok = (uc[0].x & 0xffu) == HNB_OP_M_AGE_TICK && ((uc[0].y >> 16) & 1u);
    for (uint32_t i = 1; i < h.update_len && ok; ++i) {
        const uint32_t op = uc[i].x & 0xffu, dst = (uc[i].x >> 8) & 0xffu;
        if (op == HNB_OP_M_AGE_TICK) ok = false;
        if (op == HNB_OP_M_PIN_SET && (dst == HNB_REG_AGE || dst == HNB_REG_LIFETIME)) ok = false;
    }
    bool loads_life = false, stores_life = false, has_age = false;
    for (uint32_t a = 0; a < h.n_attrs; ++a) {
        if (attrs[a].reg == HNB_REG_LIFETIME) { loads_life = (attrs[a].update_flags & HNB_ATTR_UPD_LOAD) != 0; stores_life = (attrs[a].update_flags & HNB_ATTR_UPD_STORE) != 0; }
        if (attrs[a].reg == HNB_REG_AGE) has_age = (attrs[a].update_flags & HNB_ATTR_UPD_LOAD) != 0;
    }
    if (!(ok && loads_life && !stores_life && has_age)) return false;
    if (dt_operand) *dt_operand = HNB_OPERAND_DECODE((uc[0].x >> 16) & 0xffu, uc[0].y >> 13);
    return true;
}
constexpr uint32_t kAutoMaterialiseMinSlots = kSceneMaxChunks * kChunk;   // HNB_AGE_COHORT_AUTO: from this capacity on (65,536: an instance no longer "small" for the merged launches) an AGE-reading asset keeps the cohorts and its update keeps the plane current
bool age_cohort_eligible(const uint8_t* b, const HnbProgramHeader& h, const HnbAttrEntry* attrs, bool streams, const ProgramOptions& opt) {
    if (!cull_eligible(b, h, attrs, streams, opt, nullptr) || (h.flags & HNB_PROG_HAS_RIBBONS) || opt.age_cohort == HNB_AGE_COHORT_OFF) return false;
    // HNB_AGE_COHORT_AUTO (the default) chooses from the asset: where the RENDER modifiers read AGE after every frame (HnbProgramHeader::
    // render_reads_*: ColorOverLifetime / SizeOverLifetime, src/modifier/output.rs:310-312) the plane must be current after every frame.
    // Effects of 65,536 slots or more per instance keep the cohorts and their update kernel writes the common age of a cohort chunk into the plane
    // as it goes (SlotArgs::age_current: write-only, 4 of the 8 bytes per particle the cohort saves; round 5 ran a k_materialise_age pass behind
    // every update instead - one more launch and 16 us per frame at 16.7M particles - and only from 2^20 slots on). Smaller ones - the programs
    // that share the merged launches of a scene - keep their ages in the plane: at those sizes the bytes do not matter and the cohort
    // instantiation only costs registers. Assets whose renderer does not read AGE get LEAN.
    if (opt.age_cohort == HNB_AGE_COHORT_AUTO && (h.render_reads_lo >> HNB_ATTR_AGE & 1u) && h.capacity < kAutoMaterialiseMinSlots) return false;
    // only the lean (bandwidth-bound) stacks: an update that is bound by VALU issue (ConformToSphere, Radial / TangentAccel: divisions, square
    // roots) gains nothing from 8 bytes less per particle and pays for the bookkeeping (force_field: 0.0955 -> 0.099 ms with it, measured)
    // (HNB_AGE_COHORT_ALL: every eligible stack, for A/B runs)
    if (opt.age_cohort == HNB_AGE_COHORT_ALL) return true;
    const Ins* uc = reinterpret_cast<const Ins*>(b + h.update_off);
    for (uint32_t i = 0; i < h.update_len; ++i)
        if (!vm_op_is_lean(uc[i].x & 0xffu)) return false;
    return true;
}

// Slot-major init of large spawns (hnb_kernels.hip.h): what a spawn writes must not depend on its rank among the frame's spawns
bool slot_init_eligible(const uint8_t* b, const HnbProgramHeader& h) {
    if (h.flags & (HNB_PROG_HAS_RIBBONS | HNB_PROG_READS_PARENT)) return false;
    const Ins* ic = reinterpret_cast<const Ins*>(b + h.init_off);
    for (uint32_t i = 0; i < h.init_len; ++i)
        if ((ic[i].x & 0xffu) == HNB_OP_LDPC || (ic[i].x & 0xffu) == HNB_OP_LDPARENT) return false;
    return true;
}

// What to specialise for a program (see hnb_jit.h). `aot_static`: a pre-built ProgStatic kernel matches.
jit::Request make_jit_request(const uint8_t* b, const HnbProgramHeader& h, const HnbAttrEntry* attrs, bool streams, bool aot_static, const ProgramOptions& opt) {
    jit::Request rq;
    rq.attrs = attrs; rq.n_attrs = h.n_attrs;
    rq.init = reinterpret_cast<const Ins*>(b + h.init_off); rq.init_len = h.init_len;
    rq.update = reinterpret_cast<const Ins*>(b + h.update_off); rq.update_len = h.update_len;
    rq.want_init = h.init_len > 0;
    rq.want_init_slots = slot_init_eligible(b, h);
    rq.wide_file = std::max(h.init_regs, h.update_regs) > HNB_VM_MAX_REGS;
    rq.want_update_stream = streams && !aot_static && h.update_len > 0;
    rq.want_update_generic = !streams && h.update_len > 0;
    bool lean = true;
    for (uint32_t i = 0; i < h.update_len; ++i) lean = lean && vm_op_is_lean(rq.update[i].x & 0xffu);
    rq.stream_waves = lean ? HNB_STREAM_WAVES : HNB_STREAM_WAVES_FULL;
    rq.stream_cohort = age_cohort_eligible(b, h, attrs, streams, opt);
    if (rq.stream_cohort && rq.stream_waves > HNB_STREAM_WAVES_COHORT) rq.stream_waves = HNB_STREAM_WAVES_COHORT;
    return rq;
}

// The specialised kernels of one program: load the code object, take the functions, say so in kernel_info.
void install_program_jit(HnbProgram* p, const jit::Result& res) {
    const bool aot_static = p->update_streams && strcmp(p->stream_kernel_name, "ProgInterp") != 0;
    hipModule_t mod = nullptr;
    hipFunction_t fi = nullptr, fu = nullptr;
    hipError_t je = hipModuleLoadData(&mod, res.code.data());
    if (je == hipSuccess && !res.init_name.empty()) je = hipModuleGetFunction(&fi, mod, res.init_name.c_str());
    hipFunction_t fs = nullptr;
    if (je == hipSuccess && !res.init_slots_name.empty()) je = hipModuleGetFunction(&fs, mod, res.init_slots_name.c_str());
    if (je == hipSuccess && !res.update_name.empty()) je = hipModuleGetFunction(&fu, mod, res.update_name.c_str());
    if (je != hipSuccess) {
        p->jit_log = std::string("loading the specialised code object failed: ") + hipGetErrorString(je);
        if (mod) hipModuleUnload(mod);
        (void)hipGetLastError();
        return;
    }
    p->jit_module = mod; p->jit_init = fi; p->jit_update = fu; p->jit_init_slots = fs;
    p->kernel_info = std::string(p->wide_file ? "wide-file " : "") + "init=" + (p->jit_init ? "jit" : (p->hdr.init_len ? "interp" : "none")) + " update=" +
                     (p->jit_update ? (p->update_streams ? "jit-stream" : "jit-generic")
                                    : (p->update_streams ? (aot_static ? std::string("aot-stream:") + p->stream_kernel_name : std::string("interp-stream"))
                                                         : std::string("interp-generic")));
    if (res.from_cache) p->kernel_info += " (jit cache hit)";
}

// HNB_OPT_JIT_ASYNC: the context's compilation thread (JitWorker)
void jit_worker_main(JitWorker* w) {
    for (;;) {
        std::unique_ptr<JitJob> job;
        {
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv.wait(lk, [&] { return w->stop || !w->queue.empty(); });
            if (w->stop) return;   // (jobs still queued are dropped with the context)
            job = std::move(w->queue.front());
            w->queue.pop_front();
        }
        const jit::Request rq = make_jit_request(job->blob.data(), job->hdr, job->attrs.data(), job->streams, job->aot_static, job->opt);
        job->ok = jit::build(rq, job->res);
        std::lock_guard<std::mutex> lk(w->mu);
        w->done.push_back(std::move(job));
    }
}
// ... and hnb_simulate's side of it: what has been compiled since the last frame goes into the programs that still exist. The kernels it
// replaces computed the same bits (tests run every program both ways), so the hand-over needs no synchronisation with frames in flight.
void install_finished_jit(HnbContext* ctx) {
    if (!ctx->jit_worker) return;
    std::deque<std::unique_ptr<JitJob>> done;
    {
        std::lock_guard<std::mutex> lk(ctx->jit_worker->mu);
        done.swap(ctx->jit_worker->done);
        ctx->jit_worker->in_flight -= (uint32_t)done.size();
    }
    for (std::unique_ptr<JitJob>& j : done) {
        HnbProgram* p = nullptr;
        for (HnbProgram* q : ctx->programs)
            if (q->serial == j->program_serial) p = q;
        if (!p) continue;   // destroyed in the meantime
        p->jit_pending = false;
        if (j->ok) install_program_jit(p, j->res);
        else if (!j->res.log.empty()) p->jit_log = j->res.log;
    }
}

// A program as a member of a set module: the cases of every pass, whether or not it has instructions (an init pass without code still
// zeroes the attributes and marks the slots; the family of the update - streaming plain / cohort, V register file - is the one
// plan_merged_launches sorts it into)
jit::Request make_set_request(const Ins* init, uint32_t init_len, const Ins* update, uint32_t update_len, const HnbAttrEntry* attrs, uint32_t n_attrs,
                              bool streams, bool cohort) {
    jit::Request rq;
    rq.attrs = attrs; rq.n_attrs = n_attrs;
    rq.init = init; rq.init_len = init_len; rq.update = update; rq.update_len = update_len;
    rq.want_init = true;
    rq.want_update_stream = streams;
    rq.want_update_generic = !streams;
    rq.stream_cohort = streams && cohort;
    rq.wide_file = false;
    return rq;
}

}  // namespace

extern "C" {

const char* hnb_last_error(void) { return g_last_error.c_str(); }
const char* hnb_version(void) { return "bevy_hanabi_amd 0.2.0 (gfx950)"; }

int hnb_ctx_create(int device_id, HnbContext** out_ctx) {
    if (!out_ctx) return fail(HNB_ERR_INVALID_ARG, "out_ctx is NULL");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(HNB_ERR_NO_DEVICE, "no HIP device available (the hanabi hot path has no CPU fallback)");
    if (device_id < 0 || device_id >= count) return fail(HNB_ERR_INVALID_ARG, "device %d out of range [0,%d)", device_id, count);
    HIP_TRY(hipSetDevice(device_id));
    HnbContext* ctx = new HnbContext();
    ctx->device = device_id;
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return fail(HNB_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e)); }
    ctx->own_stream = ctx->stream;
    e = hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { hipStreamDestroy(ctx->stream); delete ctx; return fail(HNB_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e)); }
    if (hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        ctx->overlap_updates = false;   // (the frame then runs on one stream, as it did before round 4)
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) {
        if (prop.multiProcessorCount > 0) ctx->num_cus = (uint32_t)prop.multiProcessorCount;
        ctx->large_bar = prop.isLargeBar != 0;
    }
    *out_ctx = ctx;
    return HNB_OK;
}

int hnb_ctx_destroy(HnbContext* ctx) {
    if (!ctx) return HNB_OK;
    if (ctx->comm_refs) return fail(HNB_ERR_INVALID_ARG, "the context is still part of %u communicator(s): call hnb_comm_destroy first", ctx->comm_refs);
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    while (!ctx->programs.empty()) hnb_program_destroy(ctx->programs.back());
    if (ctx->jit_worker) {   // (as below: a compilation in flight is waited for; queued ones are dropped)
        { std::lock_guard<std::mutex> lk(ctx->jit_worker->mu); ctx->jit_worker->stop = true; }
        ctx->jit_worker->cv.notify_all();
        ctx->jit_worker->thread.join();
        ctx->jit_worker.reset();
    }
    if (ctx->set_job) { ctx->set_job->worker.join(); ctx->set_job.reset(); }   // (hiprtc cannot be interrupted: destroying a context waits for a compilation it started)
    if (ctx->set.module) hipModuleUnload(ctx->set.module);
    recycle_timing_events(ctx);
    for (hipEvent_t e : ctx->event_pool) hipEventDestroy(e);
    if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
    if (ctx->upload_stream) hipStreamDestroy(ctx->upload_stream);
    if (ctx->side_stream) { hipStreamSynchronize(ctx->side_stream); hipStreamDestroy(ctx->side_stream); }
    if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
    for (uint32_t i = 0; i < kFrameRing; ++i) {
        hipFree(ctx->d_stage[i]);
        if (ctx->h_stage[i]) hipHostFree(ctx->h_stage[i]);
        if (ctx->stage_done[i]) hipEventDestroy(ctx->stage_done[i]);
    }
    delete ctx;
    return HNB_OK;
}

int hnb_ctx_set_stream(HnbContext* ctx, void* hip_stream) {
    if (!ctx) return fail(HNB_ERR_INVALID_ARG, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(ctx->stream));   // everything enqueued so far completes on the stream it was enqueued on
    ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return HNB_OK;
}

int hnb_ctx_set_option(HnbContext* ctx, uint32_t option, uint32_t value) {
    if (!ctx) return fail(HNB_ERR_INVALID_ARG, "ctx is NULL");
    switch (option) {
        case HNB_OPT_LIST_ORDER:
            if (value != HNB_LIST_ORDER_SPAWN && value != HNB_LIST_ORDER_SLOT) return fail(HNB_ERR_INVALID_ARG, "unknown list order %u", value);
            ctx->list_order = value;
            return HNB_OK;
        case HNB_OPT_ALTERNATE: ctx->alternate = value != 0u; return HNB_OK;
        case HNB_OPT_SKIP_LISTS: ctx->skip_lists = value != 0u; return HNB_OK;
        case HNB_OPT_AGE_COHORT:
            if (value > HNB_AGE_COHORT_AUTO) return fail(HNB_ERR_INVALID_ARG, "unknown age-cohort mode %u", value);
            ctx->popt.age_cohort = value;
            return HNB_OK;
        case HNB_OPT_CULL_LIFETIME: ctx->popt.cull_lifetime = value != 0u; return HNB_OK;
        case HNB_OPT_HORIZON: ctx->popt.horizon = value != 0u; return HNB_OK;
        case HNB_OPT_TEST_BREAK_PROOF: {   // a test hook that corrupts state on purpose: only for a process that says it is a test
            const char* e = getenv("HNB_ENABLE_TEST_HOOKS");
            if (value != 0u && !(e && e[0] == '1')) return fail(HNB_ERR_INVALID_ARG, "HNB_OPT_TEST_BREAK_PROOF is a test hook: refused unless HNB_ENABLE_TEST_HOOKS=1 is set in the environment");
            ctx->break_proof = value != 0u;
            return HNB_OK;
        }
        case HNB_OPT_TRANSPOSE: ctx->transpose = value != 0u; return HNB_OK;
        case HNB_OPT_SCENE_MERGE: ctx->scene_merge = value != 0u; return HNB_OK;
        case HNB_OPT_SUFFIX_PROOF: ctx->suffix_proof = value != 0u; return HNB_OK;
        case HNB_OPT_RING_LISTS: ctx->ring_lists = value != 0u; return HNB_OK;
        case HNB_OPT_SLOT_INIT:
            if (value > 2u) return fail(HNB_ERR_INVALID_ARG, "unknown slot-init mode %u", value);
            ctx->slot_init = value;
            return HNB_OK;
        case HNB_OPT_STREAM_HINTS: ctx->stream_hints = value != 0u; return HNB_OK;
        case HNB_OPT_DIRECT_UPLOAD: ctx->direct_upload = value != 0u; return HNB_OK;   // (ensure_stage re-creates the slots of the other kind before the next frame)
        case HNB_OPT_JIT_ASYNC: ctx->jit_async = value != 0u; return HNB_OK;
        case HNB_OPT_SET_MODULE:
            if (value > HNB_SET_MODULE_BACKGROUND) return fail(HNB_ERR_INVALID_ARG, "unknown set-module mode %u", value);
            ctx->set_mode = value;
            ctx->set_lookup = plan::SetLookupState();   // (look again: the mode decides whether a missing module is compiled)
            return HNB_OK;
        case HNB_OPT_OVERLAP_UPDATES: ctx->overlap_updates = value != 0u && ctx->side_stream && ctx->ev_fork && ctx->ev_join; return HNB_OK;   // (all three exist unless their creation failed)
        default: return fail(HNB_ERR_INVALID_ARG, "unknown option %u", option);
    }
}

int hnb_ctx_synchronize(HnbContext* ctx) {
    if (!ctx) return fail(HNB_ERR_INVALID_ARG, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return HNB_OK;
}

int hnb_program_validate(const void* blob, size_t blob_size) { return validate_blob(blob, blob_size, nullptr); }

int hnb_program_create(HnbContext* ctx, const void* blob, size_t blob_size, HnbProgram** out_prog) {
    if (!ctx || !out_prog) return fail(HNB_ERR_INVALID_ARG, "NULL argument");
    HnbProgramHeader h;
    int rc = validate_blob(blob, blob_size, &h);
    if (rc != HNB_OK) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    const uint8_t* b = static_cast<const uint8_t*>(blob);
    HnbProgram* p = new HnbProgram();
    p->ctx = ctx;
    p->hdr = h;
    p->attrs.resize(h.n_attrs);
    memcpy(p->attrs.data(), b + h.attrs_off, h.n_attrs * sizeof(HnbAttrEntry));
    p->props.resize(h.n_props);
    if (h.n_props) memcpy(p->props.data(), b + h.props_off, h.n_props * sizeof(HnbPropEntry));

    DevProgram& d = p->dev;
    size_t slab_bytes = 0;
    p->update_streams = update_is_streamable(b, h, p->attrs.data());
    if (!layout_slab(h, p->attrs.data(), d, p->sort, &slab_bytes)) { delete p; return fail(HNB_ERR_BAD_PROGRAM, "effect slab exceeds 1 TiB (capacity %u)", h.capacity); }
    p->slot_order = ctx->list_order == HNB_LIST_ORDER_SLOT && !(h.flags & HNB_PROG_HAS_RIBBONS);  // ribbons are re-sorted anyway
    p->has_ribbons = (h.flags & HNB_PROG_HAS_RIBBONS) != 0;
    p->slab_bytes = slab_bytes;
    p->parent_attrs.resize(h.parent_n_attrs);
    if (h.parent_n_attrs) memcpy(p->parent_attrs.data(), b + h.parent_attrs_off, (size_t)h.parent_n_attrs * 4);
    {
        uint32_t planes[HNB_ATTR_COUNT];
        for (uint32_t i = 0; i < HNB_ATTR_COUNT; ++i) planes[i] = kNoPlane;
        for (uint32_t i = 0; i < h.n_attrs; ++i) planes[p->attrs[i].attr] = d.attrs[i].plane_off.v;   // (256-byte units)
        hipError_t pe = hipMalloc(&p->d_plane_by_attr, sizeof planes);
        if (pe != hipSuccess) { delete p; return fail(HNB_ERR_OUT_OF_MEMORY, "hipMalloc failed: %s", hipGetErrorString(pe)); }
        hipMemcpy(p->d_plane_by_attr, planes, sizeof planes, hipMemcpyHostToDevice);
    }
    p->uniform_code.resize(h.uniform_len);
    if (h.uniform_len) memcpy(p->uniform_code.data(), b + h.uniform_off, (size_t)h.uniform_len * 8);
    p->wide_file = std::max(h.init_regs, h.update_regs) > HNB_VM_MAX_REGS;
    p->slot_init_eligible = slot_init_eligible(b, h);
    if (cull_eligible(b, h, p->attrs.data(), p->update_streams, ctx->popt, &p->cull_dt_operand)) {
        d.cull_lifetime = 1u;
        d.age_cohort = age_cohort_eligible(b, h, p->attrs.data(), p->update_streams, ctx->popt) ? 1u : 0u;
    }
    p->age_cohort_mode = ctx->popt.age_cohort;
    p->auto_materialise = ctx->popt.age_cohort == HNB_AGE_COHORT_AUTO && d.age_cohort != 0u && (h.render_reads_lo >> HNB_ATTR_AGE & 1u) != 0u;
    if (p->update_streams) select_stream_kernel(reinterpret_cast<const Ins*>(b + h.update_off), h.update_len, &p->stream_launch, &p->stream_kernel_name);
    if (!p->wide_file && jit::enabled()) {
        p->h_init.assign(reinterpret_cast<const Ins*>(b + h.init_off), reinterpret_cast<const Ins*>(b + h.init_off) + h.init_len);
        p->h_update.assign(reinterpret_cast<const Ins*>(b + h.update_off), reinterpret_cast<const Ins*>(b + h.update_off) + h.update_len);
        p->set_sig = jit::set_signature(make_set_request(p->h_init.data(), h.init_len, p->h_update.data(), h.update_len, p->attrs.data(), h.n_attrs, p->update_streams, d.age_cohort != 0u));
        p->set_sig_hash = jit::fnv1a(p->set_sig);
    }
    const bool aot_static = p->update_streams && strcmp(p->stream_kernel_name, "ProgInterp") != 0;
    p->kernel_info = std::string(p->wide_file ? "wide-file " : "") + "init=" + (h.init_len ? "interp" : "none") + " update=" +
                     (p->update_streams ? (aot_static ? std::string("aot-stream:") + p->stream_kernel_name : std::string("interp-stream")) : std::string("interp-generic"));
    p->serial = ctx->next_program_serial++;
    if (jit::enabled()) {
        const jit::Request rq = make_jit_request(b, h, p->attrs.data(), p->update_streams, aot_static, ctx->popt);
        jit::Result res;
        if (jit::build(rq, res, /*cache_only=*/ctx->jit_async)) {
            install_program_jit(p, res);
        } else if (ctx->jit_async && res.log.empty() && (rq.want_init || rq.want_update_generic || rq.want_update_stream)) {
            // not in the cache: the ahead-of-time / interpreter kernels run until the context's compilation thread has the program's own
            if (!ctx->jit_worker) {
                ctx->jit_worker.reset(new JitWorker());
                ctx->jit_worker->thread = std::thread(jit_worker_main, ctx->jit_worker.get());
            }
            std::unique_ptr<JitJob> job(new JitJob());
            job->program_serial = p->serial;
            job->blob.assign(b, b + blob_size);
            job->hdr = h; job->attrs = p->attrs; job->streams = p->update_streams; job->aot_static = aot_static; job->opt = ctx->popt;
            {
                std::lock_guard<std::mutex> lk(ctx->jit_worker->mu);
                ctx->jit_worker->queue.push_back(std::move(job));
                ctx->jit_worker->in_flight += 1;
            }
            ctx->jit_worker->cv.notify_one();
            p->jit_pending = true;
        } else if (!res.log.empty()) {
            p->jit_log = res.log;
        }
    }
    if (p->has_ribbons && p->update_streams) {  // ribbon sort: static part of "the head stays sorted" (see plan::RibbonFacts::provable)
        bool ok = true;
        uint32_t ticks = 0;
        const Ins* uc = reinterpret_cast<const Ins*>(b + h.update_off);
        for (uint32_t i = 0; i < h.update_len && ok; ++i) {
            const uint32_t op = uc[i].x & 0xffu, dst = (uc[i].x >> 8) & 0xffu;
            if (op == HNB_OP_M_AGE_TICK) { ticks += 1; p->ribbon_facts.tick_operand = HNB_OPERAND_DECODE((uc[i].x >> 16) & 0xffu, uc[i].y >> 13); ok = (p->ribbon_facts.tick_operand & HNB_OPERAND_DECODED_U) != 0; }
            if (op == HNB_OP_M_PIN_SET && dst == HNB_REG_AGE) ok = false;
        }
        ok = ok && ticks == 1;
        for (uint32_t a = 0; a < h.n_attrs; ++a)   // (a streamable update touches no non-pinned attribute, RIBBON_ID included)
            if (p->attrs[a].attr == HNB_ATTR_RIBBON_ID && (p->attrs[a].update_flags & HNB_ATTR_UPD_STORE)) ok = false;
        const Ins* ic = reinterpret_cast<const Ins*>(b + h.init_off);
        for (uint32_t i = 0; i < h.init_len && ok; ++i) {
            const uint32_t op = ic[i].x & 0xffu, dst = (ic[i].x >> 8) & 0xffu, wd = ((ic[i].y >> 8) & 3u) + 1u;
            if (op == HNB_OP_M_PIN_SET && dst == HNB_REG_AGE) {
                p->ribbon_facts.age_init_operand = HNB_OPERAND_DECODE((ic[i].x >> 16) & 0xffu, ic[i].y >> 13);
                p->ribbon_facts.age_init_set = true;
                ok = (p->ribbon_facts.age_init_operand & HNB_OPERAND_DECODED_U) != 0;   // a per-particle age could be negative: key order != age order
            } else if (op != HNB_OP_STA && op != HNB_OP_ALIVE_SET && op != HNB_OP_ALIVE_AND && op != HNB_OP_KILL_IF && !(op >= HNB_OP_M_AGE_TICK) &&
                       dst <= HNB_REG_AGE && dst + wd > HNB_REG_AGE) {
                ok = false;  // some other instruction writes the AGE register
            }
        }
        p->ribbon_facts.provable = ok;
        // the front proof (see plan::RibbonFacts::front_static): RIBBON_ID stored at most once by the init, from a uniform value
        int rid_index = -1;
        for (uint32_t a = 0; a < h.n_attrs; ++a) if (p->attrs[a].attr == HNB_ATTR_RIBBON_ID) rid_index = (int)a;
        uint32_t rid_stores = 0;
        bool front = ok && rid_index >= 0;
        for (uint32_t i = 0; i < h.init_len && front; ++i) {
            const uint32_t op = ic[i].x & 0xffu;
            if (op == HNB_OP_STA && (ic[i].y >> 16) == (uint32_t)rid_index) {
                rid_stores += 1;
                p->ribbon_facts.rid_operand = HNB_OPERAND_DECODE((ic[i].x >> 16) & 0xffu, ic[i].y >> 13);
                front = (p->ribbon_facts.rid_operand & HNB_OPERAND_DECODED_U) != 0;
            }
        }
        p->ribbon_facts.rid_set = rid_stores == 1;
        // ... and every spawn must still be in the list when it is sorted (the rotation moves exactly `spawned` rows): nothing but old age
        // kills, and the lifetime - one uniform value, compared with the tick per frame - outlasts the first frame
        uint32_t life_sets = 0;
        for (uint32_t i = 0; i < h.init_len && front; ++i) {
            const uint32_t op = ic[i].x & 0xffu, dst = (ic[i].x >> 8) & 0xffu, wd = ((ic[i].y >> 8) & 3u) + 1u;
            if (op == HNB_OP_M_PIN_SET && dst == HNB_REG_LIFETIME) {
                life_sets += 1;
                p->ribbon_facts.life_operand = HNB_OPERAND_DECODE((ic[i].x >> 16) & 0xffu, ic[i].y >> 13);
                front = (p->ribbon_facts.life_operand & HNB_OPERAND_DECODED_U) != 0;
            } else if (op != HNB_OP_STA && op != HNB_OP_ALIVE_SET && op != HNB_OP_ALIVE_AND && op != HNB_OP_KILL_IF && !(op >= HNB_OP_M_AGE_TICK) &&
                       dst <= HNB_REG_LIFETIME && dst + wd > HNB_REG_LIFETIME) {
                front = false;  // some other instruction writes the LIFETIME register
            }
        }
        for (uint32_t i = 0; i < h.update_len && front; ++i) {
            const uint32_t op = uc[i].x & 0xffu, dst = (uc[i].x >> 8) & 0xffu;
            if (op == HNB_OP_M_KILL_SPHERE || op == HNB_OP_M_KILL_AABB || op == HNB_OP_KILL_IF || op == HNB_OP_ALIVE_SET || op == HNB_OP_ALIVE_AND) front = false;
            if (op == HNB_OP_M_PIN_SET && dst == HNB_REG_LIFETIME) front = false;
        }
        p->ribbon_facts.front_static = front && rid_stores <= 1 && life_sets == 1;
    }
    {
        bool kills = false;
        const Ins* uc = reinterpret_cast<const Ins*>(b + h.update_off);
        for (uint32_t i = 0; i < h.update_len; ++i) {
            const uint32_t op = uc[i].x & 0xffu;
            kills = kills || op == HNB_OP_M_KILL_SPHERE || op == HNB_OP_M_KILL_AABB;
        }
        p->skip_facts.eligible = p->update_streams && d.cull_lifetime && !kills && h.n_event_channels == 0 && !(h.flags & HNB_PROG_READS_PARENT);
        p->skip_facts.dt_operand = p->cull_dt_operand;
        p->horizon_eligible = p->update_streams && d.cull_lifetime && !kills && !p->has_ribbons && !p->slot_order && ctx->popt.horizon;
        if (hipMalloc(&p->d_fault, 4) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void**>(&p->h_safe), 8, hipHostMallocDefault) != hipSuccess) {
            if (p->jit_module) hipModuleUnload(p->jit_module);
            hipFree(p->d_fault); hipFree(p->d_plane_by_attr);
            delete p;
            return fail(HNB_ERR_OUT_OF_MEMORY, "allocating the program's bookkeeping words failed");
        }
        hipMemset(p->d_fault, 0, 4);
        *p->h_safe = 0xffffffffull;  // tag = none
        p->dev.horizon = p->horizon_eligible ? 1u : 0u;
    }
    const size_t code_bytes = ((size_t)h.init_len + h.update_len) * 8;
    hipError_t e = hipMalloc(&p->d_code, std::max<size_t>(code_bytes, 8));
    if (e != hipSuccess) {
        if (p->jit_module) hipModuleUnload(p->jit_module);
        hipFree(p->d_plane_by_attr); hipFree(p->d_fault); hipHostFree(p->h_safe);
        delete p;
        return fail(HNB_ERR_OUT_OF_MEMORY, "hipMalloc(code) failed: %s", hipGetErrorString(e));
    }
    if (h.init_len) hipMemcpy(p->d_code, b + h.init_off, (size_t)h.init_len * 8, hipMemcpyHostToDevice);
    if (h.update_len) hipMemcpy(p->d_code + h.init_len, b + h.update_off, (size_t)h.update_len * 8, hipMemcpyHostToDevice);
    d.init_code = p->d_code;
    d.update_code = p->d_code + h.init_len;
    ctx->programs.push_back(p);
    *out_prog = p;
    return HNB_OK;
}

int hnb_program_destroy(HnbProgram* p) {
    if (!p) return HNB_OK;
    HnbContext* ctx = p->ctx;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    while (!p->effects.empty()) hnb_effect_destroy(p->effects.back());
    free_tables(p);
    if (p->jit_module) hipModuleUnload(p->jit_module);
    for (auto& blk : p->slab_blocks) hipFree(blk.base);
    hipFree(p->d_plane_by_attr);
    hipFree(p->d_code);
    hipFree(p->d_fault);
    if (p->h_safe) hipHostFree(p->h_safe);
    ctx->programs.erase(std::remove(ctx->programs.begin(), ctx->programs.end(), p), ctx->programs.end());
    delete p;
    return HNB_OK;
}

// Where a large block lands in physical memory decides how evenly its planes spread over the HBM stacks, and the
// driver's choice varies from one allocation to the next: on MI355X the same update kernel over the same 822 MB
// slab at the same virtual address measured 0.201, 0.223 or 0.231 ms depending on the allocation (stable for the
// life of the allocation; tools/bimodal_probe.py), and a plain pass that streams the planes the way the update does
// tells the placements apart (0.231 vs 0.241 ms). For blocks of kPlacementMinBytes or more, up to
// HNB_SLAB_CANDIDATES (default 24) allocations are tried, each timed with that pass; the search stops as soon as one
// candidate is 3 % faster than the median of those tried (two classes seen, the fast one in hand), the fastest is kept and the
// others are released. About one placement in eight is a fast one. Candidates cost ~3 ms each and transient
// memory (bounded by kPlacementMaxTransient), once per block; a failed candidate allocation just ends the search.
constexpr size_t kPlacementMinBytes = (size_t)256 << 20;
constexpr size_t kPlacementMaxTransient = (size_t)32 << 30;
hipError_t alloc_slab_block(HnbProgram* p, size_t bytes, char** out) {
    int want = 24;
    if (const char* e = getenv("HNB_SLAB_CANDIDATES")) want = std::max(1, std::min(32, atoi(e)));
    want = (int)std::min<size_t>((size_t)want, std::max<size_t>(1, kPlacementMaxTransient / std::max<size_t>(bytes, 1)));
    // Blocks shared by many small instances showed a single class (24 candidates of 1 GiB within 1 %): not searched.
    if (bytes < kPlacementMinBytes || bytes != p->slab_stride) want = 1;
    const bool debug = getenv("HNB_DEBUG_ALLOC") != nullptr;
    if (want == 1) return hipMalloc(reinterpret_cast<void**>(out), bytes);
    ProbeArgs pa{};
    pa.n_quads = (p->dev.capacity + 3u) / 4u;
    for (uint32_t a = 0; a < p->dev.n_attrs && pa.n_planes < 8; ++a) {  // the pinned attribute planes, as a streaming update touches them
        const DevAttr& at = p->dev.attrs[a];
        if (at.reg == HNB_REG_NONE) continue;
        pa.off[pa.n_planes] = at.plane_off;
        pa.stride16[pa.n_planes] = at.ncomp;   // ncomp * 4 B * 4 slots = ncomp 16-byte words per quad
        if (at.reg != HNB_REG_LIFETIME) pa.write_mask |= 1u << pa.n_planes;
        pa.n_planes += 1;
    }
    hipStream_t st = p->ctx->stream;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (hipEventCreate(&ev0) != hipSuccess || hipEventCreate(&ev1) != hipSuccess) {
        if (ev0) hipEventDestroy(ev0);
        (void)hipGetLastError();
        return hipMalloc(reinterpret_cast<void**>(out), bytes);
    }
    char* best = nullptr;
    float best_ms = 0.0f;
    std::vector<float> times;
    std::vector<char*> rejected;
    hipError_t first_error = hipSuccess;
    for (int c = 0; c < want; ++c) {
        char* cand = nullptr;
        const hipError_t e = hipMalloc(reinterpret_cast<void**>(&cand), bytes);
        if (e != hipSuccess) { if (!best) first_error = e; (void)hipGetLastError(); break; }
        hipMemsetAsync(cand, 0, bytes, st);  // touch every page before timing
        float ms = 1e30f;
        const uint32_t grid = (uint32_t)((pa.n_quads + 255u) / 256u);
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(ev0, st);
            k_probe_placement<<<grid, 256, 0, st>>>(cand, pa);
            hipEventRecord(ev1, st);
            hipEventSynchronize(ev1);
            float t = 0.0f;
            hipEventElapsedTime(&t, ev0, ev1);
            if (rep > 0 && t < ms) ms = t;  // the first pass warms the TLBs
        }
        if (debug) fprintf(stderr, "hanabi_amd: slab candidate %d at %p: %.4f ms\n", c, (void*)cand, ms);
        if (!best || ms < best_ms) { if (best) rejected.push_back(best); best = cand; best_ms = ms; }
        else rejected.push_back(cand);
        times.push_back(ms);
        if (times.size() >= 3) {  // two classes of placement seen and the fast one is in hand: 4 % apart, the noise within a class is ~1 %
            std::vector<float> sorted = times;
            std::sort(sorted.begin(), sorted.end());
            if (best_ms < 0.97f * sorted[sorted.size() / 2]) break;
        }
    }
    for (char* r : rejected) hipFree(r);
    hipEventDestroy(ev0);
    hipEventDestroy(ev1);
    if (!best) return first_error;
    *out = best;
    return hipSuccess;
}

int hnb_effect_create(HnbProgram* p, uint32_t slot_base, HnbEffect** out_fx) {
    if (!p || !out_fx) return fail(HNB_ERR_INVALID_ARG, "NULL argument");
    HnbContext* ctx = p->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t index = (uint32_t)p->effects.size();
    int rc = ensure_tables(p, index + 1);
    if (rc != HNB_OK) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HnbEffect* fx = new HnbEffect();
    fx->prog = p;
    fx->index = index;
    fx->slot_base = slot_base;
    hipError_t e = hipSuccess;
    if (p->free_slabs.empty()) {
        p->slab_stride = align_up(p->slab_bytes, 64 * 1024);
        const size_t want = std::max<size_t>(1, ((size_t)1 << 30) / p->slab_stride);
        // grow geometrically with the instance count, capped at ~1 GiB blocks
        const uint32_t n_slots = (uint32_t)std::min<size_t>(want, std::max<size_t>(1, p->effects.size()));
        HnbProgram::SlabBlock blk;
        blk.n_slots = n_slots;
        e = alloc_slab_block(p, p->slab_stride * n_slots, &blk.base);
        if (e == hipSuccess) {
            p->slab_blocks.push_back(blk);
            for (uint32_t i = n_slots; i-- > 0;) p->free_slabs.push_back(blk.base + (size_t)i * p->slab_stride);
        }
    }
    if (e == hipSuccess) { fx->slab = p->free_slabs.back(); p->free_slabs.pop_back(); }
    if (e == hipSuccess && getenv("HNB_DEBUG_ALLOC")) fprintf(stderr, "hanabi_amd: effect slab %p (%zu bytes)\n", fx->slab, p->slab_bytes);
    if (e != hipSuccess) { delete fx; return fail(HNB_ERR_OUT_OF_MEMORY, "hipMalloc(%zu bytes) for effect slab failed: %s", p->slab_bytes, hipGetErrorString(e)); }
    char* base = static_cast<char*>(fx->slab);
    const uint32_t cap = p->dev.capacity;
    k_reset_lists<<<(uint32_t)(((uint64_t)cap + 255u) / 256u), 256, 0, ctx->stream>>>(reinterpret_cast<uint32_t*>(base + p->dev.dead_off),
                                                              reinterpret_cast<uint32_t*>(base + p->dev.alive_off[0]),
                                                              reinterpret_cast<uint32_t*>(base + p->dev.alive_off[1]), cap);
    // Attribute planes start zeroed (the reference pre-fills with 0xFF only in debug builds).
    HIP_TRY(hipMemsetAsync(base + p->dev.attrs[0].plane_off, 0, p->slab_bytes - p->dev.attrs[0].plane_off, ctx->stream));
    if (p->horizon_eligible) {   // death horizons: no row yet, nobody can die (k_init takes minima into these); the clock starts at 0 (zeroed above)
        const size_t d_bytes = (size_t)p->dev.chunks_per_inst * 16;   // D[2][chunks]: kHorizonNever is the byte 0x7f repeated; BF[2][chunks]: 0xffffffff
        HIP_TRY(hipMemsetAsync(base + p->dev.horizon_off + 256, 0x7f, d_bytes, ctx->stream));
        HIP_TRY(hipMemsetAsync(base + p->dev.horizon_off + 256 + d_bytes, 0xff, (size_t)p->dev.chunks_per_inst * 8, ctx->stream));
    }
    if (p->has_ribbons) {  // {OR, AND} accumulators of the sort keys, both frame parities
        SortState st[2];
        for (SortState& z : st) { z.or_all = 0ull; z.and_all = ~0ull; z.or_tail = 0ull; z.and_tail = ~0ull; z.head_unsorted = 0u; z.pad[0] = z.pad[1] = z.pad[2] = 0u; }
        HIP_TRY(hipMemcpyAsync(base + p->sort.bits_off, st, sizeof st, hipMemcpyHostToDevice, ctx->stream));
    }
    // alive_count = 0, max_spawn = capacity, indirect_write_index = 0 (src/render/mod.rs:6048-6070)
    DevMeta m{};
    uint64_t slab_addr = reinterpret_cast<uint64_t>(fx->slab);
    HIP_TRY(hipMemcpyAsync(p->d_inst_base + index, &slab_addr, 8, hipMemcpyHostToDevice, ctx->stream));
    for (int i = 0; i < 2; ++i) HIP_TRY(hipMemcpyAsync(p->d_meta[i] + index, &m, sizeof m, hipMemcpyHostToDevice, ctx->stream));
    // an index that a destroyed instance used before still holds its last casualty count in one of the two rows
    for (int i = 0; i < 2; ++i) HIP_TRY(hipMemsetAsync(p->d_deaths + (size_t)i * p->table_cap + index, 0, 4, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    fx->props.assign(p->hdr.prop_words, 0u);
    for (const HnbPropEntry& pe : p->props)
        for (uint32_t c = 0; c < pe.ncomp; ++c) fx->props[pe.word_offset + c] = pe.default_bits[c];
    p->effects.push_back(fx);
    p->dev.n_inst = (uint32_t)p->effects.size();
    p->skip_hist.dirty = true;
    *out_fx = fx;
    return HNB_OK;
}

int hnb_effect_destroy(HnbEffect* fx) {
    if (!fx) return HNB_OK;
    HnbProgram* p = fx->prog;
    hipSetDevice(p->ctx->device);
    hipStreamSynchronize(p->ctx->stream);
    // unlink spawn-event relations in both directions
    if (fx->parent) {
        EventChannel& ch = fx->parent->channels[fx->parent_channel];
        if (ch.child == fx) { hipFree(ch.buf); ch = EventChannel(); }
    }
    for (EventChannel& ch : fx->channels) {
        if (ch.child) { ch.child->parent = nullptr; }
        hipFree(ch.buf);
        ch = EventChannel();
    }
    // swap-remove: move the last instance's table rows into the freed index
    const uint32_t last = (uint32_t)p->effects.size() - 1;
    if (p->h_ev_counts)   // the host copies of the event counts are indexed by instance: forget those of the two rows that change
        for (uint32_t c = 0; c < HNB_MAX_EVENT_CHANNELS; ++c) p->h_ev_counts[(size_t)fx->index * HNB_MAX_EVENT_CHANNELS + c] = p->h_ev_counts[(size_t)last * HNB_MAX_EVENT_CHANNELS + c] = 0xffffffffull;
    if (fx->index != last) {
        HnbEffect* moved = p->effects[last];
        hipMemcpy(p->d_inst_base + fx->index, p->d_inst_base + last, 8, hipMemcpyDeviceToDevice);
        for (int i = 0; i < 2; ++i) hipMemcpy(p->d_meta[i] + fx->index, p->d_meta[i] + last, sizeof(DevMeta), hipMemcpyDeviceToDevice);
        // casualty counters travel with the instance (the row of the coming frame is armed = 0, the other one is stale)
        for (int i = 0; i < 2; ++i) hipMemcpy(p->d_deaths + (size_t)i * p->table_cap + fx->index, p->d_deaths + (size_t)i * p->table_cap + last, 4, hipMemcpyDeviceToDevice);
        moved->index = fx->index;
        p->effects[fx->index] = moved;
    }
    p->effects.pop_back();
    p->dev.n_inst = (uint32_t)p->effects.size();
    p->skip_hist.dirty = true;
    p->free_slabs.push_back(fx->slab);  // the block itself is released with the program
    delete fx;
    return HNB_OK;
}

int hnb_effect_set_parent(HnbEffect* child, HnbEffect* parent, uint32_t channel, uint32_t event_capacity) {
    if (!child || !parent) return fail(HNB_ERR_INVALID_ARG, "NULL argument");
    if (child == parent) return fail(HNB_ERR_INVALID_ARG, "an effect cannot be its own parent");
    HnbProgram* cp = child->prog;
    HnbProgram* pp = parent->prog;
    if (cp->ctx != pp->ctx) return fail(HNB_ERR_INVALID_ARG, "parent and child must live in the same context (same GPU)");
    if (channel >= pp->dev.n_event_channels)
        return fail(HNB_ERR_INVALID_ARG, "the parent effect emits no spawn events on channel %u (EmitSpawnEventModifier::child_index)", channel);
    if (event_capacity == 0) return fail(HNB_ERR_INVALID_ARG, "event_capacity must be positive");
    for (HnbEffect* a = parent; a; a = a->parent)
        if (a == child) return fail(HNB_ERR_INVALID_ARG, "parent/child cycle");
    // A program's init pass is ONE launch for all its instances and parents must run before children, so the dependency
    // graph is checked at PROGRAM granularity: child->prog must not be reachable from parent->prog's ancestors (which
    // includes parent and child being instances of the same program).
    if (cp == pp) return fail(HNB_ERR_INVALID_ARG, "parent and child are instances of the same program: their init passes are one launch and cannot be ordered");
    {
        std::vector<HnbProgram*> stack{pp}, seen;
        while (!stack.empty()) {
            HnbProgram* q = stack.back();
            stack.pop_back();
            if (q == cp) return fail(HNB_ERR_INVALID_ARG, "parent/child cycle between programs: the parent's program already depends on the child's program");
            if (std::find(seen.begin(), seen.end(), q) != seen.end()) continue;
            seen.push_back(q);
            for (HnbEffect* e : q->effects)
                if (e->parent && e != child) stack.push_back(e->parent->prog);   // (child's current link is about to be replaced)
        }
    }
    // every attribute the child's init stream reads from the parent particle must exist in the parent layout
    for (uint32_t id : cp->parent_attrs)
        if (find_attr(pp, id) < 0) return fail(HNB_ERR_NOT_FOUND, "the parent layout has no attribute %u read by the child's init modifiers", id);
    HIP_TRY(hipSetDevice(cp->ctx->device));
    HIP_TRY(hipStreamSynchronize(cp->ctx->stream));
    if (child->parent) {  // re-parenting: release the old channel
        EventChannel& old = child->parent->channels[child->parent_channel];
        if (old.child == child) { hipFree(old.buf); old = EventChannel(); }
    }
    EventChannel& ch = parent->channels[channel];
    if (ch.child && ch.child != child) ch.child->parent = nullptr;  // the N-th child reads channel N: one consumer
    hipFree(ch.buf);
    const size_t bytes = sizeof(DevEventBuffer) + (size_t)event_capacity * 4;
    void* buf = nullptr;
    HIP_TRY(hipMalloc(&buf, bytes));
    HIP_TRY(hipMemset(buf, 0, bytes));
    HIP_TRY(hipMemcpy(static_cast<char*>(buf) + offsetof(DevEventBuffer, capacity), &event_capacity, 4, hipMemcpyHostToDevice));
    ch.buf = static_cast<DevEventBuffer*>(buf);
    ch.capacity = event_capacity;
    if (pp->h_ev_counts) pp->h_ev_counts[(size_t)parent->index * HNB_MAX_EVENT_CHANNELS + channel] = 0xffffffffull;  // a fresh (empty) buffer
    ch.child = child;
    child->parent = parent;
    child->parent_channel = channel;
    cp->skip_hist.dirty = true;
    // parents before children: dependency level per program. The program graph is acyclic (checked above), so the levels
    // settle within one pass per program; the bound is a backstop, never a hang.
    const size_t n_prog = cp->ctx->programs.size();
    for (HnbProgram* q : cp->ctx->programs) q->level = 0;
    for (size_t pass = 0; pass <= n_prog; ++pass) {
        bool changed = false;
        for (HnbProgram* q : cp->ctx->programs)
            for (HnbEffect* e : q->effects)
                if (e->parent && q->level <= e->parent->prog->level) { q->level = e->parent->prog->level + 1; changed = true; }
        if (!changed) break;
    }
    return HNB_OK;
}

int hnb_frame_begin(HnbContext* ctx, const HnbSimParams* params) {
    if (!ctx || !params) return fail(HNB_ERR_INVALID_ARG, "NULL argument");
    ctx->sim = *params;
    return HNB_OK;
}

int hnb_effect_set_frame(HnbEffect* fx, uint32_t spawn_count, uint32_t seed, const float* transform3x4) {
    if (!fx) return fail(HNB_ERR_INVALID_ARG, "fx is NULL");
    fx->spawn_count = spawn_count;
    fx->seed = seed;
    if (transform3x4) memcpy(fx->xf, transform3x4, sizeof fx->xf);
    return HNB_OK;
}

int hnb_program_set_frames(HnbProgram* prog, uint32_t first, uint32_t count, const uint32_t* spawn_counts, const uint32_t* seeds,
                           const float* transforms3x4) {
    if (!prog || !spawn_counts || !seeds) return fail(HNB_ERR_INVALID_ARG, "NULL argument");
    if ((size_t)first + count > prog->effects.size()) return fail(HNB_ERR_INVALID_ARG, "instance range [%u, %u) exceeds the %zu instances of the program", first, first + count, prog->effects.size());
    for (uint32_t i = 0; i < count; ++i) {
        HnbEffect* fx = prog->effects[first + i];
        fx->spawn_count = spawn_counts[i];
        fx->seed = seeds[i];
        if (transforms3x4) memcpy(fx->xf, transforms3x4 + (size_t)i * 12, sizeof fx->xf);
    }
    return HNB_OK;
}

int hnb_effect_index(HnbEffect* fx, uint32_t* out_index) {
    if (!fx || !out_index) return fail(HNB_ERR_INVALID_ARG, "NULL argument");
    *out_index = fx->index;
    return HNB_OK;
}

int hnb_effect_set_simulated(HnbEffect* fx, int simulated) {
    if (!fx) return fail(HNB_ERR_INVALID_ARG, "fx is NULL");
    if (fx->simulated != (simulated != 0)) fx->prog->skip_hist.dirty = true;  // a thawed instance ages again: older no-death bounds do not cover it
    fx->simulated = simulated != 0;
    return HNB_OK;
}

int hnb_effect_set_property(HnbEffect* fx, const char* name, const void* value, uint32_t n_words) {
    if (!fx || !name || !value) return fail(HNB_ERR_INVALID_ARG, "NULL argument");
    for (const HnbPropEntry& pe : fx->prog->props) {
        if (strncmp(pe.name, name, sizeof pe.name) == 0) {
            if (n_words != pe.ncomp) return fail(HNB_ERR_INVALID_ARG, "property '%s' has %u components, got %u", name, pe.ncomp, n_words);
            memcpy(&fx->props[pe.word_offset], value, (size_t)n_words * 4);
            return HNB_OK;
        }
    }
    return fail(HNB_ERR_NOT_FOUND, "unknown property '%s'", name);
}

static CompactBufs compact_bufs_of(const HnbContext* ctx, const HnbProgram* p, uint32_t n) {
    CompactBufs cb;
    cb.counts = p->d_counts;
    cb.deaths = p->d_deaths;
    cb.table_cap = p->table_cap;
    cb.parity = p->parity;
    cb.ev_totals = p->d_ev_totals;
    cb.xcd_remap = (n > 1 ? 1u : 0u) | (ctx->alternate && !(p->frames_run & 1u) ? 2u : 0u);  // see chunk_of_workgroup; the first frame walks DOWN: a burst's init wrote the planes upwards
    return cb;
}
// the streaming update's arguments for this frame (hnb_simulate: the program's own launch, or its row of the job table)
static SlotArgs slot_args_of(const HnbContext* ctx, const HnbProgram* p, uint32_t n, uint32_t write_died) {
    const uint32_t par = p->parity;
    SlotArgs sa{};
    sa.capacity = p->dev.capacity; sa.n_uregs = p->dev.n_uregs; sa.chunks_per_inst = p->dev.chunks_per_inst; sa.n_inst = n;
    sa.alive_flag_off = p->dev.alive_flag_off;
    sa.update_len = p->dev.update_len;
    sa.update_code = p->dev.update_code;
    sa.died_bits_off = p->dev.died_bits_off; sa.write_died = write_died;
    sa.cull_lifetime = p->dev.cull_lifetime; sa.lmin_off = p->dev.lmin_off; sa.dt_operand = p->cull_dt_operand;
    sa.age_cohort = p->dev.age_cohort;
    sa.quarters = 1u;
    sa.age_current = p->auto_materialise ? 1u : 0u;   // HNB_AGE_COHORT_AUTO: the render modifiers read AGE after every frame
    sa.frame_phase = p->frames_run & 15u;
    sa.horizon_off = p->dev.horizon_off; sa.horizon = p->horizon_eligible ? 1u : 0u;
    if (p->skip_facts.eligible) { sa.safe_words = p->d_safe; sa.safe_host = p->h_safe; sa.safe_parity = p->frames_run & 1u; sa.publish_tag = p->frames_run - 1u; sa.safe_stride = p->table_cap * p->dev.chunks_per_inst; }
    sa.skip_lists = p->plan.skip_lists ? 1u : 0u;
    sa.meta_in = p->d_meta[par]; sa.meta_out = p->d_meta[par ^ 1];
    sa.fault = p->d_fault;
    sa.transpose = ctx->transpose ? 1u : 0u;
    sa.stream_hint = p->plan.stream_hint ? 1u : 0u;
    sa.store_hint = p->plan.store_hint ? 1u : 0u;
    for (uint32_t a = 0; a < p->dev.n_attrs; ++a) {
        const DevAttr& at = p->dev.attrs[a];
        const int pi = at.reg == HNB_REG_POSITION ? 0 : at.reg == HNB_REG_VELOCITY ? 1 : at.reg == HNB_REG_AGE ? 2 : at.reg == HNB_REG_LIFETIME ? 3 : -1;
        if (pi < 0) continue;
        sa.plane_off[pi] = at.plane_off;
        if (at.upd_flags & HNB_ATTR_UPD_LOAD) sa.flags |= 1u << pi;
        if (at.upd_flags & HNB_ATTR_UPD_STORE) sa.flags |= 16u << pi;
    }
    return sa;
}

static CompactArgs compact_args_of(const HnbProgram* p) {
    CompactArgs ca{};
    ca.capacity = p->dev.capacity; ca.chunks_per_inst = p->dev.chunks_per_inst;
    ca.alive_off[0] = p->dev.alive_off[0]; ca.alive_off[1] = p->dev.alive_off[1]; ca.dead_off = p->dev.dead_off;
    ca.alive_flag_off = p->dev.alive_flag_off;
    ca.died_bits_off = p->dev.died_bits_off; ca.row_mask_off = p->dev.row_mask_off;
    ca.horizon_off = p->dev.horizon_off;
    ca.hz = p->horizon_eligible ? 1u : 0u; ca.hz_use = p->plan.hz_use ? 1u : 0u; ca.hz_parity = p->hz_parity; ca.frame_no = p->frames_run;
    ca.fault = p->d_fault;
    ca.slot_order = p->slot_order ? 1u : 0u;
    ca.suffix_dead = p->plan.ribbon.suffix ? 1u : 0u;
    ca.stream_hint = p->plan.stream_hint ? 1u : 0u;
    ca.rotate_front = p->plan.ribbon.rotate ? 1u : 0u;   // k_compact writes the survivors [spawns | older ones]: see CompactArgs (ribbon programs never have slot-ordered lists)
    ca.ring = p->plan.ribbon.ring ? 1u : 0u;
    ca.force_rewrite = (p->ring_live && !p->plan.ribbon.ring) ? 1u : 0u;
    return ca;
}

// ---- one simulated frame ---------------------------------------------------------------------------------------------------------------
// hnb_simulate = (1) make room in the staging ring, (2) per program: stage the frame's inputs and PLAN the frame (hnb_plan.h: every
// proof is a pure function, its result an immutable FramePlan), (3) fill the job tables of the launches several programs share,
// (4) one upload, (5) every init pass, parents first, (6) every update pass with its list maintenance, (7) advance the frame.
struct FrameJobs {           // the launches several programs share this frame (tables inside the frame's staging slot)
    struct Family { const void* d_jobs = nullptr; uint32_t n = 0, wgs = 0; };
    const ListsJob* d_lists = nullptr;
    uint32_t n_lists = 0, lists_wgs = 0;
    Family init[2], generic[2], stream[2];   // [wide register file] / [age cohorts]
    bool forked = false;                     // enqueue_init_passes: the side stream already waits behind the heavy program's init
    bool init_set = false, update_set = false;   // every job of init[0] / of the shared update launch has its case in the context's set module: hnb_set_init / hnb_set_update serve it
    const HnbProgram* heavy = nullptr;       // enqueue_update_passes: the one program whose update phase stays on the context's stream while
                                             // every other program's runs next to it on the side stream (null: one stream)
};

// The update phase of a frame is one independent chain per program - update -> spawn-event ordering -> lists -> ribbon sort touch only the
// program's own slabs and the event buffers it appends to, which nobody reads before the next frame's init passes. When one program is
// much heavier than the rest (a 16.7M-particle trail effect next to the rockets that spawn it: c2_events), the light chains - a dozen
// launches of microseconds each, 0.1 ms of dependent latency - hide behind the heavy update on a second stream.
constexpr uint32_t kOverlapMinChunks = 256;   // the heavy program: >= 1M slots ...
static const HnbProgram* pick_heavy_program(const HnbContext* ctx, const std::vector<HnbProgram*>& order, bool timed) {
    if (!ctx->overlap_updates || timed || order.size() < 2u) return nullptr;
    const HnbProgram* best = nullptr;
    uint64_t best_chunks = 0, others = 0;
    for (const HnbProgram* p : order) {
        const uint64_t c = (uint64_t)p->effects.size() * p->dev.chunks_per_inst;
        others += c;
        if (c > best_chunks) { best_chunks = c; best = p; }
    }
    others -= best_chunks;
    return (best_chunks >= kOverlapMinChunks && best_chunks >= 4u * others) ? best : nullptr;   // ... and at least 4x the rest together
}

// The host writes `n` bytes into host-visible device memory (HnbContext::stage_direct). The mapping is write-combining: the store fence drains the
// combining buffers, so that the writes are posted in front of whatever the caller does next - the doorbell of the frame's first launch.
static void direct_write(void* dst, const void* src, size_t n) {
    memcpy(dst, src, n);
#if !defined(__HIP_DEVICE_COMPILE__)
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_sfence();
#else
    __atomic_thread_fence(__ATOMIC_SEQ_CST);   // (a full barrier: orders the device writes in front of the doorbell store on hosts without a dedicated store fence)
#endif
#endif
}

// (1) The per-frame parameters of every program are filled into the context's next ring slot and uploaded on the upload stream. The host
// waits for the (tiny) copy itself, so the simulation stream carries no cross-stream wait: such a wait costs an ~11 us bubble in front
// of every frame's first kernel (measured), the host has ~200 us of slack per frame.
static int ensure_stage(HnbContext* ctx, const std::vector<HnbProgram*>& order, uint32_t slot) {
    size_t need = 0;
    for (const HnbProgram* p : order) need += (frame_bytes_for(p, (uint32_t)p->effects.size()) + 255u) & ~(size_t)255u;
    need += order.size() * sizeof(ListsJob) + 256u;   // the job table of the multi-program list launches
    need += order.size() * (2u * sizeof(ProgJob) + sizeof(StreamJob)) + 6u * 256u;   // ... and of the merged init / update launches of small programs
    const bool want_direct = ctx->direct_upload && ctx->large_bar && !ctx->direct_failed;
    if (need > ctx->stage_bytes || (ctx->stage_bytes && want_direct != ctx->stage_direct)) {  // grows rarely (a new program, more instances): nothing may still be reading the old buffers
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        const size_t nb = std::max<size_t>(std::max<size_t>(2 * need, ctx->stage_bytes), 64u << 10);
        for (uint32_t i = 0; i < kFrameRing; ++i) {
            hipFree(ctx->d_stage[i]); ctx->d_stage[i] = nullptr;
            if (ctx->h_stage[i]) hipHostFree(ctx->h_stage[i]);
            ctx->h_stage[i] = nullptr;
        }
        ctx->stage_bytes = 0;
        ctx->stage_direct = false;
        bool direct = want_direct;
        for (uint32_t i = 0; i < kFrameRing && direct; ++i) {
            // fine-grained device memory: host-visible through the BAR, coherent for the kernels that read it. Checked once per slot: what the host writes
            // must be what a device-side copy reads back (a device that reports a large BAR without honouring it would fault or differ here, not in a frame)
            if (hipExtMallocWithFlags(&ctx->d_stage[i], nb, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); ctx->d_stage[i] = nullptr; direct = false; break; }
            uint64_t probe[8], back[8] = {};
            for (int q = 0; q < 8; ++q) probe[q] = 0x9e3779b97f4a7c15ull * (uint64_t)(q + 1 + (int)i);
            direct_write(ctx->d_stage[i], probe, sizeof probe);
            if (hipMemcpy(back, ctx->d_stage[i], sizeof back, hipMemcpyDeviceToHost) != hipSuccess || memcmp(back, probe, sizeof probe) != 0) { (void)hipGetLastError(); direct = false; }
        }
        if (want_direct && !direct) {   // not on this device after all: ordinary slots, copies (never tried again in this context)
            ctx->direct_failed = true;
            for (uint32_t i = 0; i < kFrameRing; ++i) { hipFree(ctx->d_stage[i]); ctx->d_stage[i] = nullptr; }
        }
        for (uint32_t i = 0; i < kFrameRing; ++i) {
            if (!direct) HIP_TRY(hipMalloc(&ctx->d_stage[i], nb));
            HIP_TRY(hipHostMalloc(&ctx->h_stage[i], nb, hipHostMallocDefault));
            if (!ctx->stage_done[i]) HIP_TRY(hipEventCreateWithFlags(&ctx->stage_done[i], hipEventDisableTiming));
        }
        ctx->stage_bytes = nb;
        ctx->stage_direct = direct;
    }
    if (!order.empty()) {   // the frame that last used this slot (kFrameRing frames ago): the one place where hnb_simulate waits for the device
        if (hipEventQuery(ctx->stage_done[slot]) != hipSuccess) {
            (void)hipGetLastError();
            const auto t0 = std::chrono::steady_clock::now();
            HIP_TRY(hipEventSynchronize(ctx->stage_done[slot]));
            ctx->ring_wait_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
            ctx->ring_waits += 1;
        }
    }
    return HNB_OK;
}

// (2) One program's inputs of the frame (DevFrameInst rows, parameter blocks, init grid) into the staging slot, and the frame's plan.
static void stage_program_frame(HnbContext* ctx, HnbProgram* p, uint32_t slot, size_t& stage_off, std::vector<plan::InstanceFrame>& inst_frames) {
    const uint32_t ev_parity = ctx->frame & 1u;
    const uint32_t n = (uint32_t)p->effects.size();
    char* h = static_cast<char*>(ctx->h_stage[slot]) + stage_off;
    p->d_frame_cur = static_cast<const char*>(ctx->d_stage[slot]) + stage_off;
    DevFrameInst* fi = reinterpret_cast<DevFrameInst*>(h);
    uint32_t* ublocks = reinterpret_cast<uint32_t*>(h + (size_t)n * sizeof(DevFrameInst));
    const float sim[6] = {ctx->sim.time, ctx->sim.delta_time, ctx->sim.virtual_time, ctx->sim.virtual_delta_time,
                          ctx->sim.real_time, ctx->sim.real_delta_time};
    const uint32_t nu = p->dev.n_uregs;
    uint32_t blocks = 0;
    uint64_t cpu_spawns = 0;
    for (uint32_t i = 0; i < n; ++i) if (p->effects[i]->simulated && !p->effects[i]->parent) cpu_spawns += p->effects[i]->spawn_count;
    const bool big_burst = cpu_spawns >= (1ull << 20);
    inst_frames.assign(n, plan::InstanceFrame());
    for (uint32_t i = 0; i < n; ++i) {
        HnbEffect* fx = p->effects[i];
        memset(&fi[i], 0, sizeof fi[i]);
        fi[i].spawn_count = fx->parent ? 0u : fx->spawn_count;  // the CPU spawner of a child effect is unused (firework.rs:161)
        fi[i].seed = fx->seed;
        fi[i].slot_base = fx->slot_base;
        fi[i].init_block_start = blocks;
        fi[i].ev_parity = ev_parity;
        fi[i].skip = fx->simulated ? 0u : 1u;
        plan::InitGridInputs gi;
        gi.simulated = fx->simulated; gi.has_parent = fx->parent != nullptr; gi.spawn_count = fx->spawn_count;
        if (fx->parent && fx->simulated) {
            const EventChannel& ch = fx->parent->channels[fx->parent_channel];
            fi[i].parent_base = reinterpret_cast<uint64_t>(fx->parent->slab);
            fi[i].parent_planes = reinterpret_cast<uint64_t>(fx->parent->prog->d_plane_by_attr);
            fi[i].ev_in = reinterpret_cast<uint64_t>(ch.buf);
            gi.event_capacity = ch.capacity;  // the event count lives on the device: the grid is sized for the worst case ...
            // ... unless last frame's count has already arrived in host memory (k_emit_events writes {frame, count} there): plan::size_init_grid
            const HnbProgram* pp = fx->parent->prog;
            if (pp->h_ev_counts && ctx->frame > 0u) {
                const unsigned long long v = *reinterpret_cast<volatile unsigned long long*>(pp->h_ev_counts + (size_t)fx->parent->index * HNB_MAX_EVENT_CHANNELS + fx->parent_channel);
                gi.events_known = (uint32_t)v == ctx->frame - 1u;
                gi.known_events = (uint32_t)(v >> 32);
            }
        }
        for (uint32_t c = 0; c < HNB_MAX_EVENT_CHANNELS; ++c) fi[i].ev_out[c] = reinterpret_cast<uint64_t>(fx->channels[c].buf);
        blocks += plan::size_init_grid(gi, p->dev.capacity, kInitBlock, kInitRounds, big_burst, ctx->num_cus);
        plan::InstanceFrame& inf = inst_frames[i];
        inf.simulated = fx->simulated; inf.has_parent = fx->parent != nullptr; inf.spawn_count = fx->spawn_count;
        inf.event_capacity = fx->parent ? fx->parent->channels[fx->parent_channel].capacity : 0u;
        inf.ublock = ublocks + (size_t)i * nu;
        memcpy(fi[i].xf, fx->xf, sizeof fx->xf);
        // Parameter block: the uniform stream (literals, properties, sim params and every
        // expression built only from them) evaluated here, once per instance per frame.
        // (instances with the same property values share the result: thousands of instances of one effect cost one evaluation)
        if (nu) {
            if (i > 0 && fx->props == p->effects[i - 1]->props) memcpy(ublocks + (size_t)i * nu, ublocks + (size_t)(i - 1) * nu, (size_t)nu * 4);
            else uniform_run(p->uniform_code.data(), (uint32_t)p->uniform_code.size(), fx->props.data(), sim, ublocks + (size_t)i * nu, nu);
        }
    }
    // ---- the frame's proofs (hnb_plan.h): pure functions of the facts fixed at program creation, the history they carry and these inputs
    plan::FramePlan& pl = p->plan;
    pl = plan::FramePlan();
    {
        const unsigned long long pub = *reinterpret_cast<volatile unsigned long long*>(p->h_safe);   // {frame, bound} the device published: no read-back
        pl.skip_lists = plan::prove_skip_lists(p->skip_facts, p->skip_hist, p->frames_run, inst_frames.data(), n, plan::SkipPublished{(uint32_t)pub, (uint32_t)(pub >> 32)}, ctx->skip_lists);
    }
    if (ctx->break_proof && p->skip_facts.eligible && !pl.skip_lists) {   // the test hook: "nothing can die" claimed without proof in every frame that spawns nothing.
        bool spawns = false;                                                // The verification paths must notice (fault flag, hnb_effect_check, bench.py's gate): tests only.
        for (uint32_t i = 0; i < n; ++i) spawns = spawns || (inst_frames[i].simulated && (inst_frames[i].has_parent || inst_frames[i].spawn_count != 0u));
        pl.skip_lists = !spawns && p->frames_run > 0u;
    }
    if (p->has_ribbons) {
        // (ring lists: not for programs that emit spawn events - k_emit_* walks the rows as the update saw them - nor in timed frames of nothing: always)
        pl.ribbon = plan::prove_ribbon_order(p->ribbon_facts, p->ribbon_hist, p->dev.capacity, inst_frames.data(), n, ctx->skip_lists, ctx->suffix_proof,
                                             ctx->ring_lists && p->hdr.n_event_channels == 0u && p->update_streams && !p->slot_order);
        // a list that may stand behind a head must be rewritten (linear, head 0) in the first frame that is not a ring frame: the sort kernels and the
        // slot-ordered rebuild only know linear lists. That frame therefore runs its list kernels whatever the no-death proof says.
        if (p->ring_live && !pl.ribbon.ring) pl.skip_lists = false;
    }
    pl.lists = !(p->update_streams && pl.skip_lists);  // false: proven no spawn, no casualty; the update kernel rotates the counters
    if (p->skip_facts.eligible && !p->tick_sign_seen)
        for (uint32_t i = 0; i < n; ++i)
            if (inst_frames[i].simulated && !plan::nonneg_not_nan(plan::uword(inst_frames[i], p->skip_facts.dt_operand)) ) p->tick_sign_seen = true;
    if (pl.ribbon.ring && !pl.lists) pl.ribbon.ring = false;   // (nothing spawns, nothing can die: the list stands, head and all)
    p->dev.ring = pl.ribbon.ring ? 1u : 0u;
    pl.slot_init = plan::plan_slot_init(p->slot_init_eligible, ctx->slot_init, p->dev.capacity, p->dev.chunks_per_inst, inst_frames.data(), n);
    pl.hz_use = plan::horizon_usable(p->horizon_eligible, p->cull_dt_operand, inst_frames.data(), n);   // k_count_rows may skip row chunks
    if (pl.hz_use && pl.lists) p->hz_frames += 1;
    p->dev.hz_parity = p->hz_parity;
    p->dev.frame_no = p->frames_run;
    {   // cache policy: bytes of attribute planes the update loads + stores per slot
        uint32_t bytes = 0, stored = 0;
        for (uint32_t a = 0; a < p->dev.n_attrs; ++a) {
            const DevAttr& at = p->dev.attrs[a];
            bytes += ((at.upd_flags & HNB_ATTR_UPD_LOAD) ? 4u * at.ncomp : 0u) + ((at.upd_flags & HNB_ATTR_UPD_STORE) ? 4u * at.ncomp : 0u);
            stored += (at.upd_flags & HNB_ATTR_UPD_STORE) ? 4u * at.ncomp : 0u;
        }
        pl.stream_hint = ctx->stream_hints && plan::use_streaming_hints((uint64_t)n * p->dev.capacity, bytes);
        pl.store_hint = pl.stream_hint && plan::use_store_hints((uint64_t)n * p->dev.capacity, stored);
        p->dev.stream_hint = pl.stream_hint ? 1u : 0u;
    }
    pl.independent = p->hdr.n_event_channels == 0 && !(p->hdr.flags & HNB_PROG_READS_PARENT) && n != 0u;
    for (uint32_t i = 0; i < n; ++i) if (p->effects[i]->parent) pl.independent = false;
    uint32_t* init_start = ublocks + (size_t)n * nu;  // packed copy of init_block_start for k_init's search
    for (uint32_t i = 0; i < n; ++i) init_start[i] = fi[i].init_block_start;
    stage_off += (frame_bytes_for(p, n) + 255u) & ~(size_t)255u;
    pl.init_blocks = blocks;
    p->dev.n_inst = n;

}

// (3a) Programs whose list kernels can share two launches
static void fill_lists_jobs(HnbContext* ctx, const std::vector<HnbProgram*>& order, uint32_t slot, size_t& stage_off, bool timed, FrameJobs& jobs_out) {
    uint32_t candidates = 0;
    auto shares = [&](const HnbProgram* p) { return p->plan.lists && !p->slot_order && p != jobs_out.heavy; };   // (the heavy program's lists follow its update on the main stream)
    for (const HnbProgram* p : order) candidates += shares(p) ? 1u : 0u;
    if (candidates >= 2u && !timed) {
        ListsJob* jobs = reinterpret_cast<ListsJob*>(static_cast<char*>(ctx->h_stage[slot]) + stage_off);
        jobs_out.d_lists = reinterpret_cast<const ListsJob*>(static_cast<const char*>(ctx->d_stage[slot]) + stage_off);
        for (HnbProgram* p : order) {
            if (!shares(p)) continue;
            const uint32_t n = (uint32_t)p->effects.size();
            ListsJob jb{};
            jb.args = compact_args_of(p);
            jb.cb = compact_bufs_of(ctx, p, n);
            jb.inst_base = reinterpret_cast<const uint64_t*>(p->d_inst_base);
            jb.meta_in = p->d_meta[p->parity];
            jb.meta_out = p->d_meta[p->parity ^ 1u];
            jb.fi = reinterpret_cast<const DevFrameInst*>(p->d_frame_cur);
            jb.first_wg = jobs_out.lists_wgs;
            jb.n_wg = n * p->dev.chunks_per_inst;
            jobs_out.lists_wgs += jb.n_wg;
            jobs[jobs_out.n_lists++] = jb;
            p->plan.lists_merged = true;
        }
        stage_off += ((size_t)jobs_out.n_lists * sizeof(ListsJob) + 255u) & ~(size_t)255u;
    }

}

// The set module of the context's small programs (hnb_kernels.hip.h "SET MODULES"). The POPULATION is every program that can take part in merged
// launches at all (independent, small, narrow register file); a module is looked up - in the cache, or compiled: HNB_OPT_SET_MODULE - when a merged
// frame finds a candidate without a case and the population has stood for two frames (an application that creates an effect per frame must not
// generate a module source per frame). A module is replaced, never extended; programs it does not know leave their launches to the interpreters.
static uint32_t set_case_of(HnbContext* ctx, HnbProgram* p) {
    if (p->set_gen != ctx->set.gen) { p->set_gen = ctx->set.gen; p->set_case = ctx->set.module ? ctx->set.plan.case_of(p->set_sig) : kNoSetCase; }
    return p->set_case;
}
static void install_set_module(HnbContext* ctx, jit::SetResult& res);
static void refresh_set_module(HnbContext* ctx, const std::vector<HnbProgram*>& order, const std::vector<plan::MergeFacts>& facts) {
    if (ctx->set_job && ctx->set_job->done.load(std::memory_order_acquire)) {   // a background compilation has finished: take its module, whatever the mode is now
        ctx->set_job->worker.join();
        if (ctx->set_job->ok) {
            install_set_module(ctx, ctx->set_job->res);
            ctx->set_lookup.tried = 0;   // (the population may have moved on while it was compiled: look again)
        } else {
            // a set that does not compile (hiprtc error, a module too large) must not be compiled again and again for the life of the context:
            // `tried` keeps naming the population the job was started for, so the lookup below is not due until the population changes, and the
            // log keeps the compiler's message (round 4: tried was reset here too, the next merged frame found nothing in the cache and started
            // the same compilation over - a CPU thread in hiprtc for ever, the error overwritten by "compiling ... in the background")
            ctx->set_log = "the background compilation failed, this population is not tried again: " + ctx->set_job->res.log.substr(0, 600);
            ctx->set_failed_builds += 1;
        }
        ctx->set_job.reset();
    }
    if (ctx->set_mode == HNB_SET_MODULE_OFF || ctx->set_job) return;   // (while a compilation runs nothing is looked up: generating a module source costs a millisecond)
    std::vector<HnbProgram*> cand;
    bool covered = ctx->set.module != nullptr;
    for (size_t i = 0; i < order.size(); ++i) {
        HnbProgram* p = order[i];
        if (p->set_sig.empty() || !facts[i].independent || facts[i].total_chunks > kSceneMaxChunks) continue;
        cand.push_back(p);
        if (covered && set_case_of(ctx, p) == kNoSetCase) covered = false;
    }
    if (covered || cand.size() < 2u) return;
    std::vector<uint64_t> hs;
    for (const HnbProgram* p : cand) hs.push_back(p->set_sig_hash);
    std::sort(hs.begin(), hs.end());
    hs.erase(std::unique(hs.begin(), hs.end()), hs.end());
    const uint64_t pop = jit::hash_bytes(reinterpret_cast<const char*>(hs.data()), hs.size() * sizeof(uint64_t)) | 1ull;
    if (!plan::set_lookup_due(ctx->set_lookup, true, false, false, (uint32_t)cand.size(), pop)) return;
    std::vector<jit::Request> members;
    for (const HnbProgram* p : cand)
        members.push_back(make_set_request(p->h_init.data(), (uint32_t)p->h_init.size(), p->h_update.data(), (uint32_t)p->h_update.size(), p->attrs.data(),
                                           (uint32_t)p->attrs.size(), p->update_streams, p->dev.age_cohort != 0u));
    jit::SetResult res;
    if (!jit::build_set(members, res, ctx->set_mode != HNB_SET_MODULE_COMPILE)) {
        if (ctx->set_mode == HNB_SET_MODULE_BACKGROUND && res.log.empty()) {   // not in the cache: compile it beside the frames (one job at a time)
            auto job = std::make_shared<SetBuildJob>();
            for (const HnbProgram* p : cand) {
                SetBuildJob::Member m;
                m.attrs = p->attrs; m.init = p->h_init; m.update = p->h_update; m.streams = p->update_streams; m.cohort = p->dev.age_cohort != 0u;
                job->members.push_back(std::move(m));
            }
            SetBuildJob* j = job.get();   // (the context keeps the job alive until it has joined the worker)
            job->worker = std::thread([j]() {
                std::vector<jit::Request> rq;
                for (const SetBuildJob::Member& m : j->members)
                    rq.push_back(make_set_request(m.init.data(), (uint32_t)m.init.size(), m.update.data(), (uint32_t)m.update.size(), m.attrs.data(), (uint32_t)m.attrs.size(), m.streams, m.cohort));
                j->ok = jit::build_set(rq, j->res, false);
                j->done.store(true, std::memory_order_release);
            });
            ctx->set_job = std::move(job);
            ctx->set_log = "compiling a module for this set of " + std::to_string(res.plan.signatures.size()) + " programs in the background";
            return;
        }
        ctx->set_log = res.log.empty() ? std::string("no cache entry for this set of ") + std::to_string(res.plan.signatures.size()) + " programs (hnb_jit_precompile_set, or HNB_SET_MODULE_COMPILE / _BACKGROUND)" : res.log;
        return;
    }
    install_set_module(ctx, res);
}
static void install_set_module(HnbContext* ctx, jit::SetResult& res) {
    hipModule_t mod = nullptr;
    hipFunction_t fi = nullptr, fu = nullptr;
    hipError_t e = hipModuleLoadData(&mod, res.code.data());
    if (e == hipSuccess) e = hipModuleGetFunction(&fi, mod, "hnb_set_init");
    if (e == hipSuccess) e = hipModuleGetFunction(&fu, mod, "hnb_set_update");
    if (e != hipSuccess) {
        ctx->set_log = std::string("loading the set module failed: ") + hipGetErrorString(e);
        if (mod) hipModuleUnload(mod);
        (void)hipGetLastError();
        return;
    }
    if (ctx->set.module) {   // (kernels of the module being replaced may still be running)
        hipStreamSynchronize(ctx->stream);
        hipModuleUnload(ctx->set.module);
    }
    ctx->set.module = mod; ctx->set.init = fi; ctx->set.update = fu;
    ctx->set.plan = std::move(res.plan);
    ctx->set.gen += 1;
    ctx->set_log = std::to_string(ctx->set.plan.signatures.size()) + " programs" + (res.from_cache ? " (jit cache hit)" : " (compiled)");
}

// (3b) Small programs share their init and update launches
static void fill_merge_jobs(HnbContext* ctx, const std::vector<HnbProgram*>& order, uint32_t slot, size_t& stage_off, bool timed, FrameJobs& fj) {
    std::vector<plan::MergeFacts> facts(order.size());
    std::vector<plan::MergeDecision> decisions(order.size());
    for (size_t i = 0; i < order.size(); ++i) {
        const HnbProgram* p = order[i];
        plan::MergeFacts& f = facts[i];
        f.independent = p->plan.independent;
        f.total_chunks = (uint32_t)std::min<uint64_t>((uint64_t)p->effects.size() * p->dev.chunks_per_inst, 0xffffffffull);
        f.init_blocks = p->plan.slot_init.use ? 0u : p->plan.init_blocks;   // (a slot-major init pass is the program's own launch)
        f.init_len = p->dev.init_len; f.update_len = p->dev.update_len;
        f.wide_file = p->wide_file; f.update_streams = p->update_streams; f.age_cohort = p->dev.age_cohort != 0u;
    }
    plan::MergeLimits lim;
    lim.max_chunks = kSceneMaxChunks; lim.max_init_blocks = kSceneMaxInitBlocks; lim.max_code_len = kSceneMaxCodeLen;
    plan::plan_merged_launches(facts.data(), decisions.data(), (uint32_t)order.size(), ctx->scene_merge, timed, lim);
    bool any_merged = false;
    for (size_t i = 0; i < order.size(); ++i) { order[i]->plan.merge = decisions[i]; any_merged = any_merged || decisions[i].init_family >= 0 || decisions[i].update_family >= 0; }
    if (any_merged) refresh_set_module(ctx, order, facts);
    {   // programs the loaded module does not know: out of the shared launches while they are few (plan::split_uncovered)
        std::vector<uint8_t> has_case(order.size()), own(order.size()), out(order.size());
        for (size_t i = 0; i < order.size(); ++i) {
            HnbProgram* p = order[i];
            has_case[i] = ctx->set.module && set_case_of(ctx, p) != kNoSetCase ? 1u : 0u;
            const bool aot_static = p->update_streams && strcmp(p->stream_kernel_name, "ProgInterp") != 0;
            own[i] = ((p->jit_init || p->hdr.init_len == 0u) && (p->jit_update || aot_static || p->hdr.update_len == 0u)) ? 1u : 0u;   // (nothing of it would be interpreted)
        }
        plan::split_uncovered(decisions.data(), has_case.data(), own.data(), (uint32_t)order.size(), ctx->set.module != nullptr, out.data());
        for (size_t i = 0; i < order.size(); ++i)
            if (out[i]) { order[i]->plan.merge = plan::MergeDecision(); order[i]->unmerged_frames += 1; }
    }
    bool init_cased = ctx->set.module != nullptr, update_cased = ctx->set.module != nullptr;
    // k_update_jobs serves streaming (no cohorts), streaming (cohorts), V register file (narrow) in this order: first_wg runs over its whole grid
    const int seq[6][2] = {{0, 0}, {0, 1}, {2, plan::kStream}, {2, plan::kStreamCohort}, {1, plan::kGeneric}, {1, plan::kGenericWide}};   // {kind: 0 init / 1 generic / 2 stream, family}
    uint32_t update_base = 0;
    for (const auto& kv : seq) {
        const int kind = kv[0], fam = kv[1];
        auto member = [&](const HnbProgram* p) { return kind == 0 ? p->plan.merge.init_family == fam : p->plan.merge.update_family == fam; };
        const bool in_shared = kind == 2 || fam == plan::kGeneric;
        FrameJobs::Family& f = kind == 0 ? fj.init[fam] : kind == 1 ? fj.generic[fam == plan::kGenericWide ? 1 : 0] : fj.stream[fam == plan::kStreamCohort ? 1 : 0];
        const uint32_t base = in_shared ? update_base : 0u;
        char* hj = static_cast<char*>(ctx->h_stage[slot]) + stage_off;
        f.d_jobs = static_cast<const char*>(ctx->d_stage[slot]) + stage_off;
        for (HnbProgram* p : order) {
            if (!member(p)) continue;
            const uint32_t n = (uint32_t)p->effects.size();
            const char* d = p->d_frame_cur;
            const DevFrameInst* dfi = reinterpret_cast<const DevFrameInst*>(d);
            const uint32_t* dub = reinterpret_cast<const uint32_t*>(d + (size_t)n * sizeof(DevFrameInst));
            const uint32_t write_died = (p->plan.lists && !p->slot_order) ? 1u : 0u;
            // (streaming programs without cohorts: four workgroups per chunk, SlotArgs::quarters)
            const uint32_t quarters = (kind == 2 && fam == plan::kStream && HNB_STREAM_QUARTERS) ? 4u : 1u;
            const uint32_t wgs = kind == 0 ? p->plan.init_blocks : n * p->dev.chunks_per_inst * (kind == 1 ? kGenericSubs : quarters);
            if (kind == 2) {
                StreamJob jb{};
                jb.args = slot_args_of(ctx, p, n, write_died);
                jb.args.quarters = quarters;
                jb.inst_base = reinterpret_cast<const uint64_t*>(p->d_inst_base); jb.fi = dfi; jb.ublocks = dub;
                jb.cb = compact_bufs_of(ctx, p, n);
                jb.first_wg = base + f.wgs; jb.n_wg = wgs;
                jb.set_case = set_case_of(ctx, p);
                update_cased = update_cased && jb.set_case != kNoSetCase;
                reinterpret_cast<StreamJob*>(hj)[f.n] = jb;
            } else {
                ProgJob jb{};
                jb.prog = p->dev;
                jb.inst_base = reinterpret_cast<const uint64_t*>(p->d_inst_base); jb.meta_in = p->d_meta[p->parity]; jb.fi = dfi; jb.ublocks = dub;
                jb.cb = compact_bufs_of(ctx, p, n);
                jb.write_died = write_died;
                jb.first_wg = base + f.wgs; jb.n_wg = wgs;
                jb.set_case = set_case_of(ctx, p);
                if (kind == 0 && fam == 0) init_cased = init_cased && jb.set_case != kNoSetCase;
                if (kind == 1 && fam == plan::kGeneric) update_cased = update_cased && jb.set_case != kNoSetCase;
                reinterpret_cast<ProgJob*>(hj)[f.n] = jb;
            }
            f.n += 1; f.wgs += wgs;
            if (kind != 0) p->merged_frames += 1;
        }
        stage_off += ((size_t)f.n * (kind == 2 ? sizeof(StreamJob) : sizeof(ProgJob)) + 255u) & ~(size_t)255u;
        if (in_shared) update_base += f.wgs;
    }
    fj.init_set = init_cased && fj.init[0].n != 0u;
    fj.update_set = update_cased && (fj.stream[0].wgs + fj.stream[1].wgs + fj.generic[0].wgs) != 0u;
    if (fj.init_set || fj.update_set) {
        ctx->set_frames += 1;
        for (HnbProgram* p : order)
            if ((fj.init_set && p->plan.merge.init_family == 0) || (fj.update_set && p->plan.merge.update_family >= 0 && p->plan.merge.update_family != plan::kGenericWide)) p->set_frames += 1;
    }

}

// ribbon sort of a program's compacted lists by (RIBBON_ID, AGE) (src/render/mod.rs:7372-7612)
static void enqueue_ribbon_sort(HnbContext* ctx, HnbProgram* p, hipStream_t st) {
    const uint32_t n = (uint32_t)p->effects.size();
    const uint32_t par = p->parity;
    // The list is last frame's sorted list minus the casualties (stable compaction), every age advanced by the same
    // non-negative tick (monotone under rounding; non-negative floats order like their bits), plus this frame's spawns
    // at the end. Where the host can prove the premises (plan::RibbonFacts::provable + this frame's values + no host write)
    // the radix range is at most the largest spawn request: nothing to do without spawns, one single-workgroup launch
    // for a small range. Otherwise the device decides (k_sort_fill's order check) and all launches are issued.
    const bool proven = p->plan.ribbon.head_sorted;
    if (proven && p->plan.ribbon.max_spawn == 0u) return;
    // (r6) A frame whose list kernels were skipped - proven: nothing spawns, nothing can die, every tick finite and non-negative, no host write - leaves the
    // list exactly as the previous frame's sort left it, and every key moved by the same tick (monotone under rounding): still sorted, whatever the
    // program's ages and ribbon ids look like. (The lightning bolt of lightning.rs - ages and ribbon ids hashed from PARTICLE_COUNTER, nothing provable -
    // launched a one-workgroup sort in every frame of its 1.5 s between two strikes: 5 us of a 36 us scene frame.)
    // Keys are age BITS: a uniform tick keeps their order only while no age crosses zero. No tick of this program was ever negative or NaN
    // (HnbProgram::tick_sign_seen, sticky), and an age that carries the sign bit makes the update publish a no-death bound of 0 (update_stream_chunk), so
    // a frame whose lists were skipped follows a frame in which every alive age was +0 or above.
    if (!p->plan.lists && !p->ribbon_hist.dirty && p->frames_run > 0u && !p->tick_sign_seen) { p->sort_skipped_frames += 1; return; }
    if (p->plan.ribbon.rotate) {  // the spawns go in front and k_compact has written the survivors in that order (CompactArgs::rotate_front): nothing to sort
        p->sort_rotated_frames += 1;
        return;
    }
    SortArgs so = p->sort;
    so.parity = p->sort_parity & 1u;
    p->sort_parity += 1;
    const DevMeta* mo = p->d_meta[par ^ 1];
    const uint32_t tiles = n * so.chunks_per_inst;
    if (so.chunks_per_inst == 1u) {   // the instance is one tile: fill, the range's radix passes and the merge in one launch
        k_sort_tile1<<<n, kBlock, 0, st>>>(so, p->d_inst_base, mo);
        p->ribbon_hist.dirty = false;
        return;
    }
    k_sort_fill<<<tiles, kBlock, 0, st>>>(so, p->d_inst_base, mo);
    // ... or when the whole list is small: whatever range the device finds, one workgroup sorts it faster than sixteen launches
    // are issued (a 40-particle lightning bolt whose ages are not provably ordered took 8 + 8 empty launches per frame)
    if ((proven && p->plan.ribbon.max_spawn <= kSortSmallMax) || p->dev.capacity <= kSortSmallMax / 4u) {
        k_sort_small<<<n, kBlock, 0, st>>>(so, p->d_inst_base, mo);
    } else {
        for (uint32_t pass = 0; pass < 8; ++pass) {
            k_sort_hist<<<tiles, kBlock, 0, st>>>(so, p->d_inst_base, mo, pass);
            k_sort_scatter<<<tiles, kBlock, 0, st>>>(so, p->d_inst_base, mo, pass);
        }
    }
    k_sort_merge<<<tiles, kBlock, 0, st>>>(so, p->d_inst_base, mo);
    p->ribbon_hist.dirty = false;
}

// (5) init passes, parents first. With a heavy program (enqueue_update_passes) the frame forks HERE, behind the heavy program's own init: what
// has to precede that - the programs its instances' parents belong to, transitively: the init reads the parent particle - and what must not run
// beside its update - the programs that read ITS particles at init - stay on the context's stream in front of the fork; every other init pass
// goes to the side stream, in the same order, followed there by the light programs' update phases (c2_events: the sparkle effect's init no
// longer stands between the trails' init and update).
static bool reads_particles_of(const HnbProgram* child, const HnbProgram* parent) {
    for (const HnbEffect* fx : child->effects)
        if (fx->parent && fx->parent->prog == parent) return true;
    return false;
}
static int enqueue_init_passes(HnbContext* ctx, const std::vector<HnbProgram*>& order, FrameJobs& fj, bool timed) {
    const uint32_t np = (uint32_t)order.size();
    std::vector<uint8_t> on_side(np, 0);
    if (fj.heavy) {   // (contexts with a heavy program hold a handful of programs: the n x n relation is small)
        std::vector<uint8_t> reads((size_t)np * np, 0);
        int heavy = -1;
        for (uint32_t c = 0; c < np; ++c) {
            if (order[c] == fj.heavy) heavy = (int)c;
            for (uint32_t p = 0; p < np; ++p) reads[(size_t)c * np + p] = reads_particles_of(order[c], order[p]) ? 1u : 0u;
        }
        plan::partition_init_passes(reads.data(), np, heavy, on_side.data());
    }
    for (int pass = 0; pass < (fj.heavy ? 2 : 1); ++pass) {
    hipStream_t st = ctx->stream;
    if (pass == 1) {   // fork
        st = ctx->side_stream;
        HIP_TRY(hipEventRecord(ctx->ev_fork, ctx->stream));
        HIP_TRY(hipStreamWaitEvent(st, ctx->ev_fork, 0));
        fj.forked = true;
    }
    for (size_t pi = 0; pi < order.size(); ++pi) {
        HnbProgram* p = order[pi];
        if ((int)on_side[pi] != pass) continue;
        const uint32_t n = (uint32_t)p->effects.size();
        const uint32_t par = p->parity;
        const uint32_t blocks = p->plan.init_blocks;
        const char* d = p->d_frame_cur;
        const DevFrameInst* dfi = reinterpret_cast<const DevFrameInst*>(d);
        const uint32_t* dub = reinterpret_cast<const uint32_t*>(d + (size_t)n * sizeof(DevFrameInst));
        if (blocks && p->plan.merge.init_family < 0) {   // (>= 0: served by k_init_jobs below)
            TimingPair ti{};
            ti.prog = p;
            if (timed) { ti.a = take_event(ctx); ti.b = take_event(ctx); hipEventRecord(ti.a, st); }
            if (p->plan.slot_init.use) {   // a large spawn: the init walks the slots (hnb_kernels.hip.h "slot-major init")
                const uint32_t grid = n * p->dev.chunks_per_inst * (kChunk / kSlotInitWg);
                const DevMeta* mi = p->d_meta[par];
                if (p->plan.slot_init.marks) k_spawn_mark<<<grid, kBlock, 0, st>>>(p->dev, p->d_inst_base, mi, dfi);
                if (p->jit_init_slots) {
                    void* ka[] = {&p->dev, &p->d_inst_base, &mi, &dfi, &dub};
                    HIP_TRY(hipModuleLaunchKernel(p->jit_init_slots, grid, 1, 1, kBlock, 1, 1, 0, st, ka, nullptr));
                } else if (p->wide_file) k_init_slots<InterpCodeWide><<<grid, kBlock, 0, st>>>(p->dev, p->d_inst_base, mi, dfi, dub);
                else k_init_slots<InterpCode><<<grid, kBlock, 0, st>>>(p->dev, p->d_inst_base, mi, dfi, dub);
                p->slot_init_frames += 1;
            } else if (p->jit_init) {
                const DevMeta* mi = p->d_meta[par];
                void* ka[] = {&p->dev, &p->d_inst_base, &mi, &dfi, &dub};
                HIP_TRY(hipModuleLaunchKernel(p->jit_init, blocks, 1, 1, kInitBlock, 1, 1, 0, st, ka, nullptr));
            } else {
                if (p->wide_file) k_init<InterpCodeWide><<<blocks, kInitBlock, 0, st>>>(p->dev, p->d_inst_base, p->d_meta[par], dfi, dub);
                else k_init<InterpCode><<<blocks, kInitBlock, 0, st>>>(p->dev, p->d_inst_base, p->d_meta[par], dfi, dub);
            }
            if (timed) { hipEventRecord(ti.b, st); ctx->t_init.push_back(ti); }
        }
    }
    }

    // (the merged programs have no parent and no child: their init passes are independent of the ones above - behind the fork, if there is one)
    hipStream_t ms = fj.forked ? ctx->side_stream : ctx->stream;
    if (fj.init_set) {
        const void* jobs = fj.init[0].d_jobs;
        uint32_t n_jobs = fj.init[0].n;
        void* ka[] = {&jobs, &n_jobs};
        HIP_TRY(hipModuleLaunchKernel(ctx->set.init, fj.init[0].wgs, 1, 1, kInitBlock, 1, 1, 0, ms, ka, nullptr));
    } else if (fj.init[0].n) k_init_jobs<InterpCode><<<fj.init[0].wgs, kInitBlock, 0, ms>>>(static_cast<const ProgJob*>(fj.init[0].d_jobs), fj.init[0].n);
    if (fj.init[1].n) k_init_jobs<InterpCodeWide><<<fj.init[1].wgs, kInitBlock, 0, ms>>>(static_cast<const ProgJob*>(fj.init[1].d_jobs), fj.init[1].n);
    return HNB_OK;
}

// (6) update + kill + compaction (+ spawn-event ordering, + ribbon sort)
// One program's update phase on stream `st`: update (unless a merged launch serves it), spawn-event ordering, lists, ribbon sort.
static int enqueue_program_update(HnbContext* ctx, HnbProgram* p, hipStream_t st, bool timed) {
    const uint32_t n = (uint32_t)p->effects.size();
    const uint32_t par = p->parity;
    const char* d = p->d_frame_cur;
    const DevFrameInst* dfi = reinterpret_cast<const DevFrameInst*>(d);
    const uint32_t* dub = reinterpret_cast<const uint32_t*>(d + (size_t)n * sizeof(DevFrameInst));
    // one workgroup per 4096-slot chunk of every instance
    const uint32_t total_chunks = n * p->dev.chunks_per_inst;
    CompactBufs cb = compact_bufs_of(ctx, p, n);
    TimingPair tu{}, tc{};
    tu.prog = tc.prog = p;
    if (timed) { tu.a = take_event(ctx); tu.b = take_event(ctx); tc.b = take_event(ctx); hipEventRecord(tu.a, st); }
    const uint32_t write_died = (p->plan.lists && !p->slot_order) ? 1u : 0u;   // k_count_rows follows: the update leaves one died bit per slot
    if (p->plan.merge.update_family >= 0) {
        // (served by k_update_jobs / k_update_generic_wide_jobs: enqueue_update_passes)
    } else if (p->update_streams) {
        SlotArgs sa = slot_args_of(ctx, p, n, write_died);
        if (p->jit_update) {
            void* ka[] = {&sa, &p->d_inst_base, &dfi, &dub, &cb};
            HIP_TRY(hipModuleLaunchKernel(p->jit_update, total_chunks, 1, 1, kBlock, 1, 1, 0, st, ka, nullptr));
        } else {
            p->stream_launch(total_chunks, st, sa, p->d_inst_base, dfi, dub, cb);
        }
    } else {
        // a generic program of a few chunks: one workgroup per 256 slots (k_update_slots_generic's `split`), while the grid still fits the GPU at once
        uint32_t split = total_chunks <= kGenericSplitMaxChunks ? 1u : 0u;
        const uint32_t grid = split ? total_chunks * (kChunk / kBlock) : total_chunks;
        if (p->jit_update) {
            uint32_t dm = write_died;
            void* ka[] = {&p->dev, &p->d_inst_base, &dfi, &dub, &cb, &dm, &split};
            HIP_TRY(hipModuleLaunchKernel(p->jit_update, grid, 1, 1, kBlock, 1, 1, 0, st, ka, nullptr));
        } else if (p->wide_file) k_update_slots_generic<InterpCodeWide><<<grid, kBlock, 0, st>>>(p->dev, p->d_inst_base, dfi, dub, cb, write_died, split);
        else k_update_slots_generic<InterpCode><<<grid, kBlock, 0, st>>>(p->dev, p->d_inst_base, dfi, dub, cb, write_died, split);
    }
    if (timed) { hipEventRecord(tu.b, st); ctx->t_update.push_back(tu); }
    const CompactArgs ca = compact_args_of(p);
    const bool lists = p->plan.lists;
    if (!lists) p->skipped_frames += 1;
    else if (ca.suffix_dead) p->suffix_frames += 1;
    if (lists && p->dev.n_event_channels) {  // order this frame's spawn events (by list row) into the children's buffers
        k_emit_count<<<total_chunks, kBlock, 0, st>>>(p->dev, p->d_inst_base, p->d_meta[par], dfi, cb);
        // (gridDim.y splits every chunk's events over several workgroups: sized for the largest event buffer that listens, 16,384 events per split)
        uint32_t max_ev = 0;
        for (const HnbEffect* fx : p->effects)
            for (const EventChannel& ch : fx->channels) max_ev = std::max(max_ev, ch.capacity);
        const uint32_t splits = plan::size_event_grid(max_ev, total_chunks);
        k_emit_events<<<dim3(total_chunks, splits), kBlock, 0, st>>>(p->dev, p->d_inst_base, p->d_meta[par], dfi, cb, p->h_ev_counts, ctx->frame);
    }
    // lists: only the instances that lost particles have anything to do
    if (!p->plan.lists_merged) {  // (merged: the lists, and the ribbon sort behind them, follow after the last program's update)
        if (lists && !p->slot_order && !ca.suffix_dead) k_count_rows<<<total_chunks, kBlock, 0, st>>>(ca, p->d_inst_base, p->d_meta[par], dfi, cb);
        if (lists) k_compact<<<total_chunks, kBlock, 0, st>>>(ca, p->d_inst_base, p->d_meta[par], p->d_meta[par ^ 1], dfi, cb);
        if (lists && p->slot_order) {  // rebuild the lists in increasing slot order (instances without a casualty or spawn return at once)
            k_order_count<<<total_chunks, kBlock, 0, st>>>(ca, p->d_inst_base, p->d_meta[par], dfi, cb);
            k_order_write<<<total_chunks, kBlock, 0, st>>>(ca, p->d_inst_base, p->d_meta[par], p->d_meta[par ^ 1], dfi, cb);
        }
        if (timed) { tc.a = tu.b; hipEventRecord(tc.b, st); ctx->t_compact.push_back(tc); }
        if (p->has_ribbons) enqueue_ribbon_sort(ctx, p, st);
    }
    HIP_TRY(hipGetLastError());
    return HNB_OK;
}

static int enqueue_update_passes(HnbContext* ctx, const std::vector<HnbProgram*>& order, const FrameJobs& fj, bool timed) {
    // fork (unless enqueue_init_passes did): the light programs' chains (and every shared launch) go to the side stream behind everything enqueued so far (the init passes);
    // the heavy program's chain stays on the context's stream; the context's stream waits for the side stream at the end of the frame
    hipStream_t light = ctx->stream;
    if (fj.heavy) {
        light = ctx->side_stream;
        if (!fj.forked) {
            HIP_TRY(hipEventRecord(ctx->ev_fork, ctx->stream));
            HIP_TRY(hipStreamWaitEvent(light, ctx->ev_fork, 0));
        }
    }
    // (the merged updates first: every init pass is enqueued, and a merged program's own list kernels may follow in the loop below)
    if (fj.update_set) {
        const void *sj0 = fj.stream[0].d_jobs, *sj1 = fj.stream[1].d_jobs, *pj = fj.generic[0].d_jobs;
        uint32_t n0 = fj.stream[0].n, n1 = fj.stream[1].n, np = fj.generic[0].n, b0 = fj.stream[0].wgs, b1 = fj.stream[0].wgs + fj.stream[1].wgs;
        void* ka[] = {&sj0, &n0, &sj1, &n1, &pj, &np, &b0, &b1};
        HIP_TRY(hipModuleLaunchKernel(ctx->set.update, fj.stream[0].wgs + fj.stream[1].wgs + fj.generic[0].wgs, 1, 1, kBlock, 1, 1, 0, light, ka, nullptr));
    } else if (fj.stream[0].wgs + fj.stream[1].wgs + fj.generic[0].wgs)
        k_update_jobs<<<fj.stream[0].wgs + fj.stream[1].wgs + fj.generic[0].wgs, kBlock, 0, light>>>(
            static_cast<const StreamJob*>(fj.stream[0].d_jobs), fj.stream[0].n, static_cast<const StreamJob*>(fj.stream[1].d_jobs), fj.stream[1].n,
            static_cast<const ProgJob*>(fj.generic[0].d_jobs), fj.generic[0].n, fj.stream[0].wgs, fj.stream[0].wgs + fj.stream[1].wgs);
    if (fj.generic[1].wgs) k_update_generic_wide_jobs<<<fj.generic[1].wgs, kBlock, 0, light>>>(static_cast<const ProgJob*>(fj.generic[1].d_jobs), fj.generic[1].n);
    for (HnbProgram* p : order) {
        if (p == fj.heavy) continue;
        const int rc = enqueue_program_update(ctx, p, light, timed);
        if (rc != HNB_OK) return rc;
    }
    if (fj.n_lists) {
        k_count_rows_multi<<<fj.lists_wgs, kBlock, 0, light>>>(fj.d_lists, fj.n_lists);
        k_compact_multi<<<fj.lists_wgs, kBlock, 0, light>>>(fj.d_lists, fj.n_lists);
        for (HnbProgram* p : order)
            if (p->plan.lists_merged && p->has_ribbons) enqueue_ribbon_sort(ctx, p, light);
        HIP_TRY(hipGetLastError());
    }
    if (fj.heavy) {
        const int rc = enqueue_program_update(ctx, const_cast<HnbProgram*>(fj.heavy), ctx->stream, timed);
        if (rc != HNB_OK) return rc;
        HIP_TRY(hipEventRecord(ctx->ev_join, light));
        HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));   // join: whatever follows on the context's stream sees the whole frame
    }
    return HNB_OK;
}

// One simulated frame, in the reference's order (src/render/mod.rs:6975-7370): every effect's init
// pass, parents before children, THEN every effect's update pass. Spawn events appended by a parent's
// update in frame N are consumed by its children's init in frame N+1.
int hnb_simulate(HnbContext* ctx) {
    if (!ctx) return fail(HNB_ERR_INVALID_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    install_finished_jit(ctx);
    std::vector<HnbProgram*> order;
    for (HnbProgram* p : ctx->programs)
        if (!p->effects.empty()) order.push_back(p);
    std::stable_sort(order.begin(), order.end(), [](const HnbProgram* x, const HnbProgram* y) { return x->level < y->level; });
    const bool timed = ctx->timing && (ctx->timing_tick % ctx->timing) == 0;
    // validate every instance before any per-frame state is touched: a failed call must leave the frame's inputs intact
    for (HnbProgram* p : order)
        if (!p->parent_attrs.empty())
            for (size_t i = 0; i < p->effects.size(); ++i)
                if (!p->effects[i]->parent)
                    return fail(HNB_ERR_INVALID_ARG, "effect #%zu reads its parent particle (InheritAttributeModifier / parent_attr) but has no parent: call hnb_effect_set_parent", i);
    const uint32_t slot = ctx->frame % kFrameRing;
    int rc = ensure_stage(ctx, order, slot);
    if (rc != HNB_OK) return rc;
    size_t stage_off = 0;
    std::vector<plan::InstanceFrame> inst_frames;
    for (HnbProgram* p : order) stage_program_frame(ctx, p, slot, stage_off, inst_frames);
    FrameJobs fj;
    fj.heavy = pick_heavy_program(ctx, order, timed);
    fill_lists_jobs(ctx, order, slot, stage_off, timed, fj);
    fill_merge_jobs(ctx, order, slot, stage_off, timed, fj);
    if (stage_off) {
        // (Round 5, tried: the copy on the simulation stream in front of the frame's first kernel instead - no host wait, no copy beside the previous frame's
        // kernels. Slower everywhere: C5 0.0359 -> 0.0378 ms, c2_mixed 0.353 -> 0.360, the 26-effect scene 0.050 -> 0.064: profiles/r05j_ab_upload.log.)
        // (Round 6, tried: no host-side copy at all - one small kernel on the simulation stream reads the pinned block over the link and writes the device block,
        // the host never waits. Slower everywhere as well: C5 0.0374 -> 0.0380 ms, c2 0.139 -> 0.142, the scene 0.051 -> 0.053, time inside hnb_simulate unchanged:
        // the frame of a small effect is bound by its CHAIN of dependent launches on the device - the host only blocks on the staging ring - and the copy kernel
        // is one more link of it: profiles/r06c_ab_stage_kernel.log, r06c_stage_kernel.patch.)
        // (Round 6, what did work: no copy - HnbContext::stage_direct. The slot's previous readers are done: ensure_stage waited for the frame that last used it.)
        if (ctx->stage_direct) {
            direct_write(ctx->d_stage[slot], ctx->h_stage[slot], stage_off);
            ctx->direct_frames += 1;
        } else {
            HIP_TRY(hipMemcpyAsync(ctx->d_stage[slot], ctx->h_stage[slot], stage_off, hipMemcpyHostToDevice, ctx->upload_stream));
            HIP_TRY(hipStreamSynchronize(ctx->upload_stream));
            ctx->copied_frames += 1;
        }
    }
    rc = enqueue_init_passes(ctx, order, fj, timed);
    if (rc == HNB_OK) rc = enqueue_update_passes(ctx, order, fj, timed);
    if (rc != HNB_OK) {
        // a launch failed somewhere behind the fork: what the side stream already holds must not be left running beside whatever the caller does
        // next (hnb_ctx_synchronize, read-backs and hnb_program_destroy wait for the context's stream only). The frame is not advanced.
        if (ctx->side_stream && (fj.forked || fj.heavy)) hipStreamSynchronize(ctx->side_stream);
        return rc;
    }
    for (HnbProgram* p : order) {
        p->ring += 1;
        p->parity ^= 1u;
        p->frames_run += 1;
        if (p->horizon_eligible && p->plan.lists) p->hz_parity ^= 1u;   // k_count_rows / k_compact moved the horizons to the other half
        if (p->plan.ribbon.ring) { p->ring_live = true; p->ring_frames += 1; }
        else if (p->plan.lists) p->ring_live = false;                   // (k_compact ran with force_rewrite: every instance's list is linear again)
    }
    if (!order.empty()) HIP_TRY(hipEventRecord(ctx->stage_done[slot], ctx->stream));
    for (HnbProgram* p : order)
        for (HnbEffect* fx : p->effects) fx->spawn_count = 0;  // a spawn request is consumed by exactly one (enqueued) frame
    ctx->frame += 1;
    if (ctx->timing) ctx->timing_tick += 1;
    return HNB_OK;
}

int hnb_effect_metadata(HnbEffect* fx, HnbEffectMetadata* out) {
    if (!fx || !out) return fail(HNB_ERR_INVALID_ARG, "NULL argument");
    DevMeta m;
    int rc = read_meta(fx, &m);
    if (rc != HNB_OK) return rc;
    const uint32_t cap = fx->prog->dev.capacity;
    out->capacity = cap;
    out->alive_count = m.alive_count;
    out->max_update = m.max_update;
    out->max_spawn = cap - m.alive_count;
    out->indirect_write_index = m.ref_write_index;  // as the reference counts it: flips every frame (vfx_indirect.wgsl:80-85)
    out->particle_counter = m.particle_counter;
    out->instance_count = m.instance_count;
    out->dispatch_x = (m.alive_count + 63u) >> 6;
    out->dead_count = m.dead_count;
    out->spawned = m.spawned;
    out->fault = 0;
    if (fx->prog->d_fault) HIP_TRY(hipMemcpy(&out->fault, fx->prog->d_fault, 4, hipMemcpyDeviceToHost));
    out->reserved = 0;
    return HNB_OK;
}

int hnb_effect_alive_count(HnbEffect* fx, uint32_t* out) {
    if (!fx || !out) return fail(HNB_ERR_INVALID_ARG, "NULL argument");
    DevMeta m;
    int rc = read_meta(fx, &m);
    if (rc != HNB_OK) return rc;
    *out = m.alive_count;
    return HNB_OK;
}

int hnb_effect_read_attr(HnbEffect* fx, uint32_t attr, void* dst, size_t dst_size) {
    if (!fx || !dst) return fail(HNB_ERR_INVALID_ARG, "NULL argument");
    HnbProgram* p = fx->prog;
    const int ai = find_attr(p, attr);
    if (ai < 0) return fail(HNB_ERR_NOT_FOUND, "attribute %u is not part of the particle layout", attr);
    const size_t bytes = (size_t)p->dev.capacity * p->attrs[ai].ncomp * 4;
    if (dst_size < bytes) return fail(HNB_ERR_INVALID_ARG, "destination too small (%zu < %zu)", dst_size, bytes);
    HIP_TRY(hipSetDevice(p->ctx->device));
    if (p->dev.age_cohort && attr == HNB_ATTR_AGE)   // chunks whose particles share one age keep it in a word: write it out first
        k_materialise_age<<<p->dev.chunks_per_inst, kBlock, 0, p->ctx->stream>>>(p->d_inst_base + fx->index, p->dev.capacity, p->dev.chunks_per_inst, p->dev.lmin_off,
                                                                                  p->dev.attrs[ai].plane_off, p->dev.alive_flag_off);
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    HIP_TRY(hipMemcpy(dst, static_cast<char*>(fx->slab) + p->dev.attrs[ai].plane_off, bytes, hipMemcpyDeviceToHost));
    return HNB_OK;
}

int hnb_effect_write_attr(HnbEffect* fx, uint32_t attr, const void* src, size_t src_size) {
    if (!fx || !src) return fail(HNB_ERR_INVALID_ARG, "NULL argument");
    HnbProgram* p = fx->prog;
    const int ai = find_attr(p, attr);
    if (ai < 0) return fail(HNB_ERR_NOT_FOUND, "attribute %u is not part of the particle layout", attr);
    const size_t bytes = (size_t)p->dev.capacity * p->attrs[ai].ncomp * 4;
    if (src_size != bytes) return fail(HNB_ERR_INVALID_ARG, "source size %zu != plane size %zu", src_size, bytes);
    HIP_TRY(hipSetDevice(p->ctx->device));
    hipStream_t st = p->ctx->stream;
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(static_cast<char*>(fx->slab) + p->dev.attrs[ai].plane_off, src, bytes, hipMemcpyHostToDevice));
    p->skip_hist.dirty = true;  // ... nor does the published no-death bound
    p->ribbon_hist.dirty = true;  // ... and a ribbon list may no longer be in key order
    p->ribbon_hist.front_broken = true;  // ... nor its ages and ribbon ids what the front proof assumes
    // (the resets below are enqueued on the context's stream - the device the slab lives on -, then waited for)
    if (p->horizon_eligible)   // the death horizons were computed from the particles as they were: zero = "may die now" (the clock restarts with them)
        HIP_TRY(hipMemsetAsync(static_cast<char*>(fx->slab) + p->dev.horizon_off, 0, 256 + (size_t)p->dev.chunks_per_inst * 24, st));
    if (attr == HNB_ATTR_AGE) p->ribbon_hist.values_broken = true;  // ... and the written ages may be negative (they change key order when they cross zero later)
    // the chunks' lifetime bounds (lifetime culling) no longer describe the planes
    HIP_TRY(hipMemsetAsync(static_cast<char*>(fx->slab) + p->dev.lmin_off, 0, (size_t)p->dev.chunks_per_inst * 4, st));  // (the "completely alive" flags that follow stay valid)
    if (p->dev.age_cohort && attr == HNB_ATTR_AGE)   // the plane is the truth again: forget the cohort states (and values)
        HIP_TRY(hipMemsetAsync(static_cast<char*>(fx->slab) + p->dev.lmin_off + (size_t)p->dev.chunks_per_inst * 8, 0, (size_t)p->dev.chunks_per_inst * 8, st));
    HIP_TRY(hipStreamSynchronize(st));
    return HNB_OK;
}

// ---- device-side output boundary (include/hanabi_amd.h "Device-side output") ---------------------------------------------------
// What the reference's render passes bind after `simulate` (src/render/mod.rs:5152-5820: the particle buffer, the indirect index buffer,
// the effect metadata row vfx_indirect.wgsl:57-85 turned into draw-indirect arguments) as device pointers. Nothing here synchronises.
static_assert(sizeof(HnbDeviceMeta) == sizeof(DevMeta), "HnbDeviceMeta is the public face of DevMeta");
static_assert(offsetof(HnbDeviceMeta, alive_count) == offsetof(DevMeta, alive_count) && offsetof(HnbDeviceMeta, list_column) == offsetof(DevMeta, write_index) &&
              offsetof(HnbDeviceMeta, max_update) == offsetof(DevMeta, max_update) && offsetof(HnbDeviceMeta, indirect_write_index) == offsetof(DevMeta, ref_write_index) &&
              offsetof(HnbDeviceMeta, instance_count) == offsetof(DevMeta, instance_count), "HnbDeviceMeta field order");
static_assert(HNB_VIEW_MAX_ATTRS >= kMaxAttrs, "a view holds every attribute a layout may have");

int hnb_effect_device_view(HnbEffect* fx, HnbDeviceView* out) {
    if (!fx || !out) return fail(HNB_ERR_INVALID_ARG, "NULL argument");
    HnbProgram* p = fx->prog;
    char* base = static_cast<char*>(fx->slab);
    memset(out, 0, sizeof *out);
    out->struct_size = (uint32_t)sizeof *out;
    out->device = p->ctx->device;
    out->stream = p->ctx->stream;
    out->capacity = p->dev.capacity;
    out->slot_base = fx->slot_base;
    out->n_attrs = p->dev.n_attrs;
    out->stale_attr_mask = (p->dev.age_cohort && !p->auto_materialise) ? (1ull << HNB_ATTR_AGE) : 0ull;   // (never set for an asset whose render modifiers read AGE under the default HNB_AGE_COHORT_AUTO)
    out->alive_list[0] = reinterpret_cast<const uint32_t*>(base + p->dev.alive_off[0]);
    out->alive_list[1] = reinterpret_cast<const uint32_t*>(base + p->dev.alive_off[1]);
    out->dead_list = reinterpret_cast<const uint32_t*>(base + p->dev.dead_off);
    out->meta = reinterpret_cast<const HnbDeviceMeta*>(p->d_meta[p->parity] + fx->index);
    out->meta_next = reinterpret_cast<const HnbDeviceMeta*>(p->d_meta[p->parity ^ 1u] + fx->index);
    for (uint32_t a = 0; a < p->dev.n_attrs; ++a) {
        HnbDeviceAttr& v = out->attrs[a];
        v.attr = p->attrs[a].attr;
        v.ncomp = p->attrs[a].ncomp;
        v.scalar_type = p->attrs[a].scalar_type;
        v.stride_bytes = (uint16_t)(p->attrs[a].ncomp * 4u);
        v.plane = base + p->dev.attrs[a].plane_off;
    }
    return HNB_OK;
}

int hnb_program_device_view(HnbProgram* p, HnbProgramView* out) {
    if (!p || !out) return fail(HNB_ERR_INVALID_ARG, "NULL argument");
    memset(out, 0, sizeof *out);
    out->struct_size = (uint32_t)sizeof *out;
    out->device = p->ctx->device;
    out->stream = p->ctx->stream;
    out->capacity = p->dev.capacity;
    out->n_instances = (uint32_t)p->effects.size();
    out->n_attrs = p->dev.n_attrs;
    out->stale_attr_mask = (p->dev.age_cohort && !p->auto_materialise) ? (1ull << HNB_ATTR_AGE) : 0ull;
    out->slabs = p->d_inst_base;
    out->meta = reinterpret_cast<const HnbDeviceMeta*>(p->d_meta[p->parity]);
    out->meta_next = reinterpret_cast<const HnbDeviceMeta*>(p->d_meta[p->parity ^ 1u]);
    out->alive_list_off[0] = (uint64_t)(size_t)p->dev.alive_off[0]; out->alive_list_off[1] = (uint64_t)(size_t)p->dev.alive_off[1];
    out->dead_list_off = (uint64_t)(size_t)p->dev.dead_off;
    for (uint32_t a = 0; a < p->dev.n_attrs; ++a) {
        HnbProgramAttr& v = out->attrs[a];
        v.attr = p->attrs[a].attr;
        v.ncomp = p->attrs[a].ncomp;
        v.scalar_type = p->attrs[a].scalar_type;
        v.stride_bytes = (uint16_t)(p->attrs[a].ncomp * 4u);
        v.plane_off = (uint64_t)(size_t)p->dev.attrs[a].plane_off;
    }
    return HNB_OK;
}

int hnb_effect_materialise(HnbEffect* fx, uint64_t attr_mask) {
    if (!fx) return fail(HNB_ERR_INVALID_ARG, "fx is NULL");
    HnbProgram* p = fx->prog;
    for (uint32_t a = 0; a < HNB_ATTR_COUNT; ++a)
        if ((attr_mask >> a & 1ull) && find_attr(p, a) < 0) return fail(HNB_ERR_NOT_FOUND, "attribute %u is not part of the particle layout", a);
    if (attr_mask >> HNB_ATTR_COUNT) return fail(HNB_ERR_INVALID_ARG, "attribute mask 0x%llx names attributes that do not exist", (unsigned long long)attr_mask);
    if (p->dev.age_cohort && (attr_mask >> HNB_ATTR_AGE & 1ull)) {   // the one plane the update may leave stale: chunks whose particles share one age keep it in a word
        HIP_TRY(hipSetDevice(p->ctx->device));
        const int ai = find_attr(p, HNB_ATTR_AGE);
        k_materialise_age<<<p->dev.chunks_per_inst, kBlock, 0, p->ctx->stream>>>(p->d_inst_base + fx->index, p->dev.capacity, p->dev.chunks_per_inst, p->dev.lmin_off,
                                                                                  p->dev.attrs[ai].plane_off, p->dev.alive_flag_off);
        HIP_TRY(hipGetLastError());
    }
    return HNB_OK;
}

int hnb_effect_read_alive_list(HnbEffect* fx, uint32_t* dst, size_t dst_count) {
    if (!fx || !dst) return fail(HNB_ERR_INVALID_ARG, "NULL argument");
    HnbProgram* p = fx->prog;
    DevMeta m;
    int rc = read_meta(fx, &m);
    if (rc != HNB_OK) return rc;
    if (dst_count < m.alive_count) return fail(HNB_ERR_INVALID_ARG, "destination too small");
    {   // rows [0, alive_count) of the column, behind the list's head ("Ring lists": 0 for everything but single-ribbon trails): at most two pieces
        const char* col = static_cast<char*>(fx->slab) + p->dev.alive_off[list_column(m.write_index)];
        const uint32_t head = list_head(m.write_index), cap = p->dev.capacity;
        const uint32_t first = std::min(m.alive_count, cap - std::min(head, cap));
        HIP_TRY(hipMemcpy(dst, col + (size_t)head * 4, (size_t)first * 4, hipMemcpyDeviceToHost));
        if (first < m.alive_count) HIP_TRY(hipMemcpy(dst + first, col, (size_t)(m.alive_count - first) * 4, hipMemcpyDeviceToHost));
    }
    return HNB_OK;
}

int hnb_effect_read_dead_list(HnbEffect* fx, uint32_t* dst, size_t dst_count) {
    if (!fx || !dst) return fail(HNB_ERR_INVALID_ARG, "NULL argument");
    HnbProgram* p = fx->prog;
    DevMeta m;
    int rc = read_meta(fx, &m);
    if (rc != HNB_OK) return rc;
    const uint32_t nd = p->dev.capacity - m.alive_count;
    if (dst_count < nd) return fail(HNB_ERR_INVALID_ARG, "destination too small");
    // rows [alive_count, capacity) hold the free slots; row alive_count is the top of the stack
    HIP_TRY(hipMemcpy(dst, static_cast<char*>(fx->slab) + p->dev.dead_off + (size_t)m.alive_count * 4, (size_t)nd * 4,
                      hipMemcpyDeviceToHost));
    return HNB_OK;
}

// ---- verification on the device: invariants of one effect, bitwise comparison of two (include/hanabi_amd.h "Verification") ----------------
// For state too large to read back and compare on the host in the time a benchmark may take (16.7M particles: 0.7 GB per effect): bench.py's
// gate on the state its TIMED frames produced, and anything else that wants to know "is this effect consistent" without moving it.
namespace {
__global__ void __launch_bounds__(256)
k_check_rows(const uint32_t* __restrict__ alive, uint32_t head, const uint32_t* __restrict__ dead, uint32_t alive_count, uint32_t capacity,
             const uint8_t* __restrict__ flags, const float* __restrict__ age, const float* __restrict__ life, uint32_t* __restrict__ seen, uint32_t* __restrict__ rep) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= capacity) return;
    const bool is_alive = r < alive_count;
    const uint32_t slot = is_alive ? alive[ring_row(head, r, capacity)] : dead[r];   // rows [0, alive_count) of the list column (behind its head), rows [alive_count, capacity) of the dead stack
    if (slot >= capacity) { atomicAdd(rep + 0, 1u); return; }
    const uint32_t bit = 1u << (slot & 31u);
    if (atomicOr(seen + (slot >> 5), bit) & bit) atomicAdd(rep + 1, 1u);          // listed twice
    if ((flags[slot] == 1u) != is_alive) atomicAdd(rep + 2, 1u);                 // the byte that drives the slot-major update disagrees with the lists
    if (is_alive && age && life && !(age[slot] < life[slot])) atomicAdd(rep + 3, 1u);   // the reaping rule of src/lib.rs:1223-1258 left it alive
}
__global__ void __launch_bounds__(256)
k_compare_words(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint64_t n, unsigned long long* __restrict__ out,   // out: {differing words, first index}
                uint32_t head_a = 0, uint32_t head_b = 0, uint32_t ring = 0) {   // ring != 0: word i of either side is at (head + i) % ring (list rows behind a head)
    unsigned long long diffs = 0, first = ~0ull;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256u) {
        const uint32_t x = ring ? a[ring_row(head_a, (uint32_t)i, ring)] : a[i], y = ring ? b[ring_row(head_b, (uint32_t)i, ring)] : b[i];
        if (x != y) { diffs += 1; first = first < i ? first : i; }
    }
    if (diffs) { atomicAdd(out, diffs); atomicMin(out + 1, first); }
}
}  // namespace

int hnb_effect_check(HnbEffect* fx, HnbEffectCheck* out) {
    if (!fx || !out) return fail(HNB_ERR_INVALID_ARG, "NULL argument");
    HnbProgram* p = fx->prog;
    memset(out, 0, sizeof *out);
    int rc = hnb_effect_materialise(fx, p->dev.age_cohort ? (1ull << HNB_ATTR_AGE) : 0ull);
    if (rc != HNB_OK) return rc;
    DevMeta m;
    rc = read_meta(fx, &m);
    if (rc != HNB_OK) return rc;
    const uint32_t cap = p->dev.capacity;
    char* base = static_cast<char*>(fx->slab);
    uint32_t* scratch = nullptr;
    const size_t seen_words = ((size_t)cap + 31u) / 32u;
    HIP_TRY(hipMalloc(&scratch, (seen_words + 4u) * 4u));
    hipStream_t st = p->ctx->stream;
    hipError_t e = hipMemsetAsync(scratch, 0, (seen_words + 4u) * 4u, st);   // (the duplicate-slot bitmap and the four counters must start from zero)
    if (e != hipSuccess) { hipFree(scratch); return fail(HNB_ERR_HIP, "hnb_effect_check: %s", hipGetErrorString(e)); }
    const int ia = find_attr(p, HNB_ATTR_AGE), il = find_attr(p, HNB_ATTR_LIFETIME);
    const bool reaps = ia >= 0 && il >= 0 && p->dev.cull_lifetime;   // (the update program starts with the AGE_TICK that tests the lifetime)
    k_check_rows<<<(uint32_t)(((uint64_t)cap + 255u) / 256u), 256, 0, st>>>(
        reinterpret_cast<const uint32_t*>(base + p->dev.alive_off[list_column(m.write_index)]), list_head(m.write_index), reinterpret_cast<const uint32_t*>(base + p->dev.dead_off), m.alive_count, cap,
        reinterpret_cast<const uint8_t*>(base + p->dev.alive_flag_off), reaps ? reinterpret_cast<const float*>(base + p->dev.attrs[ia].plane_off) : nullptr,
        reaps ? reinterpret_cast<const float*>(base + p->dev.attrs[il].plane_off) : nullptr, scratch + 4, scratch);
    uint32_t rep[4] = {};
    e = hipGetLastError();   // (the launch)
    if (e == hipSuccess) e = hipMemcpyAsync(rep, scratch, sizeof rep, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(scratch);
    if (e != hipSuccess) return fail(HNB_ERR_HIP, "hnb_effect_check: %s", hipGetErrorString(e));
    out->capacity = cap; out->alive_count = m.alive_count;
    out->bad_slots = rep[0]; out->duplicate_slots = rep[1]; out->alive_byte_mismatches = rep[2]; out->alive_past_lifetime = rep[3];
    if (p->d_fault) HIP_TRY(hipMemcpy(&out->fault, p->d_fault, 4, hipMemcpyDeviceToHost));
    // (capacity rows, every slot in range, none twice: alive rows and dead rows together are a permutation of the slots)
    out->ok = (rep[0] | rep[1] | rep[2] | rep[3] | out->fault) == 0u && m.alive_count <= cap;
    return HNB_OK;
}

int hnb_effect_compare(HnbEffect* a, HnbEffect* b, HnbEffectDiff* out) {
    if (!a || !b || !out) return fail(HNB_ERR_INVALID_ARG, "NULL argument");
    HnbProgram *pa = a->prog, *pb = b->prog;
    memset(out, 0, sizeof *out);
    out->first_section = -1;
    if (pa->ctx->device != pb->ctx->device) return fail(HNB_ERR_INVALID_ARG, "the two effects live on different devices");
    if (pa->dev.capacity != pb->dev.capacity || pa->attrs.size() != pb->attrs.size()) return fail(HNB_ERR_INVALID_ARG, "the two effects have different layouts");
    for (size_t i = 0; i < pa->attrs.size(); ++i)
        if (pa->attrs[i].attr != pb->attrs[i].attr || pa->attrs[i].ncomp != pb->attrs[i].ncomp) return fail(HNB_ERR_INVALID_ARG, "the two effects have different layouts");
    int rc = hnb_effect_materialise(a, pa->dev.age_cohort ? (1ull << HNB_ATTR_AGE) : 0ull);
    if (rc == HNB_OK) rc = hnb_effect_materialise(b, pb->dev.age_cohort ? (1ull << HNB_ATTR_AGE) : 0ull);
    DevMeta ma, mb;
    if (rc == HNB_OK) rc = read_meta(a, &ma);   // (synchronises the effect's stream: what follows on the default stream sees both effects complete)
    if (rc == HNB_OK) rc = read_meta(b, &mb);
    if (rc != HNB_OK) return rc;
    const uint32_t wa[8] = {ma.alive_count, ma.particle_counter, ma.write_index & 1u, ma.max_update, ma.dead_count, ma.spawned, ma.ref_write_index & 1u, ma.instance_count};
    const uint32_t wb[8] = {mb.alive_count, mb.particle_counter, mb.write_index & 1u, mb.max_update, mb.dead_count, mb.spawned, mb.ref_write_index & 1u, mb.instance_count};
    // (word 2, the list COLUMN, is where the rows live, not what they are: a ribbon effect whose sorted list is made by rotation inside k_compact
    // changes column in frames in which a radix-sorted one does not; the rows themselves are compared below, each list through its own column)
    for (int i = 0; i < 8; ++i) out->counter_diffs += (i != 2 && wa[i] != wb[i]) ? 1u : 0u;
    const uint32_t cap = pa->dev.capacity;
    const char *sa = static_cast<const char*>(a->slab), *sb = static_cast<const char*>(b->slab);
    const size_t n_sections = 2 + pa->attrs.size();
    unsigned long long* d_out = nullptr;
    HIP_TRY(hipMalloc(&d_out, n_sections * 16));
    std::vector<unsigned long long> init(n_sections * 2);
    for (size_t i = 0; i < n_sections; ++i) { init[2 * i] = 0ull; init[2 * i + 1] = ~0ull; }
    hipError_t e = hipMemcpy(d_out, init.data(), n_sections * 16, hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(d_out); return fail(HNB_ERR_HIP, "hnb_effect_compare: %s", hipGetErrorString(e)); }
    auto cmp = [&](size_t section, const void* x, const void* y, uint64_t words, uint32_t head_x = 0, uint32_t head_y = 0, uint32_t ring = 0) {
        if (!words) return;
        const uint32_t grid = (uint32_t)std::min<uint64_t>((words + 255u) / 256u, 16384u);
        k_compare_words<<<grid, 256, 0, nullptr>>>(static_cast<const uint32_t*>(x), static_cast<const uint32_t*>(y), words, d_out + 2 * section, head_x, head_y, ring);
    };
    if (ma.alive_count == mb.alive_count) {   // (else the counters already differ and the rows do not correspond)
        // (row r of either list, through its own column and behind its own head: one effect may keep a ring where the other rewrites)
        cmp(0, sa + pa->dev.alive_off[list_column(ma.write_index)], sb + pb->dev.alive_off[list_column(mb.write_index)], ma.alive_count, list_head(ma.write_index), list_head(mb.write_index), cap);
        cmp(1, sa + pa->dev.dead_off + (size_t)ma.alive_count * 4, sb + pb->dev.dead_off + (size_t)mb.alive_count * 4, cap - ma.alive_count);
    }
    for (size_t i = 0; i < pa->attrs.size(); ++i) cmp(2 + i, sa + pa->dev.attrs[i].plane_off, sb + pb->dev.attrs[i].plane_off, (uint64_t)cap * pa->attrs[i].ncomp);
    std::vector<unsigned long long> res(n_sections * 2);
    e = hipGetLastError();   // (the launches)
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(res.data(), d_out, n_sections * 16, hipMemcpyDeviceToHost);
    hipFree(d_out);
    if (e != hipSuccess) return fail(HNB_ERR_HIP, "hnb_effect_compare: %s", hipGetErrorString(e));
    out->alive_list_diffs = res[0]; out->dead_list_diffs = res[2];
    for (size_t i = 0; i < pa->attrs.size(); ++i) out->attr_diffs += res[2 * (2 + i)];
    for (size_t i = 0; i < n_sections && out->first_section < 0; ++i)
        if (res[2 * i]) { out->first_section = i < 2 ? (int32_t)i : (int32_t)(2 + pa->attrs[i - 2].attr); out->first_index = res[2 * i + 1]; }
    out->equal = out->counter_diffs == 0u && out->alive_list_diffs == 0ull && out->dead_list_diffs == 0ull && out->attr_diffs == 0ull;
    return HNB_OK;
}

int hnb_effect_sort_ribbons(HnbEffect* fx) {
    if (!fx) return fail(HNB_ERR_INVALID_ARG, "fx is NULL");
    if (!fx->prog->has_ribbons) return fail(HNB_ERR_INVALID_ARG, "the particle layout has no RIBBON_ID attribute: nothing to sort");
    // hnb_simulate sorts ribbon effects after every update, as the reference does (src/render/mod.rs:7372-7612):
    // the list read by hnb_effect_read_alive_list is already in (RIBBON_ID, AGE) order.
    return HNB_OK;
}

int hnb_program_kernel_info(HnbProgram* prog, char* buf, size_t buf_size) {
    if (!prog || !buf || !buf_size) return fail(HNB_ERR_INVALID_ARG, "NULL argument");
    std::string s = prog->kernel_info;
    if (prog->jit_pending) s += " (specialisation pending: HNB_OPT_JIT_ASYNC)";
    if (!prog->jit_log.empty()) s += "\njit log: " + prog->jit_log;
    if (!prog->dev.age_cohort && prog->dev.cull_lifetime && prog->age_cohort_mode == HNB_AGE_COHORT_AUTO && (prog->hdr.render_reads_lo >> HNB_ATTR_AGE & 1u))
        s += "\nage cohorts: off (HNB_AGE_COHORT_AUTO: the asset's render modifiers read AGE after every frame)";
    if (prog->dev.age_cohort) {   // (debug statistics: synchronises and reads the per-chunk state words of every instance)
        hipStreamSynchronize(prog->ctx->stream);
        std::vector<uint32_t> st(prog->dev.chunks_per_inst);
        size_t in_cohort = 0;
        for (const HnbEffect* fx : prog->effects) {
            if (hipMemcpy(st.data(), static_cast<const char*>(fx->slab) + prog->dev.lmin_off + (size_t)prog->dev.chunks_per_inst * 8, st.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) break;
            for (uint32_t v : st) in_cohort += v == 1u ? 1u : 0u;
        }
        s += "\nage cohorts: " + std::to_string(in_cohort) + " of " + std::to_string(prog->effects.size() * (size_t)prog->dev.chunks_per_inst) + " chunks";
        if (prog->auto_materialise) s += " (HNB_AGE_COHORT_AUTO: the asset's render modifiers read AGE, the update keeps the plane current)";
    }
    if (prog->slot_init_frames) s += "\nslot-major init (large spawns): " + std::to_string(prog->slot_init_frames) + " frames";
    if (prog->has_ribbons && prog->suffix_frames) s += "\ncasualties proven to be the list's last rows (no k_count_rows): " + std::to_string(prog->suffix_frames) + " frames";
    if (prog->has_ribbons && prog->ring_frames) s += "\nlist kept as a ring (no row rewritten): " + std::to_string(prog->ring_frames) + " frames";
    if (prog->has_ribbons && prog->sort_skipped_frames) s += "\nribbon sorts skipped in list-free frames: " + std::to_string(prog->sort_skipped_frames) + " frames";
    if (prog->has_ribbons) s += "\nribbon sorts by rotation: " + std::to_string(prog->sort_rotated_frames) + " of " + std::to_string(prog->frames_run) + " frames" + (prog->ribbon_facts.front_static ? "" : " (not eligible)");
    if (prog->merged_frames) s += "\nupdate served by a merged launch (small programs of the context share one): " + std::to_string(prog->merged_frames) + " frames";
    if (prog->unmerged_frames) s += "\nkept out of the shared launches (the loaded set module does not know this program; its own specialised kernels): " + std::to_string(prog->unmerged_frames) + " frames";
    if (prog->set_frames) s += "\n... by the context's set module (the program's specialised code behind the shared launch): " + std::to_string(prog->set_frames) + " frames";
    if (!prog->ctx->set_log.empty() && !prog->set_sig.empty()) s += "\nset module: " + prog->ctx->set_log;
    if (prog->ctx->set_failed_builds && !prog->set_sig.empty()) s += "\nset module builds that failed: " + std::to_string(prog->ctx->set_failed_builds);
    if (prog->horizon_eligible) s += "\ndeath horizons in use: " + std::to_string(prog->hz_frames) + " frames";
    s += "\nlists skipped: " + std::to_string(prog->skipped_frames) + " of " + std::to_string(prog->frames_run) + " frames" + (prog->skip_facts.eligible ? "" : " (not eligible)");
    s += "\nframe parameters (context): written by the host into device memory in " + std::to_string(prog->ctx->direct_frames) + " frames, copied in " + std::to_string(prog->ctx->copied_frames) +
         (prog->ctx->large_bar ? "" : " (the device memory is not host-visible: no large BAR)") + (prog->ctx->direct_failed ? " (host-visible slots failed their check)" : "");
    s += "\nhnb_simulate waited for the device (the frame " + std::to_string(kFrameRing) + " before still running) in " + std::to_string(prog->ctx->ring_waits) + " of " + std::to_string(prog->ctx->frame) +
         " frames, " + std::to_string(prog->ctx->ring_wait_ns / 1000) + " us in all";
    snprintf(buf, buf_size, "%s", s.c_str());
    return HNB_OK;
}

int hnb_jit_precompile(const void* blob, size_t blob_size) {
    HnbProgramHeader h;
    int rc = validate_blob(blob, blob_size, &h);
    if (rc != HNB_OK) return rc;
    const uint8_t* b = static_cast<const uint8_t*>(blob);
    std::vector<HnbAttrEntry> attrs(h.n_attrs);
    memcpy(attrs.data(), b + h.attrs_off, h.n_attrs * sizeof(HnbAttrEntry));
    const bool streams = update_is_streamable(b, h, attrs.data());
    bool aot_static = false;
    if (streams) {
        StreamLaunchFn fn = nullptr;
        const char* name = "";
        select_stream_kernel(reinterpret_cast<const Ins*>(b + h.update_off), h.update_len, &fn, &name);
        aot_static = strcmp(name, "ProgInterp") != 0;
    }
    const jit::Request rq = make_jit_request(b, h, attrs.data(), streams, aot_static, ProgramOptions());   // (the default options: what a context starts with)
    if (!rq.want_init && !rq.want_update_generic && !rq.want_update_stream) return HNB_OK;
    jit::Result res;
    if (!jit::build(rq, res)) return fail(HNB_ERR_BAD_PROGRAM, "kernel specialisation failed: %s", res.log.c_str());
    return HNB_OK;
}

int hnb_jit_precompile_set(const void* const* blobs, const size_t* blob_sizes, uint32_t n_blobs) {
    if (!blobs || !blob_sizes || n_blobs == 0u) return fail(HNB_ERR_INVALID_ARG, "NULL argument");
    struct Member { std::vector<HnbAttrEntry> attrs; std::vector<Ins> init, update; bool streams = false, cohort = false; };
    std::vector<Member> mem;
    mem.reserve(n_blobs);
    for (uint32_t i = 0; i < n_blobs; ++i) {
        HnbProgramHeader h;
        const int rc = validate_blob(blobs[i], blob_sizes[i], &h);
        if (rc != HNB_OK) return rc;
        // what can never take part in a merged launch is skipped, as hnb_simulate skips it: the wide register file, more slots than a "small"
        // program may have with a single instance, spawn events in or out
        if (std::max(h.init_regs, h.update_regs) > HNB_VM_MAX_REGS) continue;
        if ((uint64_t)h.capacity > (uint64_t)kSceneMaxChunks * kChunk || h.n_event_channels != 0u || (h.flags & HNB_PROG_READS_PARENT)) continue;
        const uint8_t* b = static_cast<const uint8_t*>(blobs[i]);
        Member m;
        m.attrs.resize(h.n_attrs);
        memcpy(m.attrs.data(), b + h.attrs_off, h.n_attrs * sizeof(HnbAttrEntry));
        m.init.assign(reinterpret_cast<const Ins*>(b + h.init_off), reinterpret_cast<const Ins*>(b + h.init_off) + h.init_len);
        m.update.assign(reinterpret_cast<const Ins*>(b + h.update_off), reinterpret_cast<const Ins*>(b + h.update_off) + h.update_len);
        m.streams = update_is_streamable(b, h, m.attrs.data());
        uint32_t dt_operand = 0;
        const ProgramOptions opt;   // (the default options: what a context starts with)
        m.cohort = cull_eligible(b, h, m.attrs.data(), m.streams, opt, &dt_operand) && age_cohort_eligible(b, h, m.attrs.data(), m.streams, opt);
        mem.push_back(std::move(m));
    }
    std::vector<jit::Request> members;
    for (const Member& m : mem)
        members.push_back(make_set_request(m.init.data(), (uint32_t)m.init.size(), m.update.data(), (uint32_t)m.update.size(), m.attrs.data(), (uint32_t)m.attrs.size(), m.streams, m.cohort));
    if (members.size() < 2u) return HNB_OK;   // (a set module serves launches that at least two programs share)
    jit::SetResult res;
    if (!jit::build_set(members, res, false)) return fail(HNB_ERR_BAD_PROGRAM, "set module: kernel specialisation failed: %s", res.log.c_str());
    return HNB_OK;
}

// ---- multi-GPU reporting (hnb_comm.h) ----------------------------------------------------------------------------------
int hnb_comm_unique_id(void* out_id) {
    if (!out_id) return fail(HNB_ERR_INVALID_ARG, "out_id is NULL");
    comm::Api& a = comm::api();
    if (!a.ok) return fail(HNB_ERR_NOT_FOUND, "RCCL is not available: %s", a.why.c_str());
    comm::ncclUniqueId id;
    const int rc = a.GetUniqueId(&id);
    if (rc != comm::kNcclSuccess) return fail(HNB_ERR_HIP, "ncclGetUniqueId failed: %s", a.GetErrorString(rc));
    memcpy(out_id, id.internal, HNB_COMM_ID_BYTES);
    return HNB_OK;
}

int hnb_comm_set_library(const char* path, uint32_t flags) {
    if (flags & ~(uint32_t)(HNB_COMM_LIB_DUPLICATE_DEVICES | HNB_COMM_LIB_SINGLE_RANK)) return fail(HNB_ERR_INVALID_ARG, "unknown flags 0x%x", flags);
    comm::LibraryChoice& ch = comm::library_choice();
    std::lock_guard<std::mutex> g(ch.mu);
    if (ch.loaded) return fail(HNB_ERR_INVALID_ARG, "the collective library is already loaded: hnb_comm_set_library must precede the first hnb_comm_* call of the process");
    ch.path = path ? path : "";
    ch.duplicate_devices = (flags & HNB_COMM_LIB_DUPLICATE_DEVICES) != 0u;
    ch.single_rank = (flags & HNB_COMM_LIB_SINGLE_RANK) != 0u;
    return HNB_OK;
}

int hnb_comm_create_local(HnbContext* const* ctxs, uint32_t n_ctx, HnbComm** out_comm) {
    if (!ctxs || !n_ctx || !out_comm) return fail(HNB_ERR_INVALID_ARG, "NULL / empty argument");
    std::vector<int> devs;
    bool distinct = true;
    for (uint32_t i = 0; i < n_ctx; ++i) {
        if (!ctxs[i]) return fail(HNB_ERR_INVALID_ARG, "context #%u is NULL", i);
        for (uint32_t j = 0; j < i; ++j) if (ctxs[j] == ctxs[i]) return fail(HNB_ERR_INVALID_ARG, "context #%u is listed twice", i);
        for (int d : devs) distinct = distinct && d != ctxs[i]->device;
        devs.push_back(ctxs[i]->device);
    }
    HnbComm* c = new HnbComm();
    c->ctxs.assign(ctxs, ctxs + n_ctx);
    c->n_ranks = n_ctx;
    // one communicator per device. RCCL refuses a device twice: contexts sharing one are reduced through the host - unless the library
    // chosen with hnb_comm_set_library says it takes duplicates (the stand-in of tests/fake_rccl, which lets the collective branch run on a one-GPU box)
    bool duplicates_ok = false, single_rank = false;   // (asked of the CHOICE, not of the library: a one-GPU host whose contexts share a device never loads librccl)
    { comm::LibraryChoice& ch = comm::library_choice(); std::lock_guard<std::mutex> g(ch.mu); duplicates_ok = ch.duplicate_devices; single_rank = ch.single_rank; }
    // (HNB_COMM_LIB_SINGLE_RANK: a communicator of ONE context goes through the library as well - ncclCommInitAll over one device, a one-rank
    // all-reduce on the context's stream: every call of the collective branch executes for real on a one-GPU machine)
    if ((n_ctx > 1 && (distinct || duplicates_ok)) || (n_ctx == 1 && single_rank)) {
        comm::Api& a = comm::api();
        if (!a.ok) { delete c; return fail(HNB_ERR_NOT_FOUND, "RCCL is not available: %s", a.why.c_str()); }
        c->comms.resize(n_ctx);
        const int rc = a.CommInitAll(c->comms.data(), (int)n_ctx, devs.data());
        if (rc != comm::kNcclSuccess) { delete c; return fail(HNB_ERR_HIP, "ncclCommInitAll failed: %s", a.GetErrorString(rc)); }
    }
    for (HnbContext* x : c->ctxs) x->comm_refs += 1;
    *out_comm = c;
    return HNB_OK;
}

int hnb_comm_create_rank(HnbContext* ctx, const void* id, uint32_t rank, uint32_t n_ranks, HnbComm** out_comm) {
    if (!ctx || !id || !out_comm || !n_ranks || rank >= n_ranks) return fail(HNB_ERR_INVALID_ARG, "bad argument");
    std::unique_ptr<HnbComm> c(new HnbComm());   // (released on every error path)
    c->ctxs.push_back(ctx);
    c->n_ranks = n_ranks; c->rank = rank;
    bool single_rank = false;
    { comm::LibraryChoice& ch = comm::library_choice(); std::lock_guard<std::mutex> g(ch.mu); single_rank = ch.single_rank; }
    if (n_ranks > 1 || single_rank) {
        comm::Api& a = comm::api();
        if (!a.ok) return fail(HNB_ERR_NOT_FOUND, "RCCL is not available: %s", a.why.c_str());
        HIP_TRY(hipSetDevice(ctx->device));
        comm::ncclUniqueId uid;
        memcpy(uid.internal, id, HNB_COMM_ID_BYTES);
        c->comms.resize(1);
        const int rc = a.CommInitRank(&c->comms[0], (int)n_ranks, uid, (int)rank);
        if (rc != comm::kNcclSuccess) return fail(HNB_ERR_HIP, "ncclCommInitRank failed: %s", a.GetErrorString(rc));
    }
    ctx->comm_refs += 1;
    *out_comm = c.release();
    return HNB_OK;
}

// What a communicator reduces through, as text: "rccl <file the symbols were resolved from> ranks=<n> local=<contexts>" or
// "host-sum ranks=.. local=.." (contexts that share a device, or a single context without HNB_COMM_LIB_SINGLE_RANK).
int hnb_comm_describe(HnbComm* c, char* buf, size_t buf_size) {
    if (!c || !buf || !buf_size) return fail(HNB_ERR_INVALID_ARG, "NULL argument");
    std::string s;
    if (!c->comms.empty()) s = "rccl " + (comm::api().resolved.empty() ? std::string("?") : comm::api().resolved);
    else s = "host-sum";
    s += " ranks=" + std::to_string(c->n_ranks) + " local=" + std::to_string(c->ctxs.size());
    snprintf(buf, buf_size, "%s", s.c_str());
    return HNB_OK;
}

// (a communicator holds its contexts: hnb_ctx_destroy refuses a context that is still part of one)
int hnb_comm_destroy(HnbComm* c) {
    if (!c) return HNB_OK;
    for (size_t i = 0; i < c->ctxs.size(); ++i) {
        hipSetDevice(c->ctxs[i]->device);
        hipStreamSynchronize(c->ctxs[i]->stream);
        if (i < c->comms.size() && c->comms[i]) comm::api().CommDestroy(c->comms[i]);
        if (i < c->d_rows.size()) { hipFree(c->d_rows[i]); hipFree(c->d_counts[i]); hipFree(c->d_totals[i]); }
        if (c->ctxs[i]->comm_refs) c->ctxs[i]->comm_refs -= 1;
    }
    delete c;
    return HNB_OK;
}

// out_totals[e] = sum over every context of every rank of the alive count of its e-th effect. `effects` is context-major,
// [local context][n_effects]; a NULL entry counts as 0 (a rank that holds no shard of that effect).
int hnb_comm_allreduce_alive(HnbComm* c, HnbEffect* const* effects, uint32_t n_effects, uint64_t* out_totals) {
    if (!c || !effects || !n_effects || !out_totals) return fail(HNB_ERR_INVALID_ARG, "NULL / empty argument");
    const size_t nl = c->ctxs.size();
    if (n_effects > c->scratch_cap) {
        const uint32_t cap = std::max(n_effects, 2u * c->scratch_cap);
        c->d_rows.resize(nl, nullptr); c->d_counts.resize(nl, nullptr); c->d_totals.resize(nl, nullptr);
        for (size_t i = 0; i < nl; ++i) {
            HIP_TRY(hipSetDevice(c->ctxs[i]->device));
            HIP_TRY(hipStreamSynchronize(c->ctxs[i]->stream));
            hipFree(c->d_rows[i]); hipFree(c->d_counts[i]); hipFree(c->d_totals[i]);
            c->d_rows[i] = nullptr; c->d_counts[i] = c->d_totals[i] = nullptr;
            HIP_TRY(hipMalloc(&c->d_rows[i], (size_t)cap * 8));
            HIP_TRY(hipMalloc(&c->d_counts[i], (size_t)cap * 8));
            HIP_TRY(hipMalloc(&c->d_totals[i], (size_t)cap * 8));
        }
        c->scratch_cap = cap;
    }
    std::vector<uint64_t> rows(n_effects);
    for (size_t i = 0; i < nl; ++i) {
        HnbContext* ctx = c->ctxs[i];
        for (uint32_t e = 0; e < n_effects; ++e) {
            HnbEffect* fx = effects[i * n_effects + e];
            if (fx && fx->prog->ctx != ctx) return fail(HNB_ERR_INVALID_ARG, "effect #%u of context #%zu belongs to another context", e, i);
            rows[e] = fx ? reinterpret_cast<uint64_t>(fx->prog->d_meta[fx->prog->parity] + fx->index) : 0ull;
        }
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(hipMemcpyAsync(c->d_rows[i], rows.data(), (size_t)n_effects * 8, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));   // (`rows` is reused for the next context; the copy is a few hundred bytes)
        comm::k_gather_alive<<<(n_effects + 255u) / 256u, 256, 0, ctx->stream>>>(c->d_rows[i], c->d_counts[i], n_effects);
        HIP_TRY(hipGetLastError());
    }
    std::vector<unsigned long long> host(n_effects);
    if (!c->comms.empty()) {   // RCCL over xGMI: one all-reduce, n_effects x u64 (latency-bound: 8 B x #effects)
        comm::Api& a = comm::api();
        int rc = a.GroupStart();
        for (size_t i = 0; i < nl && rc == comm::kNcclSuccess; ++i) {
            hipSetDevice(c->ctxs[i]->device);
            rc = a.AllReduce(c->d_counts[i], c->d_totals[i], n_effects, comm::kNcclUint64, comm::kNcclSum, c->comms[i], c->ctxs[i]->stream);
        }
        const int rc2 = a.GroupEnd();
        if (rc != comm::kNcclSuccess || rc2 != comm::kNcclSuccess) return fail(HNB_ERR_HIP, "ncclAllReduce failed: %s", a.GetErrorString(rc != comm::kNcclSuccess ? rc : rc2));
        for (size_t i = 0; i < nl; ++i) { HIP_TRY(hipSetDevice(c->ctxs[i]->device)); HIP_TRY(hipStreamSynchronize(c->ctxs[i]->stream)); }
        HIP_TRY(hipSetDevice(c->ctxs[0]->device));
        HIP_TRY(hipMemcpy(host.data(), c->d_totals[0], (size_t)n_effects * 8, hipMemcpyDeviceToHost));
        for (uint32_t e = 0; e < n_effects; ++e) out_totals[e] = host[e];
    } else {                   // contexts that share a device, or a single context: summed on the host
        for (uint32_t e = 0; e < n_effects; ++e) out_totals[e] = 0;
        for (size_t i = 0; i < nl; ++i) {
            HIP_TRY(hipSetDevice(c->ctxs[i]->device));
            HIP_TRY(hipStreamSynchronize(c->ctxs[i]->stream));
            HIP_TRY(hipMemcpy(host.data(), c->d_counts[i], (size_t)n_effects * 8, hipMemcpyDeviceToHost));
            for (uint32_t e = 0; e < n_effects; ++e) out_totals[e] += host[e];
        }
    }
    return HNB_OK;
}

__global__ void k_marker(uint32_t) {}
int hnb_ctx_profile_marker(HnbContext* ctx, uint32_t tag) {
    if (!ctx || tag == 0u || tag > 65535u) return fail(HNB_ERR_INVALID_ARG, "marker tag must be in 1..65535");
    HIP_TRY(hipSetDevice(ctx->device));
    k_marker<<<tag, 1, 0, ctx->stream>>>(tag);
    HIP_TRY(hipGetLastError());
    return HNB_OK;
}

int hnb_ctx_enable_kernel_timing(HnbContext* ctx, int enable) {
    if (!ctx) return fail(HNB_ERR_INVALID_ARG, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    recycle_timing_events(ctx);
    ctx->timing = enable > 0 ? (uint32_t)enable : 0u;
    ctx->timing_tick = 0;
    return HNB_OK;
}

static int timing_of(HnbContext* ctx, const HnbProgram* only, double* update_ms_avg, double* compact_ms_avg, double* init_ms_avg, uint32_t* frames) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    double su = 0, si = 0, sc = 0;
    size_t nu = 0, ni = 0, nc = 0;
    for (auto& t : ctx->t_compact) { if (only && t.prog != only) continue; float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, t.a, t.b)); sc += ms; ++nc; }
    for (auto& t : ctx->t_update) { if (only && t.prog != only) continue; float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, t.a, t.b)); su += ms; ++nu; }
    for (auto& t : ctx->t_init) { if (only && t.prog != only) continue; float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, t.a, t.b)); si += ms; ++ni; }
    if (compact_ms_avg) *compact_ms_avg = nc ? sc / nc : 0.0;
    if (update_ms_avg) *update_ms_avg = nu ? su / nu : 0.0;
    if (init_ms_avg) *init_ms_avg = ni ? si / ni : 0.0;
    if (frames) *frames = (uint32_t)nu;
    return HNB_OK;
}
int hnb_ctx_kernel_timing(HnbContext* ctx, double* update_ms_avg, double* compact_ms_avg, double* init_ms_avg, uint32_t* frames) {
    if (!ctx) return fail(HNB_ERR_INVALID_ARG, "ctx is NULL");
    return timing_of(ctx, nullptr, update_ms_avg, compact_ms_avg, init_ms_avg, frames);
}
int hnb_program_kernel_timing(HnbProgram* prog, double* update_ms_avg, double* compact_ms_avg, double* init_ms_avg, uint32_t* frames) {
    if (!prog) return fail(HNB_ERR_INVALID_ARG, "prog is NULL");
    return timing_of(prog->ctx, prog, update_ms_avg, compact_ms_avg, init_ms_avg, frames);
}

}  // extern "C"
