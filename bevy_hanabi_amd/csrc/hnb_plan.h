// Host-side frame planning of hnb_simulate: every PROOF the frame's launch sequence rests on, as a pure function of
// (static program facts, the history the proof carries from frame to frame, this frame's inputs) -> a decision (+ the new history).
// No HIP, no device state: tests/cpu_plan drives these functions with synthetic inputs in `-m "not gpu"` (tests/test_frame_plan.py),
// hanabi_amd.hip calls them once per program and frame and stores the result in an immutable FramePlan.
//
//   prove_skip_lists      "nothing spawns and nothing can die this frame": the update rotates the counters itself, no list kernels
//   prove_ribbon_order    "the head of a ribbon effect's list is still sorted" / "this frame's spawns sort in front of everything" /
//                         "the frame's casualties are the last rows of the list": partial sort, sort by rotation, no k_count_rows
//   horizon_usable        "every tick of the frame is finite": k_count_rows may trust the row-chunk death horizons
//   plan_slot_init        a frame that spawns a large share of a program's slots runs its init slot-major
//   size_init_grid        init workgroups of one instance (HIP has no indirect dispatch: sized on the host for what the host knows)
//   size_event_grid       splits of a chunk's spawn events over workgroups
//   plan_merged_launches  which small, independent programs share the job-table launches of the frame
//   use_streaming_hints   cache policy: is the program big enough for its list traffic to be streamed past the Infinity Cache
//   partition_init_passes which init passes stay in front of the fork behind a heavy program's own init (parents first; nobody beside its update)
//   set_lookup_due        when a context looks the set module of its small programs up (once per population, after two merged frames)
//   split_uncovered       which programs a loaded set module does not know stay out of the shared launches (so that the others keep their set kernels)
//
// What each proof assumes about the device code is stated at the device side (hnb_kernels.hip.h); a wrong proof is reported by the
// kernels through HnbEffectMetadata::fault, never silently.
#pragma once
#include <cstdint>
#include <cstring>

namespace hnb {
namespace plan {

// One instance's inputs of the frame being enqueued.
struct InstanceFrame {
    bool simulated = true;          // hnb_effect_set_simulated
    bool has_parent = false;        // spawned by its parent's events: the CPU spawn count is unused, the event count lives on the device
    uint32_t spawn_count = 0;       // hnb_effect_set_frame
    uint32_t event_capacity = 0;    // has_parent: capacity of the event buffer the init pass consumes
    const uint32_t* ublock = nullptr;  // the instance's parameter block: the uniform stream evaluated for this frame (n_uregs words)
};
inline uint32_t uword(const InstanceFrame& f, uint32_t decoded_operand) { return f.ublock[decoded_operand & 0xffu]; }
inline float as_float(uint32_t bits) { float f; std::memcpy(&f, &bits, 4); return f; }
inline bool nonneg_not_nan(uint32_t float_bits) { return float_bits <= 0x7f800000u; }   // +0 .. +inf: sign clear, not a NaN

// ---- list-free frames ---------------------------------------------------------------------------------------------------------------
// For effects whose particles only die of old age the update publishes {frame, R} to host-mapped memory: R = a lower bound of the
// remaining life of every particle alive after `frame` (SlotArgs::safe_words). A later frame F needs no list kernels if nothing spawns
// in it and the ticks accumulated over frames frame+1 .. F stay below R. The bound must postdate everything that could have invalidated
// it (a spawn, a host write, a thawed instance, an unknown tick: `last_dirty`) and is trusted for at most 64 frames (the f32 ages
// accumulate one rounding per frame; 64 of them stay inside the 1e-5 * lifetime margin the device subtracts).
struct SkipFacts {
    bool eligible = false;          // streamable, lifetime-culled, no kill modifier, no spawn events in or out
    uint32_t dt_operand = 0;        // decoded U operand of the update stream's AGE_TICK
};
struct SkipHistory {
    double cum_tick[128] = {};      // sum of the ticks of frames 0 .. F, ring indexed by F & 127
    uint32_t last_dirty = 0;
    bool dirty = true;              // something outside the frame inputs changed the particles since the last frame
};
struct SkipPublished { uint32_t tag = 0xffffffffu, bound_bits = 0; };   // tag 0xffffffff: nothing published yet
constexpr uint32_t kSkipMaxAge = 64;

inline bool prove_skip_lists(const SkipFacts& facts, SkipHistory& h, uint32_t frame_no, const InstanceFrame* inst, uint32_t n, SkipPublished pub, bool option_enabled) {
    bool any_spawn = false, any_parent = false, tick_known = facts.eligible, have_tick = false;
    uint32_t tick_bits = 0;
    for (uint32_t i = 0; i < n && facts.eligible; ++i) {
        any_parent = any_parent || inst[i].has_parent;
        if (!inst[i].simulated) continue;
        any_spawn = any_spawn || inst[i].spawn_count != 0u;
        const uint32_t t = uword(inst[i], facts.dt_operand);
        if (!have_tick) { tick_bits = t; have_tick = true; }
        else if (t != tick_bits) tick_known = false;       // instances ticking differently: no common clock
    }
    const float tick = as_float(tick_bits);
    if (!(tick >= 0.0f)) tick_known = false;                // negative or NaN ticks: no statement about the future
    if (any_spawn || any_parent || h.dirty || !tick_known) h.last_dirty = frame_no;   // only a bound computed in this frame or later covers it
    h.dirty = false;
    h.cum_tick[frame_no & 127u] = (frame_no ? h.cum_tick[(frame_no - 1u) & 127u] : 0.0) + (tick_known ? (double)tick : 0.0);
    if (!(facts.eligible && option_enabled && !any_spawn && !any_parent && tick_known)) return false;
    if (pub.tag == 0xffffffffu || !(pub.tag < frame_no) || pub.tag < h.last_dirty || frame_no - pub.tag > kSkipMaxAge || !nonneg_not_nan(pub.bound_bits)) return false;
    const double ticks = h.cum_tick[frame_no & 127u] - h.cum_tick[pub.tag & 127u];   // frames tag + 1 .. F
    return ticks * (1.0 + 1e-6) < (double)as_float(pub.bound_bits);
}

// ---- ribbon effects: what the host can prove about the order of the alive list ---------------------------------------------------------
// The list of a ribbon effect is kept sorted by (RIBBON_ID, AGE bits). It is last frame's sorted list minus the casualties (stable
// compaction), every age advanced by the same tick, plus this frame's spawns at the end.
//   values_ok   every tick and initial age so far was >= +0 and not NaN: ages stay non-negative, key order == age order, the HEAD of the
//               list (everything but this frame's spawns) is still sorted: the radix range is at most the spawns. A violation is STICKY
//               (a negative age outlives the frame that made it and changes key order when it crosses zero later).
//   front       this frame's spawns sort IN FRONT of every older particle: one RIBBON_ID for every particle the effect ever had (one
//               uniform value that never changed, or never set), spawns start at AGE +0 and are ticked once, so the spawns' keys are
//               (rid, tick_now); every older particle has age >= fl(t_g + tick_now) >= fl(min_tick + tick_now) for the smallest tick of
//               any earlier frame: the host evaluates that f32 sum and asks for it to be > tick_now. And every spawn must survive its
//               first frame (tick < lifetime: the rotation moves exactly `spawned` rows).
//   suffix      in addition every particle has the same lifetime (one uniform value that never changed): age + tick < lifetime is
//               monotone along the age-ordered list, the casualties are its LAST rows (CompactArgs::suffix_dead; checked on the device).
struct RibbonFacts {
    bool provable = false;              // static: the update only advances AGE through one AGE_TICK, the init sets AGE from a uniform value or not at all
    bool front_static = false;          // static: + RIBBON_ID and LIFETIME set at most once by the init, from uniform values; only old age kills
    bool age_init_set = false, rid_set = false;
    uint32_t tick_operand = 0, age_init_operand = 0, rid_operand = 0, life_operand = 0;   // decoded U operands
};
struct RibbonHistory {
    bool values_broken = false;         // sticky
    bool front_broken = false;          // sticky: a negative / NaN tick, a second RIBBON_ID value, a host write
    bool rid_known = false, life_known = false, life_changed = false;
    uint32_t rid_value = 0, life_value = 0;
    float min_tick = __builtin_inff();  // smallest tick of any frame so far
    bool dirty = true;                  // a host write (or nothing yet) since the last sort: the whole list is the range
};
struct RibbonDecision {
    uint32_t max_spawn = 0;             // largest spawn request of an instance this frame (bounds the radix range)
    bool values_ok = false, front = false;
    bool head_sorted = false;           // values_ok + no host write + the option: the sort may be partial / skipped without spawns
    bool rotate = false;                // k_compact writes [spawns | older ones]: no sort kernel
    bool suffix = false;                // k_count_rows does not run
    bool ring = false;                  // both hold (or nothing spawns): the list is kept as a ring - k_init writes the spawns in front of the head, k_compact
                                        // moves the head and drops the last rows; nothing is rewritten (hnb_kernels.hip.h "Ring lists")
};
inline RibbonDecision prove_ribbon_order(const RibbonFacts& facts, RibbonHistory& h, uint32_t capacity, const InstanceFrame* inst, uint32_t n,
                                         bool option_skip_lists, bool option_suffix, bool option_ring = false) {
    RibbonDecision d;
    bool ok = facts.provable;
    for (uint32_t i = 0; i < n; ++i) {
        if (!inst[i].simulated) continue;
        const uint32_t req = inst[i].has_parent ? inst[i].event_capacity : inst[i].spawn_count;
        const uint32_t capped = req < capacity ? req : capacity;
        d.max_spawn = capped > d.max_spawn ? capped : d.max_spawn;
        if (!ok) continue;
        const uint32_t tick_bits = uword(inst[i], facts.tick_operand);
        const uint32_t age_bits = facts.age_init_set ? uword(inst[i], facts.age_init_operand) : 0u;
        ok = nonneg_not_nan(tick_bits) && nonneg_not_nan(age_bits);
    }
    if (facts.provable && !ok) h.values_broken = true;
    d.values_ok = ok && !h.values_broken;
    bool front = facts.front_static && d.values_ok && !h.front_broken;
    if (facts.front_static) {
        uint32_t tick_now_bits = 0;
        float frame_min_tick = __builtin_inff();   // (the smallest tick of ANY simulated instance enters min_tick)
        bool have = false;
        for (uint32_t i = 0; i < n; ++i) {
            if (!inst[i].simulated) continue;
            const uint32_t tick_bits = uword(inst[i], facts.tick_operand);
            const uint32_t age_bits = facts.age_init_set ? uword(inst[i], facts.age_init_operand) : 0u;
            const uint32_t rid_bits = facts.rid_set ? uword(inst[i], facts.rid_operand) : 0u;
            if (!nonneg_not_nan(tick_bits)) h.front_broken = true;    // ages are no longer what the proof assumes
            if (!h.rid_known) { h.rid_known = true; h.rid_value = rid_bits; }
            else if (rid_bits != h.rid_value) h.front_broken = true;
            if (age_bits != 0u) front = false;                         // spawns that do not start at +0 this frame
            const uint32_t life_bits = uword(inst[i], facts.life_operand);
            if (!h.life_known) { h.life_known = true; h.life_value = life_bits; }
            else if (life_bits != h.life_value) h.life_changed = true; // older particles keep the lifetime they were born with
            const float life = as_float(life_bits), t = as_float(tick_bits);
            if (!(life_bits < 0x7f800000u && t < life)) front = false; // a spawn would die in its first frame and miss the list
            if (t == t) frame_min_tick = t < frame_min_tick ? t : frame_min_tick;
            if (!have) { tick_now_bits = tick_bits; have = true; }
            else if (tick_bits != tick_now_bits) front = false;
        }
        if (h.front_broken || !have) front = false;
        if (front) {
            const float tick_now = as_float(tick_now_bits);
            volatile float bound = h.min_tick + tick_now;              // f32, as the device adds (no contraction, no excess precision)
            front = tick_now > 0.0f && bound > tick_now;
        }
        if (have) h.min_tick = frame_min_tick < h.min_tick ? frame_min_tick : h.min_tick;
    }
    d.front = front;
    d.head_sorted = facts.provable && d.values_ok && !h.dirty && option_skip_lists;
    d.rotate = front && d.head_sorted && d.max_spawn > 0u;
    d.suffix = front && d.head_sorted && !h.life_changed && option_suffix;
    d.ring = option_ring && d.suffix && (d.rotate || d.max_spawn == 0u);
    return d;
}

// ---- death horizons: usable iff every simulated instance's tick is finite this frame ----------------------------------------------------
inline bool horizon_usable(bool eligible, uint32_t dt_operand, const InstanceFrame* inst, uint32_t n) {
    if (!eligible) return false;
    for (uint32_t i = 0; i < n; ++i)
        if (inst[i].simulated && (uword(inst[i], dt_operand) & 0x7f800000u) == 0x7f800000u) return false;   // inf / NaN
    return true;
}

// ---- grids ------------------------------------------------------------------------------------------------------------------------------
// Init workgroups of one instance. `request`: the CPU spawner's count, or - for an effect with a parent - what the host knows about the
// event count: the buffer's capacity, or last frame's count if its host-mapped copy has arrived (`known_events`, exact for zero).
// The kernel reads the true count itself and strides: this only decides how many workgroups are launched.
struct InitGridInputs {
    bool simulated = true, has_parent = false;
    uint32_t spawn_count = 0, event_capacity = 0;
    bool events_known = false;          // the copy in host memory is last frame's
    uint32_t known_events = 0;
};
inline uint32_t size_init_grid(const InitGridInputs& in, uint32_t capacity, uint32_t init_block, uint32_t rounds_if_big, bool big_burst, uint32_t num_cus) {
    uint32_t request = in.simulated ? in.spawn_count : 0u;
    if (in.has_parent && in.simulated) {
        request = in.event_capacity;
        if (in.events_known) request = in.known_events < in.event_capacity ? in.known_events : in.event_capacity;
    }
    const uint32_t capped = request < capacity ? request : capacity;     // never more init workgroups than the capacity allows
    uint32_t blocks = (uint32_t)(((uint64_t)capped + init_block - 1u) / init_block);
    if (big_burst) blocks = (blocks + rounds_if_big - 1u) / rounds_if_big;    // a large burst: several groups of spawns per workgroup
    if (in.has_parent) blocks = blocks < num_cus * 8u ? blocks : num_cus * 8u; // event-driven: a bounded grid that strides
    return blocks;
}
// gridDim.y of k_emit_events: the events of a chunk are dealt over several workgroups, 16,384 events per split, sized for the largest
// event buffer that listens. (Round 6: 2048 per split - eight times the workgroups for the rocket effect of firework.rs, whose k_emit_events is the
// longest link of the chain that runs beside the trails' update - made c2_events 2 % slower, 0.357 -> 0.363 ms in three rounds on one box: every split
// repeats the chunk's prefix and scan. profiles/r06ae_ab_event_splits.log, r06ad_c2_events_timeline.log.)
inline uint32_t size_event_grid(uint32_t max_event_capacity, uint32_t total_chunks) {
    const uint32_t per_chunk = max_event_capacity / (total_chunks ? total_chunks : 1u);
    const uint32_t splits = (per_chunk + 16383u) / 16384u;
    return splits < 1u ? 1u : (splits > 64u ? 64u : splits);
}

// ---- slot-major init (hnb_kernels.hip.h "slot-major init: large spawns") -----------------------------------------------------------------------
// k_init_slots walks the SLOTS of every instance and is correct for any spawn; it pays when the frame spawns a large share of the program's slots
// (a burst, a re-burst after a die-off: the row-major k_init then scatters every plane store). Eligible programs: the init does not read
// PARTICLE_COUNTER (the rank of a spawn is not known slot-major), no ribbons (ring lists append in front), no parent particle. `marks`: some
// instance may spawn fewer particles than it has free slots - only a request of `capacity` or more provably takes them all - so k_spawn_mark runs
// first (it decides per instance from the device counters and leaves instances that do fill up alone).
struct SlotInitDecision { bool use = false, marks = false; };
constexpr uint32_t kSlotInitMinChunks = 17;   // (programs of <= 16 chunks are candidates for the merged launches)
inline SlotInitDecision plan_slot_init(bool eligible, uint32_t option, uint32_t capacity, uint32_t chunks_per_inst, const InstanceFrame* inst, uint32_t n) {
    SlotInitDecision d;   // option (HNB_OPT_SLOT_INIT): 0 never, 1 where it pays, 2 wherever it is correct (any size, any spawn: tests and A/B runs)
    if (!eligible || option == 0u || n == 0u || (option == 1u && (uint64_t)n * chunks_per_inst < kSlotInitMinChunks)) return d;
    uint64_t spawns = 0;
    bool partial = false;
    for (uint32_t i = 0; i < n; ++i) {
        if (inst[i].has_parent) return d;               // GPU-spawned: the init reads the parent particle of event i
        if (!inst[i].simulated || inst[i].spawn_count == 0u) continue;
        spawns += inst[i].spawn_count < capacity ? inst[i].spawn_count : capacity;
        partial = partial || inst[i].spawn_count < capacity;
    }
    d.use = spawns != 0u && (option >= 2u || spawns * 8u >= (uint64_t)n * capacity);
    d.marks = d.use && partial;
    return d;
}

// ---- merged launches of small programs ----------------------------------------------------------------------------------------------------
// A program is served by the frame's job-table launches (k_init_jobs / k_update_jobs / k_update_generic_wide_jobs: INTERPRETER
// instantiations) if it is small this frame, independent of every other program (no spawn events in or out, no parent), its pass is
// short (the interpreter's latency is set by the longest program of the launch) and at least two programs share the launch.
struct MergeFacts {
    bool independent = false;       // no event channels, does not read a parent, no instance has a parent, >= 1 instance
    uint32_t total_chunks = 0;      // instances x chunks per instance
    uint32_t init_blocks = 0;       // this frame
    uint32_t init_len = 0, update_len = 0;
    bool wide_file = false, update_streams = false, age_cohort = false;
};
enum UpdateFamily : int8_t { kNotMerged = -1, kStream = 0, kStreamCohort = 1, kGeneric = 2, kGenericWide = 3 };
struct MergeDecision { int8_t init_family = -1; /* -1, 0 narrow file, 1 wide */ int8_t update_family = kNotMerged; };
struct MergeLimits { uint32_t max_chunks = 16, max_init_blocks = 64, max_code_len = 64; };

inline void plan_merged_launches(const MergeFacts* progs, MergeDecision* out, uint32_t n, bool option_enabled, bool timed_frame, MergeLimits lim = MergeLimits()) {
    for (uint32_t i = 0; i < n; ++i) out[i] = MergeDecision();
    if (!option_enabled || timed_frame || n < 2u) return;   // (timed frames keep one launch per program: the timings stay attributable)
    auto small = [&](const MergeFacts& p) { return p.independent && p.total_chunks <= lim.max_chunks; };
    auto init_family = [&](const MergeFacts& p) -> int { return small(p) && p.init_blocks != 0u && p.init_blocks <= lim.max_init_blocks && p.init_len <= lim.max_code_len ? (p.wide_file ? 1 : 0) : -1; };
    auto update_family = [&](const MergeFacts& p) -> int {
        if (!small(p) || p.update_len > lim.max_code_len) return kNotMerged;
        if (p.update_streams) return p.age_cohort ? kStreamCohort : kStream;
        return p.wide_file ? kGenericWide : kGeneric;
    };
    uint32_t n_init[2] = {0, 0}, n_upd[4] = {0, 0, 0, 0};
    for (uint32_t i = 0; i < n; ++i) {
        const int fi = init_family(progs[i]), fu = update_family(progs[i]);
        if (fi >= 0) n_init[fi] += 1;
        if (fu >= 0) n_upd[fu] += 1;
    }
    // k_update_jobs serves the two streaming families and the narrow generic one in ONE launch: worth it from two programs in total
    const bool shared_update = n_upd[kStream] + n_upd[kStreamCohort] + n_upd[kGeneric] >= 2u;
    for (uint32_t i = 0; i < n; ++i) {
        const int fi = init_family(progs[i]), fu = update_family(progs[i]);
        if (fi >= 0 && n_init[fi] >= 2u) out[i].init_family = (int8_t)fi;
        if (fu == kGenericWide ? n_upd[kGenericWide] >= 2u : (fu >= 0 && shared_update)) out[i].update_family = (int8_t)fu;
    }
}

// ---- a scene that has outgrown its set module (fill_merge_jobs) -----------------------------------------------------------------------------------
// A shared launch runs on the set kernels only if EVERY job of it has a case in the loaded module; one program without a case - an effect that
// joined a running scene - used to send the whole scene back to the byte-code interpreters until a module for the new set existed (65 s of hiprtc for
// 27 programs). The reference compiles one pipeline per effect and an added effect costs one more pipeline, nobody else's
// (src/render/mod.rs:3852-3900). Counterpart: a few newcomers with kernels of their own (every program is specialised at creation: hnb_jit.h) are LEFT
// OUT of the shared launches - their own launches, their own specialised code, from their first frame - while the covered programs stay on the set
// kernels; the module of the grown set is looked up / compiled beside the frames as before and takes everybody back in when it is there.
// Only where the covered programs are the clear majority (>= 4x): a context whose population changed wholesale keeps sharing its launches on the
// interpreters, which beats one launch per program. out[i] = 1: program i stays out of the shared launches this frame.
inline void split_uncovered(const MergeDecision* dec, const uint8_t* has_case, const uint8_t* has_own_kernels, uint32_t n, bool module_loaded, uint8_t* out) {
    for (uint32_t i = 0; i < n; ++i) out[i] = 0u;
    if (!module_loaded) return;
    auto in_set_launch = [&](uint32_t i) { return dec[i].init_family == 0 || (dec[i].update_family >= 0 && dec[i].update_family != kGenericWide); };
    uint32_t covered = 0, uncovered = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (!in_set_launch(i)) continue;
        if (has_case[i]) covered += 1; else uncovered += 1;
    }
    if (uncovered == 0u || covered < 2u || covered < 4u * uncovered) return;
    for (uint32_t i = 0; i < n; ++i)
        if (in_set_launch(i) && !has_case[i] && has_own_kernels[i]) out[i] = 1u;
}

// ---- which init passes stay in front of the fork (enqueue_init_passes) -------------------------------------------------------------------------
// With a heavy program the frame forks behind its own init pass. `reads[c * n + p]`: an instance of program c has its parent in program p (the init
// of c reads p's particles and must follow p's init; p's UPDATE must not run beside it). In front of the fork stay: the heavy program, the programs
// it reads from (transitively: parents first), the programs that read ITS particles, and - again parents first - whatever those read from. Every
// other init pass goes to the side stream. side[i] = 1: behind the fork.
inline void partition_init_passes(const uint8_t* reads, uint32_t n, int heavy, uint8_t* side) {
    for (uint32_t i = 0; i < n; ++i) side[i] = heavy >= 0 ? 1u : 0u;
    if (heavy < 0) return;
    side[heavy] = 0u;
    for (uint32_t c = 0; c < n; ++c)
        if (reads[c * n + (uint32_t)heavy]) side[c] = 0u;          // would race with the heavy update
    for (bool changed = true; changed;) {                            // parents first, also across the fork
        changed = false;
        for (uint32_t c = 0; c < n; ++c)
            if (!side[c])
                for (uint32_t p = 0; p < n; ++p)
                    if (side[p] && reads[c * n + p]) { side[p] = 0u; changed = true; }
    }
}

// ---- set modules: when a context looks for (or builds) the module of its small programs (refresh_set_module) ----------------------------------------
// covered: every candidate has a case in the loaded module; population: hash over the candidates' signatures (never 0). A population is looked up
// once, and only after it has stood for two merged frames; nothing is looked up while a background compilation runs.
struct SetLookupState { uint64_t seen = 0, tried = 0; };
inline bool set_lookup_due(SetLookupState& st, bool enabled, bool job_running, bool covered, uint32_t n_candidates, uint64_t population) {
    if (!enabled || job_running || covered || n_candidates < 2u) return false;
    if (population == st.tried) return false;
    if (population != st.seen) { st.seen = population; return false; }
    st.tried = population;
    return true;
}

// ---- cache policy ----------------------------------------------------------------------------------------------------------------------------
// Streaming hints (nontemporal list traffic, nontemporal loads of the update's read-only planes: hnb_kernels.hip.h "cache policy of streamed
// data") pay when one frame of the program touches more than the 256 MiB Infinity Cache holds - a 16.7M-particle effect moves a gigabyte per
// frame, and what the cache keeps of it should be the planes the next frame starts on, not the lists - and cost a small effect the cache hits
// it lives on (C5, 4.19M particles, 35 MB per frame: 12 % slower with the hints; profiles/r04i_ab_nt3.log).
constexpr uint64_t kStreamHintBytes = 256ull << 20;
inline bool use_streaming_hints(uint64_t total_slots, uint32_t update_bytes_per_slot) {
    return total_slots * (uint64_t)(update_bytes_per_slot + 8u) > kStreamHintBytes;   // attributes the update loads + stores, + the list rows
}

// ... and the update's own plane STORES are streamed only when the planes it writes exceed the cache by half: what fits (force_field.rs at
// 8.4M particles writes 235 MB) is found there by the next frame's reversed walk, and the hint would throw that away (C3: +22 %).
constexpr uint64_t kStoreHintBytes = 384ull << 20;
inline bool use_store_hints(uint64_t total_slots, uint32_t stored_bytes_per_slot) { return total_slots * (uint64_t)stored_bytes_per_slot > kStoreHintBytes; }

// ---- the plan of one program for one frame ------------------------------------------------------------------------------------------------
struct FramePlan {
    bool skip_lists = false;        // proven: no spawn, no casualty - the update kernel is the program's only launch
    bool lists = true;              // k_count_rows / k_compact (or the slot-order kernels) run
    bool lists_merged = false;      // ... inside the context's multi-program list launches
    bool hz_use = false;            // k_count_rows may trust the death horizons
    bool stream_hint = false;       // the frame touches more than the Infinity Cache holds: list traffic and read-only planes are streamed
    bool store_hint = false;        // ... and the planes the update writes exceed it by half: its plane stores are streamed too
    RibbonDecision ribbon;
    bool independent = false;
    MergeDecision merge;
    uint32_t init_blocks = 0;
    SlotInitDecision slot_init;     // the init pass walks the slots (k_init_slots), behind k_spawn_mark where a partial re-fill is possible
};

}  // namespace plan
}  // namespace hnb
