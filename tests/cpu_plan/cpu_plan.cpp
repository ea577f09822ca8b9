// Host build of bevy_hanabi_amd/csrc/hnb_plan.h - the proofs hnb_simulate's launch sequence rests on - behind a C interface, so that
// tests/test_frame_plan.py can drive every proof with synthetic program state and frame inputs WITHOUT a device (`-m "not gpu"`).
// The product includes the same header; nothing here is product code.
#include "../../bevy_hanabi_amd/csrc/hnb_plan.h"

#include <vector>

using namespace hnb::plan;

extern "C" {
struct Row { uint32_t simulated, has_parent, spawn_count, event_capacity; uint32_t ublock[8]; };   // operand k reads ublock[k]

static std::vector<InstanceFrame> frames_of(const Row* rows, uint32_t n) {
    std::vector<InstanceFrame> v(n);
    for (uint32_t i = 0; i < n; ++i) {
        v[i].simulated = rows[i].simulated != 0; v[i].has_parent = rows[i].has_parent != 0;
        v[i].spawn_count = rows[i].spawn_count; v[i].event_capacity = rows[i].event_capacity; v[i].ublock = rows[i].ublock;
    }
    return v;
}

struct Skip { SkipFacts facts; SkipHistory hist; };
void* cpl_skip_new(int eligible, uint32_t dt_operand) { Skip* s = new Skip(); s->facts.eligible = eligible != 0; s->facts.dt_operand = dt_operand; return s; }
void cpl_skip_free(void* h) { delete static_cast<Skip*>(h); }
void cpl_skip_mark_dirty(void* h) { static_cast<Skip*>(h)->hist.dirty = true; }
uint32_t cpl_skip_last_dirty(void* h) { return static_cast<Skip*>(h)->hist.last_dirty; }
int cpl_skip_step(void* h, uint32_t frame_no, const Row* rows, uint32_t n, uint32_t tag, uint32_t bound_bits, int option) {
    Skip* s = static_cast<Skip*>(h);
    const std::vector<InstanceFrame> v = frames_of(rows, n);
    return prove_skip_lists(s->facts, s->hist, frame_no, v.data(), n, SkipPublished{tag, bound_bits}, option != 0) ? 1 : 0;
}

struct Ribbon { RibbonFacts facts; RibbonHistory hist; };
void* cpl_ribbon_new(int provable, int front_static, int age_init_set, int rid_set, uint32_t tick_op, uint32_t age_op, uint32_t rid_op, uint32_t life_op) {
    Ribbon* r = new Ribbon();
    r->facts.provable = provable != 0; r->facts.front_static = front_static != 0; r->facts.age_init_set = age_init_set != 0; r->facts.rid_set = rid_set != 0;
    r->facts.tick_operand = tick_op; r->facts.age_init_operand = age_op; r->facts.rid_operand = rid_op; r->facts.life_operand = life_op;
    return r;
}
void cpl_ribbon_free(void* h) { delete static_cast<Ribbon*>(h); }
void cpl_ribbon_sorted(void* h) { static_cast<Ribbon*>(h)->hist.dirty = false; }                       // what hnb_simulate does after a sort ran
void cpl_ribbon_host_write(void* h, int wrote_age) {                                                    // what hnb_effect_write_attr does
    Ribbon* r = static_cast<Ribbon*>(h);
    r->hist.dirty = true; r->hist.front_broken = true;
    if (wrote_age) r->hist.values_broken = true;
}
// out: max_spawn, values_ok, front, head_sorted, rotate, suffix, ring (opt_ring: bit 1 of opt_suffix)
void cpl_ribbon_step(void* h, uint32_t capacity, const Row* rows, uint32_t n, int opt_skip_lists, int opt_suffix, uint32_t* out) {
    Ribbon* r = static_cast<Ribbon*>(h);
    const std::vector<InstanceFrame> v = frames_of(rows, n);
    const RibbonDecision d = prove_ribbon_order(r->facts, r->hist, capacity, v.data(), n, opt_skip_lists != 0, (opt_suffix & 1) != 0, (opt_suffix & 2) != 0);
    out[0] = d.max_spawn; out[1] = d.values_ok; out[2] = d.front; out[3] = d.head_sorted; out[4] = d.rotate; out[5] = d.suffix; out[6] = d.ring;
}

// out: use, marks
void cpl_slot_init(int eligible, int option, uint32_t capacity, uint32_t chunks_per_inst, const Row* rows, uint32_t n, uint32_t* out) {
    const std::vector<InstanceFrame> v = frames_of(rows, n);
    const SlotInitDecision d = plan_slot_init(eligible != 0, (uint32_t)option, capacity, chunks_per_inst, v.data(), n);
    out[0] = d.use; out[1] = d.marks;
}
int cpl_horizon_usable(int eligible, uint32_t dt_operand, const Row* rows, uint32_t n) {
    const std::vector<InstanceFrame> v = frames_of(rows, n);
    return horizon_usable(eligible != 0, dt_operand, v.data(), n) ? 1 : 0;
}
uint32_t cpl_init_grid(int simulated, int has_parent, uint32_t spawn_count, uint32_t event_capacity, int events_known, uint32_t known_events,
                       uint32_t capacity, uint32_t init_block, uint32_t rounds, int big_burst, uint32_t num_cus) {
    InitGridInputs in;
    in.simulated = simulated != 0; in.has_parent = has_parent != 0; in.spawn_count = spawn_count; in.event_capacity = event_capacity;
    in.events_known = events_known != 0; in.known_events = known_events;
    return size_init_grid(in, capacity, init_block, rounds, big_burst != 0, num_cus);
}
int cpl_stream_hints(uint64_t total_slots, uint32_t update_bytes_per_slot) { return use_streaming_hints(total_slots, update_bytes_per_slot) ? 1 : 0; }
int cpl_store_hints(uint64_t total_slots, uint32_t stored_bytes_per_slot) { return use_store_hints(total_slots, stored_bytes_per_slot) ? 1 : 0; }
uint32_t cpl_event_grid(uint32_t max_event_capacity, uint32_t total_chunks) { return size_event_grid(max_event_capacity, total_chunks); }

struct MergeRow { uint32_t independent, total_chunks, init_blocks, init_len, update_len, wide_file, update_streams, age_cohort; };
// out: per program {init_family, update_family} as int32
void cpl_merge(const MergeRow* rows, uint32_t n, int option, int timed, int32_t* out) {
    std::vector<MergeFacts> f(n);
    std::vector<MergeDecision> d(n);
    for (uint32_t i = 0; i < n; ++i) {
        f[i].independent = rows[i].independent != 0; f[i].total_chunks = rows[i].total_chunks; f[i].init_blocks = rows[i].init_blocks;
        f[i].init_len = rows[i].init_len; f[i].update_len = rows[i].update_len; f[i].wide_file = rows[i].wide_file != 0;
        f[i].update_streams = rows[i].update_streams != 0; f[i].age_cohort = rows[i].age_cohort != 0;
    }
    plan_merged_launches(f.data(), d.data(), n, option != 0, timed != 0);
    for (uint32_t i = 0; i < n; ++i) { out[2 * i] = d[i].init_family; out[2 * i + 1] = d[i].update_family; }
}

void cpl_partition_init_passes(const uint8_t* reads, uint32_t n, int heavy, uint8_t* side) { partition_init_passes(reads, n, heavy, side); }
// dec: [n][2] = {init_family, update_family}
void cpl_split_uncovered(const int32_t* dec, const uint8_t* has_case, const uint8_t* own, uint32_t n, int module_loaded, uint8_t* out) {
    MergeDecision d[64];
    for (uint32_t i = 0; i < n && i < 64u; ++i) { d[i].init_family = (int8_t)dec[2 * i]; d[i].update_family = (int8_t)dec[2 * i + 1]; }
    split_uncovered(d, has_case, own, n < 64u ? n : 64u, module_loaded != 0, out);
}
void* cpl_set_lookup_new() { return new SetLookupState(); }
void cpl_set_lookup_free(void* h) { delete static_cast<SetLookupState*>(h); }
void cpl_set_lookup_reset_tried(void* h) { static_cast<SetLookupState*>(h)->tried = 0; }
int cpl_set_lookup_due(void* h, int enabled, int job_running, int covered, uint32_t n_candidates, uint64_t population) {
    return set_lookup_due(*static_cast<SetLookupState*>(h), enabled != 0, job_running != 0, covered != 0, n_candidates, population) ? 1 : 0;
}
}
