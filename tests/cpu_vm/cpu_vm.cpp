// TEST-ONLY host harness: runs a lowered HnbProgram blob through the product's program
// interpreters (bevy_hanabi_amd/csrc/hnb_vm.h compiled for the host) with the serial frame
// semantics of the kernels. It lets the CPU test suite check lowering + interpreter
// semantics against the oracle without a GPU. It is NOT part of the package and is never a
// fallback for the HIP path (the C ABI fails with HNB_ERR_NO_DEVICE when no GPU exists).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#include "../../bevy_hanabi_amd/csrc/hnb_vm.h"

using namespace hnb;

struct CpuVm {
    HnbProgramHeader h;
    std::vector<HnbAttrEntry> attrs;
    std::vector<HnbPropEntry> props;
    std::vector<Ins> ucode, icode, ucode_update;
    std::vector<uint8_t> slab;                  // attribute planes, laid out like the runtime's slab
    std::vector<AttrDesc> adesc;
    std::vector<uint32_t> list[2], dead, prop_words, ublock;
    uint32_t alive = 0, counter = 0, write_index = 0, slot_base = 0, max_update = 0, dead_count = 0, spawned = 0;
    bool streamable = true;
};

// force_generic: run the update stream through vm_run even when it is streamable
template <class FILE_T>
static void step_impl(CpuVm* v, const float* sim, uint32_t spawn_count, uint32_t seed, const float* xf_in, int force_generic) {
    static const float identity[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    const float* xf = xf_in ? xf_in : identity;
    const HnbProgramHeader& h = v->h;
    uniform_run(v->ucode.data(), (uint32_t)v->ucode.size(), v->prop_words.data(), sim, v->ublock.data(), h.n_uregs);
    VmUniforms U;
    U.u = v->ublock.data();
    U.xf = xf;
    // init
    const uint32_t alive0 = v->alive, max_spawn = h.capacity - alive0;
    const uint32_t n_spawn = spawn_count < max_spawn ? spawn_count : max_spawn;
    const uint32_t wi = v->write_index;
    for (uint32_t i = 0; i < n_spawn; ++i) {
        const uint32_t slot = v->dead[alive0 + i];
        VmState<FILE_T> S;
        S.r = FILE_T{};
        S.pindex = slot + v->slot_base;
        S.seed = pcg_hash(S.pindex ^ seed);
        S.pcounter = v->counter + i;
        S.alive = true;
        VmAttrIO io;
        io.slab = reinterpret_cast<char*>(v->slab.data()); io.attrs = v->adesc.data(); io.slot = slot;
        for (size_t a = 0; a < v->attrs.size(); ++a)
            if (v->attrs[a].reg == HNB_REG_NONE)
                for (uint32_t c = 0; c < v->attrs[a].ncomp; ++c) vm_attr_ptr(io, (uint32_t)a)[c] = 0u;
        vm_run<true, false>(v->icode.data(), (uint32_t)v->icode.size(), S, U, nullptr, nullptr, io);
        v->list[wi][alive0 + i] = slot;
        for (size_t a = 0; a < v->attrs.size(); ++a)
            if (v->attrs[a].reg != HNB_REG_NONE)
                for (uint32_t c = 0; c < v->attrs[a].ncomp; ++c) vm_attr_ptr(io, (uint32_t)a)[c] = S.r[v->attrs[a].reg + c];
    }
    const uint32_t n = alive0 + n_spawn;
    // update + stable compaction
    uint32_t survivors = 0, casualties = 0;
    std::vector<uint32_t>& rd = v->list[wi];
    std::vector<uint32_t>& wr = v->list[wi ^ 1u];
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t slot = rd[i];
        bool alive = true;
        VmAttrIO io;
        io.slab = reinterpret_cast<char*>(v->slab.data()); io.attrs = v->adesc.data(); io.slot = slot;
        if (v->streamable && !force_generic) {
            Pinned<1> X;
            X.pos[0] = V3{0, 0, 0}; X.vel[0] = V3{0, 0, 0}; X.age[0] = 0; X.lifetime[0] = 0; X.alive[0] = true;
            for (size_t a = 0; a < v->attrs.size(); ++a) {
                const HnbAttrEntry& at = v->attrs[a];
                if (!(at.update_flags & HNB_ATTR_UPD_LOAD)) continue;
                const uint32_t* p = vm_attr_ptr(io, (uint32_t)a);
                if (at.reg == HNB_REG_POSITION) X.pos[0] = V3{u2f(p[0]), u2f(p[1]), u2f(p[2])};
                else if (at.reg == HNB_REG_VELOCITY) X.vel[0] = V3{u2f(p[0]), u2f(p[1]), u2f(p[2])};
                else if (at.reg == HNB_REG_AGE) X.age[0] = u2f(p[0]);
                else if (at.reg == HNB_REG_LIFETIME) X.lifetime[0] = u2f(p[0]);
            }
            fast_run<1, true>(v->ucode_update.data(), (uint32_t)v->ucode_update.size(), X, U);
            for (size_t a = 0; a < v->attrs.size(); ++a) {
                const HnbAttrEntry& at = v->attrs[a];
                if (!(at.update_flags & HNB_ATTR_UPD_STORE)) continue;
                uint32_t* p = vm_attr_ptr(io, (uint32_t)a);
                if (at.reg == HNB_REG_POSITION) { p[0] = f2u(X.pos[0].x); p[1] = f2u(X.pos[0].y); p[2] = f2u(X.pos[0].z); }
                else if (at.reg == HNB_REG_VELOCITY) { p[0] = f2u(X.vel[0].x); p[1] = f2u(X.vel[0].y); p[2] = f2u(X.vel[0].z); }
                else if (at.reg == HNB_REG_AGE) p[0] = f2u(X.age[0]);
                else if (at.reg == HNB_REG_LIFETIME) p[0] = f2u(X.lifetime[0]);
            }
            alive = X.alive[0];
        } else {
            VmState<FILE_T> S;
            S.r = FILE_T{};
            for (size_t a = 0; a < v->attrs.size(); ++a) {
                const HnbAttrEntry& at = v->attrs[a];
                if (!(at.update_flags & HNB_ATTR_UPD_LOAD) || at.reg == HNB_REG_NONE) continue;
                for (uint32_t c = 0; c < at.ncomp; ++c) S.r[at.reg + c] = vm_attr_ptr(io, (uint32_t)a)[c];
            }
            S.pindex = slot + v->slot_base;
            S.seed = pcg_hash(S.pindex ^ seed);
            S.pcounter = 0;
            S.alive = true;
            vm_run<true, false>(v->ucode_update.data(), (uint32_t)v->ucode_update.size(), S, U, nullptr, nullptr, io);
            for (size_t a = 0; a < v->attrs.size(); ++a) {
                const HnbAttrEntry& at = v->attrs[a];
                if (!(at.update_flags & HNB_ATTR_UPD_STORE) || at.reg == HNB_REG_NONE) continue;
                for (uint32_t c = 0; c < at.ncomp; ++c) vm_attr_ptr(io, (uint32_t)a)[c] = S.r[at.reg + c];
            }
            alive = S.alive;
        }
        if (alive) wr[survivors++] = slot;
        else { v->dead[n - 1u - casualties] = slot; ++casualties; }
    }
    v->alive = survivors;
    if (v->h.flags & HNB_PROG_HAS_RIBBONS) {  // ribbon sort: stable by (RIBBON_ID, AGE bits) (vfx_sort*.wgsl)
        const uint32_t *rid = nullptr, *age = nullptr;
        for (size_t a = 0; a < v->attrs.size(); ++a) {
            if (v->attrs[a].attr == HNB_ATTR_RIBBON_ID) rid = reinterpret_cast<const uint32_t*>(v->slab.data() + v->adesc[a].plane_off);
            if (v->attrs[a].attr == HNB_ATTR_AGE) age = reinterpret_cast<const uint32_t*>(v->slab.data() + v->adesc[a].plane_off);
        }
        std::stable_sort(wr.begin(), wr.begin() + survivors, [&](uint32_t x, uint32_t y) {
            const uint64_t kx = ((uint64_t)rid[x] << 32) | (age ? age[x] : 0u), ky = ((uint64_t)rid[y] << 32) | (age ? age[y] : 0u);
            return kx < ky;
        });
    }
    v->counter += n_spawn;
    v->write_index = wi ^ 1u;
    v->max_update = n;
    v->dead_count = casualties;
    v->spawned = n_spawn;
}


extern "C" {

CpuVm* cvm_create(const uint8_t* blob, size_t size, uint32_t slot_base) {
    if (size < sizeof(HnbProgramHeader)) return nullptr;
    CpuVm* v = new CpuVm();
    memcpy(&v->h, blob, sizeof v->h);
    const HnbProgramHeader& h = v->h;
    if (h.magic != HNB_PROGRAM_MAGIC || h.total_size != size) { delete v; return nullptr; }
    v->attrs.resize(h.n_attrs);
    memcpy(v->attrs.data(), blob + h.attrs_off, h.n_attrs * sizeof(HnbAttrEntry));
    v->props.resize(h.n_props);
    if (h.n_props) memcpy(v->props.data(), blob + h.props_off, h.n_props * sizeof(HnbPropEntry));
    v->ucode.resize(h.uniform_len); if (h.uniform_len) memcpy(v->ucode.data(), blob + h.uniform_off, h.uniform_len * 8);
    v->icode.resize(h.init_len); if (h.init_len) memcpy(v->icode.data(), blob + h.init_off, h.init_len * 8);
    v->ucode_update.resize(h.update_len); if (h.update_len) memcpy(v->ucode_update.data(), blob + h.update_off, h.update_len * 8);
    size_t off = 0;
    for (auto& a : v->attrs) {
        AttrDesc d;
        d.plane_off = hnb::soff_of(off); d.ncomp = a.ncomp; d.reg = a.reg; d.upd_flags = a.update_flags; d.pad = 0;
        v->adesc.push_back(d);
        off += ((size_t)h.capacity * a.ncomp * 4 + 255) / 256 * 256;
    }
    v->slab.assign(off, 0);
    v->list[0].assign(h.capacity, 0); v->list[1].assign(h.capacity, 0);
    v->dead.resize(h.capacity);
    for (uint32_t i = 0; i < h.capacity; ++i) v->dead[i] = i;
    v->prop_words.assign(h.prop_words, 0);
    for (auto& p : v->props) for (uint32_t c = 0; c < p.ncomp; ++c) v->prop_words[p.word_offset + c] = p.default_bits[c];
    v->ublock.assign(h.n_uregs ? h.n_uregs : 1, 0);
    v->slot_base = slot_base;
    // same eligibility rule as hnb_program_create
    for (const Ins& in : v->ucode_update) {
        const uint32_t op = in.x & 0xff, a = (in.x >> 16) & 0xff, b = in.x >> 24, c = in.y & 0xff;
        bool ok = vm_op_is_streamable(op) && (a & HNB_OPERAND_U);
        if (ok && (op == HNB_OP_M_RADIAL_ACCEL || op == HNB_OP_M_TANGENT_ACCEL || op == HNB_OP_M_CONFORM_SPHERE || op == HNB_OP_M_KILL_SPHERE || op == HNB_OP_M_KILL_AABB)) ok = (b & HNB_OPERAND_U) != 0;
        if (ok && op == HNB_OP_M_TANGENT_ACCEL) ok = (c & HNB_OPERAND_U) != 0;
        if (!ok) v->streamable = false;
    }
    for (auto& a : v->attrs) if (a.update_flags && a.reg == HNB_REG_NONE) v->streamable = false;
    return v;
}
void cvm_destroy(CpuVm* v) { delete v; }
int cvm_streamable(CpuVm* v) { return v->streamable ? 1 : 0; }

int cvm_set_property(CpuVm* v, const char* name, const uint32_t* words, uint32_t n) {
    for (auto& p : v->props)
        if (strncmp(p.name, name, sizeof p.name) == 0) {
            if (n != p.ncomp) return -1;
            memcpy(&v->prop_words[p.word_offset], words, n * 4);
            return 0;
        }
    return -2;
}

void cvm_step(CpuVm* v, const float* sim, uint32_t spawn_count, uint32_t seed, const float* xf_in, int force_generic) {
    // same file selection as hnb_program_create
    if (v->h.init_regs > HNB_VM_MAX_REGS || v->h.update_regs > HNB_VM_MAX_REGS) step_impl<vreg_file_wide_t>(v, sim, spawn_count, seed, xf_in, force_generic);
    else step_impl<vreg_file_t>(v, sim, spawn_count, seed, xf_in, force_generic);
}

void cvm_counters(CpuVm* v, uint32_t* out8) {
    out8[0] = v->h.capacity; out8[1] = v->alive; out8[2] = v->max_update; out8[3] = v->h.capacity - v->alive;
    out8[4] = v->write_index; out8[5] = v->counter; out8[6] = v->alive; out8[7] = v->dead_count;
}
int cvm_read_attr(CpuVm* v, uint32_t attr, uint32_t* dst) {
    for (size_t a = 0; a < v->attrs.size(); ++a)
        if (v->attrs[a].attr == attr) {
            memcpy(dst, v->slab.data() + v->adesc[a].plane_off, (size_t)v->h.capacity * v->attrs[a].ncomp * 4);
            return (int)v->attrs[a].ncomp;
        }
    return -1;
}
void cvm_read_alive_list(CpuVm* v, uint32_t* dst) { memcpy(dst, v->list[v->write_index].data(), (size_t)v->alive * 4); }
void cvm_read_dead_list(CpuVm* v, uint32_t* dst) { memcpy(dst, v->dead.data() + v->alive, (size_t)(v->h.capacity - v->alive) * 4); }
void cvm_read_ublock(CpuVm* v, uint32_t* dst) { memcpy(dst, v->ublock.data(), (size_t)v->h.n_uregs * 4); }

}  // extern "C"
