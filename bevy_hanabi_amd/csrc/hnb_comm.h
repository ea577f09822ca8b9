// Multi-GPU reporting: the all-reduce of the alive-particle counters over RCCL (the ONLY collective of the
// design - effects shard by instance / capacity slab with no inter-GPU dependency per frame).
//
// The reference's seam is one process (src/plugin.rs:202-256, src/render/mod.rs:126-131); the counterpart here is one process
// with one HnbContext per GPU, each driven by its own submit thread (examples/multi_gpu.c), plus this communicator:
//   hnb_comm_create_local   one process, n contexts           -> ncclCommInitAll over the contexts' devices
//   hnb_comm_create_rank    one rank (context) per process    -> ncclCommInitRank with a shared ncclUniqueId
// librccl is resolved at run time (dlopen): a single-GPU host does not need it, and a process that already loaded another copy
// (PyTorch bundles one) keeps using that copy. RCCL refuses a communicator that names one device twice; contexts that share a
// device (tests on a one-GPU box) are reduced through the host instead - same entry point, same result.
#pragma once
#include <dlfcn.h>

#include <mutex>
#include <string>

#include "hnb_comm_decl.h"

namespace hnb {
namespace comm {

struct Api {
    void* lib = nullptr;
    GetUniqueId_fn GetUniqueId = nullptr;
    CommInitRank_fn CommInitRank = nullptr;
    CommInitAll_fn CommInitAll = nullptr;
    CommDestroy_fn CommDestroy = nullptr;
    AllReduce_fn AllReduce = nullptr;
    GroupStart_fn GroupStart = nullptr;
    GroupEnd_fn GroupEnd = nullptr;
    GetErrorString_fn GetErrorString = nullptr;
    bool ok = false;
    bool allows_duplicate_devices = false;   // hnb_comm_set_library(.., HNB_COMM_LIB_DUPLICATE_DEVICES): a stand-in library that accepts one device twice
    std::string why;
    std::string resolved;                    // the file the symbols came from (dladdr of ncclAllReduce): what hnb_comm_describe reports
};

// hnb_comm_set_library: an explicit library path (and what the library tolerates), honoured by the FIRST use of the API only
struct LibraryChoice { std::mutex mu; std::string path; bool duplicate_devices = false; bool single_rank = false; bool loaded = false; };
inline LibraryChoice& library_choice() { static LibraryChoice c; return c; }

inline Api load_api() {
    Api a;
    LibraryChoice& ch = library_choice();
    std::string path;
    {
        std::lock_guard<std::mutex> g(ch.mu);
        ch.loaded = true;
        path = ch.path;
        a.allows_duplicate_devices = ch.duplicate_devices;
    }
    if (!path.empty()) {
        a.lib = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!a.lib) { a.why = std::string("cannot load ") + path + ": " + dlerror(); return a; }
    } else {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            a.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (a.lib) break;
        }
        if (!a.lib) { a.why = std::string("librccl not found: ") + dlerror(); return a; }
    }
#define HNB_SYM(field, name) *reinterpret_cast<void**>(&a.field) = dlsym(a.lib, name); if (!a.field) { a.why = std::string("librccl lacks ") + name; return a; }
    HNB_SYM(GetUniqueId, "ncclGetUniqueId")
    HNB_SYM(CommInitRank, "ncclCommInitRank")
    HNB_SYM(CommInitAll, "ncclCommInitAll")
    HNB_SYM(CommDestroy, "ncclCommDestroy")
    HNB_SYM(AllReduce, "ncclAllReduce")
    HNB_SYM(GroupStart, "ncclGroupStart")
    HNB_SYM(GroupEnd, "ncclGroupEnd")
    HNB_SYM(GetErrorString, "ncclGetErrorString")
#undef HNB_SYM
    {
        Dl_info info;
        if (dladdr(reinterpret_cast<const void*>(a.AllReduce), &info) && info.dli_fname) a.resolved = info.dli_fname;
    }
    a.ok = true;
    return a;
}
// One load per process, safe against concurrent first calls (a function-local static is initialised exactly once: the submit threads of
// several contexts may reach hnb_comm_* together).
inline Api& api() {
    static Api a = load_api();
    return a;
}

// counts[i] = alive_count of the instance whose DevMeta row lives at rows[i] (0 for a null row)
__global__ void k_gather_alive(const uint64_t* __restrict__ rows, unsigned long long* __restrict__ counts, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DevMeta* m = reinterpret_cast<const DevMeta*>(rows[i]);
    counts[i] = m ? (unsigned long long)m->alive_count : 0ull;
}

}  // namespace comm
}  // namespace hnb
