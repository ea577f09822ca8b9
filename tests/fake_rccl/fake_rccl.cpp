// A stand-in for librccl, for ONE purpose: letting the collective branch of hnb_comm_* (ncclCommInitAll -> grouped ncclAllReduce on the
// contexts' streams -> read-back) execute on a box with a single GPU (tests/test_multi_gpu.py). It implements the eight entry points the
// product binds, with the signatures of /opt/rocm/include/rccl/rccl.h (this file includes the real header, so a wrong signature here
// does not compile), for communicators created by ncclCommInitAll inside one process; the reduction goes through host memory and a device
// may appear twice. Round 6: ncclGetUniqueId + ncclCommInitRank too - one rank per PROCESS, the ranks meet in a POSIX shared-memory segment
// named after the unique id (a generation counter, one slot of up to kMaxWords words per rank; every wait is bounded: a missing rank is an
// error after kWaitSeconds, never a hang) - so that hnb_comm_create_rank / bench.py --gpus N can run as N processes on a one-GPU box.
// Test infrastructure: nothing under bevy_hanabi_amd/ references it; it is selected with hnb_comm_set_library(path, ..).
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

namespace {
constexpr size_t kMaxWords = 8192;        // u64 words per all-reduce in rank mode
constexpr int kMaxRanks = 16;
constexpr double kWaitSeconds = 120.0;
struct Shared {                            // the shared segment of one unique id
    std::atomic<int> joined;               // ranks that have attached
    std::atomic<int> arrived[2];           // per phase of a round: ranks that have written their slot / read everybody's
    std::atomic<unsigned> round;
    unsigned long long slot[kMaxRanks][kMaxWords];
};
struct Clique { int n; Shared* sh = nullptr; int nranks = 1; char shm_name[64] = {0}; };
struct Pending { const void* send; void* recv; size_t count; ncclDataType_t dt; ncclRedOp_t op; ncclComm_t comm; hipStream_t stream; };
std::mutex g_mu;
int g_depth = 0;
std::vector<Pending> g_pending;
int g_calls[4] = {0, 0, 0, 0};   // CommInitAll, AllReduce, completed group reductions, CommDestroy: read by the test through fake_rccl_calls
}  // namespace
struct ncclComm { Clique* clique; int rank; int device; };

static ncclResult_t flush() {
    // group the pending operations by clique; every member must have posted one operation of the same shape
    while (!g_pending.empty()) {
        Clique* cl = g_pending.front().comm->clique;
        std::vector<Pending> ops;
        for (size_t i = 0; i < g_pending.size();) {
            if (g_pending[i].comm->clique == cl) { ops.push_back(g_pending[i]); g_pending.erase(g_pending.begin() + i); } else ++i;
        }
        if ((int)ops.size() != cl->n) return ncclInvalidUsage;
        const size_t count = ops[0].count;
        for (const Pending& o : ops) if (o.count != count || o.dt != ncclUint64 || o.op != ncclSum) return ncclInvalidArgument;
        std::vector<unsigned long long> acc(count, 0ull), tmp(count);
        for (const Pending& o : ops) {   // stream order: everything enqueued before the all-reduce has completed when its input is read
            if (hipSetDevice(o.comm->device) != hipSuccess || hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
            if (hipMemcpy(tmp.data(), o.send, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
            for (size_t i = 0; i < count; ++i) acc[i] += tmp[i];
        }
        if (cl->sh) {   // rank mode: this process's sum meets the other ranks' in the shared segment
            if (count > kMaxWords) return ncclInvalidArgument;
            Shared* sh = cl->sh;
            const int rank = ops[0].comm->rank, n = cl->nranks;
            auto wait_for = [&](std::atomic<int>& a, int want) {
                timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
                while (a.load(std::memory_order_acquire) < want) {
                    usleep(50);
                    timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
                    if ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec) > kWaitSeconds) return false;
                }
                return true;
            };
            const unsigned round = sh->round.load(std::memory_order_acquire);
            memcpy(sh->slot[rank], acc.data(), count * 8);
            const int phase = (int)(round & 1u);
            sh->arrived[phase].fetch_add(1, std::memory_order_acq_rel);
            if (!wait_for(sh->arrived[phase], n)) return ncclSystemError;          // everybody's slot is written
            for (size_t i = 0; i < count; ++i) { unsigned long long v = 0; for (int r = 0; r < n; ++r) v += sh->slot[r][i]; acc[i] = v; }
            // second barrier: nobody overwrites a slot before everybody has read it; the last one through re-arms the phase and opens the next round
            const int done = sh->arrived[phase].fetch_add(1, std::memory_order_acq_rel) + 1;
            if (done == 2 * n) { sh->arrived[phase ^ 1].store(0, std::memory_order_release); sh->round.fetch_add(1, std::memory_order_acq_rel); }
            else {
                timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
                while (sh->round.load(std::memory_order_acquire) == round) {
                    usleep(50);
                    timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
                    if ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec) > kWaitSeconds) return ncclSystemError;
                }
            }
        }
        for (const Pending& o : ops) {
            if (hipSetDevice(o.comm->device) != hipSuccess) return ncclUnhandledCudaError;
            if (hipMemcpyAsync(o.recv, acc.data(), count * 8, hipMemcpyHostToDevice, o.stream) != hipSuccess) return ncclUnhandledCudaError;
            if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;   // (acc dies with this scope)
        }
        g_calls[2] += 1;
    }
    return ncclSuccess;
}

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    memset(id->internal, 0x5a, sizeof id->internal);
    // 16 random bytes name the rendezvous segment (the rest keeps the recognisable fill)
    int fd = open("/dev/urandom", O_RDONLY);
    if (fd < 0 || read(fd, id->internal, 16) != 16) { if (fd >= 0) close(fd); return ncclSystemError; }
    close(fd);
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks <= 0 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Clique* cl = new Clique{1};
    cl->nranks = nranks;
    char hex[33];
    for (int i = 0; i < 16; ++i) snprintf(hex + 2 * i, 3, "%02x", (unsigned char)id.internal[i]);
    snprintf(cl->shm_name, sizeof cl->shm_name, "/hnb_fake_rccl_%s", hex);
    const int fd = shm_open(cl->shm_name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Shared)) != 0) { if (fd >= 0) close(fd); delete cl; return ncclSystemError; }   // (a fresh segment is zero-filled: all counters start at 0)
    void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete cl; return ncclSystemError; }
    cl->sh = static_cast<Shared*>(p);
    cl->sh->joined.fetch_add(1, std::memory_order_acq_rel);
    timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
    while (cl->sh->joined.load(std::memory_order_acquire) < nranks) {   // as ncclCommInitRank: returns when every rank has joined
        usleep(200);
        timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec) > kWaitSeconds) { munmap(p, sizeof(Shared)); shm_unlink(cl->shm_name); delete cl; return ncclSystemError; }
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> g(g_mu);
    *comm = new ncclComm{cl, rank, dev};
    g_calls[0] += 1;
    return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
    if (!comms || ndev <= 0) return ncclInvalidArgument;
    std::lock_guard<std::mutex> g(g_mu);
    Clique* cl = new Clique{ndev};
    for (int i = 0; i < ndev; ++i) comms[i] = new ncclComm{cl, i, devlist ? devlist[i] : i};
    g_calls[0] += 1;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    std::lock_guard<std::mutex> g(g_mu);
    Clique* cl = c->clique;
    delete c;
    if (--cl->n == 0) {
        if (cl->sh) { munmap(cl->sh, sizeof(Shared)); shm_unlink(cl->shm_name); }   // (the name goes with the first rank that leaves; the others' mappings stay valid)
        delete cl;
    }
    g_calls[3] += 1;
    return ncclSuccess;
}
ncclResult_t ncclGroupStart() { std::lock_guard<std::mutex> g(g_mu); g_depth += 1; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    std::lock_guard<std::mutex> g(g_mu);
    if (g_depth <= 0) return ncclInvalidUsage;
    return --g_depth == 0 ? flush() : ncclSuccess;
}
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    if (!send || !recv || !comm) return ncclInvalidArgument;
    std::lock_guard<std::mutex> g(g_mu);
    g_pending.push_back(Pending{send, recv, count, dt, op, comm, stream});
    g_calls[1] += 1;
    return g_depth == 0 ? flush() : ncclSuccess;   // outside a group only a one-member clique can complete
}
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : r == ncclInvalidUsage ? "invalid usage (fake_rccl)" : "error (fake_rccl)"; }
void fake_rccl_calls(int out[4]) { std::lock_guard<std::mutex> g(g_mu); for (int i = 0; i < 4; ++i) out[i] = g_calls[i]; }
}
