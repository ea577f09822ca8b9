// A stand-in for librccl, for ONE purpose: letting the collective branch of hnb_comm_* (ncclCommInitAll -> grouped ncclAllReduce on the
// contexts' streams -> read-back) execute on a box with a single GPU (tests/test_multi_gpu.py). It implements the eight entry points the
// product binds, with the signatures of /opt/rocm/include/rccl/rccl.h (this file includes the real header, so a wrong signature here
// does not compile), for communicators created by ncclCommInitAll inside one process; the reduction goes through host memory and a device
// may appear twice. Test infrastructure: nothing under bevy_hanabi_amd/ references it; it is selected with hnb_comm_set_library(path, ..).
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <vector>

namespace {
struct Clique { int n; };
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
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { if (!id) return ncclInvalidArgument; memset(id->internal, 0x5a, sizeof id->internal); return ncclSuccess; }
ncclResult_t ncclCommInitRank(ncclComm_t*, int, ncclUniqueId, int) { return ncclInvalidUsage; }   // one process only
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
    if (--cl->n == 0) delete cl;
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
