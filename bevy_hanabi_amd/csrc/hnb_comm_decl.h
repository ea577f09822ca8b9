// The slice of the RCCL API the product binds at run time (hnb_comm.h), declared by hand so that the product needs neither rccl.h
// at build time nor librccl at load time. tests/rccl_abi/check_rccl_abi.cpp compiles this header TOGETHER with
// /opt/rocm/include/rccl/rccl.h and static_asserts every value and every signature below against the real declarations
// (tests/test_rccl_abi.py, `-m "not gpu"`): a drift between this file and the installed RCCL fails the CPU suite.
// No HIP dependency beyond the stream handle type, which the including file provides (hipStream_t).
#pragma once
#include <cstddef>

namespace hnb {
namespace comm {

typedef struct ncclComm* ncclComm_t;
constexpr int kNcclUniqueIdBytes = 128;            // NCCL_UNIQUE_ID_BYTES (== HNB_COMM_ID_BYTES of the C ABI)
struct ncclUniqueId { char internal[kNcclUniqueIdBytes]; };
enum { kNcclSuccess = 0, kNcclUint64 = 5, kNcclSum = 0 };   // ncclSuccess, ncclUint64 (ncclDataType_t), ncclSum (ncclRedOp_t)

// ncclResult_t, ncclDataType_t and ncclRedOp_t are plain C enums without a fixed underlying type: 4-byte integers in the ABI.
typedef int (*GetUniqueId_fn)(ncclUniqueId*);
typedef int (*CommInitRank_fn)(ncclComm_t*, int, ncclUniqueId, int);
typedef int (*CommInitAll_fn)(ncclComm_t*, int, const int*);
typedef int (*CommDestroy_fn)(ncclComm_t);
typedef int (*AllReduce_fn)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
typedef int (*GroupStart_fn)();
typedef int (*GroupEnd_fn)();
typedef const char* (*GetErrorString_fn)(int);

}  // namespace comm
}  // namespace hnb
