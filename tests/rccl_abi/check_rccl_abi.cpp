// Compiled (never run) by tests/test_rccl_abi.py: hnb_comm_decl.h - the RCCL declarations the product binds with dlsym, written by
// hand so that libhanabi_amd.so has no build- or load-time dependency on RCCL - against the real header of the installed RCCL.
// Every value and every signature the product relies on is pinned here; a mismatch is a compile error.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <type_traits>

#include "hnb_comm_decl.h"

namespace hc = hnb::comm;

// values
static_assert(hc::kNcclSuccess == (int)ncclSuccess, "ncclSuccess");
static_assert(hc::kNcclUint64 == (int)ncclUint64, "ncclUint64");
static_assert(hc::kNcclSum == (int)ncclSum, "ncclSum");
static_assert(hc::kNcclUniqueIdBytes == NCCL_UNIQUE_ID_BYTES, "NCCL_UNIQUE_ID_BYTES");
// the unique id travels BY VALUE through ncclCommInitRank: size, alignment and layout class must agree
static_assert(sizeof(hc::ncclUniqueId) == sizeof(::ncclUniqueId) && alignof(hc::ncclUniqueId) == alignof(::ncclUniqueId), "ncclUniqueId layout");
static_assert(std::is_standard_layout<::ncclUniqueId>::value && std::is_trivially_copyable<::ncclUniqueId>::value, "ncclUniqueId is a plain struct");
static_assert(std::is_same<decltype(::ncclUniqueId::internal), char[NCCL_UNIQUE_ID_BYTES]>::value, "ncclUniqueId::internal");
// the enums are passed as 4-byte integers
static_assert(sizeof(ncclResult_t) == sizeof(int) && sizeof(ncclDataType_t) == sizeof(int) && sizeof(ncclRedOp_t) == sizeof(int), "enum width");
static_assert(std::is_pointer<::ncclComm_t>::value && sizeof(::ncclComm_t) == sizeof(hc::ncclComm_t), "ncclComm_t is an opaque pointer");

// signatures: the real function's type, with the real enum / handle types replaced by what hnb_comm_decl.h declares in their place
template <class T> struct Map { typedef T type; };
template <> struct Map<ncclResult_t> { typedef int type; };
template <> struct Map<ncclDataType_t> { typedef int type; };
template <> struct Map<ncclRedOp_t> { typedef int type; };
template <> struct Map<::ncclComm_t> { typedef hc::ncclComm_t type; };
template <> struct Map<::ncclComm_t*> { typedef hc::ncclComm_t* type; };
template <> struct Map<::ncclUniqueId> { typedef hc::ncclUniqueId type; };
template <> struct Map<::ncclUniqueId*> { typedef hc::ncclUniqueId* type; };
template <class F> struct MapFn;
template <class R, class... A> struct MapFn<R (*)(A...)> { typedef typename Map<R>::type (*type)(typename Map<A>::type...); };
#define SAME(real, mine) static_assert(std::is_same<MapFn<decltype(&real)>::type, hc::mine>::value, #real " does not have the signature hnb_comm_decl.h declares")
SAME(ncclGetUniqueId, GetUniqueId_fn);
SAME(ncclCommInitRank, CommInitRank_fn);
SAME(ncclCommInitAll, CommInitAll_fn);
SAME(ncclCommDestroy, CommDestroy_fn);
SAME(ncclAllReduce, AllReduce_fn);
SAME(ncclGroupStart, GroupStart_fn);
SAME(ncclGroupEnd, GroupEnd_fn);
SAME(ncclGetErrorString, GetErrorString_fn);

int main() { return 0; }
