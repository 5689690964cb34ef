// rccl_abi_check.cpp -- build-time check (syntax only, build.sh) that the RCCL prototypes dist.cpp declares by hand and resolves with dlsym still match the
// installed <rccl/rccl.h>.  Compiled only where the header exists; nothing here is linked into libmsstitch.so.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <type_traits>

static_assert(NCCL_MAJOR == 2, "dist.cpp's declarations are for NCCL / RCCL major version 2");
static_assert(sizeof(ncclUniqueId) == 128, "dist.cpp: ncclUniqueId is 128 opaque bytes");
static_assert(ncclSuccess == 0 && ncclInt8 == 0 && ncclUint8 == 1 && ncclInt32 == 2, "dist.cpp: enum values");
#define SAME(f, ...) static_assert(std::is_same<decltype(&f), ncclResult_t (*)(__VA_ARGS__)>::value, #f " has another prototype than dist.cpp declares")
SAME(ncclGetVersion, int *);
SAME(ncclGetUniqueId, ncclUniqueId *);
SAME(ncclCommInitRank, ncclComm_t *, int, ncclUniqueId, int);
SAME(ncclCommDestroy, ncclComm_t);
SAME(ncclCommCount, const ncclComm_t, int *);
SAME(ncclSend, const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
SAME(ncclRecv, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
SAME(ncclBroadcast, const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
SAME(ncclAllGather, const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
SAME(ncclGroupStart);
SAME(ncclGroupEnd);
static_assert(std::is_same<decltype(&ncclGetErrorString), const char *(*)(ncclResult_t)>::value, "ncclGetErrorString");
static_assert(sizeof(ncclDataType_t) == sizeof(int) && sizeof(ncclResult_t) == sizeof(int), "enums are passed as int");
