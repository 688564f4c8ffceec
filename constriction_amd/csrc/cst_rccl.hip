// cst_rccl.hip -- the ONE exchange step of the multi-GPU path behind the C ABI: gather of the per-stream compressed words
// (packed buffer + offsets, cst_compact_words) from every rank to one root over RCCL / xGMI (BASELINE config C5,
// SURVEY.md 8b "cst_gather_rccl", 8e).  Streams shard trivially -- one process per GPU, no collective on the coding path;
// RCCL has no allgatherv, so the gather is
//   (1) cst_gather_sizes_rccl : one ncclAllGather of (n_streams, total_words) per rank (16 bytes each),
//   (2) cst_gather_rccl       : grouped ncclSend / ncclRecv of the packed words and of the local offsets straight into
//                               their final positions on the root (every peer uses its own xGMI link to the root), then
//                               one small kernel on the root turns local offsets into global ones.
// librccl is opened at first use (dlopen "librccl.so.1"): the coder library itself has no link-time dependency on it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "cst_common.hpp"

namespace cst {

struct RcclApi {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    bool ok = false;
};

static const RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
#define CST_SYM(field, name) api.field = reinterpret_cast<decltype(api.field)>(dlsym(h, name))
        CST_SYM(GetUniqueId, "ncclGetUniqueId"); CST_SYM(CommInitRank, "ncclCommInitRank"); CST_SYM(CommDestroy, "ncclCommDestroy");
        CST_SYM(AllGather, "ncclAllGather"); CST_SYM(Send, "ncclSend"); CST_SYM(Recv, "ncclRecv");
        CST_SYM(GroupStart, "ncclGroupStart"); CST_SYM(GroupEnd, "ncclGroupEnd");
#undef CST_SYM
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather && api.Send && api.Recv && api.GroupStart && api.GroupEnd;
    });
    return api;
}

#define CST_NCCL_TRY(expr)                                                          \
    do {                                                                            \
        ncclResult_t _r = (expr);                                                   \
        if (_r != ncclSuccess) { set_hip_error(hipErrorUnknown, "rccl: " #expr); return CST_ERR_HIP; } \
    } while (0)

__global__ void own_sizes_kernel(const uint64_t* __restrict__ offsets, size_t n_streams, uint64_t* __restrict__ slot) {
    slot[0] = n_streams;
    slot[1] = offsets[n_streams];
}

// global offsets of rank r's streams: local offset + words of the ranks before it
__global__ void rebase_offsets_kernel(uint64_t* __restrict__ all_offsets, size_t first, size_t n, uint64_t words_before) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) all_offsets[first + i] += words_before;
}

} // namespace cst

using namespace cst;

extern "C" {

cst_status cst_rccl_get_unique_id(void* h_id) {
    if (!h_id) return CST_ERR_INVALID_ARGUMENT;
    if (!rccl().ok) return CST_ERR_NO_DEVICE;
    CST_NCCL_TRY(rccl().GetUniqueId(reinterpret_cast<ncclUniqueId*>(h_id)));
    return CST_OK;
}

cst_status cst_rccl_comm_init(const void* h_id, int32_t n_ranks, int32_t rank, void** out_comm) {
    if (!h_id || !out_comm || n_ranks < 1 || rank < 0 || rank >= n_ranks) return CST_ERR_INVALID_ARGUMENT;
    *out_comm = nullptr;
    if (!rccl().ok) return CST_ERR_NO_DEVICE;
    ncclUniqueId id;
    __builtin_memcpy(&id, h_id, sizeof id);
    ncclComm_t comm = nullptr;
    CST_NCCL_TRY(rccl().CommInitRank(&comm, n_ranks, id, rank));
    *out_comm = comm;
    return CST_OK;
}

cst_status cst_rccl_comm_destroy(void* comm) {
    if (!comm) return CST_OK;
    if (!rccl().ok) return CST_ERR_NO_DEVICE;
    CST_NCCL_TRY(rccl().CommDestroy(reinterpret_cast<ncclComm_t>(comm)));
    return CST_OK;
}

cst_status cst_gather_sizes_rccl(void* comm, int32_t n_ranks, int32_t rank, const uint64_t* d_offsets, size_t n_streams_local,
                                 uint64_t* d_sizes, void* stream) {
    if (!comm || !d_offsets || !d_sizes || n_ranks < 1 || rank < 0 || rank >= n_ranks) return CST_ERR_INVALID_ARGUMENT;
    if (!rccl().ok) return CST_ERR_NO_DEVICE;
    hipStream_t hs = (hipStream_t)stream;
    hipLaunchKernelGGL(own_sizes_kernel, dim3(1), dim3(1), 0, hs, d_offsets, n_streams_local, d_sizes + 2 * (size_t)rank);
    CST_HIP_TRY(hipGetLastError());
    CST_NCCL_TRY(rccl().AllGather(d_sizes + 2 * (size_t)rank, d_sizes, 2, ncclUint64, reinterpret_cast<ncclComm_t>(comm), hs));   // in place
    return CST_OK;
}

cst_status cst_gather_rccl(void* comm, int32_t n_ranks, int32_t rank, int32_t root, const uint32_t* d_packed, const uint64_t* d_offsets,
                           const uint64_t* h_sizes, uint32_t* d_all_packed, uint64_t* d_all_offsets, void* stream) {
    if (!comm || !h_sizes || n_ranks < 1 || rank < 0 || rank >= n_ranks || root < 0 || root >= n_ranks) return CST_ERR_INVALID_ARGUMENT;
    if (!rccl().ok) return CST_ERR_NO_DEVICE;
    hipStream_t hs = (hipStream_t)stream;
    ncclComm_t c = reinterpret_cast<ncclComm_t>(comm);
    const size_t my_streams = (size_t)h_sizes[2 * rank], my_words = (size_t)h_sizes[2 * rank + 1];
    if ((my_words > 0 && !d_packed) || (my_streams > 0 && !d_offsets)) return CST_ERR_INVALID_ARGUMENT;
    if (rank != root) {
        CST_NCCL_TRY(rccl().GroupStart());
        if (my_words) CST_NCCL_TRY(rccl().Send(d_packed, my_words, ncclUint32, root, c, hs));
        if (my_streams) CST_NCCL_TRY(rccl().Send(d_offsets, my_streams, ncclUint64, root, c, hs));
        CST_NCCL_TRY(rccl().GroupEnd());
        return CST_OK;
    }
    if (!d_all_packed || !d_all_offsets) return CST_ERR_INVALID_ARGUMENT;
    size_t words_before = 0, streams_before = 0;
    CST_NCCL_TRY(rccl().GroupStart());
    for (int r = 0; r < n_ranks; ++r) {
        const size_t ns = (size_t)h_sizes[2 * r], nw = (size_t)h_sizes[2 * r + 1];
        if (r == rank) {
            if (nw) CST_HIP_TRY(hipMemcpyAsync(d_all_packed + words_before, d_packed, 4 * nw, hipMemcpyDeviceToDevice, hs));
            if (ns) CST_HIP_TRY(hipMemcpyAsync(d_all_offsets + streams_before, d_offsets, 8 * ns, hipMemcpyDeviceToDevice, hs));
        } else {
            if (nw) CST_NCCL_TRY(rccl().Recv(d_all_packed + words_before, nw, ncclUint32, r, c, hs));
            if (ns) CST_NCCL_TRY(rccl().Recv(d_all_offsets + streams_before, ns, ncclUint64, r, c, hs));
        }
        words_before += nw; streams_before += ns;
    }
    CST_NCCL_TRY(rccl().GroupEnd());
    // local offsets -> global offsets (after the receives, same stream)
    words_before = 0; streams_before = 0;
    for (int r = 0; r < n_ranks; ++r) {
        const size_t ns = (size_t)h_sizes[2 * r], nw = (size_t)h_sizes[2 * r + 1];
        if (ns && words_before)
            hipLaunchKernelGGL(rebase_offsets_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, hs, d_all_offsets, streams_before, ns, words_before);
        words_before += nw; streams_before += ns;
    }
    CST_HIP_TRY(hipGetLastError());
    const uint64_t total = words_before;
    CST_HIP_TRY(hipMemcpyAsync(d_all_offsets + streams_before, &total, 8, hipMemcpyHostToDevice, hs));   // offsets[n_total] = all words
    CST_HIP_TRY(hipStreamSynchronize(hs));   // (`total` lives on this stack frame)
    return CST_OK;
}

} // extern "C"
