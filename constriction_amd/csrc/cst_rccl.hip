// cst_rccl.hip -- the ONE exchange step of the multi-GPU path behind the C ABI: gather of the per-stream compressed words
// (packed buffer + offsets, cst_compact_words) from every rank to one root over RCCL / xGMI (BASELINE config C5,
// SURVEY.md 8b "cst_gather_rccl", 8e).  Streams shard trivially -- one process per GPU, no collective on the coding path;
// RCCL has no allgatherv, so the gather is
//   (1) cst_gather_sizes_rccl : one ncclAllGather of (n_streams, total_words) per rank (16 bytes each),
//   (2) cst_gather_rccl       : grouped ncclSend / ncclRecv of the packed words and of the local offsets straight into
//                               their final positions on the root (every peer uses its own xGMI link to the root), then
//                               one small kernel on the root turns local offsets into global ones;
//   (3) cst_scatter_rccl      : the inverse, for decoding on the ranks what one rank holds.
// librccl is opened at first use (dlopen "librccl.so.1", or the path in CST_RCCL_LIB): the coder library itself has no
// link-time dependency on it.
#include <dlfcn.h>
#include <stdlib.h>
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
        // CST_RCCL_LIB names the library to open instead (a site's own RCCL build; the test double of
        // tests/rccl_double/, which lets 2-3 processes on ONE GPU run the exchange: real RCCL refuses duplicate devices)
        const char* override_path = getenv("CST_RCCL_LIB");
        void* h = nullptr;
        if (override_path && *override_path) {
            h = dlopen(override_path, RTLD_NOW | RTLD_LOCAL);
            if (!h) return;                       // a named library that does not load is an error, not a reason to fall back
        } else {
            h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) return;
        }
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
__global__ void rebase_offsets_kernel(uint64_t* __restrict__ all_offsets, size_t first, size_t n, int64_t delta) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) all_offsets[first + i] += (uint64_t)delta;
}

__global__ void set_u64_kernel(uint64_t* __restrict__ slot, uint64_t value) { *slot = value; }

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

// One group of point-to-point operations: the first failure is REMEMBERED, the group is always closed (a thread that
// returned between ncclGroupStart and ncclGroupEnd would leave the group open: later collectives on the communicator hang,
// and the peers block in their own GroupEnd), and only then is the failure reported.
struct Group {
    cst_status st = CST_OK;
    bool open = false;
    Group() {
        if (rccl().GroupStart() == ncclSuccess) open = true;
        else fail("rccl: ncclGroupStart");
    }
    void fail(const char* what) { if (st == CST_OK) { set_hip_error(hipErrorUnknown, what); st = CST_ERR_HIP; } }
    void nccl(ncclResult_t r, const char* what) { if (r != ncclSuccess) fail(what); }
    void hip(hipError_t e, const char* what) { if (e != hipSuccess && st == CST_OK) { set_hip_error(e, what); st = CST_ERR_HIP; } }
    bool good() const { return open && st == CST_OK; }       // (after a failure nothing more is posted)
    cst_status end() {
        if (open) { open = false; nccl(rccl().GroupEnd(), "rccl: ncclGroupEnd"); }
        return st;
    }
    ~Group() { if (open) (void)rccl().GroupEnd(); }
};

cst_status cst_gather_rccl(void* comm, int32_t n_ranks, int32_t rank, int32_t root, const uint32_t* d_packed, const uint64_t* d_offsets,
                           const uint64_t* h_sizes, uint32_t* d_all_packed, uint64_t* d_all_offsets, void* stream) {
    if (!comm || !h_sizes || n_ranks < 1 || rank < 0 || rank >= n_ranks || root < 0 || root >= n_ranks) return CST_ERR_INVALID_ARGUMENT;
    if (!rccl().ok) return CST_ERR_NO_DEVICE;
    hipStream_t hs = (hipStream_t)stream;
    ncclComm_t c = reinterpret_cast<ncclComm_t>(comm);
    const size_t my_streams = (size_t)h_sizes[2 * rank], my_words = (size_t)h_sizes[2 * rank + 1];
    if ((my_words > 0 && !d_packed) || (my_streams > 0 && !d_offsets)) return CST_ERR_INVALID_ARGUMENT;
    if (rank != root) {
        Group g;
        if (g.good() && my_words) g.nccl(rccl().Send(d_packed, my_words, ncclUint32, root, c, hs), "rccl: ncclSend(words)");
        if (g.good() && my_streams) g.nccl(rccl().Send(d_offsets, my_streams, ncclUint64, root, c, hs), "rccl: ncclSend(offsets)");
        return g.end();
    }
    if (!d_all_packed || !d_all_offsets) return CST_ERR_INVALID_ARGUMENT;
    size_t words_before = 0, streams_before = 0;
    {
        Group g;
        for (int r = 0; r < n_ranks; ++r) {
            const size_t ns = (size_t)h_sizes[2 * r], nw = (size_t)h_sizes[2 * r + 1];
            if (r == rank) {
                if (g.good() && nw) g.hip(hipMemcpyAsync(d_all_packed + words_before, d_packed, 4 * nw, hipMemcpyDeviceToDevice, hs), "gather: own words");
                if (g.good() && ns) g.hip(hipMemcpyAsync(d_all_offsets + streams_before, d_offsets, 8 * ns, hipMemcpyDeviceToDevice, hs), "gather: own offsets");
            } else {
                if (g.good() && nw) g.nccl(rccl().Recv(d_all_packed + words_before, nw, ncclUint32, r, c, hs), "rccl: ncclRecv(words)");
                if (g.good() && ns) g.nccl(rccl().Recv(d_all_offsets + streams_before, ns, ncclUint64, r, c, hs), "rccl: ncclRecv(offsets)");
            }
            words_before += nw; streams_before += ns;
        }
        if (cst_status st = g.end()) return st;
    }
    // local offsets -> global offsets (after the receives, same stream); offsets[n_total] = all words.  No host
    // synchronisation: the call is asynchronous on `stream` like every other one.
    words_before = 0; streams_before = 0;
    for (int r = 0; r < n_ranks; ++r) {
        const size_t ns = (size_t)h_sizes[2 * r], nw = (size_t)h_sizes[2 * r + 1];
        if (ns && words_before)
            hipLaunchKernelGGL(rebase_offsets_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, hs, d_all_offsets, streams_before, ns, (int64_t)words_before);
        words_before += nw; streams_before += ns;
    }
    hipLaunchKernelGGL(set_u64_kernel, dim3(1), dim3(1), 0, hs, d_all_offsets + streams_before, (uint64_t)words_before);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

// The inverse ("decode needs the inverse scatter"): the root holds the packed words of all ranks' streams and
// the global offsets[n_total + 1]; rank r receives its own words and its offsets[n_r + 1] rebased to start at 0 -- the
// form cst_ans_decode_batch / cst_range_decode_batch take.  h_sizes as in cst_gather_rccl (known on every rank).
cst_status cst_scatter_rccl(void* comm, int32_t n_ranks, int32_t rank, int32_t root, const uint32_t* d_all_packed,
                            const uint64_t* d_all_offsets, const uint64_t* h_sizes, uint32_t* d_packed, uint64_t* d_offsets, void* stream) {
    if (!comm || !h_sizes || n_ranks < 1 || rank < 0 || rank >= n_ranks || root < 0 || root >= n_ranks) return CST_ERR_INVALID_ARGUMENT;
    if (!rccl().ok) return CST_ERR_NO_DEVICE;
    hipStream_t hs = (hipStream_t)stream;
    ncclComm_t c = reinterpret_cast<ncclComm_t>(comm);
    const size_t my_streams = (size_t)h_sizes[2 * rank], my_words = (size_t)h_sizes[2 * rank + 1];
    if ((my_words > 0 && !d_packed) || !d_offsets) return CST_ERR_INVALID_ARGUMENT;
    size_t my_words_before = 0;
    for (int r = 0; r < rank; ++r) my_words_before += (size_t)h_sizes[2 * r + 1];
    if (rank != root) {
        Group g;
        if (g.good() && my_words) g.nccl(rccl().Recv(d_packed, my_words, ncclUint32, root, c, hs), "rccl: ncclRecv(words)");
        if (g.good()) g.nccl(rccl().Recv(d_offsets, my_streams + 1, ncclUint64, root, c, hs), "rccl: ncclRecv(offsets)");
        if (cst_status st = g.end()) return st;
    } else {
        if (!d_all_packed || !d_all_offsets) return CST_ERR_INVALID_ARGUMENT;
        size_t words_before = 0, streams_before = 0;
        Group g;
        for (int r = 0; r < n_ranks; ++r) {
            const size_t ns = (size_t)h_sizes[2 * r], nw = (size_t)h_sizes[2 * r + 1];
            if (r == rank) {
                if (g.good() && nw) g.hip(hipMemcpyAsync(d_packed, d_all_packed + words_before, 4 * nw, hipMemcpyDeviceToDevice, hs), "scatter: own words");
                if (g.good()) g.hip(hipMemcpyAsync(d_offsets, d_all_offsets + streams_before, 8 * (ns + 1), hipMemcpyDeviceToDevice, hs), "scatter: own offsets");
            } else {
                if (g.good() && nw) g.nccl(rccl().Send(d_all_packed + words_before, nw, ncclUint32, r, c, hs), "rccl: ncclSend(words)");
                if (g.good()) g.nccl(rccl().Send(d_all_offsets + streams_before, ns + 1, ncclUint64, r, c, hs), "rccl: ncclSend(offsets)");
            }
            words_before += nw; streams_before += ns;
        }
        if (cst_status st = g.end()) return st;
    }
    if (my_words_before)      // global -> local offsets
        hipLaunchKernelGGL(rebase_offsets_kernel, dim3((unsigned)((my_streams + 256) / 256)), dim3(256), 0, hs, d_offsets, (size_t)0, my_streams + 1,
                           -(int64_t)my_words_before);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

} // extern "C"
