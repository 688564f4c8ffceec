// fake_rccl.cpp -- TEST DOUBLE (test infrastructure, never shipped) of the eight nccl* entry points the library's multi-GPU
// exchange uses (constriction_amd/csrc/cst_rccl.hip: ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllGather,
// ncclSend, ncclRecv, ncclGroupStart, ncclGroupEnd).  Real RCCL refuses two ranks on one GPU, and the build and test boxes
// have one GPU at most; with CST_RCCL_LIB pointing here, cst_gather_sizes_rccl / cst_gather_rccl / cst_scatter_rccl run
// with 2 or 3 PROCESSES on that one GPU -- the displacement arithmetic for root != 0, ranks without streams and the
// group-guard paths execute for real, only the wire is different: messages are files in a rendezvous directory whose name is
// the "unique id" (write to .tmp, rename: a message is either absent or complete), payloads are staged through the host with
// hipMemcpy.  Semantics kept from NCCL: point-to-point operations between ncclGroupStart and ncclGroupEnd are posted at
// ncclGroupEnd (sends first: a file write never blocks, so no ordering of the ranks can deadlock), matched per (source,
// destination) pair in posting order, element counts of a matched send / recv must agree, and everything is ordered with the
// HIP stream it was given (the stream is synchronised before a buffer is read, the copy into a receive buffer has completed
// before the call returns).
//
//   FAKE_RCCL_HOST=1            buffers are host memory (plain memcpy, no HIP call at all: the CPU test of the double itself)
//   FAKE_RCCL_FAIL_SEND=k       the k-th ncclSend of this process (1-based) returns ncclInternalError (nothing is sent)
//   FAKE_RCCL_TIMEOUT_MS=t      a receive that finds no message within t ms returns ncclSystemError (default 20000)
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int kMaxRanks = 64;

struct Comm {
    int n = 0, rank = 0;
    std::string dir;
    uint64_t sent[kMaxRanks] = {}, received[kMaxRanks] = {}, collectives = 0;
};

struct Op { bool send; const void* src; void* dst; size_t bytes; int peer; Comm* comm; hipStream_t stream; };

int g_depth = 0, g_sends = 0, g_group_ends = 0;
std::vector<Op> g_ops;

bool host_mode() { const char* e = getenv("FAKE_RCCL_HOST"); return e && *e == '1'; }
long env_long(const char* name, long dflt) { const char* e = getenv(name); return e && *e ? atol(e) : dflt; }

size_t elem_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

bool to_host(void* h, const void* d, size_t bytes, hipStream_t s) {
    if (host_mode()) { memcpy(h, d, bytes); return true; }
    return hipStreamSynchronize(s) == hipSuccess && hipMemcpy(h, d, bytes, hipMemcpyDefault) == hipSuccess;
}
bool from_host(void* d, const void* h, size_t bytes, hipStream_t s) {
    if (host_mode()) { memcpy(d, h, bytes); return true; }
    return hipStreamSynchronize(s) == hipSuccess && hipMemcpy(d, h, bytes, hipMemcpyDefault) == hipSuccess;
}

bool write_message(const std::string& path, const void* data, size_t bytes) {
    const std::string tmp = path + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(data, 1, bytes, f) == bytes;
    return fclose(f) == 0 && ok && rename(tmp.c_str(), path.c_str()) == 0;
}

// waits for the message; false on timeout or on a size that does not match the posted receive
bool read_message(const std::string& path, void* data, size_t bytes, bool consume) {
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(env_long("FAKE_RCCL_TIMEOUT_MS", 20000));
    struct stat st;
    while (stat(path.c_str(), &st) != 0) {
        if (std::chrono::steady_clock::now() > deadline) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    if ((size_t)st.st_size != bytes) return false;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    const bool ok = fread(data, 1, bytes, f) == bytes;
    fclose(f);
    if (consume) unlink(path.c_str());
    return ok;
}

std::string p2p_path(const Comm* c, int src, int dst, uint64_t seq) {
    char name[96];
    snprintf(name, sizeof name, "/p2p_%d_%d_%llu.msg", src, dst, (unsigned long long)seq);
    return c->dir + name;
}

ncclResult_t run(std::vector<Op>& ops) {
    ncclResult_t res = ncclSuccess;
    std::vector<char> buf;
    for (const Op& op : ops) {                       // sends first
        if (!op.send) continue;
        buf.resize(op.bytes ? op.bytes : 1);
        Comm* c = op.comm;
        if (!to_host(buf.data(), op.src, op.bytes, op.stream) || !write_message(p2p_path(c, c->rank, op.peer, c->sent[op.peer]++), buf.data(), op.bytes))
            res = ncclSystemError;
    }
    for (const Op& op : ops) {
        if (op.send) continue;
        buf.resize(op.bytes ? op.bytes : 1);
        Comm* c = op.comm;
        if (!read_message(p2p_path(c, op.peer, c->rank, c->received[op.peer]++), buf.data(), op.bytes, true) ||
            !from_host(op.dst, buf.data(), op.bytes, op.stream))
            res = ncclSystemError;
    }
    ops.clear();
    return res;
}

ncclResult_t post(const Op& op) {
    g_ops.push_back(op);
    return g_depth > 0 ? ncclSuccess : run(g_ops);
}

} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    const char* base = getenv("FAKE_RCCL_DIR");
    std::string templ = std::string(base && *base ? base : "/tmp") + "/fake_rccl_XXXXXX";
    if (templ.size() + 1 > sizeof id->internal) return ncclInvalidArgument;
    std::vector<char> path(templ.begin(), templ.end());
    path.push_back(0);
    if (!mkdtemp(path.data())) return ncclSystemError;
    memcpy(id->internal, path.data(), path.size());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    id.internal[sizeof id.internal - 1] = 0;
    Comm* c = new Comm;
    c->n = nranks; c->rank = rank; c->dir = id.internal;
    struct stat st;
    if (stat(c->dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) { delete c; return ncclInvalidArgument; }
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    delete reinterpret_cast<Comm*>(comm);
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) return ncclInvalidUsage;
    ++g_group_ends;
    return --g_depth == 0 ? run(g_ops) : ncclSuccess;
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c || peer < 0 || peer >= c->n || peer == c->rank || !elem_size(datatype) || (count && !sendbuff)) return ncclInvalidArgument;
    if (++g_sends == env_long("FAKE_RCCL_FAIL_SEND", 0)) return ncclInternalError;
    return post(Op{true, sendbuff, nullptr, count * elem_size(datatype), peer, c, stream});
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c || peer < 0 || peer >= c->n || peer == c->rank || !elem_size(datatype) || (count && !recvbuff)) return ncclInvalidArgument;
    return post(Op{false, nullptr, recvbuff, count * elem_size(datatype), peer, c, stream});
}

// every rank contributes `sendcount` elements; rank r's land at recvbuff + r * sendcount (sendbuff may be that very place)
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    const size_t bytes = sendcount * elem_size(datatype);
    if (!c || !elem_size(datatype) || (bytes && (!sendbuff || !recvbuff))) return ncclInvalidArgument;
    const uint64_t seq = c->collectives++;
    auto path = [&](int r) {
        char name[96];
        snprintf(name, sizeof name, "/allgather_%llu_%d.msg", (unsigned long long)seq, r);
        return c->dir + name;
    };
    std::vector<char> buf(bytes ? bytes : 1);
    if (!to_host(buf.data(), sendbuff, bytes, stream) || !write_message(path(c->rank), buf.data(), bytes)) return ncclSystemError;
    for (int r = 0; r < c->n; ++r) {
        if (!read_message(path(r), buf.data(), bytes, false)) return ncclSystemError;      // (n readers per message: left for the directory's owner to remove)
        if (!from_host(static_cast<char*>(recvbuff) + (size_t)r * bytes, buf.data(), bytes, stream)) return ncclSystemError;
    }
    return ncclSuccess;
}

// what the tests look at
int fake_rccl_group_depth() { return g_depth; }
int fake_rccl_group_ends() { return g_group_ends; }
int fake_rccl_sends() { return g_sends; }

} // extern "C"
