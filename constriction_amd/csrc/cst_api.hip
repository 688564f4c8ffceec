// cst_api.hip -- C-ABI entry points (include/constriction_amd.h): argument checks, kernel selection, launches.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "cst_ans_kernels.hpp"

namespace cst {

static thread_local std::string g_last_error;

static thread_local const char* g_last_kernel = "";
cst_status note_kernel(const char* name, cst_status rc) { g_last_kernel = name; return rc; }

void set_hip_error(hipError_t e, const char* what) {
    char buf[512];
    std::snprintf(buf, sizeof buf, "%s: %s (%d)", what ? what : "hip", hipGetErrorString(e), (int)e);
    g_last_error = buf;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// a model's tables live on the device it was built on: coding calls must run there (one process may drive several GPUs)
static inline bool on_model_device(const cst_model* m) {
    int dev = -1;
    return hipGetDevice(&dev) == hipSuccess && dev == m->device;
}

static constexpr size_t kRingBytesPerBlock = (size_t)(kBlock / kWave) * kRingWords * sizeof(uint32_t);
static constexpr size_t kTileBytesPerBlock = (size_t)(kBlock / kWave) * kWave * kTileStride * sizeof(int32_t) + kRingBytesPerBlock;
static constexpr size_t kMaxLds = 160 * 1024;

template <typename K, typename A>
static cst_status launch(K kernel, size_t n_streams, size_t lds_bytes, hipStream_t hs, const A& args) {
    const size_t blocks = (n_streams + kBlock - 1) / kBlock;
    if (blocks == 0) return CST_OK;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    if (lds_bytes > 64 * 1024) {
        CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds_bytes));
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kBlock), lds_bytes, hs, args);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

// ---- encode dispatch ----
template <int W, int S, int G, bool FAST>
static cst_status encode_dispatch_g(const AnsEncodeArgs& a, cst_layout layout, hipStream_t hs) {
    size_t table_bytes = (((size_t)a.n_symbols * sizeof(EncEntry)) + 15) & ~(size_t)15;
    const bool vec0 = (a.n_per_stream % 4 == 0) && aligned16(a.symbols);
    if (table_bytes + kTileBytesPerBlock > kMaxLds) {
        // alphabet too large for an LDS-resident table: entries are read from HBM / L2 (generic steps)
        if (layout == CST_LAYOUT_SYMBOL_MAJOR)
            return launch(ans_encode_kernel<W, S, CST_LAYOUT_SYMBOL_MAJOR, false, G, false, true>, a.n_streams, kTileBytesPerBlock, hs, a);
        if (vec0) return launch(ans_encode_kernel<W, S, CST_LAYOUT_STREAM_MAJOR, true, G, false, true>, a.n_streams, kTileBytesPerBlock, hs, a);
        return launch(ans_encode_kernel<W, S, CST_LAYOUT_STREAM_MAJOR, false, G, false, true>, a.n_streams, kTileBytesPerBlock, hs, a);
    }
    // the main-loop statement of the (32,64), P <= 12 encoder alternates between two tile buffers per wave; without
    // room for the second one the kernel is told so (flag) and stays on the per-tile path
    constexpr size_t kSecondTiles = (size_t)(kBlock / kWave) * kWave * kTileStride * sizeof(int32_t);
    const bool second = FAST && W == 32 && S == 64 && G == 8 && table_bytes + kTileBytesPerBlock + kSecondTiles <= kMaxLds;
    AnsEncodeArgs args = a;
    args.flags = (a.flags & ~CST_KFLAG_TWO_TILES) | (second ? CST_KFLAG_TWO_TILES : 0u);
    if (second) table_bytes += kSecondTiles;
    if (layout == CST_LAYOUT_SYMBOL_MAJOR)
        return launch(ans_encode_kernel<W, S, CST_LAYOUT_SYMBOL_MAJOR, false, G, FAST>, a.n_streams, table_bytes + kTileBytesPerBlock, hs, args);
    const bool vec = (a.n_per_stream % 4 == 0) && aligned16(a.symbols);
    if (vec) return launch(ans_encode_kernel<W, S, CST_LAYOUT_STREAM_MAJOR, true, G, FAST>, a.n_streams, table_bytes + kTileBytesPerBlock, hs, args);
    return launch(ans_encode_kernel<W, S, CST_LAYOUT_STREAM_MAJOR, false, G, FAST>, a.n_streams, table_bytes + kTileBytesPerBlock, hs, args);
}

template <int W, int S>
static cst_status encode_dispatch(const AnsEncodeArgs& a, cst_layout layout, hipStream_t hs) {
    // FAST = the 32-bit-halves step of the (32,64) preset, valid for P >= 8
    const bool fast = (W == 32) && a.precision >= 8;
    switch (groups_per_point(W, a.precision)) {
        case 8: return fast ? encode_dispatch_g<W, S, 8, W == 32>(a, layout, hs) : encode_dispatch_g<W, S, 8, false>(a, layout, hs);
        case 4: return fast ? encode_dispatch_g<W, S, 4, W == 32>(a, layout, hs) : encode_dispatch_g<W, S, 4, false>(a, layout, hs);
        default: return encode_dispatch_g<W, S, 2, false>(a, layout, hs);
    }
}

// ---- decode dispatch ----
template <int W, int S, int MODE, bool LDS, int G, bool FAST>
static cst_status decode_dispatch3(const AnsDecodeArgs& a, cst_layout layout, size_t table_lds, hipStream_t hs) {
    const size_t lds = decode_uses_tile_asm(W, S, MODE, LDS, G, FAST) ? kDecTileAsmLdsBytes
                                                                      : ((table_lds + 15) & ~(size_t)15) + kTileBytesPerBlock;
    if (layout == CST_LAYOUT_SYMBOL_MAJOR)
        return launch(ans_decode_kernel<W, S, CST_LAYOUT_SYMBOL_MAJOR, false, MODE, LDS, G, FAST>, a.n_streams, lds, hs, a);
    const bool vec = (a.n_per_stream % 4 == 0) && aligned16(a.symbols);
    if (vec) return launch(ans_decode_kernel<W, S, CST_LAYOUT_STREAM_MAJOR, true, MODE, LDS, G, FAST>, a.n_streams, lds, hs, a);
    return launch(ans_decode_kernel<W, S, CST_LAYOUT_STREAM_MAJOR, false, MODE, LDS, G, FAST>, a.n_streams, lds, hs, a);
}

template <int W, int S, int MODE, bool LDS>
static cst_status decode_dispatch2(const AnsDecodeArgs& a, cst_layout layout, size_t table_lds, hipStream_t hs) {
    const bool fast = (W == 32) && a.precision >= 8;
    switch (groups_per_point(W, a.precision)) {
        case 8: return fast ? decode_dispatch3<W, S, MODE, LDS, 8, W == 32>(a, layout, table_lds, hs)
                            : decode_dispatch3<W, S, MODE, LDS, 8, false>(a, layout, table_lds, hs);
        case 4: return fast ? decode_dispatch3<W, S, MODE, LDS, 4, W == 32>(a, layout, table_lds, hs)
                            : decode_dispatch3<W, S, MODE, LDS, 4, false>(a, layout, table_lds, hs);
        default: return decode_dispatch3<W, S, MODE, LDS, 2, false>(a, layout, table_lds, hs);
    }
}

template <int W, int S>
static cst_status decode_dispatch(const AnsDecodeArgs& a, cst_layout layout, hipStream_t hs) {
    const int P = a.precision;
    const size_t lds_budget = kMaxLds - kTileBytesPerBlock - 1024;
    if (a.dec_cp && ((size_t)6 << P) <= lds_budget)
        return decode_dispatch2<W, S, kDecLutCP, true>(a, layout, ((size_t)6 << P), hs);
    const size_t bucket_lds = ((((size_t)a.n_symbols + 1) * 4 + 15) & ~(size_t)15) +
                              (bucket16_usable(a.n_symbols, P) ? ((size_t)16 << a.bucket_bits) : ((((size_t)2 << a.bucket_bits) + 15) & ~(size_t)15));
    if (bucket_lds <= lds_budget) return decode_dispatch2<W, S, kDecBucket, true>(a, layout, bucket_lds, hs);
    if (a.dec_cp) return decode_dispatch2<W, S, kDecLutCP, false>(a, layout, 0, hs);
    return decode_dispatch2<W, S, kDecBucket, false>(a, layout, 0, hs);
}

// ---- compaction: single-pass exclusive scan (decoupled look-back) fused with the gather ----
// One workgroup per kCompactStreams streams.  A workgroup takes a ticket (so that every predecessor it may wait for is
// already running), scans its counts, publishes its aggregate in ONE 8-byte agent-scope store {flag, value} (no ordering
// between flag and payload to get wrong: MI355X_MICROARCH.md, inter-workgroup visibility), looks back over its
// predecessors 64 at a time, publishes its inclusive prefix, writes the offsets and copies its streams' words.
constexpr int kCompactStreams = kBlock;          // streams per workgroup (scanned by its first kBlock threads)
constexpr int kCompactThreads = 1024;            // 16 waves copy: the gather is latency bound, it needs loads in flight
constexpr uint64_t kFlagAggregate = 1ull << 62, kFlagInclusive = 2ull << 62, kFlagMask = 3ull << 62;

typedef uint32_t cv4u __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) cv4u_unaligned { cv4u v; };

__global__ __launch_bounds__(kCompactThreads) void compact_kernel(const uint32_t* __restrict__ words, size_t stride, const uint32_t* __restrict__ n_words,
                                                                  size_t n_streams, uint64_t* __restrict__ offsets, uint32_t* __restrict__ packed,
                                                                  size_t capacity, uint32_t* __restrict__ ticket, uint64_t* __restrict__ status) {
    __shared__ uint64_t wave_sums[kBlock / kWave];
    __shared__ uint64_t s_off[kCompactStreams];
    __shared__ uint32_t s_len[kCompactStreams];
    __shared__ uint64_t s_prefix;
    __shared__ uint32_t s_bid;
    if (threadIdx.x == 0) s_bid = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t bid = s_bid;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool scanner = threadIdx.x < kCompactStreams;
    const size_t s = (size_t)bid * kCompactStreams + threadIdx.x;
    const uint32_t len = scanner && s < n_streams ? n_words[s] : 0u;
    uint64_t incl = len;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (scanner && lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    uint64_t wave_off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) { if (w < wave) wave_off += wave_sums[w]; total += wave_sums[w]; }
    if (wave == 0) {
        uint64_t prefix = 0;
        if (bid > 0) {
            if (lane == 0) __hip_atomic_store(&status[bid], kFlagAggregate | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int64_t base = (int64_t)bid - 1;; base -= 64) {
                const int64_t i = base - lane;
                uint64_t st = kFlagInclusive;                       // lanes before block 0: an empty inclusive prefix
                if (i >= 0) {
                    do { st = __hip_atomic_load(&status[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((st & kFlagMask) == 0);
                }
                // nearest predecessor (lowest lane) that already knows its inclusive prefix: sum up to and including it;
                // without one in this window all 64 aggregates count and the look-back goes on
                const uint64_t incl_mask = __ballot((st & kFlagMask) == kFlagInclusive);
                const int stop = incl_mask ? __ffsll((long long)incl_mask) - 1 : 63;
                uint64_t v = lane <= stop ? (st & ~kFlagMask) : 0;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
                prefix += v;
                if (incl_mask != 0) break;
            }
        }
        if (lane == 0) {
            __hip_atomic_store(&status[bid], kFlagInclusive | (prefix + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_prefix = prefix;
        }
    }
    __syncthreads();
    if (scanner) {
        const uint64_t off = s_prefix + wave_off + incl - len;
        if (s < n_streams) offsets[s] = off;
        if (s + 1 == n_streams) offsets[n_streams] = off + len;
        s_off[threadIdx.x] = off; s_len[threadIdx.x] = len;
    }
    __syncthreads();
    if (!packed) return;
    // Gather: one wave per stream at a time.  The head of a stream goes word by word up to the first 16-byte boundary
    // of the packed buffer, the body in aligned 16-byte stores fed by (in general misaligned) 16-byte loads, four
    // independent ones per lane before the first store, the tail word by word.
    for (int k = wave; k < kCompactStreams; k += kCompactThreads / kWave) {
        const size_t sk = (size_t)bid * kCompactStreams + k;
        if (sk >= n_streams) break;
        const uint32_t n = s_len[k];
        const uint64_t o = s_off[k];
        if (o + n > capacity) continue;
        const uint32_t* src = words + sk * stride;
        uint32_t* dst = packed + o;
        const uint32_t head = min(n, (uint32_t)((4u - (uint32_t)((reinterpret_cast<uintptr_t>(dst) >> 2) & 3u)) & 3u));
        if ((uint32_t)lane < head) dst[lane] = src[lane];
        const uint32_t n4 = (n - head) >> 2;                       // whole 16-byte pieces
        const cv4u_unaligned* s4 = reinterpret_cast<const cv4u_unaligned*>(src + head);
        cv4u* d4 = reinterpret_cast<cv4u*>(dst + head);
        for (uint32_t i = lane; i < n4; i += 4 * 64) {
            cv4u v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) if (i + 64 * u < n4) v[u] = s4[i + 64 * u].v;
#pragma unroll
            for (int u = 0; u < 4; ++u) if (i + 64 * u < n4) __builtin_nontemporal_store(v[u], d4 + i + 64 * u);
        }
        const uint32_t done = head + 4 * n4;
        if (done + (uint32_t)lane < n) dst[done + lane] = src[done + lane];
    }
}

} // namespace cst

using namespace cst;

// a wave per stream swaps word i with word n - 1 - i (so the same buffer may be both sides)
__global__ __launch_bounds__(kBlock) void words_reverse_kernel(const uint32_t* in, const uint64_t* __restrict__ off_in, size_t stride_in,
                                                               const uint32_t* __restrict__ n_words, size_t n_streams, uint32_t* out,
                                                               const uint64_t* __restrict__ off_out, size_t stride_out) {
    const size_t s = (size_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (s >= n_streams) return;
    const int lane = threadIdx.x & (kWave - 1);
    const uint32_t n = n_words[s];
    // counts and offsets are caller data, as for the decoders (word_slice): a stream whose count leaves its slab, or its slice of a
    // packed buffer (offsets[s + 1] - offsets[s]), is left untouched instead of read / written out of bounds
    if ((!off_in && stride_in != 0 && n > stride_in) || (!off_out && stride_out != 0 && n > stride_out)) return;
    if ((off_in && (off_in[s + 1] < off_in[s] || n > off_in[s + 1] - off_in[s])) ||
        (off_out && (off_out[s + 1] < off_out[s] || n > off_out[s + 1] - off_out[s]))) return;
    const uint32_t* src = in + (off_in ? (size_t)off_in[s] : s * stride_in);
    uint32_t* dst = out + (off_out ? (size_t)off_out[s] : s * stride_out);
    for (uint32_t i = (uint32_t)lane; i < (n + 1) / 2; i += kWave) {
        const uint32_t a = src[i], b = src[n - 1 - i];
        dst[i] = b;
        dst[n - 1 - i] = a;
    }
}

extern "C" {

int32_t cst_abi_version(void) { return CST_ABI_VERSION; }

const char* cst_last_kernel_name(void) { return g_last_kernel; }

int32_t cst_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { set_hip_error(e, "hipGetDeviceCount"); return CST_ERR_NO_DEVICE; }
    return n;
}

const char* cst_last_hip_error(void) { return g_last_error.c_str(); }

size_t cst_ans_max_words(size_t n, cst_coder_config c) {
    if (c.word_bits <= 0) return 0;
    const size_t by_bits = (n * (size_t)c.precision + (size_t)c.word_bits - 1) / (size_t)c.word_bits;
    const size_t bound = (n < by_bits ? n : by_bits) + (size_t)(c.state_bits / c.word_bits);
    const size_t unit = (size_t)(512 / c.word_bits) > 0 ? (size_t)(512 / c.word_bits) : 1;   // words per 64 bytes
    return (bound + unit - 1) / unit * unit;
}

size_t cst_range_max_words(size_t n, cst_coder_config c) {
    if (c.word_bits <= 0) return 0;
    const size_t by_bits = (n * (size_t)c.precision + (size_t)c.word_bits - 1) / (size_t)c.word_bits;
    const size_t bound = (n < by_bits ? n : by_bits) + 2;
    const size_t unit = (size_t)(512 / c.word_bits) > 0 ? (size_t)(512 / c.word_bits) : 1;   // words per 64 bytes
    return (bound + unit - 1) / unit * unit;
}

cst_status cst_ans_encode_batch(const cst_model* model, cst_coder_config cfg, const int32_t* d_symbols, size_t n_streams,
                                size_t n_per_stream, cst_layout layout, uint32_t* d_words, size_t stride_words,
                                uint32_t* d_n_words, uint64_t* d_state, int32_t* d_status, uint32_t flags, void* stream) {
    if (!model || !d_words || !d_n_words || !d_status) return CST_ERR_INVALID_ARGUMENT;
    if (n_per_stream > 0 && !d_symbols) return CST_ERR_INVALID_ARGUMENT;
    if ((!config_supported(cfg) && !generic_config(cfg)) || cfg.precision != model->precision) return CST_ERR_INVALID_ARGUMENT;
    if (layout != CST_LAYOUT_STREAM_MAJOR && layout != CST_LAYOUT_SYMBOL_MAJOR) return CST_ERR_INVALID_ARGUMENT;
    if ((flags & CST_FLAG_RAW_STATE) && !d_state) return CST_ERR_INVALID_ARGUMENT;
    if (n_streams == 0) return CST_OK;
    if (!on_model_device(model)) return CST_ERR_INVALID_ARGUMENT;
    if ((flags & CST_FLAG_PACKED_W16) && (generic_config(cfg) || model->per_stream)) return CST_ERR_INVALID_ARGUMENT;
    if (generic_config(cfg))       // the rest of the reference's type grid (stack.rs:1293-1356): one compiler-scheduled kernel
        return note_kernel("ans_encode_generic_kernel", ans_encode_generic(model, cfg, d_symbols, n_streams, n_per_stream, layout, d_words, stride_words,
                                                                            d_n_words, d_state, d_status, flags, (hipStream_t)stream));
    if (model->per_stream && pt_usable(model, cfg, layout, n_per_stream))   // one table per stream (config C3), compact rows
        return note_kernel("ans_encode_pt_kernel", ans_encode_pt(model, cfg, d_symbols, n_streams, n_per_stream, d_words, stride_words, d_n_words, d_state,
                                                                  d_status, flags, (hipStream_t)stream));
    if (model->per_stream)   // ... full rows (any supported shape)
        return note_kernel("ans_encode_per_stream_kernel", ans_encode_per_stream(model, cfg, d_symbols, n_streams, n_per_stream, layout, d_words, stride_words,
                                                                                  d_n_words, d_state, d_status, flags, (hipStream_t)stream));
    AnsEncodeArgs a{};
    a.symbols = d_symbols; a.n_streams = n_streams; a.n_per_stream = n_per_stream; a.enc = model->d_enc;
    a.n_symbols = model->n_symbols; a.min_symbol = model->min_symbol; a.precision = model->precision;
    a.words = d_words; a.stride_words = stride_words; a.n_words = d_n_words; a.state = d_state; a.status = d_status;
    a.flags = flags;
    hipStream_t hs = (hipStream_t)stream;
    if (flags & CST_FLAG_PACKED_W16) {     // the reference's Vec<u16>: two words per slot, every count in 16-bit words
        if (!w16pk_usable(model, cfg, layout) || (flags & CST_FLAG_RAW_STATE)) return CST_ERR_INVALID_ARGUMENT;
        return note_kernel("ans_encode_w16pk_kernel", ans_encode_w16pk(a, hs));
    }
    if (small_encode_usable(a, cfg, layout, model->cu_count)) return note_kernel("ans_encode_small_kernel", ans_encode_small(a, hs));   // more than one wave per SIMD
    if (pc_encode_usable(a, cfg, layout, model->cu_count)) return note_kernel(a.precision > 12 ? "ans_encode_pc_kernel<wide>" : "ans_encode_pc_kernel", ans_encode_pc(a, hs));           // one wave per SIMD: coder + helper waves
    if (w16_encode_usable(a, cfg, layout)) return note_kernel("ans_encode_w16_kernel", ans_encode_w16(a, layout, hs));               // SmallAnsCoder preset
    if (wide_encode_usable(a, cfg, layout)) return note_kernel("ans_encode_wide_kernel", ans_encode_wide(a, layout, hs));            // 12 < P <= 24
    if (cfg.word_bits == 32) return note_kernel("ans_encode_kernel", encode_dispatch<32, 64>(a, layout, hs));
    return note_kernel("ans_encode_kernel", encode_dispatch<16, 32>(a, layout, hs));
}

cst_status cst_ans_decode_batch(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words,
                                const uint64_t* d_offsets, size_t stride_words, size_t words_capacity, const uint32_t* d_n_words,
                                int32_t* d_symbols, size_t n_streams, size_t n_per_stream, cst_layout layout,
                                uint64_t* d_state, uint32_t* d_n_words_out, int32_t* d_status, uint32_t flags, void* stream) {
    if (!model || !d_n_words || !d_status) return CST_ERR_INVALID_ARGUMENT;
    if (n_per_stream > 0 && !d_symbols) return CST_ERR_INVALID_ARGUMENT;
    if ((!config_supported(cfg) && !generic_config(cfg)) || cfg.precision != model->precision) return CST_ERR_INVALID_ARGUMENT;
    if (layout != CST_LAYOUT_STREAM_MAJOR && layout != CST_LAYOUT_SYMBOL_MAJOR) return CST_ERR_INVALID_ARGUMENT;
    if ((flags & CST_FLAG_RAW_STATE) && !d_state) return CST_ERR_INVALID_ARGUMENT;
    if (n_streams == 0) return CST_OK;
    if (!on_model_device(model)) return CST_ERR_INVALID_ARGUMENT;
    if ((flags & CST_FLAG_PACKED_W16) && (generic_config(cfg) || model->per_stream)) return CST_ERR_INVALID_ARGUMENT;
    if (generic_config(cfg))
        return note_kernel("ans_decode_generic_kernel", ans_decode_generic(model, cfg, d_words, d_offsets, stride_words, words_capacity, d_n_words, d_symbols,
                                                                            n_streams, n_per_stream, layout, d_state, d_n_words_out, d_status, flags, (hipStream_t)stream));
    if (model->per_stream && pt_usable(model, cfg, layout, n_per_stream))
        return note_kernel("ans_decode_pt_kernel", ans_decode_pt(model, cfg, d_words, d_offsets, stride_words, words_capacity, d_n_words, d_symbols, n_streams,
                                                                  n_per_stream, d_state, d_n_words_out, d_status, flags, (hipStream_t)stream));
    if (model->per_stream)
        return note_kernel("ans_decode_per_stream_kernel", ans_decode_per_stream(model, cfg, d_words, d_offsets, stride_words, words_capacity, d_n_words, d_symbols,
                                                                                  n_streams, n_per_stream, layout, d_state, d_n_words_out, d_status, flags, (hipStream_t)stream));
    AnsDecodeArgs a{};
    a.words = d_words; a.offsets = d_offsets; a.stride_words = stride_words; a.n_words = d_n_words; a.symbols = d_symbols;
    a.n_streams = n_streams; a.n_per_stream = n_per_stream; a.dec_cp = model->d_dec_cp; a.dec_idx = model->d_dec_idx;
    a.cdf = model->d_cdf; a.bucket = model->d_bucket; a.bucket_bits = model->bucket_bits; a.n_symbols = model->n_symbols;
    a.min_symbol = model->min_symbol; a.precision = model->precision; a.state = d_state; a.n_words_out = d_n_words_out;
    a.status = d_status; a.flags = flags; a.words_capacity = words_capacity;
    hipStream_t hs = (hipStream_t)stream;
    if (flags & CST_FLAG_PACKED_W16) {
        if (!w16pk_usable(model, cfg, layout)) return CST_ERR_INVALID_ARGUMENT;
        return note_kernel("ans_decode_w16pk_kernel", ans_decode_w16pk(a, hs));
    }
    if (small_decode_usable(a, cfg, layout, model->cu_count)) return note_kernel("ans_decode_small_kernel", ans_decode_small(a, hs));   // more than one wave per SIMD
    if (b16_small_decode_usable(a, cfg, layout, 4, model->cu_count)) return note_kernel("ans_decode_b16_small_kernel", ans_decode_b16_small(a, 4, hs));   // 12 < P <= 24, two waves per SIMD
    if (b16_decode_usable(a, cfg, layout)) return note_kernel("ans_decode_b16_kernel", ans_decode_b16(a, layout, hs));               // 12 < P <= 24
    if (w16_decode_usable(a, cfg, layout)) return note_kernel("ans_decode_w16_kernel", ans_decode_w16(a, layout, hs));               // SmallAnsCoder preset
    if (dq_decode_usable(a, cfg, layout)) return note_kernel("ans_decode_dq_kernel", ans_decode_dq(a, hs));                          // P <= 12, whole aligned tiles: lane-quad word loads
    if (cfg.word_bits == 32) return note_kernel("ans_decode_kernel", decode_dispatch<32, 64>(a, layout, hs));
    return note_kernel("ans_decode_kernel", decode_dispatch<16, 32>(a, layout, hs));
}

static bool ragged_order_ok(const uint32_t* d_order, size_t n_streams) { return !d_order || n_streams <= 0xffffffffull; }

cst_status cst_ans_encode_ragged_ordered(const cst_model* model, cst_coder_config cfg, const int32_t* d_symbols, const uint64_t* d_sym_offsets,
                                         size_t n_streams, const uint32_t* d_order, uint32_t* d_words, const uint64_t* d_word_offsets,
                                         size_t stride_words, uint32_t* d_n_words, int32_t* d_status, void* stream) {
    if (!model || !config_supported(cfg) || cfg.precision != model->precision) return CST_ERR_INVALID_ARGUMENT;
    if (model->per_stream) return CST_ERR_INVALID_ARGUMENT;      // one shared table (the pattern of tests/issue52.rs)
    if (n_streams == 0) return CST_OK;
    if (!d_sym_offsets || !d_words || !d_n_words || !d_status || !ragged_order_ok(d_order, n_streams)) return CST_ERR_INVALID_ARGUMENT;
    if (!on_model_device(model)) return CST_ERR_INVALID_ARGUMENT;
    return note_kernel("ans_encode_ragged_kernel", ans_encode_ragged(model, cfg, d_symbols, d_sym_offsets, n_streams, d_words, d_word_offsets, stride_words,
                                                                     d_n_words, d_status, d_order, (hipStream_t)stream));
}

cst_status cst_ans_encode_ragged(const cst_model* model, cst_coder_config cfg, const int32_t* d_symbols, const uint64_t* d_sym_offsets,
                                 size_t n_streams, uint32_t* d_words, const uint64_t* d_word_offsets, size_t stride_words,
                                 uint32_t* d_n_words, int32_t* d_status, void* stream) {
    return cst_ans_encode_ragged_ordered(model, cfg, d_symbols, d_sym_offsets, n_streams, nullptr, d_words, d_word_offsets, stride_words,
                                         d_n_words, d_status, stream);
}

cst_status cst_ans_decode_ragged_ordered(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_word_offsets,
                                         size_t stride_words, size_t words_capacity, const uint32_t* d_n_words, int32_t* d_symbols,
                                         const uint64_t* d_sym_offsets, size_t n_streams, const uint32_t* d_order, int32_t* d_status,
                                         void* stream) {
    if (!model || !config_supported(cfg) || cfg.precision != model->precision) return CST_ERR_INVALID_ARGUMENT;
    if (model->per_stream) return CST_ERR_INVALID_ARGUMENT;
    if (n_streams == 0) return CST_OK;
    if (!d_sym_offsets || !d_n_words || !d_status || !ragged_order_ok(d_order, n_streams)) return CST_ERR_INVALID_ARGUMENT;
    if (!on_model_device(model)) return CST_ERR_INVALID_ARGUMENT;
    return note_kernel("ans_decode_ragged_kernel", ans_decode_ragged(model, cfg, d_words, d_word_offsets, stride_words, words_capacity, d_n_words, d_symbols,
                                                                     d_sym_offsets, n_streams, d_status, d_order, (hipStream_t)stream));
}

cst_status cst_ans_decode_ragged(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_word_offsets,
                                 size_t stride_words, size_t words_capacity, const uint32_t* d_n_words, int32_t* d_symbols,
                                 const uint64_t* d_sym_offsets, size_t n_streams, int32_t* d_status, void* stream) {
    return cst_ans_decode_ragged_ordered(model, cfg, d_words, d_word_offsets, stride_words, words_capacity, d_n_words, d_symbols,
                                         d_sym_offsets, n_streams, nullptr, d_status, stream);
}

// ---- ABI 5: jump points for ragged batches ----
size_t cst_ragged_jump_scratch_bytes(size_t n_chunks_total) { return 20 * n_chunks_total + 128; }

cst_status cst_ans_encode_ragged_jump(const cst_model* model, cst_coder_config cfg, const int32_t* d_symbols, const uint64_t* d_sym_offsets,
                                      size_t n_streams, const uint32_t* d_order, uint32_t* d_words, const uint64_t* d_word_offsets,
                                      size_t stride_words, uint32_t* d_n_words, size_t jump_interval, const uint64_t* d_chunk_offsets,
                                      uint32_t* d_jump_pos, uint64_t* d_jump_state, int32_t* d_status, void* stream) {
    if (!model || !config_supported(cfg) || cfg.precision != model->precision || model->per_stream) return CST_ERR_INVALID_ARGUMENT;
    if (jump_interval == 0 || jump_interval % 8 != 0 || jump_interval > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    if (n_streams == 0) return CST_OK;
    if (!d_sym_offsets || !d_words || !d_n_words || !d_status || !d_chunk_offsets || !d_jump_pos || !d_jump_state || !ragged_order_ok(d_order, n_streams))
        return CST_ERR_INVALID_ARGUMENT;
    if (!on_model_device(model)) return CST_ERR_INVALID_ARGUMENT;
    return note_kernel("ans_encode_ragged_kernel<jump>", ans_encode_ragged_jump(model, cfg, d_symbols, d_sym_offsets, n_streams, d_words, d_word_offsets,
                                                                                 stride_words, d_n_words, d_status, d_order, (uint32_t)jump_interval,
                                                                                 d_chunk_offsets, d_jump_pos, d_jump_state, (hipStream_t)stream));
}

cst_status cst_ans_decode_ragged_jump(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_word_offsets,
                                      size_t stride_words, size_t words_capacity, const uint32_t* d_n_words, int32_t* d_symbols,
                                      const uint64_t* d_sym_offsets, size_t n_streams, size_t jump_interval, const uint64_t* d_chunk_offsets,
                                      size_t n_chunks_total, const uint32_t* d_jump_pos, const uint64_t* d_jump_state, void* d_scratch,
                                      int32_t* d_status, void* stream) {
    if (!model || !config_supported(cfg) || cfg.precision != model->precision || model->per_stream) return CST_ERR_INVALID_ARGUMENT;
    if (jump_interval == 0 || jump_interval % 8 != 0 || jump_interval > 0x7fffffffull || n_chunks_total > 0xffffffffull) return CST_ERR_INVALID_ARGUMENT;
    if (n_streams == 0) return CST_OK;
    if (!d_sym_offsets || !d_n_words || !d_status || !d_chunk_offsets || !d_jump_pos || !d_jump_state || !d_scratch) return CST_ERR_INVALID_ARGUMENT;
    if (!on_model_device(model)) return CST_ERR_INVALID_ARGUMENT;
    return note_kernel("ans_decode_ragged_kernel<jump>", ans_decode_ragged_jump(model, cfg, d_words, d_word_offsets, stride_words, words_capacity, d_n_words,
                                                                                 d_symbols, d_sym_offsets, n_streams, (uint32_t)jump_interval, d_chunk_offsets,
                                                                                 n_chunks_total, d_jump_pos, d_jump_state, d_scratch, d_status,
                                                                                 (hipStream_t)stream));
}

cst_status cst_ans_count_until_ordered(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_word_offsets,
                                       size_t stride_words, size_t words_capacity, const uint32_t* d_n_words, size_t n_streams,
                                       const uint32_t* d_order, int32_t eof_symbol, size_t max_symbols, uint64_t* d_lengths,
                                       int32_t* d_status, void* stream) {
    if (!model || !config_supported(cfg) || cfg.precision != model->precision) return CST_ERR_INVALID_ARGUMENT;
    if (model->per_stream) return CST_ERR_INVALID_ARGUMENT;
    if (n_streams == 0) return CST_OK;
    if (!d_n_words || !d_lengths || !d_status || !ragged_order_ok(d_order, n_streams)) return CST_ERR_INVALID_ARGUMENT;
    if (!on_model_device(model)) return CST_ERR_INVALID_ARGUMENT;
    return ans_count_until(model, cfg, d_words, d_word_offsets, stride_words, words_capacity, d_n_words, n_streams, eof_symbol, max_symbols,
                           d_lengths, d_status, d_order, (hipStream_t)stream);
}

cst_status cst_ans_count_until(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_word_offsets,
                               size_t stride_words, size_t words_capacity, const uint32_t* d_n_words, size_t n_streams,
                               int32_t eof_symbol, size_t max_symbols, uint64_t* d_lengths, int32_t* d_status, void* stream) {
    return cst_ans_count_until_ordered(model, cfg, d_words, d_word_offsets, stride_words, words_capacity, d_n_words, n_streams, nullptr,
                                       eof_symbol, max_symbols, d_lengths, d_status, stream);
}

cst_status cst_words_reverse(const uint32_t* d_words_in, const uint64_t* d_offsets_in, size_t stride_in, const uint32_t* d_n_words,
                             size_t n_streams, uint32_t* d_words_out, const uint64_t* d_offsets_out, size_t stride_out, void* stream) {
    if (n_streams == 0) return CST_OK;
    if (!d_words_in || !d_words_out || !d_n_words) return CST_ERR_INVALID_ARGUMENT;
    const size_t n_blocks = (n_streams + kBlock / kWave - 1) / (kBlock / kWave);
    if (n_blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(words_reverse_kernel, dim3((unsigned)n_blocks), dim3(kBlock), 0, (hipStream_t)stream, d_words_in, d_offsets_in, stride_in,
                       d_n_words, n_streams, d_words_out, d_offsets_out, stride_out);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

size_t cst_compact_scratch_bytes(size_t n_streams) {
    const size_t n_blocks = (n_streams + kCompactStreams - 1) / kCompactStreams;
    return 16 + 8 * (n_blocks > 0 ? n_blocks : 1);
}

cst_status cst_compact_words(const uint32_t* d_words, size_t stride_words, const uint32_t* d_n_words, size_t n_streams,
                             uint64_t* d_offsets, uint32_t* d_packed, size_t packed_capacity, void* d_scratch, void* stream) {
    if (!d_offsets) return CST_ERR_INVALID_ARGUMENT;
    hipStream_t hs = (hipStream_t)stream;
    if (n_streams == 0) {
        CST_HIP_TRY(hipMemsetAsync(d_offsets, 0, 8, hs));
        return CST_OK;
    }
    if (!d_n_words || !d_scratch || (d_packed && !d_words)) return CST_ERR_INVALID_ARGUMENT;
    const size_t n_blocks = (n_streams + kCompactStreams - 1) / kCompactStreams;
    if (n_blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    CST_HIP_TRY(hipMemsetAsync(d_scratch, 0, cst_compact_scratch_bytes(n_streams), hs));
    uint32_t* ticket = reinterpret_cast<uint32_t*>(d_scratch);
    uint64_t* status = reinterpret_cast<uint64_t*>(reinterpret_cast<unsigned char*>(d_scratch) + 16);
    hipLaunchKernelGGL(compact_kernel, dim3((unsigned)n_blocks), dim3(kCompactThreads), 0, hs, d_words, stride_words, d_n_words, n_streams,
                       d_offsets, d_packed, packed_capacity, ticket, status);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

} // extern "C"

namespace cst {

// int8 symbol matrices inside the hand-scheduled loops (cst_ans_n8.hip).  Returns false if the shape is not theirs (the caller
// then converts next to the int32 kernels: cst_symbols.hip); otherwise *rc is the call's status.
bool ans_decode_n8_try(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_offsets, size_t stride_words,
                           size_t words_capacity, const uint32_t* d_n_words, void* d_symbols8, int32_t symbol_bytes, size_t n_streams,
                           size_t n_per_stream, cst_layout layout, uint64_t* d_state, uint32_t* d_n_words_out, int32_t* d_status, uint32_t flags,
                           void* stream, cst_status* rc) {
    if (!model || !d_n_words || !d_status || !d_symbols8 || n_streams == 0 || model->per_stream || model->d_symbol_of_index) return false;
    if (!config_supported(cfg) || cfg.precision != model->precision || !on_model_device(model)) return false;
    if ((flags & ~(uint32_t)(CST_FLAG_RAW_STATE | CST_FLAG_COLD_WORDS)) != 0 || ((flags & CST_FLAG_RAW_STATE) && !d_state)) return false;
    AnsDecodeArgs a{};
    a.words = d_words; a.offsets = d_offsets; a.stride_words = stride_words; a.n_words = d_n_words;
    a.symbols = reinterpret_cast<int32_t*>(d_symbols8);
    a.n_streams = n_streams; a.n_per_stream = n_per_stream; a.dec_cp = model->d_dec_cp; a.dec_idx = model->d_dec_idx;
    a.cdf = model->d_cdf; a.bucket = model->d_bucket; a.bucket_bits = model->bucket_bits; a.n_symbols = model->n_symbols;
    a.min_symbol = model->min_symbol; a.precision = model->precision; a.state = d_state; a.n_words_out = d_n_words_out;
    a.status = d_status; a.flags = flags; a.words_capacity = words_capacity;
    if (b16_small_decode_usable(a, cfg, layout, symbol_bytes, model->cu_count)) {      // 12 < P <= 24, more than one wave per SIMD
        *rc = note_kernel(symbol_bytes == 1 ? "ans_decode_b16_small_n8_kernel" : "ans_decode_b16_small_n16_kernel", ans_decode_b16_small(a, symbol_bytes, (hipStream_t)stream));
        return true;
    }
    if (b16_narrow_decode_usable(a, cfg, layout, symbol_bytes)) {      // 12 < P <= 24
        *rc = note_kernel(symbol_bytes == 1 ? "ans_decode_b16_n8_kernel" : "ans_decode_b16_n16_kernel", ans_decode_b16_narrow(a, symbol_bytes, (hipStream_t)stream));
        return true;
    }
    if (!n8_decode_usable(a, cfg, layout, symbol_bytes)) return false;
    if (n8_decode_small(a, model->cu_count))
        *rc = note_kernel(symbol_bytes == 1 ? "ans_decode_small_n8_kernel" : "ans_decode_small_n16_kernel", ans_decode_small_n8(a, symbol_bytes, (hipStream_t)stream));
    else *rc = note_kernel(symbol_bytes == 1 ? "ans_decode_n8_kernel" : "ans_decode_n16_kernel", ans_decode_n8(a, symbol_bytes, (hipStream_t)stream));
    return true;
}

bool ans_encode_n8_try(const cst_model* model, cst_coder_config cfg, const void* d_symbols8, int32_t symbol_bytes, size_t n_streams, size_t n_per_stream,
                       cst_layout layout, uint32_t* d_words, size_t stride_words, uint32_t* d_n_words, uint64_t* d_state, int32_t* d_status,
                       uint32_t flags, void* stream, cst_status* rc) {
    if (!model || !d_words || !d_n_words || !d_status || !d_symbols8 || n_streams == 0 || model->per_stream || model->d_symbol_of_index) return false;
    if (!config_supported(cfg) || cfg.precision != model->precision || !on_model_device(model)) return false;
    if ((flags & ~(uint32_t)CST_FLAG_RAW_STATE) != 0 || ((flags & CST_FLAG_RAW_STATE) && !d_state)) return false;
    AnsEncodeArgs a{};
    a.symbols = reinterpret_cast<const int32_t*>(d_symbols8); a.n_streams = n_streams; a.n_per_stream = n_per_stream; a.enc = model->d_enc;
    a.n_symbols = model->n_symbols; a.min_symbol = model->min_symbol; a.precision = model->precision;
    a.words = d_words; a.stride_words = stride_words; a.n_words = d_n_words; a.state = d_state; a.status = d_status;
    a.flags = flags;
    if (symbol_bytes == 2) {
        if (!pc_n16_encode_usable(a, cfg, layout)) return false;
        *rc = note_kernel(a.precision > 12 ? "ans_encode_pc_n16_kernel<wide>" : "ans_encode_pc_n16_kernel", ans_encode_pc_n16(a, 0, nullptr, nullptr, (hipStream_t)stream));
        return true;
    }
    if (symbol_bytes != 1 || !pc_n8_encode_usable(a, cfg, layout)) return false;
    *rc = note_kernel(a.precision > 12 ? "ans_encode_pc_n8_kernel<wide>" : "ans_encode_pc_n8_kernel", ans_encode_pc_n8(a, (hipStream_t)stream));
    return true;
}

} // namespace cst
