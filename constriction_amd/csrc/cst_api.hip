// cst_api.hip -- C-ABI entry points (include/constriction_amd.h): argument checks, kernel selection, launches.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "cst_ans_kernels.hpp"

namespace cst {

static thread_local std::string g_last_error;

void set_hip_error(hipError_t e, const char* what) {
    char buf[512];
    std::snprintf(buf, sizeof buf, "%s: %s (%d)", what ? what : "hip", hipGetErrorString(e), (int)e);
    g_last_error = buf;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

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
    const size_t bucket_lds = ((((size_t)a.n_symbols + 1) * 4 + 15) & ~(size_t)15) + ((((size_t)2 << a.bucket_bits) + 15) & ~(size_t)15);
    if (bucket_lds <= lds_budget) return decode_dispatch2<W, S, kDecBucket, true>(a, layout, bucket_lds, hs);
    if (a.dec_cp) return decode_dispatch2<W, S, kDecLutCP, false>(a, layout, 0, hs);
    return decode_dispatch2<W, S, kDecBucket, false>(a, layout, 0, hs);
}

// ---- compaction ----
constexpr int kScanItems = 4; // per thread

__global__ __launch_bounds__(kBlock) void scan_local_kernel(const uint32_t* __restrict__ n_words, size_t n,
                                                             uint64_t* __restrict__ offsets, uint64_t* __restrict__ block_sums) {
    __shared__ uint64_t wave_sums[kBlock / kWave];
    const size_t base = ((size_t)blockIdx.x * kBlock + threadIdx.x) * kScanItems;
    uint64_t v[kScanItems], sum = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) { v[i] = (base + i < n) ? n_words[base + i] : 0; sum += v[i]; }
    // inclusive scan of `sum` across the wave, then across the 4 waves
    uint64_t incl = sum;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wave_sums[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t wave_off = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) wave_off += wave_sums[w];
    uint64_t excl = wave_off + incl - sum;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) { if (base + i < n) offsets[base + i] = excl; excl += v[i]; }
    if (threadIdx.x == kBlock - 1) block_sums[blockIdx.x] = wave_off + incl;
}

// one workgroup: exclusive scan of the block sums in place; total -> offsets[n]
__global__ __launch_bounds__(kBlock) void scan_blocks_kernel(uint64_t* __restrict__ block_sums, size_t n_blocks,
                                                              uint64_t* __restrict__ total_out) {
    __shared__ uint64_t wave_sums[kBlock / kWave];
    __shared__ uint64_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (size_t start = 0; start < n_blocks; start += kBlock) {
        const size_t i = start + threadIdx.x;
        const uint64_t v = i < n_blocks ? block_sums[i] : 0;
        uint64_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wave_sums[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint64_t off = carry_s;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += wave_sums[w];
        if (i < n_blocks) block_sums[i] = off + incl - v;
        __syncthreads();
        if (threadIdx.x == kBlock - 1) carry_s = off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry_s;
}

__global__ __launch_bounds__(kBlock) void scan_add_kernel(uint64_t* __restrict__ offsets, size_t n,
                                                           const uint64_t* __restrict__ block_sums) {
    const size_t base = ((size_t)blockIdx.x * kBlock + threadIdx.x) * kScanItems;
    const uint64_t add = block_sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < kScanItems; ++i)
        if (base + i < n) offsets[base + i] += add;
}

// one wave per stream: copy its words from the slab to the packed buffer (256-B coalesced chunks)
__global__ __launch_bounds__(kBlock) void gather_kernel(const uint32_t* __restrict__ words, size_t stride,
                                                         const uint32_t* __restrict__ n_words, const uint64_t* __restrict__ offsets,
                                                         size_t n_streams, uint32_t* __restrict__ packed) {
    const size_t s = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    if (s >= n_streams) return;
    const int lane = threadIdx.x & 63;
    const uint32_t n = n_words[s];
    const uint32_t* src = words + s * stride;
    uint32_t* dst = packed + offsets[s];
    for (uint32_t i = lane; i < n; i += 64) dst[i] = src[i];
}

} // namespace cst

using namespace cst;

extern "C" {

int32_t cst_abi_version(void) { return CST_ABI_VERSION; }

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
    return (n < by_bits ? n : by_bits) + 2;
}

cst_status cst_ans_encode_batch(const cst_model* model, cst_coder_config cfg, const int32_t* d_symbols, size_t n_streams,
                                size_t n_per_stream, cst_layout layout, uint32_t* d_words, size_t stride_words,
                                uint32_t* d_n_words, uint64_t* d_state, int32_t* d_status, uint32_t flags, void* stream) {
    if (!model || !d_words || !d_n_words || !d_status) return CST_ERR_INVALID_ARGUMENT;
    if (n_per_stream > 0 && !d_symbols) return CST_ERR_INVALID_ARGUMENT;
    if (!config_supported(cfg) || cfg.precision != model->precision) return CST_ERR_INVALID_ARGUMENT;
    if (layout != CST_LAYOUT_STREAM_MAJOR && layout != CST_LAYOUT_SYMBOL_MAJOR) return CST_ERR_INVALID_ARGUMENT;
    if ((flags & CST_FLAG_RAW_STATE) && !d_state) return CST_ERR_INVALID_ARGUMENT;
    if (n_streams == 0) return CST_OK;
    if (model->per_stream && pt_usable(model, cfg, layout, n_per_stream))   // one table per stream (config C3), compact rows
        return ans_encode_pt(model, cfg, d_symbols, n_streams, n_per_stream, d_words, stride_words, d_n_words, d_state, d_status,
                             flags, (hipStream_t)stream);
    if (model->per_stream)   // ... full rows (any supported shape)
        return ans_encode_per_stream(model, cfg, d_symbols, n_streams, n_per_stream, layout, d_words, stride_words, d_n_words,
                                     d_state, d_status, flags, (hipStream_t)stream);
    AnsEncodeArgs a{};
    a.symbols = d_symbols; a.n_streams = n_streams; a.n_per_stream = n_per_stream; a.enc = model->d_enc;
    a.n_symbols = model->n_symbols; a.min_symbol = model->min_symbol; a.precision = model->precision;
    a.words = d_words; a.stride_words = stride_words; a.n_words = d_n_words; a.state = d_state; a.status = d_status;
    a.flags = flags;
    hipStream_t hs = (hipStream_t)stream;
    if (cfg.word_bits == 32) return encode_dispatch<32, 64>(a, layout, hs);
    return encode_dispatch<16, 32>(a, layout, hs);
}

cst_status cst_ans_decode_batch(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words,
                                const uint64_t* d_offsets, size_t stride_words, const uint32_t* d_n_words,
                                int32_t* d_symbols, size_t n_streams, size_t n_per_stream, cst_layout layout,
                                uint64_t* d_state, uint32_t* d_n_words_out, int32_t* d_status, uint32_t flags, void* stream) {
    if (!model || !d_n_words || !d_status) return CST_ERR_INVALID_ARGUMENT;
    if (n_per_stream > 0 && !d_symbols) return CST_ERR_INVALID_ARGUMENT;
    if (!config_supported(cfg) || cfg.precision != model->precision) return CST_ERR_INVALID_ARGUMENT;
    if (layout != CST_LAYOUT_STREAM_MAJOR && layout != CST_LAYOUT_SYMBOL_MAJOR) return CST_ERR_INVALID_ARGUMENT;
    if ((flags & CST_FLAG_RAW_STATE) && !d_state) return CST_ERR_INVALID_ARGUMENT;
    if (n_streams == 0) return CST_OK;
    if (model->per_stream && pt_usable(model, cfg, layout, n_per_stream))
        return ans_decode_pt(model, cfg, d_words, d_offsets, stride_words, d_n_words, d_symbols, n_streams, n_per_stream, d_state,
                             d_n_words_out, d_status, flags, (hipStream_t)stream);
    if (model->per_stream)
        return ans_decode_per_stream(model, cfg, d_words, d_offsets, stride_words, d_n_words, d_symbols, n_streams, n_per_stream,
                                     layout, d_state, d_n_words_out, d_status, flags, (hipStream_t)stream);
    AnsDecodeArgs a{};
    a.words = d_words; a.offsets = d_offsets; a.stride_words = stride_words; a.n_words = d_n_words; a.symbols = d_symbols;
    a.n_streams = n_streams; a.n_per_stream = n_per_stream; a.dec_cp = model->d_dec_cp; a.dec_idx = model->d_dec_idx;
    a.cdf = model->d_cdf; a.bucket = model->d_bucket; a.bucket_bits = model->bucket_bits; a.n_symbols = model->n_symbols;
    a.min_symbol = model->min_symbol; a.precision = model->precision; a.state = d_state; a.n_words_out = d_n_words_out;
    a.status = d_status; a.flags = flags;
    hipStream_t hs = (hipStream_t)stream;
    if (cfg.word_bits == 32) return decode_dispatch<32, 64>(a, layout, hs);
    return decode_dispatch<16, 32>(a, layout, hs);
}

cst_status cst_compact_words(const uint32_t* d_words, size_t stride_words, const uint32_t* d_n_words, size_t n_streams,
                             uint64_t* d_offsets, uint32_t* d_packed, size_t packed_capacity, uint64_t* h_total_words,
                             void* stream) {
    if (!d_n_words || !d_offsets) return CST_ERR_INVALID_ARGUMENT;
    if (d_packed && !d_words) return CST_ERR_INVALID_ARGUMENT;
    hipStream_t hs = (hipStream_t)stream;
    const size_t per_block = (size_t)kBlock * kScanItems;
    const size_t n_blocks = (n_streams + per_block - 1) / per_block;
    uint64_t total = 0;
    if (n_streams == 0) {
        CST_HIP_TRY(hipMemsetAsync(d_offsets, 0, 8, hs));
    } else {
        uint64_t* block_sums = nullptr;
        CST_HIP_TRY(hipMallocAsync((void**)&block_sums, 8 * n_blocks, hs));
        hipLaunchKernelGGL(scan_local_kernel, dim3((unsigned)n_blocks), dim3(kBlock), 0, hs, d_n_words, n_streams, d_offsets,
                           block_sums);
        hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(kBlock), 0, hs, block_sums, n_blocks, d_offsets + n_streams);
        hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)n_blocks), dim3(kBlock), 0, hs, d_offsets, n_streams,
                           (const uint64_t*)block_sums);
        CST_HIP_TRY(hipGetLastError());
        CST_HIP_TRY(hipFreeAsync(block_sums, hs));
    }
    if (h_total_words || d_packed) {
        CST_HIP_TRY(hipMemcpyAsync(&total, d_offsets + n_streams, 8, hipMemcpyDeviceToHost, hs));
        CST_HIP_TRY(hipStreamSynchronize(hs));
        if (h_total_words) *h_total_words = total;
    }
    if (d_packed && n_streams > 0) {
        if (total > packed_capacity) return CST_ERR_INVALID_ARGUMENT;
        const size_t blocks = (n_streams * kWave + kBlock - 1) / kBlock;
        hipLaunchKernelGGL(gather_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, hs, d_words, stride_words, d_n_words,
                           (const uint64_t*)d_offsets, n_streams, d_packed);
        CST_HIP_TRY(hipGetLastError());
    }
    return CST_OK;
}

} // extern "C"
