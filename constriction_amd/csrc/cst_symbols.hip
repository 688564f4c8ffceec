// cst_symbols.hip -- NARROW symbol matrices at the boundary (ABI 4).
//
// The reference's coders are generic over the symbol type (`Symbol: PrimInt + AsPrimitive<Probability> + ...`,
// src/stream/model/quantize.rs:229-255; lookup models decode to indices): an alphabet of 101 symbols travels as i8 there if the
// caller says so.  The batched kernels of this library code int32 matrices; what a narrow matrix saves is the LINK -- a batch
// that comes from and goes back to host memory is bound by PCIe (bench.py `end_to_end`: 51 - 56 GB/s against 4.8 TB/s of kernel
// traffic), and int8 symbols are a quarter of its bytes.  So the narrow types are converted ON THE DEVICE, next to the coder
// call: one streaming kernel (16 symbols per lane and access, non-temporal) widens intN -> int32 into a scratch matrix in front
// of an encode, or narrows the decoder's int32 output behind it.  The coder kernels, their words and their status are untouched;
// the extra HBM traffic is 5 bytes per int8 symbol (0.27 ms for the 268 M symbols of config C2, against the 5 ms their bytes
// spend on the link).  Since the second half of round 5 the (32,64) ANS loops read and write int8 / int16 matrices THEMSELVES where the
// shape allows it (cst_ans_pc.hip, cst_ans_n8.hip, ans_decode_b16_narrow_kernel: DESIGN.md 4.13) -- cst_ans_*_batch_sym try them
// first; this file is the path of every other shape (symbol-major, other presets, rows that are not whole 128-byte lines, per-stream
// tables) and, through the two exported conversions, of the other coders.
#include "cst_common.hpp"

namespace cst {

typedef int32_t v4i32 __attribute__((ext_vector_type(4)));

// n symbols of BYTES bytes each (signed).  Lane i of a step moves symbols 4 i .. 4 i + 3: ONE narrow load (4 or 8 bytes) and ONE
// 16-byte store, so that a wave reads 256 / 512 contiguous bytes and writes a contiguous KiB per instruction (a first version
// gave every lane 16 symbols -- four 16-byte stores 64 bytes apart per lane -- and ran at 0.9 TB/s); four steps in flight per lane.
template <int BYTES>
__global__ __launch_bounds__(256) void widen_kernel(const void* __restrict__ in, int32_t* __restrict__ out, size_t n) {
    using T = typename std::conditional<BYTES == 1, int8_t, int16_t>::type;
    using Pack = typename std::conditional<BYTES == 1, uint32_t, uint64_t>::type;
    const T* src = reinterpret_cast<const T*>(in);
    const size_t n4 = n / 4;
    const bool aligned = (reinterpret_cast<uintptr_t>(in) & (4 * BYTES - 1)) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    auto widen4 = [&](size_t i, Pack pk) {
        T v[4];
        __builtin_memcpy(v, &pk, sizeof(Pack));
        v4i32 w;
        w.x = v[0]; w.y = v[1]; w.z = v[2]; w.w = v[3];
        __builtin_nontemporal_store(w, reinterpret_cast<v4i32*>(out) + i);
    };
    if (aligned) {
        const Pack* sp = reinterpret_cast<const Pack*>(src);
        size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        for (; i + 3 * stride < n4; i += 4 * stride) {
            const Pack a = __builtin_nontemporal_load(sp + i), b = __builtin_nontemporal_load(sp + i + stride),
                       c = __builtin_nontemporal_load(sp + i + 2 * stride), d = __builtin_nontemporal_load(sp + i + 3 * stride);
            widen4(i, a); widen4(i + stride, b); widen4(i + 2 * stride, c); widen4(i + 3 * stride, d);
        }
        for (; i < n4; i += stride) widen4(i, sp[i]);
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
            for (int k = 0; k < 4; ++k) out[4 * i + k] = src[4 * i + k];
    }
    if (blockIdx.x == 0) for (size_t i = 4 * n4 + threadIdx.x; i < n; i += blockDim.x) out[i] = src[i];
}

// a value that does not fit BYTES bytes cannot be stored: it is clamped (the callers make sure the model's support fits, so a
// clamped value means a decoder that reported garbage for an invalid stream: its status says so)
template <int BYTES>
__global__ __launch_bounds__(256) void narrow_kernel(const int32_t* __restrict__ in, void* __restrict__ out, size_t n) {
    using T = typename std::conditional<BYTES == 1, int8_t, int16_t>::type;
    using Pack = typename std::conditional<BYTES == 1, uint32_t, uint64_t>::type;
    constexpr int32_t lo = BYTES == 1 ? -128 : -32768, hi = BYTES == 1 ? 127 : 32767;
    T* dst = reinterpret_cast<T*>(out);
    const size_t n4 = n / 4;
    const bool aligned = (reinterpret_cast<uintptr_t>(out) & (4 * BYTES - 1)) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    auto narrow4 = [&](v4i32 w) {
        T v[4] = {(T)min(max(w.x, lo), hi), (T)min(max(w.y, lo), hi), (T)min(max(w.z, lo), hi), (T)min(max(w.w, lo), hi)};
        Pack pk;
        __builtin_memcpy(&pk, v, sizeof(Pack));
        return pk;
    };
    if (aligned) {
        const v4i32* sp = reinterpret_cast<const v4i32*>(in);
        Pack* dp = reinterpret_cast<Pack*>(dst);
        size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        for (; i + 3 * stride < n4; i += 4 * stride) {
            const v4i32 a = __builtin_nontemporal_load(sp + i), b = __builtin_nontemporal_load(sp + i + stride),
                        c = __builtin_nontemporal_load(sp + i + 2 * stride), d = __builtin_nontemporal_load(sp + i + 3 * stride);
            dp[i] = narrow4(a); dp[i + stride] = narrow4(b); dp[i + 2 * stride] = narrow4(c); dp[i + 3 * stride] = narrow4(d);
        }
        for (; i < n4; i += stride) dp[i] = narrow4(sp[i]);
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
            for (int k = 0; k < 4; ++k) dst[4 * i + k] = (T)min(max(in[4 * i + k], lo), hi);
    }
    if (blockIdx.x == 0) for (size_t i = 4 * n4 + threadIdx.x; i < n; i += blockDim.x) dst[i] = (T)min(max(in[i], lo), hi);
}

static unsigned conv_grid(size_t n) {
    const size_t want = (n / 16 + 255) / 256;       // four steps of four symbols per lane
    return (unsigned)(want < 1 ? 1 : (want > 256 * 16 ? 256 * 16 : want));
}

static bool support_fits(const cst_model* m, int symbol_bytes) {
    if (symbol_bytes == 4) return true;
    const int64_t lo = symbol_bytes == 1 ? -128 : -32768, hi = symbol_bytes == 1 ? 127 : 32767;
    if (m->d_symbol_of_index) return true;          // non-contiguous alphabets are coded as indices elsewhere; the caller maps them
    return (int64_t)m->min_symbol >= lo && (int64_t)m->min_symbol + m->n_symbols - 1 <= hi;
}

} // namespace cst

using namespace cst;

extern "C" {

cst_status cst_symbols_widen(const void* d_in, int32_t symbol_bytes, size_t n, int32_t* d_out, void* stream) {
    if (symbol_bytes != 1 && symbol_bytes != 2) return CST_ERR_INVALID_ARGUMENT;
    if (n == 0) return CST_OK;
    if (!d_in || !d_out) return CST_ERR_INVALID_ARGUMENT;
    hipStream_t hs = (hipStream_t)stream;
    if (symbol_bytes == 1) hipLaunchKernelGGL(widen_kernel<1>, dim3(conv_grid(n)), dim3(256), 0, hs, d_in, d_out, n);
    else hipLaunchKernelGGL(widen_kernel<2>, dim3(conv_grid(n)), dim3(256), 0, hs, d_in, d_out, n);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

cst_status cst_symbols_narrow(const int32_t* d_in, size_t n, void* d_out, int32_t symbol_bytes, void* stream) {
    if (symbol_bytes != 1 && symbol_bytes != 2) return CST_ERR_INVALID_ARGUMENT;
    if (n == 0) return CST_OK;
    if (!d_in || !d_out) return CST_ERR_INVALID_ARGUMENT;
    hipStream_t hs = (hipStream_t)stream;
    if (symbol_bytes == 1) hipLaunchKernelGGL(narrow_kernel<1>, dim3(conv_grid(n)), dim3(256), 0, hs, d_in, d_out, n);
    else hipLaunchKernelGGL(narrow_kernel<2>, dim3(conv_grid(n)), dim3(256), 0, hs, d_in, d_out, n);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

size_t cst_symbols_scratch_bytes(size_t n_streams, size_t n_per_stream, int32_t symbol_bytes) {
    return symbol_bytes == 4 ? 0 : 4 * n_streams * n_per_stream + 16;
}

cst_status cst_ans_encode_batch_sym(const cst_model* model, cst_coder_config cfg, const void* d_symbols, int32_t symbol_bytes, size_t n_streams,
                                    size_t n_per_stream, cst_layout layout, uint32_t* d_words, size_t stride_words, uint32_t* d_n_words,
                                    uint64_t* d_state, int32_t* d_status, uint32_t flags, void* d_scratch, void* stream) {
    if (symbol_bytes == 4)
        return cst_ans_encode_batch(model, cfg, reinterpret_cast<const int32_t*>(d_symbols), n_streams, n_per_stream, layout, d_words, stride_words,
                                    d_n_words, d_state, d_status, flags, stream);
    if (!model || (symbol_bytes != 1 && symbol_bytes != 2)) return CST_ERR_INVALID_ARGUMENT;
    {                                          // int8 / int16 inside the hand-scheduled loops where the shape allows it (cst_ans_pc.hip): no scratch, no second kernel
        cst_status rc8 = CST_OK;
        if (ans_encode_n8_try(model, cfg, d_symbols, symbol_bytes, n_streams, n_per_stream, layout, d_words, stride_words, d_n_words, d_state, d_status, flags,
                              stream, &rc8))
            return rc8;
    }
    if (!d_scratch && n_streams * n_per_stream > 0) return CST_ERR_INVALID_ARGUMENT;
    int32_t* wide = reinterpret_cast<int32_t*>((reinterpret_cast<uintptr_t>(d_scratch) + 15) & ~(uintptr_t)15);
    if (symbol_bytes == 1 && model->per_stream && flags == CST_FLAG_NONE && d_symbols && d_words && d_n_words && d_status && n_streams > 0 &&
        n_per_stream >= 32 && config_supported(cfg) && cfg.precision == model->precision) {
        // one table per stream (config C3), int8 matrix (round 6): the jump-point-noting encoder reads int8 tiles itself -- run it with
        // ONE chunk (its only jump point, the end of the stream, goes to the scratch and is not used)
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev != model->device) return CST_ERR_INVALID_ARGUMENT;
        if (pt_n8_encode_usable(model, cfg, layout, d_symbols, n_streams, n_per_stream, n_per_stream)) {
            uint64_t* jump_state = reinterpret_cast<uint64_t*>(wide);
            uint32_t* jump_pos = reinterpret_cast<uint32_t*>(jump_state + n_streams);
            return note_kernel("ans_encode_pt_n8_kernel", ans_encode_pt_ckpt_n8(model, d_symbols, n_streams, n_per_stream, d_words, stride_words, d_n_words,
                                                                                 n_per_stream, jump_pos, jump_state, d_status, (hipStream_t)stream));
        }
    }
    const cst_status rc = cst_symbols_widen(d_symbols, symbol_bytes, n_streams * n_per_stream, wide, stream);
    if (rc != CST_OK) return rc;
    return cst_ans_encode_batch(model, cfg, wide, n_streams, n_per_stream, layout, d_words, stride_words, d_n_words, d_state, d_status, flags, stream);
}

cst_status cst_ans_decode_batch_sym(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_offsets,
                                    size_t stride_words, size_t words_capacity, const uint32_t* d_n_words, void* d_symbols, int32_t symbol_bytes,
                                    size_t n_streams, size_t n_per_stream, cst_layout layout, uint64_t* d_state, uint32_t* d_n_words_out,
                                    int32_t* d_status, uint32_t flags, void* d_scratch, void* stream) {
    if (symbol_bytes == 4)
        return cst_ans_decode_batch(model, cfg, d_words, d_offsets, stride_words, words_capacity, d_n_words, reinterpret_cast<int32_t*>(d_symbols),
                                    n_streams, n_per_stream, layout, d_state, d_n_words_out, d_status, flags, stream);
    if (!model || (symbol_bytes != 1 && symbol_bytes != 2)) return CST_ERR_INVALID_ARGUMENT;
    if (!support_fits(model, symbol_bytes)) return CST_ERR_INVALID_ARGUMENT;      // a decoded symbol must be storable
    {                                          // int8 / int16 inside the loops where the shape allows it (cst_ans_n8.hip): no scratch, no second kernel
        cst_status rc = CST_OK;
        if (ans_decode_n8_try(model, cfg, d_words, d_offsets, stride_words, words_capacity, d_n_words, d_symbols, symbol_bytes, n_streams, n_per_stream, layout,
                              d_state, d_n_words_out, d_status, flags, stream, &rc))
            return rc;
    }
    if (!d_scratch && n_streams * n_per_stream > 0) return CST_ERR_INVALID_ARGUMENT;
    int32_t* wide = reinterpret_cast<int32_t*>((reinterpret_cast<uintptr_t>(d_scratch) + 15) & ~(uintptr_t)15);
    const cst_status rc = cst_ans_decode_batch(model, cfg, d_words, d_offsets, stride_words, words_capacity, d_n_words, wide, n_streams, n_per_stream,
                                               layout, d_state, d_n_words_out, d_status, flags, stream);
    if (rc != CST_OK) return rc;
    return cst_symbols_narrow(wide, n_streams * n_per_stream, d_symbols, symbol_bytes, stream);
}

} // extern "C"
