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
// spend on the link).  Native narrow loads / stores inside the hand-scheduled loops are the next step.
#include "cst_common.hpp"

namespace cst {

typedef int32_t v4i32 __attribute__((ext_vector_type(4)));

// n symbols of BYTES bytes each (signed), 16 per lane and step
template <int BYTES>
__global__ __launch_bounds__(256) void widen_kernel(const void* __restrict__ in, int32_t* __restrict__ out, size_t n) {
    using T = typename std::conditional<BYTES == 1, int8_t, int16_t>::type;
    const T* src = reinterpret_cast<const T*>(in);
    const size_t n16 = n / 16;
    const bool aligned = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        T v[16];
        if (aligned) {
#pragma unroll
            for (int k = 0; k < BYTES; ++k) reinterpret_cast<v4i32*>(v)[k] = __builtin_nontemporal_load(reinterpret_cast<const v4i32*>(src + 16 * i) + k);
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = src[16 * i + k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v4i32 w;
            w.x = v[4 * k]; w.y = v[4 * k + 1]; w.z = v[4 * k + 2]; w.w = v[4 * k + 3];
            if (aligned) __builtin_nontemporal_store(w, reinterpret_cast<v4i32*>(out + 16 * i) + k);
            else { out[16 * i + 4 * k] = w.x; out[16 * i + 4 * k + 1] = w.y; out[16 * i + 4 * k + 2] = w.z; out[16 * i + 4 * k + 3] = w.w; }
        }
    }
    if (blockIdx.x == 0) for (size_t i = 16 * n16 + threadIdx.x; i < n; i += blockDim.x) out[i] = src[i];
}

// a value that does not fit BYTES bytes cannot be stored: it is clamped and counted (the callers make sure the model's support
// fits, so a count > 0 means a decoder that reported garbage for an invalid stream: its status says so)
template <int BYTES>
__global__ __launch_bounds__(256) void narrow_kernel(const int32_t* __restrict__ in, void* __restrict__ out, size_t n) {
    using T = typename std::conditional<BYTES == 1, int8_t, int16_t>::type;
    constexpr int32_t lo = BYTES == 1 ? -128 : -32768, hi = BYTES == 1 ? 127 : 32767;
    T* dst = reinterpret_cast<T*>(out);
    const size_t n16 = n / 16;
    const bool aligned = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        T v[16];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v4i32 w;
            if (aligned) w = __builtin_nontemporal_load(reinterpret_cast<const v4i32*>(in + 16 * i) + k);
            else { w.x = in[16 * i + 4 * k]; w.y = in[16 * i + 4 * k + 1]; w.z = in[16 * i + 4 * k + 2]; w.w = in[16 * i + 4 * k + 3]; }
            v[4 * k] = (T)min(max(w.x, lo), hi); v[4 * k + 1] = (T)min(max(w.y, lo), hi);
            v[4 * k + 2] = (T)min(max(w.z, lo), hi); v[4 * k + 3] = (T)min(max(w.w, lo), hi);
        }
        if (aligned) {
#pragma unroll
            for (int k = 0; k < BYTES; ++k) __builtin_nontemporal_store(reinterpret_cast<const v4i32*>(v)[k], reinterpret_cast<v4i32*>(dst + 16 * i) + k);
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) dst[16 * i + k] = v[k];
        }
    }
    if (blockIdx.x == 0) for (size_t i = 16 * n16 + threadIdx.x; i < n; i += blockDim.x) dst[i] = (T)min(max(in[i], lo), hi);
}

static unsigned conv_grid(size_t n) {
    const size_t want = (n / 16 + 255) / 256;
    return (unsigned)(want < 1 ? 1 : (want > 256 * 32 ? 256 * 32 : want));
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
    if (!model || (symbol_bytes != 1 && symbol_bytes != 2) || (!d_scratch && n_streams * n_per_stream > 0)) return CST_ERR_INVALID_ARGUMENT;
    int32_t* wide = reinterpret_cast<int32_t*>((reinterpret_cast<uintptr_t>(d_scratch) + 15) & ~(uintptr_t)15);
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
    if (!model || (symbol_bytes != 1 && symbol_bytes != 2) || (!d_scratch && n_streams * n_per_stream > 0)) return CST_ERR_INVALID_ARGUMENT;
    if (!support_fits(model, symbol_bytes)) return CST_ERR_INVALID_ARGUMENT;      // a decoded symbol must be storable
    int32_t* wide = reinterpret_cast<int32_t*>((reinterpret_cast<uintptr_t>(d_scratch) + 15) & ~(uintptr_t)15);
    const cst_status rc = cst_ans_decode_batch(model, cfg, d_words, d_offsets, stride_words, words_capacity, d_n_words, wide, n_streams, n_per_stream,
                                               layout, d_state, d_n_words_out, d_status, flags, stream);
    if (rc != CST_OK) return rc;
    return cst_symbols_narrow(wide, n_streams * n_per_stream, d_symbols, symbol_bytes, stream);
}

} // extern "C"
