// cst_ans_ragged.hip -- batched ANS for streams of DIFFERENT lengths: thousands of small coders in one launch.
//
// The reference's own usage pattern next to "one long message" is "many short ones": a compressed index whose documents
// are coded one AnsCoder each with a shared model (tests/issue52.rs: `coder.encode_symbol` per character, last to first,
// `into_compressed` per document; `from_compressed` + `decode_symbol` per document).  Through the single-coder drop-in
// that is one device round trip per document; here it is one launch for all of them:
//     symbols of stream s   = d_symbols[d_sym_offsets[s] .. d_sym_offsets[s + 1])        (a CSR-style ragged array)
//     words of stream s     = d_words[off(s) .. off(s) + d_n_words[s]),  off(s) = d_word_offsets ? d_word_offsets[s] : s * stride
// One lane per stream, a wave runs as many steps as its longest stream (the others idle behind a predicate: sort or bucket
// the documents by length if they differ by orders of magnitude); tables stay in HBM / L2 (short streams: staging 2^P
// entries per workgroup would cost more than the lookups), words go through the per-lane LDS rings of the other kernels.
// Every stream's words are those of cst_ans_encode_batch for that stream alone (stack.rs:835-849, 891-895; 1070-1100).
#include "cst_ans_kernels.hpp"

namespace cst {

struct RaggedArgs {
    const int32_t* symbols_in;
    int32_t* symbols_out;
    const uint64_t* sym_offsets;     // [n_streams + 1]
    size_t n_streams;
    const EncEntry* enc;
    const uint32_t* cdf;
    const uint16_t* bucket;
    int32_t bucket_bits, n_symbols, min_symbol, precision;
    uint32_t* words_out;
    const uint32_t* words_in;
    const uint64_t* word_offsets;    // [n_streams + 1] (encode: slab of stream s = [off[s], off[s + 1])) or null
    size_t stride_words;
    uint32_t* n_words_out;
    const uint32_t* n_words_in;
    int32_t* status;
    uint64_t words_capacity;
};

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d));
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

template <int W, int S>
__global__ __launch_bounds__(kBlock) void ans_encode_ragged_kernel(const RaggedArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem) + (threadIdx.x >> 6) * kRingWords;
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s - lane >= a.n_streams) return;
    const bool active = s < a.n_streams;
    const int P = a.precision;
    const uint32_t nsym = (uint32_t)a.n_symbols;
    const uint64_t sym_lo = active ? a.sym_offsets[s] : 0, sym_hi = active ? a.sym_offsets[s + 1] : 0;
    const bool too_long = sym_hi - sym_lo > 0xffffffffull || sym_hi < sym_lo;
    const uint32_t len = too_long ? 0u : (uint32_t)(sym_hi - sym_lo);
    const uint64_t slab_lo = !active ? 0 : (a.word_offsets ? a.word_offsets[s] : (uint64_t)s * a.stride_words);
    const uint64_t slab_n = !active ? 0 : (a.word_offsets ? a.word_offsets[s + 1] - slab_lo : (uint64_t)a.stride_words);
    EncLane<W, S> L;
    L.init(a.words_out + slab_lo, (uint32_t)(slab_n > 0xffffffffull ? 0xffffffffull : slab_n), ring, lane);
    const int32_t* row = a.symbols_in + sym_lo;
    const uint32_t mx = wave_max_u32(len);
    // encode_iid_symbols_reverse: last symbol first (stack.rs:835-849); one step pushes at most one word, one scheduled point
    // per step moves every complete 16-byte chunk out of the ring
    for (uint32_t k = mx; k-- > 0;) {
        if (k < len) L.template step<false>(a.enc[enc_index(row[k], a.min_symbol, nsym, L.bad)], P);
        L.flush_chunks();
    }
    uint32_t n_words = 0;
    int32_t status = L.finish(true, nsym, n_words);
    if (too_long) status = CST_STREAM_CAPACITY;
    if (!active) return;
    a.n_words_out[s] = status == CST_STREAM_OK ? n_words : 0u;
    a.status[s] = status;
}

template <int W, int S>
__global__ __launch_bounds__(kBlock) void ans_decode_ragged_kernel(const RaggedArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem) + (threadIdx.x >> 6) * kRingWords;
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s - lane >= a.n_streams) return;
    const bool active = s < a.n_streams;
    const int P = a.precision;
    const int bucket_shift = P - a.bucket_bits;
    const uint64_t sym_lo = active ? a.sym_offsets[s] : 0, sym_hi = active ? a.sym_offsets[s + 1] : 0;
    const bool too_long = sym_hi - sym_lo > 0xffffffffull || sym_hi < sym_lo;
    const uint32_t len = too_long ? 0u : (uint32_t)(sym_hi - sym_lo);
    DecLane<W, S> L;
    const WordSlice ws = active ? word_slice(a.word_offsets, a.stride_words, a.n_words_in, s, a.words_capacity) : WordSlice{0, 0u, false};
    L.init(a.words_in + ws.off, ws.n, ring, lane);
    L.read_initial_state();
    L.in.prime();
    wave_lds_fence();
    const DecLut lut{};                                  // no staged image: cdf + bucket index straight from HBM / L2
    int32_t* row = a.symbols_out + sym_lo;
    const uint32_t mx = wave_max_u32(len);
    for (uint32_t k = 0; k < mx; ++k) {
        if (k < len) row[k] = a.min_symbol + (int32_t)ans_decode_step<W, S, kDecBucket, false>(L, lut, a.cdf, a.bucket, bucket_shift, a.n_symbols, P);
        L.in.advance_window();
    }
    if (!active) return;
    a.status[s] = ws.bad ? (int32_t)CST_STREAM_INVALID_DATA : (too_long ? (int32_t)CST_STREAM_CAPACITY : L.status);
}

// The reference's index stores no lengths: a document ends where its terminator symbol is decoded (tests/issue52.rs:63-80).
// First pass of that: every stream is decoded until `eof_index` appears (or `max_symbols` were decoded: CST_STREAM_CAPACITY,
// the output a caller would size from it could not hold more), nothing is stored but the count -- terminator included.  A
// prefix sum of the counts is the d_sym_offsets of cst_ans_decode_ragged.
template <int W, int S>
__global__ __launch_bounds__(kBlock) void ans_count_until_kernel(const RaggedArgs a, uint32_t eof_index, uint64_t max_symbols, uint64_t* lengths) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem) + (threadIdx.x >> 6) * kRingWords;
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s - lane >= a.n_streams) return;
    const bool active = s < a.n_streams;
    const int P = a.precision;
    const int bucket_shift = P - a.bucket_bits;
    DecLane<W, S> L;
    const WordSlice ws = active ? word_slice(a.word_offsets, a.stride_words, a.n_words_in, s, a.words_capacity) : WordSlice{0, 0u, false};
    L.init(a.words_in + ws.off, ws.n, ring, lane);
    L.read_initial_state();
    L.in.prime();
    wave_lds_fence();
    const DecLut lut{};
    uint64_t n = 0;
    bool done = !active || ws.bad || L.status != CST_STREAM_OK || max_symbols == 0;
    bool found = false;
    while (__any(!done)) {
        if (!done) {
            const uint32_t idx = ans_decode_step<W, S, kDecBucket, false>(L, lut, a.cdf, a.bucket, bucket_shift, a.n_symbols, P);
            ++n;
            found = idx == eof_index;
            done = found || n >= max_symbols;
        }
        L.in.advance_window();
    }
    if (!active) return;
    lengths[s] = n;
    a.status[s] = ws.bad ? (int32_t)CST_STREAM_INVALID_DATA : (L.status != CST_STREAM_OK ? L.status : (found ? (int32_t)CST_STREAM_OK : (int32_t)CST_STREAM_CAPACITY));
}

template <typename K>
static cst_status ragged_launch(K kernel, const RaggedArgs& a, hipStream_t hs) {
    const size_t blocks = (a.n_streams + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    const size_t lds = (size_t)(kBlock / kWave) * kRingWords * 4;
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kBlock), lds, hs, a);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

cst_status ans_encode_ragged(const cst_model* model, cst_coder_config cfg, const int32_t* d_symbols, const uint64_t* d_sym_offsets,
                             size_t n_streams, uint32_t* d_words, const uint64_t* d_word_offsets, size_t stride_words,
                             uint32_t* d_n_words, int32_t* d_status, hipStream_t hs) {
    RaggedArgs a{};
    a.symbols_in = d_symbols; a.sym_offsets = d_sym_offsets; a.n_streams = n_streams; a.enc = model->d_enc;
    a.n_symbols = model->n_symbols; a.min_symbol = model->min_symbol; a.precision = model->precision;
    a.words_out = d_words; a.word_offsets = d_word_offsets; a.stride_words = stride_words; a.n_words_out = d_n_words; a.status = d_status;
    if (cfg.word_bits == 32) return ragged_launch(ans_encode_ragged_kernel<32, 64>, a, hs);
    return ragged_launch(ans_encode_ragged_kernel<16, 32>, a, hs);
}

cst_status ans_decode_ragged(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_word_offsets,
                             size_t stride_words, size_t words_capacity, const uint32_t* d_n_words, int32_t* d_symbols,
                             const uint64_t* d_sym_offsets, size_t n_streams, int32_t* d_status, hipStream_t hs) {
    RaggedArgs a{};
    a.symbols_out = d_symbols; a.sym_offsets = d_sym_offsets; a.n_streams = n_streams; a.cdf = model->d_cdf; a.bucket = model->d_bucket;
    a.bucket_bits = model->bucket_bits; a.n_symbols = model->n_symbols; a.min_symbol = model->min_symbol; a.precision = model->precision;
    a.words_in = d_words; a.word_offsets = d_word_offsets; a.stride_words = stride_words; a.n_words_in = d_n_words; a.status = d_status;
    a.words_capacity = words_capacity;
    if (cfg.word_bits == 32) return ragged_launch(ans_decode_ragged_kernel<32, 64>, a, hs);
    return ragged_launch(ans_decode_ragged_kernel<16, 32>, a, hs);
}

cst_status ans_count_until(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_word_offsets,
                           size_t stride_words, size_t words_capacity, const uint32_t* d_n_words, size_t n_streams, int32_t eof_symbol,
                           size_t max_symbols, uint64_t* d_lengths, int32_t* d_status, hipStream_t hs) {
    RaggedArgs a{};
    a.n_streams = n_streams; a.cdf = model->d_cdf; a.bucket = model->d_bucket;
    a.bucket_bits = model->bucket_bits; a.n_symbols = model->n_symbols; a.min_symbol = model->min_symbol; a.precision = model->precision;
    a.words_in = d_words; a.word_offsets = d_word_offsets; a.stride_words = stride_words; a.n_words_in = d_n_words; a.status = d_status;
    a.words_capacity = words_capacity;
    const size_t blocks = (n_streams + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    const size_t lds = (size_t)(kBlock / kWave) * kRingWords * 4;
    const uint32_t eof_index = (uint32_t)eof_symbol - (uint32_t)model->min_symbol;
    if (cfg.word_bits == 32)
        hipLaunchKernelGGL((ans_count_until_kernel<32, 64>), dim3((unsigned)blocks), dim3(kBlock), lds, hs, a, eof_index, (uint64_t)max_symbols, d_lengths);
    else
        hipLaunchKernelGGL((ans_count_until_kernel<16, 32>), dim3((unsigned)blocks), dim3(kBlock), lds, hs, a, eof_index, (uint64_t)max_symbols, d_lengths);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

} // namespace cst
