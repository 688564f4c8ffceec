// cst_ans_generic.hip -- the rest of the reference's (Word, State, PRECISION) grid (src/stream/stack.rs:1293-1356,
// tests/random_data.rs:161-192): AnsCoder<u32, u64> at PRECISION 25..32, AnsCoder<u16, u64>, AnsCoder<u8, u64>,
// AnsCoder<u8, u32>, AnsCoder<u8, u16>.  The presets SURVEY.md 8 names -- (32,64,24), (32,64,12), (16,32,12) -- and
// everything else on W = 32 / S = 64 / P <= 24 and W = 16 / S = 32 have hand-scheduled kernels; these combinations exist for
// completeness and take ONE compiler-scheduled kernel pair: one lane per stream, the state in 64 bits whatever S is, the
// cumulatives read from global memory (they stay in L2), `state / p` by the hardware's (software) 64-bit division, a binary
// search for the quantile.  Same recurrences, same results:
//     encode  stack.rs:1014-1048 (encode_symbol), :891-895 (into_compressed: the state's words, low first, zero high words dropped)
//     decode  stack.rs:299-318, 440-462 (from_compressed / read_initial_state), :1070-1100 (decode_symbol)
// Compressed words sit one per uint32 slot (low W bits), as everywhere in this ABI.
#include "cst_ans_kernels.hpp"

namespace cst {

struct GenericArgs {
    const uint32_t* cdf;          // [n_symbols + 1]; at P = 32 the last entry has wrapped to 0
    int32_t n_symbols, min_symbol;
    int32_t W, S, P;
    size_t n_streams, n_per_stream;
    int32_t layout;
    // encode
    const int32_t* symbols;
    uint32_t* words;
    size_t stride_words;
    uint32_t* n_words;
    // decode
    const uint32_t* words_in;
    const uint64_t* offsets;
    const uint32_t* n_words_in;
    uint64_t words_capacity;
    int32_t* symbols_out;
    uint32_t* n_words_out;
    uint64_t* state;
    int32_t* status;
    uint32_t flags;
};

__device__ __forceinline__ size_t sym_index(const GenericArgs& a, size_t s, size_t t) {
    return a.layout == CST_LAYOUT_STREAM_MAJOR ? s * a.n_per_stream + t : t * a.n_streams + s;
}

__global__ __launch_bounds__(kBlock) void ans_encode_generic_kernel(const GenericArgs a) {
    const size_t s = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (s >= a.n_streams) return;
    const int W = a.W, S = a.S, P = a.P;
    const uint64_t wmask = W == 32 ? 0xffffffffull : ((1ull << W) - 1ull);
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;
    uint32_t* out = a.words + s * a.stride_words;
    const uint64_t cap = a.stride_words;
    uint64_t st = raw ? a.state[s] : 0ull, len = 0;
    int32_t status = CST_STREAM_OK;
    for (size_t t = a.n_per_stream; t-- > 0;) {
        const int64_t i = (int64_t)a.symbols[sym_index(a, s, t)] - (int64_t)a.min_symbol;
        if (i < 0 || i >= a.n_symbols) { status = CST_STREAM_IMPOSSIBLE_SYMBOL; break; }   // before any mutation: stack.rs:1031-1033
        const uint32_t c = a.cdf[i], p = a.cdf[i + 1] - c;                                // (wrapping: right at P = 32 too)
        if ((st >> (S - P)) >= p) {
            if (len < cap) out[len] = (uint32_t)(st & wmask);
            ++len;
            st >>= W;
        }
        st = ((st / p) << P) | ((uint64_t)c + st % p);
    }
    if (status == CST_STREAM_OK && !raw) {
        for (uint64_t rest = st; rest != 0; rest >>= W) {      // bit_array_to_chunks_truncated, src/lib.rs:719-731
            if (len < cap) out[len] = (uint32_t)(rest & wmask);
            ++len;
        }
    }
    if (status == CST_STREAM_OK && len > cap) status = CST_STREAM_CAPACITY;
    if (raw) a.state[s] = st;
    a.status[s] = status;
    a.n_words[s] = status == CST_STREAM_OK ? (uint32_t)len : 0u;
}

__global__ __launch_bounds__(kBlock) void ans_decode_generic_kernel(const GenericArgs a) {
    const size_t s = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (s >= a.n_streams) return;
    const int W = a.W, S = a.S, P = a.P;
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;
    const WordSlice ws = word_slice(a.offsets, a.stride_words, a.n_words_in, s, a.words_capacity);
    const uint32_t* in = a.words_in + ws.off;
    uint64_t len = ws.n;
    int32_t status = ws.bad ? CST_STREAM_INVALID_DATA : CST_STREAM_OK;
    const uint64_t thresh = 1ull << (S - W), qmask = P == 32 ? 0xffffffffull : ((1ull << P) - 1ull);
    const uint32_t wmask = W == 32 ? 0xffffffffu : ((1u << W) - 1u);
    uint64_t st = 0;
    if (raw) {
        st = a.state[s];
    } else if (len > 0) {                                      // read_initial_state, stack.rs:440-462
        // a Word of the reference cannot carry bits above W: whatever else sits in a 32-bit slot is not part of the word
        const uint32_t first = in[--len] & wmask;
        if (first == 0u) { status = CST_STREAM_INVALID_DATA; len = 0; }
        else {
            st = first;
            while (len > 0) { st = (st << W) | (in[--len] & wmask); if (st >= thresh) break; }
        }
    }
    const int n = a.n_symbols;
    for (size_t t = 0; t < a.n_per_stream; ++t) {
        const uint32_t q = (uint32_t)(st & qmask);
        int lo = 0, hi = n - 1;                                 // the bin with cdf[i] <= q < cdf[i + 1]  (cdf[n] = 2^P, never read)
        while (lo < hi) {
            const int mid = lo + (hi - lo + 1) / 2;
            if (a.cdf[mid] <= q) lo = mid; else hi = mid - 1;
        }
        const uint32_t c = a.cdf[lo], p = a.cdf[lo + 1] - c;
        a.symbols_out[sym_index(a, s, t)] = a.min_symbol + lo;
        st = (st >> P) * (uint64_t)p + (uint64_t)(q - c);
        if (st < thresh && len > 0) st = (st << W) | (in[--len] & wmask);   // decoding past the end is legal: no refill, stack.rs:1062-1065
    }
    if (raw) a.state[s] = st;
    if (a.n_words_out) a.n_words_out[s] = (uint32_t)len;
    a.status[s] = status;
}

// the combinations of the reference's grid that the hand-scheduled kernels do not take
bool generic_config(cst_coder_config c) {
    const int W = c.word_bits, S = c.state_bits, P = c.precision;
    if (P < 1) return false;
    if (W == 32 && S == 64) return P > 24 && P <= 32;
    if (W == 16 && S == 64) return P <= 16;
    if (W == 8 && (S == 64 || S == 32 || S == 16)) return P <= 8;
    return false;
}

static cst_status launch_generic(bool encode, const GenericArgs& a, hipStream_t hs) {
    const size_t blocks = (a.n_streams + kBlock - 1) / kBlock;
    if (blocks == 0) return CST_OK;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    if (encode) hipLaunchKernelGGL(ans_encode_generic_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, hs, a);
    else hipLaunchKernelGGL(ans_decode_generic_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, hs, a);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

cst_status ans_encode_generic(const cst_model* m, cst_coder_config cfg, const int32_t* d_symbols, size_t n_streams, size_t n_per_stream,
                              cst_layout layout, uint32_t* d_words, size_t stride_words, uint32_t* d_n_words, uint64_t* d_state,
                              int32_t* d_status, uint32_t flags, hipStream_t hs) {
    if (m->per_stream || !m->d_cdf) return CST_ERR_INVALID_ARGUMENT;
    GenericArgs a{};
    a.cdf = m->d_cdf; a.n_symbols = m->n_symbols; a.min_symbol = m->min_symbol;
    a.W = cfg.word_bits; a.S = cfg.state_bits; a.P = cfg.precision;
    a.n_streams = n_streams; a.n_per_stream = n_per_stream; a.layout = (int32_t)layout;
    a.symbols = d_symbols; a.words = d_words; a.stride_words = stride_words; a.n_words = d_n_words;
    a.state = d_state; a.status = d_status; a.flags = flags;
    return launch_generic(true, a, hs);
}

cst_status ans_decode_generic(const cst_model* m, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_offsets, size_t stride_words,
                              size_t words_capacity, const uint32_t* d_n_words, int32_t* d_symbols, size_t n_streams, size_t n_per_stream,
                              cst_layout layout, uint64_t* d_state, uint32_t* d_n_words_out, int32_t* d_status, uint32_t flags, hipStream_t hs) {
    if (m->per_stream || !m->d_cdf) return CST_ERR_INVALID_ARGUMENT;
    GenericArgs a{};
    a.cdf = m->d_cdf; a.n_symbols = m->n_symbols; a.min_symbol = m->min_symbol;
    a.W = cfg.word_bits; a.S = cfg.state_bits; a.P = cfg.precision;
    a.n_streams = n_streams; a.n_per_stream = n_per_stream; a.layout = (int32_t)layout;
    a.words_in = d_words; a.offsets = d_offsets; a.stride_words = stride_words; a.n_words_in = d_n_words; a.words_capacity = words_capacity;
    a.symbols_out = d_symbols; a.n_words_out = d_n_words_out; a.state = d_state; a.status = d_status; a.flags = flags;
    return launch_generic(false, a, hs);
}

} // namespace cst
