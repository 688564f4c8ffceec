// cst_ans_ckpt.hip -- checkpointed ANS streams: the reference's Pos / Seek jump tables (src/stream/stack.rs:1107-1139,
// docs of `AnsCoder::pos` / `seek`, test :1456-1548) for the batched coder.
//
// An ANS stream decodes strictly in order, so ONE long stream (BASELINE config C1: 10^6 symbols) occupies one lane of
// the GPU however many there are.  The reference's answer is a jump table: while encoding, note (pos, state) = (words
// in the bulk, coder state) every K symbols; `seek(pos, state)` later resumes decoding right there.  Here the encoder
// records such a checkpoint in front of every chunk of K symbols and the decoder treats every (stream, chunk) pair as
// an independent coder: chunk j of stream s is AnsCoder::seek(pos[s][j], state[s][j]) followed by K decoded symbols.
// The compressed words are EXACTLY those of the plain encoder (checkpoints are side information), and decoding the
// chunks is the ordinary batched decode of n_streams * n_chunks "virtual streams" of K symbols (stream-major: chunk j
// of stream s is row s * n_chunks + j of the symbol matrix viewed as [n_streams * n_chunks][K]) with
// CST_FLAG_RAW_STATE -- every decode kernel of the library applies unchanged, including the hand-scheduled ones.
#include "cst_ans_kernels.hpp"

namespace cst {

struct CkptEncodeArgs {
    const int32_t* symbols;
    size_t n_streams, n_per_stream;
    int32_t layout;
    const EncEntry* enc;
    int32_t n_symbols, min_symbol, precision;
    uint32_t* words;
    size_t stride_words;
    uint32_t* n_words;
    size_t interval, n_chunks;
    uint32_t* ckpt_pos;
    uint64_t* ckpt_state;
    int32_t* status;
    int32_t table_in_lds;
};

// One lane per stream, generic steps (any supported preset).  Meant for FEW LONG streams: symbols are read straight
// from HBM (a batch of many short streams is better served by cst_ans_encode_batch and has lanes enough without
// checkpoints).
template <int W, int S>
__global__ __launch_bounds__(kBlock) void ans_encode_ckpt_kernel(const CkptEncodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem) + (threadIdx.x >> 6) * kRingWords;
    const EncEntry* table = a.enc;
    if (a.table_in_lds) {
        EncEntry* t = reinterpret_cast<EncEntry*>(smem + (size_t)(kBlock / kWave) * kRingWords * 4);
        for (int i = threadIdx.x; i < a.n_symbols; i += blockDim.x) t[i] = a.enc[i];
        table = t;
    }
    __syncthreads();
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = s < a.n_streams;
    const size_t N = a.n_per_stream;
    const int P = a.precision;
    const uint32_t nsym = (uint32_t)a.n_symbols;
    const size_t stride_t = a.layout == CST_LAYOUT_SYMBOL_MAJOR ? a.n_streams : 1;
    const int32_t* my = a.symbols + (active ? (a.layout == CST_LAYOUT_SYMBOL_MAJOR ? s : s * N) : 0);

    EncLane<W, S> L;
    L.init(a.words + (active ? s : 0) * a.stride_words,
           active ? (uint32_t)(a.stride_words > 0xffffffffull ? 0xffffffffull : a.stride_words) : 0u, ring, lane);
    int countdown = 4 * groups_per_point(W, P);
    size_t next_ckpt = a.n_chunks;            // checkpoint j is taken once symbol j * interval has been encoded
    for (size_t t = N; t-- > 0;) {
        const int32_t v = active ? my[t * stride_t] : a.min_symbol;
        L.template step<false>(table[enc_index(v, a.min_symbol, nsym, L.bad)], P);
        if (--countdown == 0) { countdown = 4 * groups_per_point(W, P); L.flush_chunks(); }
        if (next_ckpt > 0 && t == (next_ckpt - 1) * a.interval) {
            --next_ckpt;
            if (active) {
                a.ckpt_pos[s * a.n_chunks + next_ckpt] = L.out.wr;              // AnsCoder::pos(): (bulk.len(), state)
                a.ckpt_state[s * a.n_chunks + next_ckpt] = (uint64_t)L.state;
            }
        }
    }
    uint32_t n_words = 0;
    const int32_t status = L.finish(true, nsym, n_words);
    if (!active) return;
    a.status[s] = status;
    a.n_words[s] = (status == CST_STREAM_OK) ? n_words : 0u;
}

// the virtual streams' offsets (every chunk of stream s reads stream s's words) and states (the raw decode updates its state array:
// a copy) -- one launch in front of the decoder
__global__ void ckpt_offsets_kernel(const uint64_t* __restrict__ offsets, size_t stride_words, size_t n_streams, size_t n_chunks,
                                    const uint64_t* __restrict__ state_in, uint64_t* __restrict__ out, uint64_t* __restrict__ state_out) {
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_streams * n_chunks) return;
    const size_t s = v / n_chunks;
    out[v] = offsets ? offsets[s] : s * stride_words;
    state_out[v] = state_in[v];
}

__global__ void ckpt_status_per_stream_kernel(const int32_t* __restrict__ chunk_status, size_t n_streams, size_t n_chunks, int32_t* __restrict__ out) {
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    int32_t worst = 0;
    for (size_t j = 0; j < n_chunks; ++j) worst = max(worst, chunk_status[s * n_chunks + j]);
    out[s] = worst;
}

// decoding WITHOUT the jump points (the words are the plain encoder's): chunk 0's jump point is the whole stream --
// pos[s][0] words in the bulk, state[s][0] the final coder state
__global__ void ckpt_whole_stream_kernel(const uint32_t* __restrict__ pos, const uint64_t* __restrict__ state, size_t n_streams, size_t n_chunks,
                                         uint32_t* __restrict__ n_words, uint64_t* __restrict__ st) {
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    n_words[s] = pos[s * n_chunks];
    st[s] = state[s * n_chunks];
}

__global__ void ckpt_spread_status_kernel(const int32_t* __restrict__ per_stream, size_t n_streams, size_t n_chunks, int32_t* __restrict__ out) {
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n_streams * n_chunks) out[v] = per_stream[v / n_chunks];
}

__global__ void ans_ckpt_flag_kernel(const uint32_t* __restrict__ pos, size_t n_streams, size_t n_chunks, size_t stride, int32_t* __restrict__ status) {
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_streams * n_chunks) return;
    const uint32_t whole = pos[v / n_chunks * n_chunks];
    if (pos[v] > whole || (stride != 0 && whole > stride)) status[v] = CST_STREAM_INVALID_DATA;
}

cst_status flag_bad_jump_points(const uint32_t* d_ckpt_pos, size_t n_streams, size_t n_chunks, size_t stride_words, int32_t* d_status, hipStream_t hs) {
    const size_t n = n_streams * n_chunks;
    if (n == 0) return CST_OK;
    hipLaunchKernelGGL(ans_ckpt_flag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, hs, d_ckpt_pos, n_streams, n_chunks, stride_words, d_status);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

} // namespace cst

using namespace cst;

extern "C" {

cst_status cst_ans_encode_batch_ckpt(const cst_model* model, cst_coder_config cfg, const int32_t* d_symbols, size_t n_streams,
                                     size_t n_per_stream, cst_layout layout, uint32_t* d_words, size_t stride_words,
                                     uint32_t* d_n_words, size_t ckpt_interval, uint32_t* d_ckpt_pos, uint64_t* d_ckpt_state,
                                     int32_t* d_status, void* stream) {
    if (!model || !d_words || !d_n_words || !d_status || !d_ckpt_pos || !d_ckpt_state || ckpt_interval == 0) return CST_ERR_INVALID_ARGUMENT;
    if (n_per_stream > 0 && !d_symbols) return CST_ERR_INVALID_ARGUMENT;
    if (!config_supported(cfg) || cfg.precision != model->precision) return CST_ERR_INVALID_ARGUMENT;
    if (layout != CST_LAYOUT_STREAM_MAJOR && layout != CST_LAYOUT_SYMBOL_MAJOR) return CST_ERR_INVALID_ARGUMENT;
    if (n_streams == 0) return CST_OK;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != model->device) return CST_ERR_INVALID_ARGUMENT;
    if (model->per_stream) {
        // one table per stream (config C3): the compact-row encoder notes the jump points on its way (cst_ans_pt.hip), at
        // the speed of the plain encoder for chunks of whole 32-symbol tiles
        if (!pt_usable(model, cfg, layout, n_per_stream) || model->n_tables != n_streams) return CST_ERR_INVALID_ARGUMENT;
        return note_kernel("ans_encode_pt_kernel<ckpt>", ans_encode_pt_ckpt(model, d_symbols, n_streams, n_per_stream, d_words, stride_words, d_n_words,
                                                                             ckpt_interval, d_ckpt_pos, d_ckpt_state, d_status, (hipStream_t)stream));
    }
    {   // whole workgroups of aligned rows with chunks of whole tiles: the producer / consumer encoder notes the jump points on its
        // way (cst_ans_pc.hip, round 5) at the speed of cst_ans_encode_batch
        AnsEncodeArgs e{};
        e.symbols = d_symbols; e.n_streams = n_streams; e.n_per_stream = n_per_stream; e.enc = model->d_enc; e.n_symbols = model->n_symbols;
        e.min_symbol = model->min_symbol; e.precision = model->precision; e.words = d_words; e.stride_words = stride_words;
        e.n_words = d_n_words; e.state = nullptr; e.status = d_status; e.flags = 0;
        if (pc_encode_ckpt_usable(e, cfg, layout, ckpt_interval))
            return note_kernel(e.precision > 12 ? "ans_encode_pc_kernel<wide, ckpt>" : "ans_encode_pc_kernel<ckpt>", ans_encode_pc_ckpt(e, ckpt_interval, d_ckpt_pos, d_ckpt_state, (hipStream_t)stream));
    }
    CkptEncodeArgs a{};
    a.symbols = d_symbols; a.n_streams = n_streams; a.n_per_stream = n_per_stream; a.layout = layout; a.enc = model->d_enc;
    a.n_symbols = model->n_symbols; a.min_symbol = model->min_symbol; a.precision = model->precision; a.words = d_words;
    a.stride_words = stride_words; a.n_words = d_n_words; a.interval = ckpt_interval;
    a.n_chunks = (n_per_stream + ckpt_interval - 1) / ckpt_interval; a.ckpt_pos = d_ckpt_pos; a.ckpt_state = d_ckpt_state; a.status = d_status;
    const size_t ring_bytes = (size_t)(kBlock / kWave) * kRingWords * 4, table_bytes = (size_t)model->n_symbols * sizeof(EncEntry);
    a.table_in_lds = ring_bytes + table_bytes <= 150 * 1024;
    const size_t lds = ring_bytes + (a.table_in_lds ? table_bytes : 0);
    const size_t blocks = (n_streams + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    hipStream_t hs = (hipStream_t)stream;
    if (cfg.word_bits == 32) {
        if (lds > 64 * 1024) CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(ans_encode_ckpt_kernel<32, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((ans_encode_ckpt_kernel<32, 64>), dim3((unsigned)blocks), dim3(kBlock), lds, hs, a);
    } else {
        if (lds > 64 * 1024) CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(ans_encode_ckpt_kernel<16, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((ans_encode_ckpt_kernel<16, 32>), dim3((unsigned)blocks), dim3(kBlock), lds, hs, a);
    }
    CST_HIP_TRY(hipGetLastError());
    return note_kernel("ans_encode_ckpt_kernel", CST_OK);
}

size_t cst_ckpt_scratch_bytes(size_t n_streams, size_t n_per_stream, size_t ckpt_interval) {
    if (ckpt_interval == 0) return 0;
    return 16 * n_streams * ((n_per_stream + ckpt_interval - 1) / ckpt_interval) + 16;
}

cst_status cst_ans_decode_batch_ckpt(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_offsets,
                                     size_t stride_words, size_t words_capacity, size_t ckpt_interval, const uint32_t* d_ckpt_pos, const uint64_t* d_ckpt_state,
                                     int32_t* d_symbols, size_t n_streams, size_t n_per_stream, void* d_scratch, int32_t* d_status,
                                     void* stream) {
    if (!model || !d_ckpt_pos || !d_ckpt_state || !d_scratch || !d_status || ckpt_interval == 0) return CST_ERR_INVALID_ARGUMENT;
    if (n_per_stream % ckpt_interval != 0) return CST_ERR_INVALID_ARGUMENT;      // whole chunks only: rows of the virtual matrix
    if (n_streams == 0 || n_per_stream == 0) return CST_OK;
    const size_t n_chunks = n_per_stream / ckpt_interval, n_virtual = n_streams * n_chunks;
    hipStream_t hs = (hipStream_t)stream;
    if (model->per_stream) {
        if (!d_symbols || !config_supported(cfg) || cfg.precision != model->precision || model->n_tables != n_streams) return CST_ERR_INVALID_ARGUMENT;
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev != model->device) return CST_ERR_INVALID_ARGUMENT;
        // the lanes of a stream share its table in LDS: two waves per SIMD (cst_ans_pt.hip, ans_decode_pt_sub_kernel)
        if (n_chunks >= 2 && pt_sub_usable(model, cfg, n_streams, n_per_stream, ckpt_interval))
            return note_kernel("ans_decode_pt_sub_kernel", ans_decode_pt_sub(model, d_words, d_offsets, stride_words, words_capacity, ckpt_interval, d_ckpt_pos,
                                                                              d_ckpt_state, d_symbols, n_streams, n_per_stream, d_status, hs));
        // any other shape: the jump points are side information -- decode the streams whole (the first one IS the whole stream)
        uint64_t* w_state = reinterpret_cast<uint64_t*>(d_scratch);
        uint32_t* w_n = reinterpret_cast<uint32_t*>(w_state + n_streams);
        int32_t* w_status = reinterpret_cast<int32_t*>(w_n + n_streams);
        hipLaunchKernelGGL(ckpt_whole_stream_kernel, dim3((unsigned)((n_streams + 255) / 256)), dim3(256), 0, hs, d_ckpt_pos, d_ckpt_state, n_streams,
                           n_chunks, w_n, w_state);
        CST_HIP_TRY(hipGetLastError());
        const cst_status rc = cst_ans_decode_batch(model, cfg, d_words, d_offsets, stride_words, words_capacity, w_n, d_symbols, n_streams, n_per_stream,
                                                   CST_LAYOUT_STREAM_MAJOR, w_state, nullptr, w_status, CST_FLAG_RAW_STATE, stream);
        if (rc != CST_OK) return rc;
        hipLaunchKernelGGL(ckpt_spread_status_kernel, dim3((unsigned)((n_virtual + 255) / 256)), dim3(256), 0, hs, w_status, n_streams, n_chunks, d_status);
        CST_HIP_TRY(hipGetLastError());
        return CST_OK;
    }
    uint64_t* v_offsets = reinterpret_cast<uint64_t*>(d_scratch);
    uint64_t* v_state = v_offsets + n_virtual;
    hipLaunchKernelGGL(ckpt_offsets_kernel, dim3((unsigned)((n_virtual + 255) / 256)), dim3(256), 0, hs, d_offsets, stride_words, n_streams, n_chunks,
                       d_ckpt_state, v_offsets, v_state);
    CST_HIP_TRY(hipGetLastError());
    // every virtual stream's slice [off(s), off(s) + pos) is checked against the buffer (slab form: against all slabs)
    const size_t capacity = words_capacity ? words_capacity : (d_offsets ? 0 : n_streams * stride_words);
    const cst_status rc = cst_ans_decode_batch(model, cfg, d_words, v_offsets, 0, capacity, d_ckpt_pos, d_symbols, n_virtual, ckpt_interval,
                                               CST_LAYOUT_STREAM_MAJOR, v_state, nullptr, d_status, CST_FLAG_RAW_STATE, stream);
    if (rc != CST_OK) return rc;
    return flag_bad_jump_points(d_ckpt_pos, n_streams, n_chunks, d_offsets ? 0 : stride_words, d_status, hs);
}

// ---- narrow symbol matrices through jump points (round 5) ----

cst_status cst_ans_encode_batch_ckpt_sym(const cst_model* model, cst_coder_config cfg, const void* d_symbols, int32_t symbol_bytes, size_t n_streams,
                                         size_t n_per_stream, cst_layout layout, uint32_t* d_words, size_t stride_words, uint32_t* d_n_words,
                                         size_t ckpt_interval, uint32_t* d_ckpt_pos, uint64_t* d_ckpt_state, int32_t* d_status, void* d_scratch,
                                         void* stream) {
    if (symbol_bytes == 4)
        return cst_ans_encode_batch_ckpt(model, cfg, reinterpret_cast<const int32_t*>(d_symbols), n_streams, n_per_stream, layout, d_words, stride_words,
                                         d_n_words, ckpt_interval, d_ckpt_pos, d_ckpt_state, d_status, stream);
    if (!model || (symbol_bytes != 1 && symbol_bytes != 2) || !d_words || !d_n_words || !d_status || !d_ckpt_pos || !d_ckpt_state || ckpt_interval == 0)
        return CST_ERR_INVALID_ARGUMENT;
    if (n_streams == 0) return CST_OK;
    if (d_symbols && symbol_bytes == 1 && model->per_stream && config_supported(cfg) && cfg.precision == model->precision) {
        // one table per stream, int8 matrix (round 6): the compact-row encoder reads the int8 tiles itself
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev != model->device) return CST_ERR_INVALID_ARGUMENT;
        if (pt_n8_encode_usable(model, cfg, layout, d_symbols, n_streams, n_per_stream, ckpt_interval))
            return note_kernel("ans_encode_pt_n8_kernel<ckpt>", ans_encode_pt_ckpt_n8(model, d_symbols, n_streams, n_per_stream, d_words, stride_words, d_n_words,
                                                                                       ckpt_interval, d_ckpt_pos, d_ckpt_state, d_status, (hipStream_t)stream));
    }
    if (d_symbols && !model->per_stream && !model->d_symbol_of_index && config_supported(cfg) && cfg.precision == model->precision) {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev != model->device) return CST_ERR_INVALID_ARGUMENT;
        AnsEncodeArgs e{};
        e.symbols = reinterpret_cast<const int32_t*>(d_symbols); e.n_streams = n_streams; e.n_per_stream = n_per_stream; e.enc = model->d_enc;
        e.n_symbols = model->n_symbols; e.min_symbol = model->min_symbol; e.precision = model->precision; e.words = d_words;
        e.stride_words = stride_words; e.n_words = d_n_words; e.state = nullptr; e.status = d_status; e.flags = 0;
        if (symbol_bytes == 1 && pc_n8_encode_ckpt_usable(e, cfg, layout, ckpt_interval))      // the int8 matrix inside the loops, jump points on the way
            return note_kernel(e.precision > 12 ? "ans_encode_pc_n8_kernel<wide, ckpt>" : "ans_encode_pc_n8_kernel<ckpt>", ans_encode_pc_n8_ckpt(e, ckpt_interval, d_ckpt_pos, d_ckpt_state, (hipStream_t)stream));
        if (symbol_bytes == 2 && pc_n16_encode_ckpt_usable(e, cfg, layout, ckpt_interval))
            return note_kernel(e.precision > 12 ? "ans_encode_pc_n16_kernel<wide, ckpt>" : "ans_encode_pc_n16_kernel<ckpt>", ans_encode_pc_n16(e, ckpt_interval, d_ckpt_pos, d_ckpt_state, (hipStream_t)stream));
    }
    if (!d_scratch && n_streams * n_per_stream > 0) return CST_ERR_INVALID_ARGUMENT;
    int32_t* wide = reinterpret_cast<int32_t*>((reinterpret_cast<uintptr_t>(d_scratch) + 15) & ~(uintptr_t)15);
    const cst_status rc = cst_symbols_widen(d_symbols, symbol_bytes, n_streams * n_per_stream, wide, stream);
    if (rc != CST_OK) return rc;
    return cst_ans_encode_batch_ckpt(model, cfg, wide, n_streams, n_per_stream, layout, d_words, stride_words, d_n_words, ckpt_interval, d_ckpt_pos,
                                     d_ckpt_state, d_status, stream);
}

cst_status cst_ckpt_status_per_stream(const int32_t* d_chunk_status, size_t n_streams, size_t n_chunks, int32_t* d_stream_status, void* stream) {
    if (n_streams == 0) return CST_OK;
    if (!d_chunk_status || !d_stream_status || n_chunks == 0) return CST_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(ckpt_status_per_stream_kernel, dim3((unsigned)((n_streams + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_chunk_status, n_streams,
                       n_chunks, d_stream_status);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

size_t cst_ckpt_sym_scratch_bytes(size_t n_streams, size_t n_per_stream, size_t ckpt_interval, int32_t symbol_bytes) {
    return cst_ckpt_scratch_bytes(n_streams, n_per_stream, ckpt_interval) + cst_symbols_scratch_bytes(n_streams, n_per_stream, symbol_bytes) + 16;
}

cst_status cst_ans_decode_batch_ckpt_sym(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_offsets,
                                         size_t stride_words, size_t words_capacity, size_t ckpt_interval, const uint32_t* d_ckpt_pos,
                                         const uint64_t* d_ckpt_state, void* d_symbols, int32_t symbol_bytes, size_t n_streams, size_t n_per_stream,
                                         void* d_scratch, int32_t* d_status, void* stream) {
    if (symbol_bytes == 4)
        return cst_ans_decode_batch_ckpt(model, cfg, d_words, d_offsets, stride_words, words_capacity, ckpt_interval, d_ckpt_pos, d_ckpt_state,
                                         reinterpret_cast<int32_t*>(d_symbols), n_streams, n_per_stream, d_scratch, d_status, stream);
    if (!model || (symbol_bytes != 1 && symbol_bytes != 2) || !d_ckpt_pos || !d_ckpt_state || !d_scratch || !d_status || ckpt_interval == 0)
        return CST_ERR_INVALID_ARGUMENT;
    if (n_per_stream % ckpt_interval != 0) return CST_ERR_INVALID_ARGUMENT;
    if (n_streams == 0 || n_per_stream == 0) return CST_OK;
    const size_t ck_bytes = (cst_ckpt_scratch_bytes(n_streams, n_per_stream, ckpt_interval) + 15) & ~(size_t)15;
    void* conv_scratch = reinterpret_cast<unsigned char*>(d_scratch) + ck_bytes;
    if (model->per_stream) {
        if (symbol_bytes == 1 && d_symbols && config_supported(cfg) && cfg.precision == model->precision && model->n_tables == n_streams &&
            n_per_stream / ckpt_interval >= 2 && pt_sub_n8_usable(model, cfg, n_streams, n_per_stream, ckpt_interval, d_symbols)) {
            // int8 matrix (round 6): the sub-lane decoder's byte tiles hold the symbols and leave as they are
            int dev = -1;
            if (hipGetDevice(&dev) != hipSuccess || dev != model->device) return CST_ERR_INVALID_ARGUMENT;
            return note_kernel("ans_decode_pt_sub_n8_kernel", ans_decode_pt_sub(model, d_words, d_offsets, stride_words, words_capacity, ckpt_interval, d_ckpt_pos,
                                                                                 d_ckpt_state, reinterpret_cast<int32_t*>(d_symbols), n_streams, n_per_stream,
                                                                                 d_status, (hipStream_t)stream, 1));
        }
        // any other shape: the sub-lane decoders write int32 -- decode into the wide matrix, narrow behind them
        int32_t* wide = reinterpret_cast<int32_t*>((reinterpret_cast<uintptr_t>(conv_scratch) + 15) & ~(uintptr_t)15);
        const cst_status rc = cst_ans_decode_batch_ckpt(model, cfg, d_words, d_offsets, stride_words, words_capacity, ckpt_interval, d_ckpt_pos, d_ckpt_state,
                                                        wide, n_streams, n_per_stream, d_scratch, d_status, stream);
        if (rc != CST_OK) return rc;
        return cst_symbols_narrow(wide, n_streams * n_per_stream, d_symbols, symbol_bytes, stream);
    }
    // shared table: the ordinary batched decode of the n_streams * n_chunks virtual streams (cst_ans_decode_batch_ckpt above), into
    // the narrow matrix -- int8 rows of whole 128-symbol lines are written by the decoder loops themselves (cst_ans_n8.hip)
    const size_t n_chunks = n_per_stream / ckpt_interval, n_virtual = n_streams * n_chunks;
    hipStream_t hs = (hipStream_t)stream;
    uint64_t* v_offsets = reinterpret_cast<uint64_t*>(d_scratch);
    uint64_t* v_state = v_offsets + n_virtual;
    hipLaunchKernelGGL(ckpt_offsets_kernel, dim3((unsigned)((n_virtual + 255) / 256)), dim3(256), 0, hs, d_offsets, stride_words, n_streams, n_chunks,
                       d_ckpt_state, v_offsets, v_state);
    CST_HIP_TRY(hipGetLastError());
    const size_t capacity = words_capacity ? words_capacity : (d_offsets ? 0 : n_streams * stride_words);
    const cst_status rc = cst_ans_decode_batch_sym(model, cfg, d_words, v_offsets, 0, capacity, d_ckpt_pos, d_symbols, symbol_bytes, n_virtual, ckpt_interval,
                                                   CST_LAYOUT_STREAM_MAJOR, v_state, nullptr, d_status, CST_FLAG_RAW_STATE, conv_scratch, stream);
    if (rc != CST_OK) return rc;
    return flag_bad_jump_points(d_ckpt_pos, n_streams, n_chunks, d_offsets ? 0 : stride_words, d_status, hs);
}

} // extern "C"
