// cst_range.hip -- batched range coder with a shared table (BASELINE config C4): kernels, launch, C ABI.
// The per-lane state machines live in cst_range_kernels.hpp.
#include "cst_range_kernels.hpp"

namespace cst {


// GLOBAL_TABLE: the encoder entries stay in HBM / L2 (alphabets too large for LDS)
template <int W, int S, int LAYOUT, bool VEC, int G, bool GLOBAL_TABLE = false>
__global__ __launch_bounds__(kBlock) void range_encode_kernel(const RangeEncodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const EncEntry* table;
    const size_t table_bytes = GLOBAL_TABLE ? 0 : ((((size_t)a.n_symbols * sizeof(EncEntry)) + 15) & ~(size_t)15);
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = threadIdx.x >> 6;
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem + table_bytes) + wave_in_block * kRingWords;
    int32_t* tile = reinterpret_cast<int32_t*>(smem + table_bytes + (size_t)(kBlock / kWave) * kRingWords * 4) +
                    wave_in_block * (kWave * kTileStride);
    if constexpr (GLOBAL_TABLE) {
        table = a.enc;
    } else {
        EncEntry* t = reinterpret_cast<EncEntry*>(smem);
        for (int i = threadIdx.x; i < a.n_symbols; i += blockDim.x) t[i] = a.enc[i];
        table = t;
    }
    __syncthreads();

    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t s0 = wave * kWave;
    if (s0 >= a.n_streams) return;
    const size_t s = s0 + lane;
    const bool active = s < a.n_streams;
    const size_t N = a.n_per_stream;
    const int P = a.precision;
    const uint32_t nsym = (uint32_t)a.n_symbols;

    RangeEncLane<W, S> L;
    L.init(a.words + (active ? s : 0) * a.stride_words,
           active ? (uint32_t)(a.stride_words > 0xffffffffull ? 0xffffffffull : a.stride_words) : 0u, ring, lane);
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;
    if (raw && active) {
        const cst_range_state r = a.rstate[s];
        L.lower = (typename StateT<S>::type)r.lower; L.range = (typename StateT<S>::type)r.range;
        L.inv_n = r.inverted_n; L.inv_first = r.inverted_first;
    }

    auto code = [&](int32_t v) {
        const EncEntry e = table[enc_index(v, a.min_symbol, nsym, L.bad)];
        L.step(e.c, e.p, P);
    };

    if constexpr (LAYOUT == CST_LAYOUT_SYMBOL_MAJOR) {
        const int32_t* col = a.symbols + (active ? s : 0);
        int countdown = 4 * G;
        // four symbols per request, one group ahead of their use (a lone wave has nobody to hide a load behind)
        size_t t = 0;
        int32_t nxt[4] = {0, 0, 0, 0};
        if (N >= 4 && active) {
#pragma unroll
            for (int j = 0; j < 4; ++j) nxt[j] = col[(size_t)j * a.n_streams];
        }
        for (; t + 4 <= N; t += 4) {
            int32_t cur[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
            if (t + 8 <= N && active) {
#pragma unroll
                for (int j = 0; j < 4; ++j) nxt[j] = col[(t + 4 + j) * a.n_streams];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                code(cur[j]);
                if (--countdown == 0) { countdown = 4 * G; L.out.flush_chunks(); }
            }
        }
        for (; t < N; ++t) {
            code(active ? col[t * a.n_streams] : 0);
            if (--countdown == 0) { countdown = 4 * G; L.out.flush_chunks(); }
        }
    } else {
        const int32_t* row = a.symbols + (active ? s : 0) * N;
        const size_t n_full = N / kTileSyms;
        if (n_full > 0) {
            int32_t r[kTileSyms];
            tile_fetch<VEC>(a.symbols, a.n_streams, N, s0, 0, lane, r);
            for (size_t tb = 0; tb < n_full; ++tb) {
                wave_lds_fence();
                tile_to_lds<VEC>(tile, lane, r);
                wave_lds_fence();
                if (tb + 1 < n_full) tile_fetch<VEC>(a.symbols, a.n_streams, N, s0, (tb + 1) * kTileSyms, lane, r);
                const int32_t* my = tile + lane * kTileStride;
                // Quads run the branch-free step_inline; a quad in which some lane leaves a long Inverted run is rolled
                // back and repeated with the general step (rare).  Entries of quad j+1 are fetched before quad j is coded.
                auto entry = [&](int32_t v) { return table[enc_index(v, a.min_symbol, nsym, L.bad)]; };
                int4 v = *reinterpret_cast<const int4*>(my);
                EncEntry e0 = entry(v.x), e1 = entry(v.y), e2 = entry(v.z), e3 = entry(v.w);
#pragma unroll
                for (int j = 0; j < kTileSyms / 4; ++j) {
                    EncEntry n0 = e0, n1 = e1, n2 = e2, n3 = e3;
                    if (j + 1 < kTileSyms / 4) {
                        v = *reinterpret_cast<const int4*>(my + 4 * (j + 1));
                        n0 = entry(v.x); n1 = entry(v.y); n2 = entry(v.z); n3 = entry(v.w);
                    }
                    const auto lower0 = L.lower, range0 = L.range;
                    const uint32_t wr0 = L.out.wr, inv_n0 = L.inv_n, inv_first0 = L.inv_first;
                    bool slow = false;
                    L.step_inline(e0.c, e0.p, P, slow); L.step_inline(e1.c, e1.p, P, slow);
                    L.step_inline(e2.c, e2.p, P, slow); L.step_inline(e3.c, e3.p, P, slow);
                    if (__any(slow)) {
                        L.lower = lower0; L.range = range0; L.out.wr = wr0; L.inv_n = inv_n0; L.inv_first = inv_first0;
                        L.step(e0.c, e0.p, P); L.step(e1.c, e1.p, P); L.step(e2.c, e2.p, P); L.step(e3.c, e3.p, P);
                    }
                    e0 = n0; e1 = n1; e2 = n2; e3 = n3;
                    if ((j + 1) % G == 0) L.out.flush_chunks();
                }
            }
        }
        for (size_t t = n_full * kTileSyms; t < N; ++t) {
            code(active ? row[t] : 0);
            L.out.flush_chunks();
        }
    }

    uint32_t n_words = 0;
    int32_t status;
    if (raw) {
        L.out.drain();
        n_words = L.out.wr;
        status = L.bad >= nsym ? CST_STREAM_IMPOSSIBLE_SYMBOL : (L.out.wr > L.out.cap ? CST_STREAM_CAPACITY : CST_STREAM_OK);
    } else {
        status = L.finish(nsym, n_words);
    }
    if (!active) return;
    if (raw) {
        cst_range_state r = a.rstate[s];
        r.lower = (uint64_t)L.lower; r.range = (uint64_t)L.range; r.inverted_n = L.inv_n; r.inverted_first = L.inv_first;
        a.rstate[s] = r;
    }
    a.status[s] = status;
    a.n_words[s] = (status == CST_STREAM_OK) ? n_words : 0u;
}

// ------------------------------------------------------------------------------------------------


template <int W, int S, int LAYOUT, bool VEC, int MODE, bool LUT_IN_LDS, int G>
__global__ __launch_bounds__(kBlock) void range_decode_kernel(const RangeDecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = threadIdx.x >> 6;
    const int P = a.precision;
    DecLut lut{};
    const uint32_t* cdf = a.cdf;
    const uint16_t* bucket = a.bucket;
    size_t lds_off = stage_decoder_tables<MODE, LUT_IN_LDS>(smem, P, a.dec_cp, a.dec_idx, a.cdf, a.bucket, a.bucket_bits,
                                                           a.n_symbols, lut, cdf, bucket);
    lds_off = (lds_off + 15) & ~(size_t)15;
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem + lds_off) + wave_in_block * kRingWords;
    int32_t* tile = reinterpret_cast<int32_t*>(smem + lds_off + (size_t)(kBlock / kWave) * kRingWords * 4) +
                    wave_in_block * (kWave * kTileStride);
    __syncthreads();

    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t s0 = wave * kWave;
    if (s0 >= a.n_streams) return;
    const size_t s = s0 + lane;
    const bool active = s < a.n_streams;
    const size_t N = a.n_per_stream;
    const int bucket_shift = P - a.bucket_bits;

    RangeDecLane<W, S> L;
    const WordSlice ws = active ? word_slice(a.offsets, a.stride_words, a.n_words, s, a.words_capacity) : WordSlice{0, 0u, false};
    L.init(a.words + ws.off, ws.n, ring, lane);
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;
    if (raw && active) {
        const cst_range_state r = a.rstate[s];
        L.lower = (typename StateT<S>::type)r.lower; L.range = (typename StateT<S>::type)r.range;
        L.point = (typename StateT<S>::type)r.point; L.in.pos = (uint32_t)r.position;
    }
    L.in.prime();
    wave_lds_fence();

    if constexpr (LAYOUT == CST_LAYOUT_SYMBOL_MAJOR) {
        int32_t* col = a.symbols + (active ? s : 0);
        int countdown = 4 * G;
        for (size_t t = 0; t < N; ++t) {
            const uint32_t idx = L.template step<MODE>(lut, cdf, bucket, bucket_shift, a.n_symbols, P);
            if (active) col[t * a.n_streams] = a.min_symbol + (int32_t)idx;
            if (--countdown == 0) { countdown = 4 * G; L.in.advance_window(); }
        }
    } else {
        int32_t* row = a.symbols + (active ? s : 0) * N;
        const size_t n_full = N / kTileSyms;
        int32_t* my = tile + lane * kTileStride;
        for (size_t tb = 0; tb < n_full; ++tb) {
#pragma unroll
            for (int j = 0; j < kTileSyms / 4; ++j) {
                int4 v;
                v.x = a.min_symbol + (int32_t)L.template step<MODE>(lut, cdf, bucket, bucket_shift, a.n_symbols, P);
                v.y = a.min_symbol + (int32_t)L.template step<MODE>(lut, cdf, bucket, bucket_shift, a.n_symbols, P);
                v.z = a.min_symbol + (int32_t)L.template step<MODE>(lut, cdf, bucket, bucket_shift, a.n_symbols, P);
                v.w = a.min_symbol + (int32_t)L.template step<MODE>(lut, cdf, bucket, bucket_shift, a.n_symbols, P);
                *reinterpret_cast<int4*>(my + 4 * j) = v;
                if ((j + 1) % G == 0) L.in.advance_window();
            }
            wave_lds_fence();
            tile_store<VEC>(a.symbols, a.n_streams, N, s0, tb * kTileSyms, lane, tile);
            wave_lds_fence();
        }
        for (size_t t = n_full * kTileSyms; t < N; ++t) {
            const uint32_t idx = L.template step<MODE>(lut, cdf, bucket, bucket_shift, a.n_symbols, P);
            if (active) row[t] = a.min_symbol + (int32_t)idx;
            L.in.advance_window();
        }
    }
    if (!active) return;
    a.status[s] = ws.bad ? (int32_t)CST_STREAM_INVALID_DATA : L.status;
    if (raw) {
        cst_range_state r = a.rstate[s];
        r.lower = (uint64_t)L.lower; r.range = (uint64_t)L.range; r.point = (uint64_t)L.point; r.position = L.in.pos;
        a.rstate[s] = r;
    }
}

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static constexpr size_t kPerBlockLds = (size_t)(kBlock / kWave) * (kRingWords * sizeof(uint32_t) + kWave * kTileStride * sizeof(int32_t));
static constexpr size_t kMaxLds = 160 * 1024;

template <typename K, typename A>
static cst_status launch(K kernel, size_t n_streams, size_t lds_bytes, hipStream_t hs, const A& args) {
    const size_t blocks = (n_streams + kBlock - 1) / kBlock;
    if (blocks == 0) return CST_OK;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    if (lds_bytes > 64 * 1024)
        CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kBlock), lds_bytes, hs, args);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

template <int W, int S, int G>
static cst_status range_encode_g(const RangeEncodeArgs& a, cst_layout layout, hipStream_t hs) {
    const size_t table_bytes = (((size_t)a.n_symbols * sizeof(EncEntry)) + 15) & ~(size_t)15;
    const size_t lds = table_bytes + kPerBlockLds;
    const bool vec = (a.n_per_stream % 4 == 0) && aligned16(a.symbols);
    if (lds > kMaxLds) {   // alphabet too large for an LDS-resident table: entries are read from HBM / L2
        if (layout == CST_LAYOUT_SYMBOL_MAJOR) return launch(range_encode_kernel<W, S, CST_LAYOUT_SYMBOL_MAJOR, false, G, true>, a.n_streams, kPerBlockLds, hs, a);
        if (vec) return launch(range_encode_kernel<W, S, CST_LAYOUT_STREAM_MAJOR, true, G, true>, a.n_streams, kPerBlockLds, hs, a);
        return launch(range_encode_kernel<W, S, CST_LAYOUT_STREAM_MAJOR, false, G, true>, a.n_streams, kPerBlockLds, hs, a);
    }
    if (layout == CST_LAYOUT_SYMBOL_MAJOR) return launch(range_encode_kernel<W, S, CST_LAYOUT_SYMBOL_MAJOR, false, G>, a.n_streams, lds, hs, a);
    if (vec) return launch(range_encode_kernel<W, S, CST_LAYOUT_STREAM_MAJOR, true, G>, a.n_streams, lds, hs, a);
    return launch(range_encode_kernel<W, S, CST_LAYOUT_STREAM_MAJOR, false, G>, a.n_streams, lds, hs, a);
}

template <int W, int S>
static cst_status range_encode_ws(const RangeEncodeArgs& a, cst_layout layout, hipStream_t hs) {
    switch (groups_per_point(W, a.precision)) {
        case 8: return range_encode_g<W, S, 8>(a, layout, hs);
        case 4: return range_encode_g<W, S, 4>(a, layout, hs);
        default: return range_encode_g<W, S, 2>(a, layout, hs);
    }
}

template <int W, int S, int MODE, bool LDS, int G>
static cst_status range_decode_g(const RangeDecodeArgs& a, cst_layout layout, size_t table_lds, hipStream_t hs) {
    const size_t lds = ((table_lds + 15) & ~(size_t)15) + kPerBlockLds;
    if (layout == CST_LAYOUT_SYMBOL_MAJOR) return launch(range_decode_kernel<W, S, CST_LAYOUT_SYMBOL_MAJOR, false, MODE, LDS, G>, a.n_streams, lds, hs, a);
    const bool vec = (a.n_per_stream % 4 == 0) && aligned16(a.symbols);
    if (vec) return launch(range_decode_kernel<W, S, CST_LAYOUT_STREAM_MAJOR, true, MODE, LDS, G>, a.n_streams, lds, hs, a);
    return launch(range_decode_kernel<W, S, CST_LAYOUT_STREAM_MAJOR, false, MODE, LDS, G>, a.n_streams, lds, hs, a);
}

template <int W, int S, int MODE, bool LDS>
static cst_status range_decode_m(const RangeDecodeArgs& a, cst_layout layout, size_t table_lds, hipStream_t hs) {
    switch (groups_per_point(W, a.precision)) {
        case 8: return range_decode_g<W, S, MODE, LDS, 8>(a, layout, table_lds, hs);
        case 4: return range_decode_g<W, S, MODE, LDS, 4>(a, layout, table_lds, hs);
        default: return range_decode_g<W, S, MODE, LDS, 2>(a, layout, table_lds, hs);
    }
}

template <int W, int S>
static cst_status range_decode_ws(const RangeDecodeArgs& a, cst_layout layout, hipStream_t hs) {
    const int P = a.precision;
    const size_t lds_budget = kMaxLds - kPerBlockLds - 1024;
    if (a.dec_cp && ((size_t)6 << P) <= lds_budget) return range_decode_m<W, S, kDecLutCP, true>(a, layout, ((size_t)6 << P), hs);
    const size_t bucket_lds = ((((size_t)a.n_symbols + 1) * 4 + 15) & ~(size_t)15) +
                              (bucket16_usable(a.n_symbols, P) ? ((size_t)16 << a.bucket_bits) : ((((size_t)2 << a.bucket_bits) + 15) & ~(size_t)15));
    if (bucket_lds <= lds_budget) return range_decode_m<W, S, kDecBucket, true>(a, layout, bucket_lds, hs);
    if (a.dec_cp) return range_decode_m<W, S, kDecLutCP, false>(a, layout, 0, hs);
    return range_decode_m<W, S, kDecBucket, false>(a, layout, 0, hs);
}


// ------------------------------------------------------------------------------------------------------------------
// Jump points for any preset (the hand-scheduled (32,64) kernels are in cst_range_fast.hip): one lane per stream, symbols read
// straight from HBM, RangeEncoder::pos() = (bulk.len() + num_inverted, (lower, range)) noted in front of every chunk
// (queue.rs:182-196); and the decoder's way back to the ordinary batched decode: every (stream, chunk) pair becomes a virtual
// stream that continues at its jump point (RangeDecoder::seek, queue.rs:911-926 = CST_FLAG_RAW_STATE with a position).
// ------------------------------------------------------------------------------------------------------------------
template <int W, int S>
__global__ __launch_bounds__(kBlock) void range_encode_ckpt_generic_kernel(const RangeEncodeArgs a, const RangeCkptOut ck, cst_layout layout) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem) + (threadIdx.x >> 6) * kRingWords;
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = s < a.n_streams;
    const size_t N = a.n_per_stream;
    const int P = a.precision;
    const uint32_t nsym = (uint32_t)a.n_symbols;
    const size_t stride_t = layout == CST_LAYOUT_SYMBOL_MAJOR ? a.n_streams : 1;
    const int32_t* my = a.symbols + (active ? (layout == CST_LAYOUT_SYMBOL_MAJOR ? s : s * N) : 0);
    RangeEncLane<W, S> L;
    L.init(a.words + (active ? s : 0) * a.stride_words,
           active ? (uint32_t)(a.stride_words > 0xffffffffull ? 0xffffffffull : a.stride_words) : 0u, ring, lane);
    int countdown = 4 * groups_per_point(W, P);
    for (size_t t = 0; t < N; ++t) {
        if (active && t % ck.interval == 0) {
            const size_t k = s * ck.n_chunks + t / ck.interval;
            ck.pos[k] = L.out.wr + L.inv_n; ck.lower[k] = (uint64_t)L.lower; ck.range[k] = (uint64_t)L.range;
        }
        const int32_t v = active ? my[t * stride_t] : a.min_symbol;
        const EncEntry e = a.enc[enc_index(v, a.min_symbol, nsym, L.bad)];
        L.step(e.c, e.p, P);
        if (--countdown == 0) { countdown = 4 * groups_per_point(W, P); L.out.flush_chunks(); }
    }
    uint32_t n_words = 0;
    const int32_t status = L.finish(nsym, n_words);
    if (!active) return;
    a.status[s] = status;
    a.n_words[s] = (status == CST_STREAM_OK) ? n_words : 0u;
}

// virtual stream v = (stream v / k, chunk v % k): where its words lie, and the decoder state RangeDecoder::seek leaves
template <int W, int S>
__global__ void range_ckpt_virtual_kernel(const uint32_t* __restrict__ words, const uint64_t* __restrict__ offsets, size_t stride_words,
                                          uint64_t capacity, const uint32_t* __restrict__ n_words, const uint32_t* __restrict__ ckpt_pos,
                                          const uint64_t* __restrict__ ckpt_lower, const uint64_t* __restrict__ ckpt_range, size_t n_streams,
                                          size_t n_chunks, uint64_t* __restrict__ v_offsets, uint32_t* __restrict__ v_n, cst_range_state* __restrict__ v_state) {
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_streams * n_chunks) return;
    const size_t s = v / n_chunks;
    const WordSlice ws = word_slice(offsets, stride_words, n_words, s, capacity);
    const uint32_t pos0 = ckpt_pos[v] < ws.n ? ckpt_pos[v] : ws.n;
    uint64_t pt = 0;
    uint32_t pos = pos0;
    int num_read = 0;
    while (pos < ws.n) {                                          // read_point, queue.rs:847-868
        pt = ((pt << (W % 64)) | (uint64_t)words[ws.off + pos++]) & (S == 64 ? ~0ull : ((1ull << (S % 64)) - 1ull));
        if (++num_read == S / W) break;
    }
    if (num_read < S / W && num_read != 0) pt = (pt << (S - num_read * W)) & (S == 64 ? ~0ull : ((1ull << (S % 64)) - 1ull));
    cst_range_state r{};
    r.lower = ckpt_lower[v]; r.range = ckpt_range[v]; r.point = pt; r.position = pos;
    v_state[v] = r;
    v_offsets[v] = offsets ? offsets[s] : (uint64_t)s * stride_words;   // (the decoder checks the slice again: a bad one stays bad)
    v_n[v] = n_words[s];
}

// a jump point beyond its stream's words is caller data gone wrong: that chunk reports INVALID_DATA
__global__ void range_ckpt_flag_kernel(const uint32_t* __restrict__ n_words, const uint32_t* __restrict__ ckpt_pos, size_t n_streams, size_t n_chunks,
                                       int32_t* __restrict__ status) {
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_streams * n_chunks) return;
    if (ckpt_pos[v] > n_words[v / n_chunks]) status[v] = CST_STREAM_INVALID_DATA;
}

} // namespace cst

using namespace cst;

extern "C" {

cst_status cst_range_encode_batch(const cst_model* model, cst_coder_config cfg, const int32_t* d_symbols, size_t n_streams,
                                  size_t n_per_stream, cst_layout layout, uint32_t* d_words, size_t stride_words,
                                  uint32_t* d_n_words, cst_range_state* d_rstate, int32_t* d_status, uint32_t flags, void* stream) {
    if (!model || !d_words || !d_n_words || !d_status) return CST_ERR_INVALID_ARGUMENT;
    if ((flags & CST_FLAG_RAW_STATE) && !d_rstate) return CST_ERR_INVALID_ARGUMENT;
    if (n_per_stream > 0 && !d_symbols) return CST_ERR_INVALID_ARGUMENT;
    if (!config_supported(cfg) || cfg.precision != model->precision) return CST_ERR_INVALID_ARGUMENT;
    if (layout != CST_LAYOUT_STREAM_MAJOR && layout != CST_LAYOUT_SYMBOL_MAJOR) return CST_ERR_INVALID_ARGUMENT;
    if (model->per_stream) return CST_ERR_INVALID_ARGUMENT;
    if (n_streams == 0) return CST_OK;
    RangeEncodeArgs a{};
    a.symbols = d_symbols; a.n_streams = n_streams; a.n_per_stream = n_per_stream; a.enc = model->d_enc;
    a.n_symbols = model->n_symbols; a.min_symbol = model->min_symbol; a.precision = model->precision;
    a.words = d_words; a.stride_words = stride_words; a.n_words = d_n_words; a.status = d_status;
    a.rstate = d_rstate; a.flags = flags;
    hipStream_t hs = (hipStream_t)stream;
    if (cfg.word_bits == 32 && range_encode_fast_usable(a, layout)) return note_kernel("range_encode_fast_kernel", range_encode_fast(a, layout, hs));
    if (cfg.word_bits == 32) return note_kernel("range_encode_kernel", range_encode_ws<32, 64>(a, layout, hs));
    return note_kernel("range_encode_kernel", range_encode_ws<16, 32>(a, layout, hs));
}

cst_status cst_range_decode_batch(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words,
                                  const uint64_t* d_offsets, size_t stride_words, size_t words_capacity, const uint32_t* d_n_words,
                                  int32_t* d_symbols, size_t n_streams, size_t n_per_stream, cst_layout layout,
                                  cst_range_state* d_rstate, int32_t* d_status, uint32_t flags, void* stream) {
    if (!model || !d_n_words || !d_status) return CST_ERR_INVALID_ARGUMENT;
    if ((flags & CST_FLAG_RAW_STATE) && !d_rstate) return CST_ERR_INVALID_ARGUMENT;
    if (n_per_stream > 0 && !d_symbols) return CST_ERR_INVALID_ARGUMENT;
    if (!config_supported(cfg) || cfg.precision != model->precision) return CST_ERR_INVALID_ARGUMENT;
    if (layout != CST_LAYOUT_STREAM_MAJOR && layout != CST_LAYOUT_SYMBOL_MAJOR) return CST_ERR_INVALID_ARGUMENT;
    if (model->per_stream) return CST_ERR_INVALID_ARGUMENT;
    if (n_streams == 0) return CST_OK;
    RangeDecodeArgs a{};
    a.words = d_words; a.offsets = d_offsets; a.stride_words = stride_words; a.n_words = d_n_words; a.symbols = d_symbols;
    a.n_streams = n_streams; a.n_per_stream = n_per_stream; a.dec_cp = model->d_dec_cp; a.dec_idx = model->d_dec_idx;
    a.cdf = model->d_cdf; a.bucket = model->d_bucket; a.bucket_bits = model->bucket_bits; a.n_symbols = model->n_symbols;
    a.min_symbol = model->min_symbol; a.precision = model->precision; a.status = d_status;
    a.rstate = d_rstate; a.flags = flags; a.words_capacity = words_capacity;
    hipStream_t hs = (hipStream_t)stream;
    if (cfg.word_bits == 32 && range_decode_fast_usable(a, layout)) return note_kernel("range_decode_fast_kernel", range_decode_fast(a, layout, hs));
    if (cfg.word_bits == 32) return note_kernel("range_decode_kernel", range_decode_ws<32, 64>(a, layout, hs));
    return note_kernel("range_decode_kernel", range_decode_ws<16, 32>(a, layout, hs));
}

cst_status cst_range_encode_batch_ckpt(const cst_model* model, cst_coder_config cfg, const int32_t* d_symbols, size_t n_streams,
                                       size_t n_per_stream, cst_layout layout, uint32_t* d_words, size_t stride_words, uint32_t* d_n_words,
                                       size_t ckpt_interval, uint32_t* d_ckpt_pos, uint64_t* d_ckpt_lower, uint64_t* d_ckpt_range,
                                       int32_t* d_status, void* stream) {
    if (!model || !d_words || !d_n_words || !d_status || !d_ckpt_pos || !d_ckpt_lower || !d_ckpt_range || ckpt_interval == 0) return CST_ERR_INVALID_ARGUMENT;
    if (n_per_stream > 0 && !d_symbols) return CST_ERR_INVALID_ARGUMENT;
    if (!config_supported(cfg) || cfg.precision != model->precision || model->per_stream) return CST_ERR_INVALID_ARGUMENT;
    if (layout != CST_LAYOUT_STREAM_MAJOR && layout != CST_LAYOUT_SYMBOL_MAJOR) return CST_ERR_INVALID_ARGUMENT;
    if (n_streams == 0) return CST_OK;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != model->device) return CST_ERR_INVALID_ARGUMENT;
    RangeEncodeArgs a{};
    a.symbols = d_symbols; a.n_streams = n_streams; a.n_per_stream = n_per_stream; a.enc = model->d_enc;
    a.n_symbols = model->n_symbols; a.min_symbol = model->min_symbol; a.precision = model->precision;
    a.words = d_words; a.stride_words = stride_words; a.n_words = d_n_words; a.status = d_status;
    RangeCkptOut ck{d_ckpt_pos, d_ckpt_lower, d_ckpt_range, ckpt_interval, (n_per_stream + ckpt_interval - 1) / ckpt_interval};
    hipStream_t hs = (hipStream_t)stream;
    if (cfg.word_bits == 32 && range_encode_ckpt_fast_usable(a, layout)) return note_kernel("range_encode_ckpt_kernel", range_encode_ckpt_fast(a, ck, hs));
    const size_t blocks = (n_streams + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    const size_t lds = (size_t)(kBlock / kWave) * kRingWords * 4;
    if (cfg.word_bits == 32) hipLaunchKernelGGL((range_encode_ckpt_generic_kernel<32, 64>), dim3((unsigned)blocks), dim3(kBlock), lds, hs, a, ck, layout);
    else hipLaunchKernelGGL((range_encode_ckpt_generic_kernel<16, 32>), dim3((unsigned)blocks), dim3(kBlock), lds, hs, a, ck, layout);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

size_t cst_range_ckpt_scratch_bytes(size_t n_streams, size_t n_per_stream, size_t ckpt_interval) {
    if (ckpt_interval == 0) return 0;
    return (sizeof(cst_range_state) + 16) * n_streams * ((n_per_stream + ckpt_interval - 1) / ckpt_interval) + 16;
}

cst_status cst_range_decode_batch_ckpt(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_offsets,
                                       size_t stride_words, size_t words_capacity, const uint32_t* d_n_words, size_t ckpt_interval,
                                       const uint32_t* d_ckpt_pos, const uint64_t* d_ckpt_lower, const uint64_t* d_ckpt_range,
                                       int32_t* d_symbols, size_t n_streams, size_t n_per_stream, void* d_scratch, int32_t* d_status,
                                       void* stream) {
    if (!model || !d_n_words || !d_ckpt_pos || !d_ckpt_lower || !d_ckpt_range || !d_scratch || !d_status || ckpt_interval == 0) return CST_ERR_INVALID_ARGUMENT;
    if (n_per_stream % ckpt_interval != 0) return CST_ERR_INVALID_ARGUMENT;      // whole chunks only: rows of the virtual matrix
    if (!config_supported(cfg) || cfg.precision != model->precision || model->per_stream) return CST_ERR_INVALID_ARGUMENT;
    if (n_streams == 0 || n_per_stream == 0) return CST_OK;
    if (!d_symbols || !d_words) return CST_ERR_INVALID_ARGUMENT;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != model->device) return CST_ERR_INVALID_ARGUMENT;
    const size_t n_chunks = n_per_stream / ckpt_interval, n_virtual = n_streams * n_chunks;
    if (n_virtual > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    hipStream_t hs = (hipStream_t)stream;
    const size_t capacity = words_capacity ? words_capacity : (d_offsets ? 0 : n_streams * stride_words);
    RangeDecodeArgs a{};
    a.words = d_words; a.offsets = d_offsets; a.stride_words = stride_words; a.n_words = d_n_words; a.symbols = d_symbols;
    a.n_streams = n_streams; a.n_per_stream = n_per_stream; a.dec_cp = model->d_dec_cp; a.dec_idx = model->d_dec_idx;
    a.cdf = model->d_cdf; a.bucket = model->d_bucket; a.bucket_bits = model->bucket_bits; a.n_symbols = model->n_symbols;
    a.min_symbol = model->min_symbol; a.precision = model->precision; a.status = d_status; a.words_capacity = capacity;
    a.ckpt_pos = d_ckpt_pos; a.ckpt_lower = d_ckpt_lower; a.ckpt_range = d_ckpt_range; a.interval = ckpt_interval; a.n_chunks = n_chunks;
    // k lanes per stream on the hand-scheduled statements, two waves per SIMD (cst_range_fast.hip)
    if (cfg.word_bits == 32 && range_decode_sub_usable(a)) return note_kernel("range_decode_sub_kernel", range_decode_sub(a, hs));
    // any other preset / alphabet: the ordinary batched decode of the virtual streams, continued at their jump points
    cst_range_state* v_state = reinterpret_cast<cst_range_state*>(d_scratch);
    uint64_t* v_offsets = reinterpret_cast<uint64_t*>(v_state + n_virtual);
    uint32_t* v_n = reinterpret_cast<uint32_t*>(v_offsets + n_virtual);
    const dim3 grid((unsigned)((n_virtual + 255) / 256));
    if (cfg.word_bits == 32)
        hipLaunchKernelGGL((range_ckpt_virtual_kernel<32, 64>), grid, dim3(256), 0, hs, d_words, d_offsets, stride_words, capacity, d_n_words, d_ckpt_pos,
                           d_ckpt_lower, d_ckpt_range, n_streams, n_chunks, v_offsets, v_n, v_state);
    else
        hipLaunchKernelGGL((range_ckpt_virtual_kernel<16, 32>), grid, dim3(256), 0, hs, d_words, d_offsets, stride_words, capacity, d_n_words, d_ckpt_pos,
                           d_ckpt_lower, d_ckpt_range, n_streams, n_chunks, v_offsets, v_n, v_state);
    CST_HIP_TRY(hipGetLastError());
    const cst_status rc = cst_range_decode_batch(model, cfg, d_words, v_offsets, 0, capacity, v_n, d_symbols, n_virtual, ckpt_interval,
                                                 CST_LAYOUT_STREAM_MAJOR, v_state, d_status, CST_FLAG_RAW_STATE, stream);
    if (rc != CST_OK) return rc;
    hipLaunchKernelGGL(range_ckpt_flag_kernel, grid, dim3(256), 0, hs, d_n_words, d_ckpt_pos, n_streams, n_chunks, d_status);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

// ---- narrow symbol matrices (ABI 5): the reference's coders are generic over the symbol type (queue.rs:612, 968; quantize.rs:229-255) ----
//
// int8 matrices of stream-major rows of whole 32-symbol tiles are read by the hand-scheduled encoder itself and written by the sub-lane
// decoder itself (cst_range_fast.hip, round 6); every other shape converts next to the int32 call (cst_symbols_widen / _narrow), as
// cst_ans_*_batch_sym do.  Scratch layout: [jump-point scratch of the int32 calls][one jump point per stream for the plain calls]
// [the widened matrix].

static size_t range_sym_plain_table_bytes(size_t n_streams) { return ((20 * n_streams + 15) & ~(size_t)15) + 16; }

size_t cst_range_sym_scratch_bytes(size_t n_streams, size_t n_per_stream, size_t ckpt_interval, int32_t symbol_bytes) {
    const size_t ck = ckpt_interval ? cst_range_ckpt_scratch_bytes(n_streams, n_per_stream, ckpt_interval) : 0;
    return ((ck + 15) & ~(size_t)15) + range_sym_plain_table_bytes(n_streams) + cst_symbols_scratch_bytes(n_streams, n_per_stream, symbol_bytes) + 32;
}

namespace {
struct SymScratch { unsigned char* ck; uint32_t* pos; uint64_t* lower; uint64_t* range; void* conv; };
SymScratch split_scratch(void* d_scratch, size_t n_streams, size_t n_per_stream, size_t ckpt_interval) {
    SymScratch x{};
    unsigned char* b = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(d_scratch) + 15) & ~(uintptr_t)15);
    x.ck = b;
    b += ((ckpt_interval ? cst_range_ckpt_scratch_bytes(n_streams, n_per_stream, ckpt_interval) : 0) + 15) & ~(size_t)15;
    x.lower = reinterpret_cast<uint64_t*>(b);
    x.range = x.lower + n_streams;
    x.pos = reinterpret_cast<uint32_t*>(x.range + n_streams);
    x.conv = b + range_sym_plain_table_bytes(n_streams);
    return x;
}
bool narrow_model_ok(const cst_model* model, cst_coder_config cfg) {
    return model && !model->per_stream && !model->d_symbol_of_index && config_supported(cfg) && cfg.precision == model->precision &&
           cfg.word_bits == 32 && cfg.state_bits == 64;
}
} // namespace

cst_status cst_range_encode_batch_ckpt_sym(const cst_model* model, cst_coder_config cfg, const void* d_symbols, int32_t symbol_bytes, size_t n_streams,
                                           size_t n_per_stream, cst_layout layout, uint32_t* d_words, size_t stride_words, uint32_t* d_n_words,
                                           size_t ckpt_interval, uint32_t* d_ckpt_pos, uint64_t* d_ckpt_lower, uint64_t* d_ckpt_range,
                                           int32_t* d_status, void* d_scratch, void* stream) {
    if (symbol_bytes == 4)
        return cst_range_encode_batch_ckpt(model, cfg, reinterpret_cast<const int32_t*>(d_symbols), n_streams, n_per_stream, layout, d_words, stride_words,
                                           d_n_words, ckpt_interval, d_ckpt_pos, d_ckpt_lower, d_ckpt_range, d_status, stream);
    if (!model || (symbol_bytes != 1 && symbol_bytes != 2) || !d_words || !d_n_words || !d_status || !d_ckpt_pos || !d_ckpt_lower || !d_ckpt_range ||
        ckpt_interval == 0)
        return CST_ERR_INVALID_ARGUMENT;
    if (n_per_stream > 0 && !d_symbols) return CST_ERR_INVALID_ARGUMENT;
    if (n_streams == 0) return CST_OK;
    hipStream_t hs = (hipStream_t)stream;
    if (symbol_bytes == 1 && narrow_model_ok(model, cfg)) {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev != model->device) return CST_ERR_INVALID_ARGUMENT;
        RangeEncodeArgs a{};
        a.symbols = reinterpret_cast<const int32_t*>(d_symbols); a.n_streams = n_streams; a.n_per_stream = n_per_stream; a.enc = model->d_enc;
        a.n_symbols = model->n_symbols; a.min_symbol = model->min_symbol; a.precision = model->precision;
        a.words = d_words; a.stride_words = stride_words; a.n_words = d_n_words; a.status = d_status;
        if (range_encode_n8_usable(a, layout, ckpt_interval)) {
            RangeCkptOut ck{d_ckpt_pos, d_ckpt_lower, d_ckpt_range, ckpt_interval, n_per_stream / ckpt_interval};
            return note_kernel("range_encode_ckpt_n8_kernel", range_encode_ckpt_n8(a, ck, hs));
        }
    }
    if (!d_scratch && n_streams * n_per_stream > 0) return CST_ERR_INVALID_ARGUMENT;
    const SymScratch x = split_scratch(d_scratch, n_streams, n_per_stream, ckpt_interval);
    int32_t* wide = reinterpret_cast<int32_t*>(x.conv);
    const cst_status rc = cst_symbols_widen(d_symbols, symbol_bytes, n_streams * n_per_stream, wide, stream);
    if (rc != CST_OK) return rc;
    return cst_range_encode_batch_ckpt(model, cfg, wide, n_streams, n_per_stream, layout, d_words, stride_words, d_n_words, ckpt_interval, d_ckpt_pos,
                                       d_ckpt_lower, d_ckpt_range, d_status, stream);
}

cst_status cst_range_encode_batch_sym(const cst_model* model, cst_coder_config cfg, const void* d_symbols, int32_t symbol_bytes, size_t n_streams,
                                      size_t n_per_stream, cst_layout layout, uint32_t* d_words, size_t stride_words, uint32_t* d_n_words,
                                      cst_range_state* d_rstate, int32_t* d_status, uint32_t flags, void* d_scratch, void* stream) {
    if (symbol_bytes == 4)
        return cst_range_encode_batch(model, cfg, reinterpret_cast<const int32_t*>(d_symbols), n_streams, n_per_stream, layout, d_words, stride_words,
                                      d_n_words, d_rstate, d_status, flags, stream);
    if (!model || (symbol_bytes != 1 && symbol_bytes != 2) || !d_words || !d_n_words || !d_status) return CST_ERR_INVALID_ARGUMENT;
    if (n_per_stream > 0 && !d_symbols) return CST_ERR_INVALID_ARGUMENT;
    if (n_streams == 0) return CST_OK;
    if (!d_scratch) return CST_ERR_INVALID_ARGUMENT;
    const SymScratch x = split_scratch(d_scratch, n_streams, n_per_stream, 0);
    if (symbol_bytes == 1 && flags == CST_FLAG_NONE && narrow_model_ok(model, cfg) && n_per_stream >= (size_t)cst::kTileSyms) {
        // the jump-point-noting encoder with ONE chunk: its only jump point is the start of the stream (noted into the scratch, unused)
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev != model->device) return CST_ERR_INVALID_ARGUMENT;
        RangeEncodeArgs a{};
        a.symbols = reinterpret_cast<const int32_t*>(d_symbols); a.n_streams = n_streams; a.n_per_stream = n_per_stream; a.enc = model->d_enc;
        a.n_symbols = model->n_symbols; a.min_symbol = model->min_symbol; a.precision = model->precision;
        a.words = d_words; a.stride_words = stride_words; a.n_words = d_n_words; a.status = d_status;
        if (range_encode_n8_usable(a, layout, n_per_stream)) {
            RangeCkptOut ck{x.pos, x.lower, x.range, n_per_stream, 1};
            return note_kernel("range_encode_n8_kernel", range_encode_ckpt_n8(a, ck, (hipStream_t)stream));
        }
    }
    int32_t* wide = reinterpret_cast<int32_t*>(x.conv);
    const cst_status rc = cst_symbols_widen(d_symbols, symbol_bytes, n_streams * n_per_stream, wide, stream);
    if (rc != CST_OK) return rc;
    return cst_range_encode_batch(model, cfg, wide, n_streams, n_per_stream, layout, d_words, stride_words, d_n_words, d_rstate, d_status, flags, stream);
}

cst_status cst_range_decode_batch_ckpt_sym(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_offsets,
                                           size_t stride_words, size_t words_capacity, const uint32_t* d_n_words, size_t ckpt_interval,
                                           const uint32_t* d_ckpt_pos, const uint64_t* d_ckpt_lower, const uint64_t* d_ckpt_range, void* d_symbols,
                                           int32_t symbol_bytes, size_t n_streams, size_t n_per_stream, void* d_scratch, int32_t* d_status, void* stream) {
    if (symbol_bytes == 4)
        return cst_range_decode_batch_ckpt(model, cfg, d_words, d_offsets, stride_words, words_capacity, d_n_words, ckpt_interval, d_ckpt_pos, d_ckpt_lower,
                                           d_ckpt_range, reinterpret_cast<int32_t*>(d_symbols), n_streams, n_per_stream, d_scratch, d_status, stream);
    if (!model || (symbol_bytes != 1 && symbol_bytes != 2) || !d_n_words || !d_ckpt_pos || !d_ckpt_lower || !d_ckpt_range || !d_scratch || !d_status ||
        ckpt_interval == 0)
        return CST_ERR_INVALID_ARGUMENT;
    if (n_per_stream % ckpt_interval != 0) return CST_ERR_INVALID_ARGUMENT;
    if (n_streams == 0 || n_per_stream == 0) return CST_OK;
    if (!d_symbols || !d_words) return CST_ERR_INVALID_ARGUMENT;
    if (symbol_bytes == 1 && narrow_model_ok(model, cfg)) {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev != model->device) return CST_ERR_INVALID_ARGUMENT;
        const size_t n_chunks = n_per_stream / ckpt_interval;
        RangeDecodeArgs a{};
        a.words = d_words; a.offsets = d_offsets; a.stride_words = stride_words; a.n_words = d_n_words; a.symbols = reinterpret_cast<int32_t*>(d_symbols);
        a.n_streams = n_streams; a.n_per_stream = n_per_stream; a.dec_cp = model->d_dec_cp; a.dec_idx = model->d_dec_idx;
        a.cdf = model->d_cdf; a.bucket = model->d_bucket; a.bucket_bits = model->bucket_bits; a.n_symbols = model->n_symbols;
        a.min_symbol = model->min_symbol; a.precision = model->precision; a.status = d_status;
        a.words_capacity = words_capacity ? words_capacity : (d_offsets ? 0 : n_streams * stride_words);
        a.ckpt_pos = d_ckpt_pos; a.ckpt_lower = d_ckpt_lower; a.ckpt_range = d_ckpt_range; a.interval = ckpt_interval; a.n_chunks = n_chunks;
        if (n_streams * n_chunks <= 0x7fffffffull && range_decode_sub_n8_usable(a))
            return note_kernel("range_decode_sub_n8_kernel", range_decode_sub_n8(a, (hipStream_t)stream));
    }
    const SymScratch x = split_scratch(d_scratch, n_streams, n_per_stream, ckpt_interval);
    int32_t* wide = reinterpret_cast<int32_t*>(x.conv);
    const cst_status rc = cst_range_decode_batch_ckpt(model, cfg, d_words, d_offsets, stride_words, words_capacity, d_n_words, ckpt_interval, d_ckpt_pos,
                                                      d_ckpt_lower, d_ckpt_range, wide, n_streams, n_per_stream, x.ck, d_status, stream);
    if (rc != CST_OK) return rc;
    return cst_symbols_narrow(wide, n_streams * n_per_stream, d_symbols, symbol_bytes, stream);
}

cst_status cst_range_decode_batch_sym(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_offsets,
                                      size_t stride_words, size_t words_capacity, const uint32_t* d_n_words, void* d_symbols, int32_t symbol_bytes,
                                      size_t n_streams, size_t n_per_stream, cst_layout layout, cst_range_state* d_rstate, int32_t* d_status,
                                      uint32_t flags, void* d_scratch, void* stream) {
    if (symbol_bytes == 4)
        return cst_range_decode_batch(model, cfg, d_words, d_offsets, stride_words, words_capacity, d_n_words, reinterpret_cast<int32_t*>(d_symbols),
                                      n_streams, n_per_stream, layout, d_rstate, d_status, flags, stream);
    if (!model || (symbol_bytes != 1 && symbol_bytes != 2) || !d_n_words || !d_status) return CST_ERR_INVALID_ARGUMENT;
    if (n_per_stream > 0 && !d_symbols) return CST_ERR_INVALID_ARGUMENT;
    if (n_streams == 0 || n_per_stream == 0) return CST_OK;
    if (!d_scratch || !d_words) return CST_ERR_INVALID_ARGUMENT;
    const SymScratch x = split_scratch(d_scratch, n_streams, n_per_stream, 0);
    hipStream_t hs = (hipStream_t)stream;
    if (symbol_bytes == 1 && flags == CST_FLAG_NONE && layout == CST_LAYOUT_STREAM_MAJOR && narrow_model_ok(model, cfg)) {
        // the sub-lane decoder with ONE lane per stream: its jump point is the start of the stream, (pos, lower, range) = (0, 0, 2^64 - 1)
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev != model->device) return CST_ERR_INVALID_ARGUMENT;
        RangeDecodeArgs a{};
        a.words = d_words; a.offsets = d_offsets; a.stride_words = stride_words; a.n_words = d_n_words; a.symbols = reinterpret_cast<int32_t*>(d_symbols);
        a.n_streams = n_streams; a.n_per_stream = n_per_stream; a.dec_cp = model->d_dec_cp; a.dec_idx = model->d_dec_idx;
        a.cdf = model->d_cdf; a.bucket = model->d_bucket; a.bucket_bits = model->bucket_bits; a.n_symbols = model->n_symbols;
        a.min_symbol = model->min_symbol; a.precision = model->precision; a.status = d_status;
        a.words_capacity = words_capacity ? words_capacity : (d_offsets ? 0 : n_streams * stride_words);
        a.ckpt_pos = x.pos; a.ckpt_lower = x.lower; a.ckpt_range = x.range; a.interval = n_per_stream; a.n_chunks = 1;
        if (n_streams <= 0x7fffffffull && range_decode_sub_n8_usable(a)) {
            CST_HIP_TRY(hipMemsetAsync(x.lower, 0, 8 * n_streams, hs));
            CST_HIP_TRY(hipMemsetAsync(x.range, 0xff, 8 * n_streams, hs));
            CST_HIP_TRY(hipMemsetAsync(x.pos, 0, 4 * n_streams, hs));
            return note_kernel("range_decode_n8_kernel", range_decode_sub_n8(a, hs));
        }
    }
    int32_t* wide = reinterpret_cast<int32_t*>(x.conv);
    const cst_status rc = cst_range_decode_batch(model, cfg, d_words, d_offsets, stride_words, words_capacity, d_n_words, wide, n_streams, n_per_stream, layout,
                                                 d_rstate, d_status, flags, stream);
    if (rc != CST_OK) return rc;
    return cst_symbols_narrow(wide, n_streams * n_per_stream, d_symbols, symbol_bytes, stream);
}

} // extern "C"
