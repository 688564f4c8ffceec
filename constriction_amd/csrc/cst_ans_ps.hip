// cst_ans_ps.hip -- batched ANS with ONE MODEL PER STREAM (BASELINE config C3): every lane codes its stream with
// its own quantized-Gaussian table (learned-image-compression latents: one (mean, std) per channel / tile).
//
// Each lane's cumulative table (n+1 entries of 16 bits, row length L = 2^k) is held in LDS for the whole
// kernel, TRANSPOSED per wave: entry e of lane l lives at [e][l] (like the word rings).  The bank of an access then
// depends on the lane only, whatever index the lane probes: two lanes per bank instead of the ~6-way conflicts random
// per-lane indices cause in a row-per-lane layout (PMC: 200 of 1100 cycles per decoded symbol were bank conflicts).  Encoding needs (c, p) =
// (row[i], row[i+1]-row[i]) and the reciprocal floor(2^64/p), which comes from a small table indexed by p
// (shared by all streams, L1/L2 resident).  Decoding finds the largest i with row[i] <= q by a 4-ary search
// (3 probes per round, ceil(log4 n) rounds, branch free).
// The encoder reads stream-major symbols through wave-private LDS tiles like the shared-table kernels (128-byte row
// segments per access); the decoder writes 16 B per lane per 4 steps (see there); compressed words go through the
// same per-lane LDS rings.
#include "cst_ans_kernels.hpp"

namespace cst {

struct PsArgs {
    const int32_t* symbols_in;
    int32_t* symbols_out;
    size_t n_streams, n_per_stream;
    int32_t layout, precision, n_symbols, min_symbol;
    const uint16_t* cdf16;     // [n_streams][L]
    int32_t L;                 // row length (power of two)
    const uint64_t* recip;     // [2^P]
    uint32_t* words_out;
    const uint32_t* words_in;
    const uint64_t* offsets;
    size_t stride_words;
    uint32_t* n_words_out_enc; // encode: words written
    const uint32_t* n_words_in;
    uint32_t* n_words_left;    // decode raw: words left
    uint64_t* state;
    int32_t* status;
    uint32_t flags;
    uint64_t words_capacity;   // decode: uint32 slots behind `words_in` (0 = unknown): see word_slice
};

// this lane's column of its wave's transposed table: entry i at base[i * kWave]
struct LaneRow {
    const uint16_t* base;  // &table_of_wave[0][lane]
    __device__ __forceinline__ uint32_t at(uint32_t i) const { return base[i * kWave]; }
};

// `pad`: entries behind cdf[n] become 0xFFFF, a sentinel above every quantile when P <= 15 (the decoder's search
// then needs no bounds checks)
__device__ __forceinline__ void stage_rows(uint16_t* lds_rows, const PsArgs& a, size_t block_s0, bool pad = false) {
    const uint32_t L = (uint32_t)a.L, mask = L - 1;
    const uint32_t total = blockDim.x * L;
    const int shift = __builtin_ctz(L);
    for (uint32_t idx = threadIdx.x; idx < total; idx += blockDim.x) {
        const uint32_t j = idx >> shift, e = idx & mask;
        const size_t s = block_s0 + j;
        uint16_t v = s < a.n_streams ? a.cdf16[s * L + e] : (uint16_t)0;
        if (pad && e > (uint32_t)a.n_symbols) v = 0xFFFFu;
        lds_rows[((size_t)(j >> 6) * L + e) * kWave + (j & 63u)] = v;      // [wave][entry][lane]
    }
}

// Bucket index of the per-stream decoder: kPsBuckets + 1 entries per stream, start[b] = the symbol index whose bin
// holds quantile b * 2^(P - log2 kPsBuckets); kPsIdxStride entries per lane, stored [bucket][lane] like the tables.
constexpr int kPsBucketBits = 6, kPsBuckets = 1 << kPsBucketBits, kPsIdxStride = 66;

template <int W, int S>
__global__ void ans_encode_ps_kernel(const PsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const size_t block_s0 = (size_t)blockIdx.x * blockDim.x;
    uint16_t* rows = reinterpret_cast<uint16_t*>(smem);
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem + (size_t)blockDim.x * a.L * 2) + (threadIdx.x >> 6) * kRingWords;
    stage_rows(rows, a, block_s0);
    __syncthreads();

    const size_t s = block_s0 + threadIdx.x;
    const bool active = s < a.n_streams;
    const size_t N = a.n_per_stream;
    const int P = a.precision;
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;
    const uint32_t nsym = (uint32_t)a.n_symbols;
    const int G4 = 4 * groups_per_point(W, P);
    LaneRow R{rows + (size_t)(threadIdx.x >> 6) * a.L * kWave + lane};

    EncLane<W, S> L;
    L.init(a.words_out + (active ? s : 0) * a.stride_words,
           active ? (uint32_t)(a.stride_words > 0xffffffffull ? 0xffffffffull : a.stride_words) : 0u, ring, lane);
    if (raw && active) L.state = (typename StateT<S>::type)a.state[s];

    auto entry_of = [&](int32_t sym) {
        const uint32_t i = enc_index(sym, a.min_symbol, nsym, L.bad);
        const uint32_t c = R.at(i);
        const uint32_t p = (R.at(i + 1) - c) & 0xffffu;
        const uint64_t m = a.recip[p];
        return EncEntry{c, p, (uint32_t)m, (uint32_t)(m >> 32)};
    };

    const size_t stride_t = a.layout == CST_LAYOUT_SYMBOL_MAJOR ? a.n_streams : 1;
    const int32_t* my = a.symbols_in + (active ? (a.layout == CST_LAYOUT_SYMBOL_MAJOR ? s : s * N) : 0);
    const bool vec = a.layout == CST_LAYOUT_STREAM_MAJOR && (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.symbols_in) & 15) == 0);
    int countdown = G4;
    size_t t = N;
    if (vec) {
        // ragged top part [32 * n_full, N) directly, then full tiles through LDS (last tile first)
        const size_t n_full = N / kTileSyms;
        const size_t s0 = block_s0 + (size_t)(threadIdx.x >> 6) * kWave;
        int32_t* tile = reinterpret_cast<int32_t*>(smem + (size_t)blockDim.x * a.L * 2 + (size_t)(blockDim.x / kWave) * kRingWords * 4) +
                        (threadIdx.x >> 6) * (kWave * kTileStride);
        while (t > n_full * kTileSyms) {
            --t;
            const int32_t v = active ? my[t] : a.min_symbol;
            L.template step<false>(entry_of(v), P);
            if (--countdown <= 0) { countdown = G4; L.flush_chunks(); }
        }
        if (n_full > 0 && s0 < a.n_streams) {
            int32_t r[kTileSyms];
            tile_fetch<true>(a.symbols_in, a.n_streams, N, s0, (n_full - 1) * kTileSyms, lane, r);
            const int32_t* row = tile + lane * kTileStride;
            for (size_t tb = n_full; tb-- > 0;) {
                wave_lds_fence();
                tile_to_lds<true>(tile, lane, r);
                wave_lds_fence();
                if (tb > 0) tile_fetch<true>(a.symbols_in, a.n_streams, N, s0, (tb - 1) * kTileSyms, lane, r);
#pragma unroll
                for (int j = kTileSyms / 4 - 1; j >= 0; --j) {
                    int4 v = *reinterpret_cast<const int4*>(row + 4 * j);
                    if (!active) v = make_int4(a.min_symbol, a.min_symbol, a.min_symbol, a.min_symbol);
                    const EncEntry e3 = entry_of(v.w), e2 = entry_of(v.z), e1 = entry_of(v.y), e0 = entry_of(v.x);
                    L.template step<false>(e3, P); L.template step<false>(e2, P); L.template step<false>(e1, P); L.template step<false>(e0, P);
                    countdown -= 4;
                    if (countdown <= 0) { countdown = G4; L.flush_chunks(); }
                }
            }
        }
        t = 0;
    }
    while (t > 0) {
        --t;
        const int32_t v = active ? my[t * stride_t] : a.min_symbol;
        L.template step<false>(entry_of(v), P);
        if (--countdown <= 0) { countdown = G4; L.flush_chunks(); }
    }

    uint32_t n_words = 0;
    const int32_t status = L.finish(!raw, nsym, n_words);
    if (!active) return;
    if (raw) a.state[s] = (uint64_t)L.state;
    a.status[s] = status;
    a.n_words_out_enc[s] = (status == CST_STREAM_OK) ? n_words : 0u;
}

template <int W, int S>
__global__ void ans_decode_ps_kernel(const PsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using st_t = typename StateT<S>::type;
    const int lane = threadIdx.x & (kWave - 1);
    const size_t block_s0 = (size_t)blockIdx.x * blockDim.x;
    uint16_t* rows = reinterpret_cast<uint16_t*>(smem);
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem + (size_t)blockDim.x * a.L * 2) + (threadIdx.x >> 6) * kRingWords;
    stage_rows(rows, a, block_s0, a.precision >= kPsBucketBits && a.precision <= 15);
    __syncthreads();

    const size_t s = block_s0 + threadIdx.x;
    const bool active = s < a.n_streams;
    const size_t N = a.n_per_stream;
    const int P = a.precision;
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;
    const uint32_t n = (uint32_t)a.n_symbols;
    const int G4 = 4 * groups_per_point(W, P);
    const uint32_t qmask = (1u << P) - 1u;
    LaneRow R{rows + (size_t)(threadIdx.x >> 6) * a.L * kWave + lane};

    DecLane<W, S> L;
    const WordSlice ws = active ? word_slice(a.offsets, a.stride_words, a.n_words_in, s, a.words_capacity) : WordSlice{0, 0u, false};
    L.init(a.words_in + ws.off, ws.n, ring, lane);
    if (raw) L.state = active ? (st_t)a.state[s] : 0;
    else L.read_initial_state();
    L.in.prime();
    wave_lds_fence();

    // P in [6, 15]: bucket index + sentinel-padded rows (see stage_rows); otherwise the plain bounded 4-ary search
    const bool indexed = P >= kPsBucketBits && P <= 15;
    uint16_t* bidx = reinterpret_cast<uint16_t*>(smem + (size_t)blockDim.x * a.L * 2 + (size_t)(blockDim.x / kWave) * kRingWords * 4) +
                     (size_t)(threadIdx.x >> 6) * kPsIdxStride * kWave + lane;     // [wave][bucket][lane]
    if (indexed) {
        // this lane's own row -> its bucket index (one pass over the row)
        uint32_t i = 0;
        const uint32_t w = 1u << (P - kPsBucketBits);
        for (uint32_t b = 0; b < (uint32_t)kPsBuckets; ++b) {
            const uint32_t q0 = b * w;
            while (i + 1 < n && R.at(i + 1) <= q0) ++i;
            bidx[b * kWave] = (uint16_t)i;
        }
        bidx[kPsBuckets * kWave] = (uint16_t)(n - 1);
        wave_lds_fence();
    }

    auto decode_one = [&]() -> int32_t {
        const uint32_t q = (uint32_t)L.state & qmask;
        const uint32_t next_word = *L.in.slot(L.in.rd - 1u + L.in.shift);
        uint32_t lo;
        if (indexed) {
            // the answer lies in [start[b], start[b+1]]; everything behind start[b+1] (including the padding) is > q,
            // so the 4-ary refinement needs no bounds checks and a lane that is done is not disturbed by extra rounds
            const uint32_t b = q >> (P - kPsBucketBits);
            lo = bidx[b * kWave];
            uint32_t size = (uint32_t)bidx[(b + 1) * kWave] - lo + 1u;
            while (__any(size > 1)) {
                const uint32_t step = (size + 3) >> 2;
                // (probes clamped to index n, whose entry 2^P is itself above every quantile: the row may be exactly
                // n + 1 entries long, and an index behind it would wrap around to the row's small first entries)
                const uint32_t v1 = R.at(min(lo + step, n)), v2 = R.at(min(lo + 2 * step, n)), v3 = R.at(min(lo + 3 * step, n));
                lo += ((v1 <= q ? 1u : 0u) + (v2 <= q ? 1u : 0u) + (v3 <= q ? 1u : 0u)) * step;
                size = step;
            }
        } else {
            // largest i in [0, n) with row[i] <= q: 4-ary search, wave-uniform trip count
            lo = 0;
            uint32_t size = n;
            while (size > 1) {
                const uint32_t step = (size + 3) >> 2;
                const uint32_t p1 = lo + step, p2 = p1 + step, p3 = p2 + step;
                const uint32_t v1 = p1 < n ? R.at(p1) : 0x10000u, v2 = p2 < n ? R.at(p2) : 0x10000u, v3 = p3 < n ? R.at(p3) : 0x10000u;
                lo += ((v1 <= q ? 1u : 0u) + (v2 <= q ? 1u : 0u) + (v3 <= q ? 1u : 0u)) * step;
                size = step;
            }
        }
        const uint32_t c = R.at(lo);
        const uint32_t p = (R.at(lo + 1) - c) & 0xffffu;
        st_t st = (st_t)((st_t)(L.state >> P) * (st_t)p + (st_t)(q - c));          // stack.rs:1086-1088
        const bool refill = st < ((st_t)1 << (S - W)) && L.in.rd > 0;              // stack.rs:1089-1097
        L.state = refill ? (st_t)((st << (W % S)) | (st_t)next_word) : st;
        L.in.rd -= refill ? 1u : 0u;
        return a.min_symbol + (int32_t)lo;
    };

    const size_t stride_t = a.layout == CST_LAYOUT_SYMBOL_MAJOR ? a.n_streams : 1;
    int32_t* my = a.symbols_out + (active ? (a.layout == CST_LAYOUT_SYMBOL_MAJOR ? s : s * N) : 0);
    const bool vec = a.layout == CST_LAYOUT_STREAM_MAJOR && (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.symbols_out) & 15) == 0);
    // Symbols leave 16 bytes per lane: staging them through an LDS tile (as the encoder does) was measured 1.7x
    // SLOWER here -- the 9 KiB tile per wave costs a wave of occupancy (3 -> 2 per CU at L = 256) and this kernel is
    // bound by the latency of its table search, not by HBM.
    int countdown = G4;
    size_t t = 0;
    if (vec) {
        for (; t + 4 <= N; t += 4) {
            int4 v;
            v.x = decode_one(); v.y = decode_one(); v.z = decode_one(); v.w = decode_one();
            if (active) *reinterpret_cast<int4*>(my + t) = v;
            countdown -= 4;
            if (countdown <= 0) { countdown = G4; L.in.advance_window(); }
        }
    }
    for (; t < N; ++t) {
        const int32_t v = decode_one();
        if (active) my[t * stride_t] = v;
        if (--countdown <= 0) { countdown = G4; L.in.advance_window(); }
    }
    if (!active) return;
    a.status[s] = ws.bad ? (int32_t)CST_STREAM_INVALID_DATA : L.status;
    if (raw) {
        a.state[s] = (uint64_t)L.state;
        if (a.n_words_left) a.n_words_left[s] = L.in.rd;
    }
}

// threads per block such that rows + rings + symbol tiles fit in the 160 KiB of LDS of one CU
static int pick_block(int L, bool with_tiles, size_t& lds) {
    for (int threads = 256; threads >= 64; threads -= 64) {
        lds = (size_t)threads * L * 2 + (size_t)(threads / kWave) * (kRingWords + (with_tiles ? kWave * kTileStride : 0)) * sizeof(uint32_t) +
              (with_tiles ? 0 : (size_t)threads * kPsIdxStride * 2);   // encoder: symbol tiles; decoder: bucket index
        if (lds <= 160 * 1024) return threads;
    }
    return 0;
}

template <typename K>
static cst_status launch_ps(K kernel, const PsArgs& a, bool with_tiles, hipStream_t hs) {
    size_t lds = 0;
    const int threads = pick_block(a.L, with_tiles, lds);
    if (threads == 0) return CST_ERR_INVALID_ARGUMENT;   // support too large for LDS-resident per-stream tables
    const size_t blocks = (a.n_streams + threads - 1) / threads;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    if (lds > 64 * 1024)
        CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(threads), lds, hs, a);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

cst_status ans_encode_per_stream(const cst_model* model, cst_coder_config cfg, const int32_t* d_symbols, size_t n_streams,
                                 size_t n_per_stream, cst_layout layout, uint32_t* d_words, size_t stride_words,
                                 uint32_t* d_n_words, uint64_t* d_state, int32_t* d_status, uint32_t flags, hipStream_t hs) {
    if (model->n_tables != n_streams || !model->d_cdf16 || !model->d_recip) return CST_ERR_INVALID_ARGUMENT;
    PsArgs a{};
    a.symbols_in = d_symbols; a.n_streams = n_streams; a.n_per_stream = n_per_stream; a.layout = layout;
    a.precision = model->precision; a.n_symbols = model->n_symbols; a.min_symbol = model->min_symbol;
    a.cdf16 = model->d_cdf16; a.L = model->cdf16_stride; a.recip = model->d_recip; a.words_out = d_words;
    a.stride_words = stride_words; a.n_words_out_enc = d_n_words; a.state = d_state; a.status = d_status; a.flags = flags;
    if (cfg.word_bits == 32) return launch_ps(ans_encode_ps_kernel<32, 64>, a, true, hs);
    return launch_ps(ans_encode_ps_kernel<16, 32>, a, true, hs);
}

cst_status ans_decode_per_stream(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_offsets,
                                 size_t stride_words, size_t words_capacity, const uint32_t* d_n_words, int32_t* d_symbols, size_t n_streams,
                                 size_t n_per_stream, cst_layout layout, uint64_t* d_state, uint32_t* d_n_words_out,
                                 int32_t* d_status, uint32_t flags, hipStream_t hs) {
    if (model->n_tables != n_streams || !model->d_cdf16) return CST_ERR_INVALID_ARGUMENT;
    PsArgs a{};
    a.words_capacity = words_capacity;
    a.symbols_out = d_symbols; a.n_streams = n_streams; a.n_per_stream = n_per_stream; a.layout = layout;
    a.precision = model->precision; a.n_symbols = model->n_symbols; a.min_symbol = model->min_symbol;
    a.cdf16 = model->d_cdf16; a.L = model->cdf16_stride; a.words_in = d_words; a.offsets = d_offsets;
    a.stride_words = stride_words; a.n_words_in = d_n_words; a.n_words_left = d_n_words_out; a.state = d_state;
    a.status = d_status; a.flags = flags;
    if (cfg.word_bits == 32) return launch_ps(ans_decode_ps_kernel<32, 64>, a, false, hs);
    return launch_ps(ans_decode_ps_kernel<16, 32>, a, false, hs);
}

} // namespace cst
