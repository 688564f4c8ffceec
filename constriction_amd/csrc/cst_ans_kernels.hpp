// cst_ans_kernels.hpp -- batched ANS encode/decode kernels for gfx950 (wave64).
//
// One independent AnsCoder<W,S> per LANE.  The recurrences are those of the reference's
// AnsCoder::encode_symbol / decode_symbol (src/stream/stack.rs:1014-1048, 1070-1100); what is
// new is everything around them: the shared cumulative-frequency tables live in LDS, the int32
// symbol matrix is moved in wave-private LDS tiles (coalesced 128-B row segments in HBM, b128
// transposing reads/writes in LDS), and the u64 division by the symbol's probability is an
// exact multiply-high by a per-symbol reciprocal.
#pragma once
#include "cst_common.hpp"

namespace cst {

// ------------------------------------------------------------------------------------------------
// exact state / p
// ------------------------------------------------------------------------------------------------

// high 64 bits of a * m
__device__ __forceinline__ uint64_t mulhi64(uint64_t a, uint32_t m_lo, uint32_t m_hi) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32);
    const uint64_t p00h = __umulhi(a0, m_lo);
    const uint64_t t1 = (uint64_t)a1 * m_lo + p00h;               // v_mad_u64_u32
    const uint64_t t2 = (uint64_t)a0 * m_hi + (uint32_t)t1;       // v_mad_u64_u32
    return (uint64_t)a1 * m_hi + ((t1 >> 32) + (t2 >> 32));       // v_mad_u64_u32
}

// AnsCoder state as a template on the state width.
template <int S> struct StateT;
template <> struct StateT<64> { using type = uint64_t; };
template <> struct StateT<32> { using type = uint32_t; };

// One encode step: (c, p, m) is the table entry of the symbol.
// Returns true if a word has to be emitted (the word is returned in `word`).
template <int W, int S>
__device__ __forceinline__ bool ans_encode_step(typename StateT<S>::type& state, const EncEntry e, int P,
                                                uint32_t& word) {
    using st_t = typename StateT<S>::type;
    st_t st = state;
    // stack.rs:1035-1040: flush one word if the state would overflow
    const bool emit = (uint32_t)(st >> (S - P)) >= e.p;
    word = (uint32_t)st & (W == 32 ? 0xffffffffu : ((1u << (W & 31)) - 1u));
    if (emit) st = (st_t)(st >> (W % S));
    // stack.rs:1042-1045: state = ((state / p) << P) | (c + state % p), with state / p obtained as
    // mulhi(state, floor(2^S / p)) in {q-1, q} followed by one correction.
    st_t q;
    if constexpr (S == 64) q = mulhi64(st, e.m_lo, e.m_hi);
    else q = __umulhi(st, e.m_hi);
    uint32_t r = (uint32_t)st - (uint32_t)q * e.p; // exact: true remainder < 2p < 2^25
    if (r >= e.p) { r -= e.p; q += 1; }
    state = (st_t)((q << P) + (st_t)(e.c + r));
    return emit;
}

// decode step arithmetic (stack.rs:1086-1088): state = (state >> P) * p + (q - c)
template <int S>
__device__ __forceinline__ void ans_decode_advance(typename StateT<S>::type& state, uint32_t quantile, uint32_t c,
                                                   uint32_t p, int P) {
    using st_t = typename StateT<S>::type;
    state = (st_t)((st_t)(state >> P) * (st_t)p + (st_t)(quantile - c));
}

// number of W-bit words the state serialises to (bit_array_to_chunks_truncated, src/lib.rs:719-731)
template <int W, int S>
__device__ __forceinline__ int state_word_count(typename StateT<S>::type st) {
    int bits;
    if constexpr (S == 64) bits = 64 - __clzll((long long)st);
    else bits = 32 - __clz((int)st);
    if (st == 0) bits = 0;
    return (bits + W - 1) / W;
}

// ------------------------------------------------------------------------------------------------
// kernel arguments
// ------------------------------------------------------------------------------------------------

struct AnsEncodeArgs {
    const int32_t* symbols;
    size_t n_streams, n_per_stream;
    const EncEntry* enc;      // shared table [n_symbols]
    int32_t n_symbols, min_symbol, precision;
    uint32_t* words;
    size_t stride_words;
    uint32_t* n_words;
    uint64_t* state;          // may be null unless raw
    int32_t* status;
    uint32_t flags;
};

struct AnsDecodeArgs {
    const uint32_t* words;
    const uint64_t* offsets;  // may be null -> stream * stride_words
    size_t stride_words;
    const uint32_t* n_words;
    int32_t* symbols;
    size_t n_streams, n_per_stream;
    const uint32_t* dec32;    // DecMode-dependent tables
    const uint64_t* dec64;
    const uint32_t* cdf;
    const uint16_t* bucket;
    int32_t bucket_bits;
    int32_t n_symbols, min_symbol, precision;
    uint64_t* state;
    uint32_t* n_words_out;
    int32_t* status;
    uint32_t flags;
};

// ------------------------------------------------------------------------------------------------
// wave-private symbol tiles: 64 stream rows x kTileSyms symbols, row stride kTileStride words
// ------------------------------------------------------------------------------------------------

// global -> registers.  VEC: lane l fetches the 16-B chunk (l & 7) of rows (l >> 3) + 8k, i.e. every
// instruction reads eight full 128-B row segments.  !VEC: lane l fetches word (l & 31) of rows
// (l >> 5) + 2k.  Rows past n_streams are skipped (registers zeroed).
template <bool VEC>
__device__ __forceinline__ void tile_fetch(const int32_t* __restrict__ sym, size_t n_streams, size_t N, size_t s0,
                                           size_t t0, int lane, int32_t (&r)[kTileSyms]) {
    if constexpr (VEC) {
        const int chunk = lane & 7;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const size_t s = s0 + (size_t)((lane >> 3) + 8 * k);
            int4 v = make_int4(0, 0, 0, 0);
            if (s < n_streams) v = *reinterpret_cast<const int4*>(sym + s * N + t0 + 4 * chunk);
            r[4 * k + 0] = v.x; r[4 * k + 1] = v.y; r[4 * k + 2] = v.z; r[4 * k + 3] = v.w;
        }
    } else {
        const int col = lane & 31;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const size_t s = s0 + (size_t)((lane >> 5) + 2 * k);
            r[k] = (s < n_streams) ? sym[s * N + t0 + col] : 0;
        }
    }
}

// registers -> LDS tile (same lane mapping as tile_fetch)
template <bool VEC>
__device__ __forceinline__ void tile_to_lds(int32_t* tile, int lane, const int32_t (&r)[kTileSyms]) {
    if constexpr (VEC) {
        const int chunk = lane & 7;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int row = (lane >> 3) + 8 * k;
            *reinterpret_cast<int4*>(tile + row * kTileStride + 4 * chunk) =
                make_int4(r[4 * k + 0], r[4 * k + 1], r[4 * k + 2], r[4 * k + 3]);
        }
    } else {
        const int col = lane & 31;
#pragma unroll
        for (int k = 0; k < 32; ++k) tile[((lane >> 5) + 2 * k) * kTileStride + col] = r[k];
    }
}

// LDS tile -> global (decode side), same mapping
template <bool VEC>
__device__ __forceinline__ void tile_store(int32_t* __restrict__ sym, size_t n_streams, size_t N, size_t s0, size_t t0,
                                           int lane, const int32_t* tile) {
    if constexpr (VEC) {
        const int chunk = lane & 7;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int row = (lane >> 3) + 8 * k;
            const size_t s = s0 + (size_t)row;
            const int4 v = *reinterpret_cast<const int4*>(tile + row * kTileStride + 4 * chunk);
            if (s < n_streams) *reinterpret_cast<int4*>(sym + s * N + t0 + 4 * chunk) = v;
        }
    } else {
        const int col = lane & 31;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const int row = (lane >> 5) + 2 * k;
            const size_t s = s0 + (size_t)row;
            const int32_t v = tile[row * kTileStride + col];
            if (s < n_streams) sym[s * N + t0 + col] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// encode
// ------------------------------------------------------------------------------------------------

template <int W, int S>
struct EncLane {
    using st_t = typename StateT<S>::type;
    st_t state;
    uint32_t len;       // words emitted so far
    int32_t status;
    uint32_t* out;      // this stream's slab
    uint32_t cap;

    __device__ __forceinline__ void step(int32_t sym, const EncEntry* table, int32_t min_symbol, int32_t n_symbols,
                                         int P) {
        const uint32_t idx = (uint32_t)sym - (uint32_t)min_symbol;
        if (idx >= (uint32_t)n_symbols) {
            if (status == CST_STREAM_OK) status = CST_STREAM_IMPOSSIBLE_SYMBOL; // src/lib.rs:376-385
            return;
        }
        const EncEntry e = table[idx];
        step_entry(e, P);
    }

    __device__ __forceinline__ void step_entry(const EncEntry e, int P) {
        if (status != CST_STREAM_OK) return;
        uint32_t word;
        if (ans_encode_step<W, S>(state, e, P, word)) {
            if (len < cap) out[len] = word;
            else status = CST_STREAM_CAPACITY;
            ++len;
        }
    }

    // into_compressed: append the state's words, least significant first (stack.rs:891-895)
    __device__ __forceinline__ void finish() {
        if (status != CST_STREAM_OK) return;
        const int k = state_word_count<W, S>(state);
        if (len + (uint32_t)k > cap) { status = CST_STREAM_CAPACITY; return; }
        for (int i = 0; i < k; ++i) {
            out[len++] = (uint32_t)(state >> ((i * W) % S)) & (W == 32 ? 0xffffffffu : ((1u << (W & 31)) - 1u));
        }
    }
};

// LAYOUT 0: symbols[stream][t] staged through LDS tiles; LAYOUT 1: symbols[t][stream] read directly.
template <int W, int S, int LAYOUT, bool VEC>
__global__ __launch_bounds__(kBlock) void ans_encode_kernel(const AnsEncodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    EncEntry* table = reinterpret_cast<EncEntry*>(smem);
    const size_t table_bytes = (((size_t)a.n_symbols * sizeof(EncEntry)) + 15) & ~(size_t)15;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = threadIdx.x >> 6;
    int32_t* tile = reinterpret_cast<int32_t*>(smem + table_bytes) + wave_in_block * (kWave * kTileStride);

    // stage the encoder table once per workgroup (16 B per lane per pass, coalesced)
    for (int i = threadIdx.x; i < a.n_symbols; i += blockDim.x) table[i] = a.enc[i];
    __syncthreads();

    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t s0 = wave * kWave;
    if (s0 >= a.n_streams) return; // whole wave idle (after the barrier)
    const size_t s = s0 + lane;
    const bool active = s < a.n_streams;
    const size_t N = a.n_per_stream;
    const int P = a.precision;
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;

    EncLane<W, S> L;
    L.state = (raw && active) ? (typename StateT<S>::type)a.state[s] : 0;
    L.len = 0;
    L.status = active ? CST_STREAM_OK : -1; // -1: lane has no stream, never computes
    L.out = a.words + (active ? s : 0) * a.stride_words;
    L.cap = (uint32_t)(a.stride_words > 0xffffffffull ? 0xffffffffull : a.stride_words);

    if constexpr (LAYOUT == CST_LAYOUT_SYMBOL_MAJOR) {
        // lane-coalesced: every step reads 256 contiguous bytes per wave; keep 4 steps in flight
        const int32_t* col = a.symbols + (active ? s : 0);
        size_t t = N;
        while (t >= 4) {
            t -= 4;
            int32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;
            if (active) {
                v3 = col[(t + 3) * a.n_streams]; v2 = col[(t + 2) * a.n_streams];
                v1 = col[(t + 1) * a.n_streams]; v0 = col[(t + 0) * a.n_streams];
            }
            if (L.status == CST_STREAM_OK) {
                L.step(v3, table, a.min_symbol, a.n_symbols, P);
                L.step(v2, table, a.min_symbol, a.n_symbols, P);
                L.step(v1, table, a.min_symbol, a.n_symbols, P);
                L.step(v0, table, a.min_symbol, a.n_symbols, P);
            }
        }
        while (t > 0) {
            --t;
            if (L.status == CST_STREAM_OK) L.step(col[t * a.n_streams], table, a.min_symbol, a.n_symbols, P);
        }
    } else {
        const int32_t* row = a.symbols + (active ? s : 0) * N;
        const size_t n_full = N / kTileSyms; // full tiles [32k, 32k+32)
        // ragged top part [32*n_full, N): direct (uncoalesced) reads, at most 31 symbols per stream
        for (size_t t = N; t > n_full * kTileSyms;) {
            --t;
            if (L.status == CST_STREAM_OK) L.step(row[t], table, a.min_symbol, a.n_symbols, P);
        }
        if (n_full > 0) {
            int32_t r[kTileSyms];
            tile_fetch<VEC>(a.symbols, a.n_streams, N, s0, (n_full - 1) * kTileSyms, lane, r);
            for (size_t tb = n_full; tb-- > 0;) {
                tile_to_lds<VEC>(tile, lane, r);
                if (tb > 0) tile_fetch<VEC>(a.symbols, a.n_streams, N, s0, (tb - 1) * kTileSyms, lane, r); // prefetch
                const int32_t* my = tile + lane * kTileStride;
                // walk this lane's row backwards, 4 symbols per LDS read; the table lookups of a
                // group are independent of the coder state and are issued ahead of the dependent chain
#pragma unroll 2
                for (int j = kTileSyms / 4 - 1; j >= 0; --j) {
                    const int4 v = *reinterpret_cast<const int4*>(my + 4 * j);
                    const uint32_t i3 = (uint32_t)v.w - (uint32_t)a.min_symbol, i2 = (uint32_t)v.z - (uint32_t)a.min_symbol,
                                   i1 = (uint32_t)v.y - (uint32_t)a.min_symbol, i0 = (uint32_t)v.x - (uint32_t)a.min_symbol;
                    const uint32_t nsym = (uint32_t)a.n_symbols;
                    const bool ok = i3 < nsym && i2 < nsym && i1 < nsym && i0 < nsym;
                    if (__builtin_expect(ok, 1)) {
                        const EncEntry e3 = table[i3], e2 = table[i2], e1 = table[i1], e0 = table[i0];
                        L.step_entry(e3, P); L.step_entry(e2, P); L.step_entry(e1, P); L.step_entry(e0, P);
                    } else if (L.status == CST_STREAM_OK) {
                        L.step(v.w, table, a.min_symbol, a.n_symbols, P); L.step(v.z, table, a.min_symbol, a.n_symbols, P);
                        L.step(v.y, table, a.min_symbol, a.n_symbols, P); L.step(v.x, table, a.min_symbol, a.n_symbols, P);
                    }
                }
            }
        }
    }

    if (!active) return;
    if (raw) {
        a.state[s] = (uint64_t)L.state;
    } else {
        L.finish();
    }
    a.status[s] = L.status;
    a.n_words[s] = (L.status == CST_STREAM_OK) ? L.len : 0u;
}

// ------------------------------------------------------------------------------------------------
// decode
// ------------------------------------------------------------------------------------------------

template <int W, int S>
struct DecLane {
    using st_t = typename StateT<S>::type;
    st_t state;
    uint32_t len;          // words not yet consumed
    const uint32_t* in;    // this stream's words
    uint32_t next_word;    // in[len-1], prefetched
    int32_t status;

    // from_compressed + read_initial_state (stack.rs:299-318, 440-462)
    __device__ __forceinline__ void init_from_words() {
        state = 0;
        if (len == 0) { next_word = 0; return; }
        const uint32_t first = in[--len];
        if (first == 0) { status = CST_STREAM_INVALID_DATA; next_word = 0; return; }
        st_t st = first;
        while (len > 0) {
            st = (st_t)((st << (W % S)) | (st_t)in[--len]);
            if (st >= ((st_t)1 << (S - W))) break;
        }
        state = st;
        next_word = len > 0 ? in[len - 1] : 0u;
    }

    // stack.rs:1089-1097: refill one word if the state dropped below 2^(S-W) and words remain
    __device__ __forceinline__ void refill() {
        if (state < ((st_t)1 << (S - W)) && len > 0) {
            state = (st_t)((state << (W % S)) | (st_t)next_word);
            --len;
            next_word = len > 0 ? in[len - 1] : 0u;
        }
    }
};

template <int MODE> struct DecTables;

// returns the symbol index for the current state and advances the state
template <int W, int S, int MODE>
__device__ __forceinline__ uint32_t ans_decode_symbol(DecLane<W, S>& L, const void* lut, const uint32_t* cdf,
                                                      const uint16_t* bucket, int bucket_shift, int n_symbols, int P) {
    const uint32_t qmask = (P >= 32) ? 0xffffffffu : ((1u << P) - 1u);
    const uint32_t q = (uint32_t)L.state & qmask; // stack.rs:1084
    uint32_t idx, c, p;
    if constexpr (MODE == kDecLut32) {
        const uint32_t e = reinterpret_cast<const uint32_t*>(lut)[q];
        idx = e & 0xffu; c = (e >> 8) & 0xfffu; p = e >> 20;
    } else if constexpr (MODE == kDecLut64) {
        const uint64_t e = reinterpret_cast<const uint64_t*>(lut)[q];
        c = (uint32_t)e & 0xffffffu; p = (uint32_t)(e >> 24) & 0xffffffu; idx = (uint32_t)(e >> 48);
    } else {
        // bucket[q >> shift] = first index whose bin reaches into the bucket; scan forward
        idx = bucket[q >> bucket_shift];
        uint32_t nxt = cdf[idx + 1];
        while (nxt <= q && (int)idx + 1 < n_symbols) { ++idx; nxt = cdf[idx + 1]; }
        c = cdf[idx];
        p = nxt - c;
    }
    ans_decode_advance<S>(L.state, q, c, p, P);
    L.refill();
    return idx;
}

template <int W, int S, int LAYOUT, bool VEC, int MODE, bool LUT_IN_LDS>
__global__ __launch_bounds__(kBlock) void ans_decode_kernel(const AnsDecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = threadIdx.x >> 6;
    const int P = a.precision;

    // ---- stage tables in LDS ----
    size_t lds_off = 0;
    const void* lut = nullptr;
    const uint32_t* cdf = a.cdf;
    const uint16_t* bucket = a.bucket;
    if constexpr (MODE == kDecLut32) {
        if constexpr (LUT_IN_LDS) {
            uint32_t* l = reinterpret_cast<uint32_t*>(smem);
            const int n = 1 << P;
            for (int i = threadIdx.x; i < n; i += blockDim.x) l[i] = a.dec32[i];
            lut = l; lds_off = (size_t)n * 4;
        } else lut = a.dec32;
    } else if constexpr (MODE == kDecLut64) {
        if constexpr (LUT_IN_LDS) {
            uint64_t* l = reinterpret_cast<uint64_t*>(smem);
            const int n = 1 << P;
            for (int i = threadIdx.x; i < n; i += blockDim.x) l[i] = a.dec64[i];
            lut = l; lds_off = (size_t)n * 8;
        } else lut = a.dec64;
    } else {
        if constexpr (LUT_IN_LDS) {
            uint32_t* c = reinterpret_cast<uint32_t*>(smem);
            for (int i = threadIdx.x; i <= a.n_symbols; i += blockDim.x) c[i] = a.cdf[i];
            lds_off = (((size_t)a.n_symbols + 1) * 4 + 15) & ~(size_t)15;
            uint16_t* b = reinterpret_cast<uint16_t*>(smem + lds_off);
            const int nb = (1 << a.bucket_bits);
            for (int i = threadIdx.x; i < nb; i += blockDim.x) b[i] = a.bucket[i];
            lds_off += ((size_t)nb * 2 + 15) & ~(size_t)15;
            cdf = c; bucket = b;
        }
    }
    lds_off = (lds_off + 15) & ~(size_t)15;
    int32_t* tile = reinterpret_cast<int32_t*>(smem + lds_off) + wave_in_block * (kWave * kTileStride);
    __syncthreads();

    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t s0 = wave * kWave;
    if (s0 >= a.n_streams) return;
    const size_t s = s0 + lane;
    const bool active = s < a.n_streams;
    const size_t N = a.n_per_stream;
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;
    const int bucket_shift = P - a.bucket_bits;

    DecLane<W, S> L;
    L.status = active ? CST_STREAM_OK : -1;
    L.len = active ? a.n_words[s] : 0u;
    L.in = a.words + (active ? (a.offsets ? a.offsets[s] : s * a.stride_words) : 0);
    if (raw) {
        L.state = active ? (typename StateT<S>::type)a.state[s] : 0;
        L.next_word = L.len > 0 ? L.in[L.len - 1] : 0u;
    } else {
        L.init_from_words();
    }
    const bool run = L.status == CST_STREAM_OK;

    if constexpr (LAYOUT == CST_LAYOUT_SYMBOL_MAJOR) {
        int32_t* col = a.symbols + (active ? s : 0);
        for (size_t t = 0; t < N; ++t) {
            if (run) {
                const uint32_t idx = ans_decode_symbol<W, S, MODE>(L, lut, cdf, bucket, bucket_shift, a.n_symbols, P);
                col[t * a.n_streams] = a.min_symbol + (int32_t)idx;
            }
        }
    } else {
        int32_t* row = a.symbols + (active ? s : 0) * N;
        const size_t n_full = N / kTileSyms;
        int32_t* my = tile + lane * kTileStride;
        for (size_t tb = 0; tb < n_full; ++tb) {
#pragma unroll 2
            for (int j = 0; j < kTileSyms / 4; ++j) {
                int4 v = make_int4(0, 0, 0, 0);
                if (run) {
                    v.x = a.min_symbol + (int32_t)ans_decode_symbol<W, S, MODE>(L, lut, cdf, bucket, bucket_shift, a.n_symbols, P);
                    v.y = a.min_symbol + (int32_t)ans_decode_symbol<W, S, MODE>(L, lut, cdf, bucket, bucket_shift, a.n_symbols, P);
                    v.z = a.min_symbol + (int32_t)ans_decode_symbol<W, S, MODE>(L, lut, cdf, bucket, bucket_shift, a.n_symbols, P);
                    v.w = a.min_symbol + (int32_t)ans_decode_symbol<W, S, MODE>(L, lut, cdf, bucket, bucket_shift, a.n_symbols, P);
                }
                *reinterpret_cast<int4*>(my + 4 * j) = v;
            }
            // rows of failed / absent streams hold zeros; they are written too (their content is unspecified)
            tile_store<VEC>(a.symbols, a.n_streams, N, s0, tb * kTileSyms, lane, tile);
        }
        for (size_t t = n_full * kTileSyms; t < N; ++t) {
            if (run) {
                const uint32_t idx = ans_decode_symbol<W, S, MODE>(L, lut, cdf, bucket, bucket_shift, a.n_symbols, P);
                row[t] = a.min_symbol + (int32_t)idx;
            }
        }
    }

    if (!active) return;
    a.status[s] = L.status;
    if (raw) {
        a.state[s] = (uint64_t)L.state;
        if (a.n_words_out) a.n_words_out[s] = L.len;
    }
}

} // namespace cst
