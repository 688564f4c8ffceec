// cst_ans_kernels.hpp -- batched ANS encode/decode kernels for gfx950 (wave64).
//
// One independent AnsCoder<W,S> per LANE.  The recurrences are those of the reference's
// AnsCoder::encode_symbol / decode_symbol (src/stream/stack.rs:1014-1048, 1070-1100); what is
// new is everything around them:
//   * the shared cumulative-frequency tables live in LDS,
//   * the int32 symbol matrix moves in wave-private LDS tiles (coalesced 128-B row segments in HBM,
//     b128 transposing accesses in LDS),
//   * compressed words travel through a per-lane LDS ring: the coder step itself is branch-free
//     (one unconditional ring access + selects), and HBM is touched only at scheduled points with
//     aligned 16-byte chunks per lane, issued one interval ahead of use,
//   * the u64 division by the symbol's probability is an exact multiply-high by a per-symbol
//     reciprocal plus one correction.
#pragma once
#include "cst_common.hpp"
#include "cst_ans_asm.hpp"

namespace cst {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kRingSlots = 64;            // words per lane in the LDS word ring
constexpr int kRingWords = kRingSlots * kWave;
constexpr int kAhead = 28;                // decode: words kept issued ahead of the read position
constexpr int kMaxChunksPerPoint = 5;     // 16-byte chunks a lane may move per scheduled point

// Words a coder can emit/consume in T consecutive steps is at most ceil(P*T/W)+1 (each step moves at
// most P bits of information).  Scheduled points -- the ONLY places where a coder touches HBM for compressed
// words -- are placed every 4*G steps with G as large as possible such that this bound is <= 13, which the ring
// geometry above is sized for.  Everything a point waits for was issued a whole interval
// earlier, so neither load latency nor store completion is exposed.
__host__ __device__ inline int groups_per_point(int W, int P) {
    if (32 * P <= 12 * W) return 8;
    if (16 * P <= 12 * W) return 4;
    return 2;
}

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

template <int S> struct StateT;
template <> struct StateT<64> { using type = uint64_t; };
template <> struct StateT<32> { using type = uint32_t; };

template <int W> __device__ __forceinline__ constexpr uint32_t word_mask() { return W == 32 ? 0xffffffffu : ((1u << (W & 31)) - 1u); }

// One branch-free encode step (stack.rs:1035-1045).  Returns 1 if `word` was emitted.
// Generic form (any supported W, S, P).
template <int W, int S>
__device__ __forceinline__ uint32_t ans_encode_step(typename StateT<S>::type& state, const EncEntry e, int P,
                                                    uint32_t& word) {
    using st_t = typename StateT<S>::type;
    st_t st = state;
    const bool emit = (uint32_t)(st >> (S - P)) >= e.p;     // would the state overflow?
    word = (uint32_t)st & word_mask<W>();
    st = emit ? (st_t)(st >> (W % S)) : st;
    // state / p as mulhi(state, floor(2^S / p)) in {q-1, q} followed by one correction
    st_t q;
    if constexpr (S == 64) q = mulhi64(st, e.m_lo, e.m_hi);
    else q = __umulhi(st, e.m_hi);
    uint32_t r = (uint32_t)st - (uint32_t)q * e.p;          // exact: true remainder < 2p < 2^25
    const bool fix = r >= e.p;
    r = fix ? r - e.p : r;
    q += fix ? 1 : 0;
    state = (st_t)((q << P) + (st_t)(e.c + r));
    return emit ? 1u : 0u;
}

// (W,S) = (32,64), 8 <= P <= 24, written on 32-bit halves (what the compiler makes of the 64-bit form is twice as
// many instructions, and a lone wave pays ~4 cycles for every one of them):
//   new_state = st + c + q*(2^P - p)  with q = floor(st / p); the estimate q_est in {q-1, q} is fixed up by
//   adding k = 2^P - p once more when the estimated remainder is >= p.  `p_shl` = p << (32-P), `k` = 2^P - p.
__device__ __forceinline__ uint32_t ans_encode_step_32x64(uint32_t& lo, uint32_t& hi, const EncEntry e, uint32_t p_shl,
                                                          uint32_t k, uint32_t& word) {
    const bool emit = hi >= p_shl;                     // (state >> (64-P)) >= p
    word = lo;
    const uint32_t a0 = emit ? hi : lo;
    const uint32_t a1 = emit ? 0u : hi;
    // q_est = mulhi64(a1:a0, m)
    const uint64_t t1 = (uint64_t)a1 * e.m_lo + (uint64_t)__umulhi(a0, e.m_lo);
    const uint64_t t2 = (uint64_t)a0 * e.m_hi + (uint32_t)t1;
    const uint64_t q = (uint64_t)a1 * e.m_hi + ((t1 >> 32) + (t2 >> 32));
    const uint32_t q_lo = (uint32_t)q, q_hi = (uint32_t)(q >> 32);
    // estimated remainder (exact modulo 2^32; the true value is < 2p)
    const uint32_t r = a0 - q_lo * e.p;
    // st + c + q_est * k  (independent of r)
    const uint64_t base = (((uint64_t)a1 << 32) | a0) + e.c;
    uint64_t t = (uint64_t)q_lo * k + base;
    t += (uint64_t)__umul24(q_hi, k) << 32;            // q_hi < 2^(32-P) <= 2^24, k < 2^24
    t += (r >= e.p) ? k : 0u;
    lo = (uint32_t)t; hi = (uint32_t)(t >> 32);
    return emit ? 1u : 0u;
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
    const uint32_t* dec_cp;   // DecMode-dependent tables
    const uint16_t* dec_idx;
    const uint32_t* cdf;
    const uint16_t* bucket;
    int32_t bucket_bits;
    int32_t n_symbols, min_symbol, precision;
    uint64_t* state;
    uint32_t* n_words_out;
    int32_t* status;
    uint32_t flags;
    uint64_t words_capacity;  // uint32 slots behind `words` (0 = unknown): see word_slice
};

// ------------------------------------------------------------------------------------------------
// wave-private symbol tiles: 64 stream rows x kTileSyms symbols, row stride kTileStride words
// ------------------------------------------------------------------------------------------------

// Orders this wave's LDS traffic for the compiler (the hardware executes one wave's DS ops in order).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// global -> registers.  VEC: lane l fetches the 16-B chunk (l & 7) of rows (l >> 3) + 8k, i.e. every
// instruction reads eight full 128-B row segments.  !VEC: lane l fetches word (l & 31) of rows
// (l >> 5) + 2k.  Rows past n_streams are skipped (registers zeroed).
template <bool VEC>
__device__ __forceinline__ void tile_fetch(const int32_t* __restrict__ sym, size_t n_streams, size_t N, size_t s0,
                                           size_t t0, int lane, int32_t (&r)[kTileSyms]) {
    if constexpr (VEC) {
        const int chunk = lane & 7;
        if (s0 + kWave <= n_streams) {   // wave-uniform common case: no per-row predication
            const int32_t* src = sym + (s0 + (size_t)(lane >> 3)) * N + t0 + 4 * chunk;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const v4i t = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(src + (size_t)(8 * k) * N));
                r[4 * k + 0] = t.x; r[4 * k + 1] = t.y; r[4 * k + 2] = t.z; r[4 * k + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const size_t s = s0 + (size_t)((lane >> 3) + 8 * k);
                int4 v = make_int4(0, 0, 0, 0);
                if (s < n_streams) {
                    const v4i t = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(sym + s * N + t0 + 4 * chunk));
                    v = make_int4(t.x, t.y, t.z, t.w);
                }
                r[4 * k + 0] = v.x; r[4 * k + 1] = v.y; r[4 * k + 2] = v.z; r[4 * k + 3] = v.w;
            }
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
        // all eight LDS reads first (one wait), then the eight stores back to back
        int4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const int4*>(tile + ((lane >> 3) + 8 * k) * kTileStride + 4 * chunk);
        if (s0 + kWave <= n_streams) {   // wave-uniform common case: no per-row predication
            int32_t* dst = sym + (s0 + (size_t)(lane >> 3)) * N + t0 + 4 * chunk;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                v4i t; t.x = v[k].x; t.y = v[k].y; t.z = v[k].z; t.w = v[k].w;
                __builtin_nontemporal_store(t, reinterpret_cast<v4i*>(dst + (size_t)(8 * k) * N));
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const size_t s = s0 + (size_t)((lane >> 3) + 8 * k);
                if (s < n_streams) {
                    v4i t; t.x = v[k].x; t.y = v[k].y; t.z = v[k].z; t.w = v[k].w;
                    __builtin_nontemporal_store(t, reinterpret_cast<v4i*>(sym + s * N + t0 + 4 * chunk));
                }
            }
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
// per-lane word rings in LDS (layout [slot][lane]: every access is bank-conflict free)
// ------------------------------------------------------------------------------------------------

typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(const lds_u32*)p; }

// Output side.  Positions are counted in words from `base16`, the 16-byte aligned address at or below the
// slab start (`shift` = slab start - base16, 0..3), so that every 4-aligned position group is one aligned
// 16-B chunk of HBM.  A coder step writes its candidate word unconditionally and advances `wr` only if the
// word was really emitted; complete chunks leave for HBM at scheduled points.
template <int SLOTS = kRingSlots>
struct RingWriter {
    uint32_t wr;        // words emitted so far (may exceed cap; then nothing more is stored)
    uint32_t flushed;   // positions < flushed are in HBM (multiple of 4, or 0)
    uint32_t cap;       // slab capacity in words (0 for lanes without a stream)
    uint32_t shift;
    uint32_t* base16;
    uint32_t* ring;     // this wave's ring [kRingSlots][kWave]
    int lane;
    uint32_t lane_addr; // LDS byte address of ring[0][lane]

    __device__ __forceinline__ void init(uint32_t* slab, uint32_t capacity, uint32_t* wave_ring, int lane_) {
        lane_addr = lds_addr(wave_ring + lane_);
        // pointer arithmetic (not an integer round trip) so that the accesses stay global_*, not flat_*
        shift = (uint32_t)((reinterpret_cast<uintptr_t>(slab) & 15) >> 2);
        base16 = slab - shift;
        cap = capacity; ring = wave_ring; lane = lane_;
        wr = 0; flushed = 0;
    }

    __device__ __forceinline__ uint32_t* slot(uint32_t pos) const { return ring + ((pos & (SLOTS - 1)) * kWave + lane); }

    __device__ __forceinline__ void push(uint32_t word, uint32_t emit) {
        *slot(wr + shift) = word;   // always written; only becomes part of the stream if wr advances
        wr += emit;
    }

    // scheduled point: move complete aligned 16-B chunks from the ring to HBM
    __device__ __forceinline__ void flush_chunks() {
        const uint32_t end = wr + shift;
#pragma unroll
        for (int k = 0; k < kMaxChunksPerPoint; ++k) {
            if (flushed + 4 <= end) {
                // `flushed` is a multiple of 4 and so is the ring size: the four slots are base + i * kWave
                const uint32_t* b = slot(flushed);
                uint4 v;
                v.x = b[0]; v.y = b[kWave]; v.z = b[2 * kWave]; v.w = b[3 * kWave];
                if (flushed >= shift && flushed + 4 - shift <= cap) {
                    *reinterpret_cast<uint4*>(base16 + flushed) = v;
                } else {
                    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t pos = flushed + i;
                        if (pos >= shift && pos - shift < cap) base16[pos] = w[i];
                    }
                }
                flushed += 4;
            }
        }
    }

    // The same point for 16-byte aligned slabs (shift == 0, cap % 4 == 0) and at most K whole chunks pending: all
    // ring reads first (one LDS wait instead of one per chunk), no ragged-chunk cases, exec-masked stores only.
    template <int K>
    __device__ __forceinline__ void flush_chunks_aligned() {
        const uint32_t n = min((wr - flushed) >> 2, (uint32_t)K);
        uint4 v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t* b = slot(flushed + 4 * k);
            v[k].x = b[0]; v[k].y = b[kWave]; v[k].z = b[2 * kWave]; v[k].w = b[3 * kWave];
        }
#pragma unroll
        for (int k = 0; k < K; ++k)
            if ((uint32_t)k < n && flushed + 4 * k + 4 <= cap) *reinterpret_cast<uint4*>(base16 + flushed + 4 * k) = v[k];
        flushed += 4 * n;
    }

    // rare slow path: make room for a burst (range coder carry resolution)
    __device__ __forceinline__ void push_slow(uint32_t word) {
        if (wr + shift - flushed >= (uint32_t)(SLOTS - 4)) flush_chunks();
        push(word, 1u);
    }

    // end of stream: everything still in the ring goes to HBM (whole chunks first, then the ragged tail)
    __device__ __forceinline__ void drain() {
        for (int guard = 0; guard < 8 && flushed + 4 <= wr + shift; ++guard) flush_chunks();
        for (uint32_t pos = flushed; pos < wr + shift; ++pos)
            if (pos >= shift && pos - shift < cap) base16[pos] = *slot(pos);
        flushed = wr + shift;
    }

    // append one word directly to HBM after drain()
    __device__ __forceinline__ void append_direct(uint32_t word) {
        if (wr < cap) base16[shift + wr] = word;
        ++wr;
    }
};

// Rows that do not start on cache-line boundaries (row length not a multiple of 32 symbols): a lane's tiles begin `pre`
// symbols into its row, at the row's next 128-byte boundary, so that every 128-byte segment a tile store writes is ONE whole
// cache line again (a segment that straddles two lines reaches memory as two partial lines: 1.8x the decode time at
// 65 536 x 4100).  `pre` depends only on where the row starts:
__device__ __forceinline__ uint32_t row_skew(const int32_t* sym, size_t row, size_t N) {
    return (uint32_t)(((128u - (uint32_t)((reinterpret_cast<uintptr_t>(sym) + row * N * 4) & 127u)) & 127u) >> 2);
}
// LDS tile -> HBM for such rows (full wave): row R's 32 symbols go to sym[R][skew(R) + t0 ...]
__device__ __forceinline__ void tile_store_skewed(int32_t* __restrict__ sym, size_t N, size_t s0, size_t t0, int lane, const int32_t* tile) {
    const int chunk = lane & 7;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const size_t R = s0 + (size_t)((lane >> 3) + 8 * k);
        const int4 v = *reinterpret_cast<const int4*>(tile + ((lane >> 3) + 8 * k) * kTileStride + 4 * chunk);
        v4i t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
        __builtin_nontemporal_store(t, reinterpret_cast<v4i*>(sym + R * N + row_skew(sym, R, N) + t0 + 4 * chunk));
    }
}

// LDS tile -> symbols[t][stream] (full wave, n_streams % 4 == 0, 16-byte aligned base): the mapping of the symbol-major
// decoder statements (scripts/gen_decode_loop*.py, SYMBOL_MAJOR): piece k = streams 32 (k & 1) + 4 (lane & 7) .. + 3 of symbol
// row (lane >> 3) + 8 (k >> 1), i.e. eight whole 128-byte lines per store instruction.
__device__ __forceinline__ void tile_store_sm(int32_t* __restrict__ sym, size_t n_streams, size_t s0, size_t t0, int lane, const int32_t* tile) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int r = 32 * (k & 1) + 4 * (lane & 7), t = (lane >> 3) + 8 * (k >> 1);
        v4i v;
        v.x = tile[(r + 0) * kTileStride + t]; v.y = tile[(r + 1) * kTileStride + t];
        v.z = tile[(r + 2) * kTileStride + t]; v.w = tile[(r + 3) * kTileStride + t];
        __builtin_nontemporal_store(v, reinterpret_cast<v4i*>(sym + (t0 + (size_t)t) * n_streams + s0 + (size_t)r));
    }
}

// Input side (stack semantics: words are consumed from the END of the stream's buffer).
// SLOTS words of ring per lane; AHEAD = words kept requested below the read position.
template <int SLOTS = kRingSlots, int AHEAD = kAhead>
struct RingReader {
    uint32_t rd;           // words not yet consumed (next word has stream index rd-1)
    uint32_t shift;
    uint32_t lo_issued;    // lowest position (multiple of 4) whose chunk has been requested
    const uint32_t* base16;
    uint32_t* ring;
    int lane;
    uint4 pend[kMaxChunksPerPoint];
    int32_t pend_pos[kMaxChunksPerPoint];

    __device__ __forceinline__ uint32_t* slot(uint32_t pos) const { return ring + ((pos & (SLOTS - 1)) * kWave + lane); }

    __device__ __forceinline__ void init(const uint32_t* in, uint32_t len, uint32_t* wave_ring, int lane_) {
        // pointer arithmetic (not an integer round trip) so that the accesses stay global_*, not flat_*
        shift = (uint32_t)((reinterpret_cast<uintptr_t>(in) & 15) >> 2);
        base16 = in - shift;
        ring = wave_ring; lane = lane_; rd = len;
#pragma unroll
        for (int k = 0; k < kMaxChunksPerPoint; ++k) { pend_pos[k] = -1; pend[k] = make_uint4(0, 0, 0, 0); }
    }

    // direct HBM access to word `i` of the stream (initial-state words)
    __device__ __forceinline__ uint32_t word_direct(uint32_t i) const { return base16[shift + i]; }

    // fill the ring with the kAhead words below the read position (once per stream).  Every chunk is requested before
    // the first one is waited for: the loop form paid one HBM round trip per chunk at the start of every kernel.
    __device__ __forceinline__ void prime() {
        const uint32_t top = rd + shift;
        const uint32_t start = (top + 3) & ~3u;
        const uint32_t want_lo = top > (uint32_t)AHEAD ? top - AHEAD : 0u;
        constexpr int KMAX = AHEAD / 4 + 2;       // chunks between want_lo and the rounded-up top
        uint4 v[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (start > want_lo + 4u * (uint32_t)k) v[k] = *reinterpret_cast<const uint4*>(base16 + (start - 4u * (uint32_t)(k + 1)));
        lo_issued = start;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (start > want_lo + 4u * (uint32_t)k) {
                lo_issued = start - 4u * (uint32_t)(k + 1);
                uint32_t* b = slot(lo_issued);   // chunk positions are multiples of 4: slots are base + i * kWave
                b[0] = v[k].x; b[kWave] = v[k].y; b[2 * kWave] = v[k].z; b[3 * kWave] = v[k].w;
            }
        }
    }

    // blocking top-up of the window from wherever it stands (used once, between the first tile and the main loop)
    __device__ __forceinline__ void refill_blocking() {
        const uint32_t top = rd + shift;
        const uint32_t want_lo = top > (uint32_t)AHEAD ? top - AHEAD : 0u;
        while (lo_issued > want_lo) {
            lo_issued -= 4;
            const uint4 v = *reinterpret_cast<const uint4*>(base16 + lo_issued);
            uint32_t* b = slot(lo_issued);
            b[0] = v.x; b[kWave] = v.y; b[2 * kWave] = v.z; b[3 * kWave] = v.w;
        }
    }

    // land what the last advance_window() requested, request nothing (before handing the ring to a statement that keeps its
    // own book of requests)
    __device__ __forceinline__ void land_pending() {
#pragma unroll
        for (int k = 0; k < kMaxChunksPerPoint; ++k) {
            if (pend_pos[k] >= 0) {
                uint32_t* b = slot((uint32_t)pend_pos[k]);
                b[0] = pend[k].x; b[kWave] = pend[k].y; b[2 * kWave] = pend[k].z; b[3 * kWave] = pend[k].w;
                pend_pos[k] = -1;
            }
        }
    }

    // scheduled point: land the chunks requested at the previous point, request the next ones
    __device__ __forceinline__ void advance_window() {
#pragma unroll
        for (int k = 0; k < kMaxChunksPerPoint; ++k) {
            if (pend_pos[k] >= 0) {
                uint32_t* b = slot((uint32_t)pend_pos[k]);
                b[0] = pend[k].x; b[kWave] = pend[k].y; b[2 * kWave] = pend[k].z; b[3 * kWave] = pend[k].w;
            }
        }
        const uint32_t top = rd + shift;
        const uint32_t want_lo = top > (uint32_t)AHEAD ? top - AHEAD : 0u;
#pragma unroll
        for (int k = 0; k < kMaxChunksPerPoint; ++k) {
            if (lo_issued > want_lo) {
                lo_issued -= 4;
                pend_pos[k] = (int32_t)lo_issued;
                pend[k] = *reinterpret_cast<const uint4*>(base16 + lo_issued);
            } else {
                pend_pos[k] = -1;
            }
        }
    }

    // Same point with a fixed shape: K chunk slots, the landing of ALL of them unconditional (slots without a
    // request land in the lane's dump rows `dump[i * kWave]`).  With the conditional form the compiler cannot see
    // that "no request" implies "nothing in flight" and protects the registers of pend[] with vmcnt(0) waits
    // right behind the loads of the same point, which exposes a full HBM round trip per tile.
    template <int K>
    __device__ __forceinline__ void advance_window_fixed(uint32_t* dump) {
        static_assert(K <= kMaxChunksPerPoint, "pend[] too small");
#pragma unroll
        for (int k = 0; k < K; ++k) {
            uint32_t* b = pend_pos[k] >= 0 ? slot((uint32_t)pend_pos[k]) : dump;
            b[0] = pend[k].x; b[kWave] = pend[k].y; b[2 * kWave] = pend[k].z; b[3 * kWave] = pend[k].w;
        }
        const uint32_t top = rd + shift;
        const uint32_t want_lo = top > (uint32_t)AHEAD ? top - AHEAD : 0u;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (lo_issued > want_lo) {
                lo_issued -= 4;
                pend_pos[k] = (int32_t)lo_issued;
                pend[k] = *reinterpret_cast<const uint4*>(base16 + lo_issued);
            } else {
                pend_pos[k] = -1;
            }
        }
    }
};

// ------------------------------------------------------------------------------------------------
// encode
// ------------------------------------------------------------------------------------------------

// Per-lane ANS encoder: coder state + the word ring in front of this stream's output slab.
template <int W, int S, int SLOTS = kRingSlots>
struct EncLane {
    using st_t = typename StateT<S>::type;
    st_t state;
    uint32_t bad;       // largest raw table index seen (>= n_symbols <=> impossible symbol)
    RingWriter<SLOTS> out;

    __device__ __forceinline__ void init(uint32_t* slab, uint32_t capacity, uint32_t* wave_ring, int lane_) {
        out.init(slab, capacity, wave_ring, lane_);
        bad = 0; state = 0;
    }

    // branch-free step with a prefetched table entry; FAST selects the 32-bit-halves form (needs P >= 8)
    template <bool FAST>
    __device__ __forceinline__ void step(const EncEntry e, int P) {
        uint32_t word, emit;
        if constexpr (W == 32 && S == 64 && FAST) {
            uint32_t lo = (uint32_t)state, hi = (uint32_t)(state >> 32);
            const uint32_t k = (1u << P) - e.p;
            // ring slot address as (pos << 8 & 0x3f00) | lane_addr: one v_add_lshl + one v_and_or (the kernel checks
            // that the wave's ring is 16-KiB aligned)
            const uint32_t ra = (((out.wr + out.shift) << 8) & (uint32_t)((SLOTS - 1) * kWave * 4)) | out.lane_addr;
            ans_encode_step_asm(lo, hi, out.wr, ra, e, e.p << (32 - P), k, e.c + k);
            state = ((uint64_t)hi << 32) | lo;
            return;
        } else {
            emit = ans_encode_step<W, S>(state, e, P, word);
        }
        out.push(word, emit);
    }

    __device__ __forceinline__ void flush_chunks() { out.flush_chunks(); }

    // end of stream: everything still in the ring, then (unless raw) the state words, least significant
    // first (into_compressed, stack.rs:891-895).  Returns the stream status.
    __device__ __forceinline__ int32_t finish(bool append_state, uint32_t n_symbols, uint32_t& n_words_out) {
        out.drain();
        if (append_state) {
            const int k = state_word_count<W, S>(state);
            for (int i = 0; i < k; ++i) out.append_direct((uint32_t)(state >> ((i * W) % S)) & word_mask<W>());
        }
        n_words_out = out.wr;
        if (bad >= n_symbols) return CST_STREAM_IMPOSSIBLE_SYMBOL;   // src/lib.rs:376-385
        if (out.wr > out.cap) return CST_STREAM_CAPACITY;
        return CST_STREAM_OK;
    }
};

// symbol -> table index, clamped so that the lookup is always legal.  `bad` accumulates the largest raw
// index seen; the stream is flagged at the end if it ever reached n_symbols (2 ops per symbol).
__device__ __forceinline__ uint32_t enc_index(int32_t sym, int32_t min_symbol, uint32_t n_symbols, uint32_t& bad) {
    const uint32_t idx = (uint32_t)sym - (uint32_t)min_symbol;
    bad = max(bad, idx);
    return min(idx, n_symbols - 1u);
}

// The hand-scheduled tile / loop statements of the (32,64), P <= 12 encoder read PACKED table entries
//     { c | (c + 2^P - p) << 16,  p | p << (32 - P),  m_lo,  m_hi }          (8 <= P <= 12: the high half of the second
// word is p << (16 - P), what (state >> 48) is compared with; every operand the step derives from c and p is then a
// half-word select: scripts/gen_encode_loop.py step()).
__device__ __forceinline__ EncEntry pack_entry(const EncEntry e, int P) {
    return EncEntry{e.c | ((e.c + (1u << P) - e.p) << 16), e.p | (e.p << (32 - P)), e.m_lo, e.m_hi};
}
__device__ __forceinline__ EncEntry unpack_entry(const EncEntry e) { return EncEntry{e.c & 0xffffu, e.p & 0xffffu, e.m_lo, e.m_hi}; }

// row[p0 .. p0 + cnt) coded downwards (UP: upwards, the range coder's direction), cnt <= kTileSyms per lane (the ragged ends
// of rows that are skewed onto cache-line boundaries, row_skew above): all reads in flight at once, then the steps out of the
// lane's LDS row `my`.  code(v): one coder step for symbol v; flush(): the scheduled point behind it.
template <bool UP = false, class CODE, class FLUSH>
__device__ __forceinline__ void code_ragged(const int32_t* row, size_t p0, uint32_t cnt, int32_t* my, int32_t fill, CODE&& code, FLUSH&& flush) {
    uint32_t mx = cnt;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d));
    mx = (uint32_t)__builtin_amdgcn_readfirstlane((int)mx);
    if (!mx) return;
    wave_lds_fence();
    {
        int32_t r[kTileSyms];
#pragma unroll
        for (int k = 0; k < kTileSyms; ++k) r[k] = ((uint32_t)k < cnt) ? __builtin_nontemporal_load(row + p0 + k) : fill;
#pragma unroll
        for (int k = 0; k < kTileSyms; ++k) my[k] = r[k];
    }
    wave_lds_fence();
    if constexpr (UP) {
        int32_t v = my[0];
        for (uint32_t k = 0; k < mx; ++k) {
            const int32_t cur = v;
            if (k + 1 < mx) v = my[k + 1];
            if (k < cnt) code(cur);
            flush();
        }
    } else {
        int32_t v = my[mx - 1];
        for (uint32_t k = mx; k-- > 0;) {
            const int32_t cur = v;
            if (k > 0) v = my[k - 1];
            if (k < cnt) code(cur);
            flush();
        }
    }
}

// LAYOUT 0: symbols[stream][t] staged through LDS tiles; LAYOUT 1: symbols[t][stream] read directly.
// GLOBAL_TABLE: the encoder entries stay in HBM / L2 (alphabets too large for LDS: more than ~3800 symbols)
template <int W, int S, int LAYOUT, bool VEC, int G, bool FAST, bool GLOBAL_TABLE = false>
__global__ __launch_bounds__(kBlock) void ans_encode_kernel(const AnsEncodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // packed LDS entries <=> this instantiation has the hand-scheduled tile statements
    constexpr bool PACKED = FAST && W == 32 && S == 64 && G == 8 && !GLOBAL_TABLE;
    // LDS layout: [word rings: one 16-KiB ring per wave, 16-KiB aligned][encoder table][symbol tiles]
    constexpr size_t kRingBytes = (size_t)(kBlock / kWave) * kRingWords * 4;
    const EncEntry* table;
    const size_t table_bytes = GLOBAL_TABLE ? 0 : ((((size_t)a.n_symbols * sizeof(EncEntry)) + 15) & ~(size_t)15);
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = threadIdx.x >> 6;
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem) + wave_in_block * kRingWords;
    int32_t* tile = reinterpret_cast<int32_t*>(smem + kRingBytes + table_bytes) + wave_in_block * (kWave * kTileStride);
    if constexpr (FAST) { if ((lds_addr(ring) & (kRingWords * 4u - 1u)) != 0) __builtin_trap(); }

    if constexpr (GLOBAL_TABLE) {
        table = a.enc;
    } else {
        // stage the encoder table once per workgroup (16 B per lane per pass, coalesced)
        EncEntry* t = reinterpret_cast<EncEntry*>(smem + kRingBytes);
        for (int i = threadIdx.x; i < a.n_symbols; i += blockDim.x) t[i] = PACKED ? pack_entry(a.enc[i], a.precision) : a.enc[i];
        table = t;
    }
    __syncthreads();
    auto entry = [&](uint32_t idx) { const EncEntry e = table[idx]; return PACKED ? unpack_entry(e) : e; };   // for the C++ paths

    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t s0 = wave * kWave;
    if (s0 >= a.n_streams) return; // whole wave idle (after the only barrier)
    const size_t s = s0 + lane;
    const bool active = s < a.n_streams;
    const size_t N = a.n_per_stream;
    const int P = a.precision;
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;
    const uint32_t nsym = (uint32_t)a.n_symbols;

    EncLane<W, S> L;
    L.init(a.words + (active ? s : 0) * a.stride_words,
           active ? (uint32_t)(a.stride_words > 0xffffffffull ? 0xffffffffull : a.stride_words) : 0u, ring, lane);
    L.state = (raw && active) ? (typename StateT<S>::type)a.state[s] : 0;

    if constexpr (LAYOUT == CST_LAYOUT_SYMBOL_MAJOR) {
        // lane-coalesced: every step reads 256 contiguous bytes per wave
        const int32_t* col = a.symbols + (active ? s : 0);
        size_t t = N;
        int countdown = G;
        // ---- the hand-scheduled main loop over the full tiles (full wave, 64-byte aligned slabs, 16-byte aligned rows) ----
        if constexpr (FAST && W == 32 && S == 64 && G == 8 && !GLOBAL_TABLE) {
            const size_t n_full = N / kTileSyms;
            const uint64_t slab_off = (uint64_t)(reinterpret_cast<const unsigned char*>(L.out.base16) - reinterpret_cast<const unsigned char*>(a.words));
            const bool lane_ok = slab_off + 4ull * L.out.cap < 0x100000000ull && (reinterpret_cast<uintptr_t>(L.out.base16) & 63) == 0 &&
                                 (L.out.cap & 15u) == 0 && L.out.shift == 0;
            if ((a.flags & CST_KFLAG_TWO_TILES) && n_full > 0 && s0 + kWave <= a.n_streams && N < (1u << 24) && a.n_streams % 4 == 0 &&
                a.n_streams < (1u << 24) && (reinterpret_cast<uintptr_t>(a.symbols) & 15) == 0 && !__any(!lane_ok)) {
                // the ragged top [32 * n_full, N) first (the coder walks the symbols last to first)
                while (t > n_full * kTileSyms) {
                    --t;
                    L.template step<FAST>(entry(enc_index(col[t * a.n_streams], a.min_symbol, nsym, L.bad)), P);
                    L.flush_chunks();
                }
                int32_t smin = a.min_symbol, smax = a.min_symbol;
                uint32_t goff[8];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    goff[k] = (uint32_t)((((size_t)(lane >> 2) + 16 * (k & 1)) * a.n_streams + 16 * (size_t)(k >> 1) + 4 * (size_t)(lane & 3)) * 4);
                const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(a.symbols + (n_full - 1) * kTileSyms * a.n_streams + s0);
                const uint64_t symbols_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                              (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
                uint32_t lo = (uint32_t)L.state, hi = (uint32_t)((uint64_t)L.state >> 32);
                const uint32_t tr_off = (uint32_t)(((4 * (lane & 3)) * kTileStride + (lane >> 2)) * 4);
                int32_t* tile_b = tile + (kBlock / kWave) * (kWave * kTileStride);
                const uint32_t row_addr[2] = {lds_addr(tile + lane * kTileStride), lds_addr(tile_b + lane * kTileStride)};
                const uint32_t tr_addr[2] = {lds_addr(tile) + tr_off, lds_addr(tile_b) + tr_off};
                __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the statement keeps its own book from here
                ans_encode_tiles_loop_sm(lo, hi, L.out.wr, L.out.flushed, smin, smax, row_addr, tr_addr, L.out.lane_addr, L.out.cap,
                                         (uint32_t)slab_off, lds_addr(table) - 16u * (uint32_t)a.min_symbol, (uint32_t)P, a.words, symbols_base,
                                         (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)n_full),
                                         (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(kTileSyms * a.n_streams * 4)), goff);
                L.state = ((uint64_t)hi << 32) | lo;
                L.bad = max(L.bad, max((uint32_t)smax - (uint32_t)a.min_symbol, (uint32_t)smin - (uint32_t)a.min_symbol));
                t = 0;
            }
        }
        while (t >= 4) {
            t -= 4;
            int32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;
            if (active) {
                v3 = col[(t + 3) * a.n_streams]; v2 = col[(t + 2) * a.n_streams];
                v1 = col[(t + 1) * a.n_streams]; v0 = col[(t + 0) * a.n_streams];
            }
            const EncEntry e3 = entry(enc_index(v3, a.min_symbol, nsym, L.bad)), e2 = entry(enc_index(v2, a.min_symbol, nsym, L.bad)),
                           e1 = entry(enc_index(v1, a.min_symbol, nsym, L.bad)), e0 = entry(enc_index(v0, a.min_symbol, nsym, L.bad));
            L.template step<FAST>(e3, P); L.template step<FAST>(e2, P); L.template step<FAST>(e1, P); L.template step<FAST>(e0, P);
            if (--countdown == 0) { countdown = G; L.flush_chunks(); }
        }
        while (t > 0) {
            --t;
            const int32_t v = active ? col[t * a.n_streams] : 0;
            L.template step<FAST>(entry(enc_index(v, a.min_symbol, nsym, L.bad)), P);
            L.flush_chunks();
        }
    } else {
        const int32_t* row = a.symbols + (active ? s : 0) * N;
        const size_t n_full = N / kTileSyms; // full tiles [32k, 32k+32)
        constexpr bool TILE_ASM = FAST && W == 32 && S == 64 && G == 8 && !GLOBAL_TABLE;
        [[maybe_unused]] int32_t smin = a.min_symbol, smax = a.min_symbol;
        // wave-uniform: every slab of this wave 16-byte aligned and a whole number of chunks long
        const bool aligned_slabs = !__any(L.out.shift != 0 || (L.out.cap & 3u) != 0);
        // ---- main loop as one asm statement (full wave, aligned slabs, 32-bit offsets), rows of any length and alignment:
        // lane l's tiles start row_skew() symbols into its row, so that every 128-byte segment a tile load reads is one
        // whole cache line (a straddling segment costs two: 1.4x the encode time at 65 536 x 4100); what is left of the
        // row above the highest and below the lowest whole tile goes through LDS in two bulk reads ----
        bool done = false;
        if constexpr (TILE_ASM) {
            const uint64_t slab_off = (uint64_t)(reinterpret_cast<const unsigned char*>(L.out.base16) -
                                                 reinterpret_cast<const unsigned char*>(a.words));
            const bool off_ok = slab_off + 4ull * L.out.cap < 0x100000000ull;
            // the statement writes whole 64-byte groups: slabs 64-byte aligned and a whole number of groups long
            const bool groups_ok = (reinterpret_cast<uintptr_t>(L.out.base16) & 63) == 0 && (L.out.cap & 15u) == 0 && L.out.shift == 0;
            if ((a.flags & CST_KFLAG_TWO_TILES) && aligned_slabs && s0 + kWave <= a.n_streams && N >= 4 * kTileSyms && N < (1u << 24) &&
                !__any(!off_ok || !groups_ok)) {
                int32_t* my = tile + lane * kTileStride;
                auto ragged = [&](size_t p0, uint32_t cnt) {
                    code_ragged(row, p0, cnt, my, a.min_symbol,
                                [&](int32_t v) { L.template step<FAST>(entry(enc_index(v, a.min_symbol, nsym, L.bad)), P); },
                                [&]() { L.flush_chunks(); });
                };
                const uint32_t pre = row_skew(a.symbols, s, N);
                uint32_t max_pre = pre;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) max_pre = max(max_pre, (uint32_t)__shfl_xor((int)max_pre, d));
                max_pre = (uint32_t)__builtin_amdgcn_readfirstlane((int)max_pre);
                const size_t n_t = (N - max_pre) / kTileSyms;           // whole tiles every lane has (>= 3)
                const size_t top = pre + n_t * kTileSyms;               // this lane's symbols [top, N) come first: fewer than 64
                const uint32_t n_top = (uint32_t)(N - top);
                ragged(top + kTileSyms, n_top > (uint32_t)kTileSyms ? n_top - (uint32_t)kTileSyms : 0u);
                ragged(top, min(n_top, (uint32_t)kTileSyms));
                wave_lds_fence();
                uint32_t goff[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const size_t R = (size_t)(lane >> 3) + 8 * k;
                    goff[k] = (uint32_t)((R * N + row_skew(a.symbols, s0 + R, N) + 4 * (size_t)(lane & 7)) * 4);
                }
                const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(a.symbols + s0 * N + (n_t - 1) * kTileSyms);
                // wave-uniform base in SGPRs (readfirstlane returns int: go through uint32_t)
                const uint64_t symbols_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                              (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
                uint32_t lo = (uint32_t)L.state, hi = (uint32_t)((uint64_t)L.state >> 32);
                const uint32_t tr_off = (uint32_t)(((lane >> 3) * kTileStride + 4 * (lane & 7)) * 4);
                int32_t* tile_b = tile + (kBlock / kWave) * (kWave * kTileStride);   // second buffer behind all first ones
                const uint32_t row_addr[2] = {lds_addr(tile + lane * kTileStride), lds_addr(tile_b + lane * kTileStride)};
                const uint32_t tr_addr[2] = {lds_addr(tile) + tr_off, lds_addr(tile_b) + tr_off};
                __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the statement keeps its own book from here
                ans_encode_tiles_loop(lo, hi, L.out.wr, L.out.flushed, smin, smax, row_addr, tr_addr,
                                      L.out.lane_addr, L.out.cap, (uint32_t)slab_off,
                                      lds_addr(table) - 16u * (uint32_t)a.min_symbol, (uint32_t)P, a.words, symbols_base,
                                      (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)n_t), goff);
                L.state = ((uint64_t)hi << 32) | lo;
                ragged(0, pre);
                done = true;
            }
        }
        // ragged top part [32*n_full, N): direct (uncoalesced) reads, at most 31 symbols per stream
        for (size_t t = done ? 0 : N; t > n_full * kTileSyms;) {
            --t;
            const int32_t v = active ? row[t] : 0;
            L.template step<FAST>(entry(enc_index(v, a.min_symbol, nsym, L.bad)), P);
            L.flush_chunks();
        }
        if (n_full > 0) {
            // Symbol tiles are prefetched TWO tiles ahead into two register sets: under full load an HBM round trip is
            // longer than the ~2.4 us one tile takes, and a single tile of lead left ~35 cycles per symbol of vmcnt wait.
            int32_t rA[kTileSyms], rB[kTileSyms];
            if (!done) {
                tile_fetch<VEC>(a.symbols, a.n_streams, N, s0, (n_full - 1) * kTileSyms, lane, rA);
                if (n_full >= 2) tile_fetch<VEC>(a.symbols, a.n_streams, N, s0, (n_full - 2) * kTileSyms, lane, rB);
            }
            auto tile_body = [&](size_t tb, int32_t (&r)[kTileSyms]) {
                wave_lds_fence();
                tile_to_lds<VEC>(tile, lane, r);
                wave_lds_fence();
                // words of the previous tile first (stores), then the prefetch (loads): everything younger than the
                // OTHER register set's loads is then a store, so waiting for that set never waits for these loads
                if (TILE_ASM && aligned_slabs) L.out.template flush_chunks_aligned<3>();   // <= 3 + 12 words pending
                else L.flush_chunks();
                if (tb >= 2) tile_fetch<VEC>(a.symbols, a.n_streams, N, s0, (tb - 2) * kTileSyms, lane, r);
                const int32_t* my = tile + lane * kTileStride;
                if constexpr (TILE_ASM) {
                    uint32_t lo = (uint32_t)L.state, hi = (uint32_t)((uint64_t)L.state >> 32);
                    ans_encode_tile32(lo, hi, L.out.wr, smin, smax, lds_addr(my), lds_addr(table) - 16u * (uint32_t)a.min_symbol,
                                      (uint32_t)P, L.out.shift, L.out.lane_addr);
                    L.state = ((uint64_t)hi << 32) | lo;
                } else {
                    // Walk this lane's row backwards, 4 symbols per LDS read.  Software pipeline: the symbols of group
                    // j-2 are requested and the table entries of group j-1 are fetched before the dependent chain of
                    // group j runs (none of it depends on the state), so no LDS latency is exposed inside the tile.
                    constexpr int NG = kTileSyms / 4;
                    int4 v = *reinterpret_cast<const int4*>(my + 4 * (NG - 1));
                    int4 vn = *reinterpret_cast<const int4*>(my + 4 * (NG - 2));
                    EncEntry e3 = entry(enc_index(v.w, a.min_symbol, nsym, L.bad)), e2 = entry(enc_index(v.z, a.min_symbol, nsym, L.bad)),
                             e1 = entry(enc_index(v.y, a.min_symbol, nsym, L.bad)), e0 = entry(enc_index(v.x, a.min_symbol, nsym, L.bad));
#pragma unroll
                    for (int j = NG - 1; j >= 0; --j) {
                        EncEntry n3 = e3, n2 = e2, n1 = e1, n0 = e0;
                        int4 vnn = vn;
                        if (j > 1) vnn = *reinterpret_cast<const int4*>(my + 4 * (j - 2));
                        if (j > 0) {
                            n3 = entry(enc_index(vn.w, a.min_symbol, nsym, L.bad)); n2 = entry(enc_index(vn.z, a.min_symbol, nsym, L.bad));
                            n1 = entry(enc_index(vn.y, a.min_symbol, nsym, L.bad)); n0 = entry(enc_index(vn.x, a.min_symbol, nsym, L.bad));
                        }
                        L.template step<FAST>(e3, P); L.template step<FAST>(e2, P); L.template step<FAST>(e1, P); L.template step<FAST>(e0, P);
                        e3 = n3; e2 = n2; e1 = n1; e0 = n0; vn = vnn;
                        if (j % G == 0 && j != 0) L.flush_chunks();   // static mid-tile points (G < 8 only)
                    }
                }
            };
            for (size_t tb = done ? 0 : n_full; tb > 0;) {
                tile_body(--tb, rA);
                if (tb == 0) break;
                tile_body(--tb, rB);
            }
            if constexpr (TILE_ASM) {
                // fold the extremes into `bad` (largest raw table index): a symbol below min_symbol wraps to a huge index
                L.bad = max(L.bad, max((uint32_t)smax - (uint32_t)a.min_symbol, (uint32_t)smin - (uint32_t)a.min_symbol));
            }
        }
    }

    uint32_t n_words = 0;
    const int32_t status = L.finish(!raw, nsym, n_words);
    if (!active) return;
    if (raw) a.state[s] = (uint64_t)L.state;
    a.status[s] = status;
    a.n_words[s] = (status == CST_STREAM_OK) ? n_words : 0u;
}

// ------------------------------------------------------------------------------------------------
// decode
// ------------------------------------------------------------------------------------------------

// Per-lane ANS decoder: coder state + the word ring that runs ahead of the read position.
template <int W, int S, int SLOTS = kRingSlots, int AHEAD = kAhead>
struct DecLane {
    using st_t = typename StateT<S>::type;
    st_t state;
    int32_t status;
    RingReader<SLOTS, AHEAD> in;

    __device__ __forceinline__ void init(const uint32_t* words, uint32_t len, uint32_t* wave_ring, int lane_) {
        in.init(words, len, wave_ring, lane_);
        status = CST_STREAM_OK; state = 0;
    }

    // from_compressed + read_initial_state (stack.rs:299-318, 440-462), straight from HBM
    __device__ __forceinline__ void read_initial_state() {
        if (in.rd == 0) return;
        const uint32_t first = in.word_direct(--in.rd);
        if (first == 0) { status = CST_STREAM_INVALID_DATA; in.rd = 0; return; }
        st_t st = first;
        while (in.rd > 0) {
            st = (st_t)((st << (W % S)) | (st_t)in.word_direct(--in.rd));
            if (st >= ((st_t)1 << (S - W))) break;
        }
        state = st;
    }
};

// DecoderModel::quantile_function for a tabulated model (lookup_contiguous.rs:564-605): quantile -> (index, left
// cumulative, probability), from the LDS/global image selected by MODE.
template <int MODE>
__device__ __forceinline__ void lookup_quantile(uint32_t q, const DecLut lut, const uint32_t* cdf, const uint16_t* bucket,
                                                int bucket_shift, int n_symbols, uint32_t& idx, uint32_t& c, uint32_t& p) {
    if constexpr (MODE == kDecLutCP) {
        const uint32_t e = lut.cp[q];
        c = e & 0xffffu; p = e >> 16;
        idx = lut.sym ? (uint32_t)(lut.sym[q] - lut.min_symbol) : (uint32_t)lut.idx[q];
    } else {
        if (lut.b16) {
            uint4 e = lut.b16[q >> bucket_shift];
            if (lut.sub_bits) {             // (uniform) this image has second-level tables: DecLut, cst_common.hpp
                if (__any(q >= e.w)) {
                    // slot (bucket mod kSubTables), part = the next sub_bits bits of the quantile; the entry there is this lane's if
                    // it is a real entry whose first cumulative does not lie above q (one of a lower bucket passes, too, and
                    // leads to a longer walk below)
                    const uint32_t at = (q >> (bucket_shift - lut.sub_bits)) & (((uint32_t)kSubTables << lut.sub_bits) - 1u);
                    if (q >= e.w) {
                        const uint4 e2 = lut.sub[at];
                        const uint32_t c2 = e2.x & ((1u << lut.idx_shift) - 1u);
                        if (c2 <= q && e2.y > c2) e = e2;
                    }
                }
            }
            const uint32_t c0 = e.x & ((1u << lut.idx_shift) - 1u), i0 = e.x >> lut.idx_shift;
            const uint32_t k = (q >= e.y ? 1u : 0u) + (q >= e.z ? 1u : 0u);
            c = k == 0 ? c0 : (k == 1 ? e.y : e.z);
            uint32_t nxt = k == 0 ? e.y : (k == 1 ? e.z : e.w);
            idx = i0 + k;
            if (__any(q >= e.w)) {          // more than three symbols begin inside this bucket below q (the far tails): walk
                if (q >= e.w) {
                    idx = i0 + 3u;
                    while (cdf[idx + 1u] <= q) ++idx;     // (cdf[n] = 2^P lies above every quantile)
                    c = cdf[idx]; nxt = cdf[idx + 1u];
                }
            }
            p = nxt - c;
            return;
        }
        // bucket[q >> shift] = first index whose bin reaches into the bucket.  Probe the next three boundaries at once
        // (cdf[n] = 2^P lies above every quantile, so clamped probes never count) and advance by the number of
        // boundaries at or below q; a wave-uniform loop repeats only while some lane had to advance by all three.
        idx = bucket[q >> bucket_shift];
        const uint32_t n = (uint32_t)n_symbols;
        uint32_t c0, c1, c2, c3, cnt;
        for (uint32_t guard = 0;; guard += 3) {      // (the guard only bounds the loop should a table ever be corrupt)
            c0 = cdf[idx]; c1 = cdf[min(idx + 1, n)]; c2 = cdf[min(idx + 2, n)]; c3 = cdf[min(idx + 3, n)];
            cnt = (c1 <= q ? 1u : 0u) + (c2 <= q ? 1u : 0u) + (c3 <= q ? 1u : 0u);
            if (guard > n || !__any(cnt == 3)) break;
            idx = min(idx + cnt, n - 1u);    // (lanes with cnt < 3 are already in place: their next round counts 0)
        }
        idx = min(idx + cnt, n - 1u);
        c = cnt == 0 ? c0 : (cnt == 1 ? c1 : c2);
        p = (cnt == 0 ? c1 : (cnt == 1 ? c2 : c3)) - c;
    }
}

// One branch-free decode step (stack.rs:1084-1097): returns the symbol index.
template <int W, int S, int MODE, bool FAST, class LANE>
__device__ __forceinline__ uint32_t ans_decode_step(LANE& L, const DecLut lut, const uint32_t* cdf,
                                                    const uint16_t* bucket, int bucket_shift, int n_symbols, int P) {
    using st_t = typename StateT<S>::type;
    const uint32_t qmask = (P >= 32) ? 0xffffffffu : ((1u << P) - 1u);
    const uint32_t q = (uint32_t)L.state & qmask;
    uint32_t next_word;
    if constexpr (W == 32 && S == 64 && FAST) {
        // Unconditional ring read, issued BEFORE the table lookup so that the lookup's own wait covers it.
        // volatile: the load stays HERE (the optimiser would otherwise sink it into a branch on `refill` and put the LDS
        // latency back on the critical path) and stays visible to the compiler's lgkmcnt bookkeeping
        // (through the LDS address space explicitly: address-space inference leaves a volatile access alone, and as a
        // generic access this was a flat_load -- which waits for the wave's outstanding global loads and stores, too)
        next_word = *(const volatile lds_u32*)L.in.slot(L.in.rd - 1u + L.in.shift);
    } else {
        next_word = *L.in.slot(L.in.rd - 1u + L.in.shift);   // ignored if no refill
    }
    uint32_t idx, c, p;
    lookup_quantile<MODE>(q, lut, cdf, bucket, bucket_shift, n_symbols, idx, c, p);
    if constexpr (W == 32 && S == 64 && FAST) {
        // 32-bit halves: (state >> P) * p + (q - c); the high product fits mul_u24 because P >= 8
        const uint32_t lo = (uint32_t)L.state, hi = (uint32_t)(L.state >> 32);
        const uint32_t s_lo = __builtin_amdgcn_alignbit(hi, lo, P), s_hi = hi >> P;
        const uint64_t t = (uint64_t)s_lo * p + (uint64_t)(q - c);
        const uint32_t t_lo = (uint32_t)t;
        const uint32_t t_hi = __umul24(s_hi, p) + (uint32_t)(t >> 32);
        // refill <=> t < 2^32 and words remain <=> t_hi < min(rd, 1).  Hand-scheduled so that it stays one compare
        // and three selects (2 wait states between the VALU write of vcc and its VALU readers).
        uint32_t new_lo, new_hi, new_rd, have;
        asm volatile(
            "v_min_u32 %3, %4, 1\n\t"
            "v_cmp_lt_u32 vcc, %5, %3\n\t"
            "s_nop 1\n\t"
            "v_cndmask_b32 %0, %6, %7, vcc\n\t"          // lo' = refill ? next_word : t_lo
            "v_cndmask_b32 %1, %5, %6, vcc\n\t"          // hi' = refill ? t_lo : t_hi
            "v_subbrev_co_u32 %2, vcc, 0, %4, vcc"        // rd' = rd - refill
            : "=&v"(new_lo), "=&v"(new_hi), "=&v"(new_rd), "=&v"(have)
            : "v"(L.in.rd), "v"(t_hi), "v"(t_lo), "v"(next_word)
            : "vcc");
        L.state = ((uint64_t)new_hi << 32) | new_lo;
        L.in.rd = new_rd;
    } else {
        st_t st = (st_t)((st_t)(L.state >> P) * (st_t)p + (st_t)(q - c));
        const bool refill = st < ((st_t)1 << (S - W)) && L.in.rd > 0;
        L.state = refill ? (st_t)((st << (W % S)) | (st_t)next_word) : st;
        L.in.rd -= refill ? 1u : 0u;
    }
    return idx;
}

// Copies the decoder tables selected by MODE into LDS (if LUT_IN_LDS) and returns the bytes used.
// SUB: with second-level tables behind the bucket entries (kSubAreaBytes more; every thread of the workgroup must call)
// table_bits (bucket entries only): 0 = the model's bucket array as it is; otherwise 2^table_bits entries whose first symbols are found by a
// binary search of the cdf copy (the model tabulates 2^11 buckets; a kernel with LDS to spare halves their width: fewer quantiles lie beyond
// their entry's third symbol)
template <int MODE, bool LUT_IN_LDS, bool SUB = false>
__device__ __forceinline__ size_t stage_decoder_tables(unsigned char* smem, int P, const uint32_t* dec_cp, const uint16_t* dec_idx,
                                                       const uint32_t* g_cdf, const uint16_t* g_bucket, int bucket_bits,
                                                       int n_symbols, DecLut& lut, const uint32_t*& cdf,
                                                       const uint16_t*& bucket, int table_bits = 0) {
    size_t lds_off = 0;
    lut.cp = dec_cp; lut.idx = dec_idx;
    if constexpr (MODE == kDecLutCP) {
        if constexpr (LUT_IN_LDS) {
            uint32_t* l = reinterpret_cast<uint32_t*>(smem);
            const int n = 1 << P;
            for (int i = threadIdx.x; i < n; i += blockDim.x) l[i] = dec_cp[i];
            uint16_t* x = reinterpret_cast<uint16_t*>(smem + (size_t)n * 4);
            for (int i = threadIdx.x; i < n; i += blockDim.x) x[i] = dec_idx[i];
            lut.cp = l; lut.idx = x; lds_off = (size_t)n * 6;
        }
    } else {
        if constexpr (LUT_IN_LDS) {
            uint32_t* c = reinterpret_cast<uint32_t*>(smem);
            for (int i = threadIdx.x; i <= n_symbols; i += blockDim.x) c[i] = g_cdf[i];
            lds_off = (((size_t)n_symbols + 1) * 4 + 15) & ~(size_t)15;
            const bool searched = table_bits != 0 && table_bits != bucket_bits;
            const int model_bits = bucket_bits;
            if (searched) { bucket_bits = table_bits; __syncthreads(); }        // (the searches read the cdf copy)
            const int nb = (1 << bucket_bits);
            if (bucket16_usable(n_symbols, P)) {
                uint4* b = reinterpret_cast<uint4*>(smem + lds_off);
                const uint32_t n = (uint32_t)n_symbols;
                const int ishift = bucket16_index_shift(n_symbols);
                auto entry_at = [&](uint32_t i0) {
                    return make_uint4(g_cdf[i0] | (i0 << ishift), g_cdf[min(i0 + 1u, n)], g_cdf[min(i0 + 2u, n)], g_cdf[min(i0 + 3u, n)]);
                };
                const int shift = P - bucket_bits;
                const int sb = SUB ? min(kSubBitsMax, shift) : 0;
                uint32_t* ctl = reinterpret_cast<uint32_t*>(smem + lds_off + (size_t)nb * 16 + (size_t)kSubTables * kSubTableBytes);
                if constexpr (SUB) {
                    if (threadIdx.x < kSubTables) ctl[4 + threadIdx.x] = 0xffffffffu;       // slot owners: none yet
                    __syncthreads();
                }
                for (int i = threadIdx.x; i < nb; i += blockDim.x) {
                    uint32_t first = 0;
                    if (searched) {                                                // the symbol that holds the bucket's first quantile
                        const uint32_t q0 = (uint32_t)i << shift;
                        uint32_t hi = n;                                           // c[0] = 0 <= q0 < 2^P = c[n]
                        if (bucket_bits > model_bits) {                            // ... lies between the first symbols of the model's bucket and of its successor
                            const int d = bucket_bits - model_bits;
                            first = g_bucket[i >> d];
                            if (((i >> d) + 1) < (1 << model_bits)) hi = min(n, (uint32_t)g_bucket[(i >> d) + 1] + 1u);
                        }
                        while (hi - first > 1) { const uint32_t mid = (first + hi) >> 1; if (c[mid] <= q0) first = mid; else hi = mid; }
                    } else first = g_bucket[i];
                    const uint4 e = entry_at(first);
                    if constexpr (SUB) {
                        // more than three symbols begin in this bucket <=> a quantile of it lies at or above the fourth cumulative:
                        // the bucket claims slot (i mod kSubTables) for a second-level table (the lowest claimant keeps it; the
                        // others walk: DecLut, cst_common.hpp)
                        if (sb > 0 && e.w < ((uint32_t)(i + 1) << shift)) atomicMin(&ctl[4 + (i & (kSubTables - 1))], (uint32_t)i);
                    }
                    b[i] = e;
                }
                lds_off += (size_t)nb * 16;
                if constexpr (SUB) {
                    __syncthreads();            // (the cdf copy `c` is complete, too)
                    uint4* sub = b + nb;
                    for (uint32_t t = threadIdx.x; t < ((uint32_t)kSubTables << sb); t += blockDim.x) {
                        const uint32_t slot = t >> sb, part = t & ((1u << sb) - 1u), owner = ctl[4 + slot];
                        uint4 e = make_uint4(0u, 0u, 0u, 0u);                      // free slot: not an entry (e.y > c fails)
                        if (owner != 0xffffffffu) {
                            const uint32_t q0 = (owner << shift) + (part << (shift - sb));
                            uint32_t lo = 0, hi = n;                               // c[0] = 0 <= q0 < 2^P = c[n]
                            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (c[mid] <= q0) lo = mid; else hi = mid; }
                            e = entry_at(lo);
                        }
                        sub[t] = e;                                                // entry (slot << sb | part)
                    }
                    lds_off += kSubAreaBytes;
                }
                lut.b16 = b; lut.idx_shift = ishift; lut.sub_bits = sb; lut.sub = SUB ? b + nb : nullptr; cdf = c;
            } else {
                uint16_t* b = reinterpret_cast<uint16_t*>(smem + lds_off);
                for (int i = threadIdx.x; i < nb; i += blockDim.x) b[i] = g_bucket[i];
                lds_off += ((size_t)nb * 2 + 15) & ~(size_t)15;
                cdf = c; bucket = b;
            }
        }
    }
    return lds_off;
}

// Is the hand-scheduled tile decoder used for this instantiation?  (host and kernel must agree: it changes the
// LDS image of the tables).  G == 8 <=> P <= 12 for W = 32 (groups_per_point).
constexpr bool decode_uses_tile_asm(int W, int S, int MODE, bool LUT_IN_LDS, int G, bool FAST) {
    return FAST && W == 32 && S == 64 && MODE == kDecLutCP && LUT_IN_LDS && G == 8;
}

// LDS image for ans_decode_tile32: cp[2^P] at +0, decoded symbols (int32) at +kTileSymOffset
// WITH_K: a third table at +kTileKOffset for the main loops of scripts/gen_decode_loop.py (K_PRED):
//     K[q] = floor((2^32 - 1 - (q - c)) / p)   --   (state >> P) * p + (q - c) < 2^32  <=>  state >> P <= K[q]
// i.e. stack.rs:1091's `state < 2^32` (refill) decided by ONE 32-bit compare against the table instead of the 64-bit product.
template <bool WITH_K = false>
__device__ __forceinline__ void stage_tile_tables(unsigned char* lds, int P, const uint32_t* dec_cp, const uint16_t* dec_idx,
                                                  int32_t min_symbol, DecLut& lut) {
    uint32_t* l = reinterpret_cast<uint32_t*>(lds);
    int32_t* x = reinterpret_cast<int32_t*>(lds + kTileSymOffset);
    const int n = 1 << P;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t e = dec_cp[i];
        l[i] = e; x[i] = min_symbol + (int32_t)dec_idx[i];
        if constexpr (WITH_K) {
            const uint32_t c = e & 0xffffu, p = e >> 16;           // (p >= 1: every quantile lies in a bin)
            reinterpret_cast<uint32_t*>(lds + kTileKOffset)[i] = (0xffffffffu - ((uint32_t)i - c)) / (p ? p : 1u);
        }
    }
    lut.cp = l; lut.idx = nullptr; lut.sym = x; lut.min_symbol = min_symbol;
}

// LDS bytes of a decode workgroup that uses the hand-scheduled tile decoder:
// [rings 4 x 8 KiB][cp + sym tables 32 KiB (+ K: 48 KiB)][two symbol tiles per wave][dump rows]
#ifdef CST_DEC_KPRED      // (experiment: scripts/gen_decode_loop.py, K_PRED)
constexpr bool kDecKPred = true;
#else
constexpr bool kDecKPred = false;
#endif
constexpr size_t kDecTileAsmLdsBytes = (size_t)(kBlock / kWave) * kDecRingSlots * kWave * 4 + (kDecKPred ? kTileLutKBytes : kTileLutBytes) +
                                       2 * (size_t)(kBlock / kWave) * kWave * kTileStride * 4 + kTileDumpBytes;

// LDS layout: [word rings: one per wave, aligned to their size][tables][symbol tiles][dump rows (tile asm only)]
template <int W, int S, int LAYOUT, bool VEC, int MODE, bool LUT_IN_LDS, int G, bool FAST>
__global__ __launch_bounds__(kBlock) void ans_decode_kernel(const AnsDecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool TILE_ASM = decode_uses_tile_asm(W, S, MODE, LUT_IN_LDS, G, FAST);
    constexpr int SLOTS = TILE_ASM ? kDecRingSlots : kRingSlots;
    constexpr int AHEAD = TILE_ASM ? kDecAhead : kAhead;
    constexpr size_t kWaveRingWords = (size_t)SLOTS * kWave;
    constexpr size_t kRingBytes = (size_t)(kBlock / kWave) * kWaveRingWords * 4;
    constexpr size_t kTileWords = (size_t)kWave * kTileStride;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = threadIdx.x >> 6;
    const int P = a.precision;

    // ---- stage tables in LDS ----
    DecLut lut{};
    const uint32_t* cdf = a.cdf;
    const uint16_t* bucket = a.bucket;
    size_t lds_off;
    if constexpr (TILE_ASM) {
        stage_tile_tables<kDecKPred>(smem + kRingBytes, P, a.dec_cp, a.dec_idx, a.min_symbol, lut);
        lds_off = kDecKPred ? kTileLutKBytes : kTileLutBytes;
    } else {
        lds_off = stage_decoder_tables<MODE, LUT_IN_LDS>(smem + kRingBytes, P, a.dec_cp, a.dec_idx, a.cdf, a.bucket, a.bucket_bits,
                                                         a.n_symbols, lut, cdf, bucket);
        lds_off = (lds_off + 15) & ~(size_t)15;
    }
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem) + wave_in_block * kWaveRingWords;
    // tile asm: two tile buffers per wave (A, B), the B buffers behind all A buffers
    int32_t* tile = reinterpret_cast<int32_t*>(smem + kRingBytes + lds_off) + wave_in_block * kTileWords;
    __syncthreads();

    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t s0 = wave * kWave;
    if (s0 >= a.n_streams) return;
    const size_t s = s0 + lane;
    const bool active = s < a.n_streams;
    const size_t N = a.n_per_stream;
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;
    const int bucket_shift = P - a.bucket_bits;

    DecLane<W, S, SLOTS, AHEAD> L;
    const WordSlice ws = active ? word_slice(a.offsets, a.stride_words, a.n_words, s, a.words_capacity) : WordSlice{0, 0u, false};
    L.init(a.words + ws.off, ws.n, ring, lane);
    if (raw) L.state = active ? (typename StateT<S>::type)a.state[s] : 0;
    else L.read_initial_state();
    L.in.prime();
    wave_lds_fence();

    // one symbol (tails, symbol-major layout)
    auto next_index = [&]() -> uint32_t {
        return ans_decode_step<W, S, MODE, FAST>(L, lut, cdf, bucket, bucket_shift, a.n_symbols, P);
    };

    if constexpr (LAYOUT == CST_LAYOUT_SYMBOL_MAJOR) {
        int32_t* col = a.symbols + (active ? s : 0);
        int countdown = 4 * G;
        size_t t_done = 0;
        if constexpr (TILE_ASM) {
            // ---- the hand-scheduled main loop over the full tiles (see the stream-major branch below for the flow) ----
            const size_t n_full = N / kTileSyms;
            const unsigned char* words_base = reinterpret_cast<const unsigned char*>(reinterpret_cast<uintptr_t>(a.words) & ~(uintptr_t)15);
            const uint64_t w_off = (uint64_t)(reinterpret_cast<const unsigned char*>(L.in.base16) - words_base);
            const bool off_ok = w_off + 4ull * ((uint64_t)L.in.rd + 8) < 0x80000000ull;
            if ((s0 + kWave <= a.n_streams) && n_full >= 2 && N < (1u << 24) && a.n_streams % 4 == 0 && a.n_streams < (1u << 24) &&
                (reinterpret_cast<uintptr_t>(a.symbols) & 15) == 0 && !__any(!off_ok)) {
                if ((lds_addr(ring) & (uint32_t)(kWaveRingWords * 4 - 1)) != 0) __builtin_trap();
                int32_t* my = tile + lane * kTileStride;
                uint32_t lo = (uint32_t)L.state, hi = (uint32_t)(L.state >> 32);
                const uint32_t qmask = (1u << P) - 1u;
                const uint32_t lut_addr = lds_addr(lut.cp), lane_addr = lds_addr(ring + lane);
                int32_t* tile_b = tile + (kBlock / kWave) * kTileWords;
                uint32_t* dump = reinterpret_cast<uint32_t*>(smem + kRingBytes + lds_off + 2 * (size_t)(kBlock / kWave) * kTileWords * 4) +
                                 wave_in_block * (4 * kWave) + lane;
                __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
                ans_decode_tile32(lo, hi, L.in.rd, lut_addr, qmask, (uint32_t)P, lds_addr(my), L.in.shift - 1u, lane_addr, kDecRingMask);
                L.in.refill_blocking();
                wave_lds_fence();
                __builtin_amdgcn_s_waitcnt(0x0F70);
                uint32_t goff[8];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    goff[k] = (uint32_t)((((size_t)(lane >> 3) + 8 * (k >> 1)) * a.n_streams + 32 * (size_t)(k & 1) + 4 * (size_t)(lane & 7)) * 4);
                const uint32_t tr_off = (uint32_t)(((4 * (lane & 7)) * kTileStride + (lane >> 3)) * 4);
                uint32_t row_cur = lds_addr(tile_b + lane * kTileStride), row_prev = lds_addr(my);
                uint32_t tr_cur = lds_addr(tile_b) + tr_off, tr_prev = lds_addr(tile) + tr_off;
                const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(a.symbols + s0);
                const uint64_t store_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                            (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
                // (64-byte pieces of rows of n_streams symbols: line-aligned iff the rows are)
                const uint32_t nt_loop = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(((a.n_streams * 4) % 128 == 0 && (sb & 127) == 0) ? 1 : 0));
                const uint32_t n_loop = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(n_full - 1));
                const uint32_t tile_step = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(kTileSyms * a.n_streams * 4));
                if (nt_loop)
                    ans_decode_tiles_loop_sm<false>(lo, hi, L.in.rd, L.in.lo_issued, row_cur, row_prev, tr_cur, tr_prev, lut_addr, qmask, (uint32_t)P,
                                                    kDecRingMask, words_base, store_base, n_loop, L.in.shift - 1u, lane_addr, lds_addr(dump),
                                                    (uint32_t)w_off, tile_step, goff);
                else
                    ans_decode_tiles_loop_sm<true>(lo, hi, L.in.rd, L.in.lo_issued, row_cur, row_prev, tr_cur, tr_prev, lut_addr, qmask, (uint32_t)P,
                                                   kDecRingMask, words_base, store_base, n_loop, L.in.shift - 1u, lane_addr, lds_addr(dump),
                                                   (uint32_t)w_off, tile_step, goff);
                wave_lds_fence();
                tile_store_sm(a.symbols, a.n_streams, s0, (n_full - 1) * kTileSyms, lane, ((n_full - 1) & 1) ? tile_b : tile);
                wave_lds_fence();
                L.state = ((uint64_t)hi << 32) | lo;
                t_done = n_full * kTileSyms;
            }
        }
        for (size_t t = t_done; t < N; ++t) {
            const uint32_t idx = next_index();
            if (active) col[t * a.n_streams] = a.min_symbol + (int32_t)idx;
            if (--countdown == 0) { countdown = 4 * G; L.in.advance_window(); }
        }
    } else {
        int32_t* row = a.symbols + (active ? s : 0) * N;
        const size_t n_full = N / kTileSyms;
        int32_t* my = tile + lane * kTileStride;
        bool all_done = false;              // the main-loop statement path also decodes the rows' ragged ends
        if constexpr (TILE_ASM) {
            // the ring address is formed with v_and_or: this wave's ring must be aligned to its size
            if ((lds_addr(ring) & (uint32_t)(kWaveRingWords * 4 - 1)) != 0) __builtin_trap();
            uint32_t lo = (uint32_t)L.state, hi = (uint32_t)(L.state >> 32);
            const uint32_t qmask = (1u << P) - 1u;
            const uint32_t lut_addr = lds_addr(lut.cp), lane_addr = lds_addr(ring + lane);
            int32_t* tile_b = tile + (kBlock / kWave) * kTileWords;
            uint32_t* dump = reinterpret_cast<uint32_t*>(smem + kRingBytes + lds_off + 2 * (size_t)(kBlock / kWave) * kTileWords * 4) +
                             wave_in_block * (4 * kWave) + lane;
            // Settle every load of the prologue here: otherwise the compiler's wait-count bookkeeping merges
            // "state still loading" (loop entry) with "window chunks in flight" (back edge) into a vmcnt(0) at the
            // top of every tile, which would expose the HBM latency of the window loads once per tile.
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)

            // The main-loop statement addresses HBM as uniform base + 32-bit lane offset: it needs a full wave, at
            // least two tiles, rows of < 2^24 symbols and this wave's compressed words within 2 GiB of a.words.
            const unsigned char* words_base = reinterpret_cast<const unsigned char*>(reinterpret_cast<uintptr_t>(a.words) & ~(uintptr_t)15);
            const uint64_t w_off = (uint64_t)(reinterpret_cast<const unsigned char*>(L.in.base16) - words_base);
            const bool off_ok = w_off + 4ull * ((uint64_t)L.in.rd + 8) < 0x80000000ull;
            const bool use_loop = (s0 + kWave <= a.n_streams) && N >= 4 * kTileSyms && N < (1u << 24) && !__any(!off_ok);
            size_t tb = 0;
            if (use_loop) {
                // ---- rows of any length and alignment: every lane first decodes the `pre` symbols in front of its row's next
                // cache-line boundary (0 for rows that start on one), so that its tiles -- and the 128-byte segments the
                // tile stores write -- are whole cache lines (row_skew above) ----
                const uint32_t pre = row_skew(a.symbols, s, N);
                uint32_t max_pre = pre;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) max_pre = max(max_pre, (uint32_t)__shfl_xor((int)max_pre, d));
                max_pre = (uint32_t)__builtin_amdgcn_readfirstlane((int)max_pre);
                if (max_pre) {
                    L.state = ((uint64_t)hi << 32) | lo;
                    for (uint32_t j = 0; j < max_pre; ++j) {
                        if (j < pre) row[j] = a.min_symbol + (int32_t)next_index();
                        L.in.advance_window();
                    }
                    L.in.land_pending();
                    L.in.refill_blocking();
                    wave_lds_fence();
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                    lo = (uint32_t)L.state; hi = (uint32_t)(L.state >> 32);
                }
                const size_t n_t = (N - max_pre) / kTileSyms;          // whole tiles every lane has (>= 3)
                ans_decode_tile32(lo, hi, L.in.rd, lut_addr, qmask, (uint32_t)P, lds_addr(my), L.in.shift - 1u, lane_addr, kDecRingMask);
                L.in.refill_blocking();
                wave_lds_fence();
                __builtin_amdgcn_s_waitcnt(0x0F70);
                uint32_t goff[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const size_t R = (size_t)(lane >> 3) + 8 * k;
                    goff[k] = (uint32_t)((R * N + row_skew(a.symbols, s0 + R, N) + 4 * (size_t)(lane & 7)) * 4);
                }
                const uint32_t tr_off = (uint32_t)(((lane >> 3) * kTileStride + 4 * (lane & 7)) * 4);
                // current = B (tile 1), previous = A (tile 0)
                uint32_t row_cur = lds_addr(tile_b + lane * kTileStride), row_prev = lds_addr(my);
                uint32_t tr_cur = lds_addr(tile_b) + tr_off, tr_prev = lds_addr(tile) + tr_off;
                // wave-uniform store base in SGPRs (s0 is derived from threadIdx, so the compiler cannot know)
                const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(a.symbols + s0 * N);
                // (readfirstlane returns int: go through uint32_t or the low half sign-extends into the high one)
                const uint64_t store_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                            (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
                // (every stored segment is a whole cache line now: the streaming form of the stores in every case)
                ans_decode_tiles_loop<false>(lo, hi, L.in.rd, L.in.lo_issued, row_cur, row_prev, tr_cur, tr_prev, lut_addr, qmask, (uint32_t)P,
                                             kDecRingMask, words_base, store_base, (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(n_t - 1)),
                                             L.in.shift - 1u, lane_addr, lds_addr(dump), (uint32_t)w_off, goff);
                // the last tile is still in LDS (buffer A if it has an even index)
                wave_lds_fence();
                tile_store_skewed(a.symbols, N, s0, (n_t - 1) * kTileSyms, lane, ((n_t - 1) & 1) ? tile_b : tile);
                wave_lds_fence();
                // ---- and what is left of each row behind its last whole tile (fewer than 64 symbols) ----
                L.state = ((uint64_t)hi << 32) | lo;
                const uint32_t done = pre + (uint32_t)(n_t * kTileSyms), rest = (uint32_t)N - done;
                const uint32_t max_rest = (uint32_t)N - (uint32_t)(n_t * kTileSyms);          // (a lane with pre == 0 exists or not: an upper bound)
                if (max_rest) {
                    L.in.refill_blocking();
                    wave_lds_fence();
                    for (uint32_t j = 0; j < max_rest; ++j) {
                        if (j < rest) row[done + j] = a.min_symbol + (int32_t)next_index();
                        L.in.advance_window();
                    }
                }
                lo = (uint32_t)L.state; hi = (uint32_t)(L.state >> 32);
                tb = n_full;
                all_done = true;
            }
            for (; tb < n_full; ++tb) {
                ans_decode_tile32(lo, hi, L.in.rd, lut_addr, qmask, (uint32_t)P, lds_addr(my), L.in.shift - 1u, lane_addr, kDecRingMask);
                L.in.template advance_window_fixed<kTileAsmChunks>(dump);
                wave_lds_fence();
                tile_store<VEC>(a.symbols, a.n_streams, N, s0, tb * kTileSyms, lane, tile);
                wave_lds_fence();
            }
            L.state = ((uint64_t)hi << 32) | lo;
        } else {
            for (size_t tb = 0; tb < n_full; ++tb) {
#pragma unroll
                for (int j = 0; j < kTileSyms / 4; ++j) {
                    int4 v;
                    v.x = a.min_symbol + (int32_t)next_index(); v.y = a.min_symbol + (int32_t)next_index();
                    v.z = a.min_symbol + (int32_t)next_index(); v.w = a.min_symbol + (int32_t)next_index();
                    *reinterpret_cast<int4*>(my + 4 * j) = v;
                    if ((j + 1) % G == 0) L.in.advance_window();   // static schedule
                }
                wave_lds_fence();
                tile_store<VEC>(a.symbols, a.n_streams, N, s0, tb * kTileSyms, lane, tile);
                wave_lds_fence();
            }
        }
        for (size_t t = all_done ? N : n_full * kTileSyms; t < N; ++t) {
            const uint32_t idx = next_index();
            if (active) row[t] = a.min_symbol + (int32_t)idx;
            L.in.advance_window();
        }
    }

    if (!active) return;
    a.status[s] = ws.bad ? (int32_t)CST_STREAM_INVALID_DATA : L.status;
    if (raw) {
        a.state[s] = (uint64_t)L.state;
        if (a.n_words_out) a.n_words_out[s] = L.in.rd;
    }
}

// small-footprint kernels for batches of more than one wave per SIMD (cst_ans_small.hip)
// the rest of the reference's (Word, State, PRECISION) grid, compiler-scheduled (cst_ans_generic.hip)
bool generic_config(cst_coder_config c);
cst_status ans_encode_generic(const cst_model* m, cst_coder_config cfg, const int32_t* d_symbols, size_t n_streams, size_t n_per_stream,
                              cst_layout layout, uint32_t* d_words, size_t stride_words, uint32_t* d_n_words, uint64_t* d_state,
                              int32_t* d_status, uint32_t flags, hipStream_t hs);
cst_status ans_decode_generic(const cst_model* m, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_offsets, size_t stride_words,
                              size_t words_capacity, const uint32_t* d_n_words, int32_t* d_symbols, size_t n_streams, size_t n_per_stream,
                              cst_layout layout, uint64_t* d_state, uint32_t* d_n_words_out, int32_t* d_status, uint32_t flags, hipStream_t hs);
// producer / consumer waves for batches of at most one wave of streams per SIMD (cst_ans_pc.hip)
bool pc_encode_usable(const AnsEncodeArgs& a, cst_coder_config cfg, cst_layout layout, int device_cus);
cst_status ans_encode_pc(const AnsEncodeArgs& a, hipStream_t hs);
// int8 symbol matrices inside the loops (cst_ans_n8.hip)
bool pc_n8_encode_usable(const AnsEncodeArgs& a, cst_coder_config cfg, cst_layout layout);
bool pc_n16_encode_usable(const AnsEncodeArgs& a, cst_coder_config cfg, cst_layout layout);
bool pc_n16_encode_ckpt_usable(const AnsEncodeArgs& a, cst_coder_config cfg, cst_layout layout, size_t interval);
cst_status ans_encode_pc_n16(const AnsEncodeArgs& a, size_t interval, uint32_t* d_ckpt_pos, uint64_t* d_ckpt_state, hipStream_t hs);   // interval 0: no jump points
bool pc_n8_encode_ckpt_usable(const AnsEncodeArgs& a, cst_coder_config cfg, cst_layout layout, size_t interval);
cst_status ans_encode_pc_n8_ckpt(const AnsEncodeArgs& a, size_t interval, uint32_t* d_ckpt_pos, uint64_t* d_ckpt_state, hipStream_t hs);
bool pc_encode_ckpt_usable(const AnsEncodeArgs& a, cst_coder_config cfg, cst_layout layout, size_t interval);
cst_status ans_encode_pc_ckpt(const AnsEncodeArgs& a, size_t interval, uint32_t* d_ckpt_pos, uint64_t* d_ckpt_state, hipStream_t hs);
cst_status ans_encode_pc_n8(const AnsEncodeArgs& a, hipStream_t hs);
bool n8_decode_usable(const AnsDecodeArgs& a, cst_coder_config cfg, cst_layout layout, int symbol_bytes);
cst_status ans_decode_n8(const AnsDecodeArgs& a, int symbol_bytes, hipStream_t hs);
bool n8_decode_small(const AnsDecodeArgs& a, int device_cus);
cst_status ans_decode_small_n8(const AnsDecodeArgs& a, int symbol_bytes, hipStream_t hs);
// lane-quad word loads for the P <= 12 decoder (cst_ans_dq.hip)
bool dq_decode_usable(const AnsDecodeArgs& a, cst_coder_config cfg, cst_layout layout);
cst_status ans_decode_dq(const AnsDecodeArgs& a, hipStream_t hs);
bool small_encode_usable(const AnsEncodeArgs& a, cst_coder_config cfg, cst_layout layout, int device_cus);
bool small_decode_usable(const AnsDecodeArgs& a, cst_coder_config cfg, cst_layout layout, int device_cus);
cst_status ans_encode_small(const AnsEncodeArgs& a, hipStream_t hs);
cst_status ans_decode_small(const AnsDecodeArgs& a, hipStream_t hs);
// the hand-scheduled decoder for 12 < P <= 24 (cst_ans_b16.hip)
bool wide_encode_usable(const AnsEncodeArgs& a, cst_coder_config cfg, cst_layout layout);
cst_status ans_encode_wide(const AnsEncodeArgs& a, cst_layout layout, hipStream_t hs);
bool b16_decode_usable(const AnsDecodeArgs& a, cst_coder_config cfg, cst_layout layout);
bool b16_narrow_decode_usable(const AnsDecodeArgs& a, cst_coder_config cfg, cst_layout layout, int symbol_bytes);      // int8 / int16 matrices at 12 < P <= 24
cst_status ans_decode_b16_narrow(const AnsDecodeArgs& a, int symbol_bytes, hipStream_t hs);
bool b16_small_decode_usable(const AnsDecodeArgs& a, cst_coder_config cfg, cst_layout layout, int symbol_bytes, int device_cus);   // ... two waves per SIMD (int32 too)
cst_status ans_decode_b16_small(const AnsDecodeArgs& a, int symbol_bytes, hipStream_t hs);
cst_status ans_decode_b16(const AnsDecodeArgs& a, cst_layout layout, hipStream_t hs);
// the hand-scheduled decoder of the (16,32) preset (cst_ans_w16.hip)
bool w16_encode_usable(const AnsEncodeArgs& a, cst_coder_config cfg, cst_layout layout);
cst_status ans_encode_w16(const AnsEncodeArgs& a, cst_layout layout, hipStream_t hs);
bool w16_decode_usable(const AnsDecodeArgs& a, cst_coder_config cfg, cst_layout layout);
cst_status ans_decode_w16(const AnsDecodeArgs& a, cst_layout layout, hipStream_t hs);

} // namespace cst
