// cst_range_fast.hip -- the hand-scheduled (32,64) range coder kernels (BASELINE config C4), 8 <= P <= 24, both symbol layouts.
//
// Encoder: the lazy carry of RangeEncoder::encode_symbol (src/stream/queue.rs:612-705, EncoderSituation :126-142) in its
// "held word" form.  The reference holds back the words of an Inverted situation until it knows whether a carry
// reaches them (first + 1, then zeros -- or first, then 0xffffffff): that IS addition with carry on the number the
// emitted words spell, a carry out of `lower + scale * c` can only happen while words are held back (in the Normal
// situation lower + range does not wrap), and a word entering an Inverted run is never 0xffffffff.  So a lane keeps the
// most recent word in a register, adds carries to it and hands it to the ring when the next word is produced; no
// situation is tracked.  A carry that must travel further than the held word (a run of two or more held words) is
// handled by RangeEncHeld::carry_back on words in the ring or already in HBM; the asm main loop only flags it and the
// wave repeats its streams with the C++ step.  scripts/gen_range_encode_loop.py has the instruction-level account.
#include <type_traits>

#include "cst_range_kernels.hpp"

namespace cst {

struct CumProb { uint32_t c, p; };

struct RangeEncHeld {
    uint64_t lower, range;
    uint32_t lw;          // the held word: stream position out.wr, not yet final
    uint32_t bad;
    bool owner;           // false for the lanes of a partial wave that merely repeat its last stream: they must not
                          // read-modify-write words that have left for HBM
    RingWriter<> out;     // out.wr = final words (-1 until the first word exists)

    __device__ __forceinline__ void init(uint32_t* slab, uint32_t capacity, uint32_t* wave_ring, int lane_) {
        out.init(slab, capacity, wave_ring, lane_);
        out.wr = 0xffffffffu;
        lower = 0; range = ~0ull; lw = 0; bad = 0;   // RangeCoderState::default, queue.rs:96-104
        owner = true;
    }

    // + 1 on the number spelled by the final words (the held word overflowed)
    __device__ __noinline__ void carry_back() {
        for (int64_t i = (int64_t)(int32_t)out.wr - 1; i >= 0; --i) {
            const uint32_t pos = (uint32_t)i + out.shift;
            uint32_t w;
            if (pos >= out.flushed) {
                w = *out.slot(pos) + 1u;
                *out.slot(pos) = w;
            } else {
                if ((uint32_t)i >= out.cap) break;                       // never stored: the stream ends as CST_STREAM_CAPACITY
                if (!owner) break;                                       // (everything older has left too, and the owner's update serves every copy)
                __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): the lane's own store of this word has landed
                w = __hip_atomic_load(out.base16 + pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
                __hip_atomic_store(out.base16 + pos, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (w != 0) break;
        }
    }

    __device__ __forceinline__ void step(uint32_t c, uint32_t p, int P) {
        const uint64_t scale = range >> P;
        uint64_t nr = scale * p;
        uint64_t nl = lower + scale * c;
        if (__builtin_expect(nl < lower, 0)) {
            if (++lw == 0) carry_back();
        }
        const bool renorm = nr < (1ull << 32);
        out.push(lw, renorm ? 1u : 0u);
        if (renorm) { lw = (uint32_t)(nl >> 32); nl <<= 32; nr <<= 32; }
        lower = nl; range = nr;
    }

    __device__ __forceinline__ void flush() {
        if ((int32_t)out.wr >= 4) out.flush_chunks();
    }

    // seal_words / iter_seal (queue.rs:458-523)
    __device__ __forceinline__ int32_t finish(uint32_t n_symbols, uint32_t& n_words_out) {
        if (range != ~0ull) {
            const uint64_t point = lower + 0xffffffffull;
            if (point < lower) {
                if (++lw == 0) carry_back();
            }
            if ((int32_t)out.wr >= 0) out.push(lw, 1u); else out.wr = 0;
            out.drain();
            const uint32_t point_word = (uint32_t)(point >> 32);
            const uint32_t upper_word = (uint32_t)((lower + range) >> 32);
            out.append_direct(point_word);
            if (upper_word == point_word) out.append_direct(0u);
        } else {
            out.wr = 0;
        }
        n_words_out = out.wr;
        if (bad >= n_symbols) return CST_STREAM_IMPOSSIBLE_SYMBOL;
        if (out.wr > out.cap) return CST_STREAM_CAPACITY;
        return CST_STREAM_OK;
    }
};

template <int FLUSHES, bool SM>
__device__ __forceinline__ void range_encode_tiles_loop(uint32_t& lo0, uint32_t& lo1, uint32_t& rg0, uint32_t& rg1, uint32_t& lw,
                                                        uint32_t& wr, uint32_t& flushed, int32_t& smin, int32_t& smax, uint32_t& slow,
                                                        const uint32_t (&tile_row_addr)[2], const uint32_t (&tile_tr_addr)[2],
                                                        uint32_t ring_lane_addr, uint32_t cap, uint32_t slab_off,
                                                        uint32_t table_addr_biased, uint32_t P, const void* words_base,
                                                        uint64_t symbols_base, uint32_t n_tiles, [[maybe_unused]] uint32_t tile_step_bytes,
                                                        const uint32_t (&goff)[8]) {
    if constexpr (FLUSHES == 1 && !SM) {
#include "cst_range_encode_loop.inc"
    } else if constexpr (FLUSHES == 1) {
#include "cst_range_encode_loop_sm.inc"
    } else if constexpr (!SM) {
#include "cst_range_encode_loop_2f.inc"
    } else {
#include "cst_range_encode_loop_2f_sm.inc"
    }
}

constexpr size_t kFastRingBytes = (size_t)(kBlock / kWave) * kRingWords * 4;
constexpr size_t kFastTileBytes = (size_t)(kBlock / kWave) * kWave * kTileStride * 4;

// SM: symbols[t][stream] (CST_LAYOUT_SYMBOL_MAJOR); the launch makes sure that every wave is full and that rows are 16-byte aligned
template <int FLUSHES, bool SM>
__global__ __launch_bounds__(kBlock) void range_encode_fast_kernel(const RangeEncodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = threadIdx.x >> 6;
    const size_t table_bytes = (((size_t)a.n_symbols * sizeof(CumProb)) + 15) & ~(size_t)15;
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem) + wave_in_block * kRingWords;
    CumProb* table = reinterpret_cast<CumProb*>(smem + kFastRingBytes);
    int32_t* tile = reinterpret_cast<int32_t*>(smem + kFastRingBytes + table_bytes) + wave_in_block * (kWave * kTileStride);
    if ((lds_addr(ring) & (kRingWords * 4u - 1u)) != 0) __builtin_trap();   // the ring address is formed with v_and_or
    for (int i = threadIdx.x; i < a.n_symbols; i += blockDim.x) table[i] = CumProb{a.enc[i].c, a.enc[i].p};
    __syncthreads();

    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t s0 = wave * kWave;
    if (s0 >= a.n_streams) return;
    const size_t s = s0 + lane;
    const bool active = s < a.n_streams;
    const size_t N = a.n_per_stream;
    const int P = a.precision;
    const uint32_t nsym = (uint32_t)a.n_symbols;
    const size_t n_full = N / kTileSyms;
    // The lanes of a partial wave beyond its last stream REPEAT that stream (same symbols, same slab, same words): the wave
    // then runs the main-loop statement like a full one, and only the per-stream results are written by the owner alone.
    const size_t se = active ? s : a.n_streams - 1;
    uint32_t* slab = a.words + se * a.stride_words;
    const uint32_t cap = (uint32_t)(a.stride_words > 0xffffffffull ? 0xffffffffull : a.stride_words);

    RangeEncHeld L;
    L.init(slab, cap, ring, lane);
    L.owner = active;
    bool done = false, all_done = false;
    {
        const uint64_t slab_off = (uint64_t)(reinterpret_cast<const unsigned char*>(L.out.base16) - reinterpret_cast<const unsigned char*>(a.words));
        const bool ok = slab_off + 4ull * cap < 0x100000000ull && (reinterpret_cast<uintptr_t>(L.out.base16) & 63) == 0 &&
                        (cap & 15u) == 0 && L.out.shift == 0;
        if (!SM && N >= 4 * kTileSyms && N < (1u << 24) && !__any(!ok)) {
            // Rows of any length and alignment (row_skew, cst_ans_kernels.hpp; as in the ANS encoders): lane l's tiles start
            // row_skew() symbols into its row, so that every 128-byte segment a tile load reads is one whole cache line; the
            // symbols in front of the first and behind the last whole tile go through LDS in bulk reads.
            const size_t last_row = min((size_t)kWave, a.n_streams - s0) - 1;
            const int32_t* in_row = a.symbols + se * N;
            int32_t* my = tile + lane * kTileStride;
            auto ragged = [&](size_t p0, uint32_t cnt) {
                code_ragged<true>(in_row, p0, cnt, my, a.min_symbol,
                                  [&](int32_t v) { const CumProb e = table[enc_index(v, a.min_symbol, nsym, L.bad)]; L.step(e.c, e.p, P); },
                                  [&]() { L.flush(); });
            };
            const uint32_t pre = row_skew(a.symbols, se, N);
            uint32_t max_pre = pre;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) max_pre = max(max_pre, (uint32_t)__shfl_xor((int)max_pre, d));
            max_pre = (uint32_t)__builtin_amdgcn_readfirstlane((int)max_pre);
            const size_t n_t = (N - max_pre) / kTileSyms;           // whole tiles every lane has (>= 3)
            ragged(0, pre);
            wave_lds_fence();
            uint32_t goff[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const size_t R = min((size_t)(lane >> 3) + 8 * k, last_row);
                goff[k] = (uint32_t)((R * N + row_skew(a.symbols, s0 + R, N) + 4 * (size_t)(lane & 7)) * 4);
            }
            const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(a.symbols + s0 * N);
            const uint64_t symbols_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                          (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
            const uint32_t tr_off = (uint32_t)(((lane >> 3) * kTileStride + 4 * (lane & 7)) * 4);
            int32_t* tile_b = tile + (kBlock / kWave) * (kWave * kTileStride);
            const uint32_t row_addr[2] = {lds_addr(tile + lane * kTileStride), lds_addr(tile_b + lane * kTileStride)};
            const uint32_t tr_addr[2] = {lds_addr(tile) + tr_off, lds_addr(tile_b) + tr_off};
            // the statement continues from the state the skew symbols left
            uint32_t lo0 = (uint32_t)L.lower, lo1 = (uint32_t)(L.lower >> 32), rg0 = (uint32_t)L.range, rg1 = (uint32_t)(L.range >> 32);
            uint32_t lw = L.lw, wr = L.out.wr, flushed = L.out.flushed, slow = 0;
            int32_t smin = a.min_symbol, smax = a.min_symbol;
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the statement keeps its own book from here
            range_encode_tiles_loop<FLUSHES, SM>(lo0, lo1, rg0, rg1, lw, wr, flushed, smin, smax, slow, row_addr, tr_addr, L.out.lane_addr, cap,
                                                 (uint32_t)slab_off, lds_addr(table) - 8u * (uint32_t)a.min_symbol, (uint32_t)P, a.words,
                                                 symbols_base, (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)n_t),
                                                 (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(kTileSyms * a.n_streams * 4)), goff);
            if (__builtin_amdgcn_readfirstlane(slow) == 0) {
                L.lower = ((uint64_t)lo1 << 32) | lo0; L.range = ((uint64_t)rg1 << 32) | rg0; L.lw = lw;
                L.out.wr = wr; L.out.flushed = flushed;
                L.bad = max(L.bad, max((uint32_t)smax - (uint32_t)a.min_symbol, (uint32_t)smin - (uint32_t)a.min_symbol));
                const size_t top = pre + n_t * kTileSyms;           // fewer than 64 symbols are left behind the last whole tile
                const uint32_t n_top = (uint32_t)(N - top);
                ragged(top, min(n_top, (uint32_t)kTileSyms));
                ragged(top + kTileSyms, n_top > (uint32_t)kTileSyms ? n_top - (uint32_t)kTileSyms : 0u);
                done = all_done = true;
            } else {
                // (rare: a carry had to travel) the wave's streams again from their first symbol, with the C++ step
                L.init(slab, cap, ring, lane);
                L.owner = active;
                wave_lds_fence();
            }
        } else if (n_full > 0 && N < (1u << 24) && !__any(!ok)) {
            const size_t last_row = min((size_t)kWave, a.n_streams - s0) - 1;
            uint32_t goff[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                goff[k] = SM ? (uint32_t)((((size_t)(lane >> 2) + 16 * (k & 1)) * a.n_streams + 16 * (size_t)(k >> 1) + 4 * (size_t)(lane & 3)) * 4)
                             : (uint32_t)((min((size_t)(lane >> 3) + 8 * k, last_row) * N + 4 * (size_t)(lane & 7)) * 4);
            const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(SM ? a.symbols + s0 : a.symbols + s0 * N);
            const uint64_t symbols_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                          (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
            // where a lane's loaded registers go in the tile (tile[stream][t], stride kTileStride): see the generators
            const uint32_t tr_off = SM ? (uint32_t)(((4 * (lane & 3)) * kTileStride + (lane >> 2)) * 4)
                                       : (uint32_t)(((lane >> 3) * kTileStride + 4 * (lane & 7)) * 4);
            int32_t* tile_b = tile + (kBlock / kWave) * (kWave * kTileStride);
            const uint32_t row_addr[2] = {lds_addr(tile + lane * kTileStride), lds_addr(tile_b + lane * kTileStride)};
            const uint32_t tr_addr[2] = {lds_addr(tile) + tr_off, lds_addr(tile_b) + tr_off};
            uint32_t lo0 = 0, lo1 = 0, rg0 = 0xffffffffu, rg1 = 0xffffffffu, lw = 0, wr = 0xffffffffu, flushed = 0, slow = 0;
            int32_t smin = a.min_symbol, smax = a.min_symbol;
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the statement keeps its own book from here
            range_encode_tiles_loop<FLUSHES, SM>(lo0, lo1, rg0, rg1, lw, wr, flushed, smin, smax, slow, row_addr, tr_addr, L.out.lane_addr, cap,
                                                 (uint32_t)slab_off, lds_addr(table) - 8u * (uint32_t)a.min_symbol, (uint32_t)P, a.words,
                                                 symbols_base, (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)n_full),
                                                 (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(kTileSyms * a.n_streams * 4)), goff);
            if (__builtin_amdgcn_readfirstlane(slow) == 0) {
                L.lower = ((uint64_t)lo1 << 32) | lo0; L.range = ((uint64_t)rg1 << 32) | rg0; L.lw = lw;
                L.out.wr = wr; L.out.flushed = flushed;
                // a symbol below min_symbol wraps to a huge index
                L.bad = max((uint32_t)smax - (uint32_t)a.min_symbol, (uint32_t)smin - (uint32_t)a.min_symbol);
                done = true;
            }
        }
    }
    if (!done) {
        // partial waves, odd slabs, and streams whose carry had to travel: tile by tile with the C++ step
        if constexpr (SM) {
            // (rare: a carry had to travel) lane-coalesced reads, one symbol row at a time
            const int32_t* col = a.symbols + se;
            for (size_t t = 0; t < n_full * kTileSyms; ++t) {
                const CumProb e = table[enc_index(col[t * a.n_streams], a.min_symbol, nsym, L.bad)];
                L.step(e.c, e.p, P);
                if ((t & 7) == 7) L.flush();
            }
        } else {
            int32_t r[kTileSyms];
            for (size_t tb = 0; tb < n_full; ++tb) {
                tile_fetch<true>(a.symbols, a.n_streams, N, s0, tb * kTileSyms, lane, r);
                wave_lds_fence();
                tile_to_lds<true>(tile, lane, r);
                wave_lds_fence();
                const int32_t* my = tile + min((size_t)lane, a.n_streams - 1 - s0) * kTileStride;   // (repeating lanes read the last stream's row)
#pragma unroll 8
                for (int j = 0; j < kTileSyms; ++j) {
                    const CumProb e = table[enc_index(my[j], a.min_symbol, nsym, L.bad)];
                    L.step(e.c, e.p, P);
                    if ((j & 7) == 7) L.flush();
                }
            }
        }
    }
    const int32_t* row = SM ? a.symbols + se : a.symbols + se * N;
    const size_t row_step = SM ? a.n_streams : 1;
    for (size_t t = all_done ? N : n_full * kTileSyms; t < N; ++t) {
        const CumProb e = table[enc_index(row[t * row_step], a.min_symbol, nsym, L.bad)];
        L.step(e.c, e.p, P);
        L.flush();
    }
    uint32_t n_words = 0;
    const int32_t status = L.finish(nsym, n_words);
    if (!active) return;
    a.status[s] = status;
    a.n_words[s] = (status == CST_STREAM_OK) ? n_words : 0u;
}

// ------------------------------------------------------------------------------------------------
// Decoder, P <= 12 (scripts/gen_range_decode_loop.py): state x = point - lower, f64 quotient estimate checked through
// the symbol it selects.
// ------------------------------------------------------------------------------------------------
constexpr int kRdSlots = kDecRingSlots, kRdAhead = kDecAhead;
constexpr size_t kRdRingBytes = (size_t)(kBlock / kWave) * kRdSlots * kWave * 4;

template <bool ENDS, bool B16, bool SM>
__device__ __forceinline__ void range_decode_tiles_loop(uint32_t& x0, uint32_t& x1, uint32_t& rg0, uint32_t& rg1, uint32_t& pos,
                                                        uint32_t& hi_issued, uint32_t& row_cur, uint32_t& row_prev, uint32_t& tr_cur,
                                                        uint32_t& tr_prev, uint32_t& tiles, uint32_t& ginc, uint32_t& bad,
                                                        uint32_t lut_addr, uint32_t qmax, uint32_t P, uint32_t ring_mask,
                                                        const void* words_base, uint32_t delta_hi, uint64_t store_base,
                                                        uint32_t lens, uint32_t endr, uint32_t ring_lane_addr, uint32_t dump_addr,
                                                        uint32_t words_off, uint32_t bucket_shift,
                                                        uint32_t cdf_addr, int32_t min_symbol, [[maybe_unused]] uint32_t tile_step_bytes,
                                                        [[maybe_unused]] uint32_t c_field_mask, [[maybe_unused]] uint32_t index_shift,
                                                        bool plain_stores) {
    if constexpr (B16 && ENDS && SM) {
    if (plain_stores) {          // rows that are not cache-line aligned: see scripts/gen_decode_loop.py (CST_STORE_MOD)
#define CST_STORE_MOD ""
#include "cst_range_decode_loop_b16_ends_sm.inc"
#undef CST_STORE_MOD
    } else {
#define CST_STORE_MOD "nt"
#include "cst_range_decode_loop_b16_ends_sm.inc"
#undef CST_STORE_MOD
    }
    } else if constexpr (B16 && ENDS) {
    if (plain_stores) {          // rows that are not cache-line aligned: see scripts/gen_decode_loop.py (CST_STORE_MOD)
#define CST_STORE_MOD ""
#include "cst_range_decode_loop_b16_ends.inc"
#undef CST_STORE_MOD
    } else {
#define CST_STORE_MOD "nt"
#include "cst_range_decode_loop_b16_ends.inc"
#undef CST_STORE_MOD
    }
    } else if constexpr (B16 && SM) {
    if (plain_stores) {          // rows that are not cache-line aligned: see scripts/gen_decode_loop.py (CST_STORE_MOD)
#define CST_STORE_MOD ""
#include "cst_range_decode_loop_b16_sm.inc"
#undef CST_STORE_MOD
    } else {
#define CST_STORE_MOD "nt"
#include "cst_range_decode_loop_b16_sm.inc"
#undef CST_STORE_MOD
    }
    } else if constexpr (B16) {
    if (plain_stores) {          // rows that are not cache-line aligned: see scripts/gen_decode_loop.py (CST_STORE_MOD)
#define CST_STORE_MOD ""
#include "cst_range_decode_loop_b16.inc"
#undef CST_STORE_MOD
    } else {
#define CST_STORE_MOD "nt"
#include "cst_range_decode_loop_b16.inc"
#undef CST_STORE_MOD
    }
    } else if constexpr (ENDS && SM) {
    if (plain_stores) {          // rows that are not cache-line aligned: see scripts/gen_decode_loop.py (CST_STORE_MOD)
#define CST_STORE_MOD ""
#include "cst_range_decode_loop_ends_sm.inc"
#undef CST_STORE_MOD
    } else {
#define CST_STORE_MOD "nt"
#include "cst_range_decode_loop_ends_sm.inc"
#undef CST_STORE_MOD
    }
    } else if constexpr (ENDS) {
    if (plain_stores) {          // rows that are not cache-line aligned: see scripts/gen_decode_loop.py (CST_STORE_MOD)
#define CST_STORE_MOD ""
#include "cst_range_decode_loop_ends.inc"
#undef CST_STORE_MOD
    } else {
#define CST_STORE_MOD "nt"
#include "cst_range_decode_loop_ends.inc"
#undef CST_STORE_MOD
    }
    } else if constexpr (SM) {
    if (plain_stores) {          // rows that are not cache-line aligned: see scripts/gen_decode_loop.py (CST_STORE_MOD)
#define CST_STORE_MOD ""
#include "cst_range_decode_loop_sm.inc"
#undef CST_STORE_MOD
    } else {
#define CST_STORE_MOD "nt"
#include "cst_range_decode_loop_sm.inc"
#undef CST_STORE_MOD
    }
    } else {
    if (plain_stores) {          // rows that are not cache-line aligned: see scripts/gen_decode_loop.py (CST_STORE_MOD)
#define CST_STORE_MOD ""
#include "cst_range_decode_loop.inc"
#undef CST_STORE_MOD
    } else {
#define CST_STORE_MOD "nt"
#include "cst_range_decode_loop.inc"
#undef CST_STORE_MOD
    }
    }
}

static size_t b16_tables_bytes(int n_symbols, int bucket_bits) {
    return ((((size_t)n_symbols + 1) * 4 + 15) & ~(size_t)15) + ((size_t)16 << bucket_bits) + kSubAreaBytes;
}

// B16: 12 < P <= 24, at most 256 symbols: the lookup is one 16-byte bucket entry (DecLut::b16) instead of the quantile table
// SM: symbols[t][stream] (CST_LAYOUT_SYMBOL_MAJOR); the launch makes sure that every wave is full and that rows are 16-byte aligned
template <bool B16, bool SM>
__global__ __launch_bounds__(kBlock) void range_decode_fast_kernel(const RangeDecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = threadIdx.x >> 6;
    const int P = a.precision;
    const size_t n_q = (size_t)1 << P;
    constexpr size_t kTileWords = (size_t)kWave * kTileStride;
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem) + wave_in_block * (kRdSlots * kWave);
    // LDS image: P <= 12 as for the ANS decoder (stage_tile_tables): cp[q] = c | p << 16 at +0, the decoded symbol at +16384;
    // B16: cdf, then the bucket entries (stage_decoder_tables)
    uint32_t* lut = reinterpret_cast<uint32_t*>(smem + kRdRingBytes);
    int32_t* symt = reinterpret_cast<int32_t*>(smem + kRdRingBytes + kTileSymOffset);
    DecLut blut{};
    const uint32_t* cdf = a.cdf;
    const uint16_t* bucket = a.bucket;
    size_t table_bytes = kTileLutBytes;
    if constexpr (B16) {
        table_bytes = stage_decoder_tables<kDecBucket, true, true>(smem + kRdRingBytes, P, a.dec_cp, a.dec_idx, a.cdf, a.bucket, a.bucket_bits,
                                                             a.n_symbols, blut, cdf, bucket);
    } else {
        for (size_t q = threadIdx.x; q < n_q; q += blockDim.x) {
            lut[q] = a.dec_cp[q];
            symt[q] = a.min_symbol + (int32_t)a.dec_idx[q];
        }
    }
    int32_t* tile = reinterpret_cast<int32_t*>(smem + kRdRingBytes + table_bytes) + wave_in_block * kTileWords;
    int32_t* tile_b = tile + (kBlock / kWave) * kTileWords;
    uint32_t* dump = reinterpret_cast<uint32_t*>(smem + kRdRingBytes + table_bytes + 2 * (size_t)(kBlock / kWave) * kTileWords * 4) +
                     wave_in_block * (4 * kWave) + lane;
    if ((lds_addr(ring) & (uint32_t)(kRdSlots * kWave * 4 - 1)) != 0) __builtin_trap();   // the ring address is formed with v_and_or
    __syncthreads();

    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t s0 = wave * kWave;
    if (s0 >= a.n_streams) return;
    const size_t s = s0 + lane;
    const bool active = s < a.n_streams;
    const size_t N = a.n_per_stream;
    const size_t n_full = N / kTileSyms;
    const int bucket_shift = P - a.bucket_bits;
    // the lanes of a partial wave beyond its last stream repeat that stream (see the encoder)
    const size_t se = active ? s : a.n_streams - 1;
    const WordSlice ws = word_slice(a.offsets, a.stride_words, a.n_words, se, a.words_capacity);
    const uint32_t* my_words = a.words + ws.off;
    const uint32_t my_len = ws.n;

    RangeDecLane<32, 64, kRdSlots, kRdAhead> L;
    L.init(my_words, my_len, ring, lane);
    L.in.prime();
    wave_lds_fence();

    // the exact step (queue.rs:968-1033); returns the symbol
    auto step = [&]() -> int32_t {
        const uint32_t q = L.peek_quantile(P);
        uint32_t c, p;
        int32_t sym;
        if constexpr (B16) {
            uint32_t idx;
            lookup_quantile<kDecBucket>(q, blut, cdf, bucket, bucket_shift, a.n_symbols, idx, c, p);
            sym = a.min_symbol + (int32_t)idx;
        } else {
            const uint32_t cp = lut[q];
            c = cp & 0xffffu; p = cp >> 16;
            sym = symt[q];
        }
        const uint32_t w = L.in.peek();
        L.in.pos += L.advance(c, p, P, w, L.in.pos < L.in.len) ? 1u : 0u;
        return sym;
    };

    size_t tb = 0;
    bool all_done = false;
    {
        const unsigned char* words_base = reinterpret_cast<const unsigned char*>(reinterpret_cast<uintptr_t>(a.words) & ~(uintptr_t)15);
        const uint64_t w_off = (uint64_t)(reinterpret_cast<const unsigned char*>(L.in.base16) - words_base);
        const bool off_ok = w_off + 4ull * ((uint64_t)my_len + 8) < 0x80000000ull;
        if (n_full > 0 && N < (1u << 24) && !__any(!off_ok)) {
            // Rows of any length and alignment (row_skew, cst_ans_kernels.hpp; as in the ANS decoders): every lane first decodes
            // the `pre` symbols in front of its row's next cache-line boundary, so that its tiles -- and the 128-byte segments the
            // tile stores write -- are whole cache lines.
            const bool skew = !SM && N >= 4 * kTileSyms;
            const size_t last_row = min((size_t)kWave, a.n_streams - s0) - 1;
            int32_t* out_row = a.symbols + se * (SM ? 0 : N);
            // `count` more symbols of every lane that has them, straight to HBM (the window is topped up every four)
            auto direct = [&](uint32_t first, uint32_t have, uint32_t count) {
                for (uint32_t j = 0; j < count; ++j) {
                    if (j < have) { const int32_t sym = step(); if (active) out_row[first + j] = sym; }
                    if ((j & 3) == 3) { L.in.fill_blocking(); wave_lds_fence(); }
                }
                L.in.fill_blocking();
                wave_lds_fence();
            };
            uint32_t pre = 0;
            size_t n_loop = n_full;
            if (skew) {
                pre = row_skew(a.symbols, se, N);
                uint32_t max_pre = pre;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) max_pre = max(max_pre, (uint32_t)__shfl_xor((int)max_pre, d));
                max_pre = (uint32_t)__builtin_amdgcn_readfirstlane((int)max_pre);
                if (max_pre) direct(0, pre, max_pre);
                n_loop = (N - max_pre) / kTileSyms;             // whole tiles every lane has (>= 3)
            }
            // the eight store offsets of a tile's pieces: the statements read them from the lane's row of their CURRENT buffer
            uint32_t goff[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if constexpr (SM) {
                    goff[k] = (uint32_t)((((size_t)(lane >> 3) + 8 * (k >> 1)) * a.n_streams + 32 * (size_t)(k & 1) + 4 * (size_t)(lane & 7)) * 4);
                } else {
                    const size_t R = min((size_t)(lane >> 3) + 8 * k, last_row);
                    goff[k] = (uint32_t)((R * N + (skew ? row_skew(a.symbols, s0 + R, N) : 0u) + 4 * (size_t)(lane & 7)) * 4);
                }
            }
            auto leave_offsets = [&](uint32_t row_cur_addr) {
                int32_t* cur = row_cur_addr == lds_addr(tile + lane * kTileStride) ? tile : tile_b;
#pragma unroll
                for (int k = 0; k < 8; ++k) cur[lane * kTileStride + k] = (int32_t)goff[k];
                wave_lds_fence();
            };
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the statements keep their own book from here
            const uint32_t shift = L.in.shift;
            // (the statements carry x = point - lower: after the skew symbols `lower` is no longer 0)
            const uint64_t x_in = (uint64_t)L.point - (uint64_t)L.lower;
            uint32_t x0 = (uint32_t)x_in, x1 = (uint32_t)(x_in >> 32), rg0 = (uint32_t)L.range, rg1 = (uint32_t)(L.range >> 32);
            uint32_t pos = L.in.pos + shift, hi_issued = L.in.hi_issued;
            const uint32_t lens = my_len + shift, endr = (lens + 3u) & ~3u;
            const uint32_t tr_off = SM ? (uint32_t)(((4 * (lane & 7)) * kTileStride + (lane >> 3)) * 4)
                                       : (uint32_t)(((lane >> 3) * kTileStride + 4 * (lane & 7)) * 4);
            uint32_t row_cur = lds_addr(tile + lane * kTileStride), row_prev = lds_addr(tile_b + lane * kTileStride);
            uint32_t tr_cur = lds_addr(tile) + tr_off, tr_prev = lds_addr(tile_b) + tr_off;
            uint32_t tiles = (uint32_t)n_loop, ginc = 0, bad = 0, bad2 = 0;
            const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(SM ? a.symbols + s0 : a.symbols + s0 * N);
            const uint32_t tile_step = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(SM ? kTileSyms * a.n_streams * 4 : kTileSyms * 4));
            const uint64_t store_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
            // rows that do not start on cache-line boundaries and are too short for the skew: plain tile stores
            // (scripts/gen_decode_loop.py, CST_STORE_MOD; symbol-major: rows of n_streams symbols, line-aligned iff the rows are)
            const bool plain_stores = __builtin_amdgcn_readfirstlane((int)((!skew && (((SM ? a.n_streams : N) * 4) % 128 != 0 || (sb & 127) != 0)) ? 1 : 0)) != 0;
            const uint32_t qmax = (1u << P) - 1u, ring_mask = (uint32_t)(kRdSlots - 1) << 8;
            // the estimate's error is 2^(P - 48.5): the bias on top of it (high word of the f64)
            const uint32_t delta_hi = P <= 16 ? 0x3e100000u : 0x3e900000u;   // 2^-30, 2^-22
            const uint32_t lut_addr = B16 ? lds_addr(blut.b16) : lds_addr(lut);
            const uint32_t cdf_addr = B16 ? lds_addr(cdf) : 0u;
            const uint32_t idx_shift = B16 ? (uint32_t)blut.idx_shift : 24u, idx_mask = (1u << idx_shift) - 1u;
            leave_offsets(row_cur);
            range_decode_tiles_loop<false, B16, SM>(x0, x1, rg0, rg1, pos, hi_issued, row_cur, row_prev, tr_cur, tr_prev, tiles, ginc, bad, lut_addr,
                                                    qmax, (uint32_t)P, ring_mask, words_base, delta_hi, store_base, lens, endr,
                                                    lds_addr(ring + lane), lds_addr(dump), (uint32_t)w_off, (uint32_t)__builtin_amdgcn_readfirstlane(bucket_shift),
                                                    cdf_addr, a.min_symbol, tile_step, idx_mask, idx_shift, plain_stores);
            tiles = (uint32_t)__builtin_amdgcn_readfirstlane(tiles);
            if (tiles > 0) {
                const uint32_t done = (uint32_t)n_loop - tiles;
                const uint64_t base2 = store_base + (done > 0 ? (uint64_t)(done - 1) * tile_step : 0);
                leave_offsets(row_cur);
                range_decode_tiles_loop<true, B16, SM>(x0, x1, rg0, rg1, pos, hi_issued, row_cur, row_prev, tr_cur, tr_prev, tiles, ginc, bad2,
                                                       lut_addr, qmax, (uint32_t)P, ring_mask, words_base, delta_hi, base2, lens, endr,
                                                       lds_addr(ring + lane), lds_addr(dump), (uint32_t)w_off, (uint32_t)__builtin_amdgcn_readfirstlane(bucket_shift),
                                                       cdf_addr, a.min_symbol, tile_step, idx_mask, idx_shift, plain_stores);
            }
            if (__builtin_amdgcn_readfirstlane(bad | bad2) == 0) {
                // the last tile is still in LDS (buffer A if it has an even index)
                wave_lds_fence();
                const int32_t* last = ((n_loop - 1) & 1) ? tile_b : tile;
                if constexpr (SM) tile_store_sm(a.symbols, a.n_streams, s0, (n_full - 1) * kTileSyms, lane, last);
                else if (!skew) tile_store<true>(a.symbols, a.n_streams, N, s0, (n_full - 1) * kTileSyms, lane, last);
                else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const size_t R = min((size_t)(lane >> 3) + 8 * k, last_row);
                        const int4 v = *reinterpret_cast<const int4*>(last + ((lane >> 3) + 8 * k) * kTileStride + 4 * (lane & 7));
                        v4i t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
                        __builtin_nontemporal_store(t, reinterpret_cast<v4i*>(a.symbols + (s0 + R) * N + row_skew(a.symbols, s0 + R, N) + (n_loop - 1) * kTileSyms + 4 * (lane & 7)));
                    }
                }
                wave_lds_fence();
                L.lower = 0; L.point = ((uint64_t)x1 << 32) | x0; L.range = ((uint64_t)rg1 << 32) | rg0;
                L.in.pos = pos - shift; L.in.hi_issued = hi_issued;
                tb = n_full;
                if (skew) {
                    // ... and what is left of each row behind its last whole tile (fewer than 64 symbols)
                    L.in.fill_blocking();
                    wave_lds_fence();
                    const uint32_t done = pre + (uint32_t)(n_loop * kTileSyms);
                    direct(done, (uint32_t)N - done, (uint32_t)N - (uint32_t)(n_loop * kTileSyms));
                    all_done = true;
                }
            } else {
                // a quantile estimate failed its check, or the data are invalid: the wave's streams again, exactly
                L.init(my_words, my_len, ring, lane);
                L.in.prime();
                wave_lds_fence();
            }
        }
    }
    int32_t* my = tile + lane * kTileStride;
    for (; tb < n_full; ++tb) {
#pragma unroll 1
        for (int j = 0; j < kTileSyms / 4; ++j) {
            int4 v;
            v.x = step(); v.y = step(); v.z = step(); v.w = step();
            *reinterpret_cast<int4*>(my + 4 * j) = v;
            L.in.advance_window();
        }
        wave_lds_fence();
        if constexpr (SM) tile_store_sm(a.symbols, a.n_streams, s0, tb * kTileSyms, lane, tile);
        else tile_store<true>(a.symbols, a.n_streams, N, s0, tb * kTileSyms, lane, tile);
        wave_lds_fence();
    }
    int32_t* row = SM ? a.symbols + (active ? s : 0) : a.symbols + (active ? s : 0) * N;
    const size_t row_step = SM ? a.n_streams : 1;
    for (size_t t = all_done ? N : n_full * kTileSyms; t < N; ++t) {
        const int32_t sym = step();
        if (active) row[t * row_step] = sym;
        L.in.advance_window();
    }
    if (!active) return;
    a.status[s] = ws.bad ? (int32_t)CST_STREAM_INVALID_DATA : L.status;
}

static size_t range_decode_fast_lds(const RangeDecodeArgs& a) {
    const size_t tables = a.precision <= 12 ? (size_t)kTileLutBytes : b16_tables_bytes(a.n_symbols, a.bucket_bits);
    return kRdRingBytes + tables + 2 * kFastTileBytes + (size_t)(kBlock / kWave) * 4 * kWave * 4;
}

// symbol-major batches take the same statements with another staging (scripts/gen_range_*_loop.py, SYMBOL_MAJOR): full waves,
// whole quads of streams per 16-byte piece, 32-bit offsets inside a tile of 32 symbol rows
static bool symbol_major_ok(size_t n_streams) { return n_streams % kWave == 0 && n_streams < (1u << 24); }

bool range_decode_fast_usable(const RangeDecodeArgs& a, cst_layout layout) {
    const bool table = a.precision >= 8 && a.precision <= 12 && a.dec_cp && a.dec_idx;
    const bool entries = a.precision > 12 && a.precision <= 24 && bucket16_usable(a.n_symbols, a.precision) && a.cdf && a.bucket;
    if (layout == CST_LAYOUT_SYMBOL_MAJOR && !symbol_major_ok(a.n_streams)) return false;
    return (table || entries) && !(a.flags & CST_FLAG_RAW_STATE) && (layout == CST_LAYOUT_SYMBOL_MAJOR || a.n_per_stream % 4 == 0) &&
           (reinterpret_cast<uintptr_t>(a.symbols) & 15) == 0 && range_decode_fast_lds(a) <= 160 * 1024;
}

cst_status range_decode_fast(const RangeDecodeArgs& a, cst_layout layout, hipStream_t hs) {
    const size_t lds = range_decode_fast_lds(a);
    const size_t blocks = (a.n_streams + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    auto go = [&](auto kernel) -> cst_status {
        CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kBlock), lds, hs, a);
        CST_HIP_TRY(hipGetLastError());
        return CST_OK;
    };
    if (layout == CST_LAYOUT_SYMBOL_MAJOR) return a.precision <= 12 ? go(range_decode_fast_kernel<false, true>) : go(range_decode_fast_kernel<true, true>);
    return a.precision <= 12 ? go(range_decode_fast_kernel<false, false>) : go(range_decode_fast_kernel<true, false>);
}

bool range_encode_fast_usable(const RangeEncodeArgs& a, cst_layout layout) {
    const size_t table_bytes = (((size_t)a.n_symbols * sizeof(CumProb)) + 15) & ~(size_t)15;
    if (layout == CST_LAYOUT_SYMBOL_MAJOR && !symbol_major_ok(a.n_streams)) return false;
    return a.precision >= 8 && a.precision <= 24 && !(a.flags & CST_FLAG_RAW_STATE) &&
           a.n_per_stream >= (size_t)kTileSyms && (layout == CST_LAYOUT_SYMBOL_MAJOR || a.n_per_stream % 4 == 0) && a.n_per_stream < (1u << 24) &&
           (reinterpret_cast<uintptr_t>(a.symbols) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.words) & 63) == 0 &&
           a.stride_words % 16 == 0 && a.n_streams * a.stride_words * 4 < 0x100000000ull &&
           kFastRingBytes + table_bytes + 2 * kFastTileBytes <= 160 * 1024;
}

cst_status range_encode_fast(const RangeEncodeArgs& a, cst_layout layout, hipStream_t hs) {
    const size_t table_bytes = (((size_t)a.n_symbols * sizeof(CumProb)) + 15) & ~(size_t)15;
    const size_t lds = kFastRingBytes + table_bytes + 2 * kFastTileBytes;
    const size_t blocks = (a.n_streams + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    auto go = [&](auto kernel) -> cst_status {
        CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kBlock), lds, hs, a);
        CST_HIP_TRY(hipGetLastError());
        return CST_OK;
    };
    if (layout == CST_LAYOUT_SYMBOL_MAJOR) return a.precision <= 16 ? go(range_encode_fast_kernel<1, true>) : go(range_encode_fast_kernel<2, true>);
    return a.precision <= 16 ? go(range_encode_fast_kernel<1, false>) : go(range_encode_fast_kernel<2, false>);
}


// ------------------------------------------------------------------------------------------------------------------
// Jump points (round 5): RangeEncoder::pos() / RangeDecoder::seek (queue.rs:172-196, 900-926; the reference's test :1333-1396).
// The encoder notes (words emitted including held-back ones, lower, range) in front of every chunk of `interval` symbols; the
// decoder runs every (stream, chunk) pair on its own lane -- k lanes per stream, EIGHT waves per workgroup, two per SIMD: a lone
// wave spends a third of the range decoder's cycles waiting for its table entry and cannot issue the ~46 VALU instructions of
// a step any faster than one per ~4.3 cycles (profiles/r04_sq_counters.md).  The words are those of the plain encoder.
// ------------------------------------------------------------------------------------------------------------------
template <int FLUSHES>
__device__ __forceinline__ void range_encode_tiles_loop_ck(uint32_t& lo0, uint32_t& lo1, uint32_t& rg0, uint32_t& rg1, uint32_t& lw,
                                                           uint32_t& wr, uint32_t& flushed, int32_t& smin, int32_t& smax, uint32_t& slow,
                                                           uint32_t& ck_index, const uint32_t (&tile_row_addr)[2],
                                                           const uint32_t (&tile_tr_addr)[2], uint32_t ring_lane_addr, uint32_t cap,
                                                           uint32_t slab_off, uint32_t table_addr_biased, uint32_t P, const void* words_base,
                                                           uint64_t symbols_base, uint32_t n_tiles, const void* ck_pos_base,
                                                           const void* ck_lower_base, const void* ck_range_base, uint32_t ck_tiles,
                                                           const uint32_t (&goff)[8]) {
    if constexpr (FLUSHES == 1) {
#include "cst_range_encode_loop_ck.inc"
    } else {
#include "cst_range_encode_loop_2f_ck.inc"
    }
}

// ... over an int8 symbol matrix (round 6, scripts/gen_range_encode_loop.py GEN_RANGE_N8): a tile is 32 bytes of a row
template <int FLUSHES>
__device__ __forceinline__ void range_encode_tiles_loop_ck_n8(uint32_t& lo0, uint32_t& lo1, uint32_t& rg0, uint32_t& rg1, uint32_t& lw,
                                                              uint32_t& wr, uint32_t& flushed, int32_t& smin, int32_t& smax, uint32_t& slow,
                                                              uint32_t& ck_index, const uint32_t (&tile_row_addr)[2],
                                                              const uint32_t (&tile_tr_addr)[2], uint32_t ring_lane_addr, uint32_t cap,
                                                              uint32_t slab_off, uint32_t table_addr_biased, uint32_t P, const void* words_base,
                                                              uint64_t symbols_base, uint32_t n_tiles, const void* ck_pos_base,
                                                              const void* ck_lower_base, const void* ck_range_base, uint32_t ck_tiles,
                                                              const uint32_t (&goff)[8]) {
    if constexpr (FLUSHES == 1) {
#include "cst_range_encode_loop_ck_n8.inc"
    } else {
#include "cst_range_encode_loop_2f_ck_n8.inc"
    }
}

// SB = bytes per symbol of the matrix behind a.symbols: 4 (int32) or 1 (int8: RangeEncodeArgs::symbols is then a const int8_t*)
template <int FLUSHES, int SB = 4>
__global__ __launch_bounds__(kBlock) void range_encode_ckpt_kernel(const RangeEncodeArgs a, const RangeCkptOut ck) {
    using SymT = typename std::conditional<SB == 1, int8_t, int32_t>::type;
    const SymT* symbols = reinterpret_cast<const SymT*>(a.symbols);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = threadIdx.x >> 6;
    const size_t table_bytes = (((size_t)a.n_symbols * sizeof(CumProb)) + 15) & ~(size_t)15;
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem) + wave_in_block * kRingWords;
    CumProb* table = reinterpret_cast<CumProb*>(smem + kFastRingBytes);
    int32_t* tile = reinterpret_cast<int32_t*>(smem + kFastRingBytes + table_bytes) + wave_in_block * (kWave * kTileStride);
    if ((lds_addr(ring) & (kRingWords * 4u - 1u)) != 0) __builtin_trap();
    for (int i = threadIdx.x; i < a.n_symbols; i += blockDim.x) table[i] = CumProb{a.enc[i].c, a.enc[i].p};
    __syncthreads();

    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t s0 = wave * kWave;
    if (s0 >= a.n_streams) return;
    const size_t s = s0 + lane;
    const bool active = s < a.n_streams;
    const size_t N = a.n_per_stream;
    const int P = a.precision;
    const uint32_t nsym = (uint32_t)a.n_symbols;
    const size_t se = active ? s : a.n_streams - 1;          // (lanes beyond the last stream repeat it: see range_encode_fast_kernel)
    uint32_t* slab = a.words + se * a.stride_words;
    const uint32_t cap = (uint32_t)(a.stride_words > 0xffffffffull ? 0xffffffffull : a.stride_words);

    RangeEncHeld L;
    L.init(slab, cap, ring, lane);
    L.owner = active;
    bool done = false;
    const uint64_t slab_off = (uint64_t)(reinterpret_cast<const unsigned char*>(L.out.base16) - reinterpret_cast<const unsigned char*>(a.words));
    const bool ok = slab_off + 4ull * cap < 0x100000000ull && (reinterpret_cast<uintptr_t>(L.out.base16) & 63) == 0 &&
                    (cap & 15u) == 0 && L.out.shift == 0;
    if (N >= (size_t)kTileSyms && N % kTileSyms == 0 && ck.interval % kTileSyms == 0 && N < (1u << 24) && a.n_streams * ck.n_chunks < (1u << 28) &&
        !__any(!ok)) {
        const size_t last_row = min((size_t)kWave, a.n_streams - s0) - 1;
        uint32_t goff[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) goff[k] = (uint32_t)((min((size_t)(lane >> 3) + 8 * k, last_row) * N + 4 * (size_t)(lane & 7)) * SB);
        const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(symbols + s0 * N);
        const uint64_t symbols_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                      (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
        const uint32_t tr_off = (uint32_t)(((lane >> 3) * kTileStride + 4 * (lane & 7)) * 4);
        int32_t* tile_b = tile + (kBlock / kWave) * (kWave * kTileStride);
        const uint32_t row_addr[2] = {lds_addr(tile + lane * kTileStride), lds_addr(tile_b + lane * kTileStride)};
        const uint32_t tr_addr[2] = {lds_addr(tile) + tr_off, lds_addr(tile_b) + tr_off};
        uint32_t lo0 = 0, lo1 = 0, rg0 = 0xffffffffu, rg1 = 0xffffffffu, lw = 0, wr = 0xffffffffu, flushed = 0, slow = 0;
        uint32_t ck_index = (uint32_t)(se * ck.n_chunks);
        int32_t smin = a.min_symbol, smax = a.min_symbol;
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the statement keeps its own book from here
        if constexpr (SB == 1)
            range_encode_tiles_loop_ck_n8<FLUSHES>(lo0, lo1, rg0, rg1, lw, wr, flushed, smin, smax, slow, ck_index, row_addr, tr_addr, L.out.lane_addr,
                                                   cap, (uint32_t)slab_off, lds_addr(table) - 8u * (uint32_t)a.min_symbol, (uint32_t)P, a.words,
                                                   symbols_base, (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(N / kTileSyms)), ck.pos, ck.lower,
                                                   ck.range, (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(ck.interval / kTileSyms)), goff);
        else
        range_encode_tiles_loop_ck<FLUSHES>(lo0, lo1, rg0, rg1, lw, wr, flushed, smin, smax, slow, ck_index, row_addr, tr_addr, L.out.lane_addr,
                                            cap, (uint32_t)slab_off, lds_addr(table) - 8u * (uint32_t)a.min_symbol, (uint32_t)P, a.words,
                                            symbols_base, (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(N / kTileSyms)), ck.pos, ck.lower,
                                            ck.range, (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(ck.interval / kTileSyms)), goff);
        if (__builtin_amdgcn_readfirstlane(slow) == 0) {
            L.lower = ((uint64_t)lo1 << 32) | lo0; L.range = ((uint64_t)rg1 << 32) | rg0; L.lw = lw;
            L.out.wr = wr; L.out.flushed = flushed;
            L.bad = max((uint32_t)smax - (uint32_t)a.min_symbol, (uint32_t)smin - (uint32_t)a.min_symbol);
            done = true;
        } else {
            L.init(slab, cap, ring, lane);      // (rare: a carry had to travel) the wave's streams again, with the C++ step
            L.owner = active;
            wave_lds_fence();
        }
    }
    if (!done) {
        // any other shape: symbol by symbol (correct, slow)
        const SymT* row = symbols + se * N;
        for (size_t t = 0; t < N; ++t) {
            if (active && t % ck.interval == 0) {
                const size_t k = s * ck.n_chunks + t / ck.interval;
                ck.pos[k] = (uint32_t)((int32_t)L.out.wr + 1); ck.lower[k] = L.lower; ck.range[k] = L.range;
            }
            const CumProb e = table[enc_index((int32_t)row[t], a.min_symbol, nsym, L.bad)];
            L.step(e.c, e.p, P);
            if ((t & 7) == 7) L.flush();
        }
    }
    uint32_t n_words = 0;
    const int32_t status = L.finish(nsym, n_words);
    if (!active) return;
    a.status[s] = status;
    a.n_words[s] = (status == CST_STREAM_OK) ? n_words : 0u;
}

bool range_encode_ckpt_fast_usable(const RangeEncodeArgs& a, cst_layout layout) {
    return layout == CST_LAYOUT_STREAM_MAJOR && range_encode_fast_usable(a, layout);
}

cst_status range_encode_ckpt_fast(const RangeEncodeArgs& a, const RangeCkptOut& ck, hipStream_t hs) {
    const size_t table_bytes = (((size_t)a.n_symbols * sizeof(CumProb)) + 15) & ~(size_t)15;
    const size_t lds = kFastRingBytes + table_bytes + 2 * kFastTileBytes;
    const size_t blocks = (a.n_streams + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    auto go = [&](auto kernel) -> cst_status {
        CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kBlock), lds, hs, a, ck);
        CST_HIP_TRY(hipGetLastError());
        return CST_OK;
    };
    return a.precision <= 16 ? go(range_encode_ckpt_kernel<1>) : go(range_encode_ckpt_kernel<2>);
}

// int8 symbol matrices read by the loop itself (round 6): a.symbols is a const int8_t* in disguise; rows of whole 32-symbol tiles whose
// chunks are whole tiles run the statement, anything else the kernel's exact symbol-by-symbol path (both read int8)
bool range_encode_n8_usable(const RangeEncodeArgs& a, cst_layout layout, size_t interval) {
    if (knobs().no_n8) return false;                                  // (A/B runs: the conversion path)
    if (layout != CST_LAYOUT_STREAM_MAJOR || !range_encode_fast_usable(a, layout)) return false;      // (16-byte aligned matrix, rows of whole dwords)
    if (a.n_per_stream % kTileSyms != 0 || interval == 0 || interval % kTileSyms != 0 || a.n_per_stream % interval != 0) return false;
    if (a.n_streams * (a.n_per_stream / interval) >= (1u << 28)) return false;
    return a.min_symbol >= -128 && a.min_symbol + a.n_symbols - 1 <= 127;
}

cst_status range_encode_ckpt_n8(const RangeEncodeArgs& a, const RangeCkptOut& ck, hipStream_t hs) {
    const size_t table_bytes = (((size_t)a.n_symbols * sizeof(CumProb)) + 15) & ~(size_t)15;
    const size_t lds = kFastRingBytes + table_bytes + 2 * kFastTileBytes;
    const size_t blocks = (a.n_streams + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    auto go = [&](auto kernel) -> cst_status {
        CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kBlock), lds, hs, a, ck);
        CST_HIP_TRY(hipGetLastError());
        return CST_OK;
    };
    return a.precision <= 16 ? go(range_encode_ckpt_kernel<1, 1>) : go(range_encode_ckpt_kernel<2, 1>);
}

// ---- sub-lane decoder ----
constexpr int kRsThreads = 512;
constexpr int kRsWaves = kRsThreads / kWave;
constexpr size_t kRsRingBytes = (size_t)kRsWaves * kRdSlots * kWave * 4;      // 64 KiB
constexpr int kRsTileRow = 36;                                                  // bytes between the rows of a byte tile
constexpr size_t kRsTileBytes = (size_t)kWave * kRsTileRow;                     // 2304 B per wave and buffer
constexpr size_t kRsDumpBytes = 4 * kWave * 4;                                  // ONE landing area for unused chunk slots (never read)

template <bool ENDS, bool B16, bool N8 = false>
__device__ __forceinline__ void range_decode_tiles_loop_sub(uint32_t& x0, uint32_t& x1, uint32_t& rg0, uint32_t& rg1, uint32_t& pos,
                                                            uint32_t& hi_issued, uint32_t& row_cur, uint32_t& row_prev, uint32_t& tr_cur,
                                                            uint32_t& tr_prev, uint32_t& tiles, uint32_t& ginc, uint32_t& bad,
                                                            uint32_t lut_addr, uint32_t qmax, uint32_t P, uint32_t ring_mask,
                                                            const void* words_base, uint32_t delta_hi, uint64_t store_base,
                                                            uint32_t lens, uint32_t endr, uint32_t ring_lane_addr, uint32_t dump_addr,
                                                            uint32_t words_off, [[maybe_unused]] uint32_t bucket_shift,
                                                            [[maybe_unused]] uint32_t cdf_addr, int32_t min_symbol,
                                                            [[maybe_unused]] uint32_t c_field_mask, [[maybe_unused]] uint32_t index_shift,
                                                            bool plain_stores) {
    if constexpr (N8) {       // int8 symbol matrices (round 6): the byte tile leaves as it is, four symbols per store
#define CST_STORE_MOD ""
        if constexpr (B16 && ENDS) {
#include "cst_range_decode_loop_b16_sub_n8_ends.inc"
        } else if constexpr (B16) {
#include "cst_range_decode_loop_b16_sub_n8.inc"
        } else if constexpr (ENDS) {
#include "cst_range_decode_loop_sub_n8_ends.inc"
        } else {
#include "cst_range_decode_loop_sub_n8.inc"
        }
#undef CST_STORE_MOD
        (void)plain_stores;
        return;
    }
    if constexpr (B16 && ENDS) {
        if (plain_stores) {
#define CST_STORE_MOD ""
#include "cst_range_decode_loop_b16_sub_ends.inc"
#undef CST_STORE_MOD
        } else {
#define CST_STORE_MOD "nt"
#include "cst_range_decode_loop_b16_sub_ends.inc"
#undef CST_STORE_MOD
        }
    } else if constexpr (B16) {
        if (plain_stores) {
#define CST_STORE_MOD ""
#include "cst_range_decode_loop_b16_sub.inc"
#undef CST_STORE_MOD
        } else {
#define CST_STORE_MOD "nt"
#include "cst_range_decode_loop_b16_sub.inc"
#undef CST_STORE_MOD
        }
    } else if constexpr (ENDS) {
        if (plain_stores) {
#define CST_STORE_MOD ""
#include "cst_range_decode_loop_sub_ends.inc"
#undef CST_STORE_MOD
        } else {
#define CST_STORE_MOD "nt"
#include "cst_range_decode_loop_sub_ends.inc"
#undef CST_STORE_MOD
        }
    } else {
        if (plain_stores) {
#define CST_STORE_MOD ""
#include "cst_range_decode_loop_sub.inc"
#undef CST_STORE_MOD
        } else {
#define CST_STORE_MOD "nt"
#include "cst_range_decode_loop_sub.inc"
#undef CST_STORE_MOD
        }
    }
}

// LDS layout: [word rings, 8 KiB per wave][tables][byte tiles A, 2304 B per wave][byte tiles B][dump 1 KiB]
// P <= 12: cp[q] = c | p << 16 at +0, the symbol INDEX of quantile q (int32) at +16384; B16: cdf, bucket entries, second level
// SB = bytes per symbol of the matrix behind a.symbols: 4 (int32) or 1 (int8, round 6: the byte tiles hold the symbols themselves)
template <bool B16, int SB = 4>
__global__ __launch_bounds__(kRsThreads) void range_decode_sub_kernel(const RangeDecodeArgs a) {
    using SymT = typename std::conditional<SB == 1, int8_t, int32_t>::type;
    SymT* symbols = reinterpret_cast<SymT*>(a.symbols);
    const int32_t tile_bias = SB == 1 ? a.min_symbol : 0;             // what the byte tile holds: index (int32 matrix) or symbol (int8)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = threadIdx.x >> 6;
    const int P = a.precision;
    const size_t n_q = (size_t)1 << P;
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem) + wave_in_block * (kRdSlots * kWave);
    uint32_t* lut = reinterpret_cast<uint32_t*>(smem + kRsRingBytes);
    int32_t* idxt = reinterpret_cast<int32_t*>(smem + kRsRingBytes + kTileSymOffset);
    DecLut blut{};
    const uint32_t* cdf = a.cdf;
    const uint16_t* bucket = a.bucket;
    size_t table_bytes = kTileLutBytes;
    if constexpr (B16) {
        table_bytes = stage_decoder_tables<kDecBucket, true, true>(smem + kRsRingBytes, P, a.dec_cp, a.dec_idx, a.cdf, a.bucket, a.bucket_bits,
                                                             a.n_symbols, blut, cdf, bucket);
        table_bytes = (table_bytes + 15) & ~(size_t)15;
    } else {
        for (size_t q = threadIdx.x; q < n_q; q += blockDim.x) {
            lut[q] = a.dec_cp[q];
            idxt[q] = (int32_t)a.dec_idx[q] + tile_bias;
        }
    }
    unsigned char* tile = smem + kRsRingBytes + table_bytes + (size_t)wave_in_block * kRsTileBytes;
    unsigned char* tile_b = tile + kRsWaves * kRsTileBytes;
    uint32_t* dump = reinterpret_cast<uint32_t*>(smem + kRsRingBytes + table_bytes + 2 * kRsWaves * kRsTileBytes) + lane;
    if ((lds_addr(ring) & (uint32_t)(kRdSlots * kWave * 4 - 1)) != 0) __builtin_trap();
    __syncthreads();

    // A wave decodes ONE chunk of 64 different streams (symbol rows a whole stream apart, like the plain decoder's).  The grid
    // visits the chunks one after another -- all groups of 64 streams at chunk 0, then at chunk 1, ...: with the k chunks of a
    // group in one workgroup instead (CST_SUB_ORDER=0) k = 4 measured 0.395 instead of 0.365 ms at P = 12, 0.472 / 0.438 at P = 24,
    // k = 2 the same (one MI355X, 65 536 x 4096).
    const size_t wave_global = (size_t)blockIdx.x * kRsWaves + wave_in_block;
    const size_t n_groups = (a.n_streams + kWave - 1) / kWave;
    const bool chunk_major = (a.flags & 0x100u) == 0;
    const size_t chunk = chunk_major ? wave_global / n_groups : wave_global % a.n_chunks;
    const size_t s0 = (chunk_major ? wave_global % n_groups : wave_global / a.n_chunks) * kWave;
    if (chunk >= a.n_chunks || s0 >= a.n_streams) return;
    const bool active = s0 + lane < a.n_streams;
    const size_t s = active ? s0 + lane : a.n_streams - 1;        // the lanes of a partial wave beyond its last stream repeat that stream
    const size_t ve = s * a.n_chunks + chunk;                     // (stream, chunk) in the jump table and in d_status
    const size_t v = ve;
    const size_t N = a.n_per_stream;
    const size_t K = a.interval;                                  // symbols per chunk
    const size_t n_full = K / kTileSyms;
    const int bucket_shift = P - a.bucket_bits;
    // the whole stream's words: a lane reads on past its chunk (the point register looks two words ahead), never past the stream
    const WordSlice ws = word_slice(a.offsets, a.stride_words, a.n_words, s, a.words_capacity);
    const uint32_t* my_words = a.words + ws.off;
    const uint32_t my_len = ws.n;
    const uint32_t pos0 = a.ckpt_pos[ve];
    const bool bad_pos = pos0 > my_len;                           // a jump point outside its stream: caller data, flagged
    const uint64_t lower0 = a.ckpt_lower[ve], range0 = a.ckpt_range[ve];

    RangeDecLane<32, 64, kRdSlots, kRdAhead> L;
    L.init_at(my_words, my_len, pos0, lower0, range0, ring, lane);
    L.in.prime();
    wave_lds_fence();

    auto step = [&]() -> int32_t {                                // the exact step (queue.rs:968-1033); returns the symbol
        const uint32_t q = L.peek_quantile(P);
        uint32_t c, p;
        int32_t sym;
        if constexpr (B16) {
            uint32_t idx;
            lookup_quantile<kDecBucket>(q, blut, cdf, bucket, bucket_shift, a.n_symbols, idx, c, p);
            sym = a.min_symbol + (int32_t)idx;
        } else {
            const uint32_t cp = lut[q];
            c = cp & 0xffffu; p = cp >> 16;
            sym = a.min_symbol + idxt[q] - tile_bias;
        }
        const uint32_t w = L.in.peek();
        L.in.pos += L.advance(c, p, P, w, L.in.pos < L.in.len) ? 1u : 0u;
        return sym;
    };

    bool tiles_done = false;
    SymT* out_row = symbols + s * N + chunk * K;
    {
        const unsigned char* words_base = reinterpret_cast<const unsigned char*>(reinterpret_cast<uintptr_t>(a.words) & ~(uintptr_t)15);
        const uint64_t w_off = (uint64_t)(reinterpret_cast<const unsigned char*>(L.in.base16) - words_base);
        const bool off_ok = w_off + 4ull * ((uint64_t)my_len + 8) < 0x80000000ull;
        if (n_full > 0 && K % kTileSyms == 0 && N < (1u << 24) && (reinterpret_cast<uintptr_t>(a.symbols) & 15) == 0 && (N * SB) % 4 == 0 &&
            !__any(!off_ok)) {
            const size_t last_row = min((size_t)kWave, a.n_streams - s0) - 1;
            uint32_t goff[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) goff[k] = (uint32_t)((min((size_t)(lane >> 3) + 8 * k, last_row) * N + 4 * (size_t)(lane & 7)) * SB);
            auto leave_offsets = [&](uint32_t row_cur_addr) {      // the statements read their store offsets from the lane's row
                uint32_t* cur = reinterpret_cast<uint32_t*>((row_cur_addr == lds_addr(tile + lane * kRsTileRow) ? tile : tile_b) + lane * kRsTileRow);
#pragma unroll
                for (int k = 0; k < 8; ++k) cur[k] = goff[k];
                wave_lds_fence();
            };
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the statements keep their own book from here
            const uint32_t shift = L.in.shift;
            const uint64_t x_in = (uint64_t)L.point - (uint64_t)L.lower;          // the statements carry x = point - lower
            uint32_t x0 = (uint32_t)x_in, x1 = (uint32_t)(x_in >> 32), rg0 = (uint32_t)L.range, rg1 = (uint32_t)(L.range >> 32);
            uint32_t pos = L.in.pos + shift, hi_issued = L.in.hi_issued;
            const uint32_t lens = my_len + shift, endr = (lens + 3u) & ~3u;
            const uint32_t tr_off = (uint32_t)((lane >> 3) * kRsTileRow + 4 * (lane & 7));
            uint32_t row_cur = lds_addr(tile + lane * kRsTileRow), row_prev = lds_addr(tile_b + lane * kRsTileRow);
            uint32_t tr_cur = lds_addr(tile) + tr_off, tr_prev = lds_addr(tile_b) + tr_off;
            uint32_t tiles = (uint32_t)n_full, ginc = 0, bad = 0, bad2 = 0;
            const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(symbols + s0 * N + chunk * K);
            const uint64_t store_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
            const bool plain_stores = __builtin_amdgcn_readfirstlane((int)((((N * 4) % 128 != 0 || (sb & 127) != 0)) ? 1 : 0)) != 0;
            const uint32_t qmax = (1u << P) - 1u, ring_mask = (uint32_t)(kRdSlots - 1) << 8;
            const uint32_t delta_hi = P <= 16 ? 0x3e100000u : 0x3e900000u;   // 2^-30, 2^-22 (see range_decode_fast_kernel)
            const uint32_t lut_addr = B16 ? lds_addr(blut.b16) : lds_addr(lut);
            const uint32_t cdf_addr = B16 ? lds_addr(cdf) : 0u;
            const uint32_t idx_shift = B16 ? (uint32_t)blut.idx_shift : 24u, idx_mask = (1u << idx_shift) - 1u;
            leave_offsets(row_cur);
            range_decode_tiles_loop_sub<false, B16, SB == 1>(x0, x1, rg0, rg1, pos, hi_issued, row_cur, row_prev, tr_cur, tr_prev, tiles, ginc, bad, lut_addr,
                                                    qmax, (uint32_t)P, ring_mask, words_base, delta_hi, store_base, lens, endr,
                                                    lds_addr(ring + lane), lds_addr(dump), (uint32_t)w_off,
                                                    (uint32_t)__builtin_amdgcn_readfirstlane(bucket_shift), cdf_addr, a.min_symbol, idx_mask, idx_shift,
                                                    plain_stores);
            tiles = (uint32_t)__builtin_amdgcn_readfirstlane(tiles);
            if (tiles > 0) {
                const uint32_t done = (uint32_t)n_full - tiles;
                const uint64_t base2 = store_base + (done > 0 ? (uint64_t)(done - 1) * (kTileSyms * SB) : 0);
                leave_offsets(row_cur);
                range_decode_tiles_loop_sub<true, B16, SB == 1>(x0, x1, rg0, rg1, pos, hi_issued, row_cur, row_prev, tr_cur, tr_prev, tiles, ginc, bad2,
                                                       lut_addr, qmax, (uint32_t)P, ring_mask, words_base, delta_hi, base2, lens, endr,
                                                       lds_addr(ring + lane), lds_addr(dump), (uint32_t)w_off,
                                                       (uint32_t)__builtin_amdgcn_readfirstlane(bucket_shift), cdf_addr, a.min_symbol, idx_mask,
                                                       idx_shift, plain_stores);
            }
            if (__builtin_amdgcn_readfirstlane(bad | bad2) == 0) {
                // the last tile is still in LDS (buffer A if it has an even index): bytes -> int32 symbols -> HBM
                wave_lds_fence();
                const unsigned char* last = ((n_full - 1) & 1) ? tile_b : tile;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const size_t R = min((size_t)(lane >> 3) + 8 * k, last_row);
                    const uint32_t pk = *reinterpret_cast<const uint32_t*>(last + ((lane >> 3) + 8 * k) * kRsTileRow + 4 * (lane & 7));
                    if constexpr (SB == 1) {      // the bytes are the symbols
                        *reinterpret_cast<uint32_t*>(symbols + (s0 + R) * N + chunk * K + (n_full - 1) * kTileSyms + 4 * (lane & 7)) = pk;
                        continue;
                    }
                    v4i t;
                    t.x = a.min_symbol + (int32_t)(pk & 0xffu); t.y = a.min_symbol + (int32_t)((pk >> 8) & 0xffu);
                    t.z = a.min_symbol + (int32_t)((pk >> 16) & 0xffu); t.w = a.min_symbol + (int32_t)(pk >> 24);
                    __builtin_nontemporal_store(t, reinterpret_cast<v4i*>(a.symbols + (s0 + R) * N + chunk * K + (n_full - 1) * kTileSyms + 4 * (lane & 7)));
                }
                wave_lds_fence();
                tiles_done = true;
            } else {
                // a quantile estimate failed its check, or the data are invalid: this wave's chunks again, exactly
                L.init_at(my_words, my_len, pos0, lower0, range0, ring, lane);
                L.in.prime();
                wave_lds_fence();
            }
        }
    }
    if (!tiles_done) {
        for (size_t t = 0; t < K; ++t) {          // shapes the statements do not take, and the exact repeat: symbol by symbol
            const int32_t sym = step();
            if (active) out_row[t] = (SymT)sym;
            if ((t & 3) == 3) { L.in.fill_blocking(); wave_lds_fence(); }
        }
    }
    if (!active) return;
    a.status[v] = (ws.bad || bad_pos) ? (int32_t)CST_STREAM_INVALID_DATA : L.status;
}

static size_t range_decode_sub_lds(const RangeDecodeArgs& a) {
    const size_t tables = a.precision <= 12 ? (size_t)kTileLutBytes : ((b16_tables_bytes(a.n_symbols, a.bucket_bits) + 15) & ~(size_t)15);
    return kRsRingBytes + tables + 2 * kRsWaves * kRsTileBytes + kRsDumpBytes;
}

// (32,64), table decoders (8 <= P <= 12) or bucket entries (P <= 24), alphabets of at most 256 symbols (the tile holds bytes)
bool range_decode_sub_usable(const RangeDecodeArgs& a) {
    const bool table = a.precision >= 8 && a.precision <= 12 && a.dec_cp && a.dec_idx;
    const bool entries = a.precision > 12 && a.precision <= 24 && bucket16_usable(a.n_symbols, a.precision) && a.cdf && a.bucket;
    if (!(table || entries) || a.n_symbols > 256 || a.n_chunks == 0 || a.n_streams > 0x7fffffffull / a.n_chunks) return false;
    return range_decode_sub_lds(a) <= 160 * 1024;
}

cst_status range_decode_sub(const RangeDecodeArgs& a, hipStream_t hs) {
    const size_t lds = range_decode_sub_lds(a);
    const size_t blocks = ((a.n_streams + kWave - 1) / kWave * a.n_chunks + kRsWaves - 1) / kRsWaves;      // a wave = 64 streams x one chunk
    RangeDecodeArgs b = a;
    if (knobs().sub_order_flat) b.flags |= 0x100u;            // (A/B runs: the chunks of a group side by side)
    auto go = [&](auto kernel) -> cst_status {
        CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kRsThreads), lds, hs, b);
        CST_HIP_TRY(hipGetLastError());
        return CST_OK;
    };
    return a.precision <= 12 ? go(range_decode_sub_kernel<false>) : go(range_decode_sub_kernel<true>);
}

// ... writing an int8 symbol matrix (a.symbols is an int8_t* in disguise): the support must fit the type
bool range_decode_sub_n8_usable(const RangeDecodeArgs& a) {
    if (knobs().no_n8) return false;                                  // (A/B runs: the conversion path)
    return range_decode_sub_usable(a) && a.min_symbol >= -128 && a.min_symbol + a.n_symbols - 1 <= 127 && a.interval % 4 == 0 &&
           a.n_per_stream % 4 == 0 && (reinterpret_cast<uintptr_t>(a.symbols) & 15) == 0;
}

cst_status range_decode_sub_n8(const RangeDecodeArgs& a, hipStream_t hs) {
    const size_t lds = range_decode_sub_lds(a);
    const size_t blocks = ((a.n_streams + kWave - 1) / kWave * a.n_chunks + kRsWaves - 1) / kRsWaves;
    RangeDecodeArgs b = a;
    if (knobs().sub_order_flat) b.flags |= 0x100u;
    auto go = [&](auto kernel) -> cst_status {
        CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kRsThreads), lds, hs, b);
        CST_HIP_TRY(hipGetLastError());
        return CST_OK;
    };
    return a.precision <= 12 ? go(range_decode_sub_kernel<false, 1>) : go(range_decode_sub_kernel<true, 1>);
}

} // namespace cst
