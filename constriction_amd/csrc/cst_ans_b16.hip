// cst_ans_b16.hip -- the hand-scheduled (32,64) ANS decoder for 12 < P <= 24 (DefaultAnsCoder's PRECISION = 24),
// shared table of at most 256 symbols (1024 for P <= 22), both symbol layouts.  At this precision there is no table of 2^P quantiles: the lookup
// (lookup_contiguous.rs:564-605, contiguous.rs:628-665) is ONE 16-byte LDS read of a bucket entry (DecLut::b16,
// cst_common.hpp); scripts/gen_decode_loop_b16.py has the instruction-level account.  The step itself is
// AnsCoder::decode_symbol, stack.rs:1084-1097.
#include <cstdlib>

#include "cst_ans_kernels.hpp"

namespace cst {

__device__ __forceinline__ void ans_decode_b16_tiles_loop(uint32_t& lo, uint32_t& hi, uint32_t& rd, uint32_t& lo_issued, uint32_t& row_cur,
                                                          uint32_t& row_prev, uint32_t& tr_cur, uint32_t& tr_prev, uint32_t lut_addr,
                                                          uint32_t cdf_addr, uint32_t mask, uint32_t P, uint32_t bucket_shift,
                                                          int32_t min_symbol, uint32_t ring_mask, const void* words_base, uint64_t store_base,
                                                          uint32_t n_tiles, uint32_t shift_minus_1,
                                                          uint32_t ring_lane_addr, uint32_t dump_addr, uint32_t words_off,
                                                          uint32_t c_field_mask, uint32_t index_shift, bool plain_stores) {
    if (plain_stores) {          // rows that are not cache-line aligned: see scripts/gen_decode_loop.py (CST_STORE_MOD)
#define CST_STORE_MOD ""
#include "cst_decode_loop_b16.inc"
#undef CST_STORE_MOD
    } else {
#define CST_STORE_MOD "nt"
#include "cst_decode_loop_b16.inc"
#undef CST_STORE_MOD
    }
}

// the same for symbols[t][stream] (scripts/gen_decode_loop_b16.py, SYMBOL_MAJOR): full waves only
__device__ __forceinline__ void ans_decode_b16_tiles_loop_sm(uint32_t& lo, uint32_t& hi, uint32_t& rd, uint32_t& lo_issued, uint32_t& row_cur,
                                                             uint32_t& row_prev, uint32_t& tr_cur, uint32_t& tr_prev, uint32_t lut_addr,
                                                             uint32_t cdf_addr, uint32_t mask, uint32_t P, uint32_t bucket_shift,
                                                             int32_t min_symbol, uint32_t ring_mask, const void* words_base, uint64_t store_base,
                                                             uint32_t n_tiles, uint32_t shift_minus_1,
                                                             uint32_t ring_lane_addr, uint32_t dump_addr, uint32_t words_off,
                                                             uint32_t tile_step_bytes, uint32_t c_field_mask, uint32_t index_shift,
                                                             bool plain_stores) {
    if (plain_stores) {
#define CST_STORE_MOD ""
#include "cst_decode_loop_b16_sm.inc"
#undef CST_STORE_MOD
    } else {
#define CST_STORE_MOD "nt"
#include "cst_decode_loop_b16_sm.inc"
#undef CST_STORE_MOD
    }
}

constexpr size_t kB16RingBytes = (size_t)(kBlock / kWave) * kDecRingSlots * kWave * 4;
constexpr size_t kB16TileWords = (size_t)kWave * kTileStride;

static size_t b16_table_bytes(int n_symbols, int bucket_bits) {
    return ((((size_t)n_symbols + 1) * 4 + 15) & ~(size_t)15) + ((size_t)16 << bucket_bits) + kSubAreaBytes;
}
static size_t b16_lds_bytes(int n_symbols, int bucket_bits) {
    return kB16RingBytes + b16_table_bytes(n_symbols, bucket_bits) + 2 * (size_t)(kBlock / kWave) * kB16TileWords * 4 + kTileDumpBytes;
}

// LDS layout: [word rings, 8 KiB per wave][cdf][bucket entries][second-level tables][symbol tiles A][symbol tiles B][dump rows]
template <int LAYOUT>
__global__ __launch_bounds__(kBlock) void ans_decode_b16_kernel(const AnsDecodeArgs a) {
    constexpr bool SM = LAYOUT == CST_LAYOUT_SYMBOL_MAJOR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = threadIdx.x >> 6;
    const int P = a.precision;
    DecLut lut{};
    const uint32_t* cdf = a.cdf;
    const uint16_t* bucket = a.bucket;
    const size_t lds_off = stage_decoder_tables<kDecBucket, true, true>(smem + kB16RingBytes, P, a.dec_cp, a.dec_idx, a.cdf, a.bucket, a.bucket_bits,
                                                                  a.n_symbols, lut, cdf, bucket);
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem) + wave_in_block * (kDecRingSlots * kWave);
    int32_t* tile = reinterpret_cast<int32_t*>(smem + kB16RingBytes + lds_off) + wave_in_block * kB16TileWords;
    int32_t* tile_b = tile + (kBlock / kWave) * kB16TileWords;
    uint32_t* dump = reinterpret_cast<uint32_t*>(smem + kB16RingBytes + lds_off + 2 * (size_t)(kBlock / kWave) * kB16TileWords * 4) +
                     wave_in_block * (4 * kWave) + lane;
    if ((lds_addr(ring) & (uint32_t)(kDecRingSlots * kWave * 4 - 1)) != 0) __builtin_trap();   // the ring address is formed with v_and_or
    __syncthreads();

    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t s0 = wave * kWave;
    if (s0 >= a.n_streams) return;
    const size_t s = s0 + lane;
    const bool active = s < a.n_streams;
    const size_t N = a.n_per_stream;
    const size_t n_full = N / kTileSyms;
    const int bucket_shift = P - a.bucket_bits;

    // the lanes of a partial wave beyond its last stream REPEAT that stream (same words, same symbols, stored onto its row
    // again): the wave then runs the main-loop statement like a full one
    const size_t se = active ? s : a.n_streams - 1;
    DecLane<32, 64, kDecRingSlots, kDecAhead> L;
    const WordSlice ws = word_slice(a.offsets, a.stride_words, a.n_words, se, a.words_capacity);
    L.init(a.words + ws.off, ws.n, ring, lane);
    // CST_FLAG_RAW_STATE: the coder continues from (d_state, d_n_words) -- AnsCoder::seek, stack.rs:1107-1139; the chunks of a
    // checkpointed stream are such coders (cst_ans_ckpt.hip) -- instead of reading its state from the end of the words
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;
    if (raw) L.state = a.state[se];
    else L.read_initial_state();
    L.in.prime();
    wave_lds_fence();

    auto next_symbol = [&]() -> int32_t {
        return a.min_symbol + (int32_t)ans_decode_step<32, 64, kDecBucket, true>(L, lut, cdf, bucket, bucket_shift, a.n_symbols, P);
    };
    // one tile with the compiler-scheduled step into `dst` (the lane's tile row); the window is topped up every half tile
    auto tile_cxx = [&](int32_t* dst) {
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {
                int4 v;
                v.x = next_symbol(); v.y = next_symbol(); v.z = next_symbol(); v.w = next_symbol();
                *reinterpret_cast<int4*>(dst + 16 * h + 4 * j) = v;
            }
            L.in.refill_blocking();
            wave_lds_fence();
        }
    };

    int32_t* my = tile + lane * kTileStride;
    size_t tb = 0;
    bool all_done = false;
    {
        const unsigned char* words_base = reinterpret_cast<const unsigned char*>(reinterpret_cast<uintptr_t>(a.words) & ~(uintptr_t)15);
        const uint64_t w_off = (uint64_t)(reinterpret_cast<const unsigned char*>(L.in.base16) - words_base);
        const bool off_ok = w_off + 4ull * ((uint64_t)L.in.rd + 8) < 0x80000000ull;
        if (SM && n_full >= 2 && N < (1u << 24) && !__any(!off_ok) && s0 + kWave <= a.n_streams && a.n_streams % 4 == 0 &&
            a.n_streams < (1u << 24) && (reinterpret_cast<uintptr_t>(a.symbols) & 15) == 0) {
            tile_cxx(my);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            uint32_t lo = (uint32_t)L.state, hi = (uint32_t)(L.state >> 32);
            const uint32_t tr_off = (uint32_t)(((4 * (lane & 7)) * kTileStride + (lane >> 3)) * 4);
            uint32_t row_cur = lds_addr(tile_b + lane * kTileStride), row_prev = lds_addr(my);
            uint32_t tr_cur = lds_addr(tile_b) + tr_off, tr_prev = lds_addr(tile) + tr_off;
            const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(a.symbols + s0);
            const uint64_t store_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
            const bool plain = __builtin_amdgcn_readfirstlane((int)(((a.n_streams * 4) % 128 != 0 || (sb & 127) != 0) ? 1 : 0)) != 0;
            // (this branch depends on s0, which the compiler takes for divergent: pin the uniform operands to SGPRs)
            const uint64_t wb = (uint64_t)reinterpret_cast<uintptr_t>(words_base);
            const void* words_base_u = reinterpret_cast<const void*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(wb >> 32)) << 32) |
                                                                     (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)wb));
            // the statement reads its eight store offsets from the lane's row of the current tile buffer
#pragma unroll
            for (int k = 0; k < 8; ++k)
                tile_b[lane * kTileStride + k] = (int32_t)(uint32_t)((((size_t)(lane >> 3) + 8 * (k >> 1)) * a.n_streams + 32 * (size_t)(k & 1) + 4 * (size_t)(lane & 7)) * 4);
            wave_lds_fence();
            ans_decode_b16_tiles_loop_sm(lo, hi, L.in.rd, L.in.lo_issued, row_cur, row_prev, tr_cur, tr_prev, lds_addr(lut.b16), lds_addr(cdf),
                                         (P >= 32) ? 0xffffffffu : ((1u << P) - 1u), (uint32_t)P, (uint32_t)__builtin_amdgcn_readfirstlane(bucket_shift), a.min_symbol, kDecRingMask,
                                         words_base_u, store_base,
                                         (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(n_full - 1)), L.in.shift - 1u, lds_addr(ring + lane),
                                         lds_addr(dump), (uint32_t)w_off,
                                         (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(kTileSyms * a.n_streams * 4)), (1u << lut.idx_shift) - 1u, (uint32_t)lut.idx_shift, plain);
            L.state = ((uint64_t)hi << 32) | lo;
            wave_lds_fence();
            tile_store_sm(a.symbols, a.n_streams, s0, (n_full - 1) * kTileSyms, lane, ((n_full - 1) & 1) ? tile_b : tile);
            wave_lds_fence();
            tb = n_full;
        } else if (!SM && N >= 4 * kTileSyms && N < (1u << 24) && !__any(!off_ok)) {
            // Rows of any length and alignment (row_skew, cst_ans_kernels.hpp): every lane first decodes the `pre` symbols in
            // front of its row's next cache-line boundary, so that its tiles -- and the 128-byte segments the tile stores
            // write -- are whole cache lines; the lanes of a partial wave beyond its last stream repeat that stream's row.
            const size_t last_row = min((size_t)kWave, a.n_streams - s0) - 1;
            int32_t* out_row = a.symbols + se * N;
            const uint32_t pre = row_skew(a.symbols, se, N);
            uint32_t max_pre = pre;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) max_pre = max(max_pre, (uint32_t)__shfl_xor((int)max_pre, d));
            max_pre = (uint32_t)__builtin_amdgcn_readfirstlane((int)max_pre);
            // `count` more symbols of every lane that has them, straight to HBM (the window is topped up every eight)
            auto direct = [&](uint32_t first, uint32_t have, uint32_t count) {
                for (uint32_t j = 0; j < count; ++j) {
                    if (j < have) { const int32_t sym = next_symbol(); if (active) out_row[first + j] = sym; }
                    if ((j & 7) == 7) { L.in.refill_blocking(); wave_lds_fence(); }
                }
                L.in.refill_blocking();
                wave_lds_fence();
            };
            if (max_pre) direct(0, pre, max_pre);
            const size_t n_t = (N - max_pre) / kTileSyms;           // whole tiles every lane has (>= 3)
            tile_cxx(my);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const size_t R = min((size_t)(lane >> 3) + 8 * k, last_row);
                tile_b[lane * kTileStride + k] = (int32_t)(uint32_t)((R * N + row_skew(a.symbols, s0 + R, N) + 4 * (size_t)(lane & 7)) * 4);
            }
            wave_lds_fence();
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the statement keeps its own book from here
            uint32_t lo = (uint32_t)L.state, hi = (uint32_t)(L.state >> 32);
            const uint32_t tr_off = (uint32_t)(((lane >> 3) * kTileStride + 4 * (lane & 7)) * 4);
            // current = B (tile 1), previous = A (tile 0)
            uint32_t row_cur = lds_addr(tile_b + lane * kTileStride), row_prev = lds_addr(my);
            uint32_t tr_cur = lds_addr(tile_b) + tr_off, tr_prev = lds_addr(tile) + tr_off;
            const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(a.symbols + s0 * N);
            const uint64_t store_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
            // (every stored segment is a whole cache line now: the streaming form of the stores in every case)
            ans_decode_b16_tiles_loop(lo, hi, L.in.rd, L.in.lo_issued, row_cur, row_prev, tr_cur, tr_prev, lds_addr(lut.b16), lds_addr(cdf),
                                      (P >= 32) ? 0xffffffffu : ((1u << P) - 1u), (uint32_t)P, (uint32_t)__builtin_amdgcn_readfirstlane(bucket_shift), a.min_symbol, kDecRingMask,
                                      words_base, store_base, (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(n_t - 1)),
                                      L.in.shift - 1u, lds_addr(ring + lane), lds_addr(dump), (uint32_t)w_off, (1u << lut.idx_shift) - 1u, (uint32_t)lut.idx_shift, false);
            L.state = ((uint64_t)hi << 32) | lo;
            // the last tile is still in LDS (buffer A if it has an even index)
            wave_lds_fence();
            {
                const int32_t* last = ((n_t - 1) & 1) ? tile_b : tile;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const size_t R = min((size_t)(lane >> 3) + 8 * k, last_row);
                    const int4 v = *reinterpret_cast<const int4*>(last + ((lane >> 3) + 8 * k) * kTileStride + 4 * (lane & 7));
                    v4i t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
                    __builtin_nontemporal_store(t, reinterpret_cast<v4i*>(a.symbols + (s0 + R) * N + row_skew(a.symbols, s0 + R, N) + (n_t - 1) * kTileSyms + 4 * (lane & 7)));
                }
            }
            wave_lds_fence();
            // ... and what is left of each row behind its last whole tile (fewer than 64 symbols)
            L.in.refill_blocking();
            wave_lds_fence();
            const uint32_t done = pre + (uint32_t)(n_t * kTileSyms);
            direct(done, (uint32_t)N - done, (uint32_t)N - (uint32_t)(n_t * kTileSyms));
            all_done = true;
            tb = n_full;
        }
    }
    for (; tb < n_full; ++tb) {
        tile_cxx(my);
        if constexpr (SM) {
            // (partial waves and odd buffers: every lane writes its own column, 64 consecutive streams per symbol row)
            if (active)
                for (int t = 0; t < kTileSyms; ++t) a.symbols[(tb * kTileSyms + t) * a.n_streams + s] = my[t];
        } else {
            tile_store<true>(a.symbols, a.n_streams, N, s0, tb * kTileSyms, lane, tile);
        }
        wave_lds_fence();
    }
    int32_t* row = SM ? a.symbols + (active ? s : 0) : a.symbols + (active ? s : 0) * N;
    const size_t step_t = SM ? a.n_streams : 1;
    for (size_t t = all_done ? N : n_full * kTileSyms; t < N; ++t) {
        const int32_t sym = next_symbol();
        if (active) row[t * step_t] = sym;
        L.in.refill_blocking();
        wave_lds_fence();
    }
    if (!active) return;
    a.status[s] = ws.bad ? (int32_t)CST_STREAM_INVALID_DATA : L.status;
    if (raw) {
        a.state[s] = (uint64_t)L.state;
        if (a.n_words_out) a.n_words_out[s] = L.in.rd;
    }
}

// ------------------------------------------------------------------------------------------------
// Encoder for 12 < P <= 24 (scripts/gen_encode_loop_wide.py): the P <= 12 pipeline on unpacked table entries
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ans_encode_wide_tiles_loop(uint32_t& lo, uint32_t& hi, uint32_t& wr, uint32_t& flushed, int32_t& smin,
                                                           int32_t& smax, const uint32_t (&tile_row_addr)[2], const uint32_t (&tile_tr_addr)[2],
                                                           uint32_t ring_lane_addr, uint32_t cap, uint32_t slab_off,
                                                           uint32_t table_addr_biased, uint32_t P, const void* words_base,
                                                           uint64_t symbols_base, uint32_t n_tiles, const uint32_t (&goff)[8]) {
#include "cst_encode_loop_wide.inc"
}

// the same for symbols[t][stream] (staging as in ans_encode_tiles_loop_sm, cst_ans_asm.hpp)
__device__ __forceinline__ void ans_encode_wide_tiles_loop_sm(uint32_t& lo, uint32_t& hi, uint32_t& wr, uint32_t& flushed, int32_t& smin,
                                                              int32_t& smax, const uint32_t (&tile_row_addr)[2], const uint32_t (&tile_tr_addr)[2],
                                                              uint32_t ring_lane_addr, uint32_t cap, uint32_t slab_off,
                                                              uint32_t table_addr_biased, uint32_t P, const void* words_base,
                                                              uint64_t symbols_base, uint32_t n_tiles, uint32_t tile_step_bytes,
                                                              const uint32_t (&goff)[8]) {
#include "cst_encode_loop_wide_sm.inc"
}

constexpr size_t kWideRingBytes = (size_t)(kBlock / kWave) * kRingWords * 4;
constexpr size_t kWideTileBytes = (size_t)(kBlock / kWave) * kWave * kTileStride * 4;

// LDS layout: [word rings, 16 KiB per wave][table][symbol tiles A][symbol tiles B]
template <int LAYOUT>
__global__ __launch_bounds__(kBlock) void ans_encode_wide_kernel(const AnsEncodeArgs a) {
    constexpr bool SM = LAYOUT == CST_LAYOUT_SYMBOL_MAJOR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = threadIdx.x >> 6;
    const int P = a.precision;
    const size_t table_bytes = (size_t)a.n_symbols * sizeof(EncEntry);
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem) + wave_in_block * kRingWords;
    EncEntry* table = reinterpret_cast<EncEntry*>(smem + kWideRingBytes);
    int32_t* tile = reinterpret_cast<int32_t*>(smem + kWideRingBytes + table_bytes) + wave_in_block * (kWave * kTileStride);
    if ((lds_addr(ring) & (kRingWords * 4u - 1u)) != 0) __builtin_trap();   // the ring address is formed with v_and_or
    for (int i = threadIdx.x; i < a.n_symbols; i += blockDim.x) table[i] = a.enc[i];
    __syncthreads();

    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t s0 = wave * kWave;
    if (s0 >= a.n_streams) return;
    const size_t s = s0 + lane;
    const bool active = s < a.n_streams;
    const size_t N = a.n_per_stream;
    const uint32_t nsym = (uint32_t)a.n_symbols;
    const size_t n_full = N / kTileSyms;

    // the lanes of a partial wave beyond its last stream REPEAT that stream (same symbols, same slab, same words)
    const size_t se = active ? s : a.n_streams - 1;
    EncLane<32, 64> L;
    L.init(a.words + se * a.stride_words, (uint32_t)(a.stride_words > 0xffffffffull ? 0xffffffffull : a.stride_words), ring, lane);
    auto code = [&](int32_t v) { L.template step<true>(table[enc_index(v, a.min_symbol, nsym, L.bad)], P); };

    const int32_t* row = SM ? a.symbols + se : a.symbols + se * N;
    const size_t step_t = SM ? a.n_streams : 1;
    const uint64_t slab_off = (uint64_t)(reinterpret_cast<const unsigned char*>(L.out.base16) - reinterpret_cast<const unsigned char*>(a.words));
    const bool ok = slab_off + 4ull * L.out.cap < 0x100000000ull && (reinterpret_cast<uintptr_t>(L.out.base16) & 63) == 0 &&
                    (L.out.cap & 15u) == 0 && L.out.shift == 0;
    size_t tb = n_full;
    if (!SM && N >= 4 * kTileSyms && N < (1u << 24) && !__any(!ok)) {
        // Rows of any length and alignment (row_skew, cst_ans_kernels.hpp): lane l's tiles start row_skew() symbols into its
        // row, so that every 128-byte segment a tile load reads is one whole cache line; what is left of the row above the
        // highest and below the lowest whole tile goes through LDS in bulk reads.  The lanes of a partial wave beyond its
        // last stream repeat that stream.
        const size_t last_row = min((size_t)kWave, a.n_streams - s0) - 1;
        int32_t* my = tile + lane * kTileStride;
        auto ragged = [&](size_t p0, uint32_t cnt) { code_ragged(row, p0, cnt, my, a.min_symbol, code, [&]() { L.flush_chunks(); }); };
        const uint32_t pre = row_skew(a.symbols, se, N);
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
            const size_t R = min((size_t)(lane >> 3) + 8 * k, last_row);
            goff[k] = (uint32_t)((R * N + row_skew(a.symbols, s0 + R, N) + 4 * (size_t)(lane & 7)) * 4);
        }
        const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(a.symbols + s0 * N + (n_t - 1) * kTileSyms);
        const uint64_t symbols_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                      (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
        const uint32_t tr_off = (uint32_t)(((lane >> 3) * kTileStride + 4 * (lane & 7)) * 4);
        int32_t* tile_b = tile + (kBlock / kWave) * (kWave * kTileStride);
        const uint32_t row_addr[2] = {lds_addr(tile + lane * kTileStride), lds_addr(tile_b + lane * kTileStride)};
        const uint32_t tr_addr[2] = {lds_addr(tile) + tr_off, lds_addr(tile_b) + tr_off};
        uint32_t lo = (uint32_t)L.state, hi = (uint32_t)(L.state >> 32);
        int32_t smin = a.min_symbol, smax = a.min_symbol;
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the statement keeps its own book from here
        ans_encode_wide_tiles_loop(lo, hi, L.out.wr, L.out.flushed, smin, smax, row_addr, tr_addr, L.out.lane_addr, L.out.cap,
                                   (uint32_t)slab_off, lds_addr(table) - 16u * (uint32_t)a.min_symbol, (uint32_t)P, a.words, symbols_base,
                                   (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)n_t), goff);
        L.state = ((uint64_t)hi << 32) | lo;
        // a symbol below min_symbol wraps to a huge index
        L.bad = max(L.bad, max((uint32_t)smax - (uint32_t)a.min_symbol, (uint32_t)smin - (uint32_t)a.min_symbol));
        ragged(0, pre);
        tb = 0;
    } else {
        // ragged top part [32 * n_full, N): direct reads, at most 31 symbols per stream (the coder runs backwards)
        for (size_t t = N; t > n_full * kTileSyms;) {
            --t;
            code(row[t * step_t]);
            L.flush_chunks();
        }
        if (n_full > 0) {
            if (SM && N < (1u << 24) && !__any(!ok) && s0 + kWave <= a.n_streams && a.n_streams % 4 == 0 && a.n_streams < (1u << 24) &&
                (reinterpret_cast<uintptr_t>(a.symbols) & 15) == 0) {
                uint32_t goff[8];
    #pragma unroll
                for (int k = 0; k < 8; ++k)
                    goff[k] = (uint32_t)((((size_t)(lane >> 2) + 16 * (k & 1)) * a.n_streams + 16 * (size_t)(k >> 1) + 4 * (size_t)(lane & 3)) * 4);
                const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(a.symbols + (n_full - 1) * kTileSyms * a.n_streams + s0);
                const uint64_t symbols_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                              (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
                const uint32_t tr_off = (uint32_t)(((4 * (lane & 3)) * kTileStride + (lane >> 2)) * 4);
                int32_t* tile_b = tile + (kBlock / kWave) * (kWave * kTileStride);
                const uint32_t row_addr[2] = {lds_addr(tile + lane * kTileStride), lds_addr(tile_b + lane * kTileStride)};
                const uint32_t tr_addr[2] = {lds_addr(tile) + tr_off, lds_addr(tile_b) + tr_off};
                uint32_t lo = (uint32_t)L.state, hi = (uint32_t)(L.state >> 32);
                int32_t smin = a.min_symbol, smax = a.min_symbol;
                __builtin_amdgcn_s_waitcnt(0x0F70);
                ans_encode_wide_tiles_loop_sm(lo, hi, L.out.wr, L.out.flushed, smin, smax, row_addr, tr_addr, L.out.lane_addr, L.out.cap,
                                              (uint32_t)slab_off, lds_addr(table) - 16u * (uint32_t)a.min_symbol, (uint32_t)P, a.words, symbols_base,
                                              (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)n_full),
                                              (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(kTileSyms * a.n_streams * 4)), goff);
                L.state = ((uint64_t)hi << 32) | lo;
                L.bad = max(L.bad, max((uint32_t)smax - (uint32_t)a.min_symbol, (uint32_t)smin - (uint32_t)a.min_symbol));
                tb = 0;
            }
        }
    }
    // partial waves and odd slabs: tile by tile with the per-step statement
    if constexpr (SM) {
        for (size_t t = tb * kTileSyms; t > 0;) {
            --t;
            code(row[t * step_t]);
            if ((t & 7) == 0) L.flush_chunks();
        }
        tb = 0;
    }
    for (; tb > 0;) {
        --tb;
        int32_t r[kTileSyms];
        tile_fetch<true>(a.symbols, a.n_streams, N, s0, tb * kTileSyms, lane, r);
        wave_lds_fence();
        tile_to_lds<true>(tile, lane, r);
        wave_lds_fence();
        const int32_t* my = tile + min((size_t)lane, a.n_streams - 1 - s0) * kTileStride;   // (repeating lanes read the last stream's row)
#pragma unroll 8
        for (int j = kTileSyms - 1; j >= 0; --j) {
            code(my[j]);
            if ((j & 7) == 0) L.flush_chunks();
        }
    }
    uint32_t n_words = 0;
    const int32_t status = L.finish(true, nsym, n_words);
    if (!active) return;
    a.status[s] = status;
    a.n_words[s] = (status == CST_STREAM_OK) ? n_words : 0u;
}

bool wide_encode_usable(const AnsEncodeArgs& a, cst_coder_config cfg, cst_layout layout) {
    if (layout == CST_LAYOUT_SYMBOL_MAJOR && (a.n_streams % 4 != 0 || a.n_streams < (size_t)kWave)) return false;
    return cfg.word_bits == 32 && a.precision > 12 && a.precision <= 24 &&
           !(a.flags & CST_FLAG_RAW_STATE) && (layout == CST_LAYOUT_SYMBOL_MAJOR || a.n_per_stream % 4 == 0) &&
           (reinterpret_cast<uintptr_t>(a.symbols) & 15) == 0 &&
           kWideRingBytes + (size_t)a.n_symbols * sizeof(EncEntry) + 2 * kWideTileBytes <= 160 * 1024;
}

cst_status ans_encode_wide(const AnsEncodeArgs& a, cst_layout layout, hipStream_t hs) {
    const size_t lds = kWideRingBytes + (size_t)a.n_symbols * sizeof(EncEntry) + 2 * kWideTileBytes;
    const size_t blocks = (a.n_streams + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    auto kernel = layout == CST_LAYOUT_SYMBOL_MAJOR ? ans_encode_wide_kernel<CST_LAYOUT_SYMBOL_MAJOR> : ans_encode_wide_kernel<CST_LAYOUT_STREAM_MAJOR>;
    CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kBlock), lds, hs, a);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

// ------------------------------------------------------------------------------------------------
// int8 / int16 symbol matrices at 12 < P <= 24 (round 5; GEN_B16_NARROW of scripts/gen_decode_loop_b16.py): the same statement with ONE
// byte tile per wave.  A decoded quad is packed into the lane's row (a 128-byte line of the matrix), and the group of 128 / 64
// symbols leaves behind its last tile in a block of eight row blocks.  Whole rows of lines, any number of streams (the spare lanes of
// a partial wave repeat the last stream), the words' span known; everything else converts next to ans_decode_b16_kernel.
// ------------------------------------------------------------------------------------------------
// SMALL (GEN_B16_SMALL): the small-footprint form for two waves per SIMD -- workgroups of 512 threads, word rings of 16 slots with a window every
// quarter tile, and the one tile per wave for int32 matrices too (BYTES = 4: rows of 36 words): more than 256 streams per CU (a C5-sized shard,
// the virtual streams of a batch decoded through jump points).
template <int BYTES, bool SMALL>
struct B16NarrowGeo {
    static constexpr int kThreads = SMALL ? 512 : kBlock;
    static constexpr int kSlots = SMALL ? 16 : kDecRingSlots;
    static constexpr int kAheadWords = SMALL ? 12 : kDecAhead;              // SMALL: two quarter tiles of at most 6 words each
    static constexpr uint32_t kRingMask = (uint32_t)(kSlots - 1) * kWave * 4;
    static constexpr size_t kRingBytes = (size_t)(kThreads / kWave) * kSlots * kWave * 4;
    static constexpr int kRowBytes = BYTES == 4 ? 144 : 132;               // a 128-byte line of the matrix + padding (int32: 16-byte aligned rows)
    static constexpr size_t kTileBytes = (size_t)kWave * kRowBytes;
    // bucket entries: the one-wave form has LDS to spare and tabulates 2^12 buckets (half as many quantiles beyond their entry's third
    // symbol: the out-of-line excursions cost the 2^11 form 11 % at P = 24 on C2's Gaussian); the eight-wave form keeps the model's 2^11
    static constexpr int kTableBits = SMALL ? 0 : 12;
    static_assert(BYTES != 4 || SMALL, "int32 matrices at one wave per SIMD: ans_decode_b16_kernel");
};

template <int BYTES, bool SMALL>
__device__ __forceinline__ void ans_decode_b16_narrow_loop(uint32_t& lo, uint32_t& hi, uint32_t& rd, uint32_t& lo_issued, uint32_t& row_cur,
                                                           uint32_t& row_prev, uint32_t& tr_cur, uint32_t& tr_prev, uint32_t lut_addr,
                                                           uint32_t cdf_addr, uint32_t mask, uint32_t P, uint32_t bucket_shift,
                                                           int32_t min_symbol, uint32_t ring_mask, const void* words_base, uint64_t store_base,
                                                           uint32_t n_tiles, uint32_t shift_minus_1, uint32_t ring_lane_addr, uint32_t dump_addr,
                                                           uint32_t words_off, uint32_t c_field_mask, uint32_t index_shift) {
    if constexpr (SMALL && BYTES == 1) {
#include "cst_decode_loop_b16_s8.inc"
    } else if constexpr (SMALL && BYTES == 2) {
#include "cst_decode_loop_b16_s16.inc"
    } else if constexpr (SMALL) {
#include "cst_decode_loop_b16_s32.inc"
    } else if constexpr (BYTES == 1) {
#include "cst_decode_loop_b16_n8.inc"
    } else {
#include "cst_decode_loop_b16_n16.inc"
    }
}

template <int BYTES, bool SMALL>
static size_t b16_narrow_lds_bytes(int n_symbols, int bucket_bits) {
    using G = B16NarrowGeo<BYTES, SMALL>;
    if (G::kTableBits) bucket_bits = G::kTableBits;
    return G::kRingBytes + ((b16_table_bytes(n_symbols, bucket_bits) + 15) & ~(size_t)15) + (size_t)(G::kThreads / kWave) * G::kTileBytes +
           (size_t)(G::kThreads / kWave) * 4 * kWave * 4;
}

// LDS layout: [word rings][cdf][bucket entries][second-level tables][one tile per wave][dump rows]
template <int BYTES, bool SMALL>
__global__ __launch_bounds__(SMALL ? 512 : kBlock) void ans_decode_b16_narrow_kernel(const AnsDecodeArgs a) {
    using G = B16NarrowGeo<BYTES, SMALL>;
    constexpr int kWaves = G::kThreads / kWave;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = threadIdx.x >> 6;
    const int P = a.precision;
    DecLut lut{};
    const uint32_t* cdf = a.cdf;
    const uint16_t* bucket = a.bucket;
    const size_t lds_off = stage_decoder_tables<kDecBucket, true, true>(smem + G::kRingBytes, P, a.dec_cp, a.dec_idx, a.cdf, a.bucket, a.bucket_bits,
                                                                  a.n_symbols, lut, cdf, bucket, G::kTableBits);
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem) + wave_in_block * (G::kSlots * kWave);
    unsigned char* tile = smem + G::kRingBytes + ((lds_off + 15) & ~(size_t)15) + wave_in_block * G::kTileBytes;
    uint32_t* dump = reinterpret_cast<uint32_t*>(smem + G::kRingBytes + ((lds_off + 15) & ~(size_t)15) + (size_t)kWaves * G::kTileBytes) +
                     wave_in_block * (4 * kWave) + lane;
    if ((lds_addr(ring) & (uint32_t)(G::kSlots * kWave * 4 - 1)) != 0) __builtin_trap();   // the ring address is formed with v_and_or
    __syncthreads();

    const size_t s0 = ((size_t)blockIdx.x * G::kThreads + (size_t)wave_in_block * kWave);
    if (s0 >= a.n_streams) return;
    const uint32_t last_row = (uint32_t)min((size_t)(kWave - 1), a.n_streams - 1 - s0);      // (a partial wave: the spare lanes repeat the last stream)
    const size_t s = s0 + min((uint32_t)lane, last_row);
    const size_t N = a.n_per_stream, row_bytes = N * BYTES;
    const int bucket_shift = P - (G::kTableBits ? G::kTableBits : a.bucket_bits);
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;
    int8_t* out = reinterpret_cast<int8_t*>(a.symbols);      // (BYTE addresses of the matrix)

    DecLane<32, 64, G::kSlots, G::kAheadWords> L;
    const WordSlice ws = word_slice(a.offsets, a.stride_words, a.n_words, s, a.words_capacity);
    L.init(a.words + ws.off, ws.n, ring, lane);
    if (raw) L.state = a.state[s];
    else L.read_initial_state();
    L.in.prime();
    wave_lds_fence();

    const unsigned char* words_base = reinterpret_cast<const unsigned char*>(reinterpret_cast<uintptr_t>(a.words) & ~(uintptr_t)15);
    const uint32_t w_off = (uint32_t)(reinterpret_cast<const unsigned char*>(L.in.base16) - words_base);
    // the statement reads its eight store offsets from the head of the lane's row (before the first symbols land there)
    uint32_t* my = reinterpret_cast<uint32_t*>(tile + lane * G::kRowBytes);
#pragma unroll
    for (int k = 0; k < 8; ++k) my[k] = (uint32_t)((size_t)min((uint32_t)((lane >> 3) + 8 * k), last_row) * row_bytes + 16 * (size_t)(lane & 7));
    wave_lds_fence();
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the statement keeps its own book from here
    uint32_t lo = (uint32_t)L.state, hi = (uint32_t)(L.state >> 32);
    uint32_t row_cur = lds_addr(my), row_prev = row_cur;
    uint32_t tr_cur = lds_addr(tile) + (uint32_t)((lane >> 3) * G::kRowBytes + 16 * (lane & 7)), tr_prev = tr_cur;
    const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(out + s0 * row_bytes);
    const uint64_t store_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
    ans_decode_b16_narrow_loop<BYTES, SMALL>(lo, hi, L.in.rd, L.in.lo_issued, row_cur, row_prev, tr_cur, tr_prev, lds_addr(lut.b16), lds_addr(cdf),
                                             (1u << P) - 1u, (uint32_t)P, (uint32_t)__builtin_amdgcn_readfirstlane(bucket_shift), a.min_symbol, G::kRingMask,
                                             words_base, store_base, (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(N / kTileSyms)), L.in.shift - 1u,
                                             lds_addr(ring + lane), lds_addr(dump), w_off, (1u << lut.idx_shift) - 1u, (uint32_t)lut.idx_shift);
    a.status[s] = ws.bad ? (int32_t)CST_STREAM_INVALID_DATA : L.status;
    if (raw) {
        a.state[s] = ((uint64_t)hi << 32) | lo;
        if (a.n_words_out) a.n_words_out[s] = L.in.rd;
    }
}

static bool b16_narrow_shape_ok(const AnsDecodeArgs& a, cst_coder_config cfg, cst_layout layout, int symbol_bytes) {
    if (cfg.word_bits != 32 || cfg.state_bits != 64 || layout != CST_LAYOUT_STREAM_MAJOR || a.precision <= 12 || a.precision > 24) return false;
    if (!bucket16_usable(a.n_symbols, a.precision) || !a.bucket || !a.cdf || a.n_streams == 0) return false;
    if (a.n_per_stream % (size_t)(128 / symbol_bytes) != 0 || a.n_per_stream == 0 || a.n_per_stream >= (1u << 23)) return false;
    if ((reinterpret_cast<uintptr_t>(a.symbols) & 127) != 0) return false;
    if (a.offsets && a.words_capacity == 0) return false;                                          // the lanes' 32-bit word offsets need a known span
    if ((a.flags & CST_FLAG_RAW_STATE) && !a.state) return false;
    const uint64_t span = a.offsets ? a.words_capacity : (uint64_t)a.n_streams * a.stride_words;
    return span * 4 + 256 < 0x80000000ull;
}

bool b16_narrow_decode_usable(const AnsDecodeArgs& a, cst_coder_config cfg, cst_layout layout, int symbol_bytes) {
    if (knobs().no_n8) return false;                     // (A/B runs: the conversion path)
    if (symbol_bytes != 1 && symbol_bytes != 2) return false;
    if (!b16_narrow_shape_ok(a, cfg, layout, symbol_bytes)) return false;
    return (symbol_bytes == 1 ? b16_narrow_lds_bytes<1, false>(a.n_symbols, a.bucket_bits) : b16_narrow_lds_bytes<2, false>(a.n_symbols, a.bucket_bits)) <= 160 * 1024;
}

// ... on the small-footprint form: more streams than one wave per SIMD of this device (symbol_bytes = 1, 2 or 4)
bool b16_small_decode_usable(const AnsDecodeArgs& a, cst_coder_config cfg, cst_layout layout, int symbol_bytes, int device_cus) {
    if (!knobs().small_decoders) return false;           // (A/B runs, as for cst_ans_small.hip)
    if (symbol_bytes != 1 && symbol_bytes != 2 && symbol_bytes != 4) return false;
    if (symbol_bytes != 4 && knobs().no_n8) return false;
    if (a.n_streams <= (size_t)device_cus * kBlock) return false;
    if (!b16_narrow_shape_ok(a, cfg, layout, symbol_bytes)) return false;
    const size_t lds = symbol_bytes == 1 ? b16_narrow_lds_bytes<1, true>(a.n_symbols, a.bucket_bits)
                     : symbol_bytes == 2 ? b16_narrow_lds_bytes<2, true>(a.n_symbols, a.bucket_bits) : b16_narrow_lds_bytes<4, true>(a.n_symbols, a.bucket_bits);
    return lds <= 160 * 1024;
}

template <int BYTES, bool SMALL>
static cst_status launch_b16_narrow(const AnsDecodeArgs& a, hipStream_t hs) {
    using G = B16NarrowGeo<BYTES, SMALL>;
    const size_t lds = b16_narrow_lds_bytes<BYTES, SMALL>(a.n_symbols, a.bucket_bits);
    const size_t blocks = (a.n_streams + G::kThreads - 1) / G::kThreads;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    auto kernel = ans_decode_b16_narrow_kernel<BYTES, SMALL>;
    CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(G::kThreads), lds, hs, a);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

cst_status ans_decode_b16_narrow(const AnsDecodeArgs& a, int symbol_bytes, hipStream_t hs) {
    return symbol_bytes == 1 ? launch_b16_narrow<1, false>(a, hs) : launch_b16_narrow<2, false>(a, hs);
}

cst_status ans_decode_b16_small(const AnsDecodeArgs& a, int symbol_bytes, hipStream_t hs) {
    return symbol_bytes == 1 ? launch_b16_narrow<1, true>(a, hs) : symbol_bytes == 2 ? launch_b16_narrow<2, true>(a, hs) : launch_b16_narrow<4, true>(a, hs);
}

bool b16_decode_usable(const AnsDecodeArgs& a, cst_coder_config cfg, cst_layout layout) {
    if (layout == CST_LAYOUT_SYMBOL_MAJOR && (a.n_streams % 4 != 0 || a.n_streams < (size_t)kWave)) return false;
    return cfg.word_bits == 32 && a.precision > 12 && a.precision <= 24 &&
           bucket16_usable(a.n_symbols, a.precision) && a.bucket && a.cdf && (!(a.flags & CST_FLAG_RAW_STATE) || a.state) &&
           (layout == CST_LAYOUT_SYMBOL_MAJOR || a.n_per_stream % 4 == 0) &&
           (reinterpret_cast<uintptr_t>(a.symbols) & 15) == 0 && b16_lds_bytes(a.n_symbols, a.bucket_bits) <= 160 * 1024;
}

cst_status ans_decode_b16(const AnsDecodeArgs& a, cst_layout layout, hipStream_t hs) {
    const size_t lds = b16_lds_bytes(a.n_symbols, a.bucket_bits);
    const size_t blocks = (a.n_streams + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    auto kernel = layout == CST_LAYOUT_SYMBOL_MAJOR ? ans_decode_b16_kernel<CST_LAYOUT_SYMBOL_MAJOR> : ans_decode_b16_kernel<CST_LAYOUT_STREAM_MAJOR>;
    CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kBlock), lds, hs, a);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

} // namespace cst
