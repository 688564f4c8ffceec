// cst_ans_w16.hip -- the hand-scheduled decoder of the (16,32) preset (SmallAnsCoder, src/stream/stack.rs:153),
// 8 <= P <= 12, shared table in LDS, stream-major.  scripts/gen_decode_loop_w16.py has the instruction-level account;
// the step is AnsCoder::decode_symbol, stack.rs:1084-1097, on a 32-bit state with 16-bit words (one per 32-bit slot).
#include "cst_ans_kernels.hpp"

namespace cst {

__device__ __forceinline__ void ans_decode_w16_tiles_loop(uint32_t& st, uint32_t& rd, uint32_t& lo_issued, uint32_t& row_cur, uint32_t& row_prev,
                                                          uint32_t& tr_cur, uint32_t& tr_prev, uint32_t lut_addr, uint32_t mask, uint32_t P,
                                                          uint32_t ring_mask, const void* words_base, uint64_t store_base,
                                                          uint32_t n_tiles, uint32_t shift_minus_1, uint32_t ring_lane_addr, uint32_t dump_addr,
                                                          uint32_t words_off, bool plain_stores) {
    if (plain_stores) {          // rows that are not cache-line aligned: see scripts/gen_decode_loop.py (CST_STORE_MOD)
#define CST_STORE_MOD ""
#include "cst_decode_loop_w16.inc"
#undef CST_STORE_MOD
    } else {
#define CST_STORE_MOD "nt"
#include "cst_decode_loop_w16.inc"
#undef CST_STORE_MOD
    }
}

// the same for symbols[t][stream] (scripts/gen_decode_loop_w16.py, SYMBOL_MAJOR): full waves only
__device__ __forceinline__ void ans_decode_w16_tiles_loop_sm(uint32_t& st, uint32_t& rd, uint32_t& lo_issued, uint32_t& row_cur, uint32_t& row_prev,
                                                             uint32_t& tr_cur, uint32_t& tr_prev, uint32_t lut_addr, uint32_t mask, uint32_t P,
                                                             uint32_t ring_mask, const void* words_base, uint64_t store_base,
                                                             uint32_t n_tiles, uint32_t shift_minus_1, uint32_t ring_lane_addr, uint32_t dump_addr,
                                                             uint32_t words_off, uint32_t tile_step_bytes,
                                                             bool plain_stores) {
    if (plain_stores) {
#define CST_STORE_MOD ""
#include "cst_decode_loop_w16_sm.inc"
#undef CST_STORE_MOD
    } else {
#define CST_STORE_MOD "nt"
#include "cst_decode_loop_w16_sm.inc"
#undef CST_STORE_MOD
    }
}

constexpr int kW16RingWords = 64;             // 16-bit words of ring per lane: [position][lane] halfwords, 8 KiB per wave
constexpr int kW16Ahead = 40;                 // 12 words of a half tile + 24 until requested chunks have landed + a chunk
constexpr uint32_t kW16RingMask = (kW16RingWords - 1) * kWave * 2;
constexpr size_t kW16RingBytes = (size_t)(kBlock / kWave) * kW16RingWords * kWave * 2;
constexpr size_t kW16TileWords = (size_t)kWave * kTileStride;
constexpr size_t kW16LdsBytes = kW16RingBytes + kTileLutBytes + 2 * (size_t)(kBlock / kWave) * kW16TileWords * 4 + kTileDumpBytes;

// Per-lane (16,32) decoder with its word ring (stack semantics: words are consumed from the END of the stream).
struct W16Lane {
    uint32_t state;
    int32_t status;
    uint32_t rd;           // words not yet consumed (next word has stream index rd - 1)
    uint32_t shift;
    uint32_t lo_issued;    // lowest position (multiple of 4) whose chunk is in the ring
    const uint32_t* base16;
    uint16_t* ring;        // this wave's ring
    int lane;

    __device__ __forceinline__ uint16_t* slot(uint32_t pos) const { return ring + ((pos & (kW16RingWords - 1)) * kWave + lane); }

    __device__ __forceinline__ void init(const uint32_t* in, uint32_t len, uint16_t* wave_ring, int lane_) {
        shift = (uint32_t)((reinterpret_cast<uintptr_t>(in) & 15) >> 2);
        base16 = in - shift;
        ring = wave_ring; lane = lane_; rd = len; status = CST_STREAM_OK; state = 0;
    }

    // from_compressed + read_initial_state (stack.rs:299-318, 440-462), straight from HBM
    __device__ __forceinline__ void read_initial_state() {
        if (rd == 0) return;
        const uint32_t first = base16[shift + --rd];
        if (first == 0) { status = CST_STREAM_INVALID_DATA; rd = 0; return; }
        uint32_t st = first;
        while (rd > 0) {
            st = (st << 16) | base16[shift + --rd];
            if (st >= (1u << 16)) break;
        }
        state = st;
    }

    // blocking top-up of the window from wherever it stands
    __device__ __forceinline__ void fill_blocking() {
        const uint32_t top = rd + shift;
        const uint32_t want_lo = top > (uint32_t)kW16Ahead ? top - kW16Ahead : 0u;
        while (lo_issued > want_lo) {
            lo_issued -= 4;
            const uint4 v = *reinterpret_cast<const uint4*>(base16 + lo_issued);
            uint16_t* b = slot(lo_issued);   // chunk positions are multiples of 4: the four rows follow each other
            b[0] = (uint16_t)v.x; b[kWave] = (uint16_t)v.y; b[2 * kWave] = (uint16_t)v.z; b[3 * kWave] = (uint16_t)v.w;
        }
    }

    __device__ __forceinline__ void prime() {
        lo_issued = (rd + shift + 3) & ~3u;
        fill_blocking();
    }

    // one step (stack.rs:1084-1097); returns the decoded symbol
    __device__ __forceinline__ int32_t step(const uint32_t* cp_table, const int32_t* sym_table, int P) {
        const uint32_t q = state & ((1u << P) - 1u);
        const uint32_t cp = cp_table[q];
        const int32_t sym = sym_table[q];
        const uint32_t st = (state >> P) * (cp >> 16) + (q - (cp & 0xffffu));
        const bool refill = st < (1u << 16) && rd > 0;
        const uint32_t w = *slot(rd - 1u + shift);     // ignored if no refill
        state = refill ? ((st << 16) | w) : st;
        rd -= refill ? 1u : 0u;
        return sym;
    }
};

// LDS layout: [word rings, 8 KiB per wave][cp | symbols (stage_tile_tables)][symbol tiles A][symbol tiles B][dump rows]
template <int LAYOUT>
__global__ __launch_bounds__(kBlock) void ans_decode_w16_kernel(const AnsDecodeArgs a) {
    constexpr bool SM = LAYOUT == CST_LAYOUT_SYMBOL_MAJOR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = threadIdx.x >> 6;
    const int P = a.precision;
    DecLut lut{};
    stage_tile_tables(smem + kW16RingBytes, P, a.dec_cp, a.dec_idx, a.min_symbol, lut);
    uint16_t* ring = reinterpret_cast<uint16_t*>(smem) + wave_in_block * (kW16RingWords * kWave);
    int32_t* tile = reinterpret_cast<int32_t*>(smem + kW16RingBytes + kTileLutBytes) + wave_in_block * kW16TileWords;
    int32_t* tile_b = tile + (kBlock / kWave) * kW16TileWords;
    uint32_t* dump = reinterpret_cast<uint32_t*>(smem + kW16RingBytes + kTileLutBytes + 2 * (size_t)(kBlock / kWave) * kW16TileWords * 4) +
                     wave_in_block * (4 * kWave) + lane;
    if ((lds_addr(ring) & (uint32_t)(kW16RingWords * kWave * 2 - 1)) != 0) __builtin_trap();   // the ring address is formed with v_and_or
    __syncthreads();

    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t s0 = wave * kWave;
    if (s0 >= a.n_streams) return;
    const size_t s = s0 + lane;
    const bool active = s < a.n_streams;
    const size_t N = a.n_per_stream;
    const size_t n_full = N / kTileSyms;

    // the lanes of a partial wave beyond its last stream REPEAT that stream (same words, same symbols, stored onto its row
    // again): the wave then runs the main-loop statement like a full one
    const size_t se = active ? s : a.n_streams - 1;
    W16Lane L;
    const WordSlice ws = word_slice(a.offsets, a.stride_words, a.n_words, se, a.words_capacity);
    L.init(a.words + ws.off, ws.n, ring, lane);
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;       // the coder continues from (d_state, d_n_words): AnsCoder::seek, stack.rs:1107-1139
    if (raw) L.state = (uint32_t)a.state[se];
    else L.read_initial_state();
    L.prime();
    wave_lds_fence();

    // one tile with the compiler-scheduled step into `dst` (the lane's tile row); the window is topped up every half tile
    auto tile_cxx = [&](int32_t* dst) {
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {
                int4 v;
                v.x = L.step(lut.cp, lut.sym, P); v.y = L.step(lut.cp, lut.sym, P); v.z = L.step(lut.cp, lut.sym, P); v.w = L.step(lut.cp, lut.sym, P);
                *reinterpret_cast<int4*>(dst + 16 * h + 4 * j) = v;
            }
            L.fill_blocking();
            wave_lds_fence();
        }
    };

    int32_t* my = tile + lane * kTileStride;
    size_t tb = 0;
    bool all_done = false;
    {
        const unsigned char* words_base = reinterpret_cast<const unsigned char*>(reinterpret_cast<uintptr_t>(a.words) & ~(uintptr_t)15);
        const uint64_t w_off = (uint64_t)(reinterpret_cast<const unsigned char*>(L.base16) - words_base);
        const bool off_ok = w_off + 4ull * ((uint64_t)L.rd + 8) < 0x80000000ull;
        if (SM && n_full >= 2 && N < (1u << 24) && !__any(!off_ok) && s0 + kWave <= a.n_streams && a.n_streams % 4 == 0 &&
            a.n_streams < (1u << 24) && (reinterpret_cast<uintptr_t>(a.symbols) & 15) == 0) {
            tile_cxx(my);
            __builtin_amdgcn_s_waitcnt(0x0F70);
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
            ans_decode_w16_tiles_loop_sm(L.state, L.rd, L.lo_issued, row_cur, row_prev, tr_cur, tr_prev, lds_addr(lut.cp), (1u << P) - 1u, (uint32_t)P,
                                         kW16RingMask, words_base_u, store_base,
                                         (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(n_full - 1)), L.shift - 1u, lds_addr(ring + lane),
                                         lds_addr(dump), (uint32_t)w_off,
                                         (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(kTileSyms * a.n_streams * 4)), plain);
            wave_lds_fence();
            tile_store_sm(a.symbols, a.n_streams, s0, (n_full - 1) * kTileSyms, lane, ((n_full - 1) & 1) ? tile_b : tile);
            wave_lds_fence();
            tb = n_full;
        } else if (!SM && N >= 4 * kTileSyms && N < (1u << 24) && !__any(!off_ok)) {
            // Rows of any length and alignment (row_skew, cst_ans_kernels.hpp; as in cst_ans_b16.hip): every lane first decodes
            // the `pre` symbols in front of its row's next cache-line boundary; the lanes of a partial wave beyond its last
            // stream repeat that stream's row.
            const size_t last_row = min((size_t)kWave, a.n_streams - s0) - 1;
            const size_t se = active ? s : a.n_streams - 1;
            int32_t* out_row = a.symbols + se * N;
            const uint32_t pre = row_skew(a.symbols, se, N);
            uint32_t max_pre = pre;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) max_pre = max(max_pre, (uint32_t)__shfl_xor((int)max_pre, d));
            max_pre = (uint32_t)__builtin_amdgcn_readfirstlane((int)max_pre);
            // `count` more symbols of every lane that has them, straight to HBM (the window is topped up every eight)
            auto direct = [&](uint32_t first, uint32_t have, uint32_t count) {
                for (uint32_t j = 0; j < count; ++j) {
                    if (j < have) { const int32_t sym = L.step(lut.cp, lut.sym, P); if (active) out_row[first + j] = sym; }
                    if ((j & 7) == 7) { L.fill_blocking(); wave_lds_fence(); }
                }
                L.fill_blocking();
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
            const uint32_t tr_off = (uint32_t)(((lane >> 3) * kTileStride + 4 * (lane & 7)) * 4);
            // current = B (tile 1), previous = A (tile 0)
            uint32_t row_cur = lds_addr(tile_b + lane * kTileStride), row_prev = lds_addr(my);
            uint32_t tr_cur = lds_addr(tile_b) + tr_off, tr_prev = lds_addr(tile) + tr_off;
            const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(a.symbols + s0 * N);
            const uint64_t store_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
            // (every stored segment is a whole cache line now: the streaming form of the stores in every case)
            ans_decode_w16_tiles_loop(L.state, L.rd, L.lo_issued, row_cur, row_prev, tr_cur, tr_prev, lds_addr(lut.cp), (1u << P) - 1u, (uint32_t)P,
                                      kW16RingMask, words_base, store_base,
                                      (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(n_t - 1)), L.shift - 1u, lds_addr(ring + lane),
                                      lds_addr(dump), (uint32_t)w_off, false);
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
            L.fill_blocking();
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
            if (active)
                for (int t = 0; t < kTileSyms; ++t) a.symbols[(tb * kTileSyms + t) * a.n_streams + s] = my[t];
        } else
        tile_store<true>(a.symbols, a.n_streams, N, s0, tb * kTileSyms, lane, tile);
        wave_lds_fence();
    }
    int32_t* row = SM ? a.symbols + (active ? s : 0) : a.symbols + (active ? s : 0) * N;
    const size_t step_t = SM ? a.n_streams : 1;
    for (size_t t = all_done ? N : n_full * kTileSyms; t < N; ++t) {
        const int32_t sym = L.step(lut.cp, lut.sym, P);
        if (active) row[t * step_t] = sym;
        L.fill_blocking();
        wave_lds_fence();
    }
    if (!active) return;
    a.status[s] = ws.bad ? (int32_t)CST_STREAM_INVALID_DATA : L.status;
    if (raw) {
        a.state[s] = (uint64_t)L.state;
        if (a.n_words_out) a.n_words_out[s] = L.rd;
    }
}

// ------------------------------------------------------------------------------------------------
// Encoder (scripts/gen_encode_loop_w16.py)
// ------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) W16Entry { uint32_t c_ck, p, m, pshl; };   // c | (c + 2^P - p) << 16, p, floor(2^32 / p), p << (32 - P)

__device__ __forceinline__ void ans_encode_w16_tiles_loop(uint32_t& st, uint32_t& wr, uint32_t& flushed, int32_t& smin, int32_t& smax,
                                                          const uint32_t (&tile_row_addr)[2], const uint32_t (&tile_tr_addr)[2],
                                                          uint32_t ring_lane_addr, uint32_t cap, uint32_t slab_off, uint32_t table_addr_biased,
                                                          uint32_t P, const void* words_base, uint64_t symbols_base, uint32_t n_tiles,
                                                          const uint32_t (&goff)[8]) {
#include "cst_encode_loop_w16.inc"
}

// the same for symbols[t][stream] (staging as in ans_encode_tiles_loop_sm, cst_ans_asm.hpp)
__device__ __forceinline__ void ans_encode_w16_tiles_loop_sm(uint32_t& st, uint32_t& wr, uint32_t& flushed, int32_t& smin, int32_t& smax,
                                                             const uint32_t (&tile_row_addr)[2], const uint32_t (&tile_tr_addr)[2],
                                                             uint32_t ring_lane_addr, uint32_t cap, uint32_t slab_off, uint32_t table_addr_biased,
                                                             uint32_t P, const void* words_base, uint64_t symbols_base, uint32_t n_tiles,
                                                             uint32_t tile_step_bytes, const uint32_t (&goff)[8]) {
#include "cst_encode_loop_w16_sm.inc"
}

constexpr size_t kW16EncRingBytes = (size_t)(kBlock / kWave) * kRingWords * 4;
constexpr size_t kW16EncTileBytes = (size_t)(kBlock / kWave) * kWave * kTileStride * 4;

// LDS layout: [word rings, 16 KiB per wave][table][symbol tiles A][symbol tiles B]
template <int LAYOUT>
__global__ __launch_bounds__(kBlock) void ans_encode_w16_kernel(const AnsEncodeArgs a) {
    constexpr bool SM = LAYOUT == CST_LAYOUT_SYMBOL_MAJOR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = threadIdx.x >> 6;
    const int P = a.precision;
    const size_t table_bytes = (size_t)a.n_symbols * sizeof(W16Entry);
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem) + wave_in_block * kRingWords;
    W16Entry* table = reinterpret_cast<W16Entry*>(smem + kW16EncRingBytes);
    int32_t* tile = reinterpret_cast<int32_t*>(smem + kW16EncRingBytes + table_bytes) + wave_in_block * (kWave * kTileStride);
    if ((lds_addr(ring) & (kRingWords * 4u - 1u)) != 0) __builtin_trap();   // the ring address is formed with v_and_or
    // the statement writes its words with ds_write_b16: the upper halves of the slots must be (and stay) zero
    for (int i = threadIdx.x; i < (kBlock / kWave) * kRingWords; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = 0u;
    for (int i = threadIdx.x; i < a.n_symbols; i += blockDim.x) {
        const EncEntry e = a.enc[i];
        table[i] = W16Entry{e.c | ((e.c + (1u << P) - e.p) << 16), e.p, e.m_hi, e.p << (32 - P)};
    }
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
    EncLane<16, 32> L;
    L.init(a.words + se * a.stride_words, (uint32_t)(a.stride_words > 0xffffffffull ? 0xffffffffull : a.stride_words), ring, lane);
    auto code = [&](int32_t v) { L.template step<false>(a.enc[enc_index(v, a.min_symbol, nsym, L.bad)], P); };

    const int32_t* row = SM ? a.symbols + se : a.symbols + se * N;
    const size_t step_t = SM ? a.n_streams : 1;
    const uint64_t slab_off = (uint64_t)(reinterpret_cast<const unsigned char*>(L.out.base16) - reinterpret_cast<const unsigned char*>(a.words));
    const bool ok = slab_off + 4ull * L.out.cap < 0x100000000ull && (reinterpret_cast<uintptr_t>(L.out.base16) & 63) == 0 &&
                    (L.out.cap & 15u) == 0 && L.out.shift == 0;
    size_t tb = n_full;
    if (!SM && N >= 4 * kTileSyms && N < (1u << 24) && !__any(!ok)) {
        // Rows of any length and alignment (row_skew, cst_ans_kernels.hpp; as in the encoder of cst_ans_b16.hip): lane l's tiles
        // start row_skew() symbols into its row; the ragged ends of the row go through LDS in bulk reads.  The lanes of a
        // partial wave beyond its last stream repeat that stream.
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
        uint32_t st = (uint32_t)L.state;
        int32_t smin = a.min_symbol, smax = a.min_symbol;
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the statement keeps its own book from here
        ans_encode_w16_tiles_loop(st, L.out.wr, L.out.flushed, smin, smax, row_addr, tr_addr, L.out.lane_addr, L.out.cap, (uint32_t)slab_off,
                                  lds_addr(table) - 16u * (uint32_t)a.min_symbol, (uint32_t)P, a.words, symbols_base,
                                  (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)n_t), goff);
        L.state = st;
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
                uint32_t st = (uint32_t)L.state;
                int32_t smin = a.min_symbol, smax = a.min_symbol;
                __builtin_amdgcn_s_waitcnt(0x0F70);
                ans_encode_w16_tiles_loop_sm(st, L.out.wr, L.out.flushed, smin, smax, row_addr, tr_addr, L.out.lane_addr, L.out.cap, (uint32_t)slab_off,
                                             lds_addr(table) - 16u * (uint32_t)a.min_symbol, (uint32_t)P, a.words, symbols_base,
                                             (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)n_full),
                                             (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(kTileSyms * a.n_streams * 4)), goff);
                L.state = st;
                L.bad = max(L.bad, max((uint32_t)smax - (uint32_t)a.min_symbol, (uint32_t)smin - (uint32_t)a.min_symbol));
                tb = 0;
            }
        }
    }
    // partial waves and odd slabs: tile by tile with the compiler-scheduled step
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

bool w16_encode_usable(const AnsEncodeArgs& a, cst_coder_config cfg, cst_layout layout) {
    if (layout == CST_LAYOUT_SYMBOL_MAJOR && (a.n_streams % 4 != 0 || a.n_streams < (size_t)kWave)) return false;
    return cfg.word_bits == 16 && cfg.state_bits == 32 && a.precision >= 8 && a.precision <= 12 &&
           !(a.flags & CST_FLAG_RAW_STATE) && (layout == CST_LAYOUT_SYMBOL_MAJOR || a.n_per_stream % 4 == 0) &&
           (reinterpret_cast<uintptr_t>(a.symbols) & 15) == 0 &&
           kW16EncRingBytes + (size_t)a.n_symbols * sizeof(W16Entry) + 2 * kW16EncTileBytes <= 160 * 1024;
}

cst_status ans_encode_w16(const AnsEncodeArgs& a, cst_layout layout, hipStream_t hs) {
    const size_t lds = kW16EncRingBytes + (size_t)a.n_symbols * sizeof(W16Entry) + 2 * kW16EncTileBytes;
    const size_t blocks = (a.n_streams + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    auto kernel = layout == CST_LAYOUT_SYMBOL_MAJOR ? ans_encode_w16_kernel<CST_LAYOUT_SYMBOL_MAJOR> : ans_encode_w16_kernel<CST_LAYOUT_STREAM_MAJOR>;
    CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kBlock), lds, hs, a);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

bool w16_decode_usable(const AnsDecodeArgs& a, cst_coder_config cfg, cst_layout layout) {
    if (layout == CST_LAYOUT_SYMBOL_MAJOR && (a.n_streams % 4 != 0 || a.n_streams < (size_t)kWave)) return false;
    return cfg.word_bits == 16 && cfg.state_bits == 32 && a.precision >= 8 && a.precision <= 12 &&
           a.dec_cp && a.dec_idx && (!(a.flags & CST_FLAG_RAW_STATE) || a.state) && (layout == CST_LAYOUT_SYMBOL_MAJOR || a.n_per_stream % 4 == 0) &&
           (reinterpret_cast<uintptr_t>(a.symbols) & 15) == 0;
}

cst_status ans_decode_w16(const AnsDecodeArgs& a, cst_layout layout, hipStream_t hs) {
    const size_t blocks = (a.n_streams + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    auto kernel = layout == CST_LAYOUT_SYMBOL_MAJOR ? ans_decode_w16_kernel<CST_LAYOUT_SYMBOL_MAJOR> : ans_decode_w16_kernel<CST_LAYOUT_STREAM_MAJOR>;
    CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kW16LdsBytes));
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kBlock), kW16LdsBytes, hs, a);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

} // namespace cst
