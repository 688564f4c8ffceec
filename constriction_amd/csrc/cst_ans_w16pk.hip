// cst_ans_w16pk.hip -- the (16,32) preset (SmallAnsCoder, src/stream/stack.rs:153) with PACKED compressed words
// (CST_FLAG_PACKED_W16, ABI 4): the reference holds the words of this preset in a Vec<u16>; everywhere else in this library a
// compressed word occupies a uint32 slot, which makes the (16,32) kernels move twice the word bytes of the (32,64) ones
// (HBM traffic 1.15 - 1.19 x algorithmic, profiles/r04_pmc_summary.md) and issue twice the chunk loads (what the decoder
// waits for: SQ_WAIT_INST_ANY 55 - 60 cycles per symbol).  With the flag the slabs, the packed buffer of the compaction and
// every offset / count / capacity are in units of 16-bit words: stream s owns words16[s * stride .. + stride), exactly the
// bytes of the reference's Vec<u16>, little endian.
//   encode   the loop of cst_ans_w16.hip with another flush (scripts/gen_encode_loop_w16.py, GEN_W16_PACKED): the LDS ring keeps
//            one word per 32-bit slot, a 64-byte group of THIRTY-TWO words leaves per tile, pairs packed by v_lshl_or_b32
//   decode   the loop of cst_ans_w16.hip with 8-word chunks (scripts/gen_decode_loop_w16.py, GEN_W16_PACKED): half the chunk loads
// Shapes the statements do not take (rows that are not whole tiles, unaligned slabs, ...) are coded symbol by symbol: correct,
// slow.  Stream-major, shared table, 8 <= P <= 12.  Same recurrences (stack.rs:1035-1045, 1084-1097), same words as every
// other (16,32) path of the library -- tests/test_gpu_packed_w16.py compares them.
#include "cst_ans_kernels.hpp"

namespace cst {

struct __attribute__((aligned(16))) W16PkEntry { uint32_t c_ck, p, m, pshl; };   // as W16Entry of cst_ans_w16.hip

__device__ __forceinline__ void ans_encode_w16pk_tiles_loop(uint32_t& st, uint32_t& wr, uint32_t& flushed, int32_t& smin, int32_t& smax,
                                                            const uint32_t (&tile_row_addr)[2], const uint32_t (&tile_tr_addr)[2],
                                                            uint32_t ring_lane_addr, uint32_t cap, uint32_t slab_off, uint32_t table_addr_biased,
                                                            uint32_t P, const void* words_base, uint64_t symbols_base, uint32_t n_tiles,
                                                            const uint32_t (&goff)[8]) {
#include "cst_encode_loop_w16_pk.inc"
}

// ... the same with jump points (GEN_W16_CK=1 of the generator, round 6): every ck_tiles tiles the lanes store (wr, state) at element
// ck_index of the two jump arrays and step to the chunk in front
__device__ __forceinline__ void ans_encode_w16pk_tiles_loop_ck(uint32_t& st, uint32_t& wr, uint32_t& flushed, int32_t& smin, int32_t& smax, uint32_t& ck_index,
                                                               const uint32_t (&tile_row_addr)[2], const uint32_t (&tile_tr_addr)[2],
                                                               uint32_t ring_lane_addr, uint32_t cap, uint32_t slab_off, uint32_t table_addr_biased,
                                                               uint32_t P, const void* words_base, uint64_t symbols_base, uint32_t n_tiles,
                                                               const void* ck_pos_base, const void* ck_state_base, uint32_t ck_tiles,
                                                               const uint32_t (&goff)[8]) {
#include "cst_encode_loop_w16_pk_ck.inc"
}

__device__ __forceinline__ void ans_decode_w16pk_tiles_loop(uint32_t& st, uint32_t& rd, uint32_t& lo_issued, uint32_t& row_cur, uint32_t& row_prev,
                                                            uint32_t& tr_cur, uint32_t& tr_prev, uint32_t lut_addr, uint32_t mask, uint32_t P,
                                                            uint32_t ring_mask, const void* words_base, uint64_t store_base,
                                                            uint32_t n_tiles, uint32_t shift_minus_1, uint32_t ring_lane_addr, uint32_t dump_addr,
                                                            uint32_t words_off, bool plain_stores) {
    if (plain_stores) {          // rows that are not cache-line aligned: see scripts/gen_decode_loop.py (CST_STORE_MOD)
#define CST_STORE_MOD ""
#include "cst_decode_loop_w16_pk.inc"
#undef CST_STORE_MOD
    } else {
#define CST_STORE_MOD "nt"
#include "cst_decode_loop_w16_pk.inc"
#undef CST_STORE_MOD
    }
}

constexpr size_t kPkEncRingBytes = (size_t)(kBlock / kWave) * kRingWords * 4;                // 64 slots x 64 lanes x 4 B per wave
constexpr size_t kPkEncTileBytes = (size_t)(kBlock / kWave) * kWave * kTileStride * 4;

// jump points (Pos, stack.rs:1107-1139): [n_streams][n_chunks] (16-bit words in the bulk, state) in front of every chunk of `interval`
// symbols; n_per_stream = interval * n_chunks, interval a multiple of the tile (the launcher checks)
struct PkJumpArgs { uint32_t* pos; uint64_t* state; uint32_t interval, n_chunks; };

// LDS layout: [word rings, 16 KiB per wave][table][symbol tiles A][symbol tiles B]
template <bool CK>
__global__ __launch_bounds__(kBlock) void ans_encode_w16pk_kernel(const AnsEncodeArgs a, const PkJumpArgs jp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = threadIdx.x >> 6;
    const int P = a.precision;
    const size_t table_bytes = (size_t)a.n_symbols * sizeof(W16PkEntry);
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem) + wave_in_block * kRingWords;
    W16PkEntry* table = reinterpret_cast<W16PkEntry*>(smem + kPkEncRingBytes);
    int32_t* tile = reinterpret_cast<int32_t*>(smem + kPkEncRingBytes + table_bytes) + wave_in_block * (kWave * kTileStride);
    if ((lds_addr(ring) & (kRingWords * 4u - 1u)) != 0) __builtin_trap();   // the ring address is formed with v_and_or
    // the statement writes its words with ds_write_b16 and packs pairs with v_lshl_or: the upper halves of the slots stay zero
    for (int i = threadIdx.x; i < (kBlock / kWave) * kRingWords; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = 0u;
    for (int i = threadIdx.x; i < a.n_symbols; i += blockDim.x) {
        const EncEntry e = a.enc[i];
        table[i] = W16PkEntry{e.c | ((e.c + (1u << P) - e.p) << 16), e.p, e.m_hi, e.p << (32 - P)};
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
    const size_t se = active ? s : a.n_streams - 1;          // the lanes beyond the last stream repeat it (same symbols, same words)
    uint16_t* slab = reinterpret_cast<uint16_t*>(a.words) + se * a.stride_words;
    const uint32_t cap = (uint32_t)(a.stride_words > 0xffffffffull ? 0xffffffffull : a.stride_words);      // 16-bit words
    const int32_t* row = a.symbols + se * N;

    uint32_t st = 0, wr = 0, flushed = 0, bad = 0;
    bool done = false;
    const uint64_t slab_off = (uint64_t)(reinterpret_cast<const unsigned char*>(slab) - reinterpret_cast<const unsigned char*>(a.words));
    const bool ok = slab_off + 2ull * cap < 0x100000000ull && (reinterpret_cast<uintptr_t>(slab) & 63) == 0 && (cap & 31u) == 0;
    if (n_full > 0 && N % kTileSyms == 0 && N < (1u << 24) && (reinterpret_cast<uintptr_t>(a.symbols) & 15) == 0 && !__any(!ok)) {
        const size_t last_row = min((size_t)kWave, a.n_streams - s0) - 1;
        uint32_t goff[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) goff[k] = (uint32_t)((min((size_t)(lane >> 3) + 8 * k, last_row) * N + 4 * (size_t)(lane & 7)) * 4);
        const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(a.symbols + s0 * N + (n_full - 1) * kTileSyms);
        const uint64_t symbols_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                      (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
        const uint32_t tr_off = (uint32_t)(((lane >> 3) * kTileStride + 4 * (lane & 7)) * 4);
        int32_t* tile_b = tile + (kBlock / kWave) * (kWave * kTileStride);
        const uint32_t row_addr[2] = {lds_addr(tile + lane * kTileStride), lds_addr(tile_b + lane * kTileStride)};
        const uint32_t tr_addr[2] = {lds_addr(tile) + tr_off, lds_addr(tile_b) + tr_off};
        int32_t smin = a.min_symbol, smax = a.min_symbol;
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the statement keeps its own book from here
        if constexpr (CK) {
            uint32_t ck_index = (uint32_t)(se * jp.n_chunks + jp.n_chunks - 1);       // the last chunk's point is noted first
            ans_encode_w16pk_tiles_loop_ck(st, wr, flushed, smin, smax, ck_index, row_addr, tr_addr, lds_addr(ring + lane), cap, (uint32_t)slab_off,
                                           lds_addr(table) - 16u * (uint32_t)a.min_symbol, (uint32_t)P, a.words, symbols_base,
                                           (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)n_full), jp.pos, jp.state,
                                           (uint32_t)__builtin_amdgcn_readfirstlane(jp.interval / (uint32_t)kTileSyms), goff);
        } else
        ans_encode_w16pk_tiles_loop(st, wr, flushed, smin, smax, row_addr, tr_addr, lds_addr(ring + lane), cap, (uint32_t)slab_off,
                                    lds_addr(table) - 16u * (uint32_t)a.min_symbol, (uint32_t)P, a.words, symbols_base,
                                    (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)n_full), goff);
        bad = max((uint32_t)smax - (uint32_t)a.min_symbol, (uint32_t)smin - (uint32_t)a.min_symbol);   // a symbol below min_symbol wraps
        // what is still in the ring (fewer than 56 words per lane): straight to the slab
        wave_lds_fence();
        for (uint32_t i = flushed; i != wr; ++i)
            if (active && i < cap) slab[i] = (uint16_t)ring[(i & (kRingSlots - 1)) * kWave + lane];
        done = true;
    }
    if (!done) {
        // any other shape: symbol by symbol (encode_symbol, stack.rs:1035-1045, on a 32-bit state), words straight to the slab
        for (size_t t = N; t-- > 0;) {
            const EncEntry e = a.enc[enc_index(row[t], a.min_symbol, nsym, bad)];
            if ((st >> (32 - P)) >= e.p) {
                if (active && wr < cap) slab[wr] = (uint16_t)st;
                ++wr;
                st >>= 16;
            }
            st = ((st / e.p) << P) + e.c + st % e.p;
            if constexpr (CK) {
                if (active && t % jp.interval == 0) { jp.pos[s * jp.n_chunks + t / jp.interval] = wr; jp.state[s * jp.n_chunks + t / jp.interval] = st; }
            }
        }
    }
    // into_compressed: the state's words, least significant first, zero high words dropped (lib.rs:719-731)
    uint32_t n_words = wr;
    for (uint32_t rest = st; rest != 0; rest >>= 16) {
        if (active && n_words < cap) slab[n_words] = (uint16_t)rest;
        ++n_words;
    }
    if (!active) return;
    const int32_t status = bad >= nsym ? CST_STREAM_IMPOSSIBLE_SYMBOL : (n_words > cap ? CST_STREAM_CAPACITY : CST_STREAM_OK);
    a.status[s] = status;
    a.n_words[s] = (status == CST_STREAM_OK) ? n_words : 0u;
}

// ------------------------------------------------------------------------------------------------
// decoder
// ------------------------------------------------------------------------------------------------
constexpr int kPkRingWords = 64;              // 16-bit words of ring per lane: [position][lane] halfwords, 8 KiB per wave
constexpr int kPkAhead = 44;                  // 12 words of a half tile + 24 until requested chunks have landed + a chunk of 8
constexpr uint32_t kPkRingMask = (kPkRingWords - 1) * kWave * 2;
constexpr size_t kPkRingBytes = (size_t)(kBlock / kWave) * kPkRingWords * kWave * 2;
constexpr size_t kPkTileWords = (size_t)kWave * kTileStride;
constexpr size_t kPkDumpBytes = (size_t)(kBlock / kWave) * 2048;        // a chunk lands as 8 halfword rows of 128 B: 1 KiB + lane offsets
constexpr size_t kPkLdsBytes = kPkRingBytes + kTileLutBytes + 2 * (size_t)(kBlock / kWave) * kPkTileWords * 4 + kPkDumpBytes;

struct W16PkLane {
    uint32_t state;
    int32_t status;
    uint32_t rd;           // words not yet consumed (next word has stream index rd - 1)
    uint32_t shift;        // 16-bit words between the 16-byte boundary below the stream's first word and that word (0 .. 7)
    uint32_t lo_issued;    // lowest position (multiple of 8) whose chunk is in the ring
    const uint16_t* base16;
    uint16_t* ring;
    int lane;

    __device__ __forceinline__ uint16_t* slot(uint32_t pos) const { return ring + ((pos & (kPkRingWords - 1)) * kWave + lane); }

    __device__ __forceinline__ void init(const uint16_t* in, uint32_t len, uint16_t* wave_ring, int lane_) {
        shift = (uint32_t)((reinterpret_cast<uintptr_t>(in) & 15) >> 1);
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

    __device__ __forceinline__ void fill_blocking() {
        const uint32_t top = rd + shift;
        const uint32_t want_lo = top > (uint32_t)kPkAhead ? top - kPkAhead : 0u;
        while (lo_issued > want_lo) {
            lo_issued -= 8;
            const uint4 v = *reinterpret_cast<const uint4*>(base16 + lo_issued);
            uint16_t* b = slot(lo_issued);   // chunk positions are multiples of 8: the eight rows follow each other
            b[0] = (uint16_t)v.x; b[kWave] = (uint16_t)(v.x >> 16); b[2 * kWave] = (uint16_t)v.y; b[3 * kWave] = (uint16_t)(v.y >> 16);
            b[4 * kWave] = (uint16_t)v.z; b[5 * kWave] = (uint16_t)(v.z >> 16); b[6 * kWave] = (uint16_t)v.w; b[7 * kWave] = (uint16_t)(v.w >> 16);
        }
    }

    __device__ __forceinline__ void prime() {
        lo_issued = (rd + shift + 7) & ~7u;
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

// LDS layout: [word rings, 8 KiB per wave][cp | symbols (stage_tile_tables)][symbol tiles A][symbol tiles B][dump rows, 2 KiB per wave]
__global__ __launch_bounds__(kBlock) void ans_decode_w16pk_kernel(const AnsDecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = threadIdx.x >> 6;
    const int P = a.precision;
    DecLut lut{};
    stage_tile_tables(smem + kPkRingBytes, P, a.dec_cp, a.dec_idx, a.min_symbol, lut);
    uint16_t* ring = reinterpret_cast<uint16_t*>(smem) + wave_in_block * (kPkRingWords * kWave);
    int32_t* tile = reinterpret_cast<int32_t*>(smem + kPkRingBytes + kTileLutBytes) + wave_in_block * kPkTileWords;
    int32_t* tile_b = tile + (kBlock / kWave) * kPkTileWords;
    uint32_t* dump = reinterpret_cast<uint32_t*>(smem + kPkRingBytes + kTileLutBytes + 2 * (size_t)(kBlock / kWave) * kPkTileWords * 4 +
                                                 (size_t)wave_in_block * 2048) + lane;
    if ((lds_addr(ring) & (uint32_t)(kPkRingWords * kWave * 2 - 1)) != 0) __builtin_trap();   // the ring address is formed with v_and_or
    __syncthreads();

    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t s0 = wave * kWave;
    if (s0 >= a.n_streams) return;
    const size_t s = s0 + lane;
    const bool active = s < a.n_streams;
    const size_t N = a.n_per_stream;
    const size_t n_full = N / kTileSyms;
    const size_t se = active ? s : a.n_streams - 1;          // (repeat the last stream: the wave runs the statement like a full one)
    W16PkLane L;
    const WordSlice ws = word_slice(a.offsets, a.stride_words, a.n_words, se, a.words_capacity);      // all in 16-bit words
    const uint16_t* words16 = reinterpret_cast<const uint16_t*>(a.words);
    L.init(words16 + ws.off, ws.n, ring, lane);
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;
    if (raw) L.state = (uint32_t)a.state[se];
    else L.read_initial_state();
    L.prime();
    wave_lds_fence();

    int32_t* my = tile + lane * kTileStride;
    int32_t* out_row = a.symbols + se * N;
    size_t t_done = 0;
    {
        const unsigned char* words_base = reinterpret_cast<const unsigned char*>(reinterpret_cast<uintptr_t>(a.words) & ~(uintptr_t)15);
        const uint64_t w_off = (uint64_t)(reinterpret_cast<const unsigned char*>(L.base16) - words_base);
        const bool off_ok = w_off + 2ull * ((uint64_t)L.rd + 16) < 0x80000000ull;
        if (n_full >= 2 && N % kTileSyms == 0 && N < (1u << 24) && (reinterpret_cast<uintptr_t>(a.symbols) & 15) == 0 && !__any(!off_ok)) {
            const size_t last_row = min((size_t)kWave, a.n_streams - s0) - 1;
            // the first tile with the compiler-scheduled step into tile A (the window is topped up every half tile)
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
#pragma unroll 1
                for (int j = 0; j < 4; ++j) {
                    int4 v;
                    v.x = L.step(lut.cp, lut.sym, P); v.y = L.step(lut.cp, lut.sym, P); v.z = L.step(lut.cp, lut.sym, P); v.w = L.step(lut.cp, lut.sym, P);
                    *reinterpret_cast<int4*>(my + 16 * h + 4 * j) = v;
                }
                L.fill_blocking();
                wave_lds_fence();
            }
            // the statement reads its eight store offsets from the lane's row of the current tile buffer (B)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                tile_b[lane * kTileStride + k] = (int32_t)(uint32_t)((min((size_t)(lane >> 3) + 8 * k, last_row) * N + 4 * (size_t)(lane & 7)) * 4);
            wave_lds_fence();
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the statement keeps its own book from here
            const uint32_t tr_off = (uint32_t)(((lane >> 3) * kTileStride + 4 * (lane & 7)) * 4);
            uint32_t row_cur = lds_addr(tile_b + lane * kTileStride), row_prev = lds_addr(my);
            uint32_t tr_cur = lds_addr(tile_b) + tr_off, tr_prev = lds_addr(tile) + tr_off;
            const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(a.symbols + s0 * N);
            const uint64_t store_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
            const bool plain = __builtin_amdgcn_readfirstlane((int)(((N * 4) % 128 != 0 || (sb & 127) != 0) ? 1 : 0)) != 0;
            ans_decode_w16pk_tiles_loop(L.state, L.rd, L.lo_issued, row_cur, row_prev, tr_cur, tr_prev, lds_addr(lut.cp), (1u << P) - 1u, (uint32_t)P,
                                        kPkRingMask, words_base, store_base, (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(n_full - 1)),
                                        L.shift - 1u, lds_addr(ring + lane), lds_addr(dump), (uint32_t)w_off, plain);
            // the last tile is still in LDS (buffer A if it has an even index)
            wave_lds_fence();
            {
                const int32_t* last = ((n_full - 1) & 1) ? tile_b : tile;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const size_t R = min((size_t)(lane >> 3) + 8 * k, last_row);
                    const int4 v = *reinterpret_cast<const int4*>(last + ((lane >> 3) + 8 * k) * kTileStride + 4 * (lane & 7));
                    v4i t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
                    __builtin_nontemporal_store(t, reinterpret_cast<v4i*>(a.symbols + (s0 + R) * N + (n_full - 1) * kTileSyms + 4 * (lane & 7)));
                }
            }
            wave_lds_fence();
            t_done = N;
        }
    }
    for (size_t t = t_done; t < N; ++t) {           // shapes the statement does not take: symbol by symbol
        const int32_t sym = L.step(lut.cp, lut.sym, P);
        if (active) out_row[t] = sym;
        if ((t & 7) == 7) { L.fill_blocking(); wave_lds_fence(); }
    }
    if (!active) return;
    a.status[s] = ws.bad ? (int32_t)CST_STREAM_INVALID_DATA : L.status;
    if (raw) {
        a.state[s] = (uint64_t)L.state;
        if (a.n_words_out) a.n_words_out[s] = L.rd;
    }
}

// gather of the packed-16 slabs behind the offsets the scan wrote (cst_compact_words16): one wave per stream, 16-byte pieces
// of the destination assembled from halfword reads (the source and the destination of a stream differ in alignment)
__global__ __launch_bounds__(256) void gather_words16_kernel(const uint16_t* __restrict__ src, size_t stride, const uint32_t* __restrict__ n_words,
                                                              const uint64_t* __restrict__ offsets, size_t n_streams, uint16_t* __restrict__ dst,
                                                              uint64_t capacity) {
    const size_t s = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (s >= n_streams) return;
    const int lane = threadIdx.x & 63;
    const uint32_t n = n_words[s];
    const uint64_t off = offsets[s];
    if (n > stride || off + n > capacity) return;            // (a stream that would end beyond the buffer is not copied)
    const uint16_t* in = src + s * stride;
    uint16_t* out = dst + off;
    // halfwords up to the first 16-byte boundary of the DESTINATION ADDRESS (d_packed16 itself need not be 16-byte aligned)
    const uint32_t head = min(n, (uint32_t)(((16 - (reinterpret_cast<uintptr_t>(out) & 15)) & 15) >> 1));
    if ((uint32_t)lane < head) out[lane] = in[lane];
    const uint32_t n8 = (n - head) >> 3;
    for (uint32_t i = lane; i < n8; i += 64) {
        const uint16_t* p = in + head + 8 * i;
        uint4 v;
        v.x = (uint32_t)p[0] | ((uint32_t)p[1] << 16); v.y = (uint32_t)p[2] | ((uint32_t)p[3] << 16);
        v.z = (uint32_t)p[4] | ((uint32_t)p[5] << 16); v.w = (uint32_t)p[6] | ((uint32_t)p[7] << 16);
        *reinterpret_cast<uint4*>(out + head + 8 * i) = v;
    }
    const uint32_t done = head + 8 * n8;
    if (done + (uint32_t)lane < n) out[done + lane] = in[done + lane];
}

// ---- launchers (declared in cst_common.hpp) ----
bool w16pk_usable(const cst_model* m, cst_coder_config cfg, cst_layout layout) {
    return cfg.word_bits == 16 && cfg.state_bits == 32 && m->precision >= 8 && m->precision <= 12 && !m->per_stream &&
           layout == CST_LAYOUT_STREAM_MAJOR && m->d_dec_cp && m->d_dec_idx &&
           kPkEncRingBytes + (size_t)m->n_symbols * sizeof(W16PkEntry) + 2 * kPkEncTileBytes <= 160 * 1024;
}

cst_status ans_encode_w16pk(const AnsEncodeArgs& a, hipStream_t hs) {
    const size_t lds = kPkEncRingBytes + (size_t)a.n_symbols * sizeof(W16PkEntry) + 2 * kPkEncTileBytes;
    const size_t blocks = (a.n_streams + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(ans_encode_w16pk_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(ans_encode_w16pk_kernel<false>, dim3((unsigned)blocks), dim3(kBlock), lds, hs, a, PkJumpArgs{nullptr, nullptr, 0u, 0u});
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

// ... noting jump points on its way: chunks of whole tiles that divide the rows; 32-bit indices into the jump arrays
bool w16pk_ckpt_usable(const cst_model* m, cst_coder_config cfg, cst_layout layout, size_t n_streams, size_t n_per_stream, size_t interval) {
    if (!w16pk_usable(m, cfg, layout) || interval == 0 || interval % kTileSyms != 0 || n_per_stream % interval != 0) return false;
    return interval < (1u << 24) && n_streams * (n_per_stream / interval) < (1u << 28);
}

cst_status ans_encode_w16pk_ckpt(const AnsEncodeArgs& a, size_t interval, uint32_t* d_ckpt_pos, uint64_t* d_ckpt_state, hipStream_t hs) {
    const size_t lds = kPkEncRingBytes + (size_t)a.n_symbols * sizeof(W16PkEntry) + 2 * kPkEncTileBytes;
    const size_t blocks = (a.n_streams + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(ans_encode_w16pk_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(ans_encode_w16pk_kernel<true>, dim3((unsigned)blocks), dim3(kBlock), lds, hs, a,
                       PkJumpArgs{d_ckpt_pos, d_ckpt_state, (uint32_t)interval, (uint32_t)(a.n_per_stream / interval)});
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

cst_status ans_decode_w16pk(const AnsDecodeArgs& a, hipStream_t hs) {
    const size_t blocks = (a.n_streams + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(ans_decode_w16pk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPkLdsBytes));
    hipLaunchKernelGGL(ans_decode_w16pk_kernel, dim3((unsigned)blocks), dim3(kBlock), kPkLdsBytes, hs, a);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

} // namespace cst

using namespace cst;

extern "C" {

cst_status cst_ans_encode_batch_ckpt_packed16(const cst_model* model, cst_coder_config cfg, const int32_t* d_symbols, size_t n_streams, size_t n_per_stream,
                                              uint16_t* d_words16, size_t stride_words, uint32_t* d_n_words, size_t ckpt_interval,
                                              uint32_t* d_ckpt_pos, uint64_t* d_ckpt_state, int32_t* d_status, void* stream) {
    if (!model || !d_words16 || !d_n_words || !d_status || !d_ckpt_pos || !d_ckpt_state || ckpt_interval == 0) return CST_ERR_INVALID_ARGUMENT;
    if (n_per_stream > 0 && !d_symbols) return CST_ERR_INVALID_ARGUMENT;
    if (!config_supported(cfg) || cfg.precision != model->precision || model->d_symbol_of_index) return CST_ERR_INVALID_ARGUMENT;
    if (n_streams == 0) return CST_OK;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != model->device) return CST_ERR_INVALID_ARGUMENT;
    if (!w16pk_ckpt_usable(model, cfg, CST_LAYOUT_STREAM_MAJOR, n_streams, n_per_stream, ckpt_interval)) return CST_ERR_INVALID_ARGUMENT;
    AnsEncodeArgs a{};
    a.symbols = d_symbols; a.n_streams = n_streams; a.n_per_stream = n_per_stream; a.enc = model->d_enc;
    a.n_symbols = model->n_symbols; a.min_symbol = model->min_symbol; a.precision = model->precision;
    a.words = reinterpret_cast<uint32_t*>(d_words16); a.stride_words = stride_words; a.n_words = d_n_words; a.state = nullptr; a.status = d_status;
    a.flags = CST_FLAG_PACKED_W16;
    return note_kernel("ans_encode_w16pk_kernel<ckpt>", ans_encode_w16pk_ckpt(a, ckpt_interval, d_ckpt_pos, d_ckpt_state, (hipStream_t)stream));
}

cst_status cst_compact_words16(const uint16_t* d_words16, size_t stride_words, const uint32_t* d_n_words, size_t n_streams, uint64_t* d_offsets,
                               uint16_t* d_packed16, size_t packed_capacity, void* d_scratch, void* stream) {
    // the offsets are those of the 32-bit call (a prefix sum of the counts, whatever the counts count) ...
    const cst_status rc = cst_compact_words(nullptr, stride_words, d_n_words, n_streams, d_offsets, nullptr, 0, d_scratch, stream);
    if (rc != CST_OK || !d_packed16 || n_streams == 0) return rc;
    if (!d_words16) return CST_ERR_INVALID_ARGUMENT;
    // ... and the gather moves halfwords
    const size_t blocks = (n_streams * 64 + 255) / 256;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(gather_words16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_words16, stride_words, d_n_words, d_offsets,
                       n_streams, d_packed16, (uint64_t)packed_capacity);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

} // extern "C"
