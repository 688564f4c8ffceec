// cst_ans_pt.hip -- batched ANS with ONE MODEL PER STREAM (BASELINE config C3), compact tables.
//
// Every lane codes its stream with its own quantized-Gaussian table (learned-image-compression latents: one (mean, std)
// per channel / tile).  What limits this shape is LDS: 256 full rows of a 255-symbol support do not fit next to the word
// rings and symbol tiles of four waves, and a workgroup of fewer waves leaves SIMDs idle.  The tables are therefore
// kept COMPACT (cst::PtMeta, built once per model by cst_model.hip): such a distribution is mostly runs of unit
// probabilities, which need no table at all.  A workgroup of 256 streams then needs ~30-45 KB for its rows and runs four
// waves per CU with rings and tiles like the shared-table coder.
//
//   encode  (stack.rs:1014-1048)   16-bit cumulatives of the symbols a..b between the unit runs at both ends, indexed by
//           symbol: i -> t = clamp(i, a, b), two 16-bit LDS reads (c[t], c[t+1]),
//           c = c[t] + (i - t), p = c[t+1] - c[t]; m = floor(2^64 / p) from a reciprocal table shared by the workgroup.
//   decode  (stack.rs:1070-1100)   entries  c << 20 | (p-1) << 8 | index  sorted by c, every run of unit probabilities
//           folded into ONE entry; quantile q -> l1[q >> (P-7)] = the 16-byte aligned quad of entries that holds the first
//           candidate, EIGHT consecutive entries from there (two ds_read_b128): the answer is the entry of the smallest wrapping
//           distance (q << 20 | 0xffffe) - entry among the first seven, provided the eighth lies above the key; otherwise the lane
//           continues four entries further in a wave-uniform loop (the narrow bins next to the unit runs).
//           (Round 6: until then six entries from the aligned PAIR -- ds_read2_b64 + ds_read_b64.  A wave's LDS read with random
//           lane addresses costs ~13 LDS cycles per CU whatever its width, ds_read2_b64 29: the six entries cost 37 cycles, the
//           eight 23 -- scripts/microbench/lds_tput.hip -- and the sub-lane decoder, LDS-bound at sixteen waves, went from 0.41 to
//           0.35 ms on C3.)
// Shapes this file does not take (P > 12 or P < 8, more than 256 symbols, the 16-bit word preset, symbol-major
// matrices, blocks whose rows do not fit in LDS) stay on the full-row kernels of cst_ans_ps.hip.
#include <type_traits>

#include "cst_ans_kernels.hpp"

namespace cst {

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

constexpr int kPtRingSlots = 32;                       // P <= 12: at most 12 words per 32-symbol tile (+ <= 15 pending)
constexpr int kPtAhead = 24;
constexpr size_t kPtRingBytes = (size_t)(kBlock / kWave) * kPtRingSlots * kWave * 4;          // 32 KiB
constexpr size_t kPtTileBytes = (size_t)(kBlock / kWave) * kWave * kTileStride * 4;           // 36 KiB
constexpr size_t kPtDumpBytes = (size_t)(kBlock / kWave) * 4 * kWave * 4;                     // decoder: landing rows
constexpr size_t kPtL1Bytes = (size_t)kBlock * kPtBuckets;                                    // 32 KiB

struct PtArgs {
    const int32_t* symbols_in;
    int32_t* symbols_out;
    size_t n_streams, n_per_stream;
    int32_t precision, n_symbols, min_symbol;
    const PtMeta* meta;
    const uint16_t* rows_enc;
    const uint32_t* rows_dec;
    const uint8_t* l1;
    const uint32_t* block_base;    // of the rows in use
    const uint64_t* recip;
    uint32_t* words_out;
    const uint32_t* words_in;
    const uint64_t* offsets;
    size_t stride_words;
    uint32_t* n_words_out_enc;
    const uint32_t* n_words_in;
    uint32_t* n_words_left;
    uint64_t* state;
    int32_t* status;
    uint32_t flags;
    uint64_t words_capacity;   // decode: uint32 slots behind `words_in` (0 = unknown): see word_slice
    // jump points (Pos / Seek, stack.rs:1107-1139): [n_streams][n_chunks] (words in the bulk, coder state) in front of every
    // chunk of `interval` symbols; written by the checkpointing encoder, read by the sub-lane decoder
    uint32_t* ckpt_pos;
    uint64_t* ckpt_state;
    size_t interval, n_chunks;
    int32_t sub_shift;         // sub-lane decoder: log2(lanes per stream)
};

// Main loop of the decoder for P = 12: all full tiles of a FULL wave in one asm statement (generated, with its wait
// counts, by scripts/gen_pt_decode_loop.py).
__device__ __forceinline__ void pt_decode_tiles_loop(uint32_t& lo, uint32_t& hi, uint32_t& rd, uint32_t& lo_issued, uint32_t bucket_mask,
                                                     uint32_t ring_mask, uint32_t P, int32_t min_symbol, const void* words_base,
                                                     uint64_t store_base, uint32_t n_tiles, uint32_t l1_lane_addr, uint32_t row_addr,
                                                     uint32_t shift_minus_1, uint32_t ring_lane_addr, uint32_t dump_addr,
                                                     uint32_t words_off, uint32_t tile_row_addr, uint32_t tile_tr_addr,
                                                     const uint32_t (&goff)[8], bool plain_stores) {
    if (plain_stores) {          // rows that are not cache-line aligned: see scripts/gen_decode_loop.py (CST_STORE_MOD)
#define CST_STORE_MOD ""
#include "cst_pt_decode_loop.inc"
#undef CST_STORE_MOD
    } else {
#define CST_STORE_MOD "nt"
#include "cst_pt_decode_loop.inc"
#undef CST_STORE_MOD
    }
}

// Main loop of the encoder: all full tiles of a FULL wave in one asm statement (generated, with its wait counts, by
// scripts/gen_pt_encode_loop.py).  sym_lo / sym_hi = first / last symbol index held by the lane's row,
// row_addr_biased = LDS address of the row word of index 0 (row address - 2 * sym_lo).
__device__ __forceinline__ void pt_encode_tiles_loop(uint32_t& lo, uint32_t& hi, uint32_t& wr, uint32_t& flushed, int32_t& smin,
                                                     int32_t& smax, uint32_t tile_row_addr, uint32_t tile_tr_addr, uint32_t ring_lane_addr,
                                                     uint32_t cap, uint32_t slab_off, uint32_t sym_lo, uint32_t sym_hi,
                                                     uint32_t row_addr_biased, uint32_t recip_addr, int32_t min_symbol, uint32_t P,
                                                     uint32_t ring_mask, const void* words_base, uint64_t symbols_base, uint32_t n_tiles,
                                                     const uint32_t (&goff)[8]) {
#include "cst_pt_encode_loop.inc"
}

// ... the same with jump points (GEN_PT_CK=1 of the generator): every ck_tiles tiles the lanes store (wr, state) at element
// ck_index of the two checkpoint arrays and step to the chunk in front
__device__ __forceinline__ void pt_encode_tiles_loop_ck(uint32_t& lo, uint32_t& hi, uint32_t& wr, uint32_t& flushed, int32_t& smin,
                                                        int32_t& smax, uint32_t& ck_index, uint32_t tile_row_addr, uint32_t tile_tr_addr,
                                                        uint32_t ring_lane_addr, uint32_t cap, uint32_t slab_off, uint32_t sym_lo,
                                                        uint32_t sym_hi, uint32_t row_addr_biased, uint32_t recip_addr, int32_t min_symbol,
                                                        uint32_t P, uint32_t ring_mask, const void* words_base, uint64_t symbols_base,
                                                        uint32_t n_tiles, const void* ck_pos_base, const void* ck_state_base,
                                                        uint32_t ck_tiles, const uint32_t (&goff)[8]) {
#include "cst_pt_encode_loop_ck.inc"
}

// ... over an INT8 symbol matrix (round 6, GEN_PT_N8=1 on top of GEN_PT_CK): a tile is 32 bytes of a row
__device__ __forceinline__ void pt_encode_tiles_loop_ck_n8(uint32_t& lo, uint32_t& hi, uint32_t& wr, uint32_t& flushed, int32_t& smin,
                                                           int32_t& smax, uint32_t& ck_index, uint32_t tile_row_addr, uint32_t tile_tr_addr,
                                                           uint32_t ring_lane_addr, uint32_t cap, uint32_t slab_off, uint32_t sym_lo,
                                                           uint32_t sym_hi, uint32_t row_addr_biased, uint32_t recip_addr, int32_t min_symbol,
                                                           uint32_t P, uint32_t ring_mask, const void* words_base, uint64_t symbols_base,
                                                           uint32_t n_tiles, const void* ck_pos_base, const void* ck_state_base,
                                                           uint32_t ck_tiles, const uint32_t (&goff)[8]) {
#include "cst_pt_encode_loop_ck_n8.inc"
}

// Main loops of the SUB-LANE decoder writing an INT8 matrix (round 6, GEN_PT_N8=1 on top of GEN_PT_SUB): a packed add of min_symbol
__device__ __forceinline__ void pt_decode_tiles_loop_sub_n8(uint32_t& lo, uint32_t& hi, uint32_t& rd, uint32_t& lo_issued, uint32_t bucket_mask,
                                                            uint32_t ring_mask, uint32_t P, int32_t min_symbol, const void* words_base,
                                                            uint64_t store_base, uint32_t n_tiles, uint32_t l1_lane_addr, uint32_t row_addr,
                                                            uint32_t shift_minus_1, uint32_t ring_lane_addr, uint32_t dump_addr,
                                                            uint32_t words_off, uint32_t tile_row_addr, uint32_t tile_tr_addr,
                                                            const uint32_t (&goff)[8]) {
#define CST_STORE_MOD ""
#include "cst_pt_decode_loop_sub_n8.inc"
#undef CST_STORE_MOD
}

__device__ __forceinline__ void pt_decode_tiles_loop_sub16_n8(uint32_t& lo, uint32_t& hi, uint32_t& rd, uint32_t& lo_issued, uint32_t bucket_mask,
                                                              uint32_t ring_mask, uint32_t P, int32_t min_symbol, const void* words_base,
                                                              uint64_t store_base, uint32_t n_tiles, uint32_t l1_lane_addr, uint32_t row_addr,
                                                              uint32_t shift_minus_1, uint32_t ring_lane_addr, uint32_t dump_addr,
                                                              uint32_t words_off, uint32_t tile_row_addr, uint32_t tile_tr_addr,
                                                              const uint32_t (&goff)[8]) {
#define CST_STORE_MOD ""
#include "cst_pt_decode_loop_sub16_n8.inc"
#undef CST_STORE_MOD
}

// Main loop of the SUB-LANE decoder (GEN_PT_SUB=1 of scripts/gen_pt_decode_loop.py): the same chain, the symbol tile in bytes
__device__ __forceinline__ void pt_decode_tiles_loop_sub(uint32_t& lo, uint32_t& hi, uint32_t& rd, uint32_t& lo_issued, uint32_t bucket_mask,
                                                         uint32_t ring_mask, uint32_t P, int32_t min_symbol, const void* words_base,
                                                         uint64_t store_base, uint32_t n_tiles, uint32_t l1_lane_addr, uint32_t row_addr,
                                                         uint32_t shift_minus_1, uint32_t ring_lane_addr, uint32_t dump_addr,
                                                         uint32_t words_off, uint32_t tile_row_addr, uint32_t tile_tr_addr,
                                                         const uint32_t (&goff)[8], bool plain_stores) {
    if (plain_stores) {
#define CST_STORE_MOD ""
#include "cst_pt_decode_loop_sub.inc"
#undef CST_STORE_MOD
    } else {
#define CST_STORE_MOD "nt"
#include "cst_pt_decode_loop_sub.inc"
#undef CST_STORE_MOD
    }
}

// ... and for sixteen waves per workgroup (GEN_PT_SUB=2): 16-slot rings, the window moves every half tile, registers v48 - v125
__device__ __forceinline__ void pt_decode_tiles_loop_sub16(uint32_t& lo, uint32_t& hi, uint32_t& rd, uint32_t& lo_issued, uint32_t bucket_mask,
                                                           uint32_t ring_mask, uint32_t P, int32_t min_symbol, const void* words_base,
                                                           uint64_t store_base, uint32_t n_tiles, uint32_t l1_lane_addr, uint32_t row_addr,
                                                           uint32_t shift_minus_1, uint32_t ring_lane_addr, uint32_t dump_addr,
                                                           uint32_t words_off, uint32_t tile_row_addr, uint32_t tile_tr_addr,
                                                           const uint32_t (&goff)[8], bool plain_stores) {
    if (plain_stores) {
#define CST_STORE_MOD ""
#include "cst_pt_decode_loop_sub16.inc"
#undef CST_STORE_MOD
    } else {
#define CST_STORE_MOD "nt"
#include "cst_pt_decode_loop_sub16.inc"
#undef CST_STORE_MOD
    }
}

// copies this block's rows into LDS (coalesced, 4 bytes per lane)
__device__ __forceinline__ void pt_stage_rows(uint32_t* rows_l, const uint32_t* src, uint32_t n_words) {
    for (uint32_t i = threadIdx.x; i < n_words; i += blockDim.x) rows_l[i] = src[i];
}

// LDS layout: [word rings, 8 KiB per wave][reciprocals 8 B x 2^P][symbol tiles][rows]
// CK: note a jump point in front of every chunk of a.interval symbols (a.ckpt_pos / a.ckpt_state)
// SB = bytes per symbol of the matrix behind a.symbols_in: 4, or 1 (int8, round 6: CK only -- the plain call is the one-chunk case)
template <bool CK, int SB = 4>
__global__ __launch_bounds__(kBlock) void ans_encode_pt_kernel(const PtArgs a) {
    static_assert(SB == 4 || (SB == 1 && CK), "int8 matrices: the checkpointing form");
    using SymT = typename std::conditional<SB == 1, int8_t, int32_t>::type;
    const SymT* symbols_in = reinterpret_cast<const SymT*>(a.symbols_in);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = threadIdx.x >> 6;
    const int P = a.precision;
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem) + wave_in_block * (kPtRingSlots * kWave);
    uint64_t* recip = reinterpret_cast<uint64_t*>(smem + kPtRingBytes);
    int32_t* tile = reinterpret_cast<int32_t*>(smem + kPtRingBytes + ((size_t)8 << P)) + wave_in_block * (kWave * kTileStride);
    uint32_t* rows_l = reinterpret_cast<uint32_t*>(smem + kPtRingBytes + ((size_t)8 << P) + kPtTileBytes);
    if ((lds_addr(ring) & (uint32_t)(kPtRingSlots * kWave * 4 - 1)) != 0) __builtin_trap();   // step<true> forms ring addresses with and/or

    for (int i = threadIdx.x; i < (1 << P); i += blockDim.x) recip[i] = a.recip[i];
    {   // 16-bit rows; blocks start at even entries (cst_model.hip), so the copy moves whole words
        const uint32_t bb = a.block_base[blockIdx.x], be = a.block_base[blockIdx.x + 1];
        pt_stage_rows(rows_l, reinterpret_cast<const uint32_t*>(a.rows_enc + bb), (be - bb + 1) / 2);
    }
    __syncthreads();

    const size_t s0 = (size_t)blockIdx.x * kBlock + (size_t)wave_in_block * kWave;
    if (s0 >= a.n_streams) return;
    const size_t s = s0 + lane;
    const bool active = s < a.n_streams;
    const size_t N = a.n_per_stream;
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;
    const uint32_t nsym = (uint32_t)a.n_symbols;
    const PtMeta mt = active ? a.meta[s] : PtMeta{0u, 0u, 0, 1, 1, 0};
    const uint32_t A = mt.a, B = (uint32_t)mt.a + mt.m - 1u;
    // cumulatives of symbol index t (A <= t <= B) and of t + 1: the 32-bit word at row_addr + 2 * (t - A)
    const uint16_t* rows16 = reinterpret_cast<const uint16_t*>(rows_l) + mt.enc_off;
    const uint32_t row_addr = lds_addr(rows16);

    EncLane<32, 64, kPtRingSlots> L;
    L.init(a.words_out + (active ? s : 0) * a.stride_words,
           active ? (uint32_t)(a.stride_words > 0xffffffffull ? 0xffffffffull : a.stride_words) : 0u, ring, lane);
    if (raw && active) L.state = a.state[s];

    auto entry_of = [&](int32_t sym) {
        const uint32_t i = enc_index(sym, a.min_symbol, nsym, L.bad);
        const uint32_t t = min(max(i, A), B);
        // (two 16-bit reads: gfx950 LDS serves a misaligned 32-bit read, but 5x slower -- scripts/microbench/lds_tput.hip)
        const uint16_t* cp = rows16 + (t - A);
        const uint32_t e = (uint32_t)cp[0] | ((uint32_t)cp[1] << 16);
        const uint32_t p = (e >> 16) - (e & 0xffffu);
        const uint64_t m = recip[p];
        return EncEntry{(e & 0xffffu) + (i - t), p, (uint32_t)m, (uint32_t)(m >> 32)};
    };

    const SymT* my = symbols_in + (active ? s : 0) * N;
    // (jump points lie on tile boundaries or the tiles are not used: chunks of 32 k symbols are the fast case)
    const bool vec = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.symbols_in) & 15) == 0) && (!CK || a.interval % kTileSyms == 0);
    const size_t n_full = vec ? N / kTileSyms : 0;
    // AnsCoder::pos() once symbols [t, N) are encoded, if a chunk starts at t
    auto note_jump_point = [&](size_t t) {
        if (CK && active && t % a.interval == 0) {
            a.ckpt_pos[s * a.n_chunks + t / a.interval] = L.out.wr;
            a.ckpt_state[s * a.n_chunks + t / a.interval] = (uint64_t)L.state;
        }
    };
    // ragged top part [32 * n_full, N): direct reads
    for (size_t t = N; t > n_full * kTileSyms;) {
        --t;
        const int32_t v = active ? (int32_t)my[t] : a.min_symbol;
        L.template step<true>(entry_of(v), P);
        L.flush_chunks();
        note_jump_point(t);
    }
    bool done = false;
    if (n_full > 0 && s0 + kWave <= a.n_streams && N < (1u << 24) && (!CK || (N % kTileSyms == 0 && a.n_streams * a.n_chunks < (1u << 28)))) {
        // ---- main loop as one asm statement (full wave, 64-byte aligned slabs of whole 64-byte groups, 32-bit offsets) ----
        const uint64_t slab_off = (uint64_t)(reinterpret_cast<const unsigned char*>(L.out.base16) - reinterpret_cast<const unsigned char*>(a.words_out));
        const bool ok = slab_off + 4ull * L.out.cap < 0x100000000ull && (reinterpret_cast<uintptr_t>(L.out.base16) & 63) == 0 &&
                        (L.out.cap & 15u) == 0 && L.out.shift == 0;
        if (!__any(!ok)) {
            uint32_t goff[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) goff[k] = (uint32_t)((((size_t)(lane >> 3) + 8 * k) * N + 4 * (size_t)(lane & 7)) * SB);
            const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(symbols_in + s0 * N + (n_full - 1) * kTileSyms);
            // wave-uniform base in SGPRs (readfirstlane returns int: go through uint32_t)
            const uint64_t symbols_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                          (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
            uint32_t lo = (uint32_t)L.state, hi = (uint32_t)((uint64_t)L.state >> 32);
            int32_t smin = a.min_symbol, smax = a.min_symbol;
            const uint32_t tr_off = (uint32_t)(((lane >> 3) * kTileStride + 4 * (lane & 7)) * 4);
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the statement keeps its own book from here
            if constexpr (CK) {
                uint32_t ck_index = (uint32_t)(s * a.n_chunks + a.n_chunks - 1);        // the last chunk's jump point comes first
                if constexpr (SB == 1)
                    pt_encode_tiles_loop_ck_n8(lo, hi, L.out.wr, L.out.flushed, smin, smax, ck_index, lds_addr(tile + lane * kTileStride),
                                               lds_addr(tile) + tr_off, L.out.lane_addr, L.out.cap, (uint32_t)slab_off, A, B, row_addr - 2u * A,
                                               lds_addr(recip), a.min_symbol, (uint32_t)P, (uint32_t)((kPtRingSlots - 1) * kWave * 4), a.words_out,
                                               symbols_base, (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)n_full), a.ckpt_pos, a.ckpt_state,
                                               (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a.interval / kTileSyms)), goff);
                else
                pt_encode_tiles_loop_ck(lo, hi, L.out.wr, L.out.flushed, smin, smax, ck_index, lds_addr(tile + lane * kTileStride),
                                        lds_addr(tile) + tr_off, L.out.lane_addr, L.out.cap, (uint32_t)slab_off, A, B, row_addr - 2u * A,
                                        lds_addr(recip), a.min_symbol, (uint32_t)P, (uint32_t)((kPtRingSlots - 1) * kWave * 4), a.words_out,
                                        symbols_base, (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)n_full), a.ckpt_pos, a.ckpt_state,
                                        (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a.interval / kTileSyms)), goff);
            } else {
                pt_encode_tiles_loop(lo, hi, L.out.wr, L.out.flushed, smin, smax, lds_addr(tile + lane * kTileStride), lds_addr(tile) + tr_off,
                                     L.out.lane_addr, L.out.cap, (uint32_t)slab_off, A, B, row_addr - 2u * A, lds_addr(recip), a.min_symbol,
                                     (uint32_t)P, (uint32_t)((kPtRingSlots - 1) * kWave * 4), a.words_out, symbols_base,
                                     (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)n_full), goff);
            }
            L.state = ((uint64_t)hi << 32) | lo;
            // fold the extremes into `bad` (largest raw table index): a symbol below min_symbol wraps to a huge index
            L.bad = max(L.bad, max((uint32_t)smax - (uint32_t)a.min_symbol, (uint32_t)smin - (uint32_t)a.min_symbol));
            done = true;
        }
    }
    if constexpr (SB == 1) {
        if (n_full > 0 && !done) {      // (a partial last wave, slabs the statement does not take: symbol by symbol, straight from the int8 rows)
            for (size_t t = n_full * kTileSyms; t-- > 0;) {
                const int32_t v = active ? (int32_t)my[t] : a.min_symbol;
                L.template step<true>(entry_of(v), P);
                L.flush_chunks();
                note_jump_point(t);
            }
        }
    } else
    if (n_full > 0 && !done) {
        int32_t r[kTileSyms];
        tile_fetch<true>(a.symbols_in, a.n_streams, N, s0, (n_full - 1) * kTileSyms, lane, r);
        const int32_t* row = tile + lane * kTileStride;
        for (size_t tb = n_full; tb-- > 0;) {
            wave_lds_fence();
            tile_to_lds<true>(tile, lane, r);
            wave_lds_fence();
            L.flush_chunks();                                   // words of the previous tile (stores before the loads)
            if (tb > 0) tile_fetch<true>(a.symbols_in, a.n_streams, N, s0, (tb - 1) * kTileSyms, lane, r);
            // the entries of quad j-1 are looked up before the steps of quad j run (nothing of it depends on the state)
            int4 v = *reinterpret_cast<const int4*>(row + 4 * (kTileSyms / 4 - 1));
            if (!active) v = make_int4(a.min_symbol, a.min_symbol, a.min_symbol, a.min_symbol);
            EncEntry e3 = entry_of(v.w), e2 = entry_of(v.z), e1 = entry_of(v.y), e0 = entry_of(v.x);
#pragma unroll
            for (int j = kTileSyms / 4 - 1; j >= 0; --j) {
                EncEntry n3 = e3, n2 = e2, n1 = e1, n0 = e0;
                if (j > 0) {
                    int4 vn = *reinterpret_cast<const int4*>(row + 4 * (j - 1));
                    if (!active) vn = make_int4(a.min_symbol, a.min_symbol, a.min_symbol, a.min_symbol);
                    n3 = entry_of(vn.w); n2 = entry_of(vn.z); n1 = entry_of(vn.y); n0 = entry_of(vn.x);
                }
                L.template step<true>(e3, P); L.template step<true>(e2, P); L.template step<true>(e1, P); L.template step<true>(e0, P);
                e3 = n3; e2 = n2; e1 = n1; e0 = n0;
            }
            note_jump_point(tb * kTileSyms);
        }
    }

    uint32_t n_words = 0;
    const int32_t status = L.finish(!raw, nsym, n_words);
    if (!active) return;
    if (raw) a.state[s] = (uint64_t)L.state;
    a.status[s] = status;
    a.n_words_out_enc[s] = (status == CST_STREAM_OK) ? n_words : 0u;
}

// LDS layout: [word rings, 8 KiB per wave][bucket index 128 B per stream][symbol tiles][dump rows][rows]
__global__ __launch_bounds__(kBlock) void ans_decode_pt_kernel(const PtArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = threadIdx.x >> 6;
    const int P = a.precision;
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem) + wave_in_block * (kPtRingSlots * kWave);
    uint8_t* l1_l = smem + kPtRingBytes;
    int32_t* tile = reinterpret_cast<int32_t*>(smem + kPtRingBytes + kPtL1Bytes) + wave_in_block * (kWave * kTileStride);
    uint32_t* dump = reinterpret_cast<uint32_t*>(smem + kPtRingBytes + kPtL1Bytes + kPtTileBytes) + wave_in_block * (4 * kWave) + lane;
    uint32_t* rows_l = reinterpret_cast<uint32_t*>(smem + kPtRingBytes + kPtL1Bytes + kPtTileBytes + kPtDumpBytes);

    const size_t block_s0 = (size_t)blockIdx.x * kBlock;
    const int bshift = P - kPtBucketBits;
    {   // Bucket index of the block's streams (32 KiB, contiguous in HBM), interleaved in LDS in groups of G = 2^bshift
        // lanes: bucket k of lane j at  (j / G) * 128 * G + k * G + j % G,  so that a lane's address for quantile q is
        // (q & bucket_mask) | lane_base -- one v_and_or from the coder state.
        const size_t have = a.n_streams - block_s0 < (size_t)kBlock ? (a.n_streams - block_s0) * kPtBuckets : kPtL1Bytes;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(a.l1 + block_s0 * kPtBuckets);
        for (uint32_t i = threadIdx.x; i < kPtL1Bytes / 4; i += blockDim.x) {
            const uint32_t w = (size_t)i * 4 < have ? src[i] : 0u;
            const uint32_t j = (4 * i) / kPtBuckets, k = (4 * i) % kPtBuckets;
            uint8_t* d = l1_l + (((j >> bshift) * kPtBuckets) << bshift) + (j & ((1u << bshift) - 1u));
#pragma unroll
            for (int b = 0; b < 4; ++b) d[(k + b) << bshift] = (uint8_t)(w >> (8 * b));
        }
    }
    {
        const uint32_t bb = a.block_base[blockIdx.x], be = a.block_base[blockIdx.x + 1];
        pt_stage_rows(rows_l, a.rows_dec + bb, be - bb);
    }
    __syncthreads();

    const size_t s0 = block_s0 + (size_t)wave_in_block * kWave;
    if (s0 >= a.n_streams) return;
    const size_t s = s0 + lane;
    const bool active = s < a.n_streams;
    const size_t N = a.n_per_stream;
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;
    const uint32_t qmask = (1u << P) - 1u;
    const uint32_t bucket_mask = (uint32_t)(kPtBuckets - 1) << bshift;
    const PtMeta mt = active ? a.meta[s] : PtMeta{0u, 0u, 0, 1, 1, 0};
    const uint32_t* rowp = rows_l + mt.dec_off;                 // 16-byte aligned
    const uint32_t row_addr = lds_addr(rowp);
    const uint8_t* l1p = l1_l + (((threadIdx.x >> bshift) * kPtBuckets) << bshift) + (threadIdx.x & ((1u << bshift) - 1u));

    DecLane<32, 64, kPtRingSlots, kPtAhead> L;
    const WordSlice ws = active ? word_slice(a.offsets, a.stride_words, a.n_words_in, s, a.words_capacity) : WordSlice{0, 0u, false};
    L.init(a.words_in + ws.off, ws.n, ring, lane);
    if (raw) L.state = active ? a.state[s] : 0;
    else L.read_initial_state();
    L.in.prime();
    wave_lds_fence();
    uint32_t lo = (uint32_t)L.state, hi = (uint32_t)(L.state >> 32);

    auto decode_one = [&]() -> int32_t {
        const uint32_t q = lo & qmask;
        // candidate word for the refill, requested before the table search so that the search's waits cover it
        const uint32_t next_word = *L.in.slot(L.in.rd - 1u + L.in.shift);
        uint32_t r0 = l1p[q & bucket_mask];
        const uint32_t qk = (q << 20) | 0xffffeu;
        // eight entries from the 16-byte aligned quad that holds the first candidate (r0 = a quarter of its position): the bin is
        // the entry of the smallest wrapping distance qk - entry among the first seven (entries above the key wrap to huge
        // distances, the sentinels behind a row to qk + 1), unless the eighth is at or below the key too: then four entries further
        uint32_t dmin = 0xffffffffu;
        for (int guard = 0;; ++guard) {
            const uint4* pr = reinterpret_cast<const uint4*>(rowp) + r0;
            const uint4 xa = pr[0], xb = pr[1];
            dmin = min(dmin, min(min(qk - xa.x, qk - xa.y), min(qk - xa.z, qk - xa.w)));
            dmin = min(dmin, min(min(qk - xb.x, qk - xb.y), qk - xb.z));
            const bool more = xb.w <= qk;
            if (!__any(more) || guard > 64) break;
            r0 += more ? 1u : 0u;
        }
        const uint32_t e = qk - dmin;
        const uint32_t pm1 = (e >> 8) & 0xfffu;
        const bool run = pm1 == kPtRunMark;               // a run of unit probabilities: symbol index + (q - c), (c, p) = (q, 1)
        const uint32_t d = q - (e >> 20);
        const uint32_t p = run ? 1u : pm1 + 1u;
        const uint32_t qc = run ? 0u : d;
        const uint32_t idx = (e & 0xffu) + (run ? d : 0u);
        // (state >> P) * p + (q - c) on 32-bit halves (P >= 8: the high product fits v_mul_u32_u24)
        const uint32_t s_lo = __builtin_amdgcn_alignbit(hi, lo, P), s_hi = hi >> P;
        const uint64_t t = (uint64_t)s_lo * p + (uint64_t)qc;
        const uint32_t t_lo = (uint32_t)t;
        const uint32_t t_hi = __umul24(s_hi, p) + (uint32_t)(t >> 32);
        const bool refill = t_hi == 0u && L.in.rd > 0u;                // stack.rs:1089-1097
        lo = refill ? next_word : t_lo;
        hi = refill ? t_lo : t_hi;
        L.in.rd -= refill ? 1u : 0u;
        return a.min_symbol + (int32_t)idx;
    };

    int32_t* row = a.symbols_out + (active ? s : 0) * N;
    const bool vec = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.symbols_out) & 15) == 0);
    const size_t n_full = vec ? N / kTileSyms : 0;
    int32_t* my = tile + lane * kTileStride;
    size_t tb = 0;
    if (P == 12 && n_full > 0 && s0 + kWave <= a.n_streams && N < (1u << 24)) {
        // The main-loop statement addresses HBM as uniform base + 32-bit lane offset: it needs a full wave, rows of
        // < 2^24 symbols and this wave's compressed words within 2 GiB of a.words_in.
        const unsigned char* words_base = reinterpret_cast<const unsigned char*>(reinterpret_cast<uintptr_t>(a.words_in) & ~(uintptr_t)15);
        const uint64_t w_off = (uint64_t)(reinterpret_cast<const unsigned char*>(L.in.base16) - words_base);
        const bool off_ok = w_off + 4ull * ((uint64_t)L.in.rd + 8) < 0x80000000ull;
        if ((lds_addr(ring) & (uint32_t)(kPtRingSlots * kWave * 4 - 1)) != 0 || (lds_addr(l1_l) & (uint32_t)((kPtBuckets << bshift) - 1)) != 0) __builtin_trap();
        if (!__any(!off_ok)) {
            uint32_t goff[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) goff[k] = (uint32_t)((((size_t)(lane >> 3) + 8 * k) * N + 4 * (size_t)(lane & 7)) * 4);
            const uint32_t tr_off = (uint32_t)(((lane >> 3) * kTileStride + 4 * (lane & 7)) * 4);
            const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(a.symbols_out + s0 * N);
            // wave-uniform store base in SGPRs (readfirstlane returns int: go through uint32_t)
            const uint64_t store_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
            // rows that do not start on cache-line boundaries: plain tile stores (scripts/gen_decode_loop.py, CST_STORE_MOD)
            const bool plain_stores = __builtin_amdgcn_readfirstlane((int)(((N * 4) % 128 != 0 || (sb & 127) != 0) ? 1 : 0)) != 0;
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the statement keeps its own book from here
            pt_decode_tiles_loop(lo, hi, L.in.rd, L.in.lo_issued, bucket_mask, (uint32_t)((kPtRingSlots - 1) * kWave * 4), (uint32_t)P,
                                 a.min_symbol, words_base, store_base, (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)n_full),
                                 lds_addr(l1p), row_addr, L.in.shift - 1u, lds_addr(ring + lane), lds_addr(dump), (uint32_t)w_off,
                                 lds_addr(my), lds_addr(tile) + tr_off, goff, plain_stores);
            tb = n_full;
        }
    }
    for (; tb < n_full; ++tb) {
#pragma unroll
        for (int j = 0; j < kTileSyms / 4; ++j) {
            int4 v;
            v.x = decode_one(); v.y = decode_one(); v.z = decode_one(); v.w = decode_one();
            *reinterpret_cast<int4*>(my + 4 * j) = v;
        }
        L.in.template advance_window_fixed<3>(dump);
        wave_lds_fence();
        tile_store<true>(a.symbols_out, a.n_streams, N, s0, tb * kTileSyms, lane, tile);
        wave_lds_fence();
    }
    for (size_t t = n_full * kTileSyms; t < N; ++t) {
        const int32_t v = decode_one();
        if (active) row[t] = v;
        L.in.advance_window();
    }
    if (!active) return;
    a.status[s] = ws.bad ? (int32_t)CST_STREAM_INVALID_DATA : L.status;
    if (raw) {
        a.state[s] = ((uint64_t)hi << 32) | lo;
        if (a.n_words_left) a.n_words_left[s] = L.in.rd;
    }
}


// ------------------------------------------------------------------------------------------------------------------
// SUB-LANE decoder (round 5): k = 2^sub_shift jump points per stream, every (stream, chunk) pair on its own lane.
//
// ans_decode_pt_kernel runs ONE wave per SIMD (its rings, tiles and rows fill the LDS) and spends 183 of its 392 cycles per
// symbol waiting for its two dependent LDS round trips (profiles/r04_sq_counters.md).  The reference's own remedy for "one
// coder decodes strictly in order" is the jump table of Pos / Seek (stack.rs:1107-1139): the encoder notes (words in the
// bulk, state) in front of every chunk, and AnsCoder::seek(pos, state) resumes there.  So a workgroup here is EIGHT waves
// (two per SIMD: one wave's lookups are in flight while its sibling issues) for 512 / k streams, lane v decoding chunk
// v % k of stream v / k; the k lanes of a stream share that stream's row and bucket index in LDS, so the tables cost what
// they cost the plain kernel (less: fewer streams per workgroup for k > 2).  What pays for the second set of rings is
// the symbol tile: bytes instead of int32 (GEN_PT_SUB of the generator), one landing area for unused chunk slots.
// LDS layout: [word rings, 8 KiB per wave][bucket index 128 B per stream][byte tiles 2304 B per wave][dump 1 KiB][rows]
// The symbol matrix [n_streams][N] is the matrix [n_streams * k][N / k] of the virtual streams; words, counts and status are
// those of any other decoder of these words (the jump points are side information).
// ------------------------------------------------------------------------------------------------------------------
// WAVES = 8: 32-slot rings (cst_pt_decode_loop_sub.inc), two waves per SIMD.  WAVES = 16: 16-slot rings, the window moves every half
// tile (cst_pt_decode_loop_sub16.inc), four waves per SIMD -- for k >= 8 jump points per stream, whose 128 or 64 streams per
// workgroup leave room for sixteen rings: 64 + 16 + 36 + 1 + rows KiB.
// (8 waves with the 16-slot rings: k = 2, whose 256 streams per workgroup do not fit beside eight 32-slot rings on wide tables)
template <int WAVES, bool RING16> struct SubGeo {
    static constexpr int kThreads = WAVES * kWave;
    static constexpr int kSlots = RING16 ? 16 : kPtRingSlots;
    static constexpr int kAhead = RING16 ? 12 : kPtAhead;
    static constexpr size_t kRingBytes = (size_t)WAVES * kSlots * kWave * 4;
};
constexpr int kSubTileRow = 36;                                                        // bytes between the rows of a byte tile
constexpr size_t kSubTileBytes = (size_t)kWave * kSubTileRow;                          // 2304 B per wave
constexpr size_t kSubDumpBytes = 4 * kWave * 4;                                        // ONE landing area (never read)

// SB = bytes per symbol of the matrix behind a.symbols_out: 4, or 1 (int8, round 6: the four index bytes of a tile piece get min_symbol
// added byte by byte and leave with one 4-byte store)
template <int WAVES, bool RING16, int SB = 4>
__global__ __launch_bounds__(WAVES * kWave) void ans_decode_pt_sub_kernel(const PtArgs a) {
    using SymT = typename std::conditional<SB == 1, int8_t, int32_t>::type;
    SymT* symbols_out = reinterpret_cast<SymT*>(a.symbols_out);
    constexpr int kSubWaves = WAVES, kSlots = SubGeo<WAVES, RING16>::kSlots;
    constexpr size_t kSubRingBytes = SubGeo<WAVES, RING16>::kRingBytes;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = threadIdx.x >> 6;
    const int P = a.precision;
    const int ks = a.sub_shift;
    // Wave w of the grid decodes chunk w % k of the 64 streams of group w / k: the lanes of a wave are 64 DIFFERENT streams at the
    // same chunk, so their symbol rows lie a whole stream apart, like the plain decoder's (lanes that were the chunks of one
    // stream -- rows n_per_stream / k apart -- measured 10-20 % slower at k = 4: 4-KiB strides meet in the memory channels).
    // A workgroup's eight waves are the k chunks of 8 / k groups (k <= 8) or eight chunks of one group (k = 16).
    const size_t wave_global = (size_t)blockIdx.x * kSubWaves + wave_in_block;
    const uint32_t chunk = (uint32_t)(wave_global & (((size_t)1 << ks) - 1));
    const size_t s0 = (wave_global >> ks) * kWave;
    const size_t block_s0 = (((size_t)blockIdx.x * kSubWaves) >> ks) * kWave;
    if (block_s0 >= a.n_streams) return;                               // (the whole workgroup)
    const uint32_t S = (uint32_t)max(kSubWaves >> ks, 1) * kWave;     // streams whose tables this workgroup stages
    const size_t l1_bytes = (size_t)S * kPtBuckets;
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem) + wave_in_block * (kSlots * kWave);
    uint8_t* l1_l = smem + kSubRingBytes;
    uint8_t* tile = smem + kSubRingBytes + l1_bytes + (size_t)wave_in_block * kSubTileBytes;
    uint32_t* dump = reinterpret_cast<uint32_t*>(smem + kSubRingBytes + l1_bytes + kSubWaves * kSubTileBytes) + lane;
    uint32_t* rows_l = reinterpret_cast<uint32_t*>(smem + kSubRingBytes + l1_bytes + kSubWaves * kSubTileBytes + kSubDumpBytes);

    const size_t block_s1 = block_s0 + S < a.n_streams ? block_s0 + S : a.n_streams;
    const int bshift = P - kPtBucketBits;
    {   // bucket index of the workgroup's streams, interleaved in groups of G = 2^bshift streams (see ans_decode_pt_kernel)
        const size_t have = (block_s1 - block_s0) * kPtBuckets;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(a.l1 + block_s0 * kPtBuckets);
        for (uint32_t i = threadIdx.x; i < l1_bytes / 4; i += blockDim.x) {
            const uint32_t w = (size_t)i * 4 < have ? src[i] : 0u;
            const uint32_t j = (4 * i) / kPtBuckets, k = (4 * i) % kPtBuckets;
            uint8_t* d = l1_l + (((j >> bshift) * kPtBuckets) << bshift) + (j & ((1u << bshift) - 1u));
#pragma unroll
            for (int b = 0; b < 4; ++b) d[(k + b) << bshift] = (uint8_t)(w >> (8 * b));
        }
    }
    // the rows of streams [block_s0, block_s1): a contiguous piece of their block of kBlock tables (rows lie back to back)
    const size_t mb = block_s0 / kBlock;
    const uint32_t first_off = a.meta[block_s0].dec_off;
    {
        const uint32_t bb = a.block_base[mb];
        const uint32_t end = (block_s1 % kBlock != 0 && block_s1 < a.n_streams) ? bb + a.meta[block_s1].dec_off : a.block_base[mb + 1];
        pt_stage_rows(rows_l, a.rows_dec + bb + first_off, end - (bb + first_off));
    }
    __syncthreads();

    if (s0 >= a.n_streams) return;
    const size_t s_lane = s0 + lane;
    const bool active = s_lane < a.n_streams;
    const size_t s = active ? s_lane : block_s0;
    const size_t v = (s << ks) + chunk;                               // index of (stream, chunk) in the jump table and in d_status
    const uint32_t ls = (uint32_t)(s - block_s0);                     // the lane's stream within the workgroup
    const size_t N = a.n_per_stream;
    const size_t K = a.interval;                                      // symbols per chunk
    const uint32_t qmask = (1u << P) - 1u;
    const uint32_t bucket_mask = (uint32_t)(kPtBuckets - 1) << bshift;
    const PtMeta mt = a.meta[s];
    const uint32_t* rowp = rows_l + (mt.dec_off - first_off);        // 16-byte aligned
    const uint32_t row_addr = lds_addr(rowp);
    const uint8_t* l1p = l1_l + (((ls >> bshift) * kPtBuckets) << bshift) + (ls & ((1u << bshift) - 1u));

    DecLane<32, 64, kSlots, SubGeo<WAVES, RING16>::kAhead> L;
    // AnsCoder::seek(pos, state): the words in front of the jump point, checked against the slab / the buffer like any count
    const WordSlice ws = active ? word_slice_n(a.offsets, a.stride_words, a.ckpt_pos[v], s, a.words_capacity) : WordSlice{0, 0u, false};
    L.init(a.words_in + ws.off, ws.n, ring, lane);
    L.state = active && !ws.bad ? a.ckpt_state[v] : 0;
    L.in.prime();
    wave_lds_fence();
    uint32_t lo = (uint32_t)L.state, hi = (uint32_t)(L.state >> 32);

    auto decode_one = [&]() -> int32_t {         // as in ans_decode_pt_kernel
        const uint32_t q = lo & qmask;
        const uint32_t next_word = *L.in.slot(L.in.rd - 1u + L.in.shift);
        uint32_t r0 = l1p[q & bucket_mask];
        const uint32_t qk = (q << 20) | 0xffffeu;
        // eight entries from the 16-byte aligned quad that holds the first candidate (r0 = a quarter of its position): the bin is
        // the entry of the smallest wrapping distance qk - entry among the first seven (entries above the key wrap to huge
        // distances, the sentinels behind a row to qk + 1), unless the eighth is at or below the key too: then four entries further
        uint32_t dmin = 0xffffffffu;
        for (int guard = 0;; ++guard) {
            const uint4* pr = reinterpret_cast<const uint4*>(rowp) + r0;
            const uint4 xa = pr[0], xb = pr[1];
            dmin = min(dmin, min(min(qk - xa.x, qk - xa.y), min(qk - xa.z, qk - xa.w)));
            dmin = min(dmin, min(min(qk - xb.x, qk - xb.y), qk - xb.z));
            const bool more = xb.w <= qk;
            if (!__any(more) || guard > 64) break;
            r0 += more ? 1u : 0u;
        }
        const uint32_t e = qk - dmin;
        const uint32_t pm1 = (e >> 8) & 0xfffu;
        const bool run = pm1 == kPtRunMark;
        const uint32_t d = q - (e >> 20);
        const uint32_t p = run ? 1u : pm1 + 1u;
        const uint32_t qc = run ? 0u : d;
        const uint32_t idx = (e & 0xffu) + (run ? d : 0u);
        const uint32_t s_lo = __builtin_amdgcn_alignbit(hi, lo, P), s_hi = hi >> P;
        const uint64_t t = (uint64_t)s_lo * p + (uint64_t)qc;
        const uint32_t t_lo = (uint32_t)t;
        const uint32_t t_hi = __umul24(s_hi, p) + (uint32_t)(t >> 32);
        const bool refill = t_hi == 0u && L.in.rd > 0u;
        lo = refill ? next_word : t_lo;
        hi = refill ? t_lo : t_hi;
        L.in.rd -= refill ? 1u : 0u;
        return a.min_symbol + (int32_t)idx;
    };

    SymT* row = symbols_out + s * N + (size_t)chunk * K;
    const size_t n_full = K / kTileSyms;
    size_t t_done = 0;
    if (P == 12 && n_full > 0 && s0 + kWave <= a.n_streams && K % 4 == 0 && N % 4 == 0 && (reinterpret_cast<uintptr_t>(a.symbols_out) & 15) == 0 &&
        N < (1u << 24)) {
        const unsigned char* words_base = reinterpret_cast<const unsigned char*>(reinterpret_cast<uintptr_t>(a.words_in) & ~(uintptr_t)15);
        const uint64_t w_off = (uint64_t)(reinterpret_cast<const unsigned char*>(L.in.base16) - words_base);
        const bool off_ok = w_off + 4ull * ((uint64_t)L.in.rd + 8) < 0x80000000ull;
        if ((lds_addr(ring) & (uint32_t)(kSlots * kWave * 4 - 1)) != 0 || (lds_addr(l1_l) & (uint32_t)((kPtBuckets << bshift) - 1)) != 0) __builtin_trap();
        if (!__any(!off_ok)) {
            uint32_t goff[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) goff[k] = (uint32_t)((((size_t)(lane >> 3) + 8 * k) * N + 4 * (size_t)(lane & 7)) * SB);
            const uint32_t tr_off = (uint32_t)((lane >> 3) * kSubTileRow + 4 * (lane & 7));
            const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(symbols_out + s0 * N + (size_t)chunk * K);
            const uint64_t store_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
            const bool plain_stores = __builtin_amdgcn_readfirstlane((int)(((N * 4) % 128 != 0 || (sb & 127) != 0) ? 1 : 0)) != 0;
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the statement keeps its own book from here
            if constexpr (SB == 1 && RING16)
                pt_decode_tiles_loop_sub16_n8(lo, hi, L.in.rd, L.in.lo_issued, bucket_mask, (uint32_t)((kSlots - 1) * kWave * 4), (uint32_t)P,
                                              a.min_symbol, words_base, store_base, (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)n_full),
                                              lds_addr(l1p), row_addr, L.in.shift - 1u, lds_addr(ring + lane), lds_addr(dump), (uint32_t)w_off,
                                              lds_addr(tile + lane * kSubTileRow), lds_addr(tile) + tr_off, goff);
            else if constexpr (SB == 1)
                pt_decode_tiles_loop_sub_n8(lo, hi, L.in.rd, L.in.lo_issued, bucket_mask, (uint32_t)((kSlots - 1) * kWave * 4), (uint32_t)P,
                                            a.min_symbol, words_base, store_base, (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)n_full),
                                            lds_addr(l1p), row_addr, L.in.shift - 1u, lds_addr(ring + lane), lds_addr(dump), (uint32_t)w_off,
                                            lds_addr(tile + lane * kSubTileRow), lds_addr(tile) + tr_off, goff);
            else if constexpr (RING16)
                pt_decode_tiles_loop_sub16(lo, hi, L.in.rd, L.in.lo_issued, bucket_mask, (uint32_t)((kSlots - 1) * kWave * 4), (uint32_t)P,
                                           a.min_symbol, words_base, store_base, (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)n_full),
                                           lds_addr(l1p), row_addr, L.in.shift - 1u, lds_addr(ring + lane), lds_addr(dump), (uint32_t)w_off,
                                           lds_addr(tile + lane * kSubTileRow), lds_addr(tile) + tr_off, goff, plain_stores);
            else
                pt_decode_tiles_loop_sub(lo, hi, L.in.rd, L.in.lo_issued, bucket_mask, (uint32_t)((kSlots - 1) * kWave * 4), (uint32_t)P,
                                         a.min_symbol, words_base, store_base, (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)n_full),
                                         lds_addr(l1p), row_addr, L.in.shift - 1u, lds_addr(ring + lane), lds_addr(dump), (uint32_t)w_off,
                                         lds_addr(tile + lane * kSubTileRow), lds_addr(tile) + tr_off, goff, plain_stores);
            t_done = n_full * kTileSyms;
        }
    }
    for (size_t t = t_done; t < K; ++t) {          // shapes the statement does not take: symbol by symbol (correct, slow)
        const int32_t sym = decode_one();
        if (active) row[t] = (SymT)sym;
        L.in.advance_window();
    }
    if (!active) return;
    a.status[v] = ws.bad ? (int32_t)CST_STREAM_INVALID_DATA : L.status;
}

static size_t pt_lds_bytes(const cst_model* m, bool encode) {
    return encode ? kPtRingBytes + ((size_t)8 << m->precision) + kPtTileBytes + 2 * ((size_t)m->pt_max_enc + 4)
                  : kPtRingBytes + kPtL1Bytes + kPtTileBytes + kPtDumpBytes + 4 * ((size_t)m->pt_max_dec + 4);
}

bool pt_usable(const cst_model* m, cst_coder_config cfg, cst_layout layout, size_t n_per_stream) {
    (void)n_per_stream;
    if (!m->pt_ok || cfg.word_bits != 32 || layout != CST_LAYOUT_STREAM_MAJOR) return false;
    if (m->precision < 8 || m->precision > 12) return false;
    return pt_lds_bytes(m, true) <= 160 * 1024 && pt_lds_bytes(m, false) <= 160 * 1024;
}

template <typename K>
static cst_status pt_launch(K kernel, const PtArgs& a, size_t lds, hipStream_t hs) {
    const size_t blocks = (a.n_streams + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    if (lds > 64 * 1024)
        CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kBlock), lds, hs, a);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

cst_status ans_encode_pt(const cst_model* model, cst_coder_config cfg, const int32_t* d_symbols, size_t n_streams, size_t n_per_stream,
                         uint32_t* d_words, size_t stride_words, uint32_t* d_n_words, uint64_t* d_state, int32_t* d_status,
                         uint32_t flags, hipStream_t hs) {
    (void)cfg;
    if (model->n_tables != n_streams) return CST_ERR_INVALID_ARGUMENT;
    PtArgs a{};
    a.symbols_in = d_symbols; a.n_streams = n_streams; a.n_per_stream = n_per_stream;
    a.precision = model->precision; a.n_symbols = model->n_symbols; a.min_symbol = model->min_symbol;
    a.meta = model->d_pt_meta; a.rows_enc = model->d_pt_enc; a.block_base = model->d_pt_block_base; a.recip = model->d_recip;
    a.words_out = d_words; a.stride_words = stride_words; a.n_words_out_enc = d_n_words; a.state = d_state; a.status = d_status;
    a.flags = flags;
    return pt_launch(ans_encode_pt_kernel<false>, a, pt_lds_bytes(model, true), hs);
}

// ans_encode_pt + a jump point in front of every chunk of `interval` symbols (the caller checked the arguments)
cst_status ans_encode_pt_ckpt(const cst_model* model, const int32_t* d_symbols, size_t n_streams, size_t n_per_stream, uint32_t* d_words,
                              size_t stride_words, uint32_t* d_n_words, size_t interval, uint32_t* d_ckpt_pos, uint64_t* d_ckpt_state,
                              int32_t* d_status, hipStream_t hs) {
    if (model->n_tables != n_streams) return CST_ERR_INVALID_ARGUMENT;
    PtArgs a{};
    a.symbols_in = d_symbols; a.n_streams = n_streams; a.n_per_stream = n_per_stream;
    a.precision = model->precision; a.n_symbols = model->n_symbols; a.min_symbol = model->min_symbol;
    a.meta = model->d_pt_meta; a.rows_enc = model->d_pt_enc; a.block_base = model->d_pt_block_base; a.recip = model->d_recip;
    a.words_out = d_words; a.stride_words = stride_words; a.n_words_out_enc = d_n_words; a.status = d_status;
    a.ckpt_pos = d_ckpt_pos; a.ckpt_state = d_ckpt_state; a.interval = interval; a.n_chunks = (n_per_stream + interval - 1) / interval;
    return pt_launch(ans_encode_pt_kernel<true>, a, pt_lds_bytes(model, true), hs);
}

// int8 symbol matrices inside the loops (round 6): rows of whole tiles in chunks of whole tiles, a support inside int8
bool pt_n8_encode_usable(const cst_model* m, cst_coder_config cfg, cst_layout layout, const void* d_symbols, size_t n_streams, size_t n_per_stream,
                         size_t interval) {
    if (knobs().no_n8 || !pt_usable(m, cfg, layout, n_per_stream) || m->n_tables != n_streams) return false;
    if (interval == 0 || interval % kTileSyms != 0 || n_per_stream % interval != 0 || n_per_stream >= (1u << 24)) return false;
    if ((reinterpret_cast<uintptr_t>(d_symbols) & 15) != 0 || n_streams * (n_per_stream / interval) >= (1u << 28)) return false;
    return m->min_symbol >= -128 && m->min_symbol + m->n_symbols - 1 <= 127;
}

cst_status ans_encode_pt_ckpt_n8(const cst_model* model, const void* d_symbols8, size_t n_streams, size_t n_per_stream, uint32_t* d_words,
                                 size_t stride_words, uint32_t* d_n_words, size_t interval, uint32_t* d_ckpt_pos, uint64_t* d_ckpt_state,
                                 int32_t* d_status, hipStream_t hs) {
    if (model->n_tables != n_streams) return CST_ERR_INVALID_ARGUMENT;
    PtArgs a{};
    a.symbols_in = reinterpret_cast<const int32_t*>(d_symbols8); a.n_streams = n_streams; a.n_per_stream = n_per_stream;
    a.precision = model->precision; a.n_symbols = model->n_symbols; a.min_symbol = model->min_symbol;
    a.meta = model->d_pt_meta; a.rows_enc = model->d_pt_enc; a.block_base = model->d_pt_block_base; a.recip = model->d_recip;
    a.words_out = d_words; a.stride_words = stride_words; a.n_words_out_enc = d_n_words; a.status = d_status;
    a.ckpt_pos = d_ckpt_pos; a.ckpt_state = d_ckpt_state; a.interval = interval; a.n_chunks = n_per_stream / interval;
    return pt_launch(ans_encode_pt_kernel<true, 1>, a, pt_lds_bytes(model, true), hs);
}

// rows of the S = max(waves / k, 1) * 64 streams of a workgroup: at most S / 64 times the largest 64-stream group (pt_max_dec64),
// and never more than the largest block of kBlock streams
static size_t pt_sub_rows_bytes(const cst_model* m, int waves, int sub_shift) {
    const size_t groups = (size_t)((waves >> sub_shift) > 1 ? (waves >> sub_shift) : 1);
    const size_t by_groups = groups * (size_t)m->pt_max_dec64, by_block = (size_t)m->pt_max_dec;
    return 4 * ((by_groups < by_block ? by_groups : by_block) + 4);
}

static size_t pt_sub_lds_bytes(const cst_model* m, int waves, bool ring16, int sub_shift) {
    const size_t S = (size_t)((waves >> sub_shift) > 1 ? (waves >> sub_shift) : 1) * kWave;        // streams per workgroup
    return (size_t)waves * (ring16 ? 16 : kPtRingSlots) * kWave * 4 + S * kPtBuckets + (size_t)waves * kSubTileBytes + kSubDumpBytes +
           pt_sub_rows_bytes(m, waves, sub_shift);
}

static int pt_sub_shift(size_t k) {
    int ks = 0;
    while (((size_t)1 << ks) < k) ++ks;
    return ks;
}

// The geometry of a launch: sixteen waves per workgroup (four per SIMD, 16-slot rings) where k >= 8 jump points per stream leave
// room for them; else eight waves with the 32-slot rings; else eight waves with the 16-slot rings (k = 2 on wide tables); 0 = none
// fits.  (CST_PT_SUB_WAVES=8 never takes sixteen: A/B runs.)  The 16-slot statement is for P = 12.
struct PtSubGeo { int waves; bool ring16; };
static PtSubGeo pt_sub_geometry(const cst_model* m, int ks) {
    const bool p12 = m->precision == 12;
    if (!knobs().pt_sub_8_waves && ks >= 3 && p12 && pt_sub_lds_bytes(m, 16, true, ks) <= 160 * 1024) return {16, true};
    if (pt_sub_lds_bytes(m, 8, false, ks) <= 160 * 1024) return {8, false};
    if (p12 && pt_sub_lds_bytes(m, 8, true, ks) <= 160 * 1024) return {8, true};
    return {0, false};
}

// k lanes per stream (k = n_per_stream / interval a power of two, 2 <= k <= 16), tables that fit next to the waves' rings
bool pt_sub_usable(const cst_model* m, cst_coder_config cfg, size_t n_streams, size_t n_per_stream, size_t interval) {
    if (!m->pt_ok || cfg.word_bits != 32 || m->precision < 8 || m->precision > 12 || m->n_tables != n_streams) return false;
    if (interval == 0 || n_per_stream % interval != 0) return false;
    const size_t k = n_per_stream / interval;
    if (k < 2 || k > 16 || (k & (k - 1)) != 0 || n_streams * k > 0x7fffffffull) return false;
    return pt_sub_geometry(m, pt_sub_shift(k)).waves != 0;
}

cst_status ans_decode_pt_sub(const cst_model* model, const uint32_t* d_words, const uint64_t* d_offsets, size_t stride_words,
                             size_t words_capacity, size_t interval, const uint32_t* d_ckpt_pos, const uint64_t* d_ckpt_state,
                             int32_t* d_symbols, size_t n_streams, size_t n_per_stream, int32_t* d_status, hipStream_t hs, int symbol_bytes) {
    PtArgs a{};
    const size_t k = n_per_stream / interval;
    const int ks = pt_sub_shift(k);
    a.words_capacity = words_capacity ? words_capacity : (d_offsets ? 0 : n_streams * stride_words);
    a.symbols_out = d_symbols; a.n_streams = n_streams; a.n_per_stream = n_per_stream;
    a.precision = model->precision; a.n_symbols = model->n_symbols; a.min_symbol = model->min_symbol;
    a.meta = model->d_pt_meta; a.rows_dec = model->d_pt_dec; a.l1 = model->d_pt_l1;
    a.block_base = model->d_pt_block_base + (n_streams + kBlock - 1) / kBlock + 1;
    a.words_in = d_words; a.offsets = d_offsets; a.stride_words = stride_words; a.status = d_status;
    a.ckpt_pos = const_cast<uint32_t*>(d_ckpt_pos); a.ckpt_state = const_cast<uint64_t*>(d_ckpt_state);
    a.interval = interval; a.n_chunks = k; a.sub_shift = ks;
    const PtSubGeo geo = pt_sub_geometry(model, ks);
    if (geo.waves == 0) return CST_ERR_INVALID_ARGUMENT;
    const size_t lds = pt_sub_lds_bytes(model, geo.waves, geo.ring16, ks);
    const size_t blocks = ((((n_streams + kWave - 1) / kWave) << ks) + geo.waves - 1) / geo.waves;       // a wave = 64 streams x one chunk
    auto go = [&](auto kernel) -> cst_status {
        CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(geo.waves * kWave), lds, hs, a);
        CST_HIP_TRY(hipGetLastError());
        return CST_OK;
    };
    if (symbol_bytes == 1) {
        if (geo.waves == 16) return go(ans_decode_pt_sub_kernel<16, true, 1>);
        return geo.ring16 ? go(ans_decode_pt_sub_kernel<8, true, 1>) : go(ans_decode_pt_sub_kernel<8, false, 1>);
    }
    if (geo.waves == 16) return go(ans_decode_pt_sub_kernel<16, true>);
    return geo.ring16 ? go(ans_decode_pt_sub_kernel<8, true>) : go(ans_decode_pt_sub_kernel<8, false>);
}

// ... writing an int8 matrix itself: the sub-lane shapes whose support fits the type (P = 12: the statement's precision)
bool pt_sub_n8_usable(const cst_model* m, cst_coder_config cfg, size_t n_streams, size_t n_per_stream, size_t interval, const void* d_symbols) {
    if (knobs().no_n8 || !pt_sub_usable(m, cfg, n_streams, n_per_stream, interval)) return false;
    if (interval % 4 != 0 || n_per_stream % 4 != 0 || (reinterpret_cast<uintptr_t>(d_symbols) & 15) != 0) return false;
    return m->min_symbol >= -128 && m->min_symbol + m->n_symbols - 1 <= 127;
}

cst_status ans_decode_pt(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_offsets,
                         size_t stride_words, size_t words_capacity, const uint32_t* d_n_words, int32_t* d_symbols, size_t n_streams,
                         size_t n_per_stream, uint64_t* d_state, uint32_t* d_n_words_out, int32_t* d_status, uint32_t flags, hipStream_t hs) {
    (void)cfg;
    if (model->n_tables != n_streams) return CST_ERR_INVALID_ARGUMENT;
    PtArgs a{};
    a.words_capacity = words_capacity;
    a.symbols_out = d_symbols; a.n_streams = n_streams; a.n_per_stream = n_per_stream;
    a.precision = model->precision; a.n_symbols = model->n_symbols; a.min_symbol = model->min_symbol;
    a.meta = model->d_pt_meta; a.rows_dec = model->d_pt_dec; a.l1 = model->d_pt_l1;
    a.block_base = model->d_pt_block_base + (n_streams + kBlock - 1) / kBlock + 1;
    a.words_in = d_words; a.offsets = d_offsets; a.stride_words = stride_words; a.n_words_in = d_n_words;
    a.n_words_left = d_n_words_out; a.state = d_state; a.status = d_status; a.flags = flags;
    return pt_launch(ans_decode_pt_kernel, a, pt_lds_bytes(model, false), hs);
}

} // namespace cst
