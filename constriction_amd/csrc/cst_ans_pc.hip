// cst_ans_pc.hip -- the shared-table ANS encoder (W,S) = (32,64), 8 <= P <= 12, as PRODUCER and CONSUMER waves (round 4).
//
// 65 536 streams (BASELINE config C2) are one wave per SIMD, and a lone wave issues one instruction per ~4.4 cycles whatever
// the instruction is.  In ans_encode_kernel (cst_ans_kernels.hpp) the coder chain's wave also stages the symbol tiles
// (8 x ds_write_b128 per tile: 36 cycles each from one wave), reads complete word groups back from its ring, stores them and
// requests the next tiles: a quarter of the kernel, all of it issue and LDS-store time of that one wave and none of it memory
// time (scripts/ablate_encoder.sh: 0.228 ms without the tile work, 0.275 ms with it, whatever the stores look like).  Here a
// workgroup is EIGHT waves for 256 streams: waves 0-3 (one per SIMD) run nothing but the coder steps -- the generated statement
// cst_encode_loop_pc.inc -- and waves 4-7, their partners on the same SIMDs, do everything else (generated statements, too):
//     loader (waves 4, 5; each for two coder waves), per tile:  next tile's symbols registers -> LDS tile buffer (requested two
//                        tiles earlier), request a later one, s_barrier                          cst_encode_loop_pc_loader.inc
//     storer (waves 6, 7; each for two coder waves), per tile:  one complete 64-byte group below the write position the coder
//                        last published ring -> slab (lane quads: whole 64-byte segments), s_barrier   cst_encode_loop_pc_storer.inc
//     coder, per tile:   32 steps, candidate words into the lane's 64-slot LDS ring as before; once per tile: publish the
//                        write position, lgkmcnt(0), s_barrier.
// (CST_PC_COMBINED=1: the first form, every helper wave loading, staging and flushing for its own coder wave --
// cst_encode_loop_pc_helper.inc; 1.5 % slower)
// The SIMD's VALU is what the two waves share (a VALU instruction of either occupies it for 4 cycles), so the helpers are kept
// nearly free of VALU work -- a dozen instructions per tile; their LDS, vector-memory and scalar instructions issue beside the
// coder's VALU stream.
// The barrier sits at the top of the coder's quad 1: every read of the current tile's row has returned (the helper may
// overwrite that buffer with the tile after next) and the helper has finished the next tile (quads 1 and 0 read ahead into it).
// Same recurrence (stack.rs:1014-1048), same slabs, same words, counts and status as ans_encode_kernel; shapes this kernel
// does not take (partial workgroups, rows that are not whole cache-line aligned tiles, unaligned slabs, large alphabets) stay
// where they were.
#include <cstdlib>

#include "cst_ans_kernels.hpp"

namespace cst {

constexpr int kPcThreads = 2 * kBlock;                                   // 4 coder waves + 4 helper waves
constexpr int kPcWaves = kBlock / kWave;
constexpr size_t kPcTableBytes = 16 * 1024;                              // the encoder table (<= 1024 symbols), in front of the 16-KiB aligned rings
constexpr size_t kPcRingWaveBytes = (size_t)kRingSlots * kWave * 4;      // 64 slots [slot][lane]: 16 KiB, 16-KiB aligned
constexpr size_t kPcTileBytes = (size_t)kWave * kTileStride * 4;         // 9 KiB per tile buffer
constexpr size_t kPcHandWaveBytes = 4 * kWave * 4;                       // coder -> helper: write position (every tile); state (lo, hi), largest table index (at the end)
constexpr size_t kPcRingOff = kPcTableBytes;
constexpr size_t kPcTileOff = kPcRingOff + kPcWaves * kPcRingWaveBytes;
constexpr size_t kPcHandOff = kPcTileOff + kPcWaves * 2 * kPcTileBytes;
constexpr size_t kPcLdsBytes = kPcHandOff + kPcWaves * kPcHandWaveBytes;
static_assert(kPcLdsBytes <= 160 * 1024, "LDS budget");

__device__ __forceinline__ void ans_encode_pc_coder_loop(uint32_t& lo, uint32_t& hi, int32_t& smin, int32_t& smax, const uint32_t (&tile_row_addr)[2],
                                                         uint32_t ring_lane_addr, uint32_t publish_addr, uint32_t table_bias, uint32_t P, uint32_t n_tiles) {
#include "cst_encode_loop_pc.inc"
}

// Jump points (round 5; scripts/gen_encode_loop_pc.py, ck_hook): in front of every chunk of `tiles` 32-symbol tiles the coder wave
// notes what AnsCoder::pos() returns there (stack.rs:1107-1139) -- pos[s][j] = words emitted so far, state[s][j] -- j counting down
// from n_chunks - 1 to 0 (the whole stream).  tiles == 0: no jump points (the plain statements).
struct PcJumpArgs {
    uint32_t* pos;
    uint64_t* state;
    uint32_t tiles, n_chunks;
};

__device__ __forceinline__ void ans_encode_pc_coder_loop_ck(uint32_t& lo, uint32_t& hi, int32_t& smin, int32_t& smax, const uint32_t (&tile_row_addr)[2],
                                                            uint32_t ring_lane_addr, uint32_t publish_addr, uint32_t table_bias, uint32_t P, uint32_t n_tiles,
                                                            const uint32_t* ckpt_pos, const uint64_t* ckpt_state, uint32_t ckpt_tiles,
                                                            uint32_t ckpt_pos_off, uint32_t ckpt_state_off) {
#include "cst_encode_loop_pc_ck.inc"
}

// ... at 12 < P <= 24 (round 5): unpacked entries {c, p, floor(2^64 / p)}, the step of cst_encode_loop_wide.inc
__device__ __forceinline__ void ans_encode_pc_w_coder_loop(uint32_t& lo, uint32_t& hi, int32_t& smin, int32_t& smax, const uint32_t (&tile_row_addr)[2],
                                                           uint32_t ring_lane_addr, uint32_t publish_addr, uint32_t table_bias, uint32_t P, uint32_t n_tiles) {
#include "cst_encode_loop_pc_w.inc"
}

__device__ __forceinline__ void ans_encode_pc_w_coder_loop_ck(uint32_t& lo, uint32_t& hi, int32_t& smin, int32_t& smax, const uint32_t (&tile_row_addr)[2],
                                                              uint32_t ring_lane_addr, uint32_t publish_addr, uint32_t table_bias, uint32_t P, uint32_t n_tiles,
                                                              const uint32_t* ckpt_pos, const uint64_t* ckpt_state, uint32_t ckpt_tiles,
                                                              uint32_t ckpt_pos_off, uint32_t ckpt_state_off) {
#include "cst_encode_loop_pc_w_ck.inc"
}

__device__ __forceinline__ void ans_encode_pc_helper_loop(uint32_t& flushed, const uint32_t (&tile_tr_addr)[2], uint32_t ring_lane_addr,
                                                          uint32_t publish_addr, uint32_t cap, uint32_t slab_off, const void* words_base,
                                                          uint64_t symbols_base, uint32_t n_tiles, const uint32_t (&goff)[8]) {
#include "cst_encode_loop_pc_helper.inc"
}

__device__ __forceinline__ void ans_encode_pc_loader_loop(const uint32_t (&tile_tr_addr)[2], uint64_t symbols_base, uint32_t row_block_bytes,
                                                          uint32_t n_tiles, const uint32_t (&goff0)[8], const uint32_t (&goff1)[8]) {
#include "cst_encode_loop_pc_loader.inc"
}

__device__ __forceinline__ void ans_encode_pc_storer_loop(uint32_t (&flushed)[2], const uint32_t (&ring_lane_addr)[2], const uint32_t (&publish_addr)[2],
                                                          const uint32_t (&cap)[2], const uint32_t (&slab_off)[2], const void* words_base, uint32_t n_tiles) {
#include "cst_encode_loop_pc_storer.inc"
}

// ... two 64-byte groups per tile and coder wave (the coders at 12 < P <= 24: 32 symbols can emit 24 words)
__device__ __forceinline__ void ans_encode_pc_storer2_loop(uint32_t (&flushed)[2], const uint32_t (&ring_lane_addr)[2], const uint32_t (&publish_addr)[2],
                                                           const uint32_t (&cap)[2], const uint32_t (&slab_off)[2], const void* words_base, uint32_t n_tiles) {
#include "cst_encode_loop_pc_storer2.inc"
}

// LDS hand-off between the two halves of a workgroup: this wave's LDS operations have completed, then the barrier.  (Not
// __syncthreads(): its fence would also wait for the helper's symbol loads, which are requested tiles ahead on purpose.)
__device__ __forceinline__ void pc_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// the storer waves (6, 7; each for the coder waves cw0, cw0 + 1): complete 64-byte groups ring -> slab while the coders run, then
// the ends of the streams.  Shared by the int32 and the int8 kernel (their rings and hand-off areas look alike).
template <bool TWO_GROUPS = false>
__device__ __forceinline__ void pc_storer(const AnsEncodeArgs& a, unsigned char* smem, int lane, int cw0, size_t s0, uint32_t n_t,
                                          size_t ring_off, size_t hand_off_) {
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;
    uint32_t* ring[2]; uint32_t* hand[2];
    uint32_t ring_addr[2], pub_addr[2], slab_off[2], cap[2], flushed[2] = {0, 0};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        ring[c] = reinterpret_cast<uint32_t*>(smem + ring_off + (cw0 + c) * kPcRingWaveBytes);
        hand[c] = reinterpret_cast<uint32_t*>(smem + hand_off_ + (cw0 + c) * kPcHandWaveBytes);
        hand[c][lane] = 0;                              // nothing published yet
        ring_addr[c] = lds_addr(ring[c] + lane);
        pub_addr[c] = lds_addr(hand[c] + lane);
        // a lane behind the last stream (the int8 kernel takes partial workgroups: its coder lane codes the last stream again) has a
        // slab of capacity 0 where the slab of stream s WOULD lie: nothing of it is ever stored (the quad stores count on slabs a
        // constant distance apart, so the offset is not clamped)
        slab_off[c] = (uint32_t)((s0 + c * kWave + lane) * a.stride_words * 4);
        cap[c] = s0 + c * kWave + lane < a.n_streams ? (uint32_t)a.stride_words : 0u;
    }
    if constexpr (TWO_GROUPS) ans_encode_pc_storer2_loop(flushed, ring_addr, pub_addr, cap, slab_off, a.words, n_t);
    else ans_encode_pc_storer_loop(flushed, ring_addr, pub_addr, cap, slab_off, a.words, n_t);
    pc_barrier();                                       // the coders have published their last write positions and final states
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const size_t s = s0 + c * kWave + lane;
        const bool mine = s < a.n_streams;               // (a spare lane of a partial workgroup: nothing to finish, nothing to report)
        EncLane<32, 64> L;
        L.init(a.words + (mine ? s : 0) * a.stride_words, mine ? (uint32_t)a.stride_words : 0u, ring[c], lane);
        L.out.flushed = flushed[c];
        L.out.wr = hand[c][lane];
        L.state = ((uint64_t)hand[c][2 * kWave + lane] << 32) | hand[c][kWave + lane];
        L.bad = hand[c][3 * kWave + lane];
        uint32_t n_words = 0;
        const int32_t status = L.finish(!raw, (uint32_t)a.n_symbols, n_words);
        if (!mine) continue;
        if (raw) a.state[s] = (uint64_t)L.state;
        a.status[s] = status;
        a.n_words[s] = (status == CST_STREAM_OK) ? n_words : 0u;
    }
}

// the helper waves of a workgroup split by role: waves 4, 5 load and stage the tiles of coder waves (0, 1), (2, 3); waves 6, 7
// flush their rings and finish their streams
template <bool TWO_GROUPS = false>
__device__ __forceinline__ void pc_split_helper(const AnsEncodeArgs& a, unsigned char* smem, int wave, int lane, uint32_t n_t) {
    const size_t N = a.n_per_stream;
    const int pair = wave & 1, cw0 = 2 * pair;          // the pair's first coder wave
    const size_t s0 = (size_t)blockIdx.x * kBlock + (size_t)cw0 * kWave;
    if (wave < kPcWaves + 2) {                          // ---- loader ----
        // rows of the two coder waves, clamped to the last stream: a partial workgroup's spare coder lanes code the last stream again
        // (round 5, as in the int8 kernel below; pc_storer stores nothing of theirs)
        const size_t last = a.n_streams - 1;
        const size_t first0 = min(s0, last), first1 = min(s0 + kWave, last);
        uint32_t goff0[8], goff1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const size_t row = (size_t)(lane >> 3) + 8 * k;
            goff0[k] = (uint32_t)((min(row, last - first0) * N + 4 * (size_t)(lane & 7)) * 4);
            goff1[k] = (uint32_t)((min(row, last - first1) * N + 4 * (size_t)(lane & 7)) * 4);
        }
        const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(a.symbols + first0 * N + (size_t)(n_t - 1) * kTileSyms);
        const uint64_t symbols_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                      (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
        const uint32_t tr_off = (uint32_t)(((lane >> 3) * kTileStride + 4 * (lane & 7)) * 4);
        const uint32_t t0 = lds_addr(smem + kPcTileOff + (2 * cw0) * kPcTileBytes) + tr_off;
        const uint32_t tr_addr[2] = {t0, t0 + (uint32_t)kPcTileBytes};
        const uint32_t row_block = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)((first1 - first0) * N * 4));
        __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0): the statement keeps its own book from here
        ans_encode_pc_loader_loop(tr_addr, symbols_base, row_block, n_t, goff0, goff1);
        pc_barrier();
        return;
    }
    pc_storer<TWO_GROUPS>(a, smem, lane, cw0, s0, n_t, kPcRingOff, kPcHandOff);
}

template <bool SPLIT, bool JUMP = false, bool WIDE = false>
__global__ __launch_bounds__(kPcThreads) void ans_encode_pc_kernel(const AnsEncodeArgs a, const PcJumpArgs jp) {
    static_assert(SPLIT || !WIDE, "12 < P <= 24: the split helpers only (a storer that moves two word groups per tile)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const bool helper = wave >= kPcWaves;
    const int cw = wave & (kPcWaves - 1);               // coder wave `cw` and helper wave `cw + 4` share a SIMD and 64 streams
    const int P = a.precision;
    const uint32_t nsym = (uint32_t)a.n_symbols;
    const size_t N = a.n_per_stream;
    const uint32_t n_t = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(N / kTileSyms));
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;

    EncEntry* table = reinterpret_cast<EncEntry*>(smem);
    for (int i = threadIdx.x; i < a.n_symbols; i += kPcThreads) table[i] = WIDE ? a.enc[i] : pack_entry(a.enc[i], P);
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem + kPcRingOff + cw * kPcRingWaveBytes);
    int32_t* tile[2] = {reinterpret_cast<int32_t*>(smem + kPcTileOff + (2 * cw) * kPcTileBytes),
                        reinterpret_cast<int32_t*>(smem + kPcTileOff + (2 * cw + 1) * kPcTileBytes)};
    uint32_t* hand = reinterpret_cast<uint32_t*>(smem + kPcHandOff + cw * kPcHandWaveBytes);

    const size_t s0 = (size_t)blockIdx.x * kBlock + (size_t)cw * kWave;
    // a partial workgroup (split helpers only): the coder lanes behind the last stream code the LAST stream again -- the loader stages
    // its symbols for them, pc_storer stores nothing of theirs, the jump points they note are its own
    const size_t s = min(s0 + lane, a.n_streams - 1);

    if (!helper) {
        uint32_t lo = 0, hi = 0;
        int32_t smin = a.min_symbol, smax = a.min_symbol;
        if (raw) { const uint64_t st = a.state[s]; lo = (uint32_t)st; hi = (uint32_t)(st >> 32); }
        const uint32_t row_addr[2] = {lds_addr(tile[0] + lane * kTileStride), lds_addr(tile[1] + lane * kTileStride)};
        pc_barrier();                                   // table and the first tile are in LDS
        const uint32_t bias = lds_addr(table) - 16u * (uint32_t)a.min_symbol;
        const uint32_t jpos = (uint32_t)((s * jp.n_chunks + jp.n_chunks - 1) * 4), jstate = (uint32_t)((s * jp.n_chunks + jp.n_chunks - 1) * 8);
        if constexpr (JUMP && WIDE)
            ans_encode_pc_w_coder_loop_ck(lo, hi, smin, smax, row_addr, lds_addr(ring + lane), lds_addr(hand + lane), bias, (uint32_t)P, n_t, jp.pos, jp.state,
                                          jp.tiles, jpos, jstate);
        else if constexpr (JUMP)
            ans_encode_pc_coder_loop_ck(lo, hi, smin, smax, row_addr, lds_addr(ring + lane), lds_addr(hand + lane), bias, (uint32_t)P, n_t, jp.pos, jp.state,
                                        jp.tiles, jpos, jstate);
        else if constexpr (WIDE)
            ans_encode_pc_w_coder_loop(lo, hi, smin, smax, row_addr, lds_addr(ring + lane), lds_addr(hand + lane), bias, (uint32_t)P, n_t);
        else
            ans_encode_pc_coder_loop(lo, hi, smin, smax, row_addr, lds_addr(ring + lane), lds_addr(hand + lane), bias, (uint32_t)P, n_t);
        // largest raw table index seen: a symbol below min_symbol wraps to a huge one
        hand[kWave + lane] = lo; hand[2 * kWave + lane] = hi;
        hand[3 * kWave + lane] = max((uint32_t)smax - (uint32_t)a.min_symbol, (uint32_t)smin - (uint32_t)a.min_symbol);
        pc_barrier();                                   // the last window and the final state are published
        return;
    }

    if constexpr (SPLIT) { pc_split_helper<WIDE>(a, smem, wave, lane, n_t); return; }
    else {

    // ---- helper (combined: loads, staging and flush of its own coder wave) ----
    EncLane<32, 64> L;
    L.init(a.words + s * a.stride_words, (uint32_t)a.stride_words, ring, lane);
    hand[lane] = 0;                                     // nothing published yet
    uint32_t goff[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) goff[k] = (uint32_t)((((size_t)(lane >> 3) + 8 * k) * N + 4 * (size_t)(lane & 7)) * 4);
    const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(a.symbols + s0 * N + (size_t)(n_t - 1) * kTileSyms);
    const uint64_t symbols_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                  (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
    const uint32_t tr_off = (uint32_t)(((lane >> 3) * kTileStride + 4 * (lane & 7)) * 4);
    const uint32_t tr_addr[2] = {lds_addr(tile[0]) + tr_off, lds_addr(tile[1]) + tr_off};
    uint32_t flushed = 0;
    __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0): the statement keeps its own book from here
    ans_encode_pc_helper_loop(flushed, tr_addr, lds_addr(ring + lane), lds_addr(hand + lane), (uint32_t)a.stride_words,
                              (uint32_t)(s * a.stride_words * 4), a.words, symbols_base, n_t, goff);
    pc_barrier();                                       // the coder has published its last write position and the final state
    L.out.flushed = flushed;
    L.out.wr = hand[lane];
    L.state = ((uint64_t)hand[2 * kWave + lane] << 32) | hand[kWave + lane];
    L.bad = hand[3 * kWave + lane];
    uint32_t n_words = 0;
    const int32_t status = L.finish(!raw, nsym, n_words);
    if (raw) a.state[s] = (uint64_t)L.state;
    a.status[s] = status;
    a.n_words[s] = (status == CST_STREAM_OK) ? n_words : 0u;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// INT8 symbol matrices (round 5; scripts/gen_encode_loop_pc.py, "INT8 symbol matrices"): the same producer / consumer workgroup
// over byte tiles.  A row of the matrix is int8 -- a 128-byte line is FOUR tiles -- so the loader stages whole lines (two row
// blocks per window), the coder reads a quad as one dword and forms each table address with one SDWA shift of the sign-extended
// byte against a 256-entry table centred at LDS address 2048.  A quarter of the symbol bytes from HBM, no conversion kernel, no
// scratch; words, counts and status are those of the int32 kernels on the widened values.
//   LDS: [table: 256 entries, symbol s at 2048 + 16 s][rings 4 x 16 KiB at 16 KiB][two line buffers per coder wave][hand-off]
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kPcN8RowBytes = 132;                                        // 128 symbols + one word of padding (conflict-free b32 accesses)
constexpr int kPcN8LineSyms = 128;
constexpr size_t kPcN8LineBytes = (size_t)kWave * kPcN8RowBytes;          // 8448 B per line buffer
constexpr size_t kPcN8TileOff = kPcRingOff + kPcWaves * kPcRingWaveBytes;
constexpr size_t kPcN8HandOff = kPcN8TileOff + kPcWaves * 2 * kPcN8LineBytes;
constexpr size_t kPcN8LdsBytes = kPcN8HandOff + kPcWaves * kPcHandWaveBytes;
static_assert(kPcN8LdsBytes <= 160 * 1024 && 256 * sizeof(EncEntry) <= kPcTableBytes, "LDS budget");

__device__ __forceinline__ void ans_encode_pc_n8_coder_loop(uint32_t& lo, uint32_t& hi, int32_t& smin, int32_t& smax, uint32_t line_row_addr,
                                                            uint32_t ring_lane_addr, uint32_t publish_addr, uint32_t P, uint32_t n_tiles) {
#include "cst_encode_loop_pc_n8.inc"
}

__device__ __forceinline__ void ans_encode_pc_n8_coder_loop_ck(uint32_t& lo, uint32_t& hi, int32_t& smin, int32_t& smax, uint32_t line_row_addr,
                                                               uint32_t ring_lane_addr, uint32_t publish_addr, uint32_t P, uint32_t n_tiles,
                                                               const uint32_t* ckpt_pos, const uint64_t* ckpt_state, uint32_t ckpt_tiles,
                                                               uint32_t ckpt_pos_off, uint32_t ckpt_state_off) {
#include "cst_encode_loop_pc_n8_ck.inc"
}

// ... at 12 < P <= 24: unpacked entries {c, p, floor(2^64 / p)} and the step of cst_encode_loop_wide.inc (round 5)
__device__ __forceinline__ void ans_encode_pc_n8w_coder_loop(uint32_t& lo, uint32_t& hi, int32_t& smin, int32_t& smax, uint32_t line_row_addr,
                                                             uint32_t ring_lane_addr, uint32_t publish_addr, uint32_t P, uint32_t n_tiles) {
#include "cst_encode_loop_pc_n8w.inc"
}

__device__ __forceinline__ void ans_encode_pc_n8w_coder_loop_ck(uint32_t& lo, uint32_t& hi, int32_t& smin, int32_t& smax, uint32_t line_row_addr,
                                                                uint32_t ring_lane_addr, uint32_t publish_addr, uint32_t P, uint32_t n_tiles,
                                                                const uint32_t* ckpt_pos, const uint64_t* ckpt_state, uint32_t ckpt_tiles,
                                                                uint32_t ckpt_pos_off, uint32_t ckpt_state_off) {
#include "cst_encode_loop_pc_n8w_ck.inc"
}

__device__ __forceinline__ void ans_encode_pc_n8_loader_loop(const uint32_t (&line_tr_addr)[2], uint64_t symbols_base, uint32_t row_block_bytes,
                                                             uint32_t n_tiles, const uint32_t (&goff0)[8], const uint32_t (&goff1)[8]) {
#include "cst_encode_loop_pc_loader_n8.inc"
}

template <bool JUMP, bool WIDE = false>
__global__ __launch_bounds__(kPcThreads) void ans_encode_pc_n8_kernel(const AnsEncodeArgs a, const PcJumpArgs jp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int cw = wave & (kPcWaves - 1);
    const int P = a.precision;
    const size_t N = a.n_per_stream;
    const uint32_t n_t = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(N / kTileSyms));
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;
    const int8_t* symbols = reinterpret_cast<const int8_t*>(a.symbols);      // (the int8 matrix travels in the int32 field of the argument block)

    // the table, by symbol VALUE: entry of symbol s at 2048 + 16 s (a value outside the support: some harmless entry -- its
    // stream is flagged by the range check and its words are never used)
    if (lds_addr(smem) != 0) __builtin_trap();          // (the coder's table reads carry the table's address as an immediate)
    EncEntry* table = reinterpret_cast<EncEntry*>(smem);
    for (int u = threadIdx.x; u < 256; u += kPcThreads) {
        const int idx = (u - 128) - a.min_symbol;
        const bool in_support = idx >= 0 && idx < a.n_symbols;
        EncEntry e = a.enc[in_support ? idx : 0];
        if constexpr (!WIDE) e = pack_entry(e, P);
        // the range check's flag (n8_fold_minmax of the generator): a bit no real entry has in its first word (c < 2^12 packed with
        // c + 2^P - p < 2^13 above it; c < 2^24 unpacked)
        if (!in_support) e.c |= WIDE ? 0x80000000u : 0x8000u;
        table[u] = e;
    }

    if (wave < kPcWaves) {                               // ---- coder ----
        uint32_t* ring = reinterpret_cast<uint32_t*>(smem + kPcRingOff + cw * kPcRingWaveBytes);
        uint32_t* hand = reinterpret_cast<uint32_t*>(smem + kPcN8HandOff + cw * kPcHandWaveBytes);
        // a partial workgroup: the lanes behind the last stream code the LAST stream again (the loader stages its symbols for them);
        // what they produce is never stored (pc_storer) except the jump points, which they write where its own lane writes them
        const size_t s = min((size_t)blockIdx.x * kBlock + (size_t)cw * kWave + lane, a.n_streams - 1);
        uint32_t lo = 0, hi = 0;
        int32_t smin = 0, smax = 0;                      // (the statement ORs the first words of the entries it codes into smax: bit 15 = outside the support)
        if (raw) { const uint64_t st = a.state[s]; lo = (uint32_t)st; hi = (uint32_t)(st >> 32); }
        const uint32_t row_addr = lds_addr(smem + kPcN8TileOff + (2 * cw) * kPcN8LineBytes) + (uint32_t)(lane * kPcN8RowBytes);
        pc_barrier();                                   // table and the first line are in LDS
        const uint32_t jpos = (uint32_t)((s * jp.n_chunks + jp.n_chunks - 1) * 4), jstate = (uint32_t)((s * jp.n_chunks + jp.n_chunks - 1) * 8);
        if constexpr (JUMP && WIDE)
            ans_encode_pc_n8w_coder_loop_ck(lo, hi, smin, smax, row_addr, lds_addr(ring + lane), lds_addr(hand + lane), (uint32_t)P, n_t, jp.pos, jp.state,
                                            jp.tiles, jpos, jstate);
        else if constexpr (JUMP)
            ans_encode_pc_n8_coder_loop_ck(lo, hi, smin, smax, row_addr, lds_addr(ring + lane), lds_addr(hand + lane), (uint32_t)P, n_t, jp.pos, jp.state,
                                           jp.tiles, jpos, jstate);
        else if constexpr (WIDE)
            ans_encode_pc_n8w_coder_loop(lo, hi, smin, smax, row_addr, lds_addr(ring + lane), lds_addr(hand + lane), (uint32_t)P, n_t);
        else
            ans_encode_pc_n8_coder_loop(lo, hi, smin, smax, row_addr, lds_addr(ring + lane), lds_addr(hand + lane), (uint32_t)P, n_t);
        hand[kWave + lane] = lo; hand[2 * kWave + lane] = hi;
        (void)smin;
        hand[3 * kWave + lane] = ((uint32_t)smax & (WIDE ? 0x80000000u : 0x8000u)) ? 0xffffffffu : 0u;   // "largest table index seen": beyond every alphabet, or inside
        pc_barrier();                                   // the last window and the final state are published
        return;
    }
    const int pair = wave & 1, cw0 = 2 * pair;          // the pair's first coder wave
    const size_t s0 = (size_t)blockIdx.x * kBlock + (size_t)cw0 * kWave;
    if (wave < kPcWaves + 2) {                          // ---- loader ----
        // rows of the two coder waves, clamped to the last stream (a partial workgroup: see the coder above)
        const size_t last = a.n_streams - 1;
        const size_t first0 = min(s0, last), first1 = min(s0 + kWave, last);
        uint32_t goff0[8], goff1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const size_t row = (size_t)(lane >> 3) + 8 * k;
            goff0[k] = (uint32_t)(min(row, last - first0) * N + 16 * (size_t)(lane & 7));
            goff1[k] = (uint32_t)(min(row, last - first1) * N + 16 * (size_t)(lane & 7));
        }
        const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(symbols + first0 * N + (N - kPcN8LineSyms));
        const uint64_t symbols_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                      (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
        const uint32_t t0 = lds_addr(smem + kPcN8TileOff + (2 * cw0) * kPcN8LineBytes) + (uint32_t)((lane >> 3) * kPcN8RowBytes + 16 * (lane & 7));
        const uint32_t tr_addr[2] = {t0, t0 + (uint32_t)kPcN8LineBytes};
        const uint32_t row_block = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)((first1 - first0) * N));
        __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0): the statement keeps its own book from here
        ans_encode_pc_n8_loader_loop(tr_addr, symbols_base, row_block, n_t, goff0, goff1);
        pc_barrier();
        return;
    }
    pc_storer<WIDE>(a, smem, lane, cw0, s0, n_t, kPcRingOff, kPcN8HandOff);
}

// ---------------------------------------------------------------------------------------------------------------------
// INT16 symbol matrices (scripts/gen_encode_loop_pc.py, "INT16 symbol matrices"): the int8 workgroup with lines of 64 symbols (two tiles).
// The table is the int32 kernel's (at most 1024 entries, at LDS address 0 .. 16 KiB); the coder forms  table + 16 (symbol - min)
// with one v_mad_i32_i16 per symbol (op_sel picks the half of the dword) and folds those addresses for the range check.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ans_encode_pc_n16_coder_loop(uint32_t& lo, uint32_t& hi, uint32_t& smin, uint32_t& smax, uint32_t line_row_addr,
                                                             uint32_t ring_lane_addr, uint32_t publish_addr, uint32_t table_bias, uint32_t P, uint32_t n_tiles) {
#include "cst_encode_loop_pc_n16.inc"
}

__device__ __forceinline__ void ans_encode_pc_n16_coder_loop_ck(uint32_t& lo, uint32_t& hi, uint32_t& smin, uint32_t& smax, uint32_t line_row_addr,
                                                                uint32_t ring_lane_addr, uint32_t publish_addr, uint32_t table_bias, uint32_t P,
                                                                uint32_t n_tiles, const uint32_t* ckpt_pos, const uint64_t* ckpt_state, uint32_t ckpt_tiles,
                                                                uint32_t ckpt_pos_off, uint32_t ckpt_state_off) {
#include "cst_encode_loop_pc_n16_ck.inc"
}

__device__ __forceinline__ void ans_encode_pc_n16w_coder_loop(uint32_t& lo, uint32_t& hi, uint32_t& smin, uint32_t& smax, uint32_t line_row_addr,
                                                              uint32_t ring_lane_addr, uint32_t publish_addr, uint32_t table_bias, uint32_t P, uint32_t n_tiles) {
#include "cst_encode_loop_pc_n16w.inc"
}

__device__ __forceinline__ void ans_encode_pc_n16w_coder_loop_ck(uint32_t& lo, uint32_t& hi, uint32_t& smin, uint32_t& smax, uint32_t line_row_addr,
                                                                 uint32_t ring_lane_addr, uint32_t publish_addr, uint32_t table_bias, uint32_t P,
                                                                 uint32_t n_tiles, const uint32_t* ckpt_pos, const uint64_t* ckpt_state, uint32_t ckpt_tiles,
                                                                 uint32_t ckpt_pos_off, uint32_t ckpt_state_off) {
#include "cst_encode_loop_pc_n16w_ck.inc"
}

__device__ __forceinline__ void ans_encode_pc_n16_loader_loop(const uint32_t (&line_tr_addr)[2], uint64_t symbols_base, uint32_t row_block_bytes,
                                                              uint32_t n_tiles, const uint32_t (&goff0)[8], const uint32_t (&goff1)[8]) {
#include "cst_encode_loop_pc_loader_n16.inc"
}

template <bool JUMP, bool WIDE = false>
__global__ __launch_bounds__(kPcThreads) void ans_encode_pc_n16_kernel(const AnsEncodeArgs a, const PcJumpArgs jp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int cw = wave & (kPcWaves - 1);
    const int P = a.precision;
    const size_t N = a.n_per_stream;
    const uint32_t n_t = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(N / kTileSyms));
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;
    const int8_t* symbols = reinterpret_cast<const int8_t*>(a.symbols);      // (BYTE addresses of the int16 matrix below)
    const size_t row_bytes = N * 2;

    EncEntry* table = reinterpret_cast<EncEntry*>(smem);
    for (int i = threadIdx.x; i < a.n_symbols; i += kPcThreads) table[i] = WIDE ? a.enc[i] : pack_entry(a.enc[i], P);

    if (wave < kPcWaves) {                               // ---- coder ----
        uint32_t* ring = reinterpret_cast<uint32_t*>(smem + kPcRingOff + cw * kPcRingWaveBytes);
        uint32_t* hand = reinterpret_cast<uint32_t*>(smem + kPcN8HandOff + cw * kPcHandWaveBytes);
        const size_t s = min((size_t)blockIdx.x * kBlock + (size_t)cw * kWave + lane, a.n_streams - 1);      // (partial workgroups: as for int8)
        uint32_t lo = 0, hi = 0;
        const uint32_t table_addr = lds_addr(table);
        uint32_t smin = table_addr, smax = table_addr;    // (the statement folds the table ADDRESSES it reads: unsigned min / max)
        if (raw) { const uint64_t st = a.state[s]; lo = (uint32_t)st; hi = (uint32_t)(st >> 32); }
        const uint32_t row_addr = lds_addr(smem + kPcN8TileOff + (2 * cw) * kPcN8LineBytes) + (uint32_t)(lane * kPcN8RowBytes);
        const uint32_t bias = table_addr - 16u * (uint32_t)a.min_symbol;
        pc_barrier();                                   // table and the first line are in LDS
        const uint32_t jpos = (uint32_t)((s * jp.n_chunks + jp.n_chunks - 1) * 4), jstate = (uint32_t)((s * jp.n_chunks + jp.n_chunks - 1) * 8);
        if constexpr (JUMP && WIDE)
            ans_encode_pc_n16w_coder_loop_ck(lo, hi, smin, smax, row_addr, lds_addr(ring + lane), lds_addr(hand + lane), bias, (uint32_t)P, n_t, jp.pos,
                                             jp.state, jp.tiles, jpos, jstate);
        else if constexpr (JUMP)
            ans_encode_pc_n16_coder_loop_ck(lo, hi, smin, smax, row_addr, lds_addr(ring + lane), lds_addr(hand + lane), bias, (uint32_t)P, n_t, jp.pos,
                                            jp.state, jp.tiles, jpos, jstate);
        else if constexpr (WIDE)
            ans_encode_pc_n16w_coder_loop(lo, hi, smin, smax, row_addr, lds_addr(ring + lane), lds_addr(hand + lane), bias, (uint32_t)P, n_t);
        else
            ans_encode_pc_n16_coder_loop(lo, hi, smin, smax, row_addr, lds_addr(ring + lane), lds_addr(hand + lane), bias, (uint32_t)P, n_t);
        hand[kWave + lane] = lo; hand[2 * kWave + lane] = hi;
        // largest raw table index seen: an address below the table wraps to a huge one
        hand[3 * kWave + lane] = max((smax - table_addr) >> 4, (smin - table_addr) >> 4);
        pc_barrier();                                   // the last window and the final state are published
        return;
    }
    const int pair = wave & 1, cw0 = 2 * pair;
    const size_t s0 = (size_t)blockIdx.x * kBlock + (size_t)cw0 * kWave;
    if (wave < kPcWaves + 2) {                          // ---- loader ----
        const size_t last = a.n_streams - 1;
        const size_t first0 = min(s0, last), first1 = min(s0 + kWave, last);
        uint32_t goff0[8], goff1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const size_t row = (size_t)(lane >> 3) + 8 * k;
            goff0[k] = (uint32_t)(min(row, last - first0) * row_bytes + 16 * (size_t)(lane & 7));
            goff1[k] = (uint32_t)(min(row, last - first1) * row_bytes + 16 * (size_t)(lane & 7));
        }
        const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(symbols + first0 * row_bytes + (row_bytes - 128));
        const uint64_t symbols_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                      (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
        const uint32_t t0 = lds_addr(smem + kPcN8TileOff + (2 * cw0) * kPcN8LineBytes) + (uint32_t)((lane >> 3) * kPcN8RowBytes + 16 * (lane & 7));
        const uint32_t tr_addr[2] = {t0, t0 + (uint32_t)kPcN8LineBytes};
        const uint32_t row_block = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)((first1 - first0) * row_bytes));
        __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0): the statement keeps its own book from here
        ans_encode_pc_n16_loader_loop(tr_addr, symbols_base, row_block, n_t, goff0, goff1);
        pc_barrier();
        return;
    }
    pc_storer<WIDE>(a, smem, lane, cw0, s0, n_t, kPcRingOff, kPcN8HandOff);
}

// Rows that are whole 128-byte aligned lines of 64 int16 symbols, a table of at most 1024 entries, slabs as for the int32 kernel.
bool pc_n16_encode_usable(const AnsEncodeArgs& a, cst_coder_config cfg, cst_layout layout) {
    if (knobs().no_n8 || knobs().no_pc_encoder) return false;      // (A/B runs: the conversion path)
    if (cfg.word_bits != 32 || cfg.state_bits != 64 || layout != CST_LAYOUT_STREAM_MAJOR || a.precision < 8 || a.precision > 24) return false;
    if (a.n_streams == 0) return false;
    if (a.n_per_stream % 64 != 0 || a.n_per_stream == 0 || a.n_per_stream >= (1u << 23)) return false;
    if ((reinterpret_cast<uintptr_t>(a.symbols) & 127) != 0) return false;
    if ((reinterpret_cast<uintptr_t>(a.words) & 63) != 0 || a.stride_words % 16 != 0 || a.stride_words == 0) return false;
    if ((a.n_streams + kBlock - 1) / kBlock * kBlock * a.stride_words * 4 >= 0x100000000ull) return false;                  // 32-bit slab offsets
    if (a.n_symbols < 1 || (size_t)a.n_symbols * sizeof(EncEntry) > kPcTableBytes) return false;
    return a.min_symbol >= -32768 && a.min_symbol + a.n_symbols - 1 <= 32767;
}

cst_status ans_encode_pc_n16(const AnsEncodeArgs& a, size_t interval, uint32_t* d_ckpt_pos, uint64_t* d_ckpt_state, hipStream_t hs) {
    const size_t blocks = (a.n_streams + kBlock - 1) / kBlock;
    const bool wide = a.precision > 12;
    if (interval) {
        auto kernel = wide ? ans_encode_pc_n16_kernel<true, true> : ans_encode_pc_n16_kernel<true>;
        CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPcN8LdsBytes));
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kPcThreads), kPcN8LdsBytes, hs, a,
                           PcJumpArgs{d_ckpt_pos, d_ckpt_state, (uint32_t)(interval / kTileSyms), (uint32_t)(a.n_per_stream / interval)});
    } else {
        auto kernel = wide ? ans_encode_pc_n16_kernel<false, true> : ans_encode_pc_n16_kernel<false>;
        CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPcN8LdsBytes));
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kPcThreads), kPcN8LdsBytes, hs, a, PcJumpArgs{nullptr, nullptr, 0u, 0u});
    }
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

bool pc_n16_encode_ckpt_usable(const AnsEncodeArgs& a, cst_coder_config cfg, cst_layout layout, size_t interval);

// Rows that are whole 128-byte aligned lines, a support inside int8, slabs as for the int32 kernel; any number of streams.
bool pc_n8_encode_usable(const AnsEncodeArgs& a, cst_coder_config cfg, cst_layout layout) {
    if (knobs().no_n8 || knobs().no_pc_encoder) return false;      // (A/B runs: the conversion path)
    if (cfg.word_bits != 32 || cfg.state_bits != 64 || layout != CST_LAYOUT_STREAM_MAJOR || a.precision < 8 || a.precision > 24) return false;
    if (a.n_streams == 0) return false;                  // (partial workgroups are taken: their spare lanes repeat the last stream)
    if (a.n_per_stream % kPcN8LineSyms != 0 || a.n_per_stream == 0 || a.n_per_stream >= (1u << 24)) return false;
    if ((reinterpret_cast<uintptr_t>(a.symbols) & 127) != 0) return false;
    if ((reinterpret_cast<uintptr_t>(a.words) & 63) != 0 || a.stride_words % 16 != 0 || a.stride_words == 0) return false;
    if ((a.n_streams + kBlock - 1) / kBlock * kBlock * a.stride_words * 4 >= 0x100000000ull) return false;                  // 32-bit slab offsets
    if (a.n_symbols < 1 || a.n_symbols > 256 || a.min_symbol < -128 || a.min_symbol + a.n_symbols - 1 > 127) return false;
    return true;
}

// jump points the coder waves can note on their way: chunks of whole tiles that divide the rows, 32-bit offsets into the two arrays
static bool pc_jump_ok(const AnsEncodeArgs& a, size_t interval) {
    if (interval == 0 || interval % kTileSyms != 0 || a.n_per_stream % interval != 0) return false;
    return a.n_streams * (a.n_per_stream / interval) * 8 < 0x100000000ull;
}

bool pc_n16_encode_ckpt_usable(const AnsEncodeArgs& a, cst_coder_config cfg, cst_layout layout, size_t interval) {
    return pc_n16_encode_usable(a, cfg, layout) && pc_jump_ok(a, interval);
}

static PcJumpArgs pc_jump_args(const AnsEncodeArgs& a, size_t interval, uint32_t* pos, uint64_t* state) {
    return PcJumpArgs{pos, state, (uint32_t)(interval / kTileSyms), (uint32_t)(a.n_per_stream / interval)};
}

cst_status ans_encode_pc_n8(const AnsEncodeArgs& a, hipStream_t hs) {
    const size_t blocks = (a.n_streams + kBlock - 1) / kBlock;
    auto kernel = a.precision > 12 ? ans_encode_pc_n8_kernel<false, true> : ans_encode_pc_n8_kernel<false>;
    CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPcN8LdsBytes));
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kPcThreads), kPcN8LdsBytes, hs, a, PcJumpArgs{nullptr, nullptr, 0u, 0u});
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

bool pc_n8_encode_ckpt_usable(const AnsEncodeArgs& a, cst_coder_config cfg, cst_layout layout, size_t interval) {
    return pc_n8_encode_usable(a, cfg, layout) && pc_jump_ok(a, interval);
}

cst_status ans_encode_pc_n8_ckpt(const AnsEncodeArgs& a, size_t interval, uint32_t* d_ckpt_pos, uint64_t* d_ckpt_state, hipStream_t hs) {
    const size_t blocks = (a.n_streams + kBlock - 1) / kBlock;
    auto kernel = a.precision > 12 ? ans_encode_pc_n8_kernel<true, true> : ans_encode_pc_n8_kernel<true>;
    CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPcN8LdsBytes));
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kPcThreads), kPcN8LdsBytes, hs, a, pc_jump_args(a, interval, d_ckpt_pos, d_ckpt_state));
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

// Any number of streams (partial workgroups since round 5), at most one workgroup per CU (more streams than that: the two-waves-per-SIMD kernels of
// cst_ans_small.hip), rows that are whole 128-byte aligned tiles, 64-byte aligned slabs of whole 64-byte groups.
bool pc_encode_usable(const AnsEncodeArgs& a, cst_coder_config cfg, cst_layout layout, int device_cus) {
    if (knobs().no_pc_encoder) return false;      // (A/B runs)
    if (cfg.word_bits != 32 || layout != CST_LAYOUT_STREAM_MAJOR || a.precision < 8 || a.precision > 24) return false;
    // 12 < P <= 24 (round 5: the wide step in the coder waves, two word groups per tile in the storers; ans_encode_wide_kernel before)
    if (a.precision > 12 && (knobs().no_pc_wide || knobs().pc_combined)) return false;
    (void)device_cus;     // (more than one workgroup per CU: they run one after another, cst_api.hip asks the small-footprint kernels first)
    if (a.n_streams == 0) return false;
    if (a.n_streams % kBlock != 0 && knobs().pc_combined) return false;                      // (partial workgroups: the split helpers only)
    if (a.n_per_stream % kTileSyms != 0 || a.n_per_stream < 2 * kTileSyms || a.n_per_stream >= (1u << 24)) return false;
    if ((reinterpret_cast<uintptr_t>(a.symbols) & 127) != 0) return false;
    if ((reinterpret_cast<uintptr_t>(a.words) & 63) != 0 || a.stride_words % 16 != 0 || a.stride_words == 0) return false;
    if ((a.n_streams + kBlock - 1) / kBlock * kBlock * a.stride_words * 4 >= 0x100000000ull || 64 * a.n_per_stream * 4 >= 0x100000000ull) return false;   // 32-bit offsets
    return (size_t)a.n_symbols * sizeof(EncEntry) <= kPcTableBytes;
}

cst_status ans_encode_pc(const AnsEncodeArgs& a, hipStream_t hs) {
    const size_t blocks = (a.n_streams + kBlock - 1) / kBlock;
    const bool combined = knobs().pc_combined;      // (A/B runs: every helper wave loads AND stores)
    auto kernel = a.precision > 12 ? ans_encode_pc_kernel<true, false, true> : combined ? ans_encode_pc_kernel<false> : ans_encode_pc_kernel<true>;
    CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPcLdsBytes));
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kPcThreads), kPcLdsBytes, hs, a, PcJumpArgs{nullptr, nullptr, 0u, 0u});
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

// ... noting jump points (cst_ans_encode_batch_ckpt on shared-table models: the one-lane-per-stream kernel of cst_ans_ckpt.hip took
// 1.4 ms at 65 536 x 4096; this one runs at the speed of the plain encoder)
bool pc_encode_ckpt_usable(const AnsEncodeArgs& a, cst_coder_config cfg, cst_layout layout, size_t interval) {
    return pc_encode_usable(a, cfg, layout, 0) && pc_jump_ok(a, interval);
}

cst_status ans_encode_pc_ckpt(const AnsEncodeArgs& a, size_t interval, uint32_t* d_ckpt_pos, uint64_t* d_ckpt_state, hipStream_t hs) {
    const size_t blocks = (a.n_streams + kBlock - 1) / kBlock;
    auto kernel = a.precision > 12 ? ans_encode_pc_kernel<true, true, true> : ans_encode_pc_kernel<true, true>;
    CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPcLdsBytes));
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kPcThreads), kPcLdsBytes, hs, a, pc_jump_args(a, interval, d_ckpt_pos, d_ckpt_state));
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

} // namespace cst
