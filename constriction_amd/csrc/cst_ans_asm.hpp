// cst_ans_asm.hpp -- the hand-scheduled gfx950 statements of the default preset (W,S) = (32,64):
//   ans_encode_step_asm      one encoder step                       (8 <= P <= 24, used by the generic tile loop)
//   ans_encode_tile32        32 encoder steps, software pipelined   (P <= 12)
//   ans_encode_tiles_loop    the encoder's whole main loop          (generated: cst_encode_loop.inc; symbol-major: cst_encode_loop_sm.inc)
//   ans_decode_tile32        32 decoder steps                       (P <= 12)
//   ans_decode_tiles_loop    the decoder's whole main loop          (generated: cst_decode_loop.inc; symbol-major: cst_decode_loop_sm.inc)
// Why inline asm at all, the machine model behind the schedules and the measured results: DESIGN.md 3.6-3.8.
// The loop statements are emitted by scripts/gen_{encode,decode}_loop.py, which also keep the s_waitcnt book.
#pragma once
#include "cst_common.hpp"

namespace cst {

// Hand-scheduled form of the same step (W,S) = (32,64), 8 <= P <= 24: 22 instructions, fixed scratch registers
// v[120:134] so that every 64-bit operand of v_mad_u64_u32 is a register pair written in place (the compiler's
// version spends ~10 v_mov per step assembling such pairs).  Also performs the ring write of the candidate word and
// the emit-count update.  Hazards: two instructions separate every VALU write of vcc from its VALU reader.
//   ring_addr : LDS byte address of this lane's ring slot for word index `wr`
//   pshl = p << (32-P), k = 2^P - p, ck = c + k
__device__ __forceinline__ void ans_encode_step_asm(uint32_t& lo, uint32_t& hi, uint32_t& wr, uint32_t ring_addr,
                                                    const EncEntry e, uint32_t pshl, uint32_t k, uint32_t ck) {
    uint64_t sdummy;
    asm volatile(
        "v_cmp_ge_u32 vcc, %[hi], %[pshl]\n\t"                       // emit <=> (state >> (64-P)) >= p
        "ds_write_b32 %[ra], %[lo]\n\t"                             // candidate word, always written
        "v_mov_b32 v123, 0\n\t"
        "v_cndmask_b32_e64 v120, %[lo], %[hi], vcc\n\t"             // a0 = emit ? hi : lo
        "v_cndmask_b32_e64 v121, %[hi], 0, vcc\n\t"                 // a1 = emit ? 0 : hi          A = v[120:121]
        "v_addc_co_u32 %[wr], vcc, 0, %[wr], vcc\n\t"               // wr += emit
        "v_mul_hi_u32 v122, v120, %[m0]\n\t"                        // W = [hi32(a0*m0), 0]
        "v_mov_b32 v127, 0\n\t"
        "v_mad_u64_u32 v[124:125], vcc, v121, %[m0], v[122:123]\n\t" // U = a1*m0 + W
        "v_mov_b32 v126, v124\n\t"                                  // X = [U_lo, 0]
        "v_mad_u64_u32 v[128:129], vcc, v120, %[m1], v[126:127]\n\t" // V = a0*m1 + U_lo
        "v_add_co_u32 v130, vcc, v125, v129\n\t"                    // S = U_hi + V_hi (33 bits)
        "v_addc_co_u32 v131, vcc, 0, v123, vcc\n\t"
        "v_mad_u64_u32 v[132:133], vcc, v121, %[m1], v[130:131]\n\t" // Q = a1*m1 + S = q_est in {q-1, q}
        "v_mul_lo_u32 v134, v132, %[p]\n\t"
        "v_sub_u32 v134, v120, v134\n\t"                            // estimated remainder (true value < 2p)
        "v_cmp_ge_u32 vcc, v134, %[p]\n\t"                          // fix <=> q = q_est + 1
        "v_mad_u64_u32 v[124:125], %[sd], v132, %[k], v[120:121]\n\t" // T = A + q_lo*k
        "v_mad_u32_u24 v125, v133, %[k], v125\n\t"                  // T_hi += q_hi*k   (q_hi < 2^24, k < 2^24)
        "v_cndmask_b32 v134, %[c], %[ck], vcc\n\t"                  // d = c + (fix ? k : 0)
        "v_add_co_u32 %[lo], vcc, v124, v134\n\t"                   // state' = T + d
        "v_addc_co_u32 %[hi], vcc, 0, v125, vcc"
        : [lo] "+v"(lo), [hi] "+v"(hi), [wr] "+v"(wr), [sd] "=&s"(sdummy)
        : [ra] "v"(ring_addr), [pshl] "v"(pshl), [m0] "v"(e.m_lo), [m1] "v"(e.m_hi), [p] "v"(e.p), [k] "v"(k), [c] "v"(e.c),
          [ck] "v"(ck)
        : "vcc", "memory", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131",
          "v132", "v133", "v134");
}

// ------------------------------------------------------------------------------------------------
// Hand-scheduled encode of one 32-symbol tile for (W,S) = (32,64), 8 <= P <= 12  
//
// The encoder's table entries do not depend on the coder state, so the only serial chain is arithmetic and the
// tile is bound by instruction issue (one instruction of any kind per ~4.7 cycles for a lone wave).  One asm
// statement per tile: 23 instructions per symbol for the step itself (PACKED table entries, see CST_ENC_STEP; the
// zero halves of its register pairs set once per tile) plus 3.75 for the software pipeline around it:
//   quad j (4 symbols, walked backwards):  request the symbols of quad j-2 (one 16-B LDS read of the lane's tile
//   row), fetch the four 16-B table entries of quad j-1, then run the four steps of quad j.
// LDS returns in order, so the hand-counted waits are: lgkmcnt(8) = "symbols of quad j-1 are back" (4 entry
// reads and 4 ring writes are younger), lgkmcnt(9) = "entries of quad j are back".
// A symbol outside the model's support reads a garbage entry (LDS never faults) and only corrupts its own stream,
// which is reported through smin/smax (the caller turns them into CST_STREAM_IMPOSSIBLE_SYMBOL).
//   v100..v111  three symbol quads     v112..v143  two sets of four packed entries
//   v[144:145] A   v[146:147] [w,0]   v[148:149] U / q_est << P   v[150:151] [U_lo,0]   v[152:153] V   v[154:155] sum
//   v[156:157] Q   v158 r   v160 k = 2^P - p   v161 c (+ k)   v162 ring address   v163 entry address
// ------------------------------------------------------------------------------------------------
#define CST_ENC_SDWA " dst_sel:DWORD dst_unused:UNUSED_PAD "
// (E0, E1, M0, M1) = packed table entry { c | (c + 2^P - p) << 16,  p | p << (32 - P),  floor(2^64 / p) }: see
// scripts/gen_encode_loop.py step() for the arithmetic -- 22 VALU instructions + the ring write
#define CST_ENC_STEP(E0, E1, M0, M1)                                                                                \
    "v_cmp_ge_u32_sdwa vcc, %[hi], " E1 " src0_sel:WORD_1 src1_sel:WORD_1\n\t"                                      \
    "v_sub_u32_sdwa v160, %[twoP], " E1 CST_ENC_SDWA "src0_sel:DWORD src1_sel:WORD_0\n\t"                            \
    "v_add_lshl_u32 v162, %[wr], %[shift], 8\n\t"                                                                   \
    "v_and_or_b32 v162, v162, %[c3f00], %[lanebase]\n\t"                                                            \
    "v_cndmask_b32_e64 v144, %[lo], %[hi], vcc\n\t"                                                                 \
    "v_cndmask_b32_e64 v145, %[hi], 0, vcc\n\t"                                                                     \
    "ds_write_b32 v162, %[lo]\n\t"                                                                                  \
    "v_addc_co_u32 %[wr], vcc, 0, %[wr], vcc\n\t"                                                                   \
    "v_mul_hi_u32 v146, v144, " M0 "\n\t"                                                                           \
    "v_mad_u64_u32 v[148:149], vcc, v145, " M0 ", v[146:147]\n\t"                                                   \
    "v_mov_b32 v150, v148\n\t"                                                                                      \
    "v_mad_u64_u32 v[152:153], vcc, v144, " M1 ", v[150:151]\n\t"                                                   \
    "v_add_co_u32 v154, vcc, v149, v153\n\t"                                                                        \
    "v_addc_co_u32 v155, vcc, 0, v147, vcc\n\t"                                                                     \
    "v_mad_u64_u32 v[156:157], vcc, v145, " M1 ", v[154:155]\n\t"                                                   \
    "v_mul_u32_u24_sdwa v158, v156, " E1 CST_ENC_SDWA "src0_sel:DWORD src1_sel:WORD_0\n\t"                           \
    "v_sub_u32 v158, v144, v158\n\t"                                                                                \
    "v_cmp_ge_u32_sdwa vcc, v158, " E1 " src0_sel:WORD_0 src1_sel:WORD_0\n\t"                                        \
    "v_mad_u64_u32 v[148:149], %[sd], v156, v160, v[144:145]\n\t"                                                   \
    "v_mad_u32_u24 v149, v157, v160, v149\n\t"                                                                      \
    "v_cndmask_b32_sdwa v161, " E0 ", " E0 ", vcc" CST_ENC_SDWA "src0_sel:WORD_0 src1_sel:WORD_1\n\t"                  \
    "v_add_co_u32 %[lo], vcc, v148, v161\n\t"                                                                       \
    "v_addc_co_u32 %[hi], vcc, 0, v149, vcc\n\t"

// entry sets (consumption order: symbol .w first)
#define CST_ENC_STEPS_E0 CST_ENC_STEP("v112", "v113", "v114", "v115") CST_ENC_STEP("v116", "v117", "v118", "v119")  \
                         CST_ENC_STEP("v120", "v121", "v122", "v123") CST_ENC_STEP("v124", "v125", "v126", "v127")
#define CST_ENC_STEPS_E1 CST_ENC_STEP("v128", "v129", "v130", "v131") CST_ENC_STEP("v132", "v133", "v134", "v135")  \
                         CST_ENC_STEP("v136", "v137", "v138", "v139") CST_ENC_STEP("v140", "v141", "v142", "v143")
#define CST_ENC_FETCH1(SYM, E) "v_lshl_add_u32 v163, " SYM ", 4, %[tbl]\n\tds_read_b128 " E ", v163\n\t"
// fetch the entries of symbols (X,Y,Z,W) of a quad into a set, .w first, and fold the quad into smin/smax
#define CST_ENC_FETCH_E0(X, Y, Z, W)                                                                                \
    CST_ENC_FETCH1(W, "v[112:115]") CST_ENC_FETCH1(Z, "v[116:119]") CST_ENC_FETCH1(Y, "v[120:123]") CST_ENC_FETCH1(X, "v[124:127]") \
    CST_ENC_MINMAX(X, Y, Z, W)
#define CST_ENC_FETCH_E1(X, Y, Z, W)                                                                                \
    CST_ENC_FETCH1(W, "v[128:131]") CST_ENC_FETCH1(Z, "v[132:135]") CST_ENC_FETCH1(Y, "v[136:139]") CST_ENC_FETCH1(X, "v[140:143]") \
    CST_ENC_MINMAX(X, Y, Z, W)
#define CST_ENC_MINMAX(X, Y, Z, W)                                                                                  \
    "v_max3_i32 %[smax], %[smax], " X ", " Y "\n\tv_max3_i32 %[smax], %[smax], " Z ", " W "\n\t"                     \
    "v_min3_i32 %[smin], %[smin], " X ", " Y "\n\tv_min3_i32 %[smin], %[smin], " Z ", " W "\n\t"
#define CST_ENC_S0 "v100", "v101", "v102", "v103"
#define CST_ENC_S1 "v104", "v105", "v106", "v107"
#define CST_ENC_S2 "v108", "v109", "v110", "v111"
#define CST_ENC_FETCH_E0_(S) CST_ENC_FETCH_E0(S)
#define CST_ENC_FETCH_E1_(S) CST_ENC_FETCH_E1(S)

// Encodes symbols [31 .. 0] of the lane's tile row (LDS), last symbol first.  On return every LDS operation of the
// statement has completed.  smin/smax accumulate the smallest/largest symbol seen.
__device__ __forceinline__ void ans_encode_tile32(uint32_t& lo, uint32_t& hi, uint32_t& wr, int32_t& smin, int32_t& smax,
                                                  uint32_t tile_row_addr, uint32_t table_addr_biased, uint32_t P,
                                                  uint32_t shift, uint32_t ring_lane_addr) {
    uint64_t sd;
    asm volatile(
        "v_mov_b32 v147, 0\n\t"
        "v_mov_b32 v151, 0\n\t"
        "ds_read_b128 v[100:103], %[tile] offset:112\n\t"      // quad 7 -> S0
        "ds_read_b128 v[104:107], %[tile] offset:96\n\t"       // quad 6 -> S1
        "s_waitcnt lgkmcnt(1)\n\t"
        CST_ENC_FETCH_E0_(CST_ENC_S0)
        // quad 7
        "s_waitcnt lgkmcnt(4)\n\t"
        "ds_read_b128 v[108:111], %[tile] offset:80\n\t"       // quad 5 -> S2
        CST_ENC_FETCH_E1_(CST_ENC_S1)
        "s_waitcnt lgkmcnt(5)\n\t"
        CST_ENC_STEPS_E0
        // quad 6
        "s_waitcnt lgkmcnt(8)\n\t"
        "ds_read_b128 v[100:103], %[tile] offset:64\n\t"       // quad 4 -> S0
        CST_ENC_FETCH_E0_(CST_ENC_S2)
        "s_waitcnt lgkmcnt(9)\n\t"
        CST_ENC_STEPS_E1
        // quad 5
        "s_waitcnt lgkmcnt(8)\n\t"
        "ds_read_b128 v[104:107], %[tile] offset:48\n\t"       // quad 3 -> S1
        CST_ENC_FETCH_E1_(CST_ENC_S0)
        "s_waitcnt lgkmcnt(9)\n\t"
        CST_ENC_STEPS_E0
        // quad 4
        "s_waitcnt lgkmcnt(8)\n\t"
        "ds_read_b128 v[108:111], %[tile] offset:32\n\t"       // quad 2 -> S2
        CST_ENC_FETCH_E0_(CST_ENC_S1)
        "s_waitcnt lgkmcnt(9)\n\t"
        CST_ENC_STEPS_E1
        // quad 3
        "s_waitcnt lgkmcnt(8)\n\t"
        "ds_read_b128 v[100:103], %[tile] offset:16\n\t"       // quad 1 -> S0
        CST_ENC_FETCH_E1_(CST_ENC_S2)
        "s_waitcnt lgkmcnt(9)\n\t"
        CST_ENC_STEPS_E0
        // quad 2
        "s_waitcnt lgkmcnt(8)\n\t"
        "ds_read_b128 v[104:107], %[tile]\n\t"                 // quad 0 -> S1
        CST_ENC_FETCH_E0_(CST_ENC_S0)
        "s_waitcnt lgkmcnt(9)\n\t"
        CST_ENC_STEPS_E1
        // quad 1
        "s_waitcnt lgkmcnt(8)\n\t"
        CST_ENC_FETCH_E1_(CST_ENC_S1)
        "s_waitcnt lgkmcnt(8)\n\t"
        CST_ENC_STEPS_E0
        // quad 0
        "s_waitcnt lgkmcnt(4)\n\t"
        CST_ENC_STEPS_E1
        "s_waitcnt lgkmcnt(0)"
        : [lo] "+v"(lo), [hi] "+v"(hi), [wr] "+v"(wr), [smin] "+v"(smin), [smax] "+v"(smax), [sd] "=&s"(sd)
        : [tile] "v"(tile_row_addr), [tbl] "s"(table_addr_biased), [twoP] "v"(1u << P),
          [c3f00] "s"(0x3f00u), [shift] "v"(shift), [lanebase] "v"(ring_lane_addr)
        : "vcc", "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",
          "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125",
          "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139",
          "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153",
          "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163");
}

// All full tiles of a FULL wave in one asm statement (generated, with its wait counts, by scripts/gen_encode_loop.py):
// ans_encode_tile32's steps as ONE quad pipeline that runs on across tile boundaries (two LDS tile buffers), plus,
// per tile, the flush of a 64-byte word group from the ring (64-byte aligned slabs only), the LDS staging of the
// next tile's symbols and the request of the symbols three tiles further down (two register sets: under load an
// HBM round trip outlasts one tile).  Nothing in it is visible to the compiler's wait-count pass.
//   tile_row_addr[b] : the lane's own row in tile buffer b;  tile_tr_addr[b] : its transposed staging address
//   symbols_base : address of the last full tile of stream s0 (uniform);  goff[k] : byte offset of row (lane>>3)+8k,
//   chunk (lane&7) from it;  words_base + slab_off : the lane's slab (16-byte aligned);  cap : slab capacity (% 4 == 0)
__device__ __forceinline__ void ans_encode_tiles_loop(uint32_t& lo, uint32_t& hi, uint32_t& wr, uint32_t& flushed, int32_t& smin,
                                                      int32_t& smax, const uint32_t (&tile_row_addr)[2], const uint32_t (&tile_tr_addr)[2],
                                                      uint32_t ring_lane_addr, uint32_t cap, uint32_t slab_off,
                                                      uint32_t table_addr_biased, uint32_t P, const void* words_base,
                                                      uint64_t symbols_base, uint32_t n_tiles, const uint32_t (&goff)[8]) {
#include "cst_encode_loop.inc"
}

// The same main loop for symbols[t][stream] (CST_LAYOUT_SYMBOL_MAJOR; generated: cst_encode_loop_sm.inc).  Only the staging
// differs: symbols_base = address of (first symbol of the last full tile, stream s0);  goff[k] : byte offset of symbol row
// (lane >> 2) + 16 (k & 1), streams 16 (k >> 1) + 4 (lane & 3) .. + 3;  tile_tr_addr[b] : tile[4 (lane & 3)][lane >> 2] in
// buffer b;  tile_step_bytes = 32 * n_streams * 4.
__device__ __forceinline__ void ans_encode_tiles_loop_sm(uint32_t& lo, uint32_t& hi, uint32_t& wr, uint32_t& flushed, int32_t& smin,
                                                         int32_t& smax, const uint32_t (&tile_row_addr)[2], const uint32_t (&tile_tr_addr)[2],
                                                         uint32_t ring_lane_addr, uint32_t cap, uint32_t slab_off,
                                                         uint32_t table_addr_biased, uint32_t P, const void* words_base,
                                                         uint64_t symbols_base, uint32_t n_tiles, uint32_t tile_step_bytes,
                                                         const uint32_t (&goff)[8]) {
#include "cst_encode_loop_sm.inc"
}

// ------------------------------------------------------------------------------------------------
// Hand-scheduled decode of one 32-symbol tile for (W,S) = (32,64), 8 <= P <= 12, tables in LDS  
//
// A lone wave per SIMD (the C2 shape: 65536 streams = 1024 waves) issues one instruction of ANY kind every 4
// cycles, dependent VALU results forward without extra latency, and an LDS round trip is ~60 cycles.  A decode
// step is the serial chain   entry c|p -> N = (state >> P) * p + (q - c) -> refill? -> state' -> q' -> LDS lookup,
// so its floor is  (chain instructions) * 4 + one LDS latency.  The whole tile is ONE asm statement so that
//   * the chain is exactly 10 instructions from the arrival of an entry to the issue of the next lookup,
//   * everything else of the step (symbol fetch, ring read of the next candidate word, the shifted state halves
//     for the next step, the read-position update) is issued in the shadow of that lookup,
//   * waits are counted by hand: lgkmcnt(2|3) at the top of a step = "the entry is back, the younger symbol and
//     ring reads may still fly", lgkmcnt(0) in the slot between the refill compare and its first select,
//   * the refill predicate lives in vcc for exactly one step; carries go to a scratch SGPR pair.
// LDS image (stage_tile_tables):  cp[q] = c | p << 16 at lut+0,  sym[q] = decoded int32 symbol at lut+16384.
// Registers v120..v142 are scratch of the statement (64-bit operands of v_mad_u64_u32 are real register pairs).
//   v[120:121] N   v[122:123] [q-c, 0]   v124 p   v125,v126 (state >> P) halves   v127 lookup address
//   v128 entry     v129 candidate word   v131 ring address   v132 min(rd,1)   v133 q   v134..v141 symbols   v142 spare
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kTileLutBytes = 32768;    // cp[4096] + sym[4096]
constexpr int kTileAsmChunks = 3;            // 32 symbols of <= 12 bits: at most 12 words = 3 chunks per tile
constexpr uint32_t kTileDumpBytes = (kBlock / kWave) * 4 * kWave * 4;   // landing area of unused chunk slots
constexpr uint32_t kTileSymOffset = 16384;
constexpr uint32_t kTileKOffset = 32768;     // K[4096]: the refill thresholds of the main loops (stage_tile_tables<true>)
constexpr uint32_t kTileLutKBytes = 49152;

#define CST_DEC_STEP(TOPW, SYM, TAIL)                                                                               \
    "s_waitcnt lgkmcnt(" #TOPW ")\n\t"                                                                              \
    "v_sub_u32_sdwa v122, v133, v128 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"       \
    "v_lshrrev_b32 v124, 16, v128\n\t"                                                                              \
    "v_mad_u64_u32 v[120:121], %[sd], v125, v124, v[122:123]\n\t"                                                   \
    "v_mad_u32_u24 v121, v126, v124, v121\n\t"                                                                      \
    "v_cmp_lt_u32 vcc, v121, v132\n\t"                                                                              \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                      \
    "v_cndmask_b32 %[lo], v120, v129, vcc\n\t"                                                                      \
    "v_and_b32 v133, %[mask], %[lo]\n\t"                                                                            \
    "v_lshl_add_u32 v127, v133, 2, %[lut]\n\t"                                                                      \
    "ds_read_b32 v128, v127\n\t"                                                                                    \
    "v_subbrev_co_u32 %[rd], %[sd], 0, %[rd], vcc\n\t"                                                              \
    "v_add_lshl_u32 v131, %[rd], %[shm1], 8\n\t"                                                                    \
    "v_and_or_b32 v131, v131, %[c3f00], %[lanebase]\n\t"                                                            \
    "ds_read_b32 v129, v131\n\t"                                                                                    \
    "ds_read_b32 " SYM ", v127 offset:16384\n\t"                                                                    \
    "v_cndmask_b32 %[hi], v121, v120, vcc\n\t"                                                                      \
    "v_min_u32 v132, 1, %[rd]\n\t"                                                                                  \
    "v_alignbit_b32 v125, %[hi], %[lo], %[P]\n\t"                                                                   \
    "v_lshrrev_b32 v126, %[P], %[hi]\n\t" TAIL

// steps 4k .. 4k+3: symbols 4k+1 .. 4k+4 are fetched (the first three complete quad Q0, the fourth opens quad Q1);
// the finished quad leaves for the tile row (one 16-B LDS write) at the end of its last step
#define CST_DEC_QUAD(TOPW0, A1, A2, A3, B0, WRITE) CST_DEC_QUAD_(TOPW0, A1, A2, A3, B0, WRITE)
#define CST_DEC_QUAD_(TOPW0, A1, A2, A3, B0, WRITE)                                                                 \
    CST_DEC_STEP(TOPW0, A1, "") CST_DEC_STEP(2, A2, "") CST_DEC_STEP(2, A3, "") CST_DEC_STEP(2, B0, WRITE)

// Decodes symbols [0, 32) of the current tile into the lane's tile row (LDS).  On return every LDS operation of
// the statement has completed.
__device__ __forceinline__ void ans_decode_tile32(uint32_t& lo, uint32_t& hi, uint32_t& rd, uint32_t lut_addr, uint32_t mask,
                                                  uint32_t P, uint32_t tile_row_addr, uint32_t shift_minus_1,
                                                  uint32_t ring_lane_addr, uint32_t ring_mask) {
    uint64_t sd;
    asm volatile(
        // prologue: first lookup, candidate word, shifted state
        "v_mov_b32 v123, 0\n\t"
        "v_and_b32 v133, %[mask], %[lo]\n\t"
        "v_lshl_add_u32 v127, v133, 2, %[lut]\n\t"
        "ds_read_b32 v128, v127\n\t"
        "ds_read_b32 v134, v127 offset:16384\n\t"
        "v_add_lshl_u32 v131, %[rd], %[shm1], 8\n\t"
        "v_and_or_b32 v131, v131, %[c3f00], %[lanebase]\n\t"
        "ds_read_b32 v129, v131\n\t"
        "v_min_u32 v132, 1, %[rd]\n\t"
        "v_alignbit_b32 v125, %[hi], %[lo], %[P]\n\t"
        "v_lshrrev_b32 v126, %[P], %[hi]\n\t"
#define CST_DEC_WR(REGS, OFF) "ds_write_b128 %[tile], " REGS " offset:" #OFF "\n\t"
#define CST_DEC_W1 3
        CST_DEC_QUAD(2, "v135", "v136", "v137", "v138", CST_DEC_WR("v[134:137]", 0))
        CST_DEC_QUAD(CST_DEC_W1, "v139", "v140", "v141", "v134", CST_DEC_WR("v[138:141]", 16))
        CST_DEC_QUAD(CST_DEC_W1, "v135", "v136", "v137", "v138", CST_DEC_WR("v[134:137]", 32))
        CST_DEC_QUAD(CST_DEC_W1, "v139", "v140", "v141", "v134", CST_DEC_WR("v[138:141]", 48))
        CST_DEC_QUAD(CST_DEC_W1, "v135", "v136", "v137", "v138", CST_DEC_WR("v[134:137]", 64))
        CST_DEC_QUAD(CST_DEC_W1, "v139", "v140", "v141", "v134", CST_DEC_WR("v[138:141]", 80))
        CST_DEC_QUAD(CST_DEC_W1, "v135", "v136", "v137", "v138", CST_DEC_WR("v[134:137]", 96))
        CST_DEC_QUAD(CST_DEC_W1, "v139", "v140", "v141", "v142", CST_DEC_WR("v[138:141]", 112))
        "s_waitcnt lgkmcnt(0)"
        : [lo] "+v"(lo), [hi] "+v"(hi), [rd] "+v"(rd), [sd] "=&s"(sd)
        : [lut] "s"(lut_addr), [mask] "s"(mask), [P] "s"(P), [c3f00] "s"(ring_mask), [tile] "v"(tile_row_addr),
          [shm1] "v"(shift_minus_1), [lanebase] "v"(ring_lane_addr)
        : "vcc", "memory", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131",
          "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142");
}

// Main loop of the same decoder: tiles 1 .. n_tiles of a FULL wave in one asm statement (generated, with its wait
// counts, by scripts/gen_decode_loop.py).  Per iteration it
//   * requests up to kDecChunks 16-B chunks of compressed words (exec-masked global loads relative to `words_base`)
//     at the top and lands them in the ring at the bottom (vmcnt(8): the eight stores of the iteration are younger),
//   * decodes 32 symbols exactly like ans_decode_tile32 into the lane's row of the CURRENT tile buffer,
//   * streams the PREVIOUS tile buffer to HBM in the idle issue slots of the steps: quad k reads rows
//     (lane >> 3) + 8k, chunk (lane & 7) (one 16-B LDS read) and stores it at store_base + goff[k] (128-B row
//     segments, eight rows per instruction), then swaps the buffers and advances store_base by 128 B.
// The decode chain leaves ~6 idle issue slots per symbol (it waits on the LDS lookup), so this work is free.
// The compiler sees no vector-memory instruction in the loop, hence no conservative vmcnt(0) of its own.
constexpr int kDecRingSlots = 32;            // ring words per lane for this decoder (8 KiB per wave, 8-KiB aligned)
constexpr int kDecAhead = 24;                // two tiles of at most 12 words each
constexpr uint32_t kDecRingMask = (kDecRingSlots - 1) * kWave * 4;

// PLAIN_STORES: the tile stores without the non-temporal hint (CST_STORE_MOD) -- for rows that are not cache-line aligned
// (scripts/gen_decode_loop.py)
template <bool PLAIN_STORES = false>
__device__ __forceinline__ void ans_decode_tiles_loop(uint32_t& lo, uint32_t& hi, uint32_t& rd, uint32_t& lo_issued,
                                                      uint32_t& row_cur, uint32_t& row_prev, uint32_t& tr_cur, uint32_t& tr_prev,
                                                      uint32_t lut_addr, uint32_t mask, uint32_t P, uint32_t ring_mask,
                                                      const void* words_base, uint64_t store_base, uint32_t n_tiles,
                                                      uint32_t shift_minus_1, uint32_t ring_lane_addr, uint32_t dump_addr,
                                                      uint32_t words_off, const uint32_t (&goff)[8]) {
    if constexpr (PLAIN_STORES) {
#define CST_STORE_MOD ""
#include "cst_decode_loop.inc"
#undef CST_STORE_MOD
    } else {
#define CST_STORE_MOD "nt"
#include "cst_decode_loop.inc"
#undef CST_STORE_MOD
    }
}

// The same main loop for symbols[t][stream] (generated: cst_decode_loop_sm.inc): quad k of the previous tile leaves as the
// 16 bytes of streams 16 (k >> 1) + 4 (lane & 3) .. + 3 of symbol row (lane >> 2) + 16 (k & 1); goff[k] is that position
// relative to store_base (tile 0, stream s0), tr_cur / tr_prev = tile[4 (lane & 3)][lane >> 2] of the two buffers,
// tile_step_bytes = 32 * n_streams * 4.
template <bool PLAIN_STORES = false>
__device__ __forceinline__ void ans_decode_tiles_loop_sm(uint32_t& lo, uint32_t& hi, uint32_t& rd, uint32_t& lo_issued,
                                                         uint32_t& row_cur, uint32_t& row_prev, uint32_t& tr_cur, uint32_t& tr_prev,
                                                         uint32_t lut_addr, uint32_t mask, uint32_t P, uint32_t ring_mask,
                                                         const void* words_base, uint64_t store_base, uint32_t n_tiles,
                                                         uint32_t shift_minus_1, uint32_t ring_lane_addr, uint32_t dump_addr,
                                                         uint32_t words_off, uint32_t tile_step_bytes, const uint32_t (&goff)[8]) {
    if constexpr (PLAIN_STORES) {
#define CST_STORE_MOD ""
#include "cst_decode_loop_sm.inc"
#undef CST_STORE_MOD
    } else {
#define CST_STORE_MOD "nt"
#include "cst_decode_loop_sm.inc"
#undef CST_STORE_MOD
    }
}

} // namespace cst
