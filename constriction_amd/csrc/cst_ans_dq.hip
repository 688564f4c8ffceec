// cst_ans_dq.hip -- the shared-table ANS decoder (W,S) = (32,64), 8 <= P <= 12, with LANE-QUAD word loads (round 4).
//
// ans_decode_kernel's main loop (cst_decode_loop.inc) lets every lane request up to three 16-byte chunks of its own stream
// per tile: three instructions of up to 64 requests to 64 different cache lines.  How long the CU's memory pipeline needs for
// them depends on the slab stride (0.254 ms at 128 x 64 bytes, 0.366 ms at 103 x 64), and the tile stores of the same wave
// queue behind them at issue (scripts/gen_decode_loop_dq.py has the measurements).  Here a stream asks for a whole 64-byte
// group when its window needs one, and four neighbouring lanes move it -- at most 16 whole 64-byte segments per instruction:
//     lane 4 j + i, pass p = 0 .. 3:  chunk i of the group of stream 16 p + j  ->  slots of THAT stream's ring column.
// Groups are 64-byte aligned in memory (RingReader64: the ring's base is the stream's first word rounded down to 64 bytes), the
// ring has 64 slots per lane (window of 24 words + a group of 16 + the prologue's rounding to whole groups), and the LDS for
// its second half is the second tile buffer's: the finished tile waits in 32 registers for its stores instead.
// Same recurrence (stack.rs:1070-1100), same symbols, read positions, states and status as ans_decode_kernel; shapes this
// kernel does not take (partial waves, rows that are not whole cache-line aligned tiles, symbol-major batches) stay there.
#include <cstdlib>

#include "cst_ans_kernels.hpp"

namespace cst {

constexpr int kDqRingSlots = 64;
constexpr uint32_t kDqRingMask = (kDqRingSlots - 1) * kWave * 4;
constexpr size_t kDqRingWaveBytes = (size_t)kDqRingSlots * kWave * 4;     // 16 KiB, 16-KiB aligned
constexpr size_t kDqWaves = kBlock / kWave;
constexpr size_t kDqTileBytes = (size_t)kWave * kTileStride * 4;
constexpr size_t kDqLutOff = kDqWaves * kDqRingWaveBytes;
constexpr size_t kDqTileOff = kDqLutOff + kTileLutBytes;
constexpr size_t kDqDumpOff = kDqTileOff + kDqWaves * kDqTileBytes;
constexpr size_t kDqLdsBytes = kDqDumpOff + kTileDumpBytes;
static_assert(kDqLdsBytes <= 160 * 1024, "LDS budget");

// The window of RingReader over a base that is 64-byte aligned (shift = 0 .. 15 words): a group of 16 window positions is
// one 64-byte segment of memory.
struct RingReader64 : RingReader<kDqRingSlots, kDecAhead> {
    __device__ __forceinline__ void init64(const uint32_t* in, uint32_t len, uint32_t* wave_ring, int lane_) {
        init(in, len, wave_ring, lane_);
        shift = (uint32_t)((reinterpret_cast<uintptr_t>(in) & 63) >> 2);
        base16 = in - shift;
    }
    // blocking: whole groups from here on (lo_issued a multiple of 16; at most three more chunks)
    __device__ __forceinline__ void align_to_groups() {
        while (lo_issued & 15u) {
            lo_issued -= 4;
            const uint4 v = *reinterpret_cast<const uint4*>(base16 + lo_issued);
            uint32_t* b = slot(lo_issued);
            b[0] = v.x; b[kWave] = v.y; b[2 * kWave] = v.z; b[3 * kWave] = v.w;
        }
    }
};

__device__ __forceinline__ void ans_decode_dq_loop(uint32_t& lo, uint32_t& hi, uint32_t& rd, uint32_t& lo_issued, uint32_t row_addr, uint32_t tr_addr,
                                                   uint32_t lut_addr, uint32_t mask, uint32_t P, uint32_t ring_mask, const void* words_base,
                                                   uint64_t store_base, uint32_t n_tiles, uint32_t shift_minus_1, uint32_t ring_lane_addr,
                                                   uint32_t dump_addr, uint32_t words_off, const uint32_t (&goff)[8]) {
#define CST_STORE_MOD "nt"
#include "cst_decode_loop_dq.inc"
#undef CST_STORE_MOD
}

// LDS: [word rings: 4 x 16 KiB][cp + sym tables 32 KiB][one symbol tile per wave][dump rows]
__global__ __launch_bounds__(kBlock) void ans_decode_dq_kernel(const AnsDecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = threadIdx.x >> 6;
    const int P = a.precision;
    DecLut lut{};
    stage_tile_tables(smem + kDqLutOff, P, a.dec_cp, a.dec_idx, a.min_symbol, lut);
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem + wave_in_block * kDqRingWaveBytes);
    int32_t* tile = reinterpret_cast<int32_t*>(smem + kDqTileOff + wave_in_block * kDqTileBytes);
    uint32_t* dump = reinterpret_cast<uint32_t*>(smem + kDqDumpOff) + wave_in_block * (4 * kWave) + lane;
    __syncthreads();

    const size_t s0 = ((size_t)blockIdx.x * kBlock + (size_t)wave_in_block * kWave);
    if (s0 >= a.n_streams) return;                       // (the launcher only takes whole waves)
    const size_t s = s0 + lane;
    const size_t N = a.n_per_stream;
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;

    const WordSlice ws = word_slice(a.offsets, a.stride_words, a.n_words, s, a.words_capacity);
    RingReader64 in;
    in.init64(a.words + ws.off, ws.n, ring, lane);
    int32_t status = CST_STREAM_OK;
    uint64_t state = 0;
    if (raw) state = a.state[s];
    else if (in.rd > 0) {                               // from_compressed + read_initial_state (stack.rs:299-318, 440-462)
        const uint32_t first = in.word_direct(--in.rd);
        if (first == 0) { status = CST_STREAM_INVALID_DATA; in.rd = 0; }
        else {
            state = first;
            while (in.rd > 0) {
                state = (state << 32) | (uint64_t)in.word_direct(--in.rd);
                if (state >= (1ull << 32)) break;
            }
        }
    }
    in.prime();
    wave_lds_fence();

    if ((lds_addr(ring) & (uint32_t)(kDqRingWaveBytes - 1)) != 0) __builtin_trap();      // (ring addresses are formed with v_and_or)
    uint32_t lo = (uint32_t)state, hi = (uint32_t)(state >> 32);
    const uint32_t qmask = (1u << P) - 1u;
    const uint32_t lut_addr = lds_addr(lut.cp), lane_addr = lds_addr(ring + lane);
    int32_t* my = tile + lane * kTileStride;
    __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0): nothing of the prologue in flight when the statements keep their own book
    const uint32_t n_t = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(N / kTileSyms));
    ans_decode_tile32(lo, hi, in.rd, lut_addr, qmask, (uint32_t)P, lds_addr(my), in.shift - 1u, lane_addr, kDqRingMask);
    in.refill_blocking();
    in.align_to_groups();
    wave_lds_fence();
    __builtin_amdgcn_s_waitcnt(0x0F70);

    const unsigned char* words_base = reinterpret_cast<const unsigned char*>(reinterpret_cast<uintptr_t>(a.words) & ~(uintptr_t)63);
    const uint32_t w_off = (uint32_t)(reinterpret_cast<const unsigned char*>(in.base16) - words_base);
    uint32_t goff[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) goff[k] = (uint32_t)((((size_t)(lane >> 3) + 8 * k) * N + 4 * (size_t)(lane & 7)) * 4);
    const uint32_t tr_addr = lds_addr(tile) + (uint32_t)(((lane >> 3) * kTileStride + 4 * (lane & 7)) * 4);
    const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(a.symbols + s0 * N);
    const uint64_t store_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
    ans_decode_dq_loop(lo, hi, in.rd, in.lo_issued, lds_addr(my), tr_addr, lut_addr, qmask, (uint32_t)P, kDqRingMask, words_base, store_base,
                       n_t - 1u, in.shift - 1u, lane_addr, lds_addr(dump), w_off, goff);
    // the last tile is still in LDS
    wave_lds_fence();
    tile_store_skewed(a.symbols, N, s0, (size_t)(n_t - 1) * kTileSyms, lane, tile);

    a.status[s] = ws.bad ? (int32_t)CST_STREAM_INVALID_DATA : status;
    if (raw) {
        a.state[s] = ((uint64_t)hi << 32) | lo;
        if (a.n_words_out) a.n_words_out[s] = in.rd;
    }
}

// Whole waves, rows that are whole 128-byte aligned tiles (at least two), every stream's words within 2 GiB of the buffer.
bool dq_decode_usable(const AnsDecodeArgs& a, cst_coder_config cfg, cst_layout layout) {
    // Taken on request: CST_FLAG_COLD_WORDS of the call, or CST_DQ_DECODER=1 (A/B runs).  Measured on one MI355X, 65 536 x 4096 at
    // P = 12, slab strides of 97 / 103 / 130 x 64 bytes (gpurun_out/r04_dq_cold.txt): words from HBM (a 1-GiB fill before the
    // launch) 0.349 / 0.343 / 0.326 ms against ans_decode_kernel's 0.405 / 0.411 / 0.367; cache-resident words 0.287 / 0.337 /
    // 0.274 against 0.296 / 0.368 / 0.255 -- whole 64-byte segments are what single reads among 4 TB/s of writes should look
    // like (scripts/microbench/rw_mix.hip shows the same without any decoder), the tile's read-back burst and the position
    // exchange are what it costs where the words sit in the caches at a well-chosen stride.
    if (!(a.flags & CST_FLAG_COLD_WORDS) && !knobs().dq_decoder) return false;
    if (cfg.word_bits != 32 || cfg.state_bits != 64 || layout != CST_LAYOUT_STREAM_MAJOR || a.precision < 8 || a.precision > 12) return false;
    if (!a.dec_cp || !a.dec_idx) return false;
    if (a.n_streams == 0 || a.n_streams % kWave != 0) return false;
    if (a.n_per_stream % kTileSyms != 0 || a.n_per_stream < 2 * kTileSyms || a.n_per_stream >= (1u << 24)) return false;
    if ((reinterpret_cast<uintptr_t>(a.symbols) & 127) != 0) return false;
    if (64 * a.n_per_stream * 4 >= 0x100000000ull) return false;                                   // 32-bit row offsets within a wave
    // the lanes address their words as 32-bit byte offsets from a.words: the span must be KNOWN (packed offsets with
    // words_capacity = 0, "the caller vouches", could point anywhere: those batches stay on the kernel with 64-bit offsets)
    if (a.offsets && a.words_capacity == 0) return false;
    const uint64_t span = a.offsets ? a.words_capacity : (uint64_t)a.n_streams * a.stride_words;   // words the lanes address relative to a.words
    return span * 4 + 256 < 0x80000000ull;
}

cst_status ans_decode_dq(const AnsDecodeArgs& a, hipStream_t hs) {
    const size_t blocks = (a.n_streams + kBlock - 1) / kBlock;
    CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(ans_decode_dq_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDqLdsBytes));
    hipLaunchKernelGGL(ans_decode_dq_kernel, dim3((unsigned)blocks), dim3(kBlock), kDqLdsBytes, hs, a);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

} // namespace cst
