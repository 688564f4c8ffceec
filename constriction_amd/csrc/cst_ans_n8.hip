// cst_ans_n8.hip -- the shared-table ANS coder (W,S) = (32,64), 8 <= P <= 12, on INT8 symbol matrices (round 5).
//
// The reference's coders are generic over the symbol type (`Symbol: PrimInt + AsPrimitive<Probability> + ...`,
// src/stream/model/quantize.rs:229-255): the 101-symbol alphabet of config C2 travels as i8 there if the caller says so.  Until this
// file a narrow matrix was widened / narrowed by a streaming kernel next to the coder call (cst_symbols.hip: 5 more bytes of HBM
// traffic per symbol, 0.27 + 0.39 ms at 65 536 x 4096 on top of 0.25 + 0.25 ms of coding).  Here the int8 matrix is what the
// hand-scheduled loops read and write:
//   decode (ans_decode_n8_kernel, cst_decode_loop_n8.inc): a decoded quad is packed into one dword and written into the lane's row
//     of a BYTE tile (rows of 128 symbols: one cache line of the matrix); a loop pass is four 32-symbol tiles, and the previous
//     group of 128 symbols leaves in its shadow, eight whole 128-byte lines per store instruction, two stores per tile.
// Same recurrence (stack.rs:1070-1100), same read positions, states and status as ans_decode_kernel; the symbols are its symbols
// as int8 (the callers check that the model's support fits).  Shapes these kernels do not take (rows that are not whole aligned
// 128-symbol groups, partial waves, symbol-major batches, int16) stay on the conversion path.
#include <cstdlib>

#include "cst_ans_kernels.hpp"

namespace cst {

constexpr int kN8GroupSyms = 128;                                          // symbols per row of a byte tile = one line of the matrix
constexpr int kN8RowBytes = 132;                                           // + one word of padding: conflict-free b32 writes and reads
constexpr size_t kN8Waves = kBlock / kWave;
constexpr size_t kN8RingWaveBytes = (size_t)kDecRingSlots * kWave * 4;     // 8 KiB, 8-KiB aligned
constexpr size_t kN8TileBytes = (size_t)kWave * kN8RowBytes;               // 8448 B per buffer and wave
constexpr size_t kN8LutOff = kN8Waves * kN8RingWaveBytes;
constexpr size_t kN8TileOff = kN8LutOff + kTileLutBytes;
constexpr size_t kN8DumpOff = kN8TileOff + 2 * kN8Waves * kN8TileBytes;
constexpr size_t kN8LdsBytes = kN8DumpOff + kTileDumpBytes;
static_assert(kN8LdsBytes <= 160 * 1024, "LDS budget");
static_assert(kN8TileBytes % 4 == 0 && kN8TileOff % 16 == 0, "tile alignment");

template <int BYTES>
__device__ __forceinline__ void ans_decode_n8_loop(uint32_t& lo, uint32_t& hi, uint32_t& rd, uint32_t& lo_issued, uint32_t& row_cur,
                                                   uint32_t& row_prev, uint32_t& tr_cur, uint32_t& tr_prev, uint32_t lut_addr, uint32_t mask,
                                                   uint32_t P, uint32_t ring_mask, const void* words_base, uint64_t store_base, uint32_t n_groups,
                                                   uint32_t shift_minus_1, uint32_t ring_lane_addr, uint32_t dump_addr, uint32_t words_off,
                                                   const uint32_t (&goff)[8]) {
    if constexpr (BYTES == 1) {
#include "cst_decode_loop_n8.inc"
    } else {
#include "cst_decode_loop_n16.inc"
    }
}

// LDS: [word rings: 4 x 8 KiB][cp + sym tables 32 KiB][two byte tiles per wave (all A, then all B)][dump rows]
// BYTES = 1: int8 matrices; BYTES = 2: int16 (a 128-byte line is 64 symbols: two tiles per pass of the statement)
template <int BYTES>
__global__ __launch_bounds__(kBlock) void ans_decode_n8_kernel(const AnsDecodeArgs a) {
    constexpr int kGroupSyms = kN8GroupSyms / BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = threadIdx.x >> 6;
    const int P = a.precision;
    DecLut lut{};
    stage_tile_tables(smem + kN8LutOff, P, a.dec_cp, a.dec_idx, a.min_symbol, lut);
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem + wave_in_block * kN8RingWaveBytes);
    unsigned char* tile_a = smem + kN8TileOff + wave_in_block * kN8TileBytes;
    unsigned char* tile_b = tile_a + kN8Waves * kN8TileBytes;
    uint32_t* dump = reinterpret_cast<uint32_t*>(smem + kN8DumpOff) + wave_in_block * (4 * kWave) + lane;
    __syncthreads();

    const size_t s0 = ((size_t)blockIdx.x * kBlock + (size_t)wave_in_block * kWave);
    if (s0 >= a.n_streams) return;
    // a partial wave: the lanes behind the last stream decode the LAST stream again -- everything a lane does is a function of its
    // stream alone, so they read what its lane reads and write the same bytes to the same places (symbols, status, state)
    const uint32_t last_row = (uint32_t)min((size_t)(kWave - 1), a.n_streams - 1 - s0);
    const size_t s = s0 + min((uint32_t)lane, last_row);
    const size_t N = a.n_per_stream;
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;
    int8_t* out = reinterpret_cast<int8_t*>(a.symbols);   // (the narrow matrix travels in the int32 field of the argument block; BYTE addresses below)
    const size_t row_bytes = N * BYTES;

    DecLane<32, 64, kDecRingSlots, kDecAhead> L;
    const WordSlice ws = word_slice(a.offsets, a.stride_words, a.n_words, s, a.words_capacity);
    L.init(a.words + ws.off, ws.n, ring, lane);
    if (raw) L.state = a.state[s];
    else L.read_initial_state();
    L.in.prime();
    wave_lds_fence();

    if ((lds_addr(ring) & (uint32_t)(kN8RingWaveBytes - 1)) != 0) __builtin_trap();      // (ring addresses are formed with v_and_or)
    uint32_t lo = (uint32_t)L.state, hi = (uint32_t)(L.state >> 32);
    const uint32_t qmask = (1u << P) - 1u;
    const uint32_t lut_addr = lds_addr(lut.cp), lane_addr = lds_addr(ring + lane);
    __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0): nothing of the prologue in flight when the statement keeps its own book
    const uint32_t n_g = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(N / kGroupSyms));

    const unsigned char* words_base = reinterpret_cast<const unsigned char*>(reinterpret_cast<uintptr_t>(a.words) & ~(uintptr_t)15);
    const uint32_t w_off = (uint32_t)(reinterpret_cast<const unsigned char*>(L.in.base16) - words_base);
    uint32_t goff[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) goff[k] = (uint32_t)((size_t)min((uint32_t)((lane >> 3) + 8 * k), last_row) * row_bytes + 16 * (size_t)(lane & 7));
    const uint32_t tr_off = (uint32_t)((lane >> 3) * kN8RowBytes + 16 * (lane & 7));
    uint32_t row_cur = lds_addr(tile_a) + (uint32_t)(lane * kN8RowBytes), row_prev = lds_addr(tile_b) + (uint32_t)(lane * kN8RowBytes);
    uint32_t tr_cur = lds_addr(tile_a) + tr_off, tr_prev = lds_addr(tile_b) + tr_off;
    const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(out + s0 * row_bytes);
    const uint64_t store_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
    ans_decode_n8_loop<BYTES>(lo, hi, L.in.rd, L.in.lo_issued, row_cur, row_prev, tr_cur, tr_prev, lut_addr, qmask, (uint32_t)P, kDecRingMask, words_base,
                       store_base, n_g, L.in.shift - 1u, lane_addr, lds_addr(dump), w_off, goff);
    // the last group is still in LDS: the statement swapped the buffers behind it, so it is the "previous" one
    wave_lds_fence();
    {
        const unsigned char* last = ((n_g - 1) & 1) ? tile_b : tile_a;
        int8_t* dst = out + s0 * row_bytes + (size_t)(n_g - 1) * kN8GroupSyms;      // (a group is 128 BYTES of every row)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(last + ((lane >> 3) + 8 * k) * kN8RowBytes + 16 * (lane & 7));
            v4i v;
            v.x = (int)src[0]; v.y = (int)src[1]; v.z = (int)src[2]; v.w = (int)src[3];
            __builtin_nontemporal_store(v, reinterpret_cast<v4i*>(dst + goff[k]));
        }
    }

    a.status[s] = ws.bad ? (int32_t)CST_STREAM_INVALID_DATA : L.status;
    if (raw) {
        a.state[s] = ((uint64_t)hi << 32) | lo;
        if (a.n_words_out) a.n_words_out[s] = L.in.rd;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// More than one wave per SIMD (more than 256 streams per CU: config C5's shard, or the k x 65 536 virtual streams of a batch decoded
// through jump points): the small-footprint decoder of cst_ans_small.hip over byte tiles.  ONE packed table  c | p << 12 | index << 24
// (16 KiB, at most 256 symbols), the 32-slot ring and ONE byte tile per wave (16.25 KiB): eight waves per CU in 147 KiB.  The
// decoded symbol leaves its step as a byte of the quad's register (the SDWA add that forms min_symbol + index writes byte `pos`
// and preserves the others), the group of 128 symbols leaves at the end of its fourth tile (the sibling wave covers the stall).
// With a quarter of the int32 decoder's output bytes the second wave is no longer spent on memory time: 131 072 x 4096 decode in
// about 0.3 ms against 0.48 ms for ans_decode_n8_kernel and 0.60 - 0.68 ms for the int32 kernels.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kN8SmThreads = 512;
constexpr size_t kN8SmWaves = kN8SmThreads / kWave;
constexpr size_t kN8SmLutBytes = (size_t)4 << 12;
constexpr size_t kN8SmLutOff = kN8SmWaves * kN8RingWaveBytes;
constexpr size_t kN8SmTileOff = kN8SmLutOff + kN8SmLutBytes;
constexpr size_t kN8SmDumpOff = kN8SmTileOff + kN8SmWaves * kN8TileBytes;
constexpr size_t kN8SmLdsBytes = kN8SmDumpOff + 4 * kWave * 4;
static_assert(kN8SmLdsBytes <= 160 * 1024, "LDS budget");

template <int BYTES>
__device__ __forceinline__ void decode_groups_loop_small_n8(uint32_t& lo, uint32_t& hi, uint32_t& rd, uint32_t& lo_issued, uint32_t lut_addr,
                                                            uint32_t mask, uint32_t ring_mask, uint32_t P, const void* words_base,
                                                            uint64_t store_base, uint32_t n_tiles, int32_t min_symbol, uint32_t shift_minus_1,
                                                            uint32_t ring_lane_addr, uint32_t dump_addr, uint32_t words_off,
                                                            uint32_t tile_row_addr, uint32_t tile_tr_addr, const uint32_t (&goff)[8]) {
#define CST_STORE_MOD "nt"
    if constexpr (BYTES == 1) {
#include "cst_decode_loop_small_n8.inc"
    } else {
#include "cst_decode_loop_small_n16.inc"
    }
#undef CST_STORE_MOD
}

// LDS: [word rings: 8 x 8 KiB][packed table 16 KiB][one byte tile per wave][one landing area for unused chunk slots]
template <int BYTES>
__global__ __launch_bounds__(kN8SmThreads) void ans_decode_small_n8_kernel(const AnsDecodeArgs a) {
    constexpr int kGroupSyms = kN8GroupSyms / BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = threadIdx.x >> 6;
    const int P = a.precision;
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem + wave_in_block * kN8RingWaveBytes);
    uint32_t* lut = reinterpret_cast<uint32_t*>(smem + kN8SmLutOff);
    unsigned char* tile = smem + kN8SmTileOff + wave_in_block * kN8TileBytes;
    uint32_t* dump = reinterpret_cast<uint32_t*>(smem + kN8SmDumpOff) + lane;
    if ((lds_addr(ring) & (uint32_t)(kN8RingWaveBytes - 1)) != 0) __builtin_trap();
    // quantile -> c | p << 12 | index << 24   (lookup_contiguous.rs:564-605 as ONE table read; c, p < 2^12, index < 2^8)
    for (int q = threadIdx.x; q < (1 << P); q += blockDim.x) {
        const uint32_t cp = a.dec_cp[q];
        lut[q] = (cp & 0xfffu) | ((cp >> 16) << 12) | ((uint32_t)a.dec_idx[q] << 24);
    }
    __syncthreads();

    const size_t s0 = (size_t)blockIdx.x * kN8SmThreads + (size_t)wave_in_block * kWave;
    if (s0 >= a.n_streams) return;
    const uint32_t last_row = (uint32_t)min((size_t)(kWave - 1), a.n_streams - 1 - s0);      // (a partial wave: see ans_decode_n8_kernel)
    const size_t s = s0 + min((uint32_t)lane, last_row);
    const size_t N = a.n_per_stream;
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;
    int8_t* out = reinterpret_cast<int8_t*>(a.symbols);
    const size_t row_bytes = N * BYTES;

    DecLane<32, 64, kDecRingSlots, kDecAhead> L;
    const WordSlice ws = word_slice(a.offsets, a.stride_words, a.n_words, s, a.words_capacity);
    L.init(a.words + ws.off, ws.n, ring, lane);
    if (raw) L.state = a.state[s];
    else L.read_initial_state();
    L.in.prime();
    wave_lds_fence();
    uint32_t lo = (uint32_t)L.state, hi = (uint32_t)(L.state >> 32);

    const unsigned char* words_base = reinterpret_cast<const unsigned char*>(reinterpret_cast<uintptr_t>(a.words) & ~(uintptr_t)15);
    const uint32_t w_off = (uint32_t)(reinterpret_cast<const unsigned char*>(L.in.base16) - words_base);
    uint32_t goff[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) goff[k] = (uint32_t)((size_t)min((uint32_t)((lane >> 3) + 8 * k), last_row) * row_bytes + 16 * (size_t)(lane & 7));
    const uint32_t tr_off = (uint32_t)((lane >> 3) * kN8RowBytes + 16 * (lane & 7));
    const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(out + s0 * row_bytes);
    const uint64_t store_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
    __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0): the statement keeps its own book from here
    decode_groups_loop_small_n8<BYTES>(lo, hi, L.in.rd, L.in.lo_issued, lds_addr(lut), (1u << P) - 1u, kDecRingMask, (uint32_t)P, words_base, store_base,
                                (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(N / kGroupSyms)), a.min_symbol, L.in.shift - 1u,
                                lds_addr(ring + lane), lds_addr(dump), w_off, lds_addr(tile) + (uint32_t)(lane * kN8RowBytes), lds_addr(tile) + tr_off,
                                goff);
    a.status[s] = ws.bad ? (int32_t)CST_STREAM_INVALID_DATA : L.status;
    if (raw) {
        a.state[s] = ((uint64_t)hi << 32) | lo;
        if (a.n_words_out) a.n_words_out[s] = L.in.rd;
    }
}

// Rows that are whole 128-byte aligned groups (at least one), every stream's words within 2 GiB of the buffer.
bool n8_decode_usable(const AnsDecodeArgs& a, cst_coder_config cfg, cst_layout layout, int symbol_bytes) {
    if (knobs().no_n8) return false;                     // (A/B runs: the conversion path)
    if (cfg.word_bits != 32 || cfg.state_bits != 64 || layout != CST_LAYOUT_STREAM_MAJOR || a.precision < 8 || a.precision > 12) return false;
    if (!a.dec_cp || !a.dec_idx) return false;
    if (a.n_streams == 0) return false;                  // (partial waves are taken: their spare lanes repeat the last stream)
    if (symbol_bytes != 1 && symbol_bytes != 2) return false;
    if (a.n_per_stream % (size_t)(kN8GroupSyms / symbol_bytes) != 0 || a.n_per_stream == 0 || a.n_per_stream >= (1u << 24)) return false;
    if ((reinterpret_cast<uintptr_t>(a.symbols) & 127) != 0) return false;
    if (a.offsets && a.words_capacity == 0) return false;                                          // the lanes' 32-bit word offsets need a known span
    const uint64_t span = a.offsets ? a.words_capacity : (uint64_t)a.n_streams * a.stride_words;
    return span * 4 + 256 < 0x80000000ull;
}

// more streams than one wave per SIMD of this device (and a table the packed form holds): the small-footprint form
bool n8_decode_small(const AnsDecodeArgs& a, int device_cus) {
    if (!knobs().small_decoders) return false;           // (A/B runs, as for cst_ans_small.hip)
    return a.n_symbols <= 256 && a.n_streams > (size_t)device_cus * kBlock;
}

cst_status ans_decode_small_n8(const AnsDecodeArgs& a, int symbol_bytes, hipStream_t hs) {
    const size_t blocks = (a.n_streams + kN8SmThreads - 1) / kN8SmThreads;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    auto kernel = symbol_bytes == 1 ? ans_decode_small_n8_kernel<1> : ans_decode_small_n8_kernel<2>;
    CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kN8SmLdsBytes));
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kN8SmThreads), kN8SmLdsBytes, hs, a);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

cst_status ans_decode_n8(const AnsDecodeArgs& a, int symbol_bytes, hipStream_t hs) {
    const size_t blocks = (a.n_streams + kBlock - 1) / kBlock;
    auto kernel = symbol_bytes == 1 ? ans_decode_n8_kernel<1> : ans_decode_n8_kernel<2>;
    CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kN8LdsBytes));
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kBlock), kN8LdsBytes, hs, a);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

} // namespace cst
