// cst_ans_small.hip -- the shared-table ANS coder (W,S) = (32,64), 8 <= P <= 12, with a SMALL LDS footprint: the kernels
// for batches of more than one wave per SIMD (more than 256 streams per CU: BASELINE config C5's 131 072 streams per GPU).
//
// cst_ans_kernels.hpp's hand-scheduled kernels give every wave two symbol tiles and a 64-slot word ring (encode) or a
// 32-KiB table pair (decode): ~140 KiB of LDS per four waves, ONE wave per SIMD -- right for 65 536 streams, where
// there is no second wave to schedule anyway.  With more streams a second resident wave per SIMD is worth more than
// the second tile buffer: a lone wave issues at most one instruction per ~4.7 cycles while the SIMD can take one per
// ~2.4 from two waves (scripts/microbench/occupancy.hip), and the sibling's instructions fill the cycles a wave waits
// for its table lookup.  So here
//   encode   one tile + a 32-slot ring per wave (17 KiB), 256-thread workgroups of <= 80 KiB: two per CU;
//            main loop = cst_encode_loop_1buf.inc (scripts/gen_encode_loop.py, SINGLE variant)
//   decode   one tile + the 32-slot ring per wave, ONE packed table  c | p << 12 | index << 24  (16 KiB, needs at most
//            256 symbols) shared by a 512-thread workgroup: eight waves per CU in 153 KiB;
//            main loop = cst_decode_loop_small.inc (scripts/gen_decode_loop_small.py)
// Same recurrences (stack.rs:1014-1048, 1070-1100), same slabs, same results as the other kernels; shapes these do not
// take (other presets, symbol-major matrices, larger alphabets) stay where they were.
#include "cst_ans_kernels.hpp"

namespace cst {

constexpr int kSmRingSlots = 32;
constexpr int kSmAhead = 24;
constexpr uint32_t kSmRingMask = (kSmRingSlots - 1) * kWave * 4;
constexpr size_t kSmWaveRingBytes = (size_t)kSmRingSlots * kWave * 4;                  // 8 KiB
constexpr size_t kSmWaveTileBytes = (size_t)kWave * kTileStride * 4;                   // 9 KiB
constexpr int kSmDecThreads = 512;
constexpr size_t kSmLutBytes = (size_t)4 << 12;                                        // packed table for P <= 12
constexpr size_t kSmDumpBytes = 4 * kWave * 4;                                         // ONE landing area for unused chunk slots (never read)

__device__ __forceinline__ void encode_tiles_loop_1buf(uint32_t& lo, uint32_t& hi, uint32_t& wr, uint32_t& flushed, int32_t& smin,
                                                       int32_t& smax, uint32_t tile_row_addr, uint32_t tile_tr_addr, uint32_t ring_lane_addr,
                                                       uint32_t cap, uint32_t slab_off, uint32_t table_addr_biased, uint32_t P,
                                                       uint32_t ring_mask, const void* words_base, uint64_t symbols_base, uint32_t n_tiles,
                                                       const uint32_t (&goff)[8]) {
#include "cst_encode_loop_1buf.inc"
}

__device__ __forceinline__ void decode_tiles_loop_small(uint32_t& lo, uint32_t& hi, uint32_t& rd, uint32_t& lo_issued, uint32_t lut_addr,
                                                        uint32_t mask, uint32_t ring_mask, uint32_t P, const void* words_base,
                                                        uint64_t store_base, uint32_t n_tiles, int32_t min_symbol, uint32_t shift_minus_1,
                                                        uint32_t ring_lane_addr, uint32_t dump_addr, uint32_t words_off,
                                                        uint32_t tile_row_addr, uint32_t tile_tr_addr, const uint32_t (&goff)[8], bool plain_stores) {
    if (plain_stores) {          // rows that are not cache-line aligned: see scripts/gen_decode_loop.py (CST_STORE_MOD)
#define CST_STORE_MOD ""
#include "cst_decode_loop_small.inc"
#undef CST_STORE_MOD
    } else {
#define CST_STORE_MOD "nt"
#include "cst_decode_loop_small.inc"
#undef CST_STORE_MOD
    }
}

// LDS layout: [word rings, 8 KiB per wave][encoder table][symbol tiles, 9 KiB per wave]
__global__ __launch_bounds__(kBlock, 2) void ans_encode_small_kernel(const AnsEncodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kWaves = kBlock / kWave;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = threadIdx.x >> 6;
    const int P = a.precision;
    const size_t table_bytes = (((size_t)a.n_symbols * sizeof(EncEntry)) + 15) & ~(size_t)15;
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem) + wave_in_block * (kSmRingSlots * kWave);
    EncEntry* table = reinterpret_cast<EncEntry*>(smem + kWaves * kSmWaveRingBytes);
    int32_t* tile = reinterpret_cast<int32_t*>(smem + kWaves * kSmWaveRingBytes + table_bytes) + wave_in_block * (kWave * kTileStride);
    if ((lds_addr(ring) & (uint32_t)(kSmWaveRingBytes - 1)) != 0) __builtin_trap();
    for (int i = threadIdx.x; i < a.n_symbols; i += blockDim.x) table[i] = pack_entry(a.enc[i], P);   // (see cst_ans_kernels.hpp)
    __syncthreads();
    auto entry = [&](uint32_t idx) { return unpack_entry(table[idx]); };                                  // for the C++ paths

    const size_t s0 = (size_t)blockIdx.x * kBlock + (size_t)wave_in_block * kWave;
    if (s0 >= a.n_streams) return;
    const size_t s = s0 + lane;
    const bool active = s < a.n_streams;
    const size_t N = a.n_per_stream;
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;
    const uint32_t nsym = (uint32_t)a.n_symbols;

    EncLane<32, 64, kSmRingSlots> L;
    L.init(a.words + (active ? s : 0) * a.stride_words,
           active ? (uint32_t)(a.stride_words > 0xffffffffull ? 0xffffffffull : a.stride_words) : 0u, ring, lane);
    if (raw && active) L.state = a.state[s];

    const int32_t* my = a.symbols + (active ? s : 0) * N;
    const size_t n_full = N / kTileSyms;          // (the launcher only takes N % 4 == 0 and a 16-byte aligned matrix)
    // ragged top part [32 * n_full, N): direct reads
    for (size_t t = N; t > n_full * kTileSyms;) {
        --t;
        const int32_t v = active ? my[t] : a.min_symbol;
        L.template step<true>(entry(enc_index(v, a.min_symbol, nsym, L.bad)), P);
        L.flush_chunks();
    }
    bool done = false;
    if (n_full > 0 && s0 + kWave <= a.n_streams && N < (1u << 24)) {
        // ---- main loop as one asm statement (full wave, 64-byte aligned slabs of whole 64-byte groups, 32-bit offsets) ----
        const uint64_t slab_off = (uint64_t)(reinterpret_cast<const unsigned char*>(L.out.base16) - reinterpret_cast<const unsigned char*>(a.words));
        const bool ok = slab_off + 4ull * L.out.cap < 0x100000000ull && (reinterpret_cast<uintptr_t>(L.out.base16) & 63) == 0 &&
                        (L.out.cap & 15u) == 0 && L.out.shift == 0;
        if (!__any(!ok)) {
            uint32_t goff[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) goff[k] = (uint32_t)((((size_t)(lane >> 3) + 8 * k) * N + 4 * (size_t)(lane & 7)) * 4);
            const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(a.symbols + s0 * N + (n_full - 1) * kTileSyms);
            const uint64_t symbols_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                          (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
            uint32_t lo = (uint32_t)L.state, hi = (uint32_t)((uint64_t)L.state >> 32);
            int32_t smin = a.min_symbol, smax = a.min_symbol;
            const uint32_t tr_off = (uint32_t)(((lane >> 3) * kTileStride + 4 * (lane & 7)) * 4);
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the statement keeps its own book from here
            encode_tiles_loop_1buf(lo, hi, L.out.wr, L.out.flushed, smin, smax, lds_addr(tile + lane * kTileStride), lds_addr(tile) + tr_off,
                                   L.out.lane_addr, L.out.cap, (uint32_t)slab_off, lds_addr(table) - 16u * (uint32_t)a.min_symbol, (uint32_t)P,
                                   kSmRingMask, a.words, symbols_base, (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)n_full), goff);
            L.state = ((uint64_t)hi << 32) | lo;
            L.bad = max(L.bad, max((uint32_t)smax - (uint32_t)a.min_symbol, (uint32_t)smin - (uint32_t)a.min_symbol));
            done = true;
        }
    }
    if (n_full > 0 && !done) {      // partial wave / unaligned slabs: compiler-scheduled tiles, hand-scheduled steps
        int32_t r[kTileSyms];
        tile_fetch<true>(a.symbols, a.n_streams, N, s0, (n_full - 1) * kTileSyms, lane, r);
        const int32_t* row = tile + lane * kTileStride;
        for (size_t tb = n_full; tb-- > 0;) {
            wave_lds_fence();
            tile_to_lds<true>(tile, lane, r);
            wave_lds_fence();
            L.flush_chunks();
            if (tb > 0) tile_fetch<true>(a.symbols, a.n_streams, N, s0, (tb - 1) * kTileSyms, lane, r);
#pragma unroll
            for (int j = kTileSyms / 4 - 1; j >= 0; --j) {
                int4 v = *reinterpret_cast<const int4*>(row + 4 * j);
                if (!active) v = make_int4(a.min_symbol, a.min_symbol, a.min_symbol, a.min_symbol);
                const EncEntry e3 = entry(enc_index(v.w, a.min_symbol, nsym, L.bad)), e2 = entry(enc_index(v.z, a.min_symbol, nsym, L.bad)),
                               e1 = entry(enc_index(v.y, a.min_symbol, nsym, L.bad)), e0 = entry(enc_index(v.x, a.min_symbol, nsym, L.bad));
                L.template step<true>(e3, P); L.template step<true>(e2, P); L.template step<true>(e1, P); L.template step<true>(e0, P);
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

// LDS layout: [word rings, 8 KiB per wave][packed table 4 B x 2^P][symbol tiles, 9 KiB per wave][landing area for unused chunk slots]
__global__ __launch_bounds__(kSmDecThreads) void ans_decode_small_kernel(const AnsDecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kWaves = kSmDecThreads / kWave;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = threadIdx.x >> 6;
    const int P = a.precision;
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem) + wave_in_block * (kSmRingSlots * kWave);
    uint32_t* lut = reinterpret_cast<uint32_t*>(smem + kWaves * kSmWaveRingBytes);
    int32_t* tile = reinterpret_cast<int32_t*>(smem + kWaves * kSmWaveRingBytes + kSmLutBytes) + wave_in_block * (kWave * kTileStride);
    uint32_t* dump = reinterpret_cast<uint32_t*>(smem + kWaves * kSmWaveRingBytes + kSmLutBytes + kWaves * kSmWaveTileBytes) + lane;
    if ((lds_addr(ring) & (uint32_t)(kSmWaveRingBytes - 1)) != 0) __builtin_trap();
    // quantile -> c | p << 12 | index << 24   (lookup_contiguous.rs:564-605 as ONE table read; c, p < 2^12, index < 2^8)
    for (int q = threadIdx.x; q < (1 << P); q += blockDim.x) {
        const uint32_t cp = a.dec_cp[q];
        lut[q] = (cp & 0xfffu) | ((cp >> 16) << 12) | ((uint32_t)a.dec_idx[q] << 24);
    }
    __syncthreads();

    const size_t s0 = (size_t)blockIdx.x * kSmDecThreads + (size_t)wave_in_block * kWave;
    if (s0 >= a.n_streams) return;
    const size_t s = s0 + lane;
    const bool active = s < a.n_streams;
    const size_t N = a.n_per_stream;
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;
    const uint32_t qmask = (1u << P) - 1u;

    DecLane<32, 64, kSmRingSlots, kSmAhead> L;
    const WordSlice ws = active ? word_slice(a.offsets, a.stride_words, a.n_words, s, a.words_capacity) : WordSlice{0, 0u, false};
    L.init(a.words + ws.off, ws.n, ring, lane);
    if (raw) L.state = active ? a.state[s] : 0;
    else L.read_initial_state();
    L.in.prime();
    wave_lds_fence();
    uint32_t lo = (uint32_t)L.state, hi = (uint32_t)(L.state >> 32);

    auto decode_one = [&]() -> int32_t {
        const uint32_t q = lo & qmask;
        const uint32_t next_word = *L.in.slot(L.in.rd - 1u + L.in.shift);
        const uint32_t e = lut[q];
        const uint32_t c = e & 0xfffu, p = (e >> 12) & 0xfffu;
        const uint32_t s_lo = __builtin_amdgcn_alignbit(hi, lo, P), s_hi = hi >> P;
        const uint64_t t = (uint64_t)s_lo * p + (uint64_t)(q - c);            // stack.rs:1086-1088 on 32-bit halves (P >= 8)
        const uint32_t t_lo = (uint32_t)t;
        const uint32_t t_hi = __umul24(s_hi, p) + (uint32_t)(t >> 32);
        const bool refill = t_hi == 0u && L.in.rd > 0u;                       // stack.rs:1089-1097
        lo = refill ? next_word : t_lo;
        hi = refill ? t_lo : t_hi;
        L.in.rd -= refill ? 1u : 0u;
        return a.min_symbol + (int32_t)(e >> 24);
    };

    int32_t* row = a.symbols + (active ? s : 0) * N;
    const size_t n_full = N / kTileSyms;
    int32_t* my = tile + lane * kTileStride;
    size_t tb = 0;
    if (n_full > 0 && s0 + kWave <= a.n_streams && N < (1u << 24)) {
        const unsigned char* words_base = reinterpret_cast<const unsigned char*>(reinterpret_cast<uintptr_t>(a.words) & ~(uintptr_t)15);
        const uint64_t w_off = (uint64_t)(reinterpret_cast<const unsigned char*>(L.in.base16) - words_base);
        const bool off_ok = w_off + 4ull * ((uint64_t)L.in.rd + 8) < 0x80000000ull;
        if (!__any(!off_ok)) {
            uint32_t goff[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) goff[k] = (uint32_t)((((size_t)(lane >> 3) + 8 * k) * N + 4 * (size_t)(lane & 7)) * 4);
            const uint32_t tr_off = (uint32_t)(((lane >> 3) * kTileStride + 4 * (lane & 7)) * 4);
            const uint64_t sb = (uint64_t)reinterpret_cast<uintptr_t>(a.symbols + s0 * N);
            const uint64_t store_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
            // rows that do not start on cache-line boundaries: plain tile stores (scripts/gen_decode_loop.py, CST_STORE_MOD)
            const bool plain_stores = __builtin_amdgcn_readfirstlane((int)(((N * 4) % 128 != 0 || (sb & 127) != 0) ? 1 : 0)) != 0;
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the statement keeps its own book from here
            decode_tiles_loop_small(lo, hi, L.in.rd, L.in.lo_issued, lds_addr(lut), qmask, kSmRingMask, (uint32_t)P, words_base, store_base,
                                    (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)n_full), a.min_symbol, L.in.shift - 1u,
                                    lds_addr(ring + lane), lds_addr(dump), (uint32_t)w_off, lds_addr(my), lds_addr(tile) + tr_off, goff, plain_stores);
            tb = n_full;
        }
    }
    for (; tb < n_full; ++tb) {       // partial wave: compiler-scheduled
#pragma unroll
        for (int j = 0; j < kTileSyms / 4; ++j) {
            int4 v;
            v.x = decode_one(); v.y = decode_one(); v.z = decode_one(); v.w = decode_one();
            *reinterpret_cast<int4*>(my + 4 * j) = v;
        }
        L.in.template advance_window_fixed<3>(dump);
        wave_lds_fence();
        tile_store<true>(a.symbols, a.n_streams, N, s0, tb * kTileSyms, lane, tile);
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
        if (a.n_words_out) a.n_words_out[s] = L.in.rd;
    }
}

// ---- launchers ----
static size_t small_encode_lds(const AnsEncodeArgs& a) {
    return (size_t)(kBlock / kWave) * (kSmWaveRingBytes + kSmWaveTileBytes) + ((((size_t)a.n_symbols * sizeof(EncEntry)) + 15) & ~(size_t)15);
}

// More streams than one wave per SIMD of this device, and a shape the small-footprint kernels take?
// CST_SMALL_KERNELS=0 / =enc / =dec (A/B runs): never take the small-footprint kernels / only the encoder / only the decoder
static bool small_allowed(bool encode) {
    return encode ? knobs().small_encoders : knobs().small_decoders;
}

bool small_encode_usable(const AnsEncodeArgs& a, cst_coder_config cfg, cst_layout layout, int device_cus) {
    if (!small_allowed(true)) return false;
    if (cfg.word_bits != 32 || layout != CST_LAYOUT_STREAM_MAJOR || a.precision < 8 || a.precision > 12) return false;
    if (a.n_per_stream % 4 != 0 || (reinterpret_cast<uintptr_t>(a.symbols) & 15) != 0) return false;
    if (a.n_streams <= (size_t)device_cus * kBlock) return false;
    return small_encode_lds(a) <= 80 * 1024;          // two workgroups per CU
}

bool small_decode_usable(const AnsDecodeArgs& a, cst_coder_config cfg, cst_layout layout, int device_cus) {
    if (!small_allowed(false)) return false;
    if (cfg.word_bits != 32 || layout != CST_LAYOUT_STREAM_MAJOR || a.precision < 8 || a.precision > 12) return false;
    if (a.n_per_stream % 4 != 0 || (reinterpret_cast<uintptr_t>(a.symbols) & 15) != 0) return false;
    if (!a.dec_cp || !a.dec_idx || a.n_symbols > 256) return false;
    return a.n_streams > (size_t)device_cus * kBlock;
}

cst_status ans_encode_small(const AnsEncodeArgs& a, hipStream_t hs) {
    const size_t blocks = (a.n_streams + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    const size_t lds = small_encode_lds(a);
    if (lds > 64 * 1024)
        CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(ans_encode_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(ans_encode_small_kernel, dim3((unsigned)blocks), dim3(kBlock), lds, hs, a);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

cst_status ans_decode_small(const AnsDecodeArgs& a, hipStream_t hs) {
    const size_t blocks = (a.n_streams + kSmDecThreads - 1) / kSmDecThreads;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    const size_t lds = (size_t)(kSmDecThreads / kWave) * (kSmWaveRingBytes + kSmWaveTileBytes) + kSmLutBytes + kSmDumpBytes;
    CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(ans_decode_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(ans_decode_small_kernel, dim3((unsigned)blocks), dim3(kSmDecThreads), lds, hs, a);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

} // namespace cst
