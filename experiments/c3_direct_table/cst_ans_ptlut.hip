// cst_ans_ptlut.hip -- ONE MODEL PER STREAM (BASELINE config C3), decoder with a direct quantile -> symbol table per stream.
//
// The compact rows of cst_ans_pt.hip keep 256 tables next to the rings of a workgroup, and their price is the step: bucket index,
// six row entries, five compares, the run-length fields -- 37 VALU instructions and three dependent LDS reads per symbol
// (profiles/r05_sublane_counters.md: four waves per SIMD issue VALU 74 % of the time).  The reference's lookup decoder model
// (src/stream/model/categorical/lookup_contiguous.rs:564-605) is ONE table read: quantile -> symbol, then the symbol's (left
// cumulative, probability).  A stream's 2^P-byte table only fits in LDS if few streams share a workgroup -- which is what jump
// points give: with THIRTY-TWO lanes per stream (Pos / Seek, stack.rs:1107-1139: jump points every N / 32 symbols) a workgroup of
// 768 lanes decodes 24 streams, and 24 x (4096 + 520) bytes of tables + twelve 4-KiB word rings are 156 KiB: one workgroup per CU,
// three waves per SIMD.  The step is 1 + 2 LDS reads (table byte; the two 16-bit cumulatives around the symbol) and ~20 VALU.
//
//   LDS:  [word rings: 16 slots x 64 lanes x 4 B per wave][dump 1 KiB][quantile table 2^P B per stream][cdf16 row 520 B per stream]
//   the table is BUILT by the workgroup from the stream's 16-bit cdf row (model->d_cdf16): zero, mark the first quantile of every
//   symbol but the first with a 1, byte-wise inclusive prefix sum (32 lanes per stream, 2^P / 32 bytes each) -- ~250 instructions
//   per lane against 128 symbols x 20: the tables are never stored in HBM.
//   symbols leave from registers: a lane owns 32 consecutive symbols of its chunk = one 128-byte line (int32) and writes it with
//   eight back-to-back 16-byte stores; int8 matrices: 32 bytes per tile.
//
// Words, counts and status are those of any other decoder of these words; shapes this kernel does not take (chunks that are not
// N / 32 symbols of whole 32-symbol tiles, P > 12, more than 256 symbols) stay on ans_decode_pt_sub_kernel.
#include "cst_ans_kernels.hpp"

// Order of the vector-memory operations of the main loop (the compiler keeps the book, and its waits around a loop's back edge are
// vmcnt(0..1): whatever is in flight at a landing point is waited for).  Per 32-symbol tile:
//   P0: land P1's loads | the PREVIOUS tile's symbol stores | two window loads | 16 symbols | P1: land, two window loads | 16 symbols
// -- the stores are older than every load that is waited for, and sixteen symbols old when the first such wait comes.
#if defined(LUT_EXP_NO_STORES)
#define LUT_STORE(v, p) do { if ((v).x == 0x12345678u) *(p) = (v); } while (0)
#else
#define LUT_STORE(v, p) (*(p) = (v))
#endif

namespace cst {

constexpr int kLutLanes = 32;                 // lanes = jump points per stream
constexpr int kLutStreams = 24;               // streams per workgroup
constexpr int kLutThreads = kLutLanes * kLutStreams;          // 768: twelve waves, three per SIMD
constexpr int kLutWaves = kLutThreads / kWave;
constexpr int kLutSlots = 16;                 // ring slots per lane
constexpr int kLutAhead = 12;                 // words kept requested below the read position (a half tile reads at most six)
constexpr int kLutCdfStride = 260;            // uint16 entries per staged cdf row (n + 1 <= 257 used)
constexpr size_t kLutRingBytes = (size_t)kLutWaves * kLutSlots * kWave * 4;      // 48 KiB
constexpr size_t kLutDumpBytes = 4 * kWave * 4;

struct PtLutArgs {
    void* symbols_out;
    size_t n_streams, n_per_stream, interval;
    int32_t precision, n_symbols, min_symbol;
    const uint16_t* cdf16;
    int32_t cdf16_stride;
    const uint32_t* words_in;
    const uint64_t* offsets;
    size_t stride_words;
    uint64_t words_capacity;
    const uint32_t* ckpt_pos;
    const uint64_t* ckpt_state;
    int32_t* status;
};

static size_t pt_lut_lds_bytes(int P) { return kLutRingBytes + kLutDumpBytes + (size_t)kLutStreams * (((size_t)1 << P) + 2 * kLutCdfStride); }

template <int SB>
__global__ __launch_bounds__(kLutThreads) void ans_decode_pt_lut_kernel(const PtLutArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const int P = a.precision;
    const uint32_t lut_bytes = 1u << P;
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem) + wave * (kLutSlots * kWave) + lane;
    uint32_t* dump = reinterpret_cast<uint32_t*>(smem + kLutRingBytes) + lane;
    uint8_t* lut_all = smem + kLutRingBytes + kLutDumpBytes;
    uint16_t* cdf_all = reinterpret_cast<uint16_t*>(lut_all + (size_t)kLutStreams * lut_bytes);

    const size_t block_s0 = (size_t)blockIdx.x * kLutStreams;
    const uint32_t ls = (uint32_t)tid / kLutLanes, chunk = (uint32_t)tid % kLutLanes;
    const bool active = block_s0 + ls < a.n_streams;
    const size_t s = active ? block_s0 + ls : block_s0;
    const uint32_t n = (uint32_t)a.n_symbols;

    // ---- the workgroup's tables ----
    for (uint32_t i = tid; i < (uint32_t)(kLutStreams * kLutCdfStride); i += kLutThreads) {
        const uint32_t j = i / kLutCdfStride, e = i % kLutCdfStride;
        const size_t sj = block_s0 + j;
        cdf_all[i] = (sj < a.n_streams && e <= n) ? a.cdf16[sj * (size_t)a.cdf16_stride + e] : (uint16_t)0xffff;
    }
    uint8_t* lut = lut_all + (size_t)ls * lut_bytes;
    const uint16_t* cdf = cdf_all + ls * kLutCdfStride;
    const uint32_t seg = lut_bytes / kLutLanes;                       // bytes of the table this lane builds (P >= 8: >= 8)
    // (a lane's piece of the table is 2^P / 32 bytes: 32 dwords at P = 12, so the pieces of the 32 lanes of a stream start on TWO banks --
    //  every lane walks its piece rotated by its own index, (t + lane) mod 32, which puts the 32 lanes on 32 different banks)
    const uint32_t nd = seg / 4;
    uint32_t* lut32 = reinterpret_cast<uint32_t*>(lut);
#ifndef LUT_EXP_NO_BUILD
    for (uint32_t j = 0; j < nd; ++j) lut32[j * kLutLanes + chunk] = 0u;
    __syncthreads();
    if (active)
        for (uint32_t i = 1 + chunk; i < n; i += kLutLanes) {         // the first quantile of symbol i
            const uint32_t c = cdf[i];
            if (c < lut_bytes) lut[c] = 1;
        }
    __syncthreads();
    {   // inclusive prefix sum of the marks, byte by byte: table[q] = index of the symbol whose interval holds q (< 256: no carries)
        uint32_t* mine = lut32 + chunk * nd;
        uint32_t total = 0, below = 0;                                // marks of the piece; of its dwords in front of the rotation's start
        for (uint32_t t = 0; t < nd; ++t) {
            const uint32_t j = (t + chunk) & (nd - 1u);
            const uint32_t c = (uint32_t)__builtin_popcount(mine[j]);
            total += c;
            below += j < (chunk & (nd - 1u)) ? c : 0u;
        }
        uint32_t incl = total;
#pragma unroll
        for (int d = 1; d < kLutLanes; d <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl, d, kLutLanes);
            incl += chunk >= (uint32_t)d ? o : 0u;
        }
        const uint32_t base = incl - total;
        uint32_t run = base + below;
        for (uint32_t t = 0; t < nd; ++t) {
            const uint32_t j = (t + chunk) & (nd - 1u);
            run = j == 0u ? base : run;
            uint32_t x = mine[j];
            x += x << 8;
            x += x << 16;
            x += run * 0x01010101u;
            mine[j] = x;
            run = x >> 24;
        }
    }
    __syncthreads();

#endif
    // ---- AnsCoder::seek(pos, state) of the lane's chunk ----
    const size_t v = s * kLutLanes + chunk;
    const size_t N = a.n_per_stream, K = a.interval;
    const WordSlice ws = active ? word_slice_n(a.offsets, a.stride_words, a.ckpt_pos[v], s, a.words_capacity) : WordSlice{0, 0u, false};
    const uint32_t* words = a.words_in + ws.off;
    const uint32_t shift = (uint32_t)((reinterpret_cast<uintptr_t>(words) & 15) >> 2);
    const uint32_t* base16 = words - shift;
    const uint32_t top = ws.n + shift;                                    // words not yet consumed, counted from base16
    const uint64_t state0 = active && !ws.bad ? a.ckpt_state[v] : 0;
    uint32_t lo = (uint32_t)state0, hi = (uint32_t)(state0 >> 32);

    auto slot = [&](uint32_t pos) -> uint32_t* { return ring + (pos & (kLutSlots - 1)) * kWave; };
    uint32_t lo_req;
    {   // the ring: the kLutAhead words below the read position, all four chunks requested before the first is waited for
        const uint32_t start = (top + 3) & ~3u;
        const uint32_t want_lo = top > (uint32_t)kLutAhead ? top - kLutAhead : 0u;
        uint4 c[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            c[k] = start > want_lo + 4u * (uint32_t)k ? *reinterpret_cast<const uint4*>(base16 + (start - 4u * (uint32_t)(k + 1))) : make_uint4(0, 0, 0, 0);
        lo_req = start;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (start > want_lo + 4u * (uint32_t)k) {
                lo_req = start - 4u * (uint32_t)(k + 1);
                uint32_t* b = slot(lo_req);
                b[0] = c[k].x; b[kWave] = c[k].y; b[2 * kWave] = c[k].z; b[3 * kWave] = c[k].w;
            }
    }
    uint32_t tm1 = top - 1u;                           // position of the next word to read (top - 1; "none left" = shift - 1, as signed)
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    uint4 pend0 = make_uint4(0, 0, 0, 0), pend1 = make_uint4(0, 0, 0, 0);
    int32_t pos0 = -1, pos1 = -1;
    // every 16 symbols: land what the last point requested, request the next two chunks (unconditional loads and landings: slots
    // without a request read the stream's first chunk again and land in the dump rows -- see RingReader::advance_window_fixed)
    auto land = [&]() {
        uint32_t* b0 = pos0 >= 0 ? slot((uint32_t)pos0) : dump;
        b0[0] = pend0.x; b0[kWave] = pend0.y; b0[2 * kWave] = pend0.z; b0[3 * kWave] = pend0.w;
        uint32_t* b1 = pos1 >= 0 ? slot((uint32_t)pos1) : dump;
        b1[0] = pend1.x; b1[kWave] = pend1.y; b1[2 * kWave] = pend1.z; b1[3 * kWave] = pend1.w;
    };
    auto request = [&]() {
        const uint32_t top_now = tm1 + 1u;
        const uint32_t want_lo = top_now > (uint32_t)kLutAhead ? top_now - kLutAhead : 0u;
        const bool r0 = lo_req > want_lo;
        lo_req -= r0 ? 4u : 0u;
        pos0 = r0 ? (int32_t)lo_req : -1;
        pend0 = *reinterpret_cast<const uint4*>(base16 + (r0 ? lo_req : 0u));
        const bool r1 = lo_req > want_lo;
        lo_req -= r1 ? 4u : 0u;
        pos1 = r1 ? (int32_t)lo_req : -1;
        pend1 = *reinterpret_cast<const uint4*>(base16 + (r1 ? lo_req : 0u));
    };

    const uint32_t qmask = lut_bytes - 1u;
    auto decode_one = [&]() -> uint32_t {              // decode_symbol, stack.rs:1070-1100; the model: lookup_contiguous.rs:564-605
        const uint32_t q = lo & qmask;
#ifdef LUT_EXP_NO_RING
        const uint32_t next_word = q * 2654435761u;
#else
        const uint32_t next_word = *slot(tm1);
#endif
#ifdef LUT_EXP_NO_LUTREAD
        const uint32_t idx = (q * n) >> P;
#else
        const uint32_t idx = lut[q];
#endif
#ifdef LUT_EXP_CDF_B32
        const uint32_t cc = *reinterpret_cast<const uint32_t*>(cdf + (idx & ~1u));
        const uint32_t c0 = cc & 0xffffu, c1 = (cc >> 16) + 1u;
#else
        const uint32_t c0 = cdf[idx], c1 = cdf[idx + 1u];
#endif
        const uint32_t p = c1 - c0, d = q - c0;
        const uint32_t s_lo = __builtin_amdgcn_alignbit(hi, lo, P), s_hi = hi >> P;
        const uint64_t t = (uint64_t)s_lo * p + (uint64_t)d;
        const uint32_t t_lo = (uint32_t)t;
        const uint32_t t_hi = __umul24(s_hi, p) + (uint32_t)(t >> 32);
        const bool refill = t_hi == 0u && (int32_t)tm1 >= (int32_t)shift;
        lo = refill ? next_word : t_lo;
        hi = refill ? t_lo : t_hi;
        tm1 -= refill ? 1u : 0u;
        return idx;
    };

    unsigned char* row = reinterpret_cast<unsigned char*>(a.symbols_out) + (s * N + (size_t)chunk * K) * SB;
    const uint32_t min_sym = (uint32_t)a.min_symbol;
    constexpr int kOut = 32 / (SB == 1 ? 4 : 1);
    uint32_t out[kOut];
    auto store_tile = [&](size_t t0) {                 // the 32 symbols of tile t0 / 32: one 128-byte line of an int32 row, 32 bytes of an int8 row
        if (!active) return;
#ifdef LUT_EXP_COALESCED      // timing only: the same bytes per instruction, but adjacent lanes write adjacent 16-byte pieces
        v4u* dst = reinterpret_cast<v4u*>(reinterpret_cast<unsigned char*>(a.symbols_out) + ((s - ls % 2) * N + (size_t)(chunk & 31) * 0 + t0 * 64) * SB) + (tid & 63);
#pragma unroll
        for (int j = 0; j < kOut / 4; ++j) {
            v4u x = {out[4 * j], out[4 * j + 1], out[4 * j + 2], out[4 * j + 3]};
            LUT_STORE(x, dst + 64 * j);
        }
#else
        v4u* dst = reinterpret_cast<v4u*>(row + t0 * SB);
#pragma unroll
        for (int j = 0; j < kOut / 4; ++j) {
            v4u x = {out[4 * j], out[4 * j + 1], out[4 * j + 2], out[4 * j + 3]};
            LUT_STORE(x, dst + j);
        }
#endif
    };
    const uint32_t m4 = (min_sym & 0xffu) * 0x01010101u;
    for (size_t t0 = 0; t0 < K; t0 += 32) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            land();
            if (h == 0 && t0 > 0) store_tile(t0 - 32);
            request();
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const uint32_t idx = decode_one();
                const int at = 16 * h + i;
                if constexpr (SB == 1) {
                    if (at % 4 == 0) out[at / 4] = idx; else out[at / 4] |= idx << (8 * (at % 4));
                } else {
                    out[at] = idx + min_sym;
                }
            }
        }
        if constexpr (SB == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j)        // four bytes + min_symbol, byte by byte (no carries between them)
                out[j] = ((out[j] & 0x7f7f7f7fu) + (m4 & 0x7f7f7f7fu)) ^ ((out[j] ^ m4) & 0x80808080u);
        }
    }
    if (K > 0) store_tile(K - 32);
    if (active) a.status[v] = ws.bad ? (int32_t)CST_STREAM_INVALID_DATA : (int32_t)CST_STREAM_OK;
}

bool pt_lut_usable(const cst_model* m, cst_coder_config cfg, size_t n_streams, size_t n_per_stream, size_t interval, const void* d_symbols,
                   int symbol_bytes) {
    if (!knobs().pt_lut || !m->per_stream || !m->d_cdf16 || cfg.word_bits != 32 || cfg.state_bits != 64) return false;
    if (m->precision < 8 || m->precision > 12 || m->n_symbols > 256 || m->n_symbols < 1 || m->n_tables != n_streams) return false;
    if (interval == 0 || interval % 32 != 0 || n_per_stream != interval * kLutLanes) return false;
    if (symbol_bytes != 4 && symbol_bytes != 1) return false;
    if (symbol_bytes == 1 && (m->min_symbol < -128 || m->min_symbol + m->n_symbols - 1 > 127)) return false;
    if ((reinterpret_cast<uintptr_t>(d_symbols) & 15) != 0) return false;
    if ((n_streams + kLutStreams - 1) / kLutStreams > 0x7fffffffull || n_streams * kLutLanes > 0x7fffffffull) return false;
    return pt_lut_lds_bytes(m->precision) <= 160 * 1024;
}

cst_status ans_decode_pt_lut(const cst_model* model, const uint32_t* d_words, const uint64_t* d_offsets, size_t stride_words, size_t words_capacity,
                             size_t interval, const uint32_t* d_ckpt_pos, const uint64_t* d_ckpt_state, void* d_symbols, int symbol_bytes,
                             size_t n_streams, size_t n_per_stream, int32_t* d_status, hipStream_t hs) {
    PtLutArgs a{};
    a.symbols_out = d_symbols; a.n_streams = n_streams; a.n_per_stream = n_per_stream; a.interval = interval;
    a.precision = model->precision; a.n_symbols = model->n_symbols; a.min_symbol = model->min_symbol;
    a.cdf16 = model->d_cdf16; a.cdf16_stride = model->cdf16_stride;
    a.words_in = d_words; a.offsets = d_offsets; a.stride_words = stride_words;
    a.words_capacity = words_capacity ? words_capacity : (d_offsets ? 0 : n_streams * stride_words);
    a.ckpt_pos = d_ckpt_pos; a.ckpt_state = d_ckpt_state; a.status = d_status;
    const size_t lds = pt_lut_lds_bytes(model->precision);
    const unsigned blocks = (unsigned)((n_streams + kLutStreams - 1) / kLutStreams);
    auto go = [&](auto kernel) -> cst_status {
        CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kLutThreads), lds, hs, a);
        CST_HIP_TRY(hipGetLastError());
        return CST_OK;
    };
    return symbol_bytes == 1 ? go(ans_decode_pt_lut_kernel<1>) : go(ans_decode_pt_lut_kernel<4>);
}

} // namespace cst
