// cst_persymbol.hip -- coders driven by PER-SYMBOL entropy models (SURVEY.md 8f-1 and the single-coder drop-in).
//
//   * per-symbol quantized Gaussians: the reference's flagship call
//       coder.encode_reverse(symbols, QuantizedGaussian(lo, hi), means, stds) / coder.decode(family, means, stds)
//     (src/pybindings/stream/stack.rs:567-588, 733-751): every symbol gets its own
//     LeakilyQuantizedDistribution (src/stream/model/quantize.rs:525-568), i.e. two bit-exact f64 erf on device
//     per encoded symbol and a search over left cumulatives per decoded symbol;
//   * explicit per-symbol models: (left, prob) pairs for encoding and cdf rows for decoding.
//
// Encoding is two passes: a fully parallel pass turns every symbol into a coder entry (c, p, 2^64/p), then one
// LANE per stream runs the sequential recurrence over those entries.  Decoding uses one WAVE per stream:
// 64 lanes evaluate 64 candidate left cumulatives at once (two erf rounds / two coalesced row reads for a 201-symbol
// support).
#include <cstdlib>
#include <mutex>

#include "cst_range_kernels.hpp"
#include "cst_math.hpp"

namespace cst {

enum CoderKind : int { kAns = 0, kRange = 1, kChain = 2 };

__device__ __forceinline__ EncEntry make_entry(uint32_t c, uint32_t p) {
    uint64_t m = 0;
    if (p == 1) m = ~0ull;
    else if (p > 1) {                      // floor(2^64 / p) without 128-bit arithmetic
        const uint64_t q = (~0ull) / p;    // floor((2^64 - 1) / p)
        const uint64_t r = (~0ull) - q * p;
        m = q + ((r + 1 == p) ? 1 : 0);
    }
    return EncEntry{c, p, (uint32_t)m, (uint32_t)(m >> 32)};
}

__global__ void cp_entries_kernel(const uint32_t* __restrict__ left, const uint32_t* __restrict__ prob, size_t n, int P,
                                  EncEntry* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t c = left[i], p = prob[i];
    if ((uint64_t)c + p > ((uint64_t)1 << P)) p = 0;   // not a sub-interval of [0, 2^P): treat as impossible
    out[i] = make_entry(c, p);
}

__global__ void gaussian_entries_kernel(int P, int32_t lo, int32_t hi, const int32_t* __restrict__ sym,
                                        const double* __restrict__ mu, const double* __restrict__ sd, size_t n,
                                        EncEntry* __restrict__ out) {
    __shared__ double2 erf_tab[kErfTabEntries];
    erf_tab_fill(erf_tab, threadIdx.x, blockDim.x);
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t c = 0, p = 0;
    const double m = mu[i], s = sd[i];
    // `assert!(std > 0.0)` and finite parameters (pybindings/stream/model.rs:654-657); out-of-support symbols
    // (quantize.rs:537-539) and degenerate distributions (quantize.rs:562-565) all end up with p = 0 = impossible
    if (s > 0.0 && s <= 1.7976931348623157e308 && m == m && m <= 1.7976931348623157e308 && m >= -1.7976931348623157e308) {
        if (!leaky_gaussian_lcp_quick(sym[i], lo, hi, P, 32, m, s, c, p, erf_tab)) p = 0;
    }
    out[i] = make_entry(c, p);
}

struct EntriesEncodeArgs {
    const EncEntry* entries;
    size_t n_streams, n_per_stream;
    int32_t layout, precision;
    uint32_t* words;
    size_t stride_words;
    uint32_t* n_words;
    uint64_t* state;            // ANS raw state
    cst_range_state* rstate;    // range raw state
    int32_t* status;
    uint32_t flags;
    // chain coder: the remainders stack that is popped, and the heads
    const uint32_t* pop_words; const uint64_t* pop_offsets; size_t pop_stride; uint32_t* n_pop;
    cst_chain_heads* heads;
};

constexpr int kEntryGroup = 8;     // entries requested together (8 x 16 bytes = one 128-byte line of a stream-major row)

// one lane per stream over precomputed entries; ANS walks backwards, the range coder forwards
template <int W, int S, int KIND>
__global__ __launch_bounds__(kBlock) void encode_entries_kernel(const EntriesEncodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem) + (threadIdx.x >> 6) * kRingWords;
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = s < a.n_streams;
    const size_t N = a.n_per_stream;
    const int P = a.precision;
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;
    const int G4 = 4 * groups_per_point(W, P);
    const size_t stride_t = a.layout == CST_LAYOUT_SYMBOL_MAJOR ? a.n_streams : 1;
    const EncEntry* my = a.entries + (active ? (a.layout == CST_LAYOUT_SYMBOL_MAJOR ? s : s * N) : 0);
    uint32_t* slab = a.words + (active ? s : 0) * a.stride_words;
    const uint32_t cap = active ? (uint32_t)(a.stride_words > 0xffffffffull ? 0xffffffffull : a.stride_words) : 0u;
    uint32_t bad = 0, n_words = 0;
    int32_t status;
    int countdown = G4;

    if constexpr (KIND == kChain) {
        // ChainCoder::encode_symbol (src/stream/chain.rs:1140-1209), symbols last to first.  Not a hot path of the
        // library: plain per-lane loads and stores.
        using st_t = typename StateT<S>::type;
        constexpr uint32_t wmask = W == 32 ? 0xffffffffu : ((1u << (W % 32)) - 1u);
        status = CST_STREAM_OK;
        if (active) {
            cst_chain_heads h = a.heads[s];
            st_t rh = (st_t)h.remainders_head;
            uint32_t ch = h.compressed_head;
            const uint32_t* pop = a.pop_words + (a.pop_offsets ? a.pop_offsets[s] : s * a.pop_stride);
            uint32_t rd = a.n_pop[s];
            // entries kEntryGroup at a time, one group ahead of their use, and the next word of the popped stack one symbol
            // ahead (see the ANS branch)
            const uint32_t* idle = a.n_pop + s;
            uint32_t ahead = *(rd > 0 ? pop + (rd - 1) : idle);
            auto one = [&](const EncEntry e) {
                if (status != CST_STREAM_OK || bad) return;
                if (e.p == 0) { bad = 1; return; }
                bool refill = rh < ((st_t)e.p << (S - W - P));                  // refill_remainders_head, chain.rs:799-815
                if (refill && rd == 0) { status = CST_STREAM_OUT_OF_DATA; return; }
                rh = refill ? (st_t)((rh << (W % S)) | (st_t)(ahead & wmask)) : rh;
                rd -= refill ? 1u : 0u;
                st_t q;
                if constexpr (S == 64) q = mulhi64(rh, e.m_lo, e.m_hi);         // floor(rh / p) or one less
                else q = __umulhi(rh, e.m_hi);
                uint32_t r = (uint32_t)rh - (uint32_t)q * e.p;
                if (r >= e.p) { r -= e.p; q += 1; }
                const uint32_t quantile = e.c + r;
                rh = q;
                if (P != W && ch < (1u << (W - P))) ch = (ch << P) | quantile;
                else {
                    const uint32_t word = P == W ? quantile : (((ch << P) | quantile) & wmask);
                    if (P != W) ch >>= (W - P);
                    if (n_words < cap) slab[n_words] = word;
                    ++n_words;
                }
            };
            const size_t tail = N % kEntryGroup;
            for (size_t t = N; t-- > N - tail;) { one(my[t * stride_t]); ahead = *(rd > 0 ? pop + (rd - 1) : idle); }
            EncEntry cur[kEntryGroup], nxt[kEntryGroup];
            size_t g = N - tail;
            if (g > 0) {
#pragma unroll
                for (int k = 0; k < kEntryGroup; ++k) nxt[k] = my[(g - 1 - k) * stride_t];
            }
            while (g > 0 && status == CST_STREAM_OK && !bad) {
#pragma unroll
                for (int k = 0; k < kEntryGroup; ++k) cur[k] = nxt[k];
                g -= kEntryGroup;
                if (g > 0) {
#pragma unroll
                    for (int k = 0; k < kEntryGroup; ++k) nxt[k] = my[(g - 1 - k) * stride_t];
                }
#pragma unroll
                for (int k = 0; k < kEntryGroup; ++k) { one(cur[k]); ahead = *(rd > 0 ? pop + (rd - 1) : idle); }
            }
            if (n_words > cap) status = CST_STREAM_CAPACITY;
            h.remainders_head = (uint64_t)rh; h.compressed_head = ch;
            a.heads[s] = h;
            a.n_pop[s] = rd;
        }
    } else if constexpr (KIND == kAns) {
        EncLane<W, S> L;
        L.init(slab, cap, ring, lane);
        if (raw && active) L.state = (typename StateT<S>::type)a.state[s];
        // entries come kEntryGroup at a time, one group ahead of their use: a lone wave per SIMD has no other wave to
        // hide a load behind, and the address of every entry is known from the start
        const size_t tail = N % kEntryGroup;
        for (size_t t = N; t-- > N - tail;) {
            if (active) {
                const EncEntry e = my[t * stride_t];
                if (e.p == 0) bad = 1;
                else if (!bad) L.template step<false>(e, P);
            }
            if (--countdown == 0) { countdown = G4; L.flush_chunks(); }
        }
        EncEntry cur[kEntryGroup], nxt[kEntryGroup];
        size_t g = N - tail;                       // entries [g - kEntryGroup, g) are the next group
        if (g > 0 && active) {
#pragma unroll
            for (int j = 0; j < kEntryGroup; ++j) nxt[j] = my[(g - 1 - j) * stride_t];
        }
        while (g > 0) {
#pragma unroll
            for (int j = 0; j < kEntryGroup; ++j) cur[j] = nxt[j];
            g -= kEntryGroup;
            if (g > 0 && active) {
#pragma unroll
                for (int j = 0; j < kEntryGroup; ++j) nxt[j] = my[(g - 1 - j) * stride_t];
            }
#pragma unroll
            for (int j = 0; j < kEntryGroup; ++j) {
                if (active) {
                    if (cur[j].p == 0) bad = 1;
                    else if (!bad) L.template step<false>(cur[j], P);
                }
                if (--countdown == 0) { countdown = G4; L.flush_chunks(); }
            }
        }
        status = L.finish(!raw, 1u, n_words);
        if (active && raw) a.state[s] = (uint64_t)L.state;
    } else {
        RangeEncLane<W, S> L;
        L.init(slab, cap, ring, lane);
        if (raw && active) {
            const cst_range_state r = a.rstate[s];
            L.lower = (typename StateT<S>::type)r.lower; L.range = (typename StateT<S>::type)r.range;
            L.inv_n = r.inverted_n; L.inv_first = r.inverted_first;
        }
        EncEntry cur[kEntryGroup], nxt[kEntryGroup];
        const size_t n_grouped = N - N % kEntryGroup;
        size_t g = 0;                              // entries [g, g + kEntryGroup) are the next group
        if (n_grouped > 0 && active) {
#pragma unroll
            for (int j = 0; j < kEntryGroup; ++j) nxt[j] = my[j * stride_t];
        }
        while (g < n_grouped) {
#pragma unroll
            for (int j = 0; j < kEntryGroup; ++j) cur[j] = nxt[j];
            g += kEntryGroup;
            if (g < n_grouped && active) {
#pragma unroll
                for (int j = 0; j < kEntryGroup; ++j) nxt[j] = my[(g + j) * stride_t];
            }
#pragma unroll
            for (int j = 0; j < kEntryGroup; ++j) {
                if (active) {
                    if (cur[j].p == 0) bad = 1;
                    else if (!bad) L.step(cur[j].c, cur[j].p, P);
                }
                if (--countdown == 0) { countdown = G4; L.out.flush_chunks(); }
            }
        }
        for (size_t t = n_grouped; t < N; ++t) {
            if (active) {
                const EncEntry e = my[t * stride_t];
                if (e.p == 0) bad = 1;
                else if (!bad) L.step(e.c, e.p, P);
            }
            if (--countdown == 0) { countdown = G4; L.out.flush_chunks(); }
        }
        if (raw) {
            L.out.drain();
            n_words = L.out.wr;
            status = L.out.wr > L.out.cap ? CST_STREAM_CAPACITY : CST_STREAM_OK;
            if (active) {
                cst_range_state r = a.rstate[s];
                r.lower = (uint64_t)L.lower; r.range = (uint64_t)L.range; r.inverted_n = L.inv_n; r.inverted_first = L.inv_first;
                a.rstate[s] = r;
            }
        } else {
            status = L.finish(1u, n_words);
        }
    }
    if (!active) return;
    if (bad) status = CST_STREAM_IMPOSSIBLE_SYMBOL;
    a.status[s] = status;
    a.n_words[s] = (status == CST_STREAM_OK || KIND == kChain) ? n_words : 0u;
}

// ------------------------------------------------------------------------------------------------
// per-symbol Gaussians, ONE kernel (batches of many streams): a wave owns kFuStreams streams and alternates, tile by tile
// of kFuTile symbols, between
//   (A) all 64 lanes turning the tile's kFuStreams x kFuTile (symbol, mean, std) triples into coder entries
//       (two Gaussian cumulatives + floor(2^64 / p) each) in a wave-private LDS tile, and
//   (B) one lane per stream running the sequential coder recurrence over its row of that tile.
// Nothing but the inputs and the compressed words touches HBM: the two-pass form above writes a 16-byte entry per symbol
// and reads it back (4 GiB of scratch and 2.5x the algorithmic traffic at 65 536 x 4096).  The entry pass is the bulk of
// the work and runs with full lanes; the coder steps run on kFuStreams of the 64 lanes, which is why a wave takes 32
// streams, not 64: two waves per SIMD then cover each other's stalls.  Inputs are requested four items (~ 5000 cycles of
// arithmetic) before they are used.
// ------------------------------------------------------------------------------------------------
constexpr int kFuTile = 16;                               // symbols per tile
constexpr int kFuStreams = 32;                            // streams per wave
constexpr int kFuIters = kFuTile * kFuStreams / kWave;    // entries per lane and tile
constexpr int kFuRingSlots = 32;
constexpr int kFuAhead = 4;                               // items requested ahead of their use (kFuIters % kFuAhead == 0)
constexpr int kFuRowStride = kFuStreams + 1;              // entries: row t of the tile starts at t * kFuRowStride (conflict-free both ways)
constexpr int kFuBlock = 256;
constexpr size_t kFuTabBytes = kErfTabBytes;              // the erf tables (cst_math.hpp)
constexpr size_t kFuWaveBytes = (size_t)kFuRingSlots * kWave * 4 + (size_t)kFuTile * kFuRowStride * sizeof(EncEntry);

struct GaussianFusedArgs {
    const int32_t* symbols;
    const double* means;
    const double* stds;
    size_t n_streams, n_per_stream;
    int32_t layout, precision, lo, hi;
    uint32_t* words;
    size_t stride_words;
    uint32_t* n_words;
    uint64_t* state;
    cst_range_state* rstate;
    int32_t* status;
    uint32_t flags;
    // jump points (Pos: stack.rs:1130-1139, queue.rs:182-196), [n_streams][n_chunks], noted where a chunk of `interval` symbols starts; or
    // null.  ANS: (words in the bulk, state).  Range coder (round 6): (words emitted incl. held-back ones, lower, range).
    uint32_t* ckpt_pos;
    uint64_t* ckpt_state;
    uint64_t* ckpt_lower;
    uint64_t* ckpt_range;
    size_t interval, n_chunks;
};

// floor(2^64 / p) for 2 <= p <= 2^24 through two f64 quotients, each corrected by its exact remainder:
// 2^64 / p = 2^32 q1 + 2^32 r1 / p with q1 = floor(2^32 / p), r1 = 2^32 - q1 p
__device__ __forceinline__ EncEntry make_entry_f64(uint32_t c, uint32_t p) {
    // straight line (one model per lane: a branch would be taken by some lane every time); p <= 1 is patched in at the end
    const uint32_t pp = p > 1u ? p : 2u;
    const double inv = fast_rcp1((double)pp);                           // 2^-48: both quotients below are within one
    uint32_t q1 = f64_as_u32_hw(4294967296.0 * inv);
    int64_t r1 = (int64_t)(1ull << 32) - (int64_t)((uint64_t)q1 * pp);
    const uint32_t dn1 = r1 < 0 ? 1u : 0u, up1 = r1 >= (int64_t)pp ? 1u : 0u;
    q1 = q1 - dn1 + up1;
    const uint32_t r1u = (uint32_t)r1 + (dn1 ? pp : 0u) - (up1 ? pp : 0u);          // 0 <= r1 < p now
    const double x2 = __builtin_amdgcn_ldexp((double)r1u, 32);         // < 2^56: exact as a double
    uint32_t q2 = f64_as_u32_hw(x2 * inv);
    const int64_t r2 = (int64_t)((uint64_t)r1u << 32) - (int64_t)((uint64_t)q2 * pp);
    q2 = q2 - (r2 < 0 ? 1u : 0u) + (r2 >= (int64_t)pp ? 1u : 0u);
    const uint32_t ones = p ? 0xffffffffu : 0u;
    return EncEntry{c, p, p > 1u ? q2 : ones, p > 1u ? q1 : ones};
}

// From P = 18 on (kInvMinPrecision: the Python API's P = 24) the fused encoder's entries carry 1 / p as an f64 (2^-48:
// v_rcp_f64 + one Newton step) where the table kernels carry floor(2^64 / p): with one model per symbol the entry is built as
// often as it is used, and floor(2^64 / p) costs ~35 instructions against 4.  The (32,64) step that goes with it
// (encode_step_inv below) has the length of the table kernels' hand-scheduled one.
constexpr int kInvMinPrecision = 18;
__device__ __forceinline__ EncEntry make_entry_inv(uint32_t c, uint32_t p) {
    const double inv = fast_rcp1((double)p);                       // (p = 0: an impossible symbol, replaced before it is used)
    return EncEntry{c, p, f64_lo(inv), f64_hi(inv)};
}

// One ANS encoder step (stack.rs:1014-1048) on the 32-bit halves of a 64-bit state, 18 <= P <= 24, 1 <= p < 2^P, with
// inv = 1 / p to 2^-48:  A = emit ? state >> 32 : state  is below p 2^(64 - P) <= 2^46 p, so  A inv  is within 2^-1.9 of the
// quotient (2^-7.9 at P = 24) and its nearest integer q' is the quotient or one more; A - q' p then fits 32 signed bits and
// its sign says which.
template <int SLOTS>
__device__ __forceinline__ void encode_step_inv(EncLane<32, 64, SLOTS>& L, uint32_t c, uint32_t p, double inv, int P) {
    const uint32_t lo = (uint32_t)L.state, hi = (uint32_t)(L.state >> 32);
    const bool emit = hi >= (p << (32 - P));                       // (state >> (64 - P)) >= p
    L.out.push(lo, emit ? 1u : 0u);
    const uint32_t a0 = emit ? hi : lo, a1 = emit ? 0u : hi;
    const double af = __builtin_fma((double)a1, 4294967296.0, (double)a0);
    const double qm = af * inv + 0x1p52;                           // the integer nearest to A / p in the low mantissa bits
    const uint32_t ql = f64_lo(qm), qh = f64_hi(qm) & 0xfffffu;
    const int32_t r = (int32_t)(a0 - ql * p);                      // A - q' p, exact: -p <= r < p
    const int32_t y = r + (r < 0 ? (int32_t)(c + p) - (int32_t)(1u << P) : (int32_t)c);   // q' one too large: q = q' - 1, r + p
    L.state = ((((uint64_t)qh << 32) | ql) << P) + (uint64_t)(int64_t)y;
}

//
// PAIR (stream-major matrices of whole tiles and whole waves: the launcher checks): a tile takes 64 bytes of each stream's
// symbols -- half a 128-byte line whose other half is the NEXT tile's, and asked for a tile apart the line came from HBM twice
// (6.60 GB counted against 5.57 GB algorithmic, profiles/r04_pmc_summary.md).  So the symbols of both tiles of a line are
// requested together, a pair of tiles ahead, and parked lane by lane in the ring columns of lanes 32..63 (a wave codes
// kFuStreams = 32 streams: no coder ever writes there), where the items pick them up one item ahead of their use.
template <int W, int S, int KIND, bool PAIR = false>
__global__ __launch_bounds__(kFuBlock) void encode_gaussian_fused_kernel(const GaussianFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1), wave_in_block = threadIdx.x >> 6;
    // LDS: [word rings, one per wave, aligned to their size: the hand-scheduled step forms slot addresses with and/or]
    //      [erf tables][entry tiles, one per wave]
    constexpr size_t kRingBytes = (size_t)kFuRingSlots * kWave * 4;
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem + (size_t)wave_in_block * kRingBytes);
    double2* erf_tab = reinterpret_cast<double2*>(smem + (kFuBlock / kWave) * kRingBytes);
    EncEntry* tile = reinterpret_cast<EncEntry*>(smem + (kFuBlock / kWave) * kRingBytes + kFuTabBytes + (size_t)wave_in_block * (kFuWaveBytes - kRingBytes));
    if ((lds_addr(ring) & (uint32_t)(kRingBytes - 1)) != 0) __builtin_trap();
    erf_tab_fill(erf_tab, threadIdx.x, blockDim.x);
    __syncthreads();
    const size_t s0 = ((size_t)blockIdx.x * (kFuBlock / kWave) + wave_in_block) * kFuStreams;
    if (s0 >= a.n_streams) return;
    const size_t N = a.n_per_stream;
    const int P = a.precision;
    const bool symbol_major = a.layout == CST_LAYOUT_SYMBOL_MAJOR;
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;
    const bool use_inv = KIND == kAns && W == 32 && S == 64 && P >= kInvMinPrecision;      // entries with 1 / p (make_entry_inv)
    const size_t s = s0 + lane;
    const bool active = lane < kFuStreams && s < a.n_streams;            // this lane codes a stream in phase B

    // phase A's work items: item w = it * 64 + lane of a tile is (stream j, symbol tl); consecutive lanes take consecutive
    // addresses of the input matrices in either layout.  Items are requested ONE item ahead of their use (the loop stays
    // rolled: eight unrolled copies of two erf would not fit the instruction cache).
    const size_t n_tiles = (N + kFuTile - 1) / kFuTile;
    auto tile_of = [&](size_t step) { return KIND == kAns ? n_tiles - 1 - step : step; };   // ANS codes last to first
    auto item_j = [&](int it) { const int w = it * kWave + lane; return symbol_major ? w % kFuStreams : w / kFuTile; };
    auto item_t = [&](int it) { const int w = it * kWave + lane; return symbol_major ? w / kFuStreams : w % kFuTile; };
    // a queue of kFuAhead requested items (HBM latency is two to three items' worth of arithmetic); the item loop below is
    // unrolled by kFuAhead so that every queue slot is a fixed set of registers
    int32_t sy_q[kFuAhead];
    double mu_q[kFuAhead], sd_q[kFuAhead];
    bool ok_q[kFuAhead];
    // Full waves over rows of whole tiles walk the matrices by ADDING: item `it` of tile k lies at
    //   base(lane) + it * item_stride + k * tile_stride      (both strides wave-uniform, in either layout)
    // and the items are requested in exactly that order, so one running index per lane replaces the per-item index arithmetic
    // (two 64-bit multiply-adds, bounds tests and their exec masks: ~25 VALU and ~15 SALU per item).
    const bool walk = PAIR || (s0 + kFuStreams <= a.n_streams && N % kFuTile == 0);
    const int64_t item_stride = symbol_major ? (int64_t)(kWave / kFuStreams) * (int64_t)a.n_streams : (int64_t)(kWave / kFuTile) * (int64_t)N;
    const int64_t tile_stride = symbol_major ? (int64_t)kFuTile * (int64_t)a.n_streams : (int64_t)kFuTile;
    const int64_t wrap_delta = (KIND == kAns ? -tile_stride : tile_stride) - (int64_t)(kFuIters - 1) * item_stride;
    int64_t e_req = symbol_major ? (int64_t)item_t(0) * (int64_t)a.n_streams + (int64_t)(s0 + (size_t)item_j(0))
                                 : (int64_t)(s0 + (size_t)item_j(0)) * (int64_t)N + (int64_t)item_t(0);
    auto request = [&](int slot, size_t k, int it) {
        if (walk) {
            ok_q[slot] = true;
            if constexpr (!PAIR) sy_q[slot] = __builtin_nontemporal_load(a.symbols + e_req);
            mu_q[slot] = __builtin_nontemporal_load(a.means + e_req);
            sd_q[slot] = __builtin_nontemporal_load(a.stds + e_req);
            e_req += it == kFuIters - 1 ? wrap_delta : item_stride;
            return;
        }
        const size_t sj = s0 + (size_t)item_j(it), t = k * kFuTile + (size_t)item_t(it);
        ok_q[slot] = sj < a.n_streams && t < N;
        // (unconditional loads from an address that is always valid: a conditional load is waited for at once)
        const size_t e = ok_q[slot] ? (symbol_major ? t * a.n_streams + sj : sj * N + t) : 0;
        sy_q[slot] = __builtin_nontemporal_load(a.symbols + e);
        mu_q[slot] = __builtin_nontemporal_load(a.means + e);
        sd_q[slot] = __builtin_nontemporal_load(a.stds + e);
    };

    uint32_t* slab = a.words + (active ? s : 0) * a.stride_words;
    const uint32_t cap = active ? (uint32_t)(a.stride_words > 0xffffffffull ? 0xffffffffull : a.stride_words) : 0u;
    EncLane<W, S, kFuRingSlots> LA;
    RangeEncLane<W, S, kFuRingSlots> LR;
    if constexpr (KIND == kAns) {
        LA.init(slab, cap, ring, lane);
        if (raw && active) LA.state = (typename StateT<S>::type)a.state[s];
    } else {
        LR.init(slab, cap, ring, lane);
        if (raw && active) {
            const cst_range_state r = a.rstate[s];
            LR.lower = (typename StateT<S>::type)r.lower; LR.range = (typename StateT<S>::type)r.range;
            LR.inv_n = r.inverted_n; LR.inv_first = r.inverted_first;
        }
    }
    uint32_t bad = 0;

    // PAIR: the symbols of tiles 2 m and 2 m + 1 (one 128-byte line per stream), item `it` of the even tile in [0][it]
    int32_t sy_pair[2][kFuIters];
    int32_t sy_cur = 0;
    const int64_t sym_base = e_req;                           // item 0 of tile 0
    auto stash_slot = [&](int half, int it) {
        return ring + (((half * kFuIters + it) * 2 + (lane >> 5)) * kWave + kFuStreams + (lane & (kFuStreams - 1)));
    };
    auto pair_request = [&](size_t k_in_pair) {
        const size_t even = k_in_pair & ~(size_t)1, odd = even + 1 < n_tiles ? even + 1 : even;
        const int32_t* p0 = a.symbols + sym_base + (int64_t)even * tile_stride;
        const int32_t* p1 = a.symbols + sym_base + (int64_t)odd * tile_stride;
#pragma unroll
        for (int it = 0; it < kFuIters; ++it) {
            sy_pair[0][it] = __builtin_nontemporal_load(p0 + (int64_t)it * item_stride);
            sy_pair[1][it] = __builtin_nontemporal_load(p1 + (int64_t)it * item_stride);
        }
    };
    if (n_tiles > 0) {
        if constexpr (PAIR) pair_request(tile_of(0));
        e_req += (int64_t)tile_of(0) * tile_stride;
#pragma unroll
        for (int q = 0; q < kFuAhead; ++q) request(q, tile_of(0), q);
    }
    size_t step = 0;
    while (step < n_tiles) {
      // the tiles coded before the next symbol request: both tiles of a line (PAIR), or all of them
      size_t group_end = n_tiles;
      if constexpr (PAIR) {
          const size_t k = tile_of(step);
          const bool two = KIND == kAns ? (k & 1) != 0 : k + 1 < n_tiles;       // (ANS walks down: an odd tile, then its even partner)
          group_end = step + (two ? 2 : 1);
#pragma unroll
          for (int it = 0; it < kFuIters; ++it) {
              *stash_slot(0, it) = (uint32_t)sy_pair[0][it];
              *stash_slot(1, it) = (uint32_t)sy_pair[1][it];
          }
          pair_request(group_end < n_tiles ? tile_of(group_end) : k);          // (after the last pair: its own lines once more)
      }
      for (; step < group_end; ++step) {
        const size_t k = tile_of(step);
        wave_lds_fence();                                  // (the previous tile has been read)
        const uint32_t* stash_k = stash_slot((int)(k & 1), 0);
        if constexpr (PAIR) sy_cur = (int32_t)stash_k[0];
        // ---- phase A: entries of tile k ----
#pragma unroll 1
        for (int it0 = 0; it0 < kFuIters; it0 += kFuAhead) {
#pragma unroll
            for (int q = 0; q < kFuAhead; ++q) {
                const int it = it0 + q;
                int32_t sy;
                if constexpr (PAIR) {
                    sy = sy_cur;
                    sy_cur = (int32_t)stash_k[((it + 1) & (kFuIters - 1)) * 2 * kWave];     // (the next item's; wraps harmlessly)
                } else {
                    sy = ok_q[q] ? sy_q[q] : a.lo;                  // (items past the matrix: never coded)
                }
                const double m = ok_q[q] ? mu_q[q] : 0.0, sg = ok_q[q] ? sd_q[q] : 1.0;
                if (it + kFuAhead < kFuIters) request(q, k, it + kFuAhead);
                else if (step + 1 < n_tiles) request(q, tile_of(step + 1), it + kFuAhead - kFuIters);
                uint32_t c = 0, p = 0;
                // `assert!(std > 0.0)` and finite parameters (pybindings/stream/model.rs:654-657); out-of-support symbols
                // (quantize.rs:537-539) and degenerate distributions (quantize.rs:562-565) all end up with p = 0 = impossible.
                // No branches: invalid parameters are evaluated as (0, 1) and thrown away.
                const bool valid = sg > 0.0 && sg <= 1.7976931348623157e308 && fabs(m) <= 1.7976931348623157e308;
                const bool inside = leaky_gaussian_lcp_quick(sy, a.lo, a.hi, P, 32, valid ? m : 0.0, valid ? sg : 1.0, c, p, erf_tab);
                if (!valid || !inside || (uint64_t)c + p > ((uint64_t)1 << P)) p = 0;
                EncEntry entry{c, p, 0u, 0u};                                   // (the range coder divides by nothing)
                if constexpr (KIND == kAns) entry = use_inv ? make_entry_inv(c, p) : make_entry_f64(c, p);
                tile[item_t(it) * kFuRowStride + item_j(it)] = entry;
            }
        }
        wave_lds_fence();
        // ---- phase B: every stream's lane over its row ----
        const size_t t0 = k * kFuTile;
        const int n_here = (int)(N - t0 < (size_t)kFuTile ? N - t0 : (size_t)kFuTile);
        if constexpr (KIND == kRange) {
            // RangeEncoder::pos() in front of a chunk (a queue: BEFORE the chunk's first symbol is encoded; chunks are whole tiles)
            if (a.ckpt_pos && active && t0 % a.interval == 0) {
                a.ckpt_pos[s * a.n_chunks + t0 / a.interval] = LR.out.wr + LR.inv_n;
                a.ckpt_lower[s * a.n_chunks + t0 / a.interval] = (uint64_t)LR.lower;
                a.ckpt_range[s * a.n_chunks + t0 / a.interval] = (uint64_t)LR.range;
            }
        }
        if (active) {
            if constexpr (KIND == kAns) {
                constexpr bool FAST = W == 32 && S == 64;            // the 32-bit-halves step (8 <= P)
                if (FAST && P >= 8 && n_here == kFuTile) {
                    // a whole tile: all sixteen entries first (one LDS wait), then sixteen hand-scheduled steps.  An
                    // impossible symbol is coded as (0, 1) -- its stream is flagged and its words are never used.
                    EncEntry e[kFuTile];
#pragma unroll
                    for (int tl = 0; tl < kFuTile; ++tl) e[tl] = tile[tl * kFuRowStride + lane];
#pragma unroll
                    for (int tl = kFuTile - 1; tl >= 0; --tl) {
                        const bool none = e[tl].p == 0;
                        bad |= none ? 1u : 0u;
                        if constexpr (FAST) {
                            if (use_inv) encode_step_inv(LA, none ? 0u : e[tl].c, none ? 1u : e[tl].p, none ? 1.0 : f64_from(e[tl].m_lo, e[tl].m_hi), P);
                            else LA.template step<FAST>(EncEntry{none ? 0u : e[tl].c, none ? 1u : e[tl].p, none ? 0xffffffffu : e[tl].m_lo, none ? 0xffffffffu : e[tl].m_hi}, P);
                        }
                    }
                } else {
                    // (other presets, P < 8, the ragged tile)
                    for (int tl = n_here - 1; tl >= 0; --tl) {
                        const EncEntry e = tile[tl * kFuRowStride + lane];
                        if (e.p == 0) bad = 1;
                        else if (!bad) LA.template step<false>(use_inv ? make_entry(e.c, e.p) : e, P);
                    }
                }
            } else if (n_here == kFuTile) {
                // a whole tile: all sixteen (c, p) first (one LDS wait), then sixteen steps; an impossible symbol is coded as
                // (0, 1) -- its stream is flagged and its words are never used
                uint2 e[kFuTile];
#pragma unroll
                for (int tl = 0; tl < kFuTile; ++tl) e[tl] = *reinterpret_cast<const uint2*>(&tile[tl * kFuRowStride + lane]);
#pragma unroll
                for (int tl = 0; tl < kFuTile; ++tl) {
                    const bool none = e[tl].y == 0;
                    bad |= none ? 1u : 0u;
                    LR.step(none ? 0u : e[tl].x, none ? 1u : e[tl].y, P);
                }
            } else {
                for (int tl = 0; tl < n_here; ++tl) {
                    const EncEntry e = tile[tl * kFuRowStride + lane];
                    if (e.p == 0) bad = 1;
                    else if (!bad) LR.step(e.c, e.p, P);
                }
            }
        }
        if constexpr (KIND == kAns) {
            // AnsCoder::pos() in front of a chunk: the symbols from t0 on are encoded (chunks are whole tiles: the launcher checks)
            if (a.ckpt_pos && active && t0 % a.interval == 0) {
                a.ckpt_pos[s * a.n_chunks + t0 / a.interval] = LA.out.wr;
                a.ckpt_state[s * a.n_chunks + t0 / a.interval] = (uint64_t)LA.state;
            }
        }
        // at most kFuTile new words per stream and tile: whole chunks leave here (<= 19 pending before, < 4 after)
        if constexpr (KIND == kAns) LA.flush_chunks(); else LR.out.flush_chunks();
      }
    }

    uint32_t n_words = 0;
    int32_t status;
    if constexpr (KIND == kAns) {
        status = LA.finish(!raw, 1u, n_words);
        if (active && raw) a.state[s] = (uint64_t)LA.state;
    } else if (raw) {
        LR.out.drain();
        n_words = LR.out.wr;
        status = LR.out.wr > LR.out.cap ? CST_STREAM_CAPACITY : CST_STREAM_OK;
        if (active) {
            cst_range_state r = a.rstate[s];
            r.lower = (uint64_t)LR.lower; r.range = (uint64_t)LR.range; r.inverted_n = LR.inv_n; r.inverted_first = LR.inv_first;
            a.rstate[s] = r;
        }
    } else {
        status = LR.finish(1u, n_words);
    }
    if (!active) return;
    if (bad) status = CST_STREAM_IMPOSSIBLE_SYMBOL;
    a.status[s] = status;
    a.n_words[s] = status == CST_STREAM_OK ? n_words : 0u;
}

// ------------------------------------------------------------------------------------------------
// decoding with per-symbol models
// ------------------------------------------------------------------------------------------------

struct PerSymbolDecodeArgs {
    const uint32_t* words;
    const uint64_t* offsets;
    size_t stride_words;
    const uint32_t* n_words;
    int32_t* symbols;
    size_t n_streams, n_per_stream;
    int32_t layout, precision;
    int32_t min_symbol, n_symbols;
    const double* means;        // Gaussian
    const double* stds;
    const uint32_t* cdf_rows;   // explicit rows
    uint64_t* state;            // ANS raw
    uint32_t* n_words_out;
    cst_range_state* rstate;    // range raw
    size_t row_stride;          // explicit rows: entries from one symbol's row to the next (0: one row for all)
    // chain coder: the remainders pushed, and the heads (n_words_out = what is left of the popped stack)
    uint32_t* push_words; size_t push_stride; uint32_t* n_push;
    cst_chain_heads* heads;
    int32_t* status;
    uint32_t flags;
    uint64_t words_capacity;    // uint32 slots behind `words` (0 = unknown): see word_slice
    __device__ __forceinline__ WordSlice slice(size_t s) const { return word_slice(offsets, stride_words, n_words, s, words_capacity); }
};

struct DecodeResume { uint64_t s0, s1, s2; uint32_t pos; int32_t status; };

// Uniform (per-wave or per-lane) coder front end reading words straight from HBM.
template <int W, int S, int KIND> struct DirectDecoder;

template <int W, int S>
struct DirectDecoder<W, S, kAns> {
    using st_t = typename StateT<S>::type;
    st_t state; uint32_t rd; const uint32_t* in; int32_t status;
    uint32_t ahead;                                           // in[rd - 1], requested when the word before it was taken
    __device__ __forceinline__ void init(const PerSymbolDecodeArgs& a, size_t s, bool raw) {
        const WordSlice ws = a.slice(s);
        in = a.words + ws.off;
        rd = ws.n; status = ws.bad ? (int32_t)CST_STREAM_INVALID_DATA : (int32_t)CST_STREAM_OK; state = 0; ahead = 0; idle = a.n_words + s;
        if (raw) { state = (st_t)a.state[s]; look_ahead(); return; }
        if (rd == 0) return;                                  // read_initial_state, stack.rs:440-462
        const uint32_t first = in[--rd];
        if (first == 0) { status = CST_STREAM_INVALID_DATA; rd = 0; return; }
        st_t st = first;
        while (rd > 0) { st = (st_t)((st << (W % S)) | (st_t)in[--rd]); if (st >= ((st_t)1 << (S - W))) break; }
        state = st;
        look_ahead();
    }
    // (an unconditional load from a pointer that is always valid: a conditional one makes the compiler wait for it at once)
    const uint32_t* idle;
    __device__ __forceinline__ void look_ahead() { ahead = *(rd > 0 ? in + (rd - 1) : idle); }
    // the lane-per-stream decoder keeps a window of the stream's words in LDS: where the window starts / which word is next
    static constexpr bool kDownward = true;
    __device__ __forceinline__ uint32_t position() const { return rd; }
    __device__ __forceinline__ int64_t next_index() const { return (int64_t)rd - 1; }
    __device__ __forceinline__ uint32_t length() const { return 0xffffffffu; }          // (every index below rd exists)
    __device__ __forceinline__ uint32_t quantile(int P) { return (uint32_t)state & ((1u << P) - 1u); }
    __device__ __forceinline__ void advance(uint32_t q, uint32_t c, uint32_t p, int P) {      // stack.rs:1086-1097
        st_t st = (st_t)((st_t)(state >> P) * (st_t)p + (st_t)(q - c));
        const bool refill = st < ((st_t)1 << (S - W)) && rd > 0;        // (the caller looks ahead again: once per symbol, outside
        state = refill ? (st_t)((st << (W % S)) | (st_t)ahead) : st;    //  any divergent branch)
        rd -= refill ? 1u : 0u;
    }
    __device__ __forceinline__ void finish(const PerSymbolDecodeArgs& a, size_t s, bool raw) {
        if (raw) { a.state[s] = (uint64_t)state; if (a.n_words_out) a.n_words_out[s] = rd; }
    }
    // a decoder parked between two launches over consecutive pieces of the same stream
    __device__ __forceinline__ void park(DecodeResume& r) const { r.s0 = (uint64_t)state; r.pos = rd; }
    __device__ __forceinline__ void resume(const PerSymbolDecodeArgs& a, size_t s, const DecodeResume& r) {
        in = a.words + a.slice(s).off;
        idle = a.n_words + s; status = CST_STREAM_OK; ahead = 0;
        state = (st_t)r.s0; rd = r.pos;
        look_ahead();
    }
};

template <int W, int S>
struct DirectDecoder<W, S, kRange> {
    using st_t = typename StateT<S>::type;
    RangeDecLane<W, S> L; uint32_t pos, len; const uint32_t* in; int32_t status;
    uint32_t ahead;                                           // in[pos], requested when the word before it was taken
    const uint32_t* idle;                                     // (see the ANS decoder)
    __device__ __forceinline__ void look_ahead() { ahead = *(pos < len ? in + pos : idle); }
    static constexpr bool kDownward = false;
    __device__ __forceinline__ uint32_t position() const { return pos; }
    __device__ __forceinline__ int64_t next_index() const { return (int64_t)pos; }
    __device__ __forceinline__ uint32_t length() const { return len; }
    __device__ __forceinline__ void init(const PerSymbolDecodeArgs& a, size_t s, bool raw) {
        const WordSlice ws = a.slice(s);
        in = a.words + ws.off;
        len = ws.n; pos = 0; L.status = ws.bad ? (int32_t)CST_STREAM_INVALID_DATA : (int32_t)CST_STREAM_OK; idle = a.n_words + s;
        L.lower = 0; L.range = (st_t)~(st_t)0;
        if (raw) {
            const cst_range_state r = a.rstate[s];
            L.lower = (st_t)r.lower; L.range = (st_t)r.range; L.point = (st_t)r.point; pos = (uint32_t)r.position;
        } else {                                              // read_point, queue.rs:847-868
            st_t pt = 0; int num_read = 0;
            while (pos < len) { pt = (st_t)((pt << (W % S)) | (st_t)in[pos++]); if (++num_read == S / W) break; }
            if (num_read < S / W && num_read != 0) pt = (st_t)(pt << (S - num_read * W));
            L.point = pt;
        }
        status = L.status; ahead = 0;
        look_ahead();
    }
    __device__ __forceinline__ uint32_t quantile(int P) { const uint32_t q = L.peek_quantile(P); status = L.status; return q; }
    __device__ __forceinline__ void advance(uint32_t, uint32_t c, uint32_t p, int P) {
        const bool have = pos < len;
        pos += L.advance(c, p, P, have ? ahead : 0u, have) ? 1u : 0u;       // (the caller looks ahead again)
    }
    __device__ __forceinline__ void finish(const PerSymbolDecodeArgs& a, size_t s, bool raw) {
        if (raw) {
            cst_range_state r = a.rstate[s];
            r.lower = (uint64_t)L.lower; r.range = (uint64_t)L.range; r.point = (uint64_t)L.point; r.position = pos;
            a.rstate[s] = r;
        }
    }
    __device__ __forceinline__ void park(DecodeResume& r) const {
        r.s0 = (uint64_t)L.lower; r.s1 = (uint64_t)L.range; r.s2 = (uint64_t)L.point; r.pos = pos;
    }
    __device__ __forceinline__ void resume(const PerSymbolDecodeArgs& a, size_t s, const DecodeResume& r) {
        const WordSlice ws = a.slice(s);
        in = a.words + ws.off;
        idle = a.n_words + s; len = ws.n; status = CST_STREAM_OK; L.status = CST_STREAM_OK; ahead = 0;
        L.lower = (st_t)r.s0; L.range = (st_t)r.s1; L.point = (st_t)r.s2; pos = r.pos;
        look_ahead();
    }
};

// ChainCoder::decode_symbol (src/stream/chain.rs:1044-1122): P bits per symbol come off `compressed` whatever the model;
// what the symbol did not use goes onto `remainders` (flush_remainders_head, :784-796).
template <int W, int S>
struct DirectDecoder<W, S, kChain> {
    using st_t = typename StateT<S>::type;
    static constexpr uint32_t wmask = W == 32 ? 0xffffffffu : ((1u << (W % 32)) - 1u);
    st_t rh; uint32_t ch, rd, wr, cap; const uint32_t* in; uint32_t* out; int32_t status;
    uint32_t ahead; const uint32_t* idle;
    __device__ __forceinline__ void look_ahead() { ahead = *(rd > 0 ? in + (rd - 1) : idle); }
    static constexpr bool kDownward = true;
    __device__ __forceinline__ uint32_t position() const { return rd; }
    __device__ __forceinline__ int64_t next_index() const { return (int64_t)rd - 1; }
    __device__ __forceinline__ uint32_t length() const { return 0xffffffffu; }
    __device__ __forceinline__ void init(const PerSymbolDecodeArgs& a, size_t s, bool) {
        const WordSlice ws = a.slice(s);
        in = a.words + ws.off;
        rd = ws.n; idle = a.n_words + s; status = ws.bad ? (int32_t)CST_STREAM_INVALID_DATA : (int32_t)CST_STREAM_OK;
        const cst_chain_heads h = a.heads[s];
        rh = (st_t)h.remainders_head; ch = h.compressed_head;
        out = a.push_words + s * a.push_stride;
        cap = (uint32_t)(a.push_stride > 0xffffffffull ? 0xffffffffull : a.push_stride); wr = 0;
        look_ahead();
    }
    __device__ __forceinline__ uint32_t quantile(int P) {
        uint32_t word;
        if (P == W || ch < (1u << P)) {
            if (rd == 0) { status = CST_STREAM_OUT_OF_DATA; return 0u; }
            word = ahead & wmask; --rd;                       // (the caller looks ahead again after advance())
            if (P != W) ch = ((ch << (W - P)) | (word >> P)) & wmask;
        } else {
            word = ch; ch >>= P;
        }
        return P == W ? word : (word & ((1u << P) - 1u));
    }
    __device__ __forceinline__ void advance(uint32_t q, uint32_t c, uint32_t p, int P) {
        rh = (st_t)(rh * (st_t)p + (st_t)(q - c));
        if (rh >= ((st_t)1 << (S - P))) {
            if (wr < cap) out[wr] = (uint32_t)rh & wmask;
            ++wr;
            rh = (st_t)(rh >> (W % S));
        }
    }
    __device__ __forceinline__ void finish(const PerSymbolDecodeArgs& a, size_t s, bool) {
        cst_chain_heads h; h.remainders_head = (uint64_t)rh; h.compressed_head = ch; h.reserved = 0;
        a.heads[s] = h;
        a.n_words_out[s] = rd; a.n_push[s] = wr;
        if (wr > cap && a.status[s] == CST_STREAM_OK) a.status[s] = CST_STREAM_CAPACITY;
    }
};

// One WAVE per stream.  Every lane carries the same coder state; the search for quantile_function (semantics of
// quantize.rs:580-779 / lookup_contiguous.rs:564-605: the unique symbol with left(sym) <= q < left(sym+1)) evaluates up
// to 64 candidate left cumulatives per round: two rounds for a 201-symbol support.  MODEL supplies left(element, i).
struct GaussianLeft {
    static constexpr int32_t kBadModel = CST_STREAM_IMPOSSIBLE_SYMBOL;   // degenerate distribution (quantize.rs:562-565)
    const PerSymbolDecodeArgs& a;
    const double2* erf_tab;
    double mu, sd;
    __device__ __forceinline__ bool load(size_t e) {
        mu = a.means[e]; sd = a.stds[e];
        // the reference panics on an invalid model (pybindings/stream/model.rs:654-657)
        return sd > 0.0 && sd <= 1.7976931348623157e308 && mu == mu && mu <= 1.7976931348623157e308 && mu >= -1.7976931348623157e308;
    }
    __device__ __forceinline__ uint32_t left(uint32_t i) const {
        return leaky_gaussian_left_quick((int32_t)i, a.min_symbol, a.n_symbols, a.precision, 32, mu, sd, erf_tab);
    }
};
struct RowLeft {                       // explicit cdf rows [n + 1] per coded symbol: 64 coalesced entries per round
    static constexpr int32_t kBadModel = CST_STREAM_INVALID_DATA;          // a row that is not a cdf for this quantile
    const PerSymbolDecodeArgs& a;
    const double2* unused;
    const uint32_t* row;
    __device__ __forceinline__ bool load(size_t e) { row = a.cdf_rows + e * a.row_stride; return true; }
    __device__ __forceinline__ uint32_t left(uint32_t i) const { return row[i]; }
};

template <int W, int S, int KIND, class MODEL>
__global__ __launch_bounds__(kBlock) void decode_wave_kernel(const PerSymbolDecodeArgs a) {
    __shared__ double2 erf_tab[kErfTabEntries];
    erf_tab_fill(erf_tab, threadIdx.x, blockDim.x);
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1);
    const size_t s = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (s >= a.n_streams) return;
    const size_t N = a.n_per_stream;
    const int P = a.precision;
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;
    const uint32_t n = (uint32_t)a.n_symbols;
    const size_t stride_t = a.layout == CST_LAYOUT_SYMBOL_MAJOR ? a.n_streams : 1;
    const size_t e0 = a.layout == CST_LAYOUT_SYMBOL_MAJOR ? s : s * N;

    DirectDecoder<W, S, KIND> D;
    D.init(a, s, raw);
    MODEL M{a, erf_tab};
    int32_t status = D.status;
    for (size_t t = 0; t < N && status == CST_STREAM_OK; ++t) {
        if (!M.load(e0 + t * stride_t)) { status = CST_STREAM_IMPOSSIBLE_SYMBOL; break; }
        const uint32_t q = D.quantile(P);
        if (D.status != CST_STREAM_OK) { status = D.status; break; }
        uint32_t base = 0, count = n;                 // invariant: left(base) <= q
        while (count > 63) {
            const uint32_t stride = (count + 63) / 64;
            const uint32_t off = (uint32_t)lane * stride;
            const bool valid = off < count;
            const uint32_t val = valid ? M.left(base + off) : 0u;
            const unsigned long long m = __ballot(valid && val <= q);
            const uint32_t k = max((uint32_t)__popcll(m), 1u);   // (lane 0 always qualifies for a valid table)
            const uint32_t adv = (k - 1) * stride;
            base += adv;
            count = min(stride, count - adv);
        }
        const uint32_t val = ((uint32_t)lane <= count) ? M.left(base + lane) : 0u;
        const unsigned long long m = __ballot((uint32_t)lane < count && val <= q);
        const uint32_t k = (uint32_t)__popcll(m);
        const uint32_t c = __shfl(val, (int)(k > 0 ? k - 1 : 0), 64), nxt = __shfl(val, (int)min(k, 63u), 64);
        const uint32_t p = nxt - c;
        if (p == 0 || k == 0 || c > q || (uint64_t)c + p > ((uint64_t)1 << P)) { status = MODEL::kBadModel; break; }
        if (lane == 0) a.symbols[e0 + t * stride_t] = a.min_symbol + (int32_t)(base + k - 1);
        D.advance(q, c, p, P);
        D.look_ahead();
    }
    if (lane == 0) {
        a.status[s] = status;
        D.finish(a, s, raw);
    }
}

// One LANE per stream (batches of at least a wave of streams): the search starts from an inverse-normal guess of the
// symbol and brackets the quantile with EXACT left cumulatives -- typically two erf evaluations per symbol instead of
// the 64 to 128 of a wave-wide search.  Any search returns the same symbol: the one with left(sym) <= q < left(sym + 1)
// (quantize.rs:580-779).

// Acklam's rational approximation of the inverse normal CDF in f32 with the hardware's approximate log, sqrt and
// reciprocal: a STARTING POINT for the search (a guess that is off costs probes, never correctness).
// `tail` = min(p, 1 - p) in (0, 0.5]; returns the (negative) quantile of the lower tail.
__device__ __forceinline__ float ndtri_lower_f32(float tail) {
    constexpr float a1 = -3.969683028665376e+01f, a2 = 2.209460984245205e+02f, a3 = -2.759285104469687e+02f, a4 = 1.383577518672690e+02f,
        a5 = -3.066479806614716e+01f, a6 = 2.506628277459239e+00f, b1 = -5.447609879822406e+01f, b2 = 1.615858368580409e+02f,
        b3 = -1.556989798598866e+02f, b4 = 6.680131188771972e+01f, b5 = -1.328068155288572e+01f, c1 = -7.784894002430293e-03f,
        c2 = -3.223964580411365e-01f, c3 = -2.400758277161838e+00f, c4 = -2.549732539343734e+00f, c5 = 4.374664141464968e+00f,
        c6 = 2.938163982698783e+00f, d1 = 7.784695709041462e-03f, d2 = 3.224671290700398e-01f, d3 = 2.445134137142996e+00f,
        d4 = 3.754408661907416e+00f;
    // both branches, then a select: cheaper than diverging over 25 instructions
    // (explicit fused multiply-adds: the library is built with -ffp-contract=off for its bit-exact f64 paths, and a guess
    // has no bits to keep)
    auto f = [](float a, float b, float c) { return __builtin_fmaf(a, b, c); };
    const float q = __builtin_amdgcn_sqrtf(-1.3862943611f * __builtin_amdgcn_logf(tail));          // sqrt(-2 ln(tail))
    const float zt = f(f(f(f(f(c1, q, c2), q, c3), q, c4), q, c5), q, c6) * __builtin_amdgcn_rcpf(f(f(f(f(d1, q, d2), q, d3), q, d4), q, 1.0f));
    const float u = tail - 0.5f, r = u * u;
    const float zc = f(f(f(f(f(a1, r, a2), r, a3), r, a4), r, a5), r, a6) * u * __builtin_amdgcn_rcpf(f(f(f(f(f(b1, r, b2), r, b3), r, b4), r, b5), r, 1.0f));
    return tail < 0.02425f ? zt : zc;
}

// The same quantile from Abramowitz & Stegun 26.2.23 (|error| < 4.5e-4 over the whole lower half): a third of the instructions.
// Good for a first probe as long as 4.5e-4 sigma stays well below half a symbol; the lane decoder uses it when no lane of the
// wave has sigma >= 200.
__device__ __forceinline__ float ndtri_lower_coarse_f32(float tail) {
    const float t = __builtin_amdgcn_sqrtf(-1.3862943611f * __builtin_amdgcn_logf(tail));          // sqrt(-2 ln(tail))
    const float num = __builtin_fmaf(__builtin_fmaf(0.010328f, t, 0.802853f), t, 2.515517f);
    const float den = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(0.001308f, t, 0.189269f), t, 1.432788f), t, 1.0f);
    return __builtin_fmaf(num, __builtin_amdgcn_rcpf(den), -t);
}

// Out of line ON PURPOSE: inlined, the compiler hoists the 32 + 8 row addresses of both variants out of the symbol loop
// into 130 VGPRs and then spills the erf's registers around every call of it.
__device__ __noinline__ void store_symbol_tile(int32_t* sym, size_t n_streams, size_t N, size_t s0, size_t t0, int lane, const int32_t* tile, bool vec) {
    if (vec) tile_store<true>(sym, n_streams, N, s0, t0, lane, tile);
    else tile_store<false>(sym, n_streams, N, s0, t0, lane, tile);
}

// LDS of the lane-per-stream decoder, per wave: the symbol tile (stream-major output), one tile of parameters
// [kParTile][64 streams (+1)] for each of mean and std, and a window of kWordWindow words per stream.  Everything that comes
// from HBM is requested ONE TILE (16 symbols ~ 40 000 cycles of model search) before it is used, with coalesced loads
// where the layout allows: per-lane loads issued a symbol ahead exposed ~2300 cycles of latency per symbol (half the
// kernel's time: rocprofv3 SQ_WAIT_ANY), because a wave-wide load of 64 different cache lines takes longer than a symbol.
constexpr int kParStride = kWave + 1;                 // doubles per tile row: conflict-free writes (stream-major) and reads
// Two geometries (round 5).  BIG: parameter tiles of 16 symbols, a 32-slot word window, a 32-symbol output tile -- 34 KiB of LDS per
// wave, four waves per CU: right while a batch has one wave per SIMD anyway (65 536 streams).  SMALL: tiles of 8, a 16-slot window, a
// 16-symbol output tile (rows of 20 words) -- 17 KiB per wave, EIGHT waves per workgroup and CU: with more than one wave of streams
// per SIMD (more than 65 536 streams: e.g. a batch decoded through jump points) the second wave covers what a lone wave waits for
// (690 of its 1970 cycles per symbol, profiles/r04_sq_counters.md).
template <bool SMALL> struct LaneGeo {
    static constexpr int kParTile = SMALL ? 8 : 16;
    static constexpr int kWordWindow = 2 * kParTile;          // slots per stream, position p lives in slot p % kWordWindow
    static constexpr int kOutSyms = SMALL ? 16 : kTileSyms;   // symbols per output tile
    static constexpr int kOutStride = SMALL ? 20 : kTileStride;
    static constexpr int kThreads = SMALL ? 512 : kBlock;
    static constexpr size_t kWaveBytes = (size_t)kWave * kOutStride * 4 + 2 * (size_t)kParTile * kParStride * 8 + (size_t)kWordWindow * kWave * 4;
    static constexpr size_t kLdsBytes = kErfTabBytes + (size_t)(kThreads / kWave) * kWaveBytes;
};

// 16-symbol output tile of the SMALL geometry -> HBM: piece (lane & 3) of rows (lane >> 2) + 16 k, 64-byte row segments
__device__ __noinline__ void store_symbol_tile16(int32_t* sym, size_t n_streams, size_t N, size_t s0, size_t t0, int lane, const int32_t* tile, bool vec) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t r = (size_t)(lane >> 2) + 16 * k;
        if (s0 + r >= n_streams) continue;
        const int32_t* src = tile + r * 20 + 4 * (lane & 3);
        int32_t* dst = sym + (s0 + r) * N + t0 + 4 * (lane & 3);
        if (vec) {
            const int4 v = *reinterpret_cast<const int4*>(src);
            v4i t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
            __builtin_nontemporal_store(t, reinterpret_cast<v4i*>(dst));
        } else {
            dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
        }
    }
}

template <int W, int S, int KIND, bool SMALL = false>
__global__ __launch_bounds__(LaneGeo<SMALL>::kThreads) void decode_gaussian_lane_kernel(const PerSymbolDecodeArgs a) {
    using G = LaneGeo<SMALL>;
    constexpr int kParTile = G::kParTile, kWordWindow = G::kWordWindow, kOutSyms = G::kOutSyms, kOutStride = G::kOutStride;
    constexpr size_t kLaneDecWaveBytes = G::kWaveBytes;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double2* erf_tab = reinterpret_cast<double2*>(smem);
    erf_tab_fill(erf_tab, threadIdx.x, blockDim.x);
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1);
    unsigned char* mine = smem + kErfTabBytes + (size_t)(threadIdx.x >> 6) * kLaneDecWaveBytes;
    int32_t* tile = reinterpret_cast<int32_t*>(mine);
    double* par_mu = reinterpret_cast<double*>(mine + (size_t)kWave * kOutStride * 4);
    double* par_sd = par_mu + kParTile * kParStride;
    uint32_t* win = reinterpret_cast<uint32_t*>(par_sd + kParTile * kParStride);
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t s0 = s - lane;
    if (s0 >= a.n_streams) return;
    const bool active = s < a.n_streams;
    const size_t se = active ? s : a.n_streams - 1;          // idle lanes of a partial wave repeat its last stream
    const size_t N = a.n_per_stream;
    const int P = a.precision;
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;
    const uint32_t n = (uint32_t)a.n_symbols;
    const bool symbol_major = a.layout == CST_LAYOUT_SYMBOL_MAJOR;
    const size_t stride_t = symbol_major ? a.n_streams : 1;
    const size_t e0 = symbol_major ? se : se * N;
    const double free_weight = (double)((P >= 32 ? 0xffffffffu : ((1u << P) - 1u)) - (n - 1u));
    const float total_f = (float)(1ull << P), free_f = (float)free_weight, inv_total_f = 1.0f / total_f, inv_free_f = 1.0f / free_f;
    const double guess_shift = 0.5 - (double)a.min_symbol;            // symbol index of the real number x: x - min_symbol + 0.5
    const bool vec = !symbol_major && (N % 4 == 0) && (reinterpret_cast<uintptr_t>(a.symbols) & 15) == 0;
    const bool two_step_guess = (double)n * 64.0 > free_weight;      // the leak moves the guess by more than 1/64 quantile

    DirectDecoder<W, S, KIND> D;
    D.init(a, se, raw);
    int32_t status = D.status;
    int32_t* sym_p = a.symbols + e0;

    // ---- parameter tiles: item w = it * 64 + lane of a tile is (stream j, symbol tl), consecutive lanes on consecutive
    // addresses in either layout ----
    // SMALL geometry, stream-major: a tile is 8 symbols = HALF a 128-byte line of doubles per stream.  Requested tile by tile, every
    // line of means and stds came in twice, ~16 000 cycles apart (round 5's counters: 10.24 GB per launch against 5.49 GB
    // algorithmic).  So a request there covers a PAIR of tiles -- whole lines, the big geometry's item mapping -- and the second
    // tile's half waits in the registers it arrived in: landed when its tile begins (mode 1 -> 2 -> 0 below).
    constexpr int kReqTile = 16;                              // symbols a request can cover (= kParTile of the big geometry)
    double mu_r[kReqTile], sd_r[kReqTile];
    int mode = 0;                                             // (wave-uniform) 0: the registers hold ONE tile (or nothing); 1 / 2: a pair, its first / second half lands next
    // item `it` of a tile that lies wholly inside the matrix: element base_e + t0 * t_stride + it * item_stride (stream-major:
    // stream s0 + lane / 16 + 4 it, symbol t0 + lane % 16; symbol-major: symbol t0 + it, stream s0 + lane), LDS slot
    // base_l + it * l_stride -- pointer increments; tiles that stick out (last streams, last symbols) take the general form
    const size_t base_e = symbol_major ? s0 + (size_t)lane : (s0 + (size_t)(lane / kParTile)) * N + (size_t)(lane % kParTile);
    const size_t t_stride = symbol_major ? a.n_streams : 1, item_stride = symbol_major ? a.n_streams : (size_t)(kWave / kParTile) * N;
    const int base_l = symbol_major ? lane : (lane % kParTile) * kParStride + lane / kParTile;
    const int l_stride = symbol_major ? kParStride : kWave / kParTile;
    const bool pairs = SMALL && !symbol_major && s0 + kWave <= a.n_streams;
    auto par_request = [&](size_t t0) {
        if (SMALL && pairs && t0 % kReqTile == 0 && t0 + kReqTile <= N) {
            const size_t e = (s0 + (size_t)(lane / kReqTile)) * N + (size_t)(lane % kReqTile) + t0;
            const double* pm = a.means + e;
            const double* ps = a.stds + e;
#pragma unroll
            for (int it = 0; it < kReqTile; ++it) {
                mu_r[it] = __builtin_nontemporal_load(pm + (size_t)it * (size_t)(kWave / kReqTile) * N);
                sd_r[it] = __builtin_nontemporal_load(ps + (size_t)it * (size_t)(kWave / kReqTile) * N);
            }
            mode = 1;
            return;
        }
        mode = 0;
        if (s0 + kWave <= a.n_streams && t0 + kParTile <= N) {
            const double* pm = a.means + base_e + t0 * t_stride;
            const double* ps = a.stds + base_e + t0 * t_stride;
#pragma unroll
            for (int it = 0; it < kParTile; ++it) {
                mu_r[it] = __builtin_nontemporal_load(pm + (size_t)it * item_stride);
                sd_r[it] = __builtin_nontemporal_load(ps + (size_t)it * item_stride);
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < kParTile; ++it) {
            const int w = it * kWave + lane;
            const size_t j = symbol_major ? w % kWave : w / kParTile, tl = symbol_major ? w / kWave : w % kParTile;
            // (idle lanes of a partial wave repeat its last stream -- with that stream's parameters: the chain coder's lanes
            // write their remainders as they go)
            const size_t sj = s0 + j < a.n_streams ? s0 + j : a.n_streams - 1;
            const size_t e = t0 + tl < N ? (symbol_major ? (t0 + tl) * a.n_streams + sj : sj * N + t0 + tl) : 0;
            mu_r[it] = __builtin_nontemporal_load(a.means + e);
            sd_r[it] = __builtin_nontemporal_load(a.stds + e);
        }
    };
    auto par_land = [&]() {
        if (SMALL && mode != 0) {
            // half (mode - 1) of the pair: the lanes that hold symbols 8 (mode - 1) .. + 7 of the sixteen write theirs
            const int sub = (lane % kReqTile) - kParTile * (mode - 1);
            if (sub >= 0 && sub < kParTile) {
#pragma unroll
                for (int it = 0; it < kReqTile; ++it) {
                    par_mu[sub * kParStride + lane / kReqTile + (kWave / kReqTile) * it] = mu_r[it];
                    par_sd[sub * kParStride + lane / kReqTile + (kWave / kReqTile) * it] = sd_r[it];
                }
            }
            mode = mode == 1 ? 2 : 0;
            return;
        }
#pragma unroll
        for (int it = 0; it < kParTile; ++it) {
            par_mu[base_l + it * l_stride] = mu_r[it];
            par_sd[base_l + it * l_stride] = sd_r[it];
        }
    };
    // ---- word window: a tile of 16 symbols takes at most 16 words, so with the 16 words behind the read position in
    // LDS at the start of a tile the NEXT 16 can be on their way during it ----
    uint32_t w_r[kParTile];
    int64_t w_first = 0;                                      // index of w_r[0]
    auto win_request = [&](int64_t first) {
        w_first = first;
        const int64_t len = (int64_t)D.length();
        if (!__any(first < 0 || first + kParTile > len)) {    // every lane's 16 words exist: one pointer, sixteen offsets
            const uint32_t* pw = D.in + first;
#pragma unroll
            for (int i = 0; i < kParTile; ++i) w_r[i] = pw[i];
            return;
        }
#pragma unroll
        for (int i = 0; i < kParTile; ++i) {
            const int64_t p = first + i;
            w_r[i] = *(p >= 0 && p < len ? D.in + p : D.idle);
        }
    };
    auto win_land = [&]() {
#pragma unroll
        for (int i = 0; i < kParTile; ++i) win[(((uint32_t)(w_first + i)) & (kWordWindow - 1)) * kWave + lane] = w_r[i];
    };
    const auto window_of = [&](int ahead_tiles) -> int64_t {  // first index of the 16 words `ahead_tiles` tiles ahead
        return D.kDownward ? (int64_t)D.position() - (int64_t)kParTile * (ahead_tiles + 1) : (int64_t)D.position() + (int64_t)kParTile * ahead_tiles;
    };

    if (N > 0) {
        par_request(0);
        win_request(window_of(0));
    }
    for (size_t t0 = 0; t0 < N; t0 += kParTile) {
        wave_lds_fence();                                     // (the previous tile's parameters have been read)
        par_land();
        win_land();
        if (t0 + kParTile < N && mode != 2) par_request(t0 + kParTile);      // (mode 2: the next tile's parameters are here already)
        win_request(window_of(1));                            // (the words one tile further: used from the next tile on)
        wave_lds_fence();
        const int n_here = (int)(N - t0 < (size_t)kParTile ? N - t0 : (size_t)kParTile);
#pragma unroll 1
        for (int tl = 0; tl < n_here; ++tl) {
            const size_t t = t0 + (size_t)tl;
            int32_t sym = 0;
            const double mu = par_mu[tl * kParStride + lane], sd = par_sd[tl * kParStride + lane];
            D.ahead = win[(((uint32_t)D.next_index()) & (kWordWindow - 1)) * kWave + lane];
            if (status == CST_STREAM_OK) {
                // the reference panics on an invalid model (pybindings/stream/model.rs:654-657)
                const bool model_ok = sd > 0.0 && sd <= 1.7976931348623157e308 && fabs(mu) <= 1.7976931348623157e308;
                const uint32_t q = model_ok ? D.quantile(P) : 0u;
                if (!model_ok) status = CST_STREAM_IMPOSSIBLE_SYMBOL;
                else if (D.status != CST_STREAM_OK) status = D.status;
                else {
                    // guess: ignore the leak (one quantile per symbol) first, then account for the guessed symbol's share of it
                    const float below = (float)q + 0.5f, above = total_f - below;
                    const bool coarse = !__any(sd >= 200.0);          // (wave-uniform: the cheap quantile is good enough)
                    float z = coarse ? ndtri_lower_coarse_f32(fminf(below, above) * inv_total_f) : ndtri_lower_f32(fminf(below, above) * inv_total_f);
                    double x = mu + sd * (double)(below < above ? z : -z) + guess_shift;
                    if (two_step_guess) {
                        const float b1 = below - (float)fmin(fmax(x, 0.0), (double)(n - 1u)), a1 = free_f - b1;
                        z = ndtri_lower_f32(fmaxf(fminf(b1, a1), 0.25f) * inv_free_f);
                        x = mu + sd * (double)(b1 < a1 ? z : -z) + guess_shift;
                    }
                    // first probe: the bin BOUNDARY nearest to the guess (boundary i lies at x = i) -- whichever side of it the
                    // quantile falls, the second probe closes the bracket as long as the guess is within half a symbol; probing
                    // floor(x) first cost a third evaluation whenever the guess was a hair low, and a wave runs as many
                    // evaluations as its worst lane
                    const uint32_t g = (uint32_t)fmin(fmax(x + 0.5, 1.0), (double)(n - 1u));
                    // bracket [lo_i, hi_i): left(lo_i) = lo_v <= q < hi_v = left(hi_i)
                    uint32_t lo_i = 0, hi_i = n, lo_v = 0, hi_v = P >= 32 ? 0u : (1u << P);
                    uint32_t probe = g, step = 1;
                    bool up = false, down = false;
                    {
                        // ... and its two neighbours in the same breath: the three evaluations share their LDS latency, and a
                        // guess within half a symbol -- almost every one -- is bracketed without a second look
                        uint32_t v3[3];
                        leaky_gaussian_left3_quick(g, a.min_symbol, n, P, mu, sd, erf_tab, v3);
                        if (v3[1] <= q) {
                            up = true;
                            if (q < v3[2]) { lo_i = g; lo_v = v3[1]; hi_i = g + 1u; hi_v = v3[2]; }
                            else { lo_i = g + 1u; lo_v = v3[2]; probe = min(g + 2u, n - 1u); step = 2; }
                        } else {
                            down = true;
                            if (v3[0] <= q) { lo_i = g - 1u; lo_v = v3[0]; hi_i = g; hi_v = v3[1]; }
                            else { hi_i = g - 1u; hi_v = v3[0]; probe = max(g - 1u, 2u) - 1u; step = 2; }
                        }
                    }
                    while (hi_i - lo_i > 1) {
                        // (every probe lies strictly inside (lo_i, hi_i), a subset of (0, n))
                        const uint32_t v = leaky_gaussian_left_quick<true>((int32_t)probe, a.min_symbol, (int32_t)n, P, 32, mu, sd, erf_tab);
                        if (v <= q) { lo_i = probe; lo_v = v; up = true; } else { hi_i = probe; hi_v = v; down = true; }
                        if (up && down) probe = lo_i + (hi_i - lo_i) / 2;
                        else if (up) probe = min(lo_i + step, hi_i - 1u);
                        else probe = max(hi_i - min(step, hi_i - 1u), lo_i + 1u);
                        step *= 2;
                    }
                    const uint32_t c = lo_v, p = hi_v - lo_v;
                    if (p == 0 || c > q || (uint64_t)c + p > ((uint64_t)1 << P)) status = CST_STREAM_IMPOSSIBLE_SYMBOL;   // degenerate distribution (quantize.rs:562-565)
                    else {
                        sym = a.min_symbol + (int32_t)lo_i;
                        D.advance(q, c, p, P);
                    }
                }
            }
            if (symbol_major) {
                if (active) *sym_p = sym;
                sym_p += stride_t;
            } else {
                tile[lane * kOutStride + (t % kOutSyms)] = sym;
                if (t % kOutSyms == kOutSyms - 1) {
                    wave_lds_fence();
                    if constexpr (SMALL) store_symbol_tile16(a.symbols, a.n_streams, N, s0, t - (kOutSyms - 1), lane, tile, vec);
                    else store_symbol_tile(a.symbols, a.n_streams, N, s0, t - (kOutSyms - 1), lane, tile, vec);
                    wave_lds_fence();
                }
            }
        }
    }
    if (!symbol_major) {
        const size_t done = N - N % kOutSyms;
        if (active) for (size_t t = done; t < N; ++t) a.symbols[se * N + t] = tile[lane * kOutStride + (t % kOutSyms)];
    }
    if (!active) return;
    a.status[s] = status;
    D.finish(a, s, raw);
}

// FEWER streams than a wave has lanes -- down to the reference's own usage, ONE coder and a long message.  Decoding a
// stream is sequential, and with the model search inside the chain every symbol costs an erf latency (1.7 us).  But
// only the QUANTILE depends on the coder state, the models do not: a first kernel tabulates every symbol's whole cdf row
// at full occupancy (256 erf per symbol: 2.4 ms per million symbols), and the sequential kernel is left with a lookup.
// A row is 256 left cumulatives (supports up to 255 symbols, padded with 2^P); lane l of the stream's wave holds
// entries 4l..4l+3 of the current row in registers, straight from one coalesced 1-KiB load issued eight symbols
// earlier.  One ballot finds the lane, three compares the entry: ~45 instructions per symbol instead of ~600.
constexpr int kRowEntries = 256;
constexpr int kRowsAhead = 16;

__global__ __launch_bounds__(kRowEntries) void gaussian_rows_kernel(int P, int32_t lo, int32_t n, const double* __restrict__ means,
                                                                   const double* __restrict__ stds, int32_t layout, size_t n_streams,
                                                                   size_t N, size_t t0, size_t count, uint32_t* __restrict__ rows) {
    __shared__ double2 erf_tab[kErfTabEntries];
    erf_tab_fill(erf_tab, threadIdx.x, blockDim.x);
    __syncthreads();
    const size_t row = blockIdx.x;                     // = stream * count + (t - t0)
    const size_t s = row / count, t = t0 + row % count;
    const size_t e = layout == CST_LAYOUT_SYMBOL_MAJOR ? t * n_streams + s : s * N + t;
    const double mu = means[e], sd = stds[e];
    const int32_t i = (int32_t)threadIdx.x;
    const uint32_t total = 1u << P;
    uint32_t v;
    if (sd > 0.0 && sd <= 1.7976931348623157e308 && fabs(mu) <= 1.7976931348623157e308)
        v = i <= n ? leaky_gaussian_left_quick(i, lo, n, P, 32, mu, sd, erf_tab) : total;
    else v = i == 0 ? 0xffffffffu : total;             // invalid model (a valid row starts with 0)
    rows[row * kRowEntries + i] = v;
}

struct RowsDecodeArgs {
    PerSymbolDecodeArgs a;
    const uint4* rows;          // [stream][count][64 lanes] x 4 entries
    size_t t0, count;
    DecodeResume* resume;       // [stream]
    int32_t first, last;
};

// The row queue and the word blocks below are plain loads: the compiler's own counter bookkeeping waits, at the use of a
// row, for exactly the loads older than the fifteen youngest -- as long as NO other load sits in the loop (a per-symbol
// look-ahead of the next compressed word, waited for every symbol, drags the whole in-order queue with it and collapses
// the 16-deep queue to depth one; measured 340 ns per symbol).  Hand-issued loads in inline asm are not an option in a
// C++ loop: the compiler copies loop-carried registers at the bottom of the loop, in flight or not.
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

// The stream's compressed words, 64 at a time in one register across the wave (lane l: the l-th word from the block's
// origin in reading direction).  A symbol consumes at most one word, so a block fetched at the start of a 64-symbol
// group lasts for the whole group.
template <int KIND>
struct WordBlock {
    uint32_t cur;
    uint32_t origin;                    // read position the block starts at (ANS: counts down, range: up)
    __device__ __forceinline__ void fetch(const uint32_t* in, uint32_t len, uint32_t position, int lane) {
        origin = position;
        const uint32_t i = KIND == kAns ? position - 1u - (uint32_t)lane : position + (uint32_t)lane;      // (ANS: wraps below 0)
        cur = in[i < len ? i : 0u];
    }
    // the word at read position `position` (ANS: the next word to pop is in[position - 1])
    __device__ __forceinline__ uint32_t at(uint32_t position) const {
        const uint32_t off = KIND == kAns ? origin - position : position - origin;
        return (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)(off & 63u));
    }
};

// one symbol of decode_rows_wave_kernel: `r` = this lane's four entries of the symbol's row
template <int W, int S, int KIND>
__device__ __forceinline__ void rows_step(DirectDecoder<W, S, KIND>& D, const WordBlock<KIND>& words, const v4u r, uint32_t slot, int lane,
                                          uint32_t n, int32_t min_symbol, int P, int32_t& status, int32_t& mine) {
    const uint32_t q = D.quantile(P);
    const bool le = r.x <= q;
    const unsigned long long m = __ballot(le);
    const uint32_t cnt = (le ? 1u : 0u) + (r.y <= q ? 1u : 0u) + (r.z <= q ? 1u : 0u) + (r.w <= q ? 1u : 0u);
    const uint32_t my_c = cnt >= 4 ? r.w : cnt == 3 ? r.z : cnt == 2 ? r.y : r.x;
    const uint32_t my_n = cnt == 3 ? r.w : cnt == 2 ? r.z : r.y;
    const uint32_t k = ((uint32_t)__popcll(m) - 1u) & 63u;   // the last lane whose first entry is <= q
    const uint32_t ck = (uint32_t)__builtin_amdgcn_readlane((int)cnt, (int)k);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)my_c, (int)k);
    const uint32_t n_in = (uint32_t)__builtin_amdgcn_readlane((int)my_n, (int)k);
    const uint32_t n_out = (uint32_t)__builtin_amdgcn_readlane((int)r.x, (int)min(k + 1u, 63u));
    const uint32_t nxt = ck >= 4 ? n_out : n_in;
    const uint32_t idx = 4u * k + ck - 1u, p = nxt - c;
    // an invalid model marks entry 0 (then no lane's first entry is <= q, or lane 0's is not); a degenerate distribution
    // (quantize.rs:562-565) has p == 0; a quantile beyond the support cannot happen for a valid row
    const bool bad = (m & 1) == 0 || idx >= n || p == 0;
    const int32_t fail = D.status != CST_STREAM_OK ? D.status : bad ? CST_STREAM_IMPOSSIBLE_SYMBOL : CST_STREAM_OK;
    if (status == CST_STREAM_OK) {
        status = fail;
        if (fail == CST_STREAM_OK) {
            if ((uint32_t)lane == slot) mine = min_symbol + (int32_t)idx;
            if constexpr (KIND == kAns) D.ahead = words.at(D.rd);
            else D.ahead = words.at(D.pos);
            D.advance(q, c, p, P);
        }
    }
}

template <int W, int S, int KIND>
__global__ __launch_bounds__(kWave) void decode_rows_wave_kernel(const RowsDecodeArgs ra) {
    const PerSymbolDecodeArgs& a = ra.a;
    const int lane = threadIdx.x;
    const size_t s = blockIdx.x;
    const size_t N = a.n_per_stream;
    const int P = a.precision;
    const bool raw = (a.flags & CST_FLAG_RAW_STATE) != 0;
    const uint32_t n = (uint32_t)a.n_symbols;
    const size_t stride_t = a.layout == CST_LAYOUT_SYMBOL_MAJOR ? a.n_streams : 1;
    int32_t* out = a.symbols + (a.layout == CST_LAYOUT_SYMBOL_MAJOR ? s : s * N) + ra.t0 * stride_t;
    const v4u* rows = reinterpret_cast<const v4u*>(ra.rows) + s * ra.count * kWave + lane;
    const size_t count = ra.count;

    DirectDecoder<W, S, KIND> D;
    int32_t status;
    if (ra.first) { D.init(a, s, raw); status = D.status; }
    else { const DecodeResume r = ra.resume[s]; D.resume(a, s, r); status = r.status; }
    WordBlock<KIND> words;
    const uint32_t n_words = max(a.slice(s).n, 1u);
    auto fetch_words = [&]() {
        if constexpr (KIND == kAns) words.fetch(D.in, n_words, D.rd, lane);
        else words.fetch(D.in, n_words, D.pos, lane);
        // waited for HERE, once per 64 symbols: left pending, the compiler waits at every use with a count that drains
        // the row queue to two
        asm volatile("" : "+v"(words.cur));
    };

    v4u ahead[kRowsAhead];
#pragma unroll
    for (int j = 0; j < kRowsAhead; ++j) ahead[j] = __builtin_nontemporal_load(rows + min((size_t)j, count - 1) * kWave);
    int32_t mine = 0;                                  // lane l keeps the symbol of position 64 g + l until the group is stored
    size_t tl = 0;
    // whole blocks whose rows ahead all exist: no bounds checks
    for (; tl + 2 * kRowsAhead <= count && status == CST_STREAM_OK; tl += kRowsAhead) {
        if ((tl & 63) == 0) fetch_words();
        const uint32_t slot0 = (uint32_t)(tl & 63);
        const v4u* next = rows + (tl + kRowsAhead) * kWave;
#pragma unroll
        for (int j = 0; j < kRowsAhead; ++j) {
            const v4u r = ahead[j];
            ahead[j] = __builtin_nontemporal_load(next + j * kWave);
            rows_step<W, S, KIND>(D, words, r, slot0 + j, lane, n, a.min_symbol, P, status, mine);
        }
        if (slot0 == 64 - kRowsAhead) out[(tl + kRowsAhead - 64 + lane) * stride_t] = mine;     // (also the group an error stopped in)
    }
    // the last one or two blocks
    for (; tl < count && status == CST_STREAM_OK; tl += kRowsAhead) {
        if ((tl & 63) == 0) fetch_words();
#pragma unroll
        for (int j = 0; j < kRowsAhead; ++j) {
            const size_t t = tl + j;
            const v4u r = ahead[j];
            if (t + kRowsAhead < count) ahead[j] = __builtin_nontemporal_load(rows + (t + kRowsAhead) * kWave);
            if (t < count) rows_step<W, S, KIND>(D, words, r, (uint32_t)(t & 63), lane, n, a.min_symbol, P, status, mine);
            if (t < count && (t & 63) == 63) out[(t - 63 + lane) * stride_t] = mine;
        }
    }
    // the last, partial group (or the group an error stopped in: its remaining symbols are unspecified)
    const size_t reached = min(tl, count);
    if ((reached & 63) != 0 && (reached & ~(size_t)63) + lane < count) out[((reached & ~(size_t)63) + lane) * stride_t] = mine;
    if (lane != 0) return;
    if (ra.last) { a.status[s] = status; D.finish(a, s, raw); }
    else { DecodeResume r; D.park(r); r.status = status; ra.resume[s] = r; }
}

// ------------------------------------------------------------------------------------------------
// A FEW CHAINS (down to the reference's own usage: one).  A chain coder's quantiles do not depend on the models: symbol
// t's P bits sit at a position of `compressed` that only the head's fill level decides.  So decoding is three kernels:
//   1. per chain, sequential but trivial: cut the quantiles out of the words (no model in the loop);
//   2. per SYMBOL, in parallel over the whole chip: the model search (exact erf brackets / binary search of a cdf row);
//   3. per chain, sequential but trivial: fold (quantile - left, probability) into the remainders head, flush words.
// One wave per chain for 1 and 3, with the coder state uniform (the compiler keeps it in scalar registers) and the
// inputs fetched 64 at a time.  Against one wave per chain searching inside the loop: ~1.7 us -> ~0.1 us per symbol.
// ------------------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(kWave) void chain_quantiles_kernel(const PerSymbolDecodeArgs a, uint32_t* __restrict__ quantiles,
                                                                uint32_t* __restrict__ n_cut) {
    constexpr uint32_t wmask = W == 32 ? 0xffffffffu : ((1u << (W % 32)) - 1u);
    const int lane = threadIdx.x;
    const size_t s = blockIdx.x, N = a.n_per_stream;
    const int P = a.precision;
    const WordSlice ws = a.slice(s);
    const uint32_t* in = a.words + ws.off;
    const uint32_t n_words = max(ws.n, 1u);
    uint32_t rd = ws.n;
    uint32_t ch = a.heads[s].compressed_head;
    WordBlock<kAns> words;
    uint32_t mine = 0;
    size_t t = 0;
    for (; t < N; ++t) {
        if ((t & 63) == 0) { words.fetch(in, n_words, rd, lane); asm volatile("" : "+v"(words.cur)); }
        uint32_t word;
        if (P == W || ch < (1u << P)) {                         // decode_symbol, chain.rs:1060-1098
            if (rd == 0) break;                                 // DecoderFrontendError::OutOfCompressedData
            word = words.at(rd) & wmask; --rd;
            if (P != W) ch = ((ch << (W - P)) | (word >> P)) & wmask;
        } else {
            word = ch; ch >>= P;
        }
        const uint32_t q = P == W ? word : (word & ((1u << P) - 1u));
        if ((uint32_t)lane == (uint32_t)(t & 63)) mine = q;
        if ((t & 63) == 63) quantiles[s * N + t - 63 + lane] = mine;
    }
    if ((t & 63) != 0 && (uint32_t)lane < (uint32_t)(t & 63)) quantiles[s * N + (t & ~(size_t)63) + lane] = mine;
    if (lane == 0) {
        n_cut[s] = (uint32_t)t;
        a.heads[s].compressed_head = ch;
        a.n_words_out[s] = rd;
    }
}

// (symbol, quantile - left, probability) of every cut quantile; probability 0 marks an invalid model
template <bool GAUSSIAN>
__global__ __launch_bounds__(kBlock) void chain_lookup_kernel(const PerSymbolDecodeArgs a, const uint32_t* __restrict__ quantiles,
                                                              const uint32_t* __restrict__ n_cut, uint2* __restrict__ pairs) {
    __shared__ double2 erf_tab[kErfTabEntries];
    erf_tab_fill(erf_tab, threadIdx.x, blockDim.x);
    __syncthreads();
    const size_t N = a.n_per_stream;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_streams * N) return;
    const size_t s = i / N, t = i - s * N;
    if (t >= n_cut[s]) return;
    const size_t e = a.layout == CST_LAYOUT_SYMBOL_MAJOR ? t * a.n_streams + s : i;
    const uint32_t q = quantiles[i];
    const int P = a.precision;
    const uint32_t n = (uint32_t)a.n_symbols;
    uint32_t lo_i = 0, hi_i = n, lo_v = 0, hi_v = P >= 32 ? 0u : (1u << P);     // left(lo_i) = lo_v <= q < hi_v = left(hi_i)
    bool ok = true;
    if constexpr (GAUSSIAN) {
        const double mu = a.means[e], sd = a.stds[e];
        ok = sd > 0.0 && sd <= 1.7976931348623157e308 && fabs(mu) <= 1.7976931348623157e308;
        if (ok) {
            // the guess and the bracketing of decode_gaussian_lane_kernel
            const double free_weight = (double)((P >= 32 ? 0xffffffffu : ((1u << P) - 1u)) - (n - 1u));
            const float total_f = (float)(1ull << P), free_f = (float)free_weight;
            const double guess_shift = 0.5 - (double)a.min_symbol;
            const float below = (float)q + 0.5f, above = total_f - below;
            float z = ndtri_lower_f32(fminf(below, above) / total_f);
            double x = mu + sd * (double)(below < above ? z : -z) + guess_shift;
            if ((double)n * 64.0 > free_weight) {
                const float b1 = below - (float)fmin(fmax(x, 0.0), (double)(n - 1u)), a1 = free_f - b1;
                z = ndtri_lower_f32(fmaxf(fminf(b1, a1), 0.25f) / free_f);
                x = mu + sd * (double)(b1 < a1 ? z : -z) + guess_shift;
            }
            uint32_t probe = (uint32_t)fmin(fmax(x, 1.0), (double)(n - 1u)), step = 1;
            bool up = false, down = false;
            while (hi_i - lo_i > 1) {
                const uint32_t v = leaky_gaussian_left_quick((int32_t)probe, a.min_symbol, (int32_t)n, P, 32, mu, sd, erf_tab);
                if (v <= q) { lo_i = probe; lo_v = v; up = true; } else { hi_i = probe; hi_v = v; down = true; }
                if (up && down) probe = lo_i + (hi_i - lo_i) / 2;
                else if (up) probe = min(lo_i + step, hi_i - 1u);
                else probe = max(hi_i - min(step, hi_i - 1u), lo_i + 1u);
                step *= 2;
            }
        }
    } else {
        const uint32_t* row = a.cdf_rows + e * a.row_stride;
        while (hi_i - lo_i > 1) {                               // lookup_contiguous.rs:564-605: any search finds the same entry
            const uint32_t mid = lo_i + (hi_i - lo_i) / 2, v = row[mid];
            if (v <= q) { lo_i = mid; lo_v = v; } else { hi_i = mid; hi_v = v; }
        }
        lo_v = row[lo_i]; hi_v = row[lo_i + 1];
        ok = lo_v <= q && hi_v > lo_v && ((uint64_t)hi_v <= ((uint64_t)1 << P)) && q < hi_v;
    }
    const uint32_t p = hi_v - lo_v;
    ok = ok && p != 0 && lo_v <= q && (uint64_t)lo_v + p <= ((uint64_t)1 << P);
    a.symbols[e] = a.min_symbol + (int32_t)lo_i;
    pairs[i] = ok ? make_uint2(q - lo_v, p) : make_uint2(0u, 0u);
}

template <int W, int S>
__global__ __launch_bounds__(kWave) void chain_fold_kernel(const PerSymbolDecodeArgs a, const uint2* __restrict__ pairs,
                                                           const uint32_t* __restrict__ n_cut, int32_t bad_model_status) {
    using st_t = typename StateT<S>::type;
    constexpr uint32_t wmask = W == 32 ? 0xffffffffu : ((1u << (W % 32)) - 1u);
    const int lane = threadIdx.x;
    const size_t s = blockIdx.x, N = a.n_per_stream;
    const int P = a.precision;
    const uint32_t n = n_cut[s];
    st_t rh = (st_t)a.heads[s].remainders_head;
    uint32_t* out = a.push_words + s * a.push_stride;
    const uint32_t cap = (uint32_t)(a.push_stride > 0xffffffffull ? 0xffffffffull : a.push_stride);
    uint32_t wr = 0, mine = 0;
    int32_t status = CST_STREAM_OK;
    for (uint32_t t0 = 0; t0 < n && status == CST_STREAM_OK; t0 += kWave) {
        const uint2 pr = t0 + lane < n ? pairs[s * N + t0 + lane] : make_uint2(0u, 1u);
        const uint32_t m = min((uint32_t)kWave, n - t0);
        for (uint32_t j = 0; j < m; ++j) {
            const uint32_t rem = (uint32_t)__builtin_amdgcn_readlane((int)pr.x, (int)j);
            const uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)pr.y, (int)j);
            if (p == 0) { status = bad_model_status; break; }
            rh = (st_t)(rh * (st_t)p + (st_t)rem);              // decode_symbol, chain.rs:1104-1116
            if (rh >= ((st_t)1 << (S - P))) {
                if ((uint32_t)lane == (wr & 63u)) mine = (uint32_t)rh & wmask;
                if ((wr & 63u) == 63u && wr < cap) out[wr - 63u + lane] = mine;     // (cap is a multiple of 64 or the tail below stores)
                ++wr;
                rh = (st_t)(rh >> (W % S));
            }
        }
    }
    if ((wr & 63u) != 0 && (uint32_t)lane < (wr & 63u) && (wr & ~63u) + lane < cap) out[(wr & ~63u) + lane] = mine;
    if (lane != 0) return;
    if (status == CST_STREAM_OK && n < N) status = CST_STREAM_OUT_OF_DATA;
    if (wr > cap) status = CST_STREAM_CAPACITY;
    a.heads[s].remainders_head = (uint64_t)rh;
    a.n_push[s] = wr;
    a.status[s] = status;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------

static cst_status check_common(cst_coder_config cfg, cst_layout layout) {
    if (!config_supported(cfg)) return CST_ERR_INVALID_ARGUMENT;
    if (layout != CST_LAYOUT_STREAM_MAJOR && layout != CST_LAYOUT_SYMBOL_MAJOR) return CST_ERR_INVALID_ARGUMENT;
    return CST_OK;
}

template <int KIND>
static cst_status launch_encode_entries(cst_coder_config cfg, const EntriesEncodeArgs& a, hipStream_t hs) {
    const size_t blocks = (a.n_streams + kBlock - 1) / kBlock;
    const size_t lds = (size_t)(kBlock / kWave) * kRingWords * sizeof(uint32_t);
    if (cfg.word_bits == 32) hipLaunchKernelGGL((encode_entries_kernel<32, 64, KIND>), dim3((unsigned)blocks), dim3(kBlock), lds, hs, a);
    else hipLaunchKernelGGL((encode_entries_kernel<16, 32, KIND>), dim3((unsigned)blocks), dim3(kBlock), lds, hs, a);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

// Scratch (encoder entries, cdf rows of few-stream decodes, chain quantiles) comes from a stream-ordered memory pool
// that belongs to THIS LIBRARY -- one per device, created at first use -- whose release threshold is raised so that it
// keeps freed memory: a pool at its defaults hands everything back at the next synchronisation, and allocating 4 GiB
// afresh costs more than coding them (measured: 70 of 88 ms per call at 65 536 x 4096).  The device's DEFAULT pool is
// never touched: a host application's own hipMallocAsync traffic (PyTorch's async allocator, say) keeps its behaviour.
static std::mutex g_pool_mutex;
static hipMemPool_t g_pools[64] = {};

static hipError_t scratch_pool(hipMemPool_t* out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    if (!g_pools[dev]) {
        hipMemPoolProps props{};
        props.allocType = hipMemAllocationTypePinned;
        props.handleTypes = hipMemHandleTypeNone;
        props.location.type = hipMemLocationTypeDevice;
        props.location.id = dev;
        hipMemPool_t pool = nullptr;
        e = hipMemPoolCreate(&pool, &props);
        if (e != hipSuccess) return e;
        uint64_t keep = ~0ull;
        (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
        g_pools[dev] = pool;
    }
    *out = g_pools[dev];
    return hipSuccess;
}

static hipError_t scratch_alloc(void** ptr, size_t bytes, hipStream_t hs) {
    hipMemPool_t pool;
    hipError_t e = scratch_pool(&pool);
    if (e != hipSuccess) return e;
    return hipMallocFromPoolAsync(ptr, bytes, pool, hs);
}

// pass 1 (entries) + pass 2 (sequential coder); `fill` launches the entry kernel into the temporary buffer
template <int KIND, typename Fill>
static cst_status encode_two_pass(cst_coder_config cfg, size_t n_streams, size_t n_per_stream, cst_layout layout,
                                  uint32_t* d_words, size_t stride_words, uint32_t* d_n_words, uint64_t* d_state,
                                  cst_range_state* d_rstate, int32_t* d_status, uint32_t flags, hipStream_t hs, Fill fill) {
    if (cst_status st = check_common(cfg, layout)) return st;
    if (!d_words || !d_n_words || !d_status) return CST_ERR_INVALID_ARGUMENT;
    if ((flags & CST_FLAG_RAW_STATE) && (KIND == kAns ? (void*)d_state : (void*)d_rstate) == nullptr) return CST_ERR_INVALID_ARGUMENT;
    if (n_streams == 0) return CST_OK;
    const size_t n = n_streams * n_per_stream;
    EncEntry* entries = nullptr;
    if (n > 0) {
        CST_HIP_TRY(scratch_alloc((void**)&entries, n * sizeof(EncEntry), hs));
        fill(entries, n);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_hip_error(e, "entry kernel"); (void)hipFreeAsync(entries, hs); return CST_ERR_HIP; }
    }
    EntriesEncodeArgs a{};
    a.entries = entries; a.n_streams = n_streams; a.n_per_stream = n_per_stream; a.layout = layout; a.precision = cfg.precision;
    a.words = d_words; a.stride_words = stride_words; a.n_words = d_n_words; a.state = d_state; a.rstate = d_rstate;
    a.status = d_status; a.flags = flags;
    const cst_status st = launch_encode_entries<KIND>(cfg, a, hs);
    if (entries) CST_HIP_TRY(hipFreeAsync(entries, hs));
    return st;
}

// The fused kernel pays one coder step per kFuStreams-stream wave and symbol whatever the batch; the two-pass form runs its
// entry pass on the whole chip however few streams there are.  From 16 384 streams on (512 waves of 32 streams: two
// for every SIMD pair) the fused kernel is the faster one; below, and for the one long stream of the drop-in API, two passes.
// (CST_FUSED_MIN_STREAMS in the environment moves the threshold: the parity tests run the fused kernel on small batches.)
static bool fused_encode_usable(size_t n_streams, size_t n_per_stream) {
    return n_streams >= knobs().fused_min_streams && n_per_stream >= 1;
}

template <int KIND>
static cst_status encode_gaussian_fused(cst_coder_config cfg, int32_t min_symbol, int32_t max_symbol, const int32_t* d_symbols,
                                        const double* d_means, const double* d_stds, size_t n_streams, size_t n_per_stream, cst_layout layout,
                                        uint32_t* d_words, size_t stride_words, uint32_t* d_n_words, uint64_t* d_state,
                                        cst_range_state* d_rstate, int32_t* d_status, uint32_t flags, hipStream_t hs,
                                        size_t ckpt_interval = 0, uint32_t* d_ckpt_pos = nullptr, uint64_t* d_ckpt_state = nullptr,
                                        uint64_t* d_ckpt_lower = nullptr, uint64_t* d_ckpt_range = nullptr) {
    if (cst_status st = check_common(cfg, layout)) return st;
    if (!d_words || !d_n_words || !d_status) return CST_ERR_INVALID_ARGUMENT;
    if ((flags & CST_FLAG_RAW_STATE) && (KIND == kAns ? (void*)d_state : (void*)d_rstate) == nullptr) return CST_ERR_INVALID_ARGUMENT;
    GaussianFusedArgs a{};
    if (ckpt_interval) {
        a.ckpt_pos = d_ckpt_pos; a.ckpt_state = d_ckpt_state; a.ckpt_lower = d_ckpt_lower; a.ckpt_range = d_ckpt_range; a.interval = ckpt_interval;
        a.n_chunks = (n_per_stream + ckpt_interval - 1) / ckpt_interval;
    }
    a.symbols = d_symbols; a.means = d_means; a.stds = d_stds; a.n_streams = n_streams; a.n_per_stream = n_per_stream;
    a.layout = layout; a.precision = cfg.precision; a.lo = min_symbol; a.hi = max_symbol;
    a.words = d_words; a.stride_words = stride_words; a.n_words = d_n_words; a.state = d_state; a.rstate = d_rstate;
    a.status = d_status; a.flags = flags;
    const size_t per_block = (size_t)(kFuBlock / kWave) * kFuStreams;
    const size_t blocks = (n_streams + per_block - 1) / per_block;
    const size_t lds = kFuTabBytes + (size_t)(kFuBlock / kWave) * kFuWaveBytes;
    const bool pair = layout == CST_LAYOUT_STREAM_MAJOR && n_streams % kFuStreams == 0 && n_per_stream % kFuTile == 0 && n_per_stream > 0;
    if (cfg.word_bits == 32 && pair) {
        CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(encode_gaussian_fused_kernel<32, 64, KIND, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((encode_gaussian_fused_kernel<32, 64, KIND, true>), dim3((unsigned)blocks), dim3(kFuBlock), lds, hs, a);
    } else if (cfg.word_bits == 32) {
        CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(encode_gaussian_fused_kernel<32, 64, KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((encode_gaussian_fused_kernel<32, 64, KIND>), dim3((unsigned)blocks), dim3(kFuBlock), lds, hs, a);
    } else {
        CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(encode_gaussian_fused_kernel<16, 32, KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((encode_gaussian_fused_kernel<16, 32, KIND>), dim3((unsigned)blocks), dim3(kFuBlock), lds, hs, a);
    }
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

// few streams: cdf rows at full occupancy, then a lookup per symbol; in pieces of at most 64 MiB of rows
template <int KIND>
static cst_status decode_gaussian_by_rows(cst_coder_config cfg, const PerSymbolDecodeArgs& a, hipStream_t hs) {
    const size_t N = a.n_per_stream;
    // 65 536 rows = 64 MiB of rows per piece (the piece size makes no measurable difference from 4096 rows up: the
    // sequential kernel is bound by its dependent instruction chain, not by the rows' memory latency)
    size_t piece = ((size_t)65536 / a.n_streams) & ~(size_t)63;        // (n_streams < 64: at least 1024 symbols)
    if (piece > N) piece = N;
    uint32_t* rows = nullptr;
    DecodeResume* resume = nullptr;
    CST_HIP_TRY(scratch_alloc((void**)&rows, a.n_streams * piece * kRowEntries * sizeof(uint32_t), hs));
    hipError_t err = scratch_alloc((void**)&resume, a.n_streams * sizeof(DecodeResume), hs);
    for (size_t t0 = 0; t0 < N && err == hipSuccess; t0 += piece) {
        const size_t count = N - t0 < piece ? N - t0 : piece;
        hipLaunchKernelGGL(gaussian_rows_kernel, dim3((unsigned)(a.n_streams * count)), dim3(kRowEntries), 0, hs, a.precision, a.min_symbol,
                           a.n_symbols, a.means, a.stds, a.layout, a.n_streams, N, t0, count, rows);
        RowsDecodeArgs ra{a, reinterpret_cast<const uint4*>(rows), t0, count, resume, t0 == 0, t0 + count == N};
        if (cfg.word_bits == 32) hipLaunchKernelGGL((decode_rows_wave_kernel<32, 64, KIND>), dim3((unsigned)a.n_streams), dim3(kWave), 0, hs, ra);
        else hipLaunchKernelGGL((decode_rows_wave_kernel<16, 32, KIND>), dim3((unsigned)a.n_streams), dim3(kWave), 0, hs, ra);
        err = hipGetLastError();
    }
    if (resume) (void)hipFreeAsync(resume, hs);
    (void)hipFreeAsync(rows, hs);
    CST_HIP_TRY(err);
    return CST_OK;
}

// a few chains: cut the quantiles, search all symbols in parallel, fold the remainders
static cst_status decode_chains_in_three(cst_coder_config cfg, const PerSymbolDecodeArgs& a, bool gaussian, hipStream_t hs) {
    const size_t n = a.n_streams * a.n_per_stream;
    uint32_t* scratch = nullptr;                 // [quantiles n][pairs 2n][n_cut n_streams]
    CST_HIP_TRY(scratch_alloc((void**)&scratch, (3 * n + a.n_streams + 2) * sizeof(uint32_t), hs));
    uint32_t* quantiles = scratch;
    uint2* pairs = reinterpret_cast<uint2*>(scratch + ((n + 1) & ~(size_t)1));
    uint32_t* n_cut = scratch + ((n + 1) & ~(size_t)1) + 2 * n;
    if (cfg.word_bits == 32) hipLaunchKernelGGL((chain_quantiles_kernel<32>), dim3((unsigned)a.n_streams), dim3(kWave), 0, hs, a, quantiles, n_cut);
    else hipLaunchKernelGGL((chain_quantiles_kernel<16>), dim3((unsigned)a.n_streams), dim3(kWave), 0, hs, a, quantiles, n_cut);
    const unsigned blocks = (unsigned)((n + kBlock - 1) / kBlock);
    if (gaussian) hipLaunchKernelGGL((chain_lookup_kernel<true>), dim3(blocks), dim3(kBlock), 0, hs, a, quantiles, n_cut, pairs);
    else hipLaunchKernelGGL((chain_lookup_kernel<false>), dim3(blocks), dim3(kBlock), 0, hs, a, quantiles, n_cut, pairs);
    const int32_t bad = gaussian ? GaussianLeft::kBadModel : RowLeft::kBadModel;
    if (cfg.word_bits == 32) hipLaunchKernelGGL((chain_fold_kernel<32, 64>), dim3((unsigned)a.n_streams), dim3(kWave), 0, hs, a, pairs, n_cut, bad);
    else hipLaunchKernelGGL((chain_fold_kernel<16, 32>), dim3((unsigned)a.n_streams), dim3(kWave), 0, hs, a, pairs, n_cut, bad);
    const hipError_t err = hipGetLastError();
    (void)hipFreeAsync(scratch, hs);
    CST_HIP_TRY(err);
    return CST_OK;
}

template <int KIND>
static cst_status decode_per_symbol(cst_coder_config cfg, const PerSymbolDecodeArgs& a, bool gaussian, hipStream_t hs) {
    if (a.n_streams == 0) return CST_OK;
    const size_t blocks = (a.n_streams * kWave + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    if (KIND == kChain && a.n_streams < (size_t)kWave && a.n_per_stream >= 64 && a.n_streams * a.n_per_stream < ((size_t)1 << 31))
        return decode_chains_in_three(cfg, a, gaussian, hs);
    if (gaussian && a.n_streams >= (size_t)kWave) {      // enough streams to give every lane its own
        // more streams than one wave per SIMD: the small geometry, eight waves per CU (CST_LANE_GEO=big|small forces one: A/B runs)
        int cus = 256;
        { int dev = 0; hipDeviceProp_t prop; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount; }
        const int geo = knobs().lane_geo;
        const bool small = KIND != kChain && (geo ? geo == 2 : a.n_streams > (size_t)cus * kBlock);
        auto go = [&](auto kernel, int threads, size_t lds) -> cst_status {
            const size_t lane_blocks = (a.n_streams + threads - 1) / threads;
            CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kernel, dim3((unsigned)lane_blocks), dim3(threads), lds, hs, a);
            return CST_OK;
        };
        cst_status rc;
        if constexpr (KIND != kChain) {
            if (small) rc = cfg.word_bits == 32 ? go(decode_gaussian_lane_kernel<32, 64, KIND, true>, LaneGeo<true>::kThreads, LaneGeo<true>::kLdsBytes)
                                                : go(decode_gaussian_lane_kernel<16, 32, KIND, true>, LaneGeo<true>::kThreads, LaneGeo<true>::kLdsBytes);
            else rc = cfg.word_bits == 32 ? go(decode_gaussian_lane_kernel<32, 64, KIND, false>, LaneGeo<false>::kThreads, LaneGeo<false>::kLdsBytes)
                                          : go(decode_gaussian_lane_kernel<16, 32, KIND, false>, LaneGeo<false>::kThreads, LaneGeo<false>::kLdsBytes);
        } else {
            rc = cfg.word_bits == 32 ? go(decode_gaussian_lane_kernel<32, 64, KIND, false>, LaneGeo<false>::kThreads, LaneGeo<false>::kLdsBytes)
                                     : go(decode_gaussian_lane_kernel<16, 32, KIND, false>, LaneGeo<false>::kThreads, LaneGeo<false>::kLdsBytes);
        }
        if (rc != CST_OK) return rc;
        note_kernel(KIND == kChain ? "chain_decode_gaussian_lane_kernel" : KIND == kRange ? (small ? "range_decode_gaussian_lane_kernel<small>" : "range_decode_gaussian_lane_kernel")
                                   : (small ? "ans_decode_gaussian_lane_kernel<small>" : "ans_decode_gaussian_lane_kernel"), CST_OK);
    } else if (KIND != kChain && gaussian && a.n_symbols < kRowEntries && a.n_per_stream >= 32) {
        note_kernel("decode_gaussian_by_rows", CST_OK);
        if constexpr (KIND != kChain) return decode_gaussian_by_rows<KIND>(cfg, a, hs);
    } else if (gaussian) {
        if (cfg.word_bits == 32) hipLaunchKernelGGL((decode_wave_kernel<32, 64, KIND, GaussianLeft>), dim3((unsigned)blocks), dim3(kBlock), 0, hs, a);
        else hipLaunchKernelGGL((decode_wave_kernel<16, 32, KIND, GaussianLeft>), dim3((unsigned)blocks), dim3(kBlock), 0, hs, a);
    } else {
        if (cfg.word_bits == 32) hipLaunchKernelGGL((decode_wave_kernel<32, 64, KIND, RowLeft>), dim3((unsigned)blocks), dim3(kBlock), 0, hs, a);
        else hipLaunchKernelGGL((decode_wave_kernel<16, 32, KIND, RowLeft>), dim3((unsigned)blocks), dim3(kBlock), 0, hs, a);
    }
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

static cst_status fill_decode_args(PerSymbolDecodeArgs& a, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_offsets,
                                   size_t stride_words, size_t words_capacity, const uint32_t* d_n_words, int32_t* d_symbols, size_t n_streams,
                                   size_t n_per_stream, cst_layout layout, int32_t min_symbol, int64_t n_symbols, int32_t* d_status,
                                   uint32_t flags) {
    if (cst_status st = check_common(cfg, layout)) return st;
    if (!d_n_words || !d_status || (n_per_stream > 0 && !d_symbols)) return CST_ERR_INVALID_ARGUMENT;
    // (the same support limit as the encoding entry points: per-symbol models hold no tables, any n <= 2^P works)
    if (n_symbols < 2 || n_symbols > ((int64_t)1 << cfg.precision)) return CST_ERR_MODEL;
    a.words = d_words; a.offsets = d_offsets; a.stride_words = stride_words; a.n_words = d_n_words; a.symbols = d_symbols;
    a.n_streams = n_streams; a.n_per_stream = n_per_stream; a.layout = layout; a.precision = cfg.precision;
    a.min_symbol = min_symbol; a.n_symbols = (int32_t)n_symbols; a.status = d_status; a.flags = flags;
    a.row_stride = (size_t)n_symbols + 1; a.words_capacity = words_capacity;
    return CST_OK;
}

// ---- chain coder: shared parts of the entry points ----
static cst_status chain_decode_common(PerSymbolDecodeArgs& a, cst_coder_config cfg, const uint32_t* d_pop_words, const uint64_t* d_pop_offsets,
                                      size_t pop_stride, uint32_t* d_n_pop, int32_t* d_symbols, size_t n_streams, size_t n_per_stream,
                                      cst_layout layout, int32_t min_symbol, int64_t n_symbols, uint32_t* d_push_words, size_t push_stride,
                                      uint32_t* d_n_push, cst_chain_heads* d_heads, int32_t* d_status) {
    if (cst_status st = fill_decode_args(a, cfg, d_pop_words, d_pop_offsets, pop_stride, 0, d_n_pop, d_symbols, n_streams, n_per_stream, layout,
                                         min_symbol, n_symbols, d_status, 0)) return st;
    // (d_pop_words must be a valid device pointer even when every d_n_pop[s] is 0: the kernels request a stream's next word
    // ahead of knowing whether they will need it)
    if (!d_heads || !d_n_push || !d_pop_words || (n_per_stream > 0 && !d_push_words)) return CST_ERR_INVALID_ARGUMENT;
    a.push_words = d_push_words; a.push_stride = push_stride; a.n_push = d_n_push; a.heads = d_heads; a.n_words_out = d_n_pop;
    return CST_OK;
}

template <typename Fill>
static cst_status chain_encode_common(cst_coder_config cfg, size_t n_streams, size_t n_per_stream, cst_layout layout, const uint32_t* d_pop_words,
                                      const uint64_t* d_pop_offsets, size_t pop_stride, uint32_t* d_n_pop, uint32_t* d_push_words,
                                      size_t push_stride, uint32_t* d_n_push, cst_chain_heads* d_heads, int32_t* d_status, hipStream_t hs,
                                      Fill fill) {
    if (cst_status st = check_common(cfg, layout)) return st;
    if (!d_heads || !d_n_pop || !d_n_push || !d_status || !d_pop_words || (n_per_stream > 0 && !d_push_words)) return CST_ERR_INVALID_ARGUMENT;
    if (n_streams == 0) return CST_OK;
    const size_t n = n_streams * n_per_stream;
    EncEntry* entries = nullptr;
    if (n > 0) {
        CST_HIP_TRY(scratch_alloc((void**)&entries, n * sizeof(EncEntry), hs));
        fill(entries, n);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_hip_error(e, "entry kernel"); (void)hipFreeAsync(entries, hs); return CST_ERR_HIP; }
    }
    EntriesEncodeArgs a{};
    a.entries = entries; a.n_streams = n_streams; a.n_per_stream = n_per_stream; a.layout = layout; a.precision = cfg.precision;
    a.words = d_push_words; a.stride_words = push_stride; a.n_words = d_n_push; a.status = d_status;
    a.pop_words = d_pop_words; a.pop_offsets = d_pop_offsets; a.pop_stride = pop_stride; a.n_pop = d_n_pop; a.heads = d_heads;
    const cst_status st = launch_encode_entries<kChain>(cfg, a, hs);
    if (entries) CST_HIP_TRY(hipFreeAsync(entries, hs));
    return st;
}

} // namespace cst

using namespace cst;

extern "C" {

cst_status cst_release_scratch(void) {
    hipMemPool_t pool;
    CST_HIP_TRY(scratch_pool(&pool));
    CST_HIP_TRY(hipDeviceSynchronize());
    CST_HIP_TRY(hipMemPoolTrimTo(pool, 0));
    return CST_OK;
}

cst_status cst_ans_encode_cp_batch(cst_coder_config cfg, const uint32_t* d_left, const uint32_t* d_prob, size_t n_streams,
                                   size_t n_per_stream, cst_layout layout, uint32_t* d_words, size_t stride_words,
                                   uint32_t* d_n_words, uint64_t* d_state, int32_t* d_status, uint32_t flags, void* stream) {
    if (n_per_stream > 0 && (!d_left || !d_prob)) return CST_ERR_INVALID_ARGUMENT;
    hipStream_t hs = (hipStream_t)stream;
    return encode_two_pass<kAns>(cfg, n_streams, n_per_stream, layout, d_words, stride_words, d_n_words, d_state, nullptr,
                                 d_status, flags, hs, [&](EncEntry* out, size_t n) {
        hipLaunchKernelGGL(cp_entries_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, hs, d_left, d_prob, n, cfg.precision, out);
    });
}

cst_status cst_range_encode_cp_batch(cst_coder_config cfg, const uint32_t* d_left, const uint32_t* d_prob, size_t n_streams,
                                     size_t n_per_stream, cst_layout layout, uint32_t* d_words, size_t stride_words,
                                     uint32_t* d_n_words, cst_range_state* d_rstate, int32_t* d_status, uint32_t flags, void* stream) {
    if (n_per_stream > 0 && (!d_left || !d_prob)) return CST_ERR_INVALID_ARGUMENT;
    hipStream_t hs = (hipStream_t)stream;
    return encode_two_pass<kRange>(cfg, n_streams, n_per_stream, layout, d_words, stride_words, d_n_words, nullptr, d_rstate,
                                   d_status, flags, hs, [&](EncEntry* out, size_t n) {
        hipLaunchKernelGGL(cp_entries_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, hs, d_left, d_prob, n, cfg.precision, out);
    });
}

cst_status cst_ans_encode_gaussian_batch(cst_coder_config cfg, int32_t min_symbol, int32_t max_symbol, const int32_t* d_symbols,
                                         const double* d_means, const double* d_stds, size_t n_streams, size_t n_per_stream,
                                         cst_layout layout, uint32_t* d_words, size_t stride_words, uint32_t* d_n_words,
                                         uint64_t* d_state, int32_t* d_status, uint32_t flags, void* stream) {
    if (n_per_stream > 0 && (!d_symbols || !d_means || !d_stds)) return CST_ERR_INVALID_ARGUMENT;
    if (max_symbol <= min_symbol || (int64_t)max_symbol - min_symbol + 1 > ((int64_t)1 << cfg.precision)) return CST_ERR_MODEL;
    hipStream_t hs = (hipStream_t)stream;
    if (fused_encode_usable(n_streams, n_per_stream))
        return note_kernel("ans_encode_gaussian_fused_kernel", encode_gaussian_fused<kAns>(cfg, min_symbol, max_symbol, d_symbols, d_means, d_stds, n_streams, n_per_stream, layout, d_words,
                                           stride_words, d_n_words, d_state, nullptr, d_status, flags, hs));
    note_kernel("ans_encode_gaussian_two_pass", CST_OK);
    return encode_two_pass<kAns>(cfg, n_streams, n_per_stream, layout, d_words, stride_words, d_n_words, d_state, nullptr,
                                 d_status, flags, hs, [&](EncEntry* out, size_t n) {
        hipLaunchKernelGGL(gaussian_entries_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, hs, cfg.precision, min_symbol,
                           max_symbol, d_symbols, d_means, d_stds, n, out);
    });
}

// Jump points for the reference's flagship call (every symbol its own (mean, std)): the fused encoder notes AnsCoder::pos() in
// front of every chunk of `ckpt_interval` symbols (a multiple of the kernel's 16-symbol tile), and the decoder runs every
// (stream, chunk) pair as a coder of its own -- the per-symbol parameters are a matrix of the symbols' shape, so chunk j of stream
// s is row s * n_chunks + j of all three matrices viewed as [n_streams * n_chunks][interval].  What that buys: the lane decoder of
// 65 536 streams is ONE wave per SIMD and spends a third of its cycles waiting; with two jump points per stream the small-geometry
// kernel (LaneGeo<true>) runs two.
cst_status cst_ans_encode_gaussian_batch_ckpt(cst_coder_config cfg, int32_t min_symbol, int32_t max_symbol, const int32_t* d_symbols,
                                              const double* d_means, const double* d_stds, size_t n_streams, size_t n_per_stream,
                                              cst_layout layout, uint32_t* d_words, size_t stride_words, uint32_t* d_n_words,
                                              size_t ckpt_interval, uint32_t* d_ckpt_pos, uint64_t* d_ckpt_state, int32_t* d_status, void* stream) {
    if (n_per_stream > 0 && (!d_symbols || !d_means || !d_stds)) return CST_ERR_INVALID_ARGUMENT;
    if (!d_ckpt_pos || !d_ckpt_state || ckpt_interval == 0 || ckpt_interval % kFuTile != 0 || n_per_stream % ckpt_interval != 0) return CST_ERR_INVALID_ARGUMENT;
    if (max_symbol <= min_symbol || (int64_t)max_symbol - min_symbol + 1 > ((int64_t)1 << cfg.precision)) return CST_ERR_MODEL;
    if (n_streams == 0) return CST_OK;
    return note_kernel("ans_encode_gaussian_fused_kernel<ckpt>", encode_gaussian_fused<kAns>(cfg, min_symbol, max_symbol, d_symbols, d_means, d_stds, n_streams, n_per_stream, layout, d_words, stride_words,
                                       d_n_words, nullptr, nullptr, d_status, CST_FLAG_NONE, (hipStream_t)stream, ckpt_interval, d_ckpt_pos, d_ckpt_state));
}

__global__ void gaussian_ckpt_offsets_kernel(const uint64_t* __restrict__ offsets, size_t stride_words, size_t n_streams, size_t n_chunks,
                                             const uint64_t* __restrict__ state_in, uint64_t* __restrict__ v_offsets, uint64_t* __restrict__ v_state) {
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_streams * n_chunks) return;
    const size_t s = v / n_chunks;
    v_offsets[v] = offsets ? offsets[s] : s * stride_words;
    v_state[v] = state_in[v];                  // (the raw decode updates its state array)
}

cst_status cst_ans_decode_gaussian_batch_ckpt(cst_coder_config cfg, int32_t min_symbol, int32_t max_symbol, const uint32_t* d_words,
                                              const uint64_t* d_offsets, size_t stride_words, size_t words_capacity, size_t ckpt_interval,
                                              const uint32_t* d_ckpt_pos, const uint64_t* d_ckpt_state, const double* d_means, const double* d_stds,
                                              int32_t* d_symbols, size_t n_streams, size_t n_per_stream, void* d_scratch, int32_t* d_status, void* stream) {
    if (!d_ckpt_pos || !d_ckpt_state || !d_scratch || !d_status || ckpt_interval == 0 || n_per_stream % ckpt_interval != 0) return CST_ERR_INVALID_ARGUMENT;
    if (n_streams == 0 || n_per_stream == 0) return CST_OK;
    const size_t n_chunks = n_per_stream / ckpt_interval, n_virtual = n_streams * n_chunks;
    uint64_t* v_offsets = reinterpret_cast<uint64_t*>(d_scratch);
    uint64_t* v_state = v_offsets + n_virtual;
    hipLaunchKernelGGL(gaussian_ckpt_offsets_kernel, dim3((unsigned)((n_virtual + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_offsets, stride_words,
                       n_streams, n_chunks, d_ckpt_state, v_offsets, v_state);
    CST_HIP_TRY(hipGetLastError());
    const size_t capacity = words_capacity ? words_capacity : (d_offsets ? 0 : n_streams * stride_words);
    const cst_status rc = cst_ans_decode_gaussian_batch(cfg, min_symbol, max_symbol, d_words, v_offsets, 0, capacity, d_ckpt_pos, d_means, d_stds, d_symbols,
                                                        n_virtual, ckpt_interval, CST_LAYOUT_STREAM_MAJOR, v_state, nullptr, d_status, CST_FLAG_RAW_STATE, stream);
    if (rc != CST_OK) return rc;
    return flag_bad_jump_points(d_ckpt_pos, n_streams, n_chunks, d_offsets ? 0 : stride_words, d_status, (hipStream_t)stream);
}

// ... and for the range coder (round 6): the fused encoder notes RangeEncoder::pos() in front of every chunk, the decoder builds the
// RangeDecoder::seek states of the (stream, chunk) pairs (point re-read at the jump point: range_ckpt_virtual_state) and runs them as
// streams of their own -- the small-geometry lane decoder, two waves per SIMD, where the plain decoder of 65 536 streams has one
cst_status cst_range_encode_gaussian_batch_ckpt(cst_coder_config cfg, int32_t min_symbol, int32_t max_symbol, const int32_t* d_symbols,
                                                const double* d_means, const double* d_stds, size_t n_streams, size_t n_per_stream,
                                                cst_layout layout, uint32_t* d_words, size_t stride_words, uint32_t* d_n_words,
                                                size_t ckpt_interval, uint32_t* d_ckpt_pos, uint64_t* d_ckpt_lower, uint64_t* d_ckpt_range,
                                                int32_t* d_status, void* stream) {
    if (n_per_stream > 0 && (!d_symbols || !d_means || !d_stds)) return CST_ERR_INVALID_ARGUMENT;
    if (!d_ckpt_pos || !d_ckpt_lower || !d_ckpt_range || ckpt_interval == 0 || ckpt_interval % kFuTile != 0 || n_per_stream % ckpt_interval != 0)
        return CST_ERR_INVALID_ARGUMENT;
    if (max_symbol <= min_symbol || (int64_t)max_symbol - min_symbol + 1 > ((int64_t)1 << cfg.precision)) return CST_ERR_MODEL;
    if (n_streams == 0) return CST_OK;
    return note_kernel("range_encode_gaussian_fused_kernel<ckpt>",
                       encode_gaussian_fused<kRange>(cfg, min_symbol, max_symbol, d_symbols, d_means, d_stds, n_streams, n_per_stream, layout, d_words,
                                                     stride_words, d_n_words, nullptr, nullptr, d_status, CST_FLAG_NONE, (hipStream_t)stream, ckpt_interval,
                                                     d_ckpt_pos, nullptr, d_ckpt_lower, d_ckpt_range));
}

size_t cst_range_gaussian_ckpt_scratch_bytes(size_t n_streams, size_t n_per_stream, size_t ckpt_interval) {
    if (ckpt_interval == 0) return 0;
    return (sizeof(cst_range_state) + 16) * n_streams * ((n_per_stream + ckpt_interval - 1) / ckpt_interval) + 64;
}

} // extern "C"  (kernels and templates have C++ linkage)

// virtual stream v = (stream v / k, chunk v % k): where its words lie, and the decoder state RangeDecoder::seek leaves (queue.rs:911-926)
template <int W, int S>
__global__ void gaussian_range_virtual_kernel(const uint32_t* __restrict__ words, const uint64_t* __restrict__ offsets, size_t stride_words,
                                              uint64_t capacity, const uint32_t* __restrict__ n_words, const uint32_t* __restrict__ ckpt_pos,
                                              const uint64_t* __restrict__ ckpt_lower, const uint64_t* __restrict__ ckpt_range, size_t n_streams,
                                              size_t n_chunks, uint64_t* __restrict__ v_offsets, uint32_t* __restrict__ v_n,
                                              cst_range_state* __restrict__ v_state) {
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_streams * n_chunks) return;
    const size_t s = v / n_chunks;
    const WordSlice ws = word_slice(offsets, stride_words, n_words, s, capacity);
    const uint32_t pos0 = ckpt_pos[v] < ws.n ? ckpt_pos[v] : ws.n;
    uint64_t pt = 0;
    uint32_t pos = pos0;
    int num_read = 0;
    const uint64_t mask = S == 64 ? ~0ull : ((1ull << (S % 64)) - 1ull);
    while (pos < ws.n) {                                          // read_point, queue.rs:847-868
        pt = ((pt << (W % 64)) | (uint64_t)words[ws.off + pos++]) & mask;
        if (++num_read == S / W) break;
    }
    if (num_read < S / W && num_read != 0) pt = (pt << (S - num_read * W)) & mask;
    cst_range_state r{};
    r.lower = ckpt_lower[v]; r.range = ckpt_range[v]; r.point = pt; r.position = pos;
    v_state[v] = r;
    v_offsets[v] = offsets ? offsets[s] : (uint64_t)s * stride_words;   // (the decoder checks the slice again: a bad one stays bad)
    v_n[v] = n_words[s];
}

__global__ void gaussian_range_flag_kernel(const uint32_t* __restrict__ n_words, const uint32_t* __restrict__ ckpt_pos, size_t n_streams, size_t n_chunks,
                                           int32_t* __restrict__ status) {
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_streams * n_chunks) return;
    if (ckpt_pos[v] > n_words[v / n_chunks]) status[v] = CST_STREAM_INVALID_DATA;      // a jump point beyond its stream's words
}

extern "C" {

cst_status cst_range_decode_gaussian_batch_ckpt(cst_coder_config cfg, int32_t min_symbol, int32_t max_symbol, const uint32_t* d_words,
                                                const uint64_t* d_offsets, size_t stride_words, size_t words_capacity, const uint32_t* d_n_words,
                                                size_t ckpt_interval, const uint32_t* d_ckpt_pos, const uint64_t* d_ckpt_lower,
                                                const uint64_t* d_ckpt_range, const double* d_means, const double* d_stds, int32_t* d_symbols,
                                                size_t n_streams, size_t n_per_stream, void* d_scratch, int32_t* d_status, void* stream) {
    if (!d_n_words || !d_ckpt_pos || !d_ckpt_lower || !d_ckpt_range || !d_scratch || !d_status || ckpt_interval == 0 || n_per_stream % ckpt_interval != 0)
        return CST_ERR_INVALID_ARGUMENT;
    if (cfg.word_bits != 32 && cfg.word_bits != 16) return CST_ERR_INVALID_ARGUMENT;
    if (n_streams == 0 || n_per_stream == 0) return CST_OK;
    if (!d_words) return CST_ERR_INVALID_ARGUMENT;
    const size_t n_chunks = n_per_stream / ckpt_interval, n_virtual = n_streams * n_chunks;
    if (n_virtual > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    hipStream_t hs = (hipStream_t)stream;
    const size_t capacity = words_capacity ? words_capacity : (d_offsets ? 0 : n_streams * stride_words);
    cst_range_state* v_state = reinterpret_cast<cst_range_state*>((reinterpret_cast<uintptr_t>(d_scratch) + 15) & ~(uintptr_t)15);
    uint64_t* v_offsets = reinterpret_cast<uint64_t*>(v_state + n_virtual);
    uint32_t* v_n = reinterpret_cast<uint32_t*>(v_offsets + n_virtual);
    const dim3 grid((unsigned)((n_virtual + 255) / 256));
    if (cfg.word_bits == 32)
        hipLaunchKernelGGL((gaussian_range_virtual_kernel<32, 64>), grid, dim3(256), 0, hs, d_words, d_offsets, stride_words, capacity, d_n_words, d_ckpt_pos,
                           d_ckpt_lower, d_ckpt_range, n_streams, n_chunks, v_offsets, v_n, v_state);
    else
        hipLaunchKernelGGL((gaussian_range_virtual_kernel<16, 32>), grid, dim3(256), 0, hs, d_words, d_offsets, stride_words, capacity, d_n_words, d_ckpt_pos,
                           d_ckpt_lower, d_ckpt_range, n_streams, n_chunks, v_offsets, v_n, v_state);
    CST_HIP_TRY(hipGetLastError());
    const cst_status rc = cst_range_decode_gaussian_batch(cfg, min_symbol, max_symbol, d_words, v_offsets, 0, capacity, v_n, d_means, d_stds, d_symbols,
                                                          n_virtual, ckpt_interval, CST_LAYOUT_STREAM_MAJOR, v_state, d_status, CST_FLAG_RAW_STATE, stream);
    if (rc != CST_OK) return rc;
    hipLaunchKernelGGL(gaussian_range_flag_kernel, grid, dim3(256), 0, hs, d_n_words, d_ckpt_pos, n_streams, n_chunks, d_status);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

cst_status cst_range_encode_gaussian_batch(cst_coder_config cfg, int32_t min_symbol, int32_t max_symbol, const int32_t* d_symbols,
                                           const double* d_means, const double* d_stds, size_t n_streams, size_t n_per_stream,
                                           cst_layout layout, uint32_t* d_words, size_t stride_words, uint32_t* d_n_words,
                                           cst_range_state* d_rstate, int32_t* d_status, uint32_t flags, void* stream) {
    if (n_per_stream > 0 && (!d_symbols || !d_means || !d_stds)) return CST_ERR_INVALID_ARGUMENT;
    if (max_symbol <= min_symbol || (int64_t)max_symbol - min_symbol + 1 > ((int64_t)1 << cfg.precision)) return CST_ERR_MODEL;
    hipStream_t hs = (hipStream_t)stream;
    if (fused_encode_usable(n_streams, n_per_stream))
        return encode_gaussian_fused<kRange>(cfg, min_symbol, max_symbol, d_symbols, d_means, d_stds, n_streams, n_per_stream, layout, d_words,
                                             stride_words, d_n_words, nullptr, d_rstate, d_status, flags, hs);
    return encode_two_pass<kRange>(cfg, n_streams, n_per_stream, layout, d_words, stride_words, d_n_words, nullptr, d_rstate,
                                   d_status, flags, hs, [&](EncEntry* out, size_t n) {
        hipLaunchKernelGGL(gaussian_entries_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, hs, cfg.precision, min_symbol,
                           max_symbol, d_symbols, d_means, d_stds, n, out);
    });
}

cst_status cst_ans_decode_gaussian_batch(cst_coder_config cfg, int32_t min_symbol, int32_t max_symbol, const uint32_t* d_words,
                                         const uint64_t* d_offsets, size_t stride_words, size_t words_capacity, const uint32_t* d_n_words,
                                         const double* d_means, const double* d_stds, int32_t* d_symbols, size_t n_streams,
                                         size_t n_per_stream, cst_layout layout, uint64_t* d_state, uint32_t* d_n_words_out,
                                         int32_t* d_status, uint32_t flags, void* stream) {
    PerSymbolDecodeArgs a{};
    if (max_symbol <= min_symbol) return CST_ERR_MODEL;
    if (cst_status st = fill_decode_args(a, cfg, d_words, d_offsets, stride_words, words_capacity, d_n_words, d_symbols, n_streams, n_per_stream, layout,
                                         min_symbol, (int64_t)max_symbol - min_symbol + 1, d_status, flags)) return st;
    if (n_per_stream > 0 && (!d_means || !d_stds)) return CST_ERR_INVALID_ARGUMENT;
    if ((flags & CST_FLAG_RAW_STATE) && !d_state) return CST_ERR_INVALID_ARGUMENT;
    a.means = d_means; a.stds = d_stds; a.state = d_state; a.n_words_out = d_n_words_out;
    return decode_per_symbol<kAns>(cfg, a, true, (hipStream_t)stream);
}

cst_status cst_range_decode_gaussian_batch(cst_coder_config cfg, int32_t min_symbol, int32_t max_symbol, const uint32_t* d_words,
                                           const uint64_t* d_offsets, size_t stride_words, size_t words_capacity, const uint32_t* d_n_words,
                                           const double* d_means, const double* d_stds, int32_t* d_symbols, size_t n_streams,
                                           size_t n_per_stream, cst_layout layout, cst_range_state* d_rstate, int32_t* d_status,
                                           uint32_t flags, void* stream) {
    PerSymbolDecodeArgs a{};
    if (max_symbol <= min_symbol) return CST_ERR_MODEL;
    if (cst_status st = fill_decode_args(a, cfg, d_words, d_offsets, stride_words, words_capacity, d_n_words, d_symbols, n_streams, n_per_stream, layout,
                                         min_symbol, (int64_t)max_symbol - min_symbol + 1, d_status, flags)) return st;
    if (n_per_stream > 0 && (!d_means || !d_stds)) return CST_ERR_INVALID_ARGUMENT;
    if ((flags & CST_FLAG_RAW_STATE) && !d_rstate) return CST_ERR_INVALID_ARGUMENT;
    a.means = d_means; a.stds = d_stds; a.rstate = d_rstate;
    return decode_per_symbol<kRange>(cfg, a, true, (hipStream_t)stream);
}

cst_status cst_ans_decode_rows_batch(cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_offsets, size_t stride_words, size_t words_capacity,
                                     const uint32_t* d_n_words, const uint32_t* d_cdf_rows, int32_t n_symbols, int32_t min_symbol,
                                     int32_t* d_symbols, size_t n_streams, size_t n_per_stream, cst_layout layout, uint64_t* d_state,
                                     uint32_t* d_n_words_out, int32_t* d_status, uint32_t flags, void* stream) {
    PerSymbolDecodeArgs a{};
    if (cst_status st = fill_decode_args(a, cfg, d_words, d_offsets, stride_words, words_capacity, d_n_words, d_symbols, n_streams, n_per_stream, layout,
                                         min_symbol, n_symbols, d_status, flags)) return st;
    if (n_per_stream > 0 && !d_cdf_rows) return CST_ERR_INVALID_ARGUMENT;
    if ((flags & CST_FLAG_RAW_STATE) && !d_state) return CST_ERR_INVALID_ARGUMENT;
    a.cdf_rows = d_cdf_rows; a.state = d_state; a.n_words_out = d_n_words_out;
    return decode_per_symbol<kAns>(cfg, a, false, (hipStream_t)stream);
}

cst_status cst_range_decode_rows_batch(cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_offsets, size_t stride_words, size_t words_capacity,
                                       const uint32_t* d_n_words, const uint32_t* d_cdf_rows, int32_t n_symbols, int32_t min_symbol,
                                       int32_t* d_symbols, size_t n_streams, size_t n_per_stream, cst_layout layout,
                                       cst_range_state* d_rstate, int32_t* d_status, uint32_t flags, void* stream) {
    PerSymbolDecodeArgs a{};
    if (cst_status st = fill_decode_args(a, cfg, d_words, d_offsets, stride_words, words_capacity, d_n_words, d_symbols, n_streams, n_per_stream, layout,
                                         min_symbol, n_symbols, d_status, flags)) return st;
    if (n_per_stream > 0 && !d_cdf_rows) return CST_ERR_INVALID_ARGUMENT;
    if ((flags & CST_FLAG_RAW_STATE) && !d_rstate) return CST_ERR_INVALID_ARGUMENT;
    a.cdf_rows = d_cdf_rows; a.rstate = d_rstate;
    return decode_per_symbol<kRange>(cfg, a, false, (hipStream_t)stream);
}

// ---- chain coder ----
cst_status cst_chain_decode_gaussian_batch(cst_coder_config cfg, int32_t min_symbol, int32_t max_symbol, const uint32_t* d_pop_words,
                                           const uint64_t* d_pop_offsets, size_t pop_stride, uint32_t* d_n_pop, const double* d_means,
                                           const double* d_stds, int32_t* d_symbols, size_t n_streams, size_t n_per_stream,
                                           cst_layout layout, uint32_t* d_push_words, size_t push_stride, uint32_t* d_n_push,
                                           cst_chain_heads* d_heads, int32_t* d_status, void* stream) {
    PerSymbolDecodeArgs a{};
    if (max_symbol <= min_symbol) return CST_ERR_MODEL;
    if (cst_status st = chain_decode_common(a, cfg, d_pop_words, d_pop_offsets, pop_stride, d_n_pop, d_symbols, n_streams, n_per_stream, layout,
                                            min_symbol, (int64_t)max_symbol - min_symbol + 1, d_push_words, push_stride, d_n_push, d_heads,
                                            d_status)) return st;
    if (n_per_stream > 0 && (!d_means || !d_stds)) return CST_ERR_INVALID_ARGUMENT;
    a.means = d_means; a.stds = d_stds;
    return decode_per_symbol<kChain>(cfg, a, true, (hipStream_t)stream);
}

cst_status cst_chain_decode_rows_batch(cst_coder_config cfg, const uint32_t* d_pop_words, const uint64_t* d_pop_offsets, size_t pop_stride,
                                       uint32_t* d_n_pop, const uint32_t* d_cdf_rows, size_t row_stride, int32_t n_symbols,
                                       int32_t min_symbol, int32_t* d_symbols, size_t n_streams, size_t n_per_stream, cst_layout layout,
                                       uint32_t* d_push_words, size_t push_stride, uint32_t* d_n_push, cst_chain_heads* d_heads,
                                       int32_t* d_status, void* stream) {
    PerSymbolDecodeArgs a{};
    if (cst_status st = chain_decode_common(a, cfg, d_pop_words, d_pop_offsets, pop_stride, d_n_pop, d_symbols, n_streams, n_per_stream, layout,
                                            min_symbol, n_symbols, d_push_words, push_stride, d_n_push, d_heads, d_status)) return st;
    if (n_per_stream > 0 && !d_cdf_rows) return CST_ERR_INVALID_ARGUMENT;
    if (row_stride != 0 && row_stride != (size_t)n_symbols + 1) return CST_ERR_INVALID_ARGUMENT;
    a.cdf_rows = d_cdf_rows; a.row_stride = row_stride;
    return decode_per_symbol<kChain>(cfg, a, false, (hipStream_t)stream);
}

cst_status cst_chain_encode_cp_batch(cst_coder_config cfg, const uint32_t* d_left, const uint32_t* d_prob, size_t n_streams,
                                     size_t n_per_stream, cst_layout layout, const uint32_t* d_pop_words, const uint64_t* d_pop_offsets,
                                     size_t pop_stride, uint32_t* d_n_pop, uint32_t* d_push_words, size_t push_stride, uint32_t* d_n_push,
                                     cst_chain_heads* d_heads, int32_t* d_status, void* stream) {
    if (n_per_stream > 0 && (!d_left || !d_prob)) return CST_ERR_INVALID_ARGUMENT;
    hipStream_t hs = (hipStream_t)stream;
    return chain_encode_common(cfg, n_streams, n_per_stream, layout, d_pop_words, d_pop_offsets, pop_stride, d_n_pop, d_push_words, push_stride,
                               d_n_push, d_heads, d_status, hs, [&](EncEntry* out, size_t n) {
        hipLaunchKernelGGL(cp_entries_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, hs, d_left, d_prob, n, cfg.precision, out);
    });
}

cst_status cst_chain_encode_gaussian_batch(cst_coder_config cfg, int32_t min_symbol, int32_t max_symbol, const int32_t* d_symbols,
                                           const double* d_means, const double* d_stds, size_t n_streams, size_t n_per_stream,
                                           cst_layout layout, const uint32_t* d_pop_words, const uint64_t* d_pop_offsets, size_t pop_stride,
                                           uint32_t* d_n_pop, uint32_t* d_push_words, size_t push_stride, uint32_t* d_n_push,
                                           cst_chain_heads* d_heads, int32_t* d_status, void* stream) {
    if (n_per_stream > 0 && (!d_symbols || !d_means || !d_stds)) return CST_ERR_INVALID_ARGUMENT;
    if (max_symbol <= min_symbol || (int64_t)max_symbol - min_symbol + 1 > ((int64_t)1 << cfg.precision)) return CST_ERR_MODEL;
    hipStream_t hs = (hipStream_t)stream;
    return chain_encode_common(cfg, n_streams, n_per_stream, layout, d_pop_words, d_pop_offsets, pop_stride, d_n_pop, d_push_words, push_stride,
                               d_n_push, d_heads, d_status, hs, [&](EncEntry* out, size_t n) {
        hipLaunchKernelGGL(gaussian_entries_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, hs, cfg.precision, min_symbol,
                           max_symbol, d_symbols, d_means, d_stds, n, out);
    });
}

} // extern "C"
