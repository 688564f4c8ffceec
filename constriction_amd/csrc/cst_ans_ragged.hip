// cst_ans_ragged.hip -- batched ANS for streams of DIFFERENT lengths: thousands of small coders in one launch.
//
// The reference's own usage pattern next to "one long message" is "many short ones": a compressed index whose documents
// are coded one AnsCoder each with a shared model (tests/issue52.rs: `coder.encode_symbol` per character, last to first,
// `into_compressed` per document; `from_compressed` + `decode_symbol` per document).  Through the single-coder drop-in
// that is one device round trip per document; here it is one launch for all of them:
//     symbols of stream s   = d_symbols[d_sym_offsets[s] .. d_sym_offsets[s + 1])        (a CSR-style ragged array)
//     words of stream s     = d_words[off(s) .. off(s) + d_n_words[s]),  off(s) = d_word_offsets ? d_word_offsets[s] : s * stride
// One lane per stream, a wave runs as many steps as its longest stream (the others idle behind a predicate).  These batches
// are bound by the LATENCY of a step, not by throughput (a document's steps are a chain and few waves are resident), so the
// kernels keep everything a step waits for within the CU:
//   * tables in LDS (STAGED: the encoder's 16-byte entries / the decoder's cdf + 16-byte bucket entries with second-level
//     tables, DecLut in cst_common.hpp) whenever they fit in 64 KiB -- a few KiB to 42 KiB per workgroup, staged once;
//     larger alphabets read them from HBM / L2;
//   * symbols in GROUPS OF EIGHT: the encoder codes the (len mod 8) symbols at the end of a row first, then whole groups
//     from two 16-byte loads per lane issued a full group ahead; the decoder collects eight symbols in registers and
//     stores them as two 16-byte pieces;
//   * one memory point per group: a gfx9 wave has ONE counter for its loads and stores (vmcnt) and the compiler waits for
//     all of them wherever it cannot count what a divergent branch issued, so every group first consumes what was
//     requested a whole group earlier (next symbols / landed word chunks), THEN issues this group's requests and stores,
//     then runs its eight steps out of LDS and registers.  The per-step form of round 3's first version (a symbol load, a
//     table read from L2, a chunk store or load per step) paid a memory round trip per step: 0.5-1 us per symbol.
// Words go through the per-lane LDS rings of the other kernels.
// Every stream's words are those of cst_ans_encode_batch for that stream alone (stack.rs:835-849, 891-895; 1070-1100).
#include "cst_ans_kernels.hpp"

namespace cst {

struct RaggedArgs {
    const int32_t* symbols_in;
    int32_t* symbols_out;
    const uint64_t* sym_offsets;     // [n_streams + 1]
    size_t n_streams;
    const EncEntry* enc;
    const uint32_t* cdf;
    const uint16_t* bucket;
    int32_t bucket_bits, n_symbols, min_symbol, precision;
    uint32_t* words_out;
    const uint32_t* words_in;
    const uint64_t* word_offsets;    // [n_streams + 1] (encode: slab of stream s = [off[s], off[s + 1])) or null
    size_t stride_words;
    uint32_t* n_words_out;
    const uint32_t* n_words_in;
    int32_t* status;
    uint64_t words_capacity;
    const uint32_t* order;           // null, or [n_streams]: lane slot i codes stream order[i] (streams of similar length side by side)
    // Jump points (round 6; the reference's Pos / Seek, stack.rs:1107-1139): AnsCoder::pos() in front of every chunk of `jump_interval`
    // symbols of every stream -- chunk j of stream s is entry jump_chunk_offsets[s] + j of jump_pos / jump_state -- noted by the encoder
    // on its way (the words are unchanged).  The decoder of such a batch runs every chunk as a stream of its own, its coder state
    // taken from `state_in` instead of the end of its words.
    uint32_t jump_interval;                 // 0 = none; a multiple of kRaggedGroup
    const uint64_t* jump_chunk_offsets;     // [n_streams + 1]
    uint32_t* jump_pos;
    uint64_t* jump_state;
    const uint64_t* state_in;               // decoder: [n_streams] raw coder states, or null (read_initial_state)
};

// lane slot -> stream: the slot itself, or order[slot] (an entry that is not a stream leaves its lane idle)
__device__ __forceinline__ size_t ragged_stream(const RaggedArgs& a, size_t slot, bool& active) {
    active = slot < a.n_streams;
    if (!active || !a.order) return slot;
    const size_t s = a.order[slot];
    active = s < a.n_streams;
    return s;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d));
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

typedef int32_t rv4i __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) rv4i_unaligned { rv4i v; };      // 16-byte access at a 4-byte boundary

constexpr size_t kRaggedRingBytes = (size_t)(kBlock / kWave) * kRingWords * 4;
constexpr int kRaggedGroup = 8;                  // symbols per memory point (<= kAhead / 2 - 4: see the decoder's window)
constexpr size_t kRaggedStageLimit = 64 * 1024;  // tables up to this size are staged in LDS

// "the values requested a group ago are needed HERE": the compiler puts its wait in front of this (everything outstanding
// is a whole group old by then) instead of behind the requests and stores of the new group
__device__ __forceinline__ void consume(rv4i& a, rv4i& b) {
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w));
}

// FAST (W = 32, P >= 8): the hand-scheduled coder steps of the other kernels (32-bit halves, cst_ans_asm.hpp)
// G = symbols per memory point: 8, or 16 / 32 (round 6) -- a lane's next group is requested one group ahead, and eight steps (~800
// cycles) do not cover a gather's round trip under load: the LONGEST stream of a batch is a chain of len / G such waits, and with
// G = 8 it set every batch's time (2000 symbols: 0.30 ms whether the batch held 2 000 or 100 000 documents)
template <int W, int S, bool STAGED, bool FAST, int G = 8>
__global__ __launch_bounds__(kBlock) void ans_encode_ragged_kernel(const RaggedArgs a) {
    constexpr int kRaggedGroup = G;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    uint32_t* ring = reinterpret_cast<uint32_t*>(smem) + (threadIdx.x >> 6) * kRingWords;
    if constexpr (FAST) { if ((lds_addr(ring) & (kRingWords * 4u - 1u)) != 0) __builtin_trap(); }     // step<true> forms ring addresses with and/or
    const EncEntry* table = a.enc;
    if constexpr (STAGED) {
        EncEntry* t = reinterpret_cast<EncEntry*>(smem + kRaggedRingBytes);
        for (int i = threadIdx.x; i < a.n_symbols; i += blockDim.x) t[i] = a.enc[i];
        table = t;
        __syncthreads();
    }
    const size_t slot = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot - lane >= a.n_streams) return;
    bool active;
    const size_t s = ragged_stream(a, slot, active);
    const int P = a.precision;
    const uint32_t nsym = (uint32_t)a.n_symbols;
    const uint64_t sym_lo = active ? a.sym_offsets[s] : 0, sym_hi = active ? a.sym_offsets[s + 1] : 0;
    const bool too_long = sym_hi - sym_lo > 0xffffffffull || sym_hi < sym_lo;
    const uint32_t len = too_long ? 0u : (uint32_t)(sym_hi - sym_lo);
    const uint64_t slab_lo = !active ? 0 : (a.word_offsets ? a.word_offsets[s] : (uint64_t)s * a.stride_words);
    // (offsets that run backwards -- corrupt metadata -- give the stream a slab of NO words: it reports CST_STREAM_CAPACITY and
    //  writes nothing, instead of an "unbounded" slab over its neighbours)
    const uint64_t slab_hi = !active ? 0 : (a.word_offsets ? a.word_offsets[s + 1] : 0);
    const uint64_t slab_n = !active ? 0 : (a.word_offsets ? (slab_hi >= slab_lo ? slab_hi - slab_lo : 0) : (uint64_t)a.stride_words);
    EncLane<W, S> L;
    L.init(a.words_out + slab_lo, (uint32_t)(slab_n > 0xffffffffull ? 0xffffffffull : slab_n), ring, lane);
    const int32_t* row = a.symbols_in + sym_lo;
    auto entry = [&](int32_t v) { return table[enc_index(v, a.min_symbol, nsym, L.bad)]; };
    // encode_iid_symbols_reverse: last symbol first (stack.rs:835-849).  The (len mod 8) symbols at the END of the row one by
    // one (all their loads in flight at once), so that what remains is whole groups
    // AnsCoder::pos() once the symbols [start, len) are encoded, whenever a chunk starts at `start`.  No division in the loop: the
    // lane counts groups down to its next chunk boundary (`to_jump`) and walks the table backwards (`next_chunk`).
    const bool jumps = a.jump_interval != 0 && active && len > 0;
    const uint32_t groups_per_chunk = a.jump_interval ? a.jump_interval / (uint32_t)kRaggedGroup : 1u;
    const uint32_t ng0 = len / kRaggedGroup;
    uint32_t to_jump = ng0 % groups_per_chunk;                   // whole groups above the highest chunk boundary at or below 8 ng0
    uint64_t next_chunk = jumps ? a.jump_chunk_offsets[s] + ng0 / groups_per_chunk : 0;       // the chunk that starts at that boundary
    // A jump point is NOTED where it occurs and STORED at the next group's memory point, next to the word chunks (the kernel's rule:
    // one memory point per group -- the 64 lanes of a wave reach their chunk boundaries in different groups, so a store right behind
    // the steps would sit in front of nearly every group's wait for its symbols).
    bool jp_pending = false;
    uint32_t jp_pos = 0;
    uint64_t jp_state = 0, jp_chunk = 0;
    auto note_here = [&]() {
        jp_pos = L.out.wr; jp_state = (uint64_t)L.state; jp_chunk = next_chunk; jp_pending = true;
        --next_chunk;
    };
    auto store_jump_point = [&]() {
        if (jp_pending) { a.jump_pos[jp_chunk] = jp_pos; a.jump_state[jp_chunk] = jp_state; jp_pending = false; }
    };
    const uint32_t pre = len & (uint32_t)(kRaggedGroup - 1);
    if (__any(pre != 0)) {
        int32_t v[kRaggedGroup - 1];
#pragma unroll
        for (int j = 0; j < kRaggedGroup - 1; ++j) v[j] = (uint32_t)j < pre ? row[len - 1u - (uint32_t)j] : 0;
#pragma unroll
        for (int j = 0; j < kRaggedGroup - 1; ++j)
            if ((uint32_t)j < pre) L.template step<FAST>(entry(v[j]), P);
        if (jumps && pre != 0 && to_jump == 0) note_here();     // the ragged top part IS a chunk (8 ng0 is a multiple of the interval)
    }
    if (jumps && to_jump == 0) { to_jump = groups_per_chunk; if (pre == 0) --next_chunk; }      // (no chunk starts at len itself)
    const uint32_t ng = len / kRaggedGroup;                     // whole groups, coded from the last one down
    const uint32_t mxg = wave_max_u32(ng);
    constexpr int Q = G / 4;                                                        // 16-byte pieces per group
    const rv4i_unaligned* g4 = reinterpret_cast<const rv4i_unaligned*>(row);        // group g = pieces Q g .. Q g + Q - 1
    rv4i nx[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) nx[q] = rv4i{0, 0, 0, 0};
    if (ng > 0) {
#pragma unroll
        for (int q = 0; q < Q; ++q) nx[q] = g4[Q * (ng - 1) + q].v;
    }
    for (uint32_t g = 0; g < mxg; ++g) {
#pragma unroll
        for (int q = 0; q < Q; q += 2) consume(nx[q], nx[q + 1]);      // group g's symbols (requested a group ago) -- and every older store
        rv4i cur[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) cur[q] = nx[q];
        if (g + 1 < ng) {
#pragma unroll
            for (int q = 0; q < Q; ++q) nx[q] = g4[Q * (ng - g - 2) + q].v;
        }
        L.flush_chunks();                   // complete 16-byte chunks of the words of earlier groups: ring -> slab (at most 5)
        if constexpr (G > 16) L.flush_chunks();
        store_jump_point();                 // (a jump point noted by the previous group)
        if (g < ng) {
#pragma unroll
            for (int q = Q - 1; q >= 0; q -= 2) {                       // eight steps at a time, the group's last symbols first
                const rv4i c0 = cur[q - 1], c1 = cur[q];
                const EncEntry e7 = entry(c1.w), e6 = entry(c1.z), e5 = entry(c1.y), e4 = entry(c1.x);
                const EncEntry e3 = entry(c0.w), e2 = entry(c0.z), e1 = entry(c0.y), e0 = entry(c0.x);
                L.template step<FAST>(e7, P); L.template step<FAST>(e6, P); L.template step<FAST>(e5, P); L.template step<FAST>(e4, P);
                L.template step<FAST>(e3, P); L.template step<FAST>(e2, P); L.template step<FAST>(e1, P); L.template step<FAST>(e0, P);
            }
            if (jumps && --to_jump == 0) { note_here(); to_jump = groups_per_chunk; }
        }
    }
    store_jump_point();
    uint32_t n_words = 0;
    int32_t status = L.finish(true, nsym, n_words);
    if (too_long) status = CST_STREAM_CAPACITY;
    if (!active) return;
    a.n_words_out[s] = status == CST_STREAM_OK ? n_words : 0u;
    a.status[s] = status;
}

// what both decoding kernels share: the tables (staged or not), the lane's coder, its window primed
// The decoders' memory point in two halves with a FIXED shape (the idea of RingReader::advance_window_fixed): all kRaggedChunks
// chunk slots land unconditionally -- a slot without a request lands in the lane's dump rows -- so that the compiler knows
// nothing is in flight into pend[] when the next requests overwrite it.  With the conditional landing of advance_window() it
// protected every pend[] register with a vmcnt(0) right in front of its load, i.e. it waited for the symbol stores and the
// previous chunk request just issued: a memory round trip per requested chunk, 720 cycles per step on a lone wave.
constexpr int kRaggedChunks = 3;                 // a group consumes at most 8 words: two chunks and a ragged one
template <class R>
__device__ __forceinline__ void ring_land_all(R& in, uint32_t* dump) {
#pragma unroll
    for (int k = 0; k < kRaggedChunks; ++k) {
        uint32_t* b = in.pend_pos[k] >= 0 ? in.slot((uint32_t)in.pend_pos[k]) : dump;
        b[0] = in.pend[k].x; b[kWave] = in.pend[k].y; b[2 * kWave] = in.pend[k].z; b[3 * kWave] = in.pend[k].w;
    }
}
template <int AHEAD, class R>
__device__ __forceinline__ void ring_request(R& in) {
    const uint32_t top = in.rd + in.shift;
    const uint32_t want_lo = top > (uint32_t)AHEAD ? top - AHEAD : 0u;
#pragma unroll
    for (int k = 0; k < kRaggedChunks; ++k) {
        if (in.lo_issued > want_lo) {
            in.lo_issued -= 4;
            in.pend_pos[k] = (int32_t)in.lo_issued;
            in.pend[k] = *reinterpret_cast<const uint4*>(in.base16 + in.lo_issued);
        } else {
            in.pend_pos[k] = -1;
        }
    }
}

constexpr int kRaggedDecSlots = 32, kRaggedDecAhead = 24;      // 8 KiB of ring per wave: two workgroups per CU next to 42 KiB of tables
constexpr size_t kRaggedDumpBytes = (size_t)(kBlock / kWave) * 4 * kWave * 4;          // per lane four words nobody reads
constexpr size_t kRaggedDecRingBytes = (size_t)(kBlock / kWave) * kRaggedDecSlots * kWave * 4 + kRaggedDumpBytes;
template <int W, int S, bool STAGED, bool FAST>
struct RaggedDecoder {
    DecLut lut{};
    const uint32_t* cdf;
    const uint16_t* bucket;
    DecLane<W, S, kRaggedDecSlots, kRaggedDecAhead> L;
    WordSlice ws;
    bool active;
    size_t s, slot;
    int lane;
    uint32_t* dump;

    // every thread of the workgroup: the tables (the only barriers)
    __device__ __forceinline__ void stage(const RaggedArgs& a, unsigned char* smem) {
        cdf = a.cdf; bucket = a.bucket;
        if constexpr (STAGED) {
            stage_decoder_tables<kDecBucket, true, true>(smem + kRaggedDecRingBytes, a.precision, nullptr, nullptr, a.cdf, a.bucket, a.bucket_bits,
                                                         a.n_symbols, lut, cdf, bucket);
            __syncthreads();
        }
        lane = threadIdx.x & (kWave - 1);
        slot = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        s = ragged_stream(a, slot, active);
    }
    __device__ __forceinline__ bool wave_has_streams(const RaggedArgs& a) const { return slot - lane < a.n_streams; }
    // the lane's coder on its slice of the words, window primed
    __device__ __forceinline__ void start(const RaggedArgs& a, uint32_t* ring) {
        ws = active ? word_slice(a.word_offsets, a.stride_words, a.n_words_in, s, a.words_capacity) : WordSlice{0, 0u, false};
        L.init(a.words_in + ws.off, ws.n, ring, lane);
        if (a.state_in) L.state = (active && !ws.bad) ? a.state_in[s] : 0;      // AnsCoder::seek(pos, state): the words in front of the jump point
        else L.read_initial_state();
        L.in.prime();
        wave_lds_fence();
    }
    __device__ __forceinline__ void land() { ring_land_all(L.in, dump); }
    __device__ __forceinline__ void request() { ring_request<kRaggedDecAhead>(L.in); }
    __device__ __forceinline__ uint32_t step(const RaggedArgs& a) {
        return ans_decode_step<W, S, kDecBucket, FAST>(L, lut, cdf, bucket, a.precision - a.bucket_bits, a.n_symbols, a.precision);
    }
};

template <int W, int S, bool STAGED, bool FAST>
__global__ __launch_bounds__(kBlock) void ans_decode_ragged_kernel(const RaggedArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    static_assert(2 * kRaggedGroup <= kRaggedDecAhead - 4, "a group may consume kRaggedGroup words before the chunks requested at its start land");
    RaggedDecoder<W, S, STAGED, FAST> D;
    D.stage(a, smem);
    if (!D.wave_has_streams(a)) return;
    D.start(a, reinterpret_cast<uint32_t*>(smem) + (threadIdx.x >> 6) * (kRaggedDecSlots * kWave));
    D.dump = reinterpret_cast<uint32_t*>(smem + kRaggedDecRingBytes - kRaggedDumpBytes) + (threadIdx.x >> 6) * (4 * kWave) + D.lane;
    const uint64_t sym_lo = D.active ? a.sym_offsets[D.s] : 0, sym_hi = D.active ? a.sym_offsets[D.s + 1] : 0;
    const bool too_long = sym_hi - sym_lo > 0xffffffffull || sym_hi < sym_lo;
    const uint32_t len = too_long ? 0u : (uint32_t)(sym_hi - sym_lo);
    int32_t* row = a.symbols_out + sym_lo;
    const uint32_t mx = wave_max_u32(len);
    int32_t o[kRaggedGroup];
#pragma unroll
    for (int j = 0; j < kRaggedGroup; ++j) o[j] = 0;
    // symbols k0 - 8 .. k0 - 1 (decoded by the previous group) -> HBM: two 16-byte pieces, or one by one at the end of a row
    auto store_group = [&](uint32_t k0) {
        if (k0 < kRaggedGroup || k0 - kRaggedGroup >= len) return;
        const uint32_t b = k0 - kRaggedGroup;
        if (k0 <= len) {
            rv4i_unaligned* d = reinterpret_cast<rv4i_unaligned*>(row + b);
            d[0].v = rv4i{o[0], o[1], o[2], o[3]};
            d[1].v = rv4i{o[4], o[5], o[6], o[7]};
        } else {
#pragma unroll
            for (int j = 0; j < kRaggedGroup; ++j)
                if (b + (uint32_t)j < len) row[b + (uint32_t)j] = o[j];
        }
    };
    for (uint32_t k0 = 0; k0 < mx; k0 += kRaggedGroup) {
        D.land();                           // the chunks requested a group ago (a group consumes at most 8 of the >= 12 words landed below it)
        store_group(k0);
        D.request();
#pragma unroll
        for (int j = 0; j < kRaggedGroup; ++j)
            if (k0 + (uint32_t)j < len) o[j] = a.min_symbol + (int32_t)D.step(a);
    }
    store_group((mx + kRaggedGroup - 1) / kRaggedGroup * kRaggedGroup);
    if (!D.active) return;
    a.status[D.s] = D.ws.bad ? (int32_t)CST_STREAM_INVALID_DATA : (too_long ? (int32_t)CST_STREAM_CAPACITY : D.L.status);
}

// The reference's index stores no lengths: a document ends where its terminator symbol is decoded (tests/issue52.rs:63-80).
// First pass of that: every stream is decoded until `eof_index` appears (or `max_symbols` were decoded: CST_STREAM_CAPACITY,
// the output a caller would size from it could not hold more), nothing is stored but the count -- terminator included.  A
// prefix sum of the counts is the d_sym_offsets of cst_ans_decode_ragged.
template <int W, int S, bool STAGED, bool FAST>
__global__ __launch_bounds__(kBlock) void ans_count_until_kernel(const RaggedArgs a, uint32_t eof_index, uint64_t max_symbols, uint64_t* lengths) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    RaggedDecoder<W, S, STAGED, FAST> D;
    D.stage(a, smem);
    if (!D.wave_has_streams(a)) return;
    D.start(a, reinterpret_cast<uint32_t*>(smem) + (threadIdx.x >> 6) * (kRaggedDecSlots * kWave));
    D.dump = reinterpret_cast<uint32_t*>(smem + kRaggedDecRingBytes - kRaggedDumpBytes) + (threadIdx.x >> 6) * (4 * kWave) + D.lane;
    uint64_t n = 0;
    bool done = !D.active || D.ws.bad || D.L.status != CST_STREAM_OK || max_symbols == 0;
    bool found = false;
    while (__any(!done)) {
        D.land();                           // what the previous group requested
        D.request();
#pragma unroll
        for (int j = 0; j < kRaggedGroup; ++j) {
            if (!done) {
                const uint32_t idx = D.step(a);
                ++n;
                found = idx == eof_index;
                done = found || n >= max_symbols;
            }
        }
    }
    if (!D.active) return;
    lengths[D.s] = n;
    a.status[D.s] = D.ws.bad ? (int32_t)CST_STREAM_INVALID_DATA
                             : (D.L.status != CST_STREAM_OK ? D.L.status : (found ? (int32_t)CST_STREAM_OK : (int32_t)CST_STREAM_CAPACITY));
}

static size_t ragged_encode_table_bytes(const cst_model* m) { return (((size_t)m->n_symbols * sizeof(EncEntry)) + 15) & ~(size_t)15; }
static size_t ragged_decode_table_bytes(const cst_model* m) {
    const size_t cdf = (((size_t)m->n_symbols + 1) * 4 + 15) & ~(size_t)15;
    return cdf + (bucket16_usable(m->n_symbols, m->precision) ? ((size_t)16 << m->bucket_bits) + kSubAreaBytes
                                                               : ((((size_t)2 << m->bucket_bits) + 15) & ~(size_t)15));
}

// largest dynamic LDS allocation of a workgroup on the current device (cached per process: the library targets one kind of GPU)
static size_t device_lds_limit() {
    static const size_t limit = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0)
            return (size_t)64 * 1024;
        return (size_t)v;
    }();
    return limit;
}

template <typename K, typename... Extra>
static cst_status ragged_launch(K kernel, const RaggedArgs& a, size_t ring_bytes, size_t table_bytes, hipStream_t hs, Extra... extra) {
    const size_t blocks = (a.n_streams + kBlock - 1) / kBlock;
    if (blocks == 0) return CST_OK;
    if (blocks > 0x7fffffffull) return CST_ERR_INVALID_ARGUMENT;
    const size_t lds = ring_bytes + table_bytes;
    if (lds > 64 * 1024)
        CST_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kBlock), lds, hs, a, extra...);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

// (W, S) by the preset, STAGED by the size of the tables, FAST where the hand-scheduled steps apply
#define CST_RAGGED_DISPATCH(KERNEL, RING_BYTES, TABLE_BYTES, ...)                                                                   \
    do {                                                                                                                             \
        const size_t tb_ = (TABLE_BYTES);                                                                                            \
        /* the tables are staged if they fit beside the rings in what THIS device gives a workgroup (gfx950: 160 KiB) */             \
        const bool staged_ = tb_ <= kRaggedStageLimit && (RING_BYTES) + tb_ <= device_lds_limit();                                   \
        const size_t t_ = staged_ ? tb_ : 0;                                                                                         \
        if (cfg.word_bits != 32)                                                                                                     \
            return staged_ ? ragged_launch(KERNEL<16, 32, true, false>, a, RING_BYTES, t_, hs, ##__VA_ARGS__)                        \
                           : ragged_launch(KERNEL<16, 32, false, false>, a, RING_BYTES, t_, hs, ##__VA_ARGS__);                      \
        if (model->precision >= 8)                                                                                                   \
            return staged_ ? ragged_launch(KERNEL<32, 64, true, true>, a, RING_BYTES, t_, hs, ##__VA_ARGS__)                         \
                           : ragged_launch(KERNEL<32, 64, false, true>, a, RING_BYTES, t_, hs, ##__VA_ARGS__);                       \
        return staged_ ? ragged_launch(KERNEL<32, 64, true, false>, a, RING_BYTES, t_, hs, ##__VA_ARGS__)                            \
                       : ragged_launch(KERNEL<32, 64, false, false>, a, RING_BYTES, t_, hs, ##__VA_ARGS__);                          \
    } while (0)

// the encoder's kernels by symbols per memory point (G): 16 by default, 8 where a jump interval is not a multiple of 16
// (CST_RAGGED_GROUP=8|16|32 forces one that divides the interval)
#define CST_RAGGED_ENC_KERNEL(G)                                                                                                      \
    do {                                                                                                                             \
        const size_t tb_ = ragged_encode_table_bytes(model);                                                                         \
        const bool staged_ = tb_ <= kRaggedStageLimit && kRaggedRingBytes + tb_ <= device_lds_limit();                               \
        const size_t t_ = staged_ ? tb_ : 0;                                                                                         \
        if (cfg.word_bits != 32)                                                                                                     \
            return staged_ ? ragged_launch(ans_encode_ragged_kernel<16, 32, true, false, G>, a, kRaggedRingBytes, t_, hs)            \
                           : ragged_launch(ans_encode_ragged_kernel<16, 32, false, false, G>, a, kRaggedRingBytes, t_, hs);          \
        if (model->precision >= 8)                                                                                                   \
            return staged_ ? ragged_launch(ans_encode_ragged_kernel<32, 64, true, true, G>, a, kRaggedRingBytes, t_, hs)             \
                           : ragged_launch(ans_encode_ragged_kernel<32, 64, false, true, G>, a, kRaggedRingBytes, t_, hs);           \
        return staged_ ? ragged_launch(ans_encode_ragged_kernel<32, 64, true, false, G>, a, kRaggedRingBytes, t_, hs)                \
                       : ragged_launch(ans_encode_ragged_kernel<32, 64, false, false, G>, a, kRaggedRingBytes, t_, hs);              \
    } while (0)

static cst_status encode_ragged_dispatch(const cst_model* model, cst_coder_config cfg, const RaggedArgs& a, hipStream_t hs) {
    int g = knobs().ragged_group;
    if (a.jump_interval != 0 && a.jump_interval % (uint32_t)g != 0) g = a.jump_interval % 16 == 0 ? 16 : 8;
    if (g == 32) CST_RAGGED_ENC_KERNEL(32);
    if (g == 16) CST_RAGGED_ENC_KERNEL(16);
    CST_RAGGED_ENC_KERNEL(8);
}

cst_status ans_encode_ragged(const cst_model* model, cst_coder_config cfg, const int32_t* d_symbols, const uint64_t* d_sym_offsets,
                             size_t n_streams, uint32_t* d_words, const uint64_t* d_word_offsets, size_t stride_words,
                             uint32_t* d_n_words, int32_t* d_status, const uint32_t* d_order, hipStream_t hs) {
    RaggedArgs a{};
    a.order = d_order;
    a.symbols_in = d_symbols; a.sym_offsets = d_sym_offsets; a.n_streams = n_streams; a.enc = model->d_enc;
    a.n_symbols = model->n_symbols; a.min_symbol = model->min_symbol; a.precision = model->precision;
    a.words_out = d_words; a.word_offsets = d_word_offsets; a.stride_words = stride_words; a.n_words_out = d_n_words; a.status = d_status;
    return encode_ragged_dispatch(model, cfg, a, hs);
}

static RaggedArgs ragged_decode_args(const cst_model* model, const uint32_t* d_words, const uint64_t* d_word_offsets, size_t stride_words,
                                     size_t words_capacity, const uint32_t* d_n_words, size_t n_streams, int32_t* d_status, const uint32_t* d_order) {
    RaggedArgs a{};
    a.order = d_order;
    a.n_streams = n_streams; a.cdf = model->d_cdf; a.bucket = model->d_bucket;
    a.bucket_bits = model->bucket_bits; a.n_symbols = model->n_symbols; a.min_symbol = model->min_symbol; a.precision = model->precision;
    a.words_in = d_words; a.word_offsets = d_word_offsets; a.stride_words = stride_words; a.n_words_in = d_n_words; a.status = d_status;
    a.words_capacity = words_capacity;
    return a;
}

cst_status ans_decode_ragged(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_word_offsets,
                             size_t stride_words, size_t words_capacity, const uint32_t* d_n_words, int32_t* d_symbols,
                             const uint64_t* d_sym_offsets, size_t n_streams, int32_t* d_status, const uint32_t* d_order, hipStream_t hs) {
    RaggedArgs a = ragged_decode_args(model, d_words, d_word_offsets, stride_words, words_capacity, d_n_words, n_streams, d_status, d_order);
    a.symbols_out = d_symbols; a.sym_offsets = d_sym_offsets;
    CST_RAGGED_DISPATCH(ans_decode_ragged_kernel, kRaggedDecRingBytes, ragged_decode_table_bytes(model));
}

cst_status ans_count_until(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_word_offsets,
                           size_t stride_words, size_t words_capacity, const uint32_t* d_n_words, size_t n_streams, int32_t eof_symbol,
                           size_t max_symbols, uint64_t* d_lengths, int32_t* d_status, const uint32_t* d_order, hipStream_t hs) {
    const RaggedArgs a = ragged_decode_args(model, d_words, d_word_offsets, stride_words, words_capacity, d_n_words, n_streams, d_status, d_order);
    const uint32_t eof_index = (uint32_t)eof_symbol - (uint32_t)model->min_symbol;
    const uint64_t mx = (uint64_t)max_symbols;
    CST_RAGGED_DISPATCH(ans_count_until_kernel, kRaggedDecRingBytes, ragged_decode_table_bytes(model), eof_index, mx, d_lengths);
}

// ---- jump points (round 6) ----
// the chunks of a batch as streams of their own: chunk c = jump_chunk_offsets[s] + j of stream s decodes symbols
// [sym_offsets[s] + j I, min(.. + I, sym_offsets[s + 1])) from stream s's words in front of its jump point
__global__ void ragged_jump_virtual_kernel(const uint64_t* __restrict__ sym_offsets, const uint64_t* __restrict__ word_offsets, size_t stride_words,
                                           const uint64_t* __restrict__ chunk_offsets, size_t n_streams, size_t n_chunks_total, uint32_t interval,
                                           uint64_t* __restrict__ v_sym_offsets, uint64_t* __restrict__ v_word_offsets) {
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // entries behind the last chunk (n_chunks_total may be an upper bound of the chunks): empty streams at the end of the symbols
    if (s <= n_chunks_total && s >= chunk_offsets[n_streams]) { v_sym_offsets[s] = sym_offsets[n_streams]; if (s < n_chunks_total) v_word_offsets[s] = 0; }
    if (s >= n_streams) return;
    const uint64_t lo = sym_offsets[s], hi = sym_offsets[s + 1];
    const uint64_t c0 = chunk_offsets[s], c1 = chunk_offsets[s + 1];
    const uint64_t w = word_offsets ? word_offsets[s] : (uint64_t)s * stride_words;
    for (uint64_t c = c0; c < c1 && c < n_chunks_total; ++c) {
        const uint64_t start = lo + (c - c0) * interval;
        v_sym_offsets[c] = start < hi ? start : hi;              // (a table with too many chunks for its stream: empty ones)
        v_word_offsets[c] = w;
    }
}

// a stream's status: the worst of its chunks'; a table that does not describe the stream (chunks != ceil(len / I)), or a jump point
// with more words than the stream has, is caller data gone wrong: CST_STREAM_INVALID_DATA
__global__ void ragged_jump_status_kernel(const int32_t* __restrict__ chunk_status, const uint64_t* __restrict__ sym_offsets,
                                          const uint64_t* __restrict__ chunk_offsets, const uint32_t* __restrict__ jump_pos,
                                          const uint32_t* __restrict__ n_words, size_t n_streams, size_t n_chunks_total, uint32_t interval,
                                          int32_t* __restrict__ status) {
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    const uint64_t len = sym_offsets[s + 1] - sym_offsets[s];
    const uint64_t c0 = chunk_offsets[s], c1 = chunk_offsets[s + 1];
    int32_t worst = CST_STREAM_OK;
    if (c1 < c0 || c1 > n_chunks_total || c1 - c0 != (len + interval - 1) / interval) worst = CST_STREAM_INVALID_DATA;
    else
        for (uint64_t c = c0; c < c1; ++c) {
            worst = max(worst, chunk_status[c]);
            if (jump_pos[c] > n_words[s]) worst = CST_STREAM_INVALID_DATA;
        }
    status[s] = worst;
}

cst_status ans_encode_ragged_jump(const cst_model* model, cst_coder_config cfg, const int32_t* d_symbols, const uint64_t* d_sym_offsets,
                                  size_t n_streams, uint32_t* d_words, const uint64_t* d_word_offsets, size_t stride_words, uint32_t* d_n_words,
                                  int32_t* d_status, const uint32_t* d_order, uint32_t interval, const uint64_t* d_chunk_offsets,
                                  uint32_t* d_jump_pos, uint64_t* d_jump_state, hipStream_t hs) {
    RaggedArgs a{};
    a.order = d_order;
    a.symbols_in = d_symbols; a.sym_offsets = d_sym_offsets; a.n_streams = n_streams; a.enc = model->d_enc;
    a.n_symbols = model->n_symbols; a.min_symbol = model->min_symbol; a.precision = model->precision;
    a.words_out = d_words; a.word_offsets = d_word_offsets; a.stride_words = stride_words; a.n_words_out = d_n_words; a.status = d_status;
    a.jump_interval = interval; a.jump_chunk_offsets = d_chunk_offsets; a.jump_pos = d_jump_pos; a.jump_state = d_jump_state;
    return encode_ragged_dispatch(model, cfg, a, hs);
}

static cst_status decode_ragged_virtual(const cst_model* model, cst_coder_config cfg, const RaggedArgs& a, hipStream_t hs) {
    CST_RAGGED_DISPATCH(ans_decode_ragged_kernel, kRaggedDecRingBytes, ragged_decode_table_bytes(model));
}

cst_status ans_decode_ragged_jump(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_word_offsets,
                                  size_t stride_words, size_t words_capacity, const uint32_t* d_n_words, int32_t* d_symbols,
                                  const uint64_t* d_sym_offsets, size_t n_streams, uint32_t interval, const uint64_t* d_chunk_offsets,
                                  size_t n_chunks_total, const uint32_t* d_jump_pos, const uint64_t* d_jump_state, void* d_scratch,
                                  int32_t* d_status, hipStream_t hs) {
    unsigned char* b = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(d_scratch) + 15) & ~(uintptr_t)15);
    uint64_t* v_sym = reinterpret_cast<uint64_t*>(b);
    uint64_t* v_word = v_sym + n_chunks_total + 2;
    int32_t* v_status = reinterpret_cast<int32_t*>(v_word + n_chunks_total + 1);
    const unsigned grid = (unsigned)((n_streams + 255) / 256);
    const size_t v_threads = n_streams > n_chunks_total + 1 ? n_streams : n_chunks_total + 1;
    hipLaunchKernelGGL(ragged_jump_virtual_kernel, dim3((unsigned)((v_threads + 255) / 256)), dim3(256), 0, hs, d_sym_offsets, d_word_offsets, stride_words, d_chunk_offsets, n_streams,
                       n_chunks_total, interval, v_sym, v_word);
    CST_HIP_TRY(hipGetLastError());
    if (n_chunks_total > 0) {
        // every chunk's slice [word offset of its stream, + pos) is checked against the buffer like any stream's (slab form: against all slabs)
        RaggedArgs a = ragged_decode_args(model, d_words, v_word, 0, words_capacity ? words_capacity : (d_word_offsets ? 0 : n_streams * stride_words),
                                          d_jump_pos, n_chunks_total, v_status, nullptr);
        a.symbols_out = d_symbols; a.sym_offsets = v_sym; a.state_in = d_jump_state;
        const cst_status rc = decode_ragged_virtual(model, cfg, a, hs);
        if (rc != CST_OK) return rc;
    }
    hipLaunchKernelGGL(ragged_jump_status_kernel, dim3(grid), dim3(256), 0, hs, v_status, d_sym_offsets, d_chunk_offsets, d_jump_pos, d_n_words, n_streams,
                       n_chunks_total, interval, d_status);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}
#undef CST_RAGGED_DISPATCH

} // namespace cst
