// cst_range_kernels.hpp -- per-lane range coder state machines (shared by cst_range.hip and cst_persymbol.hip).
//
// Recurrences: RangeEncoder::encode_symbol (src/stream/queue.rs:612-705) with its lazy carry
// (EncoderSituation::Inverted, queue.rs:126-142), seal_words (queue.rs:482-523), RangeDecoder::read_point
// (queue.rs:847-868) and decode_symbol (queue.rs:968-1033).  Same data movement as the ANS kernels
// (cst_ans_kernels.hpp): tables in LDS, LDS-tiled symbol matrix, per-lane LDS word rings; words are
// written and read front to back (a queue).
#pragma once
#include "cst_ans_kernels.hpp"

namespace cst {

struct RangeEncodeArgs {
    const int32_t* symbols;
    size_t n_streams, n_per_stream;
    const EncEntry* enc;
    int32_t n_symbols, min_symbol, precision;
    uint32_t* words;
    size_t stride_words;
    uint32_t* n_words;
    int32_t* status;
    cst_range_state* rstate;
    uint32_t flags;
};

struct RangeDecodeArgs {
    const uint32_t* words;
    const uint64_t* offsets;
    size_t stride_words;
    const uint32_t* n_words;
    int32_t* symbols;
    size_t n_streams, n_per_stream;
    const uint32_t* dec_cp;
    const uint16_t* dec_idx;
    const uint32_t* cdf;
    const uint16_t* bucket;
    int32_t bucket_bits;
    int32_t n_symbols, min_symbol, precision;
    int32_t* status;
    cst_range_state* rstate;
    uint32_t flags;
    uint64_t words_capacity;  // uint32 slots behind `words` (0 = unknown): see word_slice
    // jump points (RangeEncoder::pos / RangeDecoder::seek, queue.rs:172-196, 900-926), [n_streams][n_chunks]: the sub-lane decoder
    const uint32_t* ckpt_pos;
    const uint64_t* ckpt_lower;
    const uint64_t* ckpt_range;
    size_t interval, n_chunks;
};

struct RangeCkptOut {         // what the checkpointing encoders note in front of every chunk of `interval` symbols
    uint32_t* pos;
    uint64_t* lower;
    uint64_t* range;
    size_t interval, n_chunks;
};

// cst_range_fast.hip: the hand-scheduled (32,64) kernels; `*_usable` says whether a call qualifies
bool range_encode_fast_usable(const RangeEncodeArgs& a, cst_layout layout);
cst_status range_encode_fast(const RangeEncodeArgs& a, cst_layout layout, hipStream_t hs);
bool range_decode_fast_usable(const RangeDecodeArgs& a, cst_layout layout);
cst_status range_decode_fast(const RangeDecodeArgs& a, cst_layout layout, hipStream_t hs);
// ... with jump points: the encoder that notes them on its way, the decoder with k lanes per stream (two waves per SIMD)
bool range_encode_ckpt_fast_usable(const RangeEncodeArgs& a, cst_layout layout);
cst_status range_encode_ckpt_fast(const RangeEncodeArgs& a, const RangeCkptOut& ck, hipStream_t hs);
bool range_decode_sub_usable(const RangeDecodeArgs& a);
cst_status range_decode_sub(const RangeDecodeArgs& a, hipStream_t hs);
// ... over int8 symbol matrices (round 6): `symbols` of the argument structs is then an int8_t pointer in disguise
bool range_encode_n8_usable(const RangeEncodeArgs& a, cst_layout layout, size_t interval);
cst_status range_encode_ckpt_n8(const RangeEncodeArgs& a, const RangeCkptOut& ck, hipStream_t hs);
bool range_decode_sub_n8_usable(const RangeDecodeArgs& a);
cst_status range_decode_sub_n8(const RangeDecodeArgs& a, hipStream_t hs);

// Forward-reading counterpart of RingReader (queue semantics).
template <int SLOTS = kRingSlots, int AHEAD = kAhead>
struct RingReaderFwd {
    uint32_t pos;          // next stream index to read
    uint32_t len;          // words in the stream
    uint32_t shift;
    uint32_t hi_issued;    // positions < hi_issued (multiple of 4) have been requested
    const uint32_t* base16;
    uint32_t* ring;
    int lane;
    uint4 pend[kMaxChunksPerPoint];
    int32_t pend_pos[kMaxChunksPerPoint];

    __device__ __forceinline__ uint32_t* slot(uint32_t p) const { return ring + ((p & (SLOTS - 1)) * kWave + lane); }

    __device__ __forceinline__ void init(const uint32_t* in, uint32_t n, uint32_t* wave_ring, int lane_) {
        // pointer arithmetic (not an integer round trip) so that the accesses stay global_*, not flat_*
        shift = (uint32_t)((reinterpret_cast<uintptr_t>(in) & 15) >> 2);
        base16 = in - shift;
        ring = wave_ring; lane = lane_; pos = 0; len = n;
#pragma unroll
        for (int k = 0; k < kMaxChunksPerPoint; ++k) pend_pos[k] = -1;
    }

    __device__ __forceinline__ void prime() {
        hi_issued = (pos + shift) & ~3u;      // (nothing in front of the read position is ever read: a coder that continues
        fill_blocking();                      //  at a jump point or a raw position does not walk there from word 0)
    }

    // everything the window wants, now (no chunk may be pending)
    __device__ __forceinline__ void fill_blocking() {
        const uint32_t end = len + shift;
        const uint32_t want_hi = min(pos + shift + (uint32_t)AHEAD, end);
        while (hi_issued < want_hi) {
            const uint4 v = *reinterpret_cast<const uint4*>(base16 + hi_issued);
            *slot(hi_issued + 0) = v.x; *slot(hi_issued + 1) = v.y; *slot(hi_issued + 2) = v.z; *slot(hi_issued + 3) = v.w;
            hi_issued += 4;
        }
    }

    __device__ __forceinline__ void advance_window() {
#pragma unroll
        for (int k = 0; k < kMaxChunksPerPoint; ++k) {
            if (pend_pos[k] >= 0) {
                const uint32_t p = (uint32_t)pend_pos[k];
                *slot(p + 0) = pend[k].x; *slot(p + 1) = pend[k].y; *slot(p + 2) = pend[k].z; *slot(p + 3) = pend[k].w;
            }
        }
        const uint32_t end = len + shift;
        const uint32_t want_hi = min(pos + shift + (uint32_t)AHEAD, end);
#pragma unroll
        for (int k = 0; k < kMaxChunksPerPoint; ++k) {
            if (hi_issued < want_hi) {
                pend_pos[k] = (int32_t)hi_issued;
                pend[k] = *reinterpret_cast<const uint4*>(base16 + hi_issued);
                hi_issued += 4;
            } else {
                pend_pos[k] = -1;
            }
        }
    }

    // next word if any (ring must cover it); does not advance
    __device__ __forceinline__ uint32_t peek() const { return *slot(pos + shift); }
    __device__ __forceinline__ uint32_t word_direct(uint32_t i) const { return base16[shift + i]; }
};

template <int W, int S, int SLOTS = kRingSlots>
struct RangeEncLane {
    using st_t = typename StateT<S>::type;
    st_t lower, range;
    uint32_t inv_n, inv_first;   // EncoderSituation: inv_n == 0 <=> Normal
    uint32_t bad;
    RingWriter<SLOTS> out;

    __device__ __forceinline__ void init(uint32_t* slab, uint32_t capacity, uint32_t* wave_ring, int lane_) {
        out.init(slab, capacity, wave_ring, lane_);
        lower = 0; range = (st_t)~(st_t)0;   // RangeCoderState::default, queue.rs:96-104
        inv_n = 0; inv_first = 0; bad = 0;
    }

    // queue.rs:612-705
    __device__ __forceinline__ void step(uint32_t c, uint32_t p, int P) {
        const st_t scale = (st_t)(range >> P);
        const st_t new_range = (st_t)(scale * (st_t)p);
        const st_t new_lower = (st_t)(lower + scale * (st_t)c);
        if (__builtin_expect(inv_n != 0, 0)) {
            if ((st_t)(new_lower + new_range) > new_lower) {   // inverted -> normal
                uint32_t first, cons;
                if (new_lower < lower) { first = (inv_first + 1u) & word_mask<W>(); cons = 0u; }
                else { first = inv_first; cons = word_mask<W>(); }
                out.push_slow(first);
                for (uint32_t i = 1; i < inv_n; ++i) out.push_slow(cons);
                inv_n = 0;
            }
        }
        lower = new_lower; range = new_range;
        const bool renorm = range < ((st_t)1 << (S - W));
        const uint32_t lower_word = (uint32_t)(lower >> (S - W)) & word_mask<W>();
        const st_t sh_lower = (st_t)(lower << (W % S)), sh_range = (st_t)(range << (W % S));
        const bool no_wrap = (st_t)(sh_lower + sh_range) > sh_lower;
        // common case: Normal -> Normal (emit lower_word) -- branch free through the ring
        out.push(lower_word, (renorm && inv_n == 0 && no_wrap) ? 1u : 0u);
        if (__builtin_expect(renorm && (inv_n != 0 || !no_wrap), 0)) {
            if (inv_n != 0) inv_n += 1;                    // inverted -> inverted
            else { inv_n = 1; inv_first = lower_word; }    // normal -> inverted
        }
        lower = renorm ? sh_lower : lower;
        range = renorm ? sh_range : range;
    }

    // The same step, branch free, for every case except one: leaving an Inverted run of TWO OR MORE held-back words
    // (then `slow` is set and the caller repeats the quad with step()).  An Inverted situation as such is common
    // (the interval straddles a word boundary with probability range / 2^32 at a renormalisation); a run longer than
    // one word needs a second renormalisation inside it and is rare.  Two candidate words per step go to the ring
    // unconditionally -- the resolved first word of a run and the regular word -- and `wr` advances by what was
    // really emitted.  queue.rs:612-705.
    __device__ __forceinline__ void step_inline(uint32_t c, uint32_t p, int P, bool& slow) {
        const st_t scale = (st_t)(range >> P);
        const st_t new_range = (st_t)(scale * (st_t)p);
        const st_t new_lower = (st_t)(lower + scale * (st_t)c);
        // leaving the Inverted situation?
        const bool resolve = inv_n != 0 && (st_t)(new_lower + new_range) > new_lower;
        slow |= resolve && inv_n >= 2u;
        const uint32_t first = (inv_first + (new_lower < lower ? 1u : 0u)) & word_mask<W>();
        out.push(first, resolve ? 1u : 0u);
        const uint32_t n_inv = resolve ? 0u : inv_n;
        // renormalisation
        const bool renorm = new_range < ((st_t)1 << (S - W));
        const uint32_t lower_word = (uint32_t)(new_lower >> (S - W)) & word_mask<W>();
        const st_t sh_lower = (st_t)(new_lower << (W % S)), sh_range = (st_t)(new_range << (W % S));
        const bool no_wrap = (st_t)(sh_lower + sh_range) > sh_lower;
        out.push(lower_word, (renorm && n_inv == 0 && no_wrap) ? 1u : 0u);
        const bool enter = renorm && n_inv == 0 && !no_wrap;
        inv_first = enter ? lower_word : inv_first;
        inv_n = enter ? 1u : ((renorm && n_inv != 0) ? n_inv + 1u : n_inv);
        lower = renorm ? sh_lower : new_lower;
        range = renorm ? sh_range : new_range;
    }

    // seal_words / iter_seal (queue.rs:458-523)
    __device__ __forceinline__ int32_t finish(uint32_t n_symbols, uint32_t& n_words_out) {
        out.drain();
        if (range != (st_t)~(st_t)0) {
            const st_t point = (st_t)(lower + (((st_t)1 << (S - W)) - 1));
            if (inv_n != 0) {
                uint32_t first, cons;
                if (point >= lower) { first = inv_first; cons = word_mask<W>(); }
                else { first = (inv_first + 1u) & word_mask<W>(); cons = 0u; }
                out.append_direct(first);
                for (uint32_t i = 1; i < inv_n; ++i) out.append_direct(cons);
            }
            const uint32_t point_word = (uint32_t)(point >> (S - W)) & word_mask<W>();
            const uint32_t upper_word = (uint32_t)((st_t)(lower + range) >> (S - W)) & word_mask<W>();
            out.append_direct(point_word);
            if (upper_word == point_word) out.append_direct(0u);
        }
        n_words_out = out.wr;
        if (bad >= n_symbols) return CST_STREAM_IMPOSSIBLE_SYMBOL;
        if (out.wr > out.cap) return CST_STREAM_CAPACITY;
        return CST_STREAM_OK;
    }
};

template <int W, int S, int SLOTS = kRingSlots, int AHEAD = kAhead>
struct RangeDecLane {
    using st_t = typename StateT<S>::type;
    st_t lower, range, point;
    int32_t status;
    RingReaderFwd<SLOTS, AHEAD> in;

    // from_compressed + read_point (queue.rs:776-790, 847-868)
    __device__ __forceinline__ void init(const uint32_t* words, uint32_t len, uint32_t* wave_ring, int lane_) {
        in.init(words, len, wave_ring, lane_);
        lower = 0; range = (st_t)~(st_t)0; status = CST_STREAM_OK;
        st_t pt = 0;
        int num_read = 0;
        while (in.pos < in.len) {
            pt = (st_t)((pt << (W % S)) | (st_t)in.word_direct(in.pos++));
            if (++num_read == S / W) break;
        }
        if (num_read < S / W && num_read != 0) pt = (st_t)(pt << (S - num_read * W));
        point = pt;
    }

    // RangeDecoder::seek((pos, (lower, range))), queue.rs:911-926: continue reading at word `pos0`, read_point, take the state
    __device__ __forceinline__ void init_at(const uint32_t* words, uint32_t len, uint32_t pos0, st_t lower0, st_t range0,
                                            uint32_t* wave_ring, int lane_) {
        in.init(words, len, wave_ring, lane_);
        in.pos = pos0 < len ? pos0 : len;
        lower = lower0; range = range0; status = CST_STREAM_OK;
        st_t pt = 0;
        int num_read = 0;
        while (in.pos < in.len) {
            pt = (st_t)((pt << (W % S)) | (st_t)in.word_direct(in.pos++));
            if (++num_read == S / W) break;
        }
        if (num_read < S / W && num_read != 0) pt = (st_t)(pt << (S - num_read * W));
        point = pt;
    }

    // queue.rs:984-993: quantile = (point - lower) / (range >> P); flags InvalidData and clamps if it is >= 2^P
    __device__ __forceinline__ uint32_t peek_quantile(int P) {
        const st_t scale = (st_t)(range >> P);
        const st_t x = (st_t)(point - lower);
        // the quotient is < 2^(P+1) <= 2^25: estimate it as x * (1 / scale) in f64, then make it exact.  v_rcp_f64 alone
        // is good to 2^-24.4 only (scripts/microbench/rcp_f64_error.hip), which at P = 24 could leave the estimate two
        // off; one Newton step (two fma) puts it within one for every P.  (An IEEE division costs ~15 instructions.)
        const double sd = (double)scale;
        double r = __builtin_amdgcn_rcp(sd);
        r = __builtin_fma(__builtin_fma(-sd, r, 1.0), r, r);
        uint32_t q = (uint32_t)((double)x * r);
        const st_t prod = (st_t)((st_t)q * scale);
        if (prod > x) --q;
        else if ((st_t)(x - prod) >= scale) ++q;
        if (q >= (1u << P)) {                      // DecoderFrontendError::InvalidData
            if (status == CST_STREAM_OK) status = CST_STREAM_INVALID_DATA;
            q = (1u << P) - 1u;                    // keep the lane on legal table indices; its output is unspecified
        }
        return q;
    }

    // queue.rs:998-1030 given the model's answer; `next_word` is the word at the read position (if `have`).
    // Returns true if a word was consumed.
    __device__ __forceinline__ bool advance(uint32_t c, uint32_t p, int P, uint32_t next_word, bool have) {
        const st_t scale = (st_t)(range >> P);
        lower = (st_t)(lower + scale * (st_t)c);
        range = (st_t)(scale * (st_t)p);
        const bool renorm = range < ((st_t)1 << (S - W));
        const st_t sh_point = (st_t)((st_t)(point << (W % S)) | (st_t)(have ? next_word : 0u));
        lower = renorm ? (st_t)(lower << (W % S)) : lower;
        range = renorm ? (st_t)(range << (W % S)) : range;
        point = renorm ? sh_point : point;
        return renorm && have;
    }

    // queue.rs:968-1033 with a tabulated model; returns the symbol index
    template <int MODE>
    __device__ __forceinline__ uint32_t step(const DecLut lut, const uint32_t* cdf, const uint16_t* bucket, int bucket_shift,
                                             int n_symbols, int P) {
        const uint32_t q = peek_quantile(P);
        const uint32_t next_word = in.peek();
        uint32_t idx, c, p;
        lookup_quantile<MODE>(q, lut, cdf, bucket, bucket_shift, n_symbols, idx, c, p);
        in.pos += advance(c, p, P, next_word, in.pos < in.len) ? 1u : 0u;
        return idx;
    }
};

} // namespace cst
